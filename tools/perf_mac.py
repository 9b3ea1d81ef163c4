"""Quick MAC-kernel throughput probe (panel shape of one 8192x8192 genotype block, s=15)."""
import ctypes as C, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as ol
from sfgwas_amd import capi

ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
L = capi.lib()
N = ctx.N
for (K, R, Ncols, Lv) in [(91, 30, 91, 5), (91, 30, 32, 5), (91, 30, 64, 5), (182, 30, 91, 5)]:
    rot = ctx.malloc(K * R * Lv * N * 8); pt = ctx.malloc(K * Ncols * Lv * N * 8); out = ctx.malloc(Ncols * R * Lv * N * 8)
    # fill with something non-trivial: reuse memset patterns (values need only be < 2^36)
    import ctypes
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes=[C.c_void_p,C.c_int,C.c_size_t]; hip.hipMemset(rot, 0x01, K * R * Lv * N * 8); hip.hipMemset(pt, 0x02, K * Ncols * Lv * N * 8)
    hip.hipDeviceSynchronize()
    for it in range(3):
        t0 = time.time()
        ctx.check(L.sfg_mac_dev(ctx.h, rot, pt, out, K, R, Ncols, Lv, 0), "mac")
        ctx.sync()
        dt = time.time() - t0
    macs = K * R * Ncols * Lv * N
    byts = (K * R + K * Ncols + Ncols * R) * Lv * N * 8
    print(f"K={K} R={R} Ncols={Ncols} L={Lv}: {dt*1e3:.2f} ms  {macs/dt:.3e} MAC/s  min-bytes {byts/1e9:.2f} GB -> {byts/dt/1e12:.2f} TB/s", flush=True)
    for p in (rot, pt, out): ctx.free(p)
