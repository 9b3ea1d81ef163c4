"""HBM roofline of the Beaver element-wise product (B2, mpc/beavermult.go:112-133) at the QC vector sizes of BASELINE config 4 (10^6 .. 10^7 field
elements, qualcontrol.go:90,208): 4 operand vectors read + 1 written = 80 B per 128-bit element, 160 B per 256-bit element.  Device-resident operands
(uniform limbs below a Mersenne-like odd modulus); prints one JSON line per (limbs, pid, n)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfgwas_amd import capi, params as P

ctx = capi.Context(P.Q_PN14, P.P_PN14)
L = capi.lib()
hip = C.CDLL("libamdhip64.so")
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
for limbs, modulus in ((2, (1 << 127) - 1), (4, (1 << 255) - 19)):
    mod = np.array([(modulus >> (64 * k)) & ((1 << 64) - 1) for k in range(limbs)], dtype=np.uint64)
    for n in (10 ** 6, 10 ** 7):
        nbytes = n * limbs * 8
        bufs = [ctx.malloc(nbytes) for _ in range(5)]
        for k, b in enumerate(bufs[:4]):
            hip.hipMemset(b, 0x11 * (k + 1) & 0x7F, nbytes)              # every limb 0x1111.. / 0x2222.. : below the modulus
        for pid in (0, 1, 2):
            for _ in range(2):
                ctx.check(L.sfg_beaver_elem_dev(ctx.h, pid, limbs, capi.p64(mod), bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], n), "beaver"); ctx.sync()
            reps = 20
            t = time.perf_counter()
            for _ in range(reps):
                ctx.check(L.sfg_beaver_elem_dev(ctx.h, pid, limbs, capi.p64(mod), bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], n), "beaver")
            ctx.sync(); dt = (time.perf_counter() - t) / reps
            by = 5 * nbytes if pid else 3 * nbytes                      # pid 0 touches only the two masks (am * bm)
            print(json.dumps({"kernel": "k_beaver_elem", "limbs": limbs, "pid": pid, "n": n, "us": round(dt * 1e6, 1), "elements_per_s": n / dt,
                              "algorithmic_GBps": by / dt / 1e9, "frac_of_8TBps": by / dt / 8e12}), flush=True)
        for b in bufs:
            ctx.free(b)
