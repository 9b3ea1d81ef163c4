"""How the number of HIP streams alive in a process changes the kernels of a product on the context's own queue (round 5: profiles/r05_mgpu_queue_count.txt found
a 7 - 100 % slowdown with a fourth library stream).  One configuration per process:  python3 tools/r5_streams.py <extra streams> <mode>
  mode: unused     - the extra streams exist, nothing is ever enqueued on them
        used       - a 4-byte device copy is enqueued on every extra stream between products
        on_first   - the product runs on the FIRST extra stream, the others unused
        waits      - as on_first, and every other extra stream waits on an event of the product's queue and the product's queue waits on it back, between products
        on_extra   - the product runs on the LAST extra stream (sfg_ctx_set_stream)
        before     - as `unused`, the extra streams made BEFORE the context
prints: ms per Q*X product at 10 000 x 100 000 (kp = 15) and the k_skew / FFT / NTT phase times."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfgwas_amd import capi, params as P                      # noqa: E402

n_extra, mode = int(sys.argv[1]), sys.argv[2]
hip = C.CDLL("libamdhip64.so")
streams = []


def make(n):
    for _ in range(n):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0       # hipStreamNonBlocking
        streams.append(s)


if mode == "before":
    hip.hipSetDevice(0)
    make(n_extra)
ctx = capi.Context(P.Q_PN14, P.P_PN14)
lib = capi.lib()
if mode != "before":
    make(n_extra)
rots = P.rotations_for_matmul()
ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), "keys")
KP, LEVEL, L = 15, P.MAX_LEVEL, P.MAX_LEVEL
n_ind, m_snp = 10000, 100000
d, g = ctx.fill_geno(n_ind, m_snp, 0x5F6A)
A = ctx.fill_uniform_cts(KP * ((n_ind - 1) // P.SLOTS + 1), LEVEL, 0xC1F3)
if mode == "on_extra":
    ctx.check(lib.sfg_ctx_set_stream(ctx.h, streams[-1]), "set_stream")
scratch = ctx.malloc(64)
ev = [C.c_void_p() for _ in range(2)]
for e in ev:
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0
own = streams[0] if mode in ("waits", "on_first") else None       # (the context's own queue has no accessor: these modes run the product on the FIRST extra stream)
if own is not None:
    ctx.check(lib.sfg_ctx_set_stream(ctx.h, own), "set_stream")


def between():
    if mode == "used":
        for s in streams:
            hip.hipMemcpyAsync(scratch, C.c_void_p(scratch.value + 32), 4, 3, s)
    elif mode == "waits":
        for s in streams[1:]:
            hip.hipEventRecord(ev[0], own)
            hip.hipStreamWaitEvent(s, ev[0], 0)
            hip.hipMemcpyAsync(scratch, C.c_void_p(scratch.value + 32), 4, 3, s)
            hip.hipEventRecord(ev[1], s)
            hip.hipStreamWaitEvent(own, ev[1], 0)


def product():
    lib.sfg_ctx_clear_phases(ctx.h)
    out = ctx.matmul_resident(A, KP, LEVEL, L, g, 0)
    between()
    ctx.sync()
    return out


for _ in range(3):
    product().free()
t0 = time.perf_counter()
R = 5
ph = {}
for _ in range(R):
    product().free()
    for k in ("skew", "encode", "mac_small", "mac_i8_pack_pt", "rotate"):
        ph[k] = ph.get(k, 0.0) + max(ctx.phase_ms(k), 0.0) / R
dt = (time.perf_counter() - t0) / R
print(f"extra={n_extra} mode={mode} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '-')}: {1e3 * dt:.1f} ms per product  " + " ".join(f"{k}={v:.1f}" for k, v in ph.items()), flush=True)
