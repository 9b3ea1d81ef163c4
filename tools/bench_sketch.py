"""HBM roofline of the count-sketch + moments pass (P1, gwas/pca.go:152-162: S (kp x n) * X with S sparse +-1, column sums and sums of squares;
the one MFMA use the north star allows) and of the column-moment kernel (P2): 1 B per genotype read once."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfgwas_amd import capi, params as P
ctx = capi.Context(P.Q_PN14, P.P_PN14); L = capi.lib()
for nrow, ncol in ((100_000, 100_000), (100_000, 400_000)):
    gd, gh = ctx.fill_geno(nrow, ncol, 0x5F6A)
    kp = 15
    rnd = np.random.default_rng(1)
    bucket = rnd.integers(0, kp, nrow).astype(np.int32); sgn = (rnd.integers(0, 2, nrow) * 2 - 1).astype(np.int8)
    sk = np.zeros((kp, ncol)); xs = np.zeros(ncol, dtype=np.uint64); x2 = np.zeros(ncol, dtype=np.uint64)
    def run():
        t = time.perf_counter()
        ctx.check(L.sfg_sketch(ctx.h, gh, bucket.ctypes.data_as(C.POINTER(C.c_int32)), sgn.ctypes.data_as(C.POINTER(C.c_int8)), kp, sk.ctypes.data_as(C.POINTER(C.c_double)),
                               capi.p64(xs), capi.p64(x2)), "sketch")
        return time.perf_counter() - t
    run(); dt = min(run(), run())
    ms = ctx.phase_ms("sketch")
    s1 = np.zeros(ncol); s2 = np.zeros(ncol)
    t = time.perf_counter(); ctx.check(L.sfg_geno_colsums(ctx.h, gh, s1.ctypes.data_as(C.POINTER(C.c_double)), s2.ctypes.data_as(C.POINTER(C.c_double))), "colsums"); dc = time.perf_counter() - t
    print(json.dumps({"nrow": nrow, "ncol": ncol, "sketch_call_ms": dt * 1e3, "sketch_GBps_incl_host_copies": nrow * ncol / dt / 1e9, "colsums_call_ms": dc * 1e3,
                      "colsums_GBps_incl_host_copies": nrow * ncol / dc / 1e9, "kernel_phase_ms": ms}), flush=True)
    gd.free(); ctx.geno_free(gh)
