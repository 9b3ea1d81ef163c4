"""Config-5 shaped measurement of the streamed association path (sfg_assoc_stream_bed): one party's 500 000-sample chromosome file, batches of
8192 SNPs, s = 13 (the covariate product of assoc.go:395) - synthetic random 2-bit codes written to local storage first.
Prints one JSON line: seconds per batch, useful ring-MAC/s, file GB/s, and the same batches multiplied from HBM-resident int8 for comparison."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfgwas_amd import capi, params as P

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=500_000)
ap.add_argument("--snps", type=int, default=32768)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--s", type=int, default=13)
ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
ap.add_argument("--quick", action="store_true", help="the default streamed path only (warm-up + two calls): for kernel profiles")
a = ap.parse_args()
bps = (a.samples + 3) // 4
path = os.path.join(a.dir, "sfg_stream_bench.bed")
t0 = time.time()
rnd = np.random.default_rng(5)
with open(path, "wb") as f:
    f.write(bytes([0x6C, 0x1B, 0x01]))
    for j0 in range(0, a.snps, 1024):
        f.write(rnd.integers(0, 256, (min(1024, a.snps - j0), bps), dtype=np.uint8).tobytes())
t_write = time.time() - t0
ctx = capi.Context(P.Q_PN14, P.P_PN14)
L = capi.lib()
rots = P.rotations_for_matmul()
ctx.check(L.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), "keys")
nbr = (a.samples - 1) // P.SLOTS + 1
A = ctx.fill_uniform_cts(a.s * nbr, P.MAX_LEVEL, 0xC1F3)
nbatch = (a.snps + a.batch - 1) // a.batch
cap = nbatch * ((a.batch - 1) // P.SLOTS + 1)
out = capi.DevArray(ctx, (a.s, cap, 2, P.MAX_LEVEL, P.N))
got = C.c_size_t()
def run(flags=0, snps=None):
    t = time.time()
    ctx.check(L.sfg_assoc_stream_bed(ctx.h, (path_half if snps else path).encode(), a.samples, snps or a.snps, None, None, a.batch, A.p, a.s, P.MAX_LEVEL, P.MAX_LEVEL, flags, out.p, cap, C.byref(got), None, None), "stream")
    ctx.sync()
    return time.time() - t
run()                                  # warm-up: scratch pools, page cache
# the first half of the file alone: the call's fixed cost (the rotation cache of `mat`, built once per call) separated from the cost of one more batch
if a.quick:
    dt = min(run(), run())
    print(json.dumps({"workload": f"assoc batches streamed from a .bed: {a.samples} samples x {a.snps} SNPs, batch {a.batch}, s={a.s}", "batches": nbatch, "calls": 3, "s_per_batch_streamed": dt / nbatch}))
    os.remove(path)
    sys.exit(0)
half = (nbatch // 2) * a.batch
path_half = path + ".half"
if 0 < half < a.snps:
    with open(path, "rb") as f, open(path_half, "wb") as g:
        left = 3 + half * bps
        while left:
            buf = f.read(min(left, 1 << 26)); g.write(buf); left -= len(buf)
dt_half = min(run(0, half), run(0, half)) if 0 < half < a.snps else None
dt = min(run(), run())
phases = {k: round(ctx.phase_ms(k), 2) for k in ("rotate", "skew", "encode", "ntt_plain", "mac", "mac_small", "mac_big", "mac_i8_pack_pt", "mac_i8_pack_rot", "mac_i8_untile") if ctx.phase_ms(k) > 0}
try:                                   # the same with O_DIRECT reads: the disk, not the page cache
    dt_direct = min(run(capi.SFG_STREAM_DIRECT), run(capi.SFG_STREAM_DIRECT))
except capi.SfgError as e:
    dt_direct = None; direct_err = str(e)[:120]
if os.environ.get("SFG_ASSOC_ROTCACHE_MB") != "0":      # the per-batch rotation rebuild of round 2, for the A/B: a second context that reads the switch
    os.environ["SFG_ASSOC_ROTCACHE_MB"] = "0"
    ctx0 = capi.Context(P.Q_PN14, P.P_PN14)
    ctx0.check(L.sfg_fill_rotkeys_synthetic(ctx0.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), "keys")
    A0 = ctx0.fill_uniform_cts(a.s * nbr, P.MAX_LEVEL, 0xC1F3); out0 = capi.DevArray(ctx0, (a.s, cap, 2, P.MAX_LEVEL, P.N))
    def run0():
        t = time.time()
        ctx0.check(L.sfg_assoc_stream_bed(ctx0.h, path.encode(), a.samples, a.snps, None, None, a.batch, A0.p, a.s, P.MAX_LEVEL, P.MAX_LEVEL, 0, out0.p, cap, C.byref(got), None, None), "stream")
        ctx0.sync()
        return time.time() - t
    ctx.check(L.sfg_ctx_release_scratch(ctx.h), "release")          # two contexts share the device: the first one's kept pools (rotation cache, panels) make room
    run0(); dt_rebuild = min(run0(), run0())
    h0 = out0.host()
    A0.free(); out0.free(); ctx0.close()
    run()                              # `out` again from the cached path (the O_DIRECT runs wrote the same words)
    same = bool(np.array_equal(h0, out.host()))
else:
    dt_rebuild, same = None, None
# the same products from an HBM-resident int8 batch (no file, no decode): the compute floor
gd, gh = ctx.fill_geno(a.samples, a.batch, 0x5F6A)
o2 = capi.DevArray(ctx, (a.s, (a.batch - 1) // P.SLOTS + 1, 2, P.MAX_LEVEL, P.N))
ctx.check(L.sfg_matmul_resident_dev(ctx.h, A.p, a.s, P.MAX_LEVEL, P.MAX_LEVEL, gh, 0, o2.p), "resident"); ctx.sync()
t = time.time()
for _ in range(nbatch):
    ctx.check(L.sfg_matmul_resident_dev(ctx.h, A.p, a.s, P.MAX_LEVEL, P.MAX_LEVEL, gh, 0, o2.p), "resident")
ctx.sync(); dt_res = time.time() - t
macs = a.samples * a.snps * a.s * 2 * P.MAX_LEVEL * 2
print(json.dumps({"workload": f"assoc batches streamed from a .bed: {a.samples} samples x {a.snps} SNPs, batch {a.batch}, s={a.s}", "batches": nbatch,
                  "s_per_batch_streamed": dt / nbatch,
                  "s_per_batch_marginal": None if dt_half is None else (dt - dt_half) / (nbatch - nbatch // 2), "s_per_call_fixed": None if dt_half is None else dt - nbatch * (dt - dt_half) / (nbatch - nbatch // 2),
                  "last_batch_phases_ms": phases, "s_per_batch_resident_int8": dt_res / nbatch, "useful_ring_macs_per_s_streamed": macs / dt,
                  "file_GBps": (a.snps * bps) / dt / 1e9, "file_bytes": a.snps * bps, "file_write_s": t_write,
                  "s_per_batch_streamed_O_DIRECT": None if dt_direct is None else dt_direct / nbatch,
                  "file_GBps_O_DIRECT": None if dt_direct is None else (a.snps * bps) / dt_direct / 1e9,
                  "s_per_batch_streamed_rotations_rebuilt_per_batch": None if dt_rebuild is None else dt_rebuild / nbatch,
                  "outputs_identical_with_and_without_the_call_wide_rotation_cache": same,
                  "note": "buffered reads of a file that was just written come from the page cache; the O_DIRECT figures are the storage device's; "
                          "the resident figure rebuilds the rotation cache per call (one call = one batch there)"}))
os.remove(path)
if os.path.exists(path_half):
    os.remove(path_half)
