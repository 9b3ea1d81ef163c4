"""Round 6: tune the stand-alone mover (workgroups, depth) and close the co-run question with throttled settings.  See tools/r6_mover_ubench.py."""
import ctypes as C
import os
import sys
os.environ["SFG_ENABLE_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfgwas_amd import capi, params as P          # noqa: E402

ctx = capi.Context(P.Q_PN14, P.P_PN14)
lib = capi.lib()
f = lib.ubench_ntt_move
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 13


def run(mode, nblocks=256, depth=3, nt=0, reps=3):
    ms = C.c_double()
    ctx.check(f(ctx.h, mode, G, nblocks, depth, nt, reps, C.byref(ms)), "ubench_ntt_move")
    return ms.value


base_ntt = run(2); base_pack = run(3); base_seq = run(0)
print(f"G={G} NTTs alone {base_ntt:.2f} ms; pass alone {base_pack:.2f} ms; one after the other {base_seq:.2f} ms", flush=True)
for nb in (1024, 1280, 1536, 2048, 2560, 4096, 8192):
    for depth in (1, 2, 3):
        print(f"G={G} mover alone: {nb} workgroups depth {depth}: {run(4, nb, depth, 0):.2f} ms", flush=True)
for nb in (64, 128, 192):
    for depth in (1, 2, 3):
        t = run(5, nb, depth, 1)
        print(f"G={G} mover in front of the NTT launches: {nb} workgroups depth {depth} nt 1: {t:.2f} ms  ({t / base_seq:.3f} of sequential; gate 0.80)", flush=True)
