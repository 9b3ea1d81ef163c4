"""Round 6 gate: the low-occupancy mover (sfgwas_amd/csrc/i8_move.hpp) against the transposition pass, alone and in front of the plaintext NTT's workgroups.
One MAC launch's worth (G block rows).  Prints ms; `check` lines must say 0 differing words."""
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
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"


def run(mode, nblocks=256, depth=3, nt=0, reps=3):
    ms = C.c_double()
    ctx.check(f(ctx.h, mode, G, nblocks, depth, nt, reps, C.byref(ms)), "ubench_ntt_move")
    return ms.value


print(f"check G=2 mover alone vs pass: {run(6, 256, 3, 0):.0f} differing words", flush=True) if False else None
for depth in (1, 2, 3):
    for nt in (0, 1):
        print(f"check G={G} mover alone depth {depth} nt {nt}: {run(6, 256, depth, nt):.0f} differing words", flush=True)
print(f"check G={G} mover riding in NTT launches, depth 3: {run(7, 256, 3, 0):.0f} differing words", flush=True)
print(f"check G={G} mover riding, 512 blocks depth 2 nt: {run(7, 512, 2, 1):.0f} differing words", flush=True)
base_ntt = run(2); base_pack = run(3); base_seq = run(0)
print(f"G={G} NTTs alone {base_ntt:.2f} ms; pass alone {base_pack:.2f} ms; one after the other {base_seq:.2f} ms", flush=True)
for nb in (256, 512, 768, 1024, 1280):
    for depth in (2, 3):
        for nt in (0, 1):
            print(f"G={G} mover alone: {nb} workgroups depth {depth} nt {nt}: {run(4, nb, depth, nt):.2f} ms", flush=True)
    if quick:
        break
for nb in (256, 512, 768):
    for depth in (1, 2, 3):
        for nt in (0, 1):
            t = run(5, nb, depth, nt)
            print(f"G={G} mover in front of the NTT launches: {nb} workgroups depth {depth} nt {nt}: {t:.2f} ms  ({t / base_seq:.3f} of sequential; gate 0.80)", flush=True)
