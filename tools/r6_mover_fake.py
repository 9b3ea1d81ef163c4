"""Round 6: what binds the riding launch?  tools/r6_mover_ubench.py's harness (one MAC launch's worth, G block rows, A/B build) with the mover's accesses replaced
(SFG_UB_MOVER_FAKE, results INVALID): 1 = reads 32 KiB contiguous per unit instead of 256 rows of 128 B, 2 = writes 32 KiB contiguous instead of 128 pieces of 256 B,
3 = both, 8 = reads in the pattern of a panel laid out [column][plane][128-byte coefficient block][k] (2 KiB runs), 4 | n << 8 = sleeps n x 127 x 64 clocks per unit instead of moving (slot loss alone).  One process per setting (the switch is read per call)."""
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


def run(mode, nblocks=192, depth=1, nt=1, reps=3):
    ms = C.c_double()
    ctx.check(f(ctx.h, mode, G, nblocks, depth, nt, reps, C.byref(ms)), "ubench_ntt_move")
    return ms.value


os.environ.pop("SFG_UB_MOVER_FAKE", None)
base_ntt = run(2); base_pack = run(3); base_seq = run(0)
os.environ["SFG_UB_MOVER_FAKE"] = "16"
print(f"G={G} NTTs alone, writing the K-major pattern: {run(2):.2f} ms", flush=True)
os.environ.pop("SFG_UB_MOVER_FAKE", None)
print(f"G={G} NTTs alone {base_ntt:.2f} ms; pass alone {base_pack:.2f} ms; one after the other {base_seq:.2f} ms", flush=True)
for fake, what in ((0, "real accesses"), (1, "contiguous reads"), (2, "contiguous writes"), (3, "contiguous reads and writes"), (8, "reads as 2 KiB runs: panel [column][plane][c block][k][128 B]"), (16, "the NTTs WRITE that panel pattern (128-byte runs), real mover accesses"), (24, "K-major panel: NTT writes and mover reads"),
                   (4 | (1 << 8), "sleeping movers, 127 x 64 clocks per unit")):
    os.environ["SFG_UB_MOVER_FAKE"] = str(fake)
    for nb, depth in ((192, 1), (256, 1), (256, 2)):
        alone = run(4, max(nb, 1024), depth, 1) if not fake & 4 else float("nan")
        t = run(5, nb, depth, 1)
        print(f"G={G} fake {fake:5d} ({what}): {nb} movers depth {depth}: riding {t:.2f} ms ({t / base_seq:.3f} of sequential); mover alone with {max(nb, 1024)} workgroups {alone:.2f} ms", flush=True)
