"""Round 6: is the plaintext NTT inside a riding launch slowed by the LATENCY of its own loads behind the movers' traffic?  tools/r6_mover_ubench.py's harness with a
library whose NTT takes no phase-B twiddle loads (ntt.hip built with -DSFG_NTT_DIAG=3 into sfgwas_amd/lib_diag; results INVALID): NTTs alone, and riding."""
import ctypes as C
import os
import sys
os.environ["SFG_ENABLE_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfgwas_amd import capi, params as P          # noqa: E402

ctx = capi.Context(P.Q_PN14, P.P_PN14)
f = capi.lib().ubench_ntt_move
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 13


def run(mode, nblocks=192, depth=1, nt=1, reps=3):
    ms = C.c_double()
    ctx.check(f(ctx.h, mode, G, nblocks, depth, nt, reps, C.byref(ms)), "ubench_ntt_move")
    return ms.value


for fake in (0, 24):
    os.environ["SFG_UB_MOVER_FAKE"] = str(fake)
    print(f"{os.environ.get('SFG_LIB_PATH', 'product')} fake {fake}: NTTs alone {run(2):.2f} ms; pass alone {run(3):.2f}; riding 192 x 1: {run(5, 192, 1, 1):.2f}; 256 x 2: {run(5, 256, 2, 1):.2f}; 192 x 3: {run(5, 192, 3, 1):.2f}", flush=True)
