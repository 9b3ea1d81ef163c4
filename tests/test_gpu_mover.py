"""The low-occupancy mover (sfgwas_amd/csrc/i8_move.hpp) against the transposition pass it stands in for (k_i8_pack_pt_digits): random panel bytes through both, every
word of both tile buffers compared on the device - alone on the chip and riding in front of plaintext-NTT launches (k_ntt_half3_move), at every depth of prefetch and
with / without streaming accesses.  The hook (ubench_ntt_move, modes 6 and 7) lives in the experimenters' build, so the comparison runs in a child process on
sfgwas_amd/lib_ab; what the PRODUCT does with the mover - the riding transposition over the K-major panel - is held against the oracle by every product test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_CHILD = r"""
import ctypes as C, os, sys
os.environ["SFG_ENABLE_TEST_HOOKS"] = "1"
sys.path.insert(0, '.')
from sfgwas_amd import capi, params as P
ctx = capi.Context(P.Q_PN14, P.P_PN14)
f = capi.lib().ubench_ntt_move
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
bad = []
# K = 91 (a ragged pair of chunks) and 273 k-rows: rows past K and columns 91..95 must come out as zeros in both.  (G, mode, workgroups, depth, streaming): the
# product's shape (192 x 1 x streaming, riding) first, then the other instances of the kernel
for G, mode, nblocks, depth, nt in ((3, 7, 192, 1, 1), (1, 7, 192, 1, 1), (3, 6, 256, 1, 1), (1, 7, 8, 3, 0), (3, 7, 192, 2, 1), (3, 6, 256, 3, 0), (1, 6, 256, 2, 0), (3, 7, 64, 3, 1)):
    ms = C.c_double()
    ctx.check(f(ctx.h, mode, G, nblocks, depth, nt, 1, C.byref(ms)), "ubench_ntt_move")
    if ms.value != 0.0:
        bad.append((G, mode, nblocks, depth, nt, ms.value))
print("differing", bad)
sys.exit(1 if bad else 0)
"""


def test_mover_writes_the_tiles_of_the_pass_alone_and_riding_in_ntt_launches():
    from sfgwas_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(capi.env_for({"SFG_AB_BUILD_FOR_TEST_HOOK": "1"}))          # (any non-deployment switch selects the A/B library)
    env["SFG_ENABLE_TEST_HOOKS"] = "1"
    r = subprocess.run([sys.executable, "-c", _CHILD], cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
