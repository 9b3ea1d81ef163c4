"""Timing experiment: do transposition workgroups riding in the plaintext NTT's launches (k_ntt_half3_pack) hide the transposition pass?  See ubench_ntt_pack, mac_i8.hip."""
import ctypes as C
import os
import sys
os.environ["SFG_ENABLE_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfgwas_amd import capi, params as P          # noqa: E402

ctx = capi.Context(P.Q_PN14, P.P_PN14)
lib = capi.lib()
f = lib.ubench_ntt_pack
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 13


def run(mode, per_launch=0, cap=0, reps=3):
    ms = C.c_double()
    ctx.check(f(ctx.h, mode, G, per_launch, cap, reps, C.byref(ms)), "ubench_ntt_pack")
    return ms.value


names = {0: "NTTs then transposition", 1: "transposition riding in the NTT launches", 2: "NTTs alone", 3: "transposition alone"}
for mode in (2, 3, 0, 1):
    print(f"G={G} mode {mode} ({names[mode]}): {run(mode):.2f} ms", flush=True)
for cap in (8, 16, 24, 32, 48, 64):
    print(f"G={G} fused, period cap {cap} NTT blocks per 8 transposition blocks: {run(1, 0, cap):.2f} ms", flush=True)
for per in (1000, 2000, 3000, 4000):
    try:
        print(f"G={G} fused, {per} transposition workgroups per launch: {run(1, per):.2f} ms", flush=True)
    except capi.SfgError as e:
        print(f"G={G} fused, {per} per launch: {e}", flush=True)
