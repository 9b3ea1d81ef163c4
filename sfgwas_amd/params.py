"""CKKS parameter presets of the hot path (gwas/gwas.go:164-177 selects them by name; crypto.go:282-284 slots).

PN14QP438 is the preset of the reference's shipped configuration (config/configGlobal.toml:8).  The real integration takes
the moduli at run time from cryptoParams.Params.Qi()/Pi() (matmult.go:328 does the same); the values below are the
PN14QP438-shaped chain used by bench.py and the tests: q0 is the 46-bit prime the preset starts with, the others are
NTT-friendly primes (== 1 mod 2^15) of the preset's sizes."""

LOGN = 14
N = 1 << LOGN
SLOTS = N // 2
D = 91                      # ceil(sqrt(slots)), matmult.go:1047
MAX_LEVEL = 5               # maxLevel at every call site (matmult.go:42,91; pca.go:112-113; assoc.go:395,424)
DEFAULT_SCALE = 2.0 ** 34

Q_PN14 = [0x200000440001, 0x7fff80001, 0x800280001, 0x7ffd80001, 0x7ffc80001,
          0x7ff9c0001, 0x800008001, 0x7fffb0001, 0x8000f8001, 0x800250001]
P_PN14 = [0x80000050001, 0x7fffffd8001]


def rotations_for_matmul(slots=SLOTS, d=D):
    """left-rotation amounts the streamed products need keys for: baby steps 1..d-1 and giant steps d*g
    (crypto.go:252-263 generates exactly these; matmult.go:1375,1476 use them)"""
    return list(range(1, d)) + [g * d for g in range(1, d) if g * d < slots]
