"""Size-independent properties of the GPU product at multi-block sizes (several full 8192 x 8192 blocks), where
the CPU oracle would take hours.  All checks are bit-exact self-consistency of the HIP path:
  * block-row grouping (K fusion) does not change any output word;
  * the LDS-DMA and the register-staged MAC kernels agree;
  * computing block-column ranges separately and concatenating equals the full product (output sharding);
  * accumulate over block-row ranges, summed mod q, then finalize equals the one-shot product (contraction sharding);
  * an all-zero / all-missing genotype matrix gives the zero ciphertext;
  * X^T through SFG_TRANSPOSE equals the product with an explicitly transposed upload."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SLOTS, D, N, L, LEVEL = 8192, 91, 16384, 5, 5


class Env:
    def __init__(self):
        from sfgwas_amd import capi
        self.capi = capi
        self.ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
        self.lib = capi.lib()
        rots = list(range(1, D)) + [g * D for g in range(1, D) if g * D < SLOTS]
        arr = (C.c_int * len(rots))(*rots)
        self.ctx.check(self.lib.sfg_fill_rotkeys_synthetic(self.ctx.h, arr, len(rots), 0xBEEF), "keys")

    def geno(self, nrow, ncol, seed, host=None):
        gh = C.c_void_p()
        if host is not None:
            host = np.ascontiguousarray(host, dtype=np.int8)
            self.ctx.check(self.lib.sfg_geno_upload(self.ctx.h, host.ctypes.data_as(C.c_void_p), nrow, ncol, ncol, C.byref(gh)), "upload")
            return gh, None
        d = self.ctx.malloc(nrow * ncol)
        self.ctx.check(self.lib.sfg_fill_geno_dev(self.ctx.h, d, nrow, ncol, seed), "fill")
        self.ctx.check(self.lib.sfg_geno_from_device(self.ctx.h, d, nrow, ncol, ncol, C.byref(gh)), "geno")
        return gh, d

    def cts(self, n, seed):
        d = self.ctx.malloc(n * 2 * (LEVEL + 1) * N * 8)
        self.ctx.check(self.lib.sfg_fill_uniform_ct_dev(self.ctx.h, d, n, LEVEL, seed), "fill ct")
        return d

    def product(self, A, s, gh, flags, m_out, blk=None):
        out = self.ctx.malloc(s * m_out * 2 * L * N * 8)
        if blk is None:
            self.ctx.check(self.lib.sfg_matmul_resident_dev(self.ctx.h, A, s, LEVEL, L, gh, flags, out), "matmul")
        else:
            self.ctx.check(self.lib.sfg_matmul_resident_range_dev(self.ctx.h, A, s, LEVEL, L, gh, flags, blk[0], blk[1], out), "range")
        h = self.ctx.to_host(out, (s, m_out, 2, L, N), np.uint64)
        self.ctx.free(out)
        return h


@pytest.fixture(scope="module")
def env():
    e = Env()
    yield e
    e.ctx.close()


def test_range_concat_and_accumulate_finalize(env):
    s, nrow, ncol = 2, 2 * SLOTS + 100, 2 * SLOTS            # 3 block rows x 2 block cols
    gh, gd = env.geno(nrow, ncol, 5)
    nbr, m_ct = 3, 2
    A = env.cts(s * nbr, 11)
    full = env.product(A, s, gh, 0, m_ct)
    # output sharding: block columns computed separately
    parts = [env.product(A, s, gh, 0, 1, blk=(j, j + 1)) for j in range(m_ct)]
    assert np.array_equal(np.concatenate(parts, axis=1), full)
    # contraction sharding: accumulate two block-row ranges separately, add mod q, finalize
    accw = m_ct * D * s * 2 * L * N
    acc = [env.ctx.malloc(accw * 8) for _ in range(2)]
    for k, (b0, b1) in enumerate([(0, 1), (1, 3)]):
        env.ctx.check(env.lib.sfg_matmul_accumulate_dev(env.ctx.h, A, s, LEVEL, L, gh, 0, b0, b1, 0, m_ct, 0, acc[k]), "acc")
    h = [env.ctx.to_host(a, (m_ct, D, s, 2, L, N), np.uint64) for a in acc]
    tot = h[0] + h[1]                                          # < 2^47: no overflow
    for l in range(L):
        tot[..., l, :] %= np.uint64(ol.Q_PN14[l])
    env.ctx.check(env.lib.sfg_memcpy_h2d(env.ctx.h, acc[0], tot.ctypes.data_as(C.c_void_p), tot.nbytes), "h2d")
    out = env.ctx.malloc(s * m_ct * 2 * L * N * 8)
    env.ctx.check(env.lib.sfg_matmul_finalize_dev(env.ctx.h, acc[0], s, L, m_ct, 0, D, 0, out), "fin")
    two_phase = env.ctx.to_host(out, (s, m_ct, 2, L, N), np.uint64)
    assert np.array_equal(two_phase, full)
    # also: summing un-reduced accumulators on the device and reducing there
    env.ctx.check(env.lib.sfg_memcpy_h2d(env.ctx.h, acc[1], (h[0] + h[1]).ctypes.data_as(C.c_void_p), tot.nbytes), "h2d")
    env.ctx.check(env.lib.sfg_reduce_rows_dev(env.ctx.h, acc[1], m_ct * D * s * 2, L), "reduce")
    assert np.array_equal(env.ctx.to_host(acc[1], tot.shape, np.uint64), tot)
    for p in acc + [out, A]:
        env.ctx.free(p)
    env.lib.sfg_geno_free(env.ctx.h, gh)
    env.ctx.free(gd)


def test_transpose_flag_equals_explicit_transpose(env):
    rnd = np.random.default_rng(9)
    X = rnd.integers(-1, 3, (SLOTS + 37, 300)).astype(np.int8)
    s = 1
    gh, _ = env.geno(X.shape[0], X.shape[1], 0, host=X)
    ght, _ = env.geno(X.shape[1], X.shape[0], 0, host=X.T)
    A = env.cts(s * 1, 3)                                     # operand X^T: 300 rows -> 1 block row
    a = env.product(A, s, gh, env.capi.SFG_TRANSPOSE, 2)
    b = env.product(A, s, ght, 0, 2)
    assert np.array_equal(a, b)
    env.ctx.free(A)
    env.lib.sfg_geno_free(env.ctx.h, gh)
    env.lib.sfg_geno_free(env.ctx.h, ght)


def test_zero_and_all_missing_matrix_give_zero(env):
    s = 2
    for fill in (0, -1):
        X = np.full((700, SLOTS + 5), fill, dtype=np.int8)
        gh, _ = env.geno(X.shape[0], X.shape[1], 0, host=X)
        A = env.cts(s, 17)
        out = env.product(A, s, gh, 0, 2)
        assert not out.any()
        env.ctx.free(A)
        env.lib.sfg_geno_free(env.ctx.h, gh)


def child_env(switches):
    from sfgwas_amd import capi
    return capi.env_for(switches)


_CHILD = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib as ol
from sfgwas_amd import capi
ctx = capi.Context(ol.Q_PN14, ol.P_PN14); lib = capi.lib()
D, N, L, LEVEL = 91, 16384, 5, 5
rots = list(range(1, D)) + [g * D for g in range(1, D) if g * D < 8192]
ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), 'keys')
nrow, ncol, s = 3 * 8192 - 11, 8192 + 9, 2
g = ctx.malloc(nrow * ncol); ctx.check(lib.sfg_fill_geno_dev(ctx.h, g, nrow, ncol, 77), 'g')
gh = C.c_void_p(); ctx.check(lib.sfg_geno_from_device(ctx.h, g, nrow, ncol, ncol, C.byref(gh)), 'gh')
A = ctx.malloc(s * 3 * 2 * 6 * N * 8); ctx.check(lib.sfg_fill_uniform_ct_dev(ctx.h, A, s * 3, LEVEL, 5), 'A')
out = ctx.malloc(s * 2 * 2 * L * N * 8)
ctx.check(lib.sfg_matmul_resident_dev(ctx.h, A, s, LEVEL, L, gh, 0, out), 'mm')
h = ctx.to_host(out, (s, 2, 2, L, N), np.uint64)
np.save(sys.argv[1], h)
"""


def test_group_size_and_mac_kernel_invariance(tmp_path):
    """The product's words (outs[0], outs[1], outs[3], outs[4]: the product library with deployment switches only) against every superseded kernel, each selected by its
    A/B switch in the experimenters' build (sfgwas_amd/lib_ab, `make ab`).  The same product in separate processes: SFG_MM_GROUP=1, SFG_MM_GROUP=8, the register-staged MAC, and a small
    accumulator budget (one block column per pass -> the product-wide rotation cache is built once and reused), the round-1 LDS-DMA MAC in both
    workgroup shapes, the plain plaintext panel, the full-image NTT kernels, the fp64 DPP-broadcast MAC for every modulus (SFG_MAC_IMPL=bc; the default runs the small moduli on the int8 matrix core)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, envv in [("g1", {"SFG_MM_GROUP": "1"}), ("g8", {"SFG_MM_GROUP": "8"}), ("reg", {"SFG_MAC_IMPL": "reg"}),
                       ("budget", {"SFG_MM_ACC_BUDGET_MB": "300"}), ("budget_g1", {"SFG_MM_ACC_BUDGET_MB": "300", "SFG_MM_GROUP": "1"}),
                       ("dma", {"SFG_MAC_IMPL": "dma"}), ("dma_wc2", {"SFG_MAC_IMPL": "dma", "SFG_MAC_WC": "2"}), ("plain_pt", {"SFG_MAC_PT": "plain"}),
                       ("ntt_full", {"SFG_NTT_FWD_IMPL": "full", "SFG_NTT_HALF_IMPL": "full"}), ("bc", {"SFG_MAC_IMPL": "bc"}), ("bc_g1", {"SFG_MAC_IMPL": "bc", "SFG_MM_GROUP": "1"}), ("i8_lds", {"SFG_MAC_I8_ROT": "lds"}), ("i8_big", {"SFG_MAC_I8_BIG": "0"}),
                       ("i8_cache", {"SFG_MAC_I8_ROT": "cache"}), ("i8_wg1", {"SFG_MAC_I8_WG": "1"}), ("i8_w6", {"SFG_MAC_I8_WAVES": "6"}), ("i8_w6_big0", {"SFG_MAC_I8_WAVES": "6", "SFG_MAC_I8_BIG": "0"})]:
        f = str(tmp_path / (name + ".npy"))
        e = dict(os.environ); e.update(child_env(envv))      # kernel A/B switches exist in the A/B build only: those children load sfgwas_amd/lib_ab, the others the product
        r = subprocess.run([sys.executable, "-c", _CHILD, f], cwd=root, env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
    assert np.array_equal(outs[0], outs[1]), "block-row grouping changed the result"
    assert np.array_equal(outs[0], outs[2]), "default and register-staged MAC kernels disagree"
    assert np.array_equal(outs[0], outs[3]) and np.array_equal(outs[0], outs[4]), "column passes / shared rotation cache changed the result"
    assert np.array_equal(outs[0], outs[5]) and np.array_equal(outs[0], outs[6]), "DPP-broadcast and 8 x 3-tile LDS-DMA MAC kernels (4- / 8-wave workgroups) disagree"
    assert np.array_equal(outs[0], outs[7]), "packed-limb and plain plaintext panels disagree"
    assert np.array_equal(outs[0], outs[8]), "split and full-image NTT kernels disagree"
    assert np.array_equal(outs[0], outs[9]) and np.array_equal(outs[0], outs[10]), "the int8 matrix-core MAC (default: five signed base-256 digits, nine int32 sums) and the fp64 DPP-broadcast MAC of round 2 (SFG_MAC_IMPL=bc) disagree"
    assert np.array_equal(outs[0], outs[11]), "int8 MAC with LDS-staged rot tiles disagrees"
    assert np.array_equal(outs[0], outs[12]), "the 46-bit modulus on the int8 matrix core (six digits, the default) disagrees with the fp64 kernel (SFG_MAC_I8_BIG=0)"
    assert np.array_equal(outs[0], outs[13]) and np.array_equal(outs[0], outs[14]), "int8 MAC from the LDS prefetch ring (default) and straight from global memory (SFG_MAC_I8_ROT=cache, SFG_MAC_I8_WG=1) disagree"
    assert np.array_equal(outs[0], outs[15]) and np.array_equal(outs[0], outs[16]), "twelve-wave (default) and six-wave ring MAC disagree"
    assert outs[0].any()


def test_more_than_thirty_rows_per_k_slice_across_groups():
    """s = 17 ciphertexts (34 rows of every k-slice: MAC passes of 30 + 4 rows) over three block rows in two MAC groups (SFG_MM_GROUP=2: the second group
    accumulates).  The int8 kernels take up to 32 rows from a pass's first row, so a pass must stop at its own last row - rows 30, 31 were added twice before
    round 4.  Row independence: rows [0, 8) and [8, 17) multiplied on their own (16 and 18 rows: single passes) give the same words."""
    from sfgwas_amd import capi
    import ctypes as C
    import oracle_lib as ol
    saved = os.environ.get("SFG_MM_GROUP"); os.environ["SFG_MM_GROUP"] = "2"
    try:
        ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    finally:
        if saved is None:
            os.environ.pop("SFG_MM_GROUP", None)
        else:
            os.environ["SFG_MM_GROUP"] = saved
    lib = capi.lib()
    D, N, L, LEVEL, SLOTS = 91, 16384, 5, 5, 8192
    rots = list(range(1, D)) + [g * D for g in range(1, D) if g * D < SLOTS]
    ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), "keys")
    nrow, ncol, s = 2 * SLOTS + 500, SLOTS + 70, 17          # (two block columns: the second one's encode carries the first one's transposition - round 6 - in both row passes' tiles)
    nbr = 3
    gd, gh = ctx.fill_geno(nrow, ncol, 0x77)
    A = ctx.fill_uniform_cts(s * nbr, LEVEL, 0x1234)
    full = ctx.matmul_resident(A, s, LEVEL, L, gh).host()
    assert full.shape == (s, 2, 2, L, N) and full.any()
    Ah = A.host().reshape(s, nbr, 2, LEVEL + 1, N)
    for r0, r1 in ((0, 8), (8, 17)):
        sub = capi.DevArray.from_host(ctx, np.ascontiguousarray(Ah[r0:r1]))
        part = ctx.matmul_resident(sub, r1 - r0, LEVEL, L, gh)
        assert np.array_equal(part.host(), full[r0:r1]), f"rows [{r0}, {r1}) of the 34-row product differ from the product of those rows alone"
        sub.free(); part.free()
    A.free(); gd.free(); ctx.geno_free(gh); ctx.close()


_CHILD_LARGE = r"""
import sys, ctypes as C, hashlib, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib as ol
from sfgwas_amd import capi
ctx = capi.Context(ol.Q_PN14, ol.P_PN14); lib = capi.lib()
D, N, L, LEVEL, SLOTS = 91, 16384, 5, 5, 8192
rots = list(range(1, D)) + [g * D for g in range(1, D) if g * D < SLOTS]
ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), 'keys')
nrow, ncol, s = 50000, 131077, 15               # BASELINE c3 rows (7 block rows, ragged), 17 block columns (ragged), kp = 15
nbr, mct = -(-nrow // SLOTS), -(-ncol // SLOTS)
g = ctx.malloc(nrow * ncol); ctx.check(lib.sfg_fill_geno_dev(ctx.h, g, nrow, ncol, 0x5F6A), 'g')
gh = C.c_void_p(); ctx.check(lib.sfg_geno_from_device(ctx.h, g, nrow, ncol, ncol, C.byref(gh)), 'gh')
h = hashlib.sha256()
for flags, nin, nout in ((0, nbr, mct), (capi.SFG_TRANSPOSE, mct, nbr)):       # Q*X and Q'*X^T
    A = ctx.malloc(s * nin * 2 * (LEVEL + 1) * N * 8); ctx.check(lib.sfg_fill_uniform_ct_dev(ctx.h, A, s * nin, LEVEL, 0xC1F3 + flags), 'A')
    out = ctx.malloc(s * nout * 2 * L * N * 8)
    ctx.check(lib.sfg_matmul_resident_dev(ctx.h, A, s, LEVEL, L, gh, flags, out), 'mm')
    res = ctx.to_host(out, (s, nout, 2, L, N), np.uint64)
    assert res.any() and int(res.max()) < max(ol.Q_PN14[:L])
    h.update(res.tobytes())
    ctx.free(A); ctx.free(out)
open(sys.argv[1], 'w').write(h.hexdigest())
"""


def test_schedule_invariance_at_kp15_multi_pass_size(tmp_path):
    """Both products of a power iteration at kp = 15 on a 50 000 x 131 077 matrix (7 ragged block rows x 17 block
    columns: several block-row groups, two column passes at the default accumulator budget, the shared rotation cache
    engage).  The digest of every output word must not depend on the schedule: default (one queue) vs the two-queue overlap (SFG_MM_OVERLAP=1) + groups of 3 +
    a 3-column accumulator budget + the register-staged MAC kernel for the third run, and - round 6 - with and without the riding transposition."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for name, envv in [("default", {}),
                       ("two_queues", {"SFG_MM_OVERLAP": "1", "SFG_MM_GROUP": "3", "SFG_MM_ACC_BUDGET_MB": "6000", "SFG_UPLOAD_BLOCKING": "1"}),
                       ("full_ntt", {"SFG_NTT_HALF_IMPL": "full", "SFG_NTT_FWD_IMPL": "full", "SFG_MM_GROUP": "5", "SFG_MAC_IMPL": "dma"}),
                       # the riding transposition (round 6, the default's schedule: a MAC launch's plaintext panel is transposed inside the NTT launches of the next
                       # launch's encode): off, i.e. the pass before every MAC launch; and with other mover shapes over groups of 2 block rows (four groups: the
                       # launches held over a group boundary take the pass)
                       ("no_ride", {"SFG_PT_RIDE": "0", "SFG_PT_COMPACT": "0"}),        # (and the panel's rows as full words, round 5's layout)
                       ("ride_deep", {"SFG_PT_RIDE": "64", "SFG_PT_RIDE_DEPTH": "3", "SFG_PT_RIDE_NT": "0", "SFG_MM_GROUP": "2", "SFG_PT_KMAJOR": "0"})]:       # (and the compact panel plaintext-major)
        f = str(tmp_path / (name + ".txt"))
        e = dict(os.environ); e.update(child_env(envv))
        r = subprocess.run([sys.executable, "-c", _CHILD_LARGE, f], cwd=root, env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(open(f).read())
    assert len(set(digests)) == 1, digests
