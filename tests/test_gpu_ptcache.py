"""The plaintext coefficient cache of a resident matrix (sfg_geno_set_plaintext_cache: the device counterpart of the reference's DiagCache files,
gwas/matmult.go:1228-1334) never changes an output word: products with the cache off, filling, hitting in the orientation that filled it, and hitting in the
OTHER orientation (NTT through the automorphism permutation of the cached rows) are compared word for word at a ragged 2 x 2 block shape."""
import ctypes as C
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SLOTS, D, N, L, LEVEL = 8192, 91, 16384, 5, 5
NROW, NCOL, S = SLOTS + 300, SLOTS + 517, 2
T, SQ = 2, 1          # SFG_TRANSPOSE, SFG_SQUARE


@pytest.fixture(scope="module")
def env():
    from sfgwas_amd import capi
    assert (capi.SFG_TRANSPOSE, capi.SFG_SQUARE) == (T, SQ)
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    lib = capi.lib()
    rots = list(range(1, D)) + [g * D for g in range(1, D) if g * D < SLOTS]
    ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), "keys")
    rng = np.random.default_rng(77)
    geno = rng.integers(0, 3, (NROW, NCOL), dtype=np.int8)
    geno[rng.random((NROW, NCOL)) < 0.01] = -9                       # missing calls
    A = {0: ctx.fill_uniform_cts(S * 2, LEVEL, 0xA1), T: ctx.fill_uniform_cts(S * 2, LEVEL, 0xA2)}     # 2 block rows in either orientation

    def prod(g, flags):
        out = ctx.matmul_resident(A[flags & T], S, LEVEL, L, g, flags)
        h = out.host().copy()
        out.free()
        return h

    def stats(g):
        v = [C.c_size_t() for _ in range(4)]
        ctx.check(lib.sfg_geno_plaintext_cache_stats(ctx.h, g, *[C.byref(x) for x in v]), "stats")
        return tuple(x.value for x in v)        # blocks, bytes, hits, fills

    g0 = ctx.geno_upload(geno)
    ref = {f: prod(g0, f) for f in (0, T, SQ, T | SQ)}
    ctx.geno_free(g0)
    yield ctx, lib, geno, prod, stats, ref
    for a in A.values():
        a.free()
    ctx.close()


def test_fill_then_hit_in_both_orientations(env):
    ctx, lib, geno, prod, stats, ref = env
    g = ctx.geno_upload(geno)
    ctx.check(lib.sfg_geno_set_plaintext_cache(ctx.h, g, 8 << 30), "enable")
    assert np.array_equal(prod(g, 0), ref[0])                         # fills 4 blocks
    assert stats(g) == (4, 4 * SLOTS * SLOTS * 8, 0, 4)
    assert np.array_equal(prod(g, 0), ref[0])                         # same orientation: the rows as cached
    assert stats(g)[2:] == (4, 4)
    assert np.array_equal(prod(g, T), ref[T])                         # other orientation: automorphism images of the cached rows
    assert stats(g)[2:] == (8, 4)
    assert np.array_equal(prod(g, T | SQ), ref[T | SQ])               # the squared flavour is its own entry, filled from the transposed orientation here
    assert stats(g)[0] == 8
    assert np.array_equal(prod(g, SQ), ref[SQ])                       # ... and hit from the stored one
    assert stats(g) == (8, 8 * SLOTS * SLOTS * 8, 12, 8)
    ctx.check(lib.sfg_geno_set_plaintext_cache(ctx.h, g, 0), "drop")
    assert stats(g)[:2] == (0, 0)
    assert np.array_equal(prod(g, T), ref[T])
    ctx.geno_free(g)


def test_filled_from_the_transposed_product_and_partial_budget(env):
    ctx, lib, geno, prod, stats, ref = env
    g = ctx.geno_upload(geno)
    ctx.check(lib.sfg_geno_set_plaintext_cache(ctx.h, g, 3 * SLOTS * SLOTS * 8 + 5), "enable")     # room for 3 of the 4 blocks
    assert np.array_equal(prod(g, T), ref[T])
    assert stats(g)[0] == 3
    assert np.array_equal(prod(g, 0), ref[0])                         # 3 blocks through the permutation, 1 encoded afresh
    assert np.array_equal(prod(g, T), ref[T])
    assert stats(g) == (3, 3 * SLOTS * SLOTS * 8, 6, 3)
    ctx.geno_free(g)                                                  # frees the slots with the handle


def test_packed_matrix(env):
    ctx, lib, geno, prod, stats, ref = env
    g8 = ctx.geno_upload(geno)
    gp = C.c_void_p()
    ctx.check(lib.sfg_geno_pack(ctx.h, g8, C.byref(gp)), "pack")
    ctx.geno_free(g8)
    ctx.check(lib.sfg_geno_set_plaintext_cache(ctx.h, gp, 8 << 30), "enable")
    assert np.array_equal(prod(gp, 0), ref[0])
    assert np.array_equal(prod(gp, T), ref[T])
    assert np.array_equal(prod(gp, 0), ref[0])
    assert stats(gp)[2:] == (8, 4)
    ctx.geno_free(gp)


def test_another_context_cannot_take_the_cache(env):
    ctx, lib, geno, prod, stats, ref = env
    from sfgwas_amd import capi
    g = ctx.geno_upload(geno[:64, :64])
    ctx.check(lib.sfg_geno_set_plaintext_cache(ctx.h, g, 1 << 30), "enable")
    other = capi.Context(ol.Q_PN14, ol.P_PN14)
    assert lib.sfg_geno_set_plaintext_cache(other.h, g, 1 << 30) != 0
    assert b"another context" in lib.sfg_last_error(other.h)
    other.close()
    ctx.geno_free(g)


def test_unprovable_rounding_drops_the_cached_rows(env):
    """rows cached while the 2^-50 near-tie condition is outstanding must not be served after the documented recovery (reset): the failing
    synchronising call and the reset both make the owning context's caches forget their rows; later products refill and agree"""
    ctx, lib, geno, prod, stats, ref = env
    from sfgwas_amd import capi
    g = ctx.geno_upload(geno)
    ctx.check(lib.sfg_geno_set_plaintext_cache(ctx.h, g, 8 << 30), "enable")
    ctx.encoder_near_ties(reset=True)
    assert np.array_equal(prod(g, 0), ref[0])
    assert stats(g)[0] == 4
    ctx.check(lib.sfg_ctx_encoder_inject_unsafe_for_test(ctx.h, 1), "inject")
    with pytest.raises(capi.SfgError, match="rounding tie"):
        ctx.sync()
    assert stats(g)[:2] == (0, 0)                                     # forgotten at the report ...
    ctx.check(lib.sfg_ctx_encoder_inject_unsafe_for_test(ctx.h, 1), "inject")
    ctx.encoder_near_ties(reset=True)                                 # ... and at the reset
    assert stats(g)[:2] == (0, 0)
    assert np.array_equal(prod(g, T), ref[T])                         # refills (4 blocks, from the transposed orientation this time)
    assert stats(g)[0] == 4 and stats(g)[3] == 8
    assert np.array_equal(prod(g, 0), ref[0])
    ctx.geno_free(g)


def test_failure_path_hook_is_refused_without_the_test_switch(env):
    import os
    from sfgwas_amd import capi
    saved = os.environ.pop("SFG_ENABLE_TEST_HOOKS", None)
    try:
        other = capi.Context(ol.Q_PN14, ol.P_PN14)
    finally:
        if saved is not None:
            os.environ["SFG_ENABLE_TEST_HOOKS"] = saved
    assert capi.lib().sfg_ctx_encoder_inject_unsafe_for_test(other.h, 1) != 0
    assert b"test switch" in capi.lib().sfg_last_error(other.h)
    other.sync()
    other.close()


def test_destroying_the_owner_detaches_its_matrices(env):
    """a fork that owns a matrix's cache is destroyed BEFORE the matrix is freed on another context - the handle must not keep a pointer to the
    dead context (sfg_ctx_destroy releases the caches it owns and clears the owner); the matrix keeps multiplying, cache off, and can be cached again elsewhere."""
    ctx, lib, geno, prod, stats, ref = env
    g = ctx.geno_upload(geno)
    fork = ctx.fork()
    fork.check(lib.sfg_geno_set_plaintext_cache(fork.h, g, 8 << 30), "enable on the fork")
    Af = fork.fill_uniform_cts(S * 2, LEVEL, 0xA1)
    out = fork.matmul_resident(Af, S, LEVEL, L, g, 0)
    assert np.array_equal(out.host(), ref[0])
    assert stats(g)[0] == 4
    out.free(); Af.free()
    fork.close()                                                      # the owner goes first
    assert stats(g)[:2] == (0, 0)
    assert np.array_equal(prod(g, T), ref[T])                         # no cache: encoded afresh on the surviving context
    ctx.check(lib.sfg_geno_set_plaintext_cache(ctx.h, g, 8 << 30), "the parent may take the cache now")
    assert np.array_equal(prod(g, 0), ref[0])
    assert stats(g)[0] == 4
    ctx.geno_free(g)                                                  # (used to dereference the destroyed fork)
