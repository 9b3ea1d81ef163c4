"""2-bit packed genotype residency: packing is lossless for {missing, 0, 1, 2}, and every product taken from a packed handle -
X and X^T, ragged blocks, SFG_SQUARE - is bit-identical to the product from the int8 handle (which the other tests compare with the oracle)."""
import ctypes as C
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_packed_handle_gives_identical_products_sums_and_roundtrip():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    L = capi.lib()
    rots = list(range(1, 91)) + [g * 91 for g in range(1, 91) if g * 91 < 8192]
    ctx.check(L.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), "keys")
    rnd = np.random.default_rng(31)
    nrow, ncol, s, level = 8192 + 37, 8192 + 1003, 2, 5                      # 2 x 2 blocks, ragged both ways, columns not a multiple of 16
    geno = rnd.choice(np.array([-1, 0, 1, 2], dtype=np.int8), size=(nrow, ncol), p=[0.03, 0.5, 0.3, 0.17])
    g = ctx.geno_upload(geno)
    gp = C.c_void_p()
    ctx.check(L.sfg_geno_pack(ctx.h, g, C.byref(gp)), "pack")
    back = np.empty_like(geno)
    ctx.check(L.sfg_geno_download(ctx.h, gp, back.ctypes.data_as(C.c_void_p)), "download packed")
    assert np.array_equal(back, geno), "pack / unpack is not lossless"
    s1, q1, s2, q2 = (np.zeros(ncol) for _ in range(4))
    ctx.check(L.sfg_geno_colsums(ctx.h, g, s1.ctypes.data_as(C.POINTER(C.c_double)), q1.ctypes.data_as(C.POINTER(C.c_double))), "colsums")
    ctx.check(L.sfg_geno_colsums(ctx.h, gp, s2.ctypes.data_as(C.POINTER(C.c_double)), q2.ctypes.data_as(C.POINTER(C.c_double))), "colsums packed")
    assert np.array_equal(s1, s2) and np.array_equal(q1, q2)
    for flags, nin in ((0, 2), (capi.SFG_TRANSPOSE, 2), (capi.SFG_SQUARE, 2)):
        A = ctx.fill_uniform_cts(s * nin, level, 0x77 + flags)
        a = ctx.matmul_resident(A, s, level, 5, g, flags).host()
        b = ctx.matmul_resident(A, s, level, 5, gp, flags).host()
        assert a.any() and np.array_equal(a, b), f"flags {flags}: packed and int8 handles disagree"
        A.free()
    # values the 2-bit layout cannot hold are refused, loudly
    geno[5, 7] = 3
    gbad = ctx.geno_upload(geno); gq = C.c_void_p()
    with pytest.raises(capi.SfgError, match="do not fit the 2-bit layout"):
        ctx.check(L.sfg_geno_pack(ctx.h, gbad, C.byref(gq)), "pack")
    for h in (g, gp, gbad):
        L.sfg_geno_free(ctx.h, h)
    ctx.close()
