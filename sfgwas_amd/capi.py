"""ctypes binding of libsfgwas_hip.so (the C-ABI declared in include/sfgwas_hip.h).

This is plumbing for tests and bench.py; the product is the shared library.  There is no CPU
fallback: if the library is missing or no GPU is present, calls fail loudly.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFG_LIB_PATH") or os.path.join(_HERE, "lib", "libsfgwas_hip.so")      # SFG_LIB_PATH: same-box A/B of two builds
AB_LIB_PATH = os.path.join(_HERE, "lib_ab", "libsfgwas_hip.so")      # the experimenters' build (`make -C sfgwas_amd/csrc ab`): superseded kernels + their A/B switches
# what the product library reads from the environment (sfgwas_amd/csrc/ctx.hip: read_config); every other SFG_* switch exists in the A/B build only
DEPLOYMENT_SWITCHES = {"SFG_MM_GROUP", "SFG_MM_ACC_BUDGET_MB", "SFG_ASSOC_ROTCACHE_MB", "SFG_KSW_BUDGET_MB", "SFG_ENC_BATCH", "SFG_UPLOAD_BLOCKING",
                       "SFG_MGPU_TRANSPORT", "SFG_MGPU_CACHE_GB", "SFG_RCCL_LIB", "SFG_ENABLE_TEST_HOOKS",
                       "SFG_TEST_SCRATCH_OOM", "SFG_TEST_TIE_BAND_LOG2", "SFG_MGPU_FORCE_COLLECTIVES"}


def ab_lib():
    """path of the A/B build, made on demand (a few minutes the first time; `make ab` is part of __graft_entry__.build)"""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "ab", "-j8"], stdout=subprocess.DEVNULL)
    return AB_LIB_PATH


def env_for(switches):
    """environment additions for a child process that sets `switches`: the A/B library when any of them is not a deployment switch"""
    e = {k: str(v) for k, v in switches.items()}
    if any(k.startswith("SFG_") and k not in DEPLOYMENT_SWITCHES for k in e):
        e["SFG_LIB_PATH"] = ab_lib()
    return e


class SfgConfig(C.Structure):
    """include/sfgwas_hip.h: sfg_config"""
    _fields_ = [("struct_size", C.c_uint32), ("mm_group", C.c_int), ("acc_budget_bytes", C.c_size_t), ("assoc_rotcache_bytes", C.c_size_t),
                ("ksw_budget_bytes", C.c_size_t), ("enc_batch", C.c_int), ("upload_blocking", C.c_int), ("mgpu_transport", C.c_char_p),
                ("mgpu_cache_bytes", C.c_size_t), ("rccl_lib", C.c_char_p)]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = C.sizeof(SfgConfig)
        for k, v in kw.items():
            setattr(self, k, v)

u64p = C.POINTER(C.c_uint64)
_lib = None

SFG_SQUARE = 1
SFG_TRANSPOSE = 2
SFG_STREAM_DIRECT = 4


class SfgError(RuntimeError):
    pass


def _sig(L):
    vp, i, u64, d, sz = C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_size_t
    S = {
        "sfg_ctx_create": (i, [C.POINTER(vp), i, i, i, i, u64p, u64p, d]),
        "sfg_ctx_create_ex": (i, [C.POINTER(vp), i, i, i, i, u64p, u64p, d, vp]),
        "sfg_config_default": (None, [vp]),
        "sfg_mgpu_create_ex": (i, [C.POINTER(vp), C.POINTER(i), i, i, i, i, u64p, u64p, d, vp]),
        "sfg_mgpu_create_rank_ex": (i, [C.POINTER(vp), i, i, i, vp, i, i, i, u64p, u64p, d, vp]),
        "sfg_ctx_fork": (i, [vp, C.POINTER(vp)]),
        "sfg_ctx_destroy": (None, [vp]),
        "sfg_last_error": (C.c_char_p, [vp]),
        "sfg_ctx_synchronize": (i, [vp]),
        "sfg_ctx_set_stream": (i, [vp, vp]),
        "sfg_ctx_release_scratch": (i, [vp]),
        "sfg_ctx_scratch_bytes": (i, [vp, C.c_char_p, C.POINTER(C.c_size_t)]),
        "sfg_rotcache_invalidate": (i, [vp]),
        "sfg_ctx_use_own_stream": (i, [vp]),
        "sfg_ctx_encoder_inject_unsafe_for_test": (i, [vp, C.c_ulonglong]),
        "sfg_ctx_encoder_resolved": (i, [vp, C.POINTER(C.c_ulonglong)]),
        "sfg_ctx_encoder_unprovable": (i, [vp, C.POINTER(C.c_ulonglong)]),
        "sfg_ctx_load_rotkey": (i, [vp, u64, u64p, i]),
        "sfg_ctx_load_secret_key": (i, [vp, u64p, i]),
        "sfg_ct_galois_dev": (i, [vp, vp, vp, i, i, u64]),
        "sfg_ct_mul_scalar_add_dev": (i, [vp, vp, u64p, vp, i, i]),
        "sfg_refresh_gen_shares_dev": (i, [vp, vp, i, i, vp, vp, i, vp, vp, vp, vp]),
        "sfg_refresh_finish_dev": (i, [vp, vp, i, i, vp, vp, vp, vp]),
        "sfg_refresh_gen_shares_scaled_dev": (i, [vp, vp, i, i, d, d, vp, vp, i, vp, vp, vp, vp]),
        "sfg_refresh_finish_scaled_dev": (i, [vp, vp, i, i, d, d, vp, vp, vp, vp]),
        "sfg_ckks_to_ss_share_dev": (i, [vp, vp, i, i, vp, i, vp, vp, vp]),
        "sfg_geno_pack": (i, [vp, vp, C.POINTER(vp)]),
        "sfg_geno_unpack": (i, [vp, vp, C.POINTER(vp)]),
        "sfg_assoc_stream_bed": (i, [vp, C.c_char_p, sz, sz, vp, vp, sz, vp, i, i, i, C.c_uint, vp, sz, C.POINTER(sz), vp, vp]),
        "sfg_assoc_stream_pgen": (i, [vp, C.c_char_p, vp, vp, sz, vp, i, i, i, C.c_uint, vp, sz, C.POINTER(sz), vp, vp]),
        "sfg_assoc_pgen": (i, [vp, vp, sz, vp, vp, sz, vp, i, i, i, C.c_uint, vp, sz, C.POINTER(sz), vp, vp]),
        "sfg_ctx_has_rotkey": (i, [vp, u64]),
        "sfg_ctx_export_rotkey": (i, [vp, u64, u64p]),
        "sfg_galois_for_rotation": (u64, [vp, i]),
        "sfg_malloc": (i, [vp, C.POINTER(vp), sz]),
        "sfg_free": (i, [vp, vp]),
        "sfg_memcpy_h2d": (i, [vp, vp, vp, sz]),
        "sfg_memcpy_d2h": (i, [vp, vp, vp, sz]),
        "sfg_memcpy_d2d": (i, [vp, vp, vp, sz]),
        "sfg_ct_drop_level_dev": (i, [vp, vp, vp, i, i, i]),
        "sfg_ntt_rows": (i, [vp, vp, i, C.POINTER(i)]),
        "sfg_intt_rows": (i, [vp, vp, i, C.POINTER(i)]),
        "sfg_mac_dev": (i, [vp, vp, vp, vp, i, i, i, i, i]),
        "sfg_mac_i8_dev": (i, [vp, vp, vp, vp, i, i, i, i, i, i]),
        "sfg_encode_diags_dev": (i, [vp, vp, sz, i, i, i, i, i, i, vp]),
        "sfg_encode_coeffs_host": (i, [vp, C.POINTER(d), i, C.POINTER(C.c_int64)]),
        "sfg_encode_vectors_dev": (i, [vp, C.POINTER(d), i, i, vp]),
        "sfg_ctx_encoder_near_ties": (i, [vp, C.POINTER(C.c_ulonglong), i]),
        "sfg_rotate_right_dev": (i, [vp, vp, vp, i, i, C.POINTER(i)]),
        "sfg_ct_add_dev": (i, [vp, vp, vp, vp, i, i]),
        "sfg_ct_sub_dev": (i, [vp, vp, vp, vp, i, i]),
        "sfg_ct_mul_scalar_dev": (i, [vp, vp, u64p, vp, i, i]),
        "sfg_ct_add_scalar_dev": (i, [vp, vp, u64p, vp, i, i]),
        "sfg_ct_add_plain_dev": (i, [vp, vp, vp, sz, vp, i, i]),
        "sfg_ctx_load_relinkey": (i, [vp, u64p, i]),
        "sfg_ct_mulrelin_dev": (i, [vp, vp, vp, vp, i, i]),
        "sfg_ct_mul_plain_dev": (i, [vp, vp, vp, sz, vp, i, i]),
        "sfg_ct_rescale_dev": (i, [vp, vp, vp, i, i]),
        "sfg_ct_innersum_dev": (i, [vp, vp, i, i, vp]),
        "sfg_geno_upload": (i, [vp, vp, sz, sz, sz, C.POINTER(vp)]),
        "sfg_geno_from_device": (i, [vp, vp, sz, sz, sz, C.POINTER(vp)]),
        "sfg_geno_free": (None, [vp, vp]),
        "sfg_geno_create": (i, [vp, sz, sz, C.POINTER(vp)]),
        "sfg_geno_write_rows": (i, [vp, vp, sz, sz, vp, sz]),
        "sfg_geno_compare_rows": (i, [vp, vp, C.c_uint, sz, sz, vp, sz, u64p]),
        "sfg_pinned_alloc": (i, [vp, C.POINTER(vp), sz]),
        "sfg_pinned_free": (i, [vp, vp]),
        "sfg_geno_set_plaintext_cache": (i, [vp, vp, C.c_size_t]),
        "sfg_geno_plaintext_cache_stats": (i, [vp, vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "sfg_geno_from_bed": (i, [vp, vp, sz, sz, sz, vp, vp, C.POINTER(vp)]),
        "sfg_geno_dims": (i, [vp, C.POINTER(sz), C.POINTER(sz)]),
        "sfg_pgen_dims": (i, [vp, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "sfg_geno_from_pgen": (i, [vp, vp, sz, sz, sz, vp, vp, C.POINTER(vp)]),
        "sfg_pgen_geno_counts": (i, [vp, vp, sz, vp, vp]),
        "sfg_geno_download": (i, [vp, vp, vp]),
        "sfg_geno_transpose": (i, [vp, vp, C.POINTER(vp)]),
        "sfg_geno_concat_cols": (i, [vp, C.POINTER(vp), i, C.POINTER(vp)]),
        "sfg_geno_colsums": (i, [vp, vp, C.POINTER(d), C.POINTER(d)]),
        "sfg_matmul_resident_dev": (i, [vp, vp, i, i, i, vp, C.c_uint, vp]),
        "sfg_matmul_from_cache": (i, [vp, vp, i, i, i, C.c_char_p, i, vp]),
        "sfg_diagcache_header": (i, [vp, C.c_char_p, i, u64p]),
        "sfg_diagcache_write": (i, [vp, vp, C.c_uint, i, C.c_char_p, C.POINTER(C.c_int)]),
        "sfg_matmul_stream": (i, [vp, u64p, i, i, i, vp, sz, sz, sz, C.c_uint, u64p, C.POINTER(d), C.POINTER(d)]),
        "sfg_matmul_resident_range_dev": (i, [vp, vp, i, i, i, vp, C.c_uint, i, i, vp]),
        "sfg_matmul_accumulate_dev": (i, [vp, vp, i, i, i, vp, C.c_uint, i, i, i, i, i, vp]),
        "sfg_matmul_finalize_dev": (i, [vp, vp, i, i, i, i, i, i, vp]),
        "sfg_matmul_finalize_slots_dev": (i, [vp, vp, i, i, i, i, i, i, i, i, vp]),
        "sfg_reduce_rows_dev": (i, [vp, vp, sz, i]),
        "sfg_rotcache_layout": (i, [vp, i, i, C.POINTER(sz), C.POINTER(sz)]),
        "sfg_rotcache_build_jobs_dev": (i, [vp, vp, i, i, i, i, i, i, vp]),
        "sfg_rotcache_scatter_dev": (i, [vp, vp, i, i, i, i, i, i, vp]),
        "sfg_rotcache_build_rows_dev": (i, [vp, vp, i, i, i, i, i, i, vp]),
        "sfg_matmul_resident_range_rc_dev": (i, [vp, vp, i, i, vp, C.c_uint, i, i, vp]),
        "sfg_matmul_accumulate_rc_dev": (i, [vp, vp, i, i, vp, C.c_uint, i, i, i, i, i, vp]),
        "sfg_beaver_elem_dev": (i, [vp, i, i, u64p, vp, vp, vp, vp, vp, sz]),
        "sfg_ss_mask_dev": (i, [vp, i, u64p, u64p, vp, vp, vp, vp, sz]),
        "sfg_ss_hub_share_dev": (i, [vp, i, u64p, vp, vp, vp, sz]),
        "sfg_beaver_elem": (i, [vp, i, i, u64p, u64p, u64p, u64p, u64p, u64p, sz]),
        "sfg_beaver_matmul": (i, [vp, i, i, u64p, u64p, u64p, u64p, u64p, u64p, i, i, i]),
        "sfg_sketch": (i, [vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int8), i, C.POINTER(d), u64p, u64p]),
        "sfg_fill_uniform_ct_dev": (i, [vp, vp, i, i, u64]),
        "sfg_fill_geno_dev": (i, [vp, vp, sz, sz, u64]),
        "sfg_fill_geno_window_dev": (i, [vp, vp, sz, sz, sz, sz, sz, u64]),
        "sfg_fill_rotkeys_synthetic": (i, [vp, C.POINTER(i), i, u64]),
        "sfg_mgpu_unique_id": (i, [vp]),
        "sfg_mgpu_create": (i, [C.POINTER(vp), C.POINTER(i), i, i, i, i, u64p, u64p, d]),
        "sfg_mgpu_create_rank": (i, [C.POINTER(vp), i, i, i, vp, i, i, i, u64p, u64p, d]),
        "sfg_mgpu_destroy": (None, [vp]),
        "sfg_mgpu_last_error": (C.c_char_p, [vp]),
        "sfg_mgpu_world": (i, [vp]),
        "sfg_mgpu_nlocal": (i, [vp]),
        "sfg_mgpu_rank": (i, [vp, i]),
        "sfg_mgpu_ctx": (vp, [vp, i]),
        "sfg_mgpu_transport": (C.c_char_p, [vp]),
        "sfg_mgpu_synchronize": (i, [vp]),
        "sfg_mgpu_load_rotkey": (i, [vp, u64, u64p, i]),
        "sfg_mgpu_load_relinkey": (i, [vp, u64p, i]),
        "sfg_mgpu_fill_rotkeys_synthetic": (i, [vp, C.POINTER(i), i, u64]),
        "sfg_mgpu_shard": (i, [i, sz, i, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
        "sfg_mgpu_geno_upload": (i, [vp, vp, sz, sz, sz, C.POINTER(vp)]),
        "sfg_mgpu_geno_adopt": (i, [vp, sz, sz, C.POINTER(vp), C.POINTER(vp)]),
        "sfg_mgpu_geno_create": (i, [vp, sz, sz, C.POINTER(vp)]),
        "sfg_mgpu_geno_write_rows": (i, [vp, vp, sz, sz, vp, sz]),
        "sfg_mgpu_geno_compare_rows": (i, [vp, vp, C.c_uint, sz, sz, vp, sz, u64p]),
        "sfg_mgpu_comm_info": (i, [vp, i, C.POINTER(i), C.POINTER(i)]),
        "sfg_mgpu_preflight": (i, [vp, sz]),
        "sfg_mgpu_geno_synthetic": (i, [vp, sz, sz, u64, i, C.POINTER(vp)]),
        "sfg_mgpu_geno_free": (None, [vp, vp]),
        "sfg_mgpu_geno_shard": (vp, [vp, i]),
        "sfg_mgpu_geno_dims": (i, [vp, C.POINTER(sz), C.POINTER(sz)]),
        "sfg_mgpu_geno_blocks": (i, [vp, i, C.POINTER(sz), C.POINTER(sz)]),
        "sfg_mgpu_geno_set_plaintext_cache": (i, [vp, vp, sz]),
        "sfg_mgpu_matmul_dev": (i, [vp, C.POINTER(vp), i, i, i, vp, C.c_uint, C.POINTER(vp)]),
        "sfg_mgpu_matmul": (i, [vp, u64p, i, i, i, vp, C.c_uint, u64p]),
        "sfg_mgpu_assoc_stream_bed": (i, [vp, C.c_char_p, sz, sz, vp, vp, sz, u64p, i, i, i, C.c_uint, u64p, sz, C.POINTER(sz), vp, vp]),
        "sfg_mgpu_assoc_stream_pgen": (i, [vp, C.c_char_p, vp, vp, sz, sz, u64p, i, i, i, C.c_uint, u64p, sz, C.POINTER(sz), vp, vp]),
        "sfg_ctx_clear_phases": (i, [vp]),
        "sfg_last_phase_ms": (d, [vp, C.c_char_p]),
        "sfg_last_phase_launches": (i, [vp, C.c_char_p]),
        "sfg_last_phase_bytes": (d, [vp, C.c_char_p]),
    }
    for name, (res, args) in S.items():
        fn = getattr(L, name)         # AttributeError here = the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return list(S.keys())


EXPORTS = []


def lib():
    global _lib, EXPORTS
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfgError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)")
        _lib = C.CDLL(LIB_PATH)
        EXPORTS = _sig(_lib)
    return _lib


def p64(a):
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    return a.ctypes.data_as(u64p)


class Context:
    """sfg_ctx wrapper. moduli = q list + p list; psi=None derives lattigo's root."""

    def __init__(self, q, p, scale=2.0 ** 34, logN=14, device=0, psi=None, config=None):
        L = lib()
        self.q, self.p = list(q), list(p)
        self.nq, self.np_ = len(q), len(p)
        self.N, self.slots = 1 << logN, (1 << logN) // 2
        mods = np.array(self.q + self.p, dtype=np.uint64)
        h = C.c_void_p()
        ps = None if psi is None else p64(np.array(psi, dtype=np.uint64))
        if config is None:
            rc = L.sfg_ctx_create(C.byref(h), device, logN, self.nq, self.np_, p64(mods), ps, float(scale))
        else:                                               # config: SfgConfig (include/sfgwas_hip.h: sfg_config)
            rc = L.sfg_ctx_create_ex(C.byref(h), device, logN, self.nq, self.np_, p64(mods), ps, float(scale), C.byref(config))
        if rc:
            raise SfgError("sfg_ctx_create: " + L.sfg_last_error(None).decode())
        self.h = h
        self.beta = (self.nq + self.np_ - 1) // self.np_

    def close(self):
        if getattr(self, "h", None):
            lib().sfg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc:
            raise SfgError(f"{what}: {lib().sfg_last_error(self.h).decode()}")

    # ---- memory
    def malloc(self, nbytes):
        p = C.c_void_p()
        self.check(lib().sfg_malloc(self.h, C.byref(p), nbytes), "sfg_malloc")
        return p

    def free(self, p):
        self.check(lib().sfg_free(self.h, p), "sfg_free")

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self.check(lib().sfg_memcpy_h2d(self.h, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")
        return p

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self.check(lib().sfg_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), p, out.nbytes), "d2h")
        return out

    def sync(self):
        self.check(lib().sfg_ctx_synchronize(self.h), "sync")

    # ---- ring substrate
    def ntt_rows(self, rows, mod_idx, inverse=False):
        """rows: [nrows][N] uint64 host array; returns transformed copy (round-trips through HBM)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        n = rows.shape[0]
        mi = (C.c_int * n)(*[int(x) for x in mod_idx])
        d = self.to_device(rows)
        fn = lib().sfg_intt_rows if inverse else lib().sfg_ntt_rows
        self.check(fn(self.h, d, n, mi), "ntt_rows")
        out = self.to_host(d, rows.shape, np.uint64)
        self.free(d)
        return out

    def load_rotkey(self, galois, key, montgomery=False):
        key = np.ascontiguousarray(key, dtype=np.uint64)
        self.check(lib().sfg_ctx_load_rotkey(self.h, int(galois), p64(key), int(montgomery)), "load_rotkey")

    def export_rotkey(self, galois):
        key = np.zeros((self.beta, 2, self.nq + self.np_, self.N), dtype=np.uint64)
        self.check(lib().sfg_ctx_export_rotkey(self.h, int(galois), p64(key)), "export_rotkey")
        return key

    # ---- collective bootstrap, local work
    def load_secret_key(self, sk_rows, montgomery=False):
        sk_rows = np.ascontiguousarray(sk_rows, dtype=np.uint64)
        assert sk_rows.shape == (self.nq, self.N)
        self.check(lib().sfg_ctx_load_secret_key(self.h, p64(sk_rows), int(montgomery)), "load_secret_key")

    def refresh_gen_shares(self, cts, level, crs, mask_limbs, e0, e1, scales=None):
        """cts [nct][2][level+1][N], crs [nct][nq][N], mask_limbs [nct][N][W] uint64, e0/e1 [nct][N] int32 -> (h0 [nct][level+1][N], h1 [nct][nq][N]);
        scales = (ciphertext scale, target scale) selects the target-scale form"""
        nct, W = cts.shape[0], mask_limbs.shape[-1]
        d = [self.to_device(np.ascontiguousarray(a)) for a in (cts, crs, mask_limbs, e0.astype(np.int32), e1.astype(np.int32))]
        h0, h1 = self.malloc(nct * (level + 1) * self.N * 8), self.malloc(nct * self.nq * self.N * 8)
        if scales is None:
            self.check(lib().sfg_refresh_gen_shares_dev(self.h, d[0], nct, level, d[1], d[2], W, d[3], d[4], h0, h1), "refresh_gen_shares")
        else:
            self.check(lib().sfg_refresh_gen_shares_scaled_dev(self.h, d[0], nct, level, float(scales[0]), float(scales[1]), d[1], d[2], W, d[3], d[4], h0, h1),
                       "refresh_gen_shares_scaled")
        out = self.to_host(h0, (nct, level + 1, self.N), np.uint64), self.to_host(h1, (nct, self.nq, self.N), np.uint64)
        for p_ in d + [h0, h1]:
            self.free(p_)
        return out

    def ckks_to_ss_share(self, cts, level, mask_limbs, e0):
        nct, W = cts.shape[0], mask_limbs.shape[-1]
        d = [self.to_device(np.ascontiguousarray(a)) for a in (cts, mask_limbs, e0.astype(np.int32))]
        h0, mk = self.malloc(nct * (level + 1) * self.N * 8), self.malloc(nct * (level + 1) * self.N * 8)
        self.check(lib().sfg_ckks_to_ss_share_dev(self.h, d[0], nct, level, d[1], W, d[2], h0, mk), "ckks_to_ss_share")
        out = self.to_host(h0, (nct, level + 1, self.N), np.uint64), self.to_host(mk, (nct, level + 1, self.N), np.uint64)
        for p_ in d + [h0, mk]:
            self.free(p_)
        return out

    def refresh_finish(self, cts, level, h0agg, h1agg, crs, scales=None):
        nct = cts.shape[0]
        d = [self.to_device(np.ascontiguousarray(a)) for a in (cts, h0agg, h1agg, crs)]
        o = self.malloc(nct * 2 * self.nq * self.N * 8)
        if scales is None:
            self.check(lib().sfg_refresh_finish_dev(self.h, d[0], nct, level, d[1], d[2], d[3], o), "refresh_finish")
        else:
            self.check(lib().sfg_refresh_finish_scaled_dev(self.h, d[0], nct, level, float(scales[0]), float(scales[1]), d[1], d[2], d[3], o), "refresh_finish_scaled")
        out = self.to_host(o, (nct, 2, self.nq, self.N), np.uint64)
        for p_ in d + [o]:
            self.free(p_)
        return out

    def fork(self):
        """a second caller on the same tables and keys (sfg_ctx_fork): own queues, scratch and timers"""
        h = C.c_void_p()
        self.check(lib().sfg_ctx_fork(self.h, C.byref(h)), "sfg_ctx_fork")
        child = Context.__new__(Context)
        child.__dict__.update({k: v for k, v in self.__dict__.items()})
        child.h = h
        return child

    def encoder_near_ties(self, reset=False):
        n = C.c_ulonglong()
        self.check(lib().sfg_ctx_encoder_near_ties(self.h, C.byref(n), int(reset)), "encoder_near_ties")
        return n.value

    def galois(self, k):
        return lib().sfg_galois_for_rotation(self.h, k)

    def phase_ms(self, name):
        return lib().sfg_last_phase_ms(self.h, name.encode())


def _ctx_mac(self, rot, pt, L, out_init=None):
    """rot: [K][R][L][N], pt: [K][Ncols][L][N] -> out [Ncols][R][L][N] (host arrays, staged through HBM)."""
    rot = np.ascontiguousarray(rot, dtype=np.uint64)
    pt = np.ascontiguousarray(pt, dtype=np.uint64)
    K, R = rot.shape[0], rot.shape[1]
    Ncols = pt.shape[1]
    assert rot.shape[2] == L and pt.shape[2] == L and pt.shape[0] == K
    d_rot, d_pt = self.to_device(rot), self.to_device(pt)
    if out_init is not None:
        d_out = self.to_device(np.ascontiguousarray(out_init, dtype=np.uint64))
    else:
        d_out = self.malloc(Ncols * R * L * self.N * 8)
    self.check(lib().sfg_mac_dev(self.h, d_rot, d_pt, d_out, K, R, Ncols, L, int(out_init is not None)), "sfg_mac_dev")
    out = self.to_host(d_out, (Ncols, R, L, self.N), np.uint64)
    for p in (d_rot, d_pt, d_out):
        self.free(p)
    return out


Context.mac = _ctx_mac


def _ctx_mac_i8(self, rot, pt_half, L, K=None, out_init=None, pt_form=0):
    """sfg_mac_i8_dev (test hook): the default int8 matrix-core MAC.  rot [P][R][L][N], pt_half [P][Ncols][L][N/2]; with K > P the P k-slices are
    repeated cyclically ON THE DEVICE (k-slice k = slice k mod P) so that long contractions cost one small upload.  Returns out [Ncols][R][L][N]."""
    rot = np.ascontiguousarray(rot, dtype=np.uint64)
    pt_half = np.ascontiguousarray(pt_half, dtype=np.uint64)
    P, R = rot.shape[0], rot.shape[1]
    Ncols = pt_half.shape[1]
    K = P if K is None else K
    assert rot.shape[2] == L and pt_half.shape[2] == L and pt_half.shape[0] == P and pt_half.shape[3] == self.N // 2 and K >= P
    bufs = []
    for arr in (rot, pt_half):
        sl = arr.nbytes // P                                    # bytes per k-slice
        d = self.malloc(sl * K)
        self.check(lib().sfg_memcpy_h2d(self.h, d, arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")
        filled = P
        while filled < K:
            n = min(filled, K - filled)
            self.check(lib().sfg_memcpy_d2d(self.h, C.c_void_p(d.value + filled * sl), d, n * sl), "d2d")
            filled += n
        bufs.append(d)
    if out_init is not None:
        d_out = self.to_device(np.ascontiguousarray(out_init, dtype=np.uint64))
    else:
        d_out = self.malloc(Ncols * R * L * self.N * 8)
    try:
        self.check(lib().sfg_mac_i8_dev(self.h, bufs[0], bufs[1], d_out, K, R, Ncols, L, int(out_init is not None), pt_form), "sfg_mac_i8_dev")
        out = self.to_host(d_out, (Ncols, R, L, self.N), np.uint64)
    finally:
        for p_ in bufs + [d_out]:
            self.free(p_)
    return out


Context.mac_i8 = _ctx_mac_i8


def _ctx_encode_diags(self, block, shift0, nshift, L, transposed=False):
    """block: [r][c] int8 host array (<= slots x slots) -> pt [nshift][L][N] uint64."""
    block = np.ascontiguousarray(block, dtype=np.int8)
    r, c = block.shape
    if transposed:
        r, c = c, r
    d_blk = self.to_device(block)
    d_pt = self.malloc(nshift * L * self.N * 8)
    self.check(lib().sfg_encode_diags_dev(self.h, d_blk, block.shape[1], r, c, int(transposed), shift0, nshift, L, d_pt), "sfg_encode_diags_dev")
    out = self.to_host(d_pt, (nshift, L, self.N), np.uint64)
    self.free(d_blk)
    self.free(d_pt)
    return out


Context.encode_diags = _ctx_encode_diags


def _ctx_rotate_right(self, cts, level, nrots):
    """cts: [nct][2][level+1][N] host array; nrots: list of right-rotation amounts."""
    cts = np.ascontiguousarray(cts, dtype=np.uint64)
    nct = cts.shape[0]
    d_in = self.to_device(cts)
    d_out = self.malloc(cts.nbytes)
    arr = (C.c_int * nct)(*[int(x) for x in nrots])
    self.check(lib().sfg_rotate_right_dev(self.h, d_in, d_out, nct, level, arr), "sfg_rotate_right_dev")
    out = self.to_host(d_out, cts.shape, np.uint64)
    self.free(d_in)
    self.free(d_out)
    return out


Context.rotate_right = _ctx_rotate_right


def _ctx_matmul_stream(self, A, in_level, max_level, geno, flags=0, want_sums=False):
    """MatMult4Stream through the C-ABI with host buffers.
    A: [s][nbr][2][in_level+1][N] uint64, geno: [nrow][ncol] int8 -> (out [s][m_ct][2][max_level][N], sum, sqsum)"""
    A = np.ascontiguousarray(A, dtype=np.uint64)
    geno = np.ascontiguousarray(geno, dtype=np.int8)
    s = A.shape[0]
    nrow, ncol = geno.shape
    lrow, lcol = (ncol, nrow) if flags & SFG_TRANSPOSE else (nrow, ncol)
    nbr, m_ct = (lrow - 1) // self.slots + 1, (lcol - 1) // self.slots + 1
    assert A.shape[1] == nbr, (A.shape, nbr)
    out = np.zeros((s, m_ct, 2, max_level, self.N), dtype=np.uint64)
    sm = np.zeros(ncol) if want_sums else None
    sq = np.zeros(ncol) if want_sums else None
    dp = C.POINTER(C.c_double)
    self.check(lib().sfg_matmul_stream(self.h, p64(A), s, in_level, max_level, geno.ctypes.data_as(C.c_void_p), nrow, ncol, ncol, flags,
                                       p64(out), sm.ctypes.data_as(dp) if want_sums else None, sq.ctypes.data_as(dp) if want_sums else None),
               "sfg_matmul_stream")
    return out, sm, sq


Context.matmul_stream = _ctx_matmul_stream


def random_rotkey(ctx_or_ring_moduli, beta, N, seed):
    """uniform random key words [beta][2][nmod][N] — enough for GPU-vs-oracle parity (which does not need a valid key)"""
    rnd = np.random.default_rng(seed)
    mods = ctx_or_ring_moduli
    k = np.zeros((beta, 2, len(mods), N), dtype=np.uint64)
    for m, q in enumerate(mods):
        k[:, :, m, :] = rnd.integers(0, q, (beta, 2, N), dtype=np.uint64)
    return k


def _ctx_evalop(self, name, level, *arrays, out_level=None, extra=()):
    """run one of the batched evaluator ops (host arrays in, host array out): test plumbing"""
    devs = [self.to_device(np.ascontiguousarray(a, dtype=np.uint64)) for a in arrays]
    nct = arrays[0].shape[0]
    ol_ = level if out_level is None else out_level
    out = self.malloc(nct * 2 * (ol_ + 1) * self.N * 8)
    fn = getattr(lib(), name)
    if name in ("sfg_ct_mul_plain_dev", "sfg_ct_add_plain_dev"):
        self.check(fn(self.h, devs[0], devs[1], extra[0], out, nct, level), name)
    elif name in ("sfg_ct_mul_scalar_dev", "sfg_ct_add_scalar_dev"):
        self.check(fn(self.h, devs[0], p64(np.ascontiguousarray(extra[0], dtype=np.uint64)), out, nct, level), name)
    elif name == "sfg_ct_rescale_dev":
        self.check(fn(self.h, devs[0], out, nct, level), name)
    else:
        self.check(fn(self.h, devs[0], devs[1], out, nct, level), name)
    res = self.to_host(out, (nct, 2, ol_ + 1, self.N), np.uint64)
    for d_ in devs + [out]:
        self.free(d_)
    return res


def _ctx_innersum(self, cts, level):
    cts = np.ascontiguousarray(cts, dtype=np.uint64)
    d_in = self.to_device(cts)
    out = self.malloc(2 * (level + 1) * self.N * 8)
    self.check(lib().sfg_ct_innersum_dev(self.h, d_in, cts.shape[0], level, out), "innersum")
    res = self.to_host(out, (2, level + 1, self.N), np.uint64)
    self.free(d_in); self.free(out)
    return res


def _ctx_load_relinkey(self, key, montgomery=False):
    key = np.ascontiguousarray(key, dtype=np.uint64)
    self.check(lib().sfg_ctx_load_relinkey(self.h, p64(key), int(montgomery)), "load_relinkey")


Context.evalop = _ctx_evalop
Context.innersum = _ctx_innersum
Context.load_relinkey = _ctx_load_relinkey


def _ctx_geno_to_host(self, g):
    nr, nc = C.c_size_t(), C.c_size_t()
    lib().sfg_geno_dims(g, C.byref(nr), C.byref(nc))
    out = np.empty((nr.value, nc.value), dtype=np.int8)
    self.check(lib().sfg_geno_download(self.h, g, out.ctypes.data_as(C.c_void_p)), "geno_download")
    return out


def _ctx_geno_from_bed(self, bed, num_sample, num_snp, row_filter=None, col_filter=None):
    bed = np.ascontiguousarray(bed, dtype=np.uint8)
    rf = None if row_filter is None else np.ascontiguousarray(row_filter, dtype=np.uint8)
    cf = None if col_filter is None else np.ascontiguousarray(col_filter, dtype=np.uint8)
    g = C.c_void_p()
    self.check(lib().sfg_geno_from_bed(self.h, bed.ctypes.data_as(C.c_void_p), bed.size, num_sample, num_snp,
                                       None if rf is None else rf.ctypes.data_as(C.c_void_p),
                                       None if cf is None else cf.ctypes.data_as(C.c_void_p), C.byref(g)), "geno_from_bed")
    return g


def _ctx_geno_from_pgen(self, img, v0=0, v1=0, row_filter=None, col_filter=None):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rf = None if row_filter is None else np.ascontiguousarray(row_filter, dtype=np.uint8)
    cf = None if col_filter is None else np.ascontiguousarray(col_filter, dtype=np.uint8)
    g = C.c_void_p()
    self.check(lib().sfg_geno_from_pgen(self.h, img.ctypes.data_as(C.c_void_p), img.size, v0, v1,
                                        None if rf is None else rf.ctypes.data_as(C.c_void_p),
                                        None if cf is None else cf.ctypes.data_as(C.c_void_p), C.byref(g)), "geno_from_pgen")
    return g


def _ctx_pgen_geno_counts(self, img, row_filter=None):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ns, nv = C.c_size_t(), C.c_size_t()
    self.check(lib().sfg_pgen_dims(self.h, img.ctypes.data_as(C.c_void_p), img.size, C.byref(ns), C.byref(nv)), "pgen_dims")
    rf = None if row_filter is None else np.ascontiguousarray(row_filter, dtype=np.uint8)
    out = np.empty((6, nv.value), dtype=np.uint32)
    self.check(lib().sfg_pgen_geno_counts(self.h, img.ctypes.data_as(C.c_void_p), img.size, None if rf is None else rf.ctypes.data_as(C.c_void_p),
                                          out.ctypes.data_as(C.c_void_p)), "pgen_geno_counts")
    return out


Context.geno_to_host = _ctx_geno_to_host
Context.geno_from_bed = _ctx_geno_from_bed
Context.geno_from_pgen = _ctx_geno_from_pgen
Context.pgen_geno_counts = _ctx_pgen_geno_counts


def _ctx_encode_vectors(self, values, level):
    values = np.ascontiguousarray(values, dtype=np.float64)
    nvec = values.shape[0]
    out = self.malloc(nvec * (level + 1) * self.N * 8)
    self.check(lib().sfg_encode_vectors_dev(self.h, values.ctypes.data_as(C.POINTER(C.c_double)), nvec, level, out), "encode_vectors")
    res = self.to_host(out, (nvec, level + 1, self.N), np.uint64)
    self.free(out)
    return res


Context.encode_vectors = _ctx_encode_vectors


# ---- resident products (device-level plumbing for the tests and bench.py)
class DevArray:
    """a device buffer with a numpy-like shape (uint64 words unless dtype says otherwise); freed with .free()"""

    def __init__(self, ctx, shape, dtype=np.uint64):
        self.ctx, self.shape, self.dtype = ctx, tuple(int(x) for x in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.p = ctx.malloc(max(self.nbytes, 8))

    @classmethod
    def from_host(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        d = cls(ctx, arr.shape, arr.dtype)
        ctx.check(lib().sfg_memcpy_h2d(ctx.h, d.p, arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")
        return d

    def host(self):
        return self.ctx.to_host(self.p, self.shape, self.dtype)

    def host_slice(self, index):
        """download self[index] for a leading-axis integer index (or tuple of leading indices)"""
        index = index if isinstance(index, tuple) else (index,)
        sub = self.shape[len(index):]
        off = 0
        for k, i in enumerate(index):
            off += int(i) * int(np.prod(self.shape[k + 1:]))
        out = np.empty(sub, dtype=self.dtype)
        src = C.c_void_p(self.p.value + off * self.dtype.itemsize)
        self.ctx.check(lib().sfg_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), src, out.nbytes), "d2h")
        return out

    def free(self):
        if self.p is not None:
            self.ctx.free(self.p)
            self.p = None


def _ctx_geno_upload(self, geno):
    geno = np.ascontiguousarray(geno, dtype=np.int8)
    g = C.c_void_p()
    self.check(lib().sfg_geno_upload(self.h, geno.ctypes.data_as(C.c_void_p), geno.shape[0], geno.shape[1], geno.shape[1], C.byref(g)), "geno_upload")
    return g


def _ctx_geno_free(self, g):
    lib().sfg_geno_free(self.h, g)


def _ctx_geno_create(self, nrow, ncol):
    g = C.c_void_p()
    self.check(lib().sfg_geno_create(self.h, nrow, ncol, C.byref(g)), "geno_create")
    return g


def _ctx_geno_write_rows(self, g, row0, rows, ld=None):
    """rows: int8 [nrows][>= ncol] (a C-contiguous chunk of whole rows; ld = its row stride)"""
    rows = np.ascontiguousarray(rows, dtype=np.int8)
    self.check(lib().sfg_geno_write_rows(self.h, g, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p), ld or rows.shape[1]), "geno_write_rows")


def _ctx_geno_compare_rows(self, g, flags, row0, rows):
    rows = np.ascontiguousarray(rows, dtype=np.int8)
    nd = np.zeros(1, dtype=np.uint64)
    self.check(lib().sfg_geno_compare_rows(self.h, g, flags, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p), rows.shape[1], p64(nd)), "geno_compare_rows")
    return int(nd[0])


def _ctx_fill_uniform_cts(self, nct, level, seed):
    d = DevArray(self, (nct, 2, level + 1, self.N))
    self.check(lib().sfg_fill_uniform_ct_dev(self.h, d.p, nct, level, seed), "fill_uniform_ct")
    return d


def _ctx_fill_geno(self, nrow, ncol, seed):
    d = DevArray(self, (nrow, ncol), np.int8)
    self.check(lib().sfg_fill_geno_dev(self.h, d.p, nrow, ncol, seed), "fill_geno")
    g = C.c_void_p()
    self.check(lib().sfg_geno_from_device(self.h, d.p, nrow, ncol, ncol, C.byref(g)), "geno_from_device")
    return d, g


def _ctx_matmul_resident(self, A_dev, s, in_level, max_level, g, flags=0, blk=None):
    """A_dev: DevArray [s][nbr][2][in_level+1][N]; returns DevArray out [s][m_out][2][max_level][N]"""
    nr, nc = C.c_size_t(), C.c_size_t()
    lib().sfg_geno_dims(g, C.byref(nr), C.byref(nc))
    lcol = nr.value if flags & SFG_TRANSPOSE else nc.value
    m_ct = (lcol - 1) // self.slots + 1
    if blk is None:
        out = DevArray(self, (s, m_ct, 2, max_level, self.N))
        self.check(lib().sfg_matmul_resident_dev(self.h, A_dev.p, s, in_level, max_level, g, flags, out.p), "matmul_resident")
    else:
        m_out = m_ct if flags & SFG_TRANSPOSE else blk[1] - blk[0]
        out = DevArray(self, (s, m_out, 2, max_level, self.N))
        self.check(lib().sfg_matmul_resident_range_dev(self.h, A_dev.p, s, in_level, max_level, g, flags, blk[0], blk[1], out.p), "matmul_resident_range")
    return out


def _ctx_matmul_accumulate(self, A_dev, s, in_level, max_level, g, flags, b0, b1, j0, j1, acc=None):
    d = 91
    accumulate = acc is not None
    if acc is None:
        acc = DevArray(self, (j1 - j0, d, s, 2, max_level, self.N))
    self.check(lib().sfg_matmul_accumulate_dev(self.h, A_dev.p, s, in_level, max_level, g, flags, b0, b1, j0, j1, int(accumulate), acc.p), "matmul_accumulate")
    return acc


def _ctx_matmul_finalize(self, acc, s, max_level, ncolb, g0, g1, out=None):
    accumulate = out is not None
    if out is None:
        out = DevArray(self, (s, ncolb, 2, max_level, self.N))
    self.check(lib().sfg_matmul_finalize_dev(self.h, acc.p, s, max_level, ncolb, g0, g1, int(accumulate), out.p), "matmul_finalize")
    return out


Context.geno_upload = _ctx_geno_upload
Context.geno_free = _ctx_geno_free
Context.geno_create = _ctx_geno_create
Context.geno_write_rows = _ctx_geno_write_rows
Context.geno_compare_rows = _ctx_geno_compare_rows
Context.fill_uniform_cts = _ctx_fill_uniform_cts
Context.fill_geno = _ctx_fill_geno
Context.matmul_resident = _ctx_matmul_resident
Context.matmul_accumulate = _ctx_matmul_accumulate
Context.matmul_finalize = _ctx_matmul_finalize


# ---- the multi-GPU engine (sfg_mgpu_*, mgpu.hip); plumbing for the tests and bench.py
class MultiGpu:
    """sfg_mgpu wrapper.  devices = [0, 1, ...] makes a single-process engine (one rank per entry; a repeated device selects the in-process `direct` transport);
    rank / world / uid join a multi-process world (one rank per process)."""

    def __init__(self, q, p, devices=None, scale=2.0 ** 34, logN=14, rank=None, world=None, uid=None, device=0, config=None):
        L = lib()
        self.q, self.p = list(q), list(p)
        self.nq, self.np_ = len(q), len(p)
        self.N, self.slots = 1 << logN, (1 << logN) // 2
        mods = np.array(self.q + self.p, dtype=np.uint64)
        h = C.c_void_p()
        cfgp = None if config is None else C.byref(config)
        if uid is None:
            devs = (C.c_int * len(devices))(*devices)
            rc = L.sfg_mgpu_create_ex(C.byref(h), devs, len(devices), logN, self.nq, self.np_, p64(mods), None, float(scale), cfgp)
        else:
            buf = (C.c_uint8 * 128).from_buffer_copy(bytes(uid))
            rc = L.sfg_mgpu_create_rank_ex(C.byref(h), device, rank, world, buf, logN, self.nq, self.np_, p64(mods), None, float(scale), cfgp)
        if rc:
            raise SfgError("sfg_mgpu_create: " + L.sfg_mgpu_last_error(None).decode())
        self.h = h
        self.world, self.nlocal = L.sfg_mgpu_world(h), L.sfg_mgpu_nlocal(h)
        self.transport = L.sfg_mgpu_transport(h).decode()
        self.ctx = []
        for i in range(self.nlocal):                        # borrowed contexts (owned by the engine): for device buffers and phase timers
            c = Context.__new__(Context)
            c.__dict__.update(q=self.q, p=self.p, nq=self.nq, np_=self.np_, N=self.N, slots=self.slots, beta=(self.nq + self.np_ - 1) // self.np_)
            c.h = C.c_void_p(L.sfg_mgpu_ctx(h, i))
            c.close = lambda: None
            self.ctx.append(c)
        self.ranks = [L.sfg_mgpu_rank(h, i) for i in range(self.nlocal)]

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        if lib().sfg_mgpu_unique_id(buf):
            raise SfgError("sfg_mgpu_unique_id: " + lib().sfg_mgpu_last_error(None).decode())
        return bytes(buf)

    def check(self, rc, what):
        if rc:
            raise SfgError(f"{what}: {lib().sfg_mgpu_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            for c in self.ctx:
                c.h = None
            lib().sfg_mgpu_destroy(self.h)
            self.h = None

    def sync(self):
        self.check(lib().sfg_mgpu_synchronize(self.h), "sfg_mgpu_synchronize")

    def fill_rotkeys_synthetic(self, rots_left, seed):
        arr = (C.c_int * len(rots_left))(*rots_left)
        self.check(lib().sfg_mgpu_fill_rotkeys_synthetic(self.h, arr, len(rots_left), seed), "sfg_mgpu_fill_rotkeys_synthetic")

    def load_rotkey(self, galois, key, montgomery=False):
        key = np.ascontiguousarray(key, dtype=np.uint64)
        self.check(lib().sfg_mgpu_load_rotkey(self.h, int(galois), p64(key), int(montgomery)), "sfg_mgpu_load_rotkey")

    def geno_upload(self, geno):
        geno = np.ascontiguousarray(geno, dtype=np.int8)
        g = C.c_void_p()
        self.check(lib().sfg_mgpu_geno_upload(self.h, geno.ctypes.data_as(C.c_void_p), geno.shape[0], geno.shape[1], geno.shape[1], C.byref(g)), "sfg_mgpu_geno_upload")
        return g

    def geno_create(self, nrow, ncol):
        g = C.c_void_p()
        self.check(lib().sfg_mgpu_geno_create(self.h, nrow, ncol, C.byref(g)), "sfg_mgpu_geno_create")
        return g

    def geno_write_rows(self, g, row0, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int8)
        self.check(lib().sfg_mgpu_geno_write_rows(self.h, g, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p), rows.shape[1]), "sfg_mgpu_geno_write_rows")

    def geno_compare_rows(self, g, flags, row0, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int8)
        nd = np.zeros(1, dtype=np.uint64)
        self.check(lib().sfg_mgpu_geno_compare_rows(self.h, g, flags, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p), rows.shape[1], p64(nd)), "sfg_mgpu_geno_compare_rows")
        return int(nd[0])

    def preflight(self, count_per_rank=4096):
        self.check(lib().sfg_mgpu_preflight(self.h, count_per_rank), "sfg_mgpu_preflight")

    def comm_info(self, local=0):
        n, r = C.c_int(), C.c_int()
        self.check(lib().sfg_mgpu_comm_info(self.h, local, C.byref(n), C.byref(r)), "sfg_mgpu_comm_info")
        return n.value, r.value

    def geno_synthetic(self, nrow, ncol, seed, packed=False):
        g = C.c_void_p()
        self.check(lib().sfg_mgpu_geno_synthetic(self.h, nrow, ncol, seed, int(packed), C.byref(g)), "sfg_mgpu_geno_synthetic")
        return g

    def geno_free(self, g):
        lib().sfg_mgpu_geno_free(self.h, g)

    def geno_blocks(self, g, local):
        b0, b1 = C.c_size_t(), C.c_size_t()
        lib().sfg_mgpu_geno_blocks(g, local, C.byref(b0), C.byref(b1))
        return b0.value, b1.value

    def matmul_dev(self, A, s, in_level, max_level, g, flags, out):
        """A / out: lists of DevArray, one per local rank"""
        n = self.nlocal
        pa = (C.c_void_p * n)(*[a.p for a in A])
        po = (C.c_void_p * n)(*[o.p for o in out])
        self.check(lib().sfg_mgpu_matmul_dev(self.h, pa, s, in_level, max_level, g, flags, po), "sfg_mgpu_matmul_dev")

    def matmul(self, A_host, s, in_level, max_level, g, flags=0):
        """host form: A_host [s][nbr or m_ct][2][in_level+1][N] -> out [s][m_ct or nbr][2][max_level][N]"""
        A_host = np.ascontiguousarray(A_host, dtype=np.uint64)
        nr, nc = C.c_size_t(), C.c_size_t()
        lib().sfg_mgpu_geno_dims(g, C.byref(nr), C.byref(nc))
        ncols = ((nr.value if flags & SFG_TRANSPOSE else nc.value) - 1) // self.slots + 1
        out = np.zeros((s, ncols, 2, max_level, self.N), dtype=np.uint64)
        self.check(lib().sfg_mgpu_matmul(self.h, p64(A_host), s, in_level, max_level, g, flags, p64(out)), "sfg_mgpu_matmul")
        return out
