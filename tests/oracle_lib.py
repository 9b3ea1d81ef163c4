"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package (sfgwas_amd/).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")


def build_oracle():
    if os.environ.get("SFG_ORACLE_SO"):            # e.g. a sanitizer build of the oracle (tools/oracle_asan.sh)
        return os.environ["SFG_ORACLE_SO"]
    src = os.path.join(_ROOT, "oracle", "sfgwas_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        L = _lib
        L.orc_ring_new.restype = C.c_void_p
        L.orc_ring_new.argtypes = [C.c_int, C.c_int, C.c_int, u64p, u64p]
        L.orc_ring_free.argtypes = [C.c_void_p]
        L.orc_ring_psi.restype = C.c_uint64
        L.orc_ring_psi.argtypes = [C.c_void_p, C.c_int]
        L.orc_mred_params.restype = C.c_uint64
        L.orc_mred_params.argtypes = [C.c_uint64]
        L.orc_bred_params.argtypes = [C.c_uint64, u64p]
        L.orc_mform.restype = C.c_uint64
        L.orc_mform.argtypes = [C.c_uint64, C.c_uint64, u64p]
        L.orc_mred.restype = C.c_uint64
        L.orc_mred.argtypes = [C.c_uint64] * 4
        L.orc_ntt.argtypes = [C.c_void_p, C.c_int, u64p]
        L.orc_intt.argtypes = [C.c_void_p, C.c_int, u64p]
        L.orc_mul_coeffs_and_add128.argtypes = [u64p, u64p, u64p, C.c_int]
        L.orc_reduce_and_add_uint128.argtypes = [u64p, u64p, C.c_uint64, C.c_uint64, C.c_int]
        L.orc_mform_vec.argtypes = [u64p, C.c_int, C.c_uint64]
        L.orc_canonical_reduce.argtypes = [u64p, C.c_int, C.c_uint64]
        L.orc_get_diag_bool.restype = C.c_int
        L.orc_get_diag_bool.argtypes = [C.c_int] * 4
        L.orc_get_diag.restype = C.c_int
        L.orc_get_diag.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int8), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_encode_coeffs.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int64), C.c_int]
        L.orc_encode_ntt.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_int, u64p, C.c_int]
        L.orc_rotkeys_new.restype = C.c_void_p
        L.orc_rotkeys_new.argtypes = [C.c_void_p]
        L.orc_rotkeys_free.argtypes = [C.c_void_p]
        L.orc_rotkeys_set.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.orc_rotkeys_beta.restype = C.c_int
        L.orc_rotkeys_beta.argtypes = [C.c_void_p]
        L.orc_galois_for_rotation.restype = C.c_uint64
        L.orc_galois_for_rotation.argtypes = [C.c_void_p, C.c_int]
        L.orc_automorphism_index.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]
        L.orc_keyswitch.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p, u64p]
        L.orc_rotate_left.restype = C.c_int
        L.orc_rotate_left.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u64p, C.c_int, u64p]
        L.orc_rotate_right.restype = C.c_int
        L.orc_rotate_right.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u64p, C.c_int, u64p]
        L.orc_apply_galois.restype = C.c_int
        L.orc_apply_galois.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u64p, C.c_uint64, u64p]
        L.orc_gen_secret.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int8)]
        L.orc_gen_rotkey.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_uint64, C.c_uint64, u64p]
        L.orc_encrypt_coeffs.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_int64), C.c_uint64, u64p]
        L.orc_decrypt_residues.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_int, u64p, u64p]
        L.orc_fill_uniform.argtypes = [C.c_void_p, C.c_int, C.c_uint64, u64p]
        L.orc_matmult4stream.restype = C.c_int
        L.orc_matmult4stream.argtypes = [C.c_void_p, C.c_void_p, C.c_double, u64p, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_int8), C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                         u64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_cpmult_acc_v2.argtypes = [u64p, u64p, u64p, C.c_int, C.c_int, C.c_int]
        L.orc_matmult_accumulate.restype = C.c_int
        L.orc_matmult_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_double, u64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int8), C.c_size_t, C.c_size_t,
                                             C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.POINTER(C.c_uint8)]
        L.orc_matmult_finalize.restype = C.c_int
        L.orc_matmult_finalize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, u64p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, u64p]
        L.orc_bed_decode.restype = C.c_int
        L.orc_bed_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_filter_matrix.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pgen_decode_codes.restype = C.c_int
        L.orc_pgen_decode_codes.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_pgen_to_int8.restype = C.c_int
        L.orc_pgen_to_int8.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pgen_geno_counts.restype = C.c_int
        L.orc_pgen_geno_counts.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_scale_up_exact.restype = C.c_uint64
        L.orc_scale_up_exact.argtypes = [C.c_double, C.c_double, C.c_uint64]
        L.orc_mul_const.argtypes = [C.c_void_p, C.c_int, u64p, C.c_double, u64p, C.POINTER(C.c_double)]
        L.orc_mul_const_and_add.argtypes = [C.c_void_p, C.c_int, u64p, C.c_double, C.c_double, u64p, C.POINTER(C.c_double)]
        L.orc_add_const.argtypes = [C.c_void_p, C.c_int, u64p, C.c_double, C.c_double, u64p]
        L.orc_add_plain.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.orc_ct_addsub.argtypes = [C.c_void_p, C.c_int, u64p, u64p, C.c_int, u64p]
        L.orc_mulrelin.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p, u64p]
        L.orc_mul_plain.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p]
        L.orc_rescale.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        L.orc_innersum_all.restype = C.c_int
        L.orc_innersum_all.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u64p, C.c_int, u64p]
        L.orc_gen_rlk.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_uint64, u64p]
        L.orc_bench_mac.restype = C.c_double
        L.orc_bench_mac.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_longlong)]
        L.orc_bench_mac_ref_layout.restype = C.c_double
        L.orc_bench_mac_ref_layout.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_diagcache_create.restype = C.c_void_p
        L.orc_diagcache_create.argtypes = [C.c_char_p, C.c_int]
        L.orc_diagcache_set_tables.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.orc_diagcache_write.restype = C.c_int
        L.orc_diagcache_write.argtypes = [C.c_void_p, C.POINTER(u64p), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_uint32]
        L.orc_diagcache_open.restype = C.c_void_p
        L.orc_diagcache_open.argtypes = [C.c_char_p, C.c_int]
        L.orc_diagcache_header.argtypes = [C.c_void_p, u64p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.orc_diagcache_read.restype = C.c_int
        L.orc_diagcache_read.argtypes = [C.c_void_p, C.POINTER(u64p), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
        L.orc_diagcache_close.argtypes = [C.c_void_p]
        L.orc_beaver_elem.argtypes = [C.c_int, C.c_int, u64p, u64p, u64p, u64p, u64p, u64p, C.c_size_t]
        L.orc_beaver_matmul.argtypes = [C.c_int, C.c_int, u64p, u64p, u64p, u64p, u64p, u64p, C.c_int, C.c_int, C.c_int]
        L.orc_ss_mask.argtypes = [C.c_int, u64p, u64p, u64p, u64p, u64p, u64p, C.c_size_t]
        L.orc_ss_hub_share.argtypes = [C.c_int, u64p, u64p, u64p, u64p, C.c_size_t]
        L.orc_bigint_to_rns.argtypes = [C.c_void_p, C.c_int, u64p, C.c_int, u64p]
        L.orc_refresh_gen_shares.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p, u64p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), u64p, u64p]
        L.orc_refresh_finish.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p, u64p, u64p]
        L.orc_refresh_gen_shares_scaled.argtypes = [C.c_void_p, C.c_int, u64p, C.c_double, C.c_double, u64p, u64p, u64p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), u64p, u64p]
        L.orc_refresh_finish_scaled.argtypes = [C.c_void_p, C.c_int, u64p, C.c_double, C.c_double, u64p, u64p, u64p, u64p]
        L.orc_sketch.argtypes = [C.POINTER(C.c_int8), C.c_size_t, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int8), C.c_int,
                                 C.POINTER(C.c_double), u64p, u64p]
    return _lib


def p64(a):
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    return a.ctypes.data_as(u64p)


def pd(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


def pi8(a):
    assert a.dtype == np.int8 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int8))


# PN14QP438-shaped modulus chain: defined once, in the product package (sfgwas_amd/params.py)
from sfgwas_amd.params import Q_PN14, P_PN14  # noqa: E402,F401


def small_primes(logN, bits, count, skip=0):
    """NTT-friendly primes == 1 mod 2N just below 2^bits."""
    from sympy import isprime
    M = 2 << logN
    out, x = [], (1 << bits) + 1 - M
    while len(out) < count + skip:
        if isprime(x):
            out.append(x)
        x -= M
    return out[skip:]


class Ring:
    def __init__(self, logN, q, p, psi=None):
        self.logN, self.N, self.nq, self.np_ = logN, 1 << logN, len(q), len(p)
        self.moduli = list(q) + list(p)
        mods = np.array(self.moduli, dtype=np.uint64)
        ps = None if psi is None else p64(np.array(psi, dtype=np.uint64))
        self.h = lib().orc_ring_new(logN, len(q), len(p), p64(mods), ps)
        assert self.h, "orc_ring_new failed"
        self.psi = [lib().orc_ring_psi(self.h, m) for m in range(len(self.moduli))]
        self.slots = self.N // 2
        self.beta = lib().orc_rotkeys_beta(self.h)

    def __del__(self):
        try:
            lib().orc_ring_free(self.h)
        except Exception:
            pass

    def ntt(self, mod, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().orc_ntt(self.h, mod, p64(a))
        return a

    def intt(self, mod, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().orc_intt(self.h, mod, p64(a))
        return a

    def encode_coeffs(self, v, scale, prec=0):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros(self.N, dtype=np.int64)
        lib().orc_encode_coeffs(self.h, pd(v), float(scale), out.ctypes.data_as(C.POINTER(C.c_int64)), prec)
        return out

    def encode_ntt(self, v, scale, nlev, prec=0):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros((nlev, self.N), dtype=np.uint64)
        lib().orc_encode_ntt(self.h, pd(v), float(scale), nlev, p64(out), prec)
        return out

    def galois(self, k):
        return lib().orc_galois_for_rotation(self.h, k)

    def gen_secret(self, seed):
        s = np.zeros(self.N, dtype=np.int8)
        lib().orc_gen_secret(self.h, seed, pi8(s))
        return s

    def key_words(self):
        return self.beta * 2 * len(self.moduli) * self.N

    def gen_rotkey(self, s, g, seed):
        k = np.zeros((self.beta, 2, len(self.moduli), self.N), dtype=np.uint64)
        lib().orc_gen_rotkey(self.h, pi8(s), g, seed, p64(k))
        return k

    def encrypt(self, s, level, m_coeffs, seed):
        ct = np.zeros((2, level + 1, self.N), dtype=np.uint64)
        m = np.ascontiguousarray(m_coeffs, dtype=np.int64)
        lib().orc_encrypt_coeffs(self.h, pi8(s), level, m.ctypes.data_as(C.POINTER(C.c_int64)), seed, p64(ct))
        return ct

    def decrypt_residues(self, s, level, ct):
        out = np.zeros((level + 1, self.N), dtype=np.uint64)
        ct = np.ascontiguousarray(ct)
        lib().orc_decrypt_residues(self.h, pi8(s), level, p64(ct), p64(out))
        return out

    def fill_uniform(self, level, seed):
        ct = np.zeros((2, level + 1, self.N), dtype=np.uint64)
        lib().orc_fill_uniform(self.h, level, seed, p64(ct))
        return ct


class RotKeys:
    def __init__(self, ring):
        self.ring = ring
        self.h = lib().orc_rotkeys_new(ring.h)
        self.keys = {}

    def __del__(self):
        try:
            lib().orc_rotkeys_free(self.h)
        except Exception:
            pass

    def add(self, g, key):
        key = np.ascontiguousarray(key, dtype=np.uint64)
        self.keys[g] = key
        lib().orc_rotkeys_set(self.h, g, p64(key))

    def gen_for_rotations(self, s, rots_left, seed=77):
        for k in rots_left:
            g = self.ring.galois(k)
            if g not in self.keys:
                self.add(g, self.ring.gen_rotkey(s, g, seed + g))


def rotate_left(ring, keys, level, ct, k):
    out = np.zeros_like(ct)
    rc = lib().orc_rotate_left(ring.h, keys.h, level, p64(np.ascontiguousarray(ct)), k, p64(out))
    assert rc == 0, "missing rotation key"
    return out


def rotate_right(ring, keys, level, ct, k):
    out = np.zeros_like(ct)
    rc = lib().orc_rotate_right(ring.h, keys.h, level, p64(np.ascontiguousarray(ct)), k, p64(out))
    assert rc == 0, "missing rotation key"
    return out


def matmult4stream(ring, keys, scale, A, in_level, max_level, geno, compute_sqsum=False, square=False, enc_prec=0):
    """A: [s][nbr][2][in_level+1][N] uint64; geno: [nrow][ncol] int8. Returns (out[s][m_ct][2][L][N], sum, sqsum)."""
    s, nbr = A.shape[0], A.shape[1]
    nrow, ncol = geno.shape
    m_ct = (ncol - 1) // ring.slots + 1
    assert nbr == (nrow - 1) // ring.slots + 1
    out = np.zeros((s, m_ct, 2, max_level, ring.N), dtype=np.uint64)
    sm = np.zeros(ncol, dtype=np.float64) if compute_sqsum else None
    sq = np.zeros(ncol, dtype=np.float64) if compute_sqsum else None
    A = np.ascontiguousarray(A)
    geno = np.ascontiguousarray(geno)
    rc = lib().orc_matmult4stream(ring.h, keys.h, float(scale), p64(A), s, in_level, max_level, pi8(geno), nrow, ncol,
                                  int(compute_sqsum), int(square), enc_prec, p64(out),
                                  pd(sm) if compute_sqsum else None, pd(sq) if compute_sqsum else None)
    assert rc == 0, "oracle matmult failed (missing rotation key?)"
    return out, sm, sq


def needed_rotations(slots, nrow, ncol):
    """Left-rotation amounts MatMult4Stream needs keys for: baby steps 1..d-1 and giant steps d*g (matmult.go:1375,1476)."""
    import math
    d = int(math.ceil(math.sqrt(slots)))
    ks = set()
    for b in range(1, d):
        ks.add(b)
    for g in range(1, d):
        if g * d < slots:
            ks.add(g * d)
    return sorted(ks), d


def splitmix64_array(seed, n):
    """counter-mode splitmix64: element i = mix(seed + (i+1)*golden)"""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


# ---- collective bootstrap, local work (oracle restatement of lattigo v2.1.0 dckks/refresh.go; parity unpinned)
def bigints_to_limbs(vals, W):
    """signed Python ints -> [len][W] two's-complement little-endian uint64 limbs"""
    out = np.zeros((len(vals), W), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        u = int(v) & ((1 << (64 * W)) - 1)
        for w in range(W):
            out[i, w] = (u >> (64 * w)) & mask
    return out


def pi32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def refresh_gen_shares(ring, level, ct, sk, crs, mask_limbs, e0, e1):
    h0 = np.zeros((level + 1, ring.N), dtype=np.uint64)
    h1 = np.zeros((ring.nq, ring.N), dtype=np.uint64)
    lib().orc_refresh_gen_shares(ring.h, level, p64(np.ascontiguousarray(ct)), p64(np.ascontiguousarray(sk)), p64(np.ascontiguousarray(crs)),
                                 p64(np.ascontiguousarray(mask_limbs)), mask_limbs.shape[1], pi32(np.ascontiguousarray(e0)), pi32(np.ascontiguousarray(e1)),
                                 p64(h0), p64(h1))
    return h0, h1


def refresh_finish(ring, level, ct, h0agg, h1agg, crs):
    out = np.zeros((2, ring.nq, ring.N), dtype=np.uint64)
    lib().orc_refresh_finish(ring.h, level, p64(np.ascontiguousarray(ct)), p64(np.ascontiguousarray(h0agg)), p64(np.ascontiguousarray(h1agg)),
                             p64(np.ascontiguousarray(crs)), p64(out))
    return out


def refresh_gen_shares_scaled(ring, level, ct, ct_scale, target_scale, sk, crs, mask_limbs, e0, e1):
    h0 = np.zeros((level + 1, ring.N), dtype=np.uint64)
    h1 = np.zeros((ring.nq, ring.N), dtype=np.uint64)
    lib().orc_refresh_gen_shares_scaled(ring.h, level, p64(np.ascontiguousarray(ct)), float(ct_scale), float(target_scale), p64(np.ascontiguousarray(sk)),
                                        p64(np.ascontiguousarray(crs)), p64(np.ascontiguousarray(mask_limbs)), mask_limbs.shape[1],
                                        pi32(np.ascontiguousarray(e0)), pi32(np.ascontiguousarray(e1)), p64(h0), p64(h1))
    return h0, h1


def refresh_finish_scaled(ring, level, ct, ct_scale, target_scale, h0agg, h1agg, crs):
    out = np.zeros((2, ring.nq, ring.N), dtype=np.uint64)
    lib().orc_refresh_finish_scaled(ring.h, level, p64(np.ascontiguousarray(ct)), float(ct_scale), float(target_scale), p64(np.ascontiguousarray(h0agg)),
                                    p64(np.ascontiguousarray(h1agg)), p64(np.ascontiguousarray(crs)), p64(out))
    return out


def secret_ntt(ring, s_coeff):
    """ternary secret coefficients -> NTT-domain canonical rows [nq][N]"""
    rows = np.zeros((ring.nq, ring.N), dtype=np.uint64)
    for j in range(ring.nq):
        q = ring.moduli[j]
        rows[j] = ring.ntt(j, np.where(s_coeff < 0, q - 1, s_coeff.astype(np.int64)).astype(np.uint64))
    return rows


# ---- PLINK 2 .pgen (oracle side)
def pgen_dims(img):
    return int(img[7:11].view("<u4")[0]), int(img[3:7].view("<u4")[0])          # samples, variants


def pgen_geno_counts(img, row_filter=None):
    ns, nv = pgen_dims(img)
    out = np.zeros((6, nv), dtype=np.uint32)
    rf = None if row_filter is None else np.ascontiguousarray(row_filter, dtype=np.uint8)
    rc = lib().orc_pgen_geno_counts(img.ctypes.data, img.size, None if rf is None else rf.ctypes.data, out.ctypes.data)
    if rc:
        raise ValueError(f"orc_pgen_geno_counts failed ({rc})")
    return out


def pgen_to_int8(img, v0=0, v1=None, row_filter=None, col_filter=None):
    ns, nv = pgen_dims(img)
    v1 = nv if v1 is None else v1
    rf = None if row_filter is None else np.ascontiguousarray(row_filter, dtype=np.uint8)
    cf = None if col_filter is None else np.ascontiguousarray(col_filter, dtype=np.uint8)
    nr = ns if rf is None else int(np.count_nonzero(rf))
    nc = v1 - v0 if cf is None else int(np.count_nonzero(cf))
    out = np.empty((nr, nc), dtype=np.int8)
    rc = lib().orc_pgen_to_int8(img.ctypes.data, img.size, v0, v1, None if rf is None else rf.ctypes.data, None if cf is None else cf.ctypes.data, out.ctypes.data)
    if rc:
        raise ValueError(f"orc_pgen_to_int8 failed ({rc})")
    return out
