"""Independent pure-Python (big-int / mpmath / numpy) statements of the mathematics the oracle must satisfy.

Used only to PIN the C oracle at small ring sizes (tests/test_oracle_pinning.py) and to decode
decrypted results in the end-to-end checks.  Nothing here is derived from the C oracle's code.
"""
import math
import numpy as np


def brev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def ntt_direct(p, psi, q, logN):
    """out[i] = p(psi^(2*brev(i)+1)) mod q — the defining property of lattigo's NTT layout."""
    N = 1 << logN
    out = []
    for i in range(N):
        x = pow(psi, 2 * brev(i, logN) + 1, q)
        acc, xp = 0, 1
        for c in range(N):
            acc = (acc + int(p[c]) * xp) % q
            xp = xp * x % q
        out.append(acc)
    return out


def negacyclic_mul(a, b, q):
    N = len(a)
    out = [0] * N
    for i in range(N):
        for j in range(N):
            k = i + j
            v = int(a[i]) * int(b[j])
            if k >= N:
                out[k - N] = (out[k - N] - v) % q
            else:
                out[k] = (out[k] + v) % q
    return out


def automorphism_coeffs(p, g, q):
    """p(X) -> p(X^g) in the coefficient domain mod (X^N + 1, q)."""
    N = len(p)
    out = [0] * N
    for i in range(N):
        e = (i * g) % (2 * N)
        if e < N:
            out[e] = (out[e] + int(p[i])) % q
        else:
            out[e - N] = (out[e - N] - int(p[i])) % q
    return out


def encode_exact(v, N, scale, dps=80):
    """round(scale * sigma^-1(v)) with mpmath: w_c = (1/n) sum_t v_t zeta^(-5^t c); coeff[c]=Re, coeff[c+n]=Im."""
    import mpmath as mp
    mp.mp.dps = dps
    n, M = N // 2, 2 * N
    rot = [pow(5, t, M) for t in range(n)]
    out = [0] * N
    for c in range(n):
        acc = mp.mpc(0)
        for t in range(n):
            acc += mp.mpf(float(v[t])) * mp.expjpi(mp.mpf(-2 * ((rot[t] * c) % M)) / M)
        acc = acc / n * mp.mpf(scale)

        def rnd(x):
            return int(mp.floor(x + mp.mpf(0.5))) if x >= 0 else -int(mp.floor(-x + mp.mpf(0.5)))
        out[c] = rnd(acc.real)
        out[c + n] = rnd(acc.imag)
    return out


def decode(coeffs_float, N):
    """slot values v_t = sum_c w_c zeta^(5^t c), w_c = p_c + i p_{c+n}; coeffs already divided by scale."""
    n, M = N // 2, 2 * N
    w = np.asarray(coeffs_float[:n], dtype=np.complex128) + 1j * np.asarray(coeffs_float[n:], dtype=np.complex128)
    c = np.arange(n)
    tw = np.exp(2j * np.pi * c / M)
    V = np.fft.ifft(w * tw) * n          # V[m] = sum_c w_c zeta^c omega_n^(m c)
    idx = np.zeros(n, dtype=np.int64)
    g = 1
    for t in range(n):
        idx[t] = ((g - 1) // 4) % n
        g = g * 5 % M
    return V[idx]


def crt_centered(residues, moduli):
    """residues: [nmod][N] ints -> list of centered big ints"""
    Q = 1
    for q in moduli:
        Q *= q
    N = len(residues[0])
    out = []
    coef = []
    for q in moduli:
        Qi = Q // q
        coef.append(Qi * pow(Qi, -1, q))
    for x in range(N):
        v = sum(int(residues[m][x]) * coef[m] for m in range(len(moduli))) % Q
        if v > Q // 2:
            v -= Q
        out.append(v)
    return out


def get_diag_bruteforce(X, dim, shift):
    """dst[j] = X[(j+shift) mod dim][j] where defined else 0 (matmult.go:636-664 with index = -shift)."""
    r, c = X.shape
    dst = np.zeros(dim)
    for j in range(dim):
        i = (j + shift) % dim
        if i < r and j < c:
            dst[j] = X[i, j]
    return dst


def bsgs_d(slots):
    return int(math.ceil(math.sqrt(float(slots))))
