"""Test-side WRITER of PLINK 2 .pgen files (storage mode 0x10), written independently of the decoders from the published PGEN
specification (plink-ng 2.0 pgenlib, pgen_spec): lets the tests produce every main-track record type - the reference's example data only
contains types 0 and 1 - with difflists of several groups, all header width modes and LD-compressed runs.  Self-consistency only: the
types the reference data does not use stay "parity unpinned" (DESIGN.md)."""
import numpy as np


def varint(x):
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def pack2(codes):
    c = np.zeros((len(codes) + 3) // 4 * 4, dtype=np.uint8)
    c[:len(codes)] = codes
    c = c.reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8).tobytes()


def id_bytes(ns):
    return (ns.bit_length() - 1) // 8 + 1


def difflist(ids, vals, ns):
    """ids ascending sample indices, vals their 2-bit values"""
    n = len(ids)
    out = bytearray(varint(n))
    if n == 0:
        return bytes(out)
    idb = id_bytes(ns)
    groups = [(ids[k:k + 64], vals[k:k + 64]) for k in range(0, n, 64)]
    deltas = []
    for gi, _ in groups:
        d = bytearray()
        for a, b in zip(gi[:-1], gi[1:]):
            d += varint(int(b) - int(a))
        deltas.append(bytes(d))
    for gi, _ in groups:
        out += int(gi[0]).to_bytes(idb, "little")
    for d in deltas[:-1]:
        assert 0 <= len(d) - 63 < 256
        out.append(len(d) - 63)
    out += pack2(np.asarray(vals, dtype=np.uint8))
    for d in deltas:
        out += d
    return bytes(out)


def invert(codes):
    c = codes.copy()
    c[codes == 0] = 2
    c[codes == 2] = 0
    return c


def record(codes, vrtype, base, ns):
    codes = np.asarray(codes, dtype=np.uint8)
    if vrtype == 0:
        return pack2(codes)
    if vrtype == 1:
        cnt = np.bincount(codes, minlength=4)
        lo, hi = sorted(np.argsort(-cnt, kind="stable")[:2].tolist())
        bits = np.zeros((ns + 7) // 8 * 8, dtype=np.uint8)
        bits[:ns] = (codes == hi)
        other = np.nonzero((codes != lo) & (codes != hi))[0]
        return bytes([lo * 4 + (hi - lo)]) + np.packbits(bits, bitorder="little").tobytes() + difflist(other, codes[other], ns)
    if vrtype in (4, 6, 7):
        fill = vrtype & 3
        other = np.nonzero(codes != fill)[0]
        return difflist(other, codes[other], ns)
    if vrtype in (2, 3):
        tgt = invert(codes) if vrtype == 3 else codes
        other = np.nonzero(tgt != base)[0]
        return difflist(other, tgt[other], ns)
    raise ValueError(vrtype)


def write_pgen(codes, vrtypes, wmode=7, extra_vrtype_bits=0):
    """codes [nv][ns] in {0,1,2,3}; vrtypes[nv] main-track types; wmode = low nibble of the header control byte"""
    codes = np.asarray(codes, dtype=np.uint8)
    nv, ns = codes.shape
    recs, base = [], None
    for v in range(nv):
        t = int(vrtypes[v])
        recs.append(record(codes[v], t, base, ns))
        if t not in (2, 3):
            base = codes[v]
    lb = (wmode & 3) + 1
    assert all(len(r) < (1 << (8 * lb)) for r in recs)
    nblk = (nv + 65535) // 65536
    hdr_len = 12 + 8 * nblk
    per_blk = []
    for b in range(nblk):
        v0, v1 = b * 65536, min(nv, (b + 1) * 65536)
        vt = [int(vrtypes[v]) | extra_vrtype_bits for v in range(v0, v1)]
        if wmode < 4:
            vt = vt + [0] * (len(vt) & 1)
            vb = bytes(vt[k] | (vt[k + 1] << 4) for k in range(0, len(vt), 2))
        else:
            vb = bytes(vt)
        lens = b"".join(len(recs[v]).to_bytes(lb, "little") for v in range(v0, v1))
        per_blk.append(vb + lens)
        hdr_len += len(vb) + len(lens)
    out = bytearray([0x6C, 0x1B, 0x10]) + nv.to_bytes(4, "little") + ns.to_bytes(4, "little") + bytes([wmode])
    pos = hdr_len
    for b in range(nblk):
        out += pos.to_bytes(8, "little")
        pos += sum(len(recs[v]) for v in range(b * 65536, min(nv, (b + 1) * 65536)))
    for pb in per_blk:
        out += pb
    assert len(out) == hdr_len
    for r in recs:
        out += r
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def synthetic(nv, ns, seed, types=(0, 1, 2, 3, 4, 6, 7)):
    """a genotype matrix whose variants suit the record types drawn for them (sparse rows for difflists, correlated rows for LD)"""
    rnd = np.random.default_rng(seed)
    codes = np.zeros((nv, ns), dtype=np.uint8)
    vrt = np.zeros(nv, dtype=np.uint8)
    base = None
    for v in range(nv):
        t = int(rnd.choice(types)) if v else int(rnd.choice([x for x in types if x not in (2, 3)] or [0]))
        if t in (2, 3) and base is None:
            t = 0
        if t == 0:
            row = rnd.integers(0, 4, ns)
        elif t == 1:
            row = rnd.choice([0, 1, 2, 3], ns, p=[0.55, 0.38, 0.05, 0.02])
            row = rnd.permutation(4)[row]
        elif t in (4, 6, 7):
            row = np.full(ns, t & 3)
            k = int(rnd.integers(0, max(2, ns // 6)))
            idx = rnd.choice(ns, k, replace=False)
            row[idx] = rnd.integers(0, 4, k)
        else:
            row = base.copy()
            k = int(rnd.integers(0, max(2, ns // 5)))
            idx = rnd.choice(ns, k, replace=False)
            row[idx] = rnd.integers(0, 4, k)
            if t == 3:
                row = invert(row.astype(np.uint8))
        codes[v] = row
        vrt[v] = t
        if t not in (2, 3):
            base = codes[v].copy()
    return codes, vrt
