#!/usr/bin/env python3
"""lattigo_vectors.bin (written by main.go) -> tests/golden/lattigo_vectors.npz

    python tools/lattigo_fixtures/to_npz.py lattigo_vectors.bin tests/golden/lattigo_vectors.npz

The pn14 arrays are stored as SHA-256 digests + head samples (a PN14 plaintext is 786 KB); the small-ring arrays in full."""
import hashlib
import struct
import sys

import numpy as np


def read(path):
    arrs = {}
    with open(path, "rb") as f:
        while True:
            h = f.read(4)
            if len(h) < 4:
                break
            (nl,) = struct.unpack("<I", h)
            name = f.read(nl).decode()
            dtype, ndim = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<%dQ" % ndim, f.read(8 * ndim))
            n = int(np.prod(dims))
            a = np.frombuffer(f.read(8 * n), dtype="<u8").reshape(dims)
            arrs[name] = a.view(np.float64) if dtype == 1 else a
    return arrs


def main():
    arrs = read(sys.argv[1])
    out = {}
    for name, a in arrs.items():
        if name.startswith("pn14.") and a.size > 4096:
            out[name + ".sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
            out[name + ".head"] = a.reshape(-1)[:64].copy()
            out[name + ".shape"] = np.array(a.shape, dtype=np.int64)
        else:
            out[name] = a
    np.savez_compressed(sys.argv[2], **out)
    print("wrote", sys.argv[2], "with", len(out), "arrays")


if __name__ == "__main__":
    main()
