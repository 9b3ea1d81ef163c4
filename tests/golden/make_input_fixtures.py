"""Generates tests/golden/input_formats.npz by RUNNING the reference's own converters
(/root/reference/scripts/plinkBedToBinary.py, filterMatrix.py, transposeMatrix.py, mergeMatrices.py) on small random
inputs.  Only runs in the build container (the reference tree does not travel); the .npz holds inputs and the
outputs those scripts produced - data, not source.

    python tests/golden/make_input_fixtures.py
"""
import os, subprocess, sys, tempfile
import numpy as np

REF = "/root/reference/scripts"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "input_formats.npz")


def run(script, *args):
    subprocess.run([sys.executable, os.path.join(REF, script)] + [str(a) for a in args], check=True, stdout=subprocess.DEVNULL)


def main():
    rnd = np.random.default_rng(20240607)
    fx = {}
    with tempfile.TemporaryDirectory() as d:
        cases = [(5, 7), (13, 4), (64, 64), (257, 131), (1000, 77), (3, 1)]
        fx["bed_cases"] = np.array(cases, dtype=np.int64)
        for k, (ns, nv) in enumerate(cases):
            bps = (ns + 3) // 4
            payload = rnd.integers(0, 256, nv * bps, dtype=np.uint8)
            bed = np.concatenate([np.array([0x6C, 0x1B, 0x01], dtype=np.uint8), payload])
            bed.tofile(f"{d}/in.bed")
            run("plinkBedToBinary.py", f"{d}/in.bed", ns, nv, f"{d}/out.bin")
            geno = np.fromfile(f"{d}/out.bin", dtype=np.int8).reshape(ns, nv)
            fx[f"bed_{k}"] = bed
            fx[f"geno_{k}"] = geno
            rf = (rnd.random(ns) < 0.7).astype(np.uint8); cf = (rnd.random(nv) < 0.6).astype(np.uint8)
            if not rf.any(): rf[0] = 1
            if not cf.any(): cf[0] = 1
            rf.tofile(f"{d}/rf.bin"); cf.tofile(f"{d}/cf.bin")
            run("filterMatrix.py", f"{d}/out.bin", ns, nv, f"{d}/rf.bin", f"{d}/cf.bin", f"{d}/filt.bin")
            fx[f"rowfilt_{k}"] = rf; fx[f"colfilt_{k}"] = cf
            fx[f"filtered_{k}"] = np.fromfile(f"{d}/filt.bin", dtype=np.int8).reshape(int(rf.sum()), int(cf.sum()))
            run("transposeMatrix.py", f"{d}/out.bin", ns, nv, f"{d}/t.bin")
            fx[f"transposed_{k}"] = np.fromfile(f"{d}/t.bin", dtype=np.int8).reshape(nv, ns)
        # mergeMatrices.py: column-wise concatenation of <prefix>.<i>.bin
        nrows, widths = 9, [4, 1, 6]
        parts = [rnd.integers(-1, 3, (nrows, w)).astype(np.int8) for w in widths]
        for i, p in enumerate(parts):
            p.tofile(f"{d}/m.{i}.bin")
        open(f"{d}/ncols.txt", "w").write("\n".join(str(w) for w in widths) + "\n")
        run("mergeMatrices.py", f"{d}/m", nrows, f"{d}/ncols.txt", f"{d}/merged.bin")
        for i, p in enumerate(parts):
            fx[f"merge_part_{i}"] = p
        fx["merged"] = np.fromfile(f"{d}/merged.bin", dtype=np.int8).reshape(nrows, sum(widths))
    np.savez_compressed(OUT, **fx)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
