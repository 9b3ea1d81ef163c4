"""Run in the BUILD container (where /root/reference exists): copies the reference's shipped example data that pins the .pgen decoder
into tests/golden/example_party1/ -

  geno/chr1..22.pgen            the 22 PGEN files of party 1 (1000 samples x 100 000 SNPs; config/configLocal.Party1.toml:6), data files
  all.gcount.transpose.bin      the reference-held fixture (configLocal.Party1.toml:15): 6 x 100 000 uint32 from plink2 --geno-counts
                                (scripts/preprocessing/computeGenoCounts.py) - the expected output
  chrom_sizes.txt               SNPs per chromosome file

and, for party 2, only the expected output (example_party2_gcount.bin) plus a SHA-256 of each of its .pgen inputs: its decoder check
runs where the reference tree is present (tests/test_pgen.py skips it elsewhere).  Data only - no reference source text is copied."""
import hashlib
import json
import os
import shutil

REF = "/root/reference/example_data"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    dst = os.path.join(HERE, "example_party1")
    os.makedirs(os.path.join(dst, "geno"), exist_ok=True)
    for c in range(1, 23):
        shutil.copyfile(f"{REF}/party1/geno/chr{c}.pgen", f"{dst}/geno/chr{c}.pgen")
    shutil.copyfile(f"{REF}/party1/all.gcount.transpose.bin", f"{dst}/all.gcount.transpose.bin")
    shutil.copyfile(f"{REF}/party1/chrom_sizes.txt", f"{dst}/chrom_sizes.txt")
    shutil.copyfile(f"{REF}/party2/all.gcount.transpose.bin", os.path.join(HERE, "example_party2_gcount.bin"))
    sums = {f"chr{c}.pgen": hashlib.sha256(open(f"{REF}/party2/geno/chr{c}.pgen", "rb").read()).hexdigest() for c in range(1, 23)}
    json.dump(sums, open(os.path.join(HERE, "example_party2_pgen_sha256.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
