"""PLINK 2 .pgen decoding, pinned by a fixture the REFERENCE ITSELF holds: example_data/party{1,2}/all.gcount.transpose.bin
(config/configLocal.Party1.toml:15; made by scripts/preprocessing/computeGenoCounts.py with plink2 --geno-counts from the very .pgen
files under example_data/party*/geno/).  A decoder whose per-SNP hom-ref / het / hom-alt / missing counts reproduce that file for all
100 000 SNPs decodes the reference's data as plink2 does.  CPU part: the oracle restatement (oracle/sfgwas_oracle.c orc_pgen_*).
The record types the reference's data does not contain (LD-compressed, pure difflists) are checked against an independent writer of
the published specification (tests/pgen_writer.py) - self-consistency, parity unpinned for those."""
import hashlib
import json
import os
import numpy as np
import pytest

import oracle_lib as ol
import pgen_writer as pw

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF2 = "/root/reference/example_data/party2"


def party1_images():
    return [np.fromfile(os.path.join(GOLD, "example_party1", "geno", f"chr{c}.pgen"), dtype=np.uint8) for c in range(1, 23)]


def test_oracle_geno_counts_of_party1_equal_the_reference_fixture_for_all_100000_snps():
    ref = np.fromfile(os.path.join(GOLD, "example_party1", "all.gcount.transpose.bin"), dtype=np.uint32).reshape(6, -1)
    sizes = [int(x) for x in open(os.path.join(GOLD, "example_party1", "chrom_sizes.txt")).read().split()]
    imgs = party1_images()
    assert [ol.pgen_dims(i)[1] for i in imgs] == sizes and all(ol.pgen_dims(i)[0] == 1000 for i in imgs)
    got = np.concatenate([ol.pgen_geno_counts(i) for i in imgs], axis=1)
    assert got.shape == ref.shape == (6, 100_000)
    assert np.array_equal(got, ref)
    assert np.array_equal(got.sum(0), np.full(100_000, 1000))


def test_oracle_geno_counts_of_party2_equal_the_reference_fixture():
    if not os.path.isdir(REF2):
        pytest.skip("party 2's .pgen inputs are read from the reference tree (build container only); its expected output is committed")
    ref = np.fromfile(os.path.join(GOLD, "example_party2_gcount.bin"), dtype=np.uint32).reshape(6, -1)
    sums = json.load(open(os.path.join(GOLD, "example_party2_pgen_sha256.json")))
    cols = []
    for c in range(1, 23):
        raw = open(f"{REF2}/geno/chr{c}.pgen", "rb").read()
        assert hashlib.sha256(raw).hexdigest() == sums[f"chr{c}.pgen"]
        cols.append(ol.pgen_geno_counts(np.frombuffer(raw, dtype=np.uint8).copy()))
    assert np.array_equal(np.concatenate(cols, axis=1), ref)


def test_oracle_int8_matrix_is_the_alt_allele_count_and_honours_keep_and_extract():
    img = party1_images()[21]
    ns, nv = ol.pgen_dims(img)
    full = ol.pgen_to_int8(img)
    assert full.shape == (ns, nv) and full.min() >= 0 and full.max() == 2
    ref = np.fromfile(os.path.join(GOLD, "example_party1", "all.gcount.transpose.bin"), dtype=np.uint32).reshape(6, -1)[:, -nv:]
    assert np.array_equal((full == 1).sum(0), ref[1]) and np.array_equal((full == 2).sum(0), ref[2])
    rnd = np.random.default_rng(3)
    rf, cf = rnd.random(ns) < 0.7, rnd.random(300) < 0.5
    sub = ol.pgen_to_int8(img, 1000, 1300, rf, cf)
    assert np.array_equal(sub, full[rf][:, 1000:1300][:, cf])
    kept = ol.pgen_geno_counts(img, rf)
    assert np.array_equal(kept[0], (full[rf] == 0).sum(0)) and np.array_equal(kept[5], np.zeros(nv))


@pytest.mark.parametrize("ns,nv,wmode,seed", [(1000, 300, 0, 1), (257, 200, 4, 2), (70001, 40, 6, 3), (5, 64, 1, 4), (4096, 150, 7, 5)])
def test_oracle_decodes_every_record_type_written_by_the_independent_writer(ns, nv, wmode, seed):
    """types 0, 1, 2, 3, 4, 6, 7; 1-, 2- and 3-byte sample ids; difflists of several 64-entry groups; 4- and 8-bit vrtype tables"""
    codes, vrt = pw.synthetic(nv, ns, seed)
    assert len(set(vrt.tolist())) >= 5
    img = pw.write_pgen(codes, vrt, wmode=wmode)
    got = ol.pgen_to_int8(img)
    want = np.where(codes == 3, -1, codes).astype(np.int8).T
    assert np.array_equal(got, want)
    cnt = ol.pgen_geno_counts(img)
    assert np.array_equal(cnt[5], (codes == 3).sum(1)) and np.array_equal(cnt[1], (codes == 1).sum(1))
    # a window that starts inside an LD-compressed run needs the base in front of it
    ld = [v for v in range(1, nv) if vrt[v] in (2, 3)]
    if ld:
        v0 = ld[len(ld) // 2]
        assert np.array_equal(ol.pgen_to_int8(img, v0, min(nv, v0 + 7)), want[:, v0:v0 + 7])


def test_oracle_rejects_malformed_files():
    codes, vrt = pw.synthetic(20, 100, 9)
    img = pw.write_pgen(codes, vrt, wmode=5)
    bad = img.copy(); bad[0] = 0
    with pytest.raises(ValueError):
        ol.pgen_to_int8(bad)
    with pytest.raises(ValueError):
        ol.pgen_to_int8(img[:len(img) - 5].copy())
    multi = pw.write_pgen(codes, np.zeros(20, dtype=np.uint8), wmode=5, extra_vrtype_bits=8)
    with pytest.raises(ValueError):
        ol.pgen_to_int8(multi)
