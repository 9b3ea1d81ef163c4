"""Summarise tools/r4_pmc_mac_i8.sh: per launch of the int8 MAC kernel, exact fabric-side read bytes and the L2 hit rate, for the three launch variants."""
import csv, glob, json, sys, collections

root = sys.argv[1]
N, H = 16384, 8192


def load(d):
    fs = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    rows = collections.defaultdict(dict)           # dispatch id -> {kernel, counters}
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_mac_i8" not in k and "k_i8_pack_pt_digits" not in k and "k_ntt_half3" not in k:
            continue
        e = rows[int(r["Dispatch_Id"])]
        e["kernel"] = k
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return [rows[i] for i in sorted(rows)]


out = {}
for v in ("default", "wg1", "lds"):
    try:
        p1, p2 = load(f"{root}/{v}_P1"), load(f"{root}/{v}_P2")
    except (IndexError, FileNotFoundError):
        continue
    res = {}
    for kname in ("k_mac_i8", "k_i8_pack_pt_digits", "k_ntt_half3"):
        a = [e for e in p1 if kname in e["kernel"]]
        b = [e for e in p2 if kname in e["kernel"]]
        if not a:
            continue
        by = [32 * e.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * e.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * e.get("TCC_EA0_RDREQ_128B_sum", 0) for e in a]
        if kname == "k_mac_i8":                     # the two largest launches are Q'*X^T (K = 1183 at c2); the rest Q*X (K = 182)
            order = sorted(range(len(by)), key=lambda i: -by[i])
            big = order[:2]
        else:
            big = list(range(len(by)))
        sel = lambda lst, key: sum(lst[i].get(key, 0) for i in big) / len(big)
        res[kname] = {"launches_averaged": len(big), "read_bytes_per_launch": sum(by[i] for i in big) / len(big),
                      "rdreq_32B": sel(a, "TCC_EA0_RDREQ_32B_sum"), "rdreq_64B": sel(a, "TCC_EA0_RDREQ_64B_sum"), "rdreq_128B": sel(a, "TCC_EA0_RDREQ_128B_sum"),
                      "rdreq_total": sel(a, "TCC_EA0_RDREQ_sum")}
        if b:
            bb = sorted(range(len(b)), key=lambda i: -b[i].get("TCC_REQ_sum", 0))[:2] if kname == "k_mac_i8" else list(range(len(b)))
            hit = sum(b[i].get("TCC_HIT_sum", 0) for i in bb); miss = sum(b[i].get("TCC_MISS_sum", 0) for i in bb)
            res[kname].update(l2_hit_rate=hit / max(hit + miss, 1), tcc_req=sum(b[i].get("TCC_REQ_sum", 0) for i in bb) / len(bb),
                              tcc_read=sum(b[i].get("TCC_READ_sum", 0) for i in bb) / len(bb), tcc_hit=hit / len(bb), tcc_miss=miss / len(bb))
    out[v] = res
K = 1183; nch = (K + 63) // 64
alg = {"rot_tiles_A": 4 * N * nch * 2 * 5 * 1024, "pt_tiles_B": 4 * H * 6 * nch * 5 * 1024}
out["_algorithmic_read_bytes_K1183"] = dict(alg, total=sum(alg.values()))
json.dump(out, open(root + "/pmc_mac_i8.json", "w"), indent=1)
for v, res in out.items():
    if v.startswith("_"):
        print(v, res); continue
    for k, r in res.items():
        print(f"{v:8s} {k:22s} read {r['read_bytes_per_launch'] / 1e9:8.3f} GB  (32B {r['rdreq_32B']:.3e}  64B {r['rdreq_64B']:.3e}  128B {r['rdreq_128B']:.3e}  all {r['rdreq_total']:.3e})"
              + (f"  L2 hit {r['l2_hit_rate']:.3f} req {r['tcc_req']:.3e} read {r['tcc_read']:.3e}" if "l2_hit_rate" in r else ""))
