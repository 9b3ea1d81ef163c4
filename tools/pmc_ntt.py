"""Summarise tools/r5_pmc_ntt.sh: per launch of the encode kernels (k_ntt_half3, k_fft_encode) and of the other large kernels of a product, exact fabric-side read and
write bytes (TCC_EA0_RDREQ_* / WRREQ_* passes) and the SQ issue / wait counters.  Output: <dir>/pmc_ntt.json in the format bench.py reads for `roofline.traffic`
({kernel: {launches, read_bytes_per_launch, write_bytes_per_launch, hbm_bytes_per_launch}}, "_config")."""
import csv, glob, json, sys, collections

root, cfg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "c2")


def load(d):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    dur = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return per, n, dur


out = {"_config": cfg, "_method": "separate rocprofv3 --pmc passes (tools/r5_pmc_ntt.sh): TCC_EA0_RDREQ_{32B,64B,128B}_sum and TCC_EA0_WRREQ_{sum,64B_sum}; "
                                 "bytes = 32 n32 + 64 n64 + 128 n128 (reads), 64 n64 + 32 (n - n64) (writes); exact request sizes, no FETCH_SIZE correction"}
try:
    rd, nr, _ = load(root + "/R"); wr, nw, _ = load(root + "/W")
    for k in rd:
        if not any(s in k for s in ("k_ntt_half3", "k_fft_encode", "k_i8_pack_pt_digits", "k_mac_i8_ring", "k_i8_untile", "k_ntt_fwd_split", "k_ntt_inv", "k_ksw")):
            continue
        c = rd[k]; rb = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
        w = wr.get(k, {}); n64 = w.get("TCC_EA0_WRREQ_64B_sum", 0); wb = 64 * n64 + 32 * (w.get("TCC_EA0_WRREQ_sum", 0) - n64)
        n = max(nr[k], 1); m = max(nw.get(k, 1), 1)
        out[k] = {"launches": n, "read_bytes_per_launch": rb / n, "write_bytes_per_launch": wb / m, "hbm_bytes_per_launch": rb / n + wb / m}
except (IndexError, FileNotFoundError) as e:
    out["_traffic_error"] = repr(e)
sq = {}
for pn in ("S1", "S2"):
    try:
        c, n, dur = load(root + "/" + pn)
    except (IndexError, FileNotFoundError):
        continue
    for k in c:
        if any(s in k for s in ("k_ntt_half3", "k_fft_encode")):
            e = sq.setdefault(k, {}); e.update({kk: v for kk, v in c[k].items()}); e["launches"] = n[k]; e["ms_" + pn] = dur.get(k, 0.0)
out["_sq"] = sq
json.dump(out, open(root + "/pmc_ntt.json", "w"), indent=1)
for k, v in out.items():
    if k.startswith("_") and k != "_sq":
        continue
    if k == "_sq":
        for kk, e in v.items():
            w = e.get("SQ_WAVES", 0) or 1
            print(f"SQ {kk[:40]:40s} waves {w:.3e} VALU/wave {e.get('SQ_INSTS_VALU', 0) / w:7.0f} LDS/wave {e.get('SQ_INSTS_LDS', 0) / w:6.0f} SALU/wave {e.get('SQ_INSTS_SALU', 0) / w:6.0f} "
                  f"VMEM/wave {e.get('SQ_INSTS_VMEM', 0) / w:5.0f}  wave_cycles/wave {e.get('SQ_WAVE_CYCLES', 0) / w:8.0f} active_valu/wave {e.get('SQ_ACTIVE_INST_VALU', 0) / w:8.0f} "
                  f"wait_inst_any/wave {e.get('SQ_WAIT_INST_ANY', 0) / w:8.0f} wait_any/wave {e.get('SQ_WAIT_ANY', 0) / w:8.0f} busy_cycles {e.get('SQ_BUSY_CYCLES', 0):.3e} gui {e.get('GRBM_GUI_ACTIVE', 0):.3e} "
                  f"ms {e.get('ms_S1', 0):.1f}")
        continue
    print(f"{k[:52]:52s} n={v['launches']:5d} read {v['read_bytes_per_launch'] / 1e9:8.4f} GB  write {v['write_bytes_per_launch'] / 1e9:8.4f} GB")
