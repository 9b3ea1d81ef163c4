"""Per-step per-kernel totals from a rocprofv3 kernel trace CSV: steps are split at every n-th launch of the big MAC kernel (its launches per step are fixed)."""
import csv, sys, collections
f, per_step = sys.argv[1], int(sys.argv[2])
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
step, seen = 0, 0
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
seq = collections.defaultdict(list)
dump = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
for s, e, name, grid in rows:
    key = name.split("(")[0][-40:]
    if dump and step == 2 and ("k_ntt_inv" in key or "fwd_split" in key or "k_mac_bc" in key): dump.write(f"{key.split()[-1]} {grid} {(e - s) / 1e3:.1f}\n")
    tot[step][key][0] += 1; tot[step][key][1] += (e - s) / 1e6
    if key.endswith("k_ntt_inv"): seq[step].append((e - s) / 1e3)
    if "k_mac_bc<false" in name:
        seen += 1
        if seen % per_step == 0: step += 1
for st in sorted(tot):
    print("== step", st)
    for k, (c, ms) in sorted(tot[st].items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {k:42s} {c:6d} {ms:9.2f} ms  avg {1e3 * ms / c:9.1f} us")
    q = seq[st]
    if q: print("   k_ntt_inv durations (us), in launch order, every 10th:", [round(x) for x in q[::10]])
