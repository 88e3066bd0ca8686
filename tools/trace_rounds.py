#!/usr/bin/env python3
"""Per-launch durations of the simplification kernels from a rocprofv3 --kernel-trace CSV (launch order)."""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    seq.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in ("k_snapshot", "k_probe", "k_reserve", "k_commit", "k_select", "k_mark_big"):
    v = [x for kk, xs in seq.items() if kk.startswith(k) for x in xs] if k not in seq else seq[k]
    if v:
        print(k, "n=%d total=%.1f ms" % (len(v), sum(v) / 1e3))
        print("  us:", " ".join("%.0f" % x for x in v))
tot = sum(sum(xs) for xs in seq.values())
print("all kernels: %.1f ms;" % (tot / 1e3), "span %.1f ms" % ((int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
other = sorted(((sum(xs), k, len(xs)) for k, xs in seq.items()), reverse=True)[:14]
for s, k, n in other:
    print("  %-60s n=%-5d %.2f ms" % (k[:60], n, s / 1e3))
