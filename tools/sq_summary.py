#!/usr/bin/env python3
"""SQ counters of one rocprofv3 pass per kernel -> profiles/<tag>_sq_counters.json.

  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
            --output-format csv -d <dir> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
  python tools/sq_summary.py <tag> <dir>

WAIT_ANY = wave parked on s_waitcnt / barrier, ACTIVE_INST_ANY = wave issuing; resident waves per SIMD = 4 x SQ_WAVE_CYCLES /
(duration x clock) / 1024 SIMDs (the counter ticks once per 4 cycles and wave); issue utilisation of a SIMD ~ active share x resident waves."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_GHZ = 2.4


def main():
    tag, d = sys.argv[1:3]
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void rocprim") or k.startswith("__amd"):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in launches[k]:
            launches[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = {}
    for k, c in sorted(agg.items(), key=lambda x: -dur[x[0]]):
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if not wc or not dur[k]:
            continue
        res = 4.0 * wc / (dur[k] * 1e-3 * CLOCK_GHZ * 1e9) / 1024.0
        act = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        out[k] = {"launches": len(launches[k]), "total_ms": round(dur[k], 3), "waves": int(c.get("SQ_WAVES", 0)),
                  "wait_any_pct": round(100 * c.get("SQ_WAIT_ANY", 0.0) / wc, 1), "wait_inst_pct": round(100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 1),
                  "active_pct": round(100 * act, 1), "resident_waves_per_simd": round(res, 2), "simd_issue_utilisation_est": round(act * res, 2),
                  "valu_insts_per_wave": round(c.get("SQ_INSTS_VALU", 0.0) / max(1.0, c.get("SQ_WAVES", 1.0))),
                  "lds_insts_per_wave": round(c.get("SQ_INSTS_LDS", 0.0) / max(1.0, c.get("SQ_WAVES", 1.0)))}
    sys.path.insert(0, ROOT)
    import bench
    doc = {"round": tag, "source_digest": bench.W_source_digest(),
           "command": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS "
                      "-- python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
           "reading": __doc__.split("\n\n")[-1].replace("\n", " "), "kernels": out}
    json.dump(doc, open(os.path.join(ROOT, "profiles", tag + "_sq_counters.json"), "w"), indent=1)
    for k in list(out)[:6]:
        print(k, out[k])


if __name__ == "__main__":
    main()
