#!/usr/bin/env python3
"""Turn the rocprofv3 output of a round into the committed summaries under profiles/.

  python tools/profile_summary.py rNN <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [bench.json]

stats_dir:  rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python bench.py ...
pmc_*_dir:  rocprofv3 --kernel-trace --pmc FETCH_SIZE (resp. WRITE_SIZE) --output-format csv -d <dir> -- python bench.py ...
            (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes)
Writes profiles/rNN_kernel_stats.csv, profiles/rNN_pmc_summary.json and profiles/pmc_latest.json (read by bench.py to
fill roofline.traffic for the dominant kernel).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not f:
        raise SystemExit("no file matches " + pattern)
    return f[-1]


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void rocprim"):
            k = "rocprim::*"
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    tag, stats_dir, fdir, wdir = sys.argv[1:5]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(newest(os.path.join(stats_dir, "**", "*_kernel_stats.csv") if False else os.path.join(stats_dir, "*", "*_kernel_stats.csv")),
                os.path.join(out, tag + "_kernel_stats.csv"))
    fetch = per_kernel(newest(os.path.join(fdir, "*", "*_counter_collection.csv")))
    write = per_kernel(newest(os.path.join(wdir, "*", "*_counter_collection.csv")))
    kernels = {}
    for k, (n, v) in fetch.items():
        wn, wv = write.get(k, (0, 0.0))
        kernels[k] = {"launches": n, "fetch_kb_per_launch_raw": v / n, "write_kb_per_launch_raw": (wv / wn) if wn else 0.0,
                      # gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled (upper estimate for
                      # scattered 4-byte reads); WRITE_SIZE is uncalibrated (device-scope atomics count as fabric writes)
                      "hbm_bytes_per_launch_est": (2 * v / n + ((wv / wn) if wn else 0.0)) * 1024}
    sys.path.insert(0, ROOT)
    import bench
    bench_args = os.environ.get("PROFILE_BENCH_ARGS", "")          # e.g. "--config 3": the summaries of another bench configuration (no pmc_latest.json then)
    doc = {"round": tag, "source_digest": bench.W_source_digest(),
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python bench.py %s--steps 1 --warmup 0 --no-cpu-baseline" % (bench_args + " " if bench_args else ""),
           "workload": os.environ.get("PROFILE_WORKLOAD", "default bench workload (8 strains x 4.6 Mbp, k=25, D=150, 4 iterations)"),
           "units": "counter unit KB; hbm_bytes_per_launch_est = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md HBM section)",
           "kernels": kernels}
    for name in ((tag + "_pmc_summary.json",) if bench_args else (tag + "_pmc_summary.json", "pmc_latest.json")):
        json.dump(doc, open(os.path.join(out, name), "w"), indent=1)
    if len(sys.argv) > 5:
        shutil.copy(sys.argv[5], os.path.join(out, tag + "_bench_default.json"))
    for k, v in sorted(kernels.items(), key=lambda x: -x[1]["hbm_bytes_per_launch_est"] * x[1]["launches"])[:10]:
        print("%-24s launches %5d  est HBM/launch %10.2f MB" % (k, v["launches"], v["hbm_bytes_per_launch_est"] / 1e6))


if __name__ == "__main__":
    main()
