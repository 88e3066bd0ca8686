#!/usr/bin/env python3
"""The sharded rank doubling of the long-k enumeration (csrc/longk.hip) with R virtual ranks on ONE GPU (local transport):
per-rank exchange bytes (exact), suffixes sorted per rank and round (SBL_TRACE), wall time of the call with all ranks sharing
the device, against the single-GPU call.  usage: longk_shard_probe.py [L0=4600000] [strains=8] [k ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sibelia_amd import BlockFinder, workloads as W          # noqa: E402
from sibelia_amd.dist import LocalShardedFinder               # noqa: E402

L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ks = [int(x) for x in sys.argv[3:]] or [100, 500]
seqs = W.gen_strains(L0=L0, n=n, seed=1) if n else W.longk_case(L0, 4)
one = BlockFinder(seqs, device=0)
for k in ks:
    one.enumerate(k)
    t = time.perf_counter(); ref = one.enumerate(k); t_one = time.perf_counter() - t
    nsuf = 2 * (sum(len(s) for s in seqs) + len(seqs) + 1) - 1 + k
    print(json.dumps({"k": k, "ranks": 0, "mode": "one GPU, not sharded", "suffixes": nsuf, "bif": int(ref[0]), "enumerate_call_s": round(t_one, 4)}), flush=True)
    for R in (1, 2, 4, 8):
        f = LocalShardedFinder(seqs, [0] * R)
        f.enumerate(k)
        t = time.perf_counter(); got = f.enumerate(k); dt = time.perf_counter() - t
        assert got[0] == ref[0] and (got[1] == ref[1]).all() and (got[2] == ref[2]).all()
        st = f.stats()
        print(json.dumps({"k": k, "ranks": R, "mode": "virtual ranks sharing one GPU (times are NOT multi-GPU times; bytes are exact)", "enumerate_call_s": round(dt, 4),
                          "call_s_per_rank_share": round(dt / R, 4),
                          "exchange_bytes_per_rank": [s["exchange_bytes"] for s in st], "bytes_per_suffix_total": round(sum(s["exchange_bytes"] for s in st) / nsuf, 2),
                          "xgmi_ms_at_7x45GBps_per_rank": round(max(s["exchange_bytes"] for s in st) / (7 * 45e9) * 1e3, 2)}), flush=True)
        f.close()
