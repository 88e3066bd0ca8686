#!/usr/bin/env python3
"""Volume / cost of the hash-prefix sharded enumeration on the bench workload, with R virtual ranks on ONE GPU
(local transport).  Times are not multi-GPU times (the ranks share the device); the bytes are exact."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sibelia_amd import BlockFinder, workloads as W          # noqa: E402
from sibelia_amd.dist import LocalShardedFinder               # noqa: E402

L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
seqs = W.gen_strains(L0=L0, n=8, seed=1)
N = W.strand_kmers(seqs, 25)
one = BlockFinder(seqs, device=0)
one.enumerate(25)
t = time.perf_counter(); ref = one.enumerate(25); t_one = time.perf_counter() - t
print(json.dumps({"ranks": 0, "mode": "unsharded", "strand_kmers": N, "bif": int(ref[0]), "enumerate_call_s": t_one}))
for R in (1, 2, 4, 8):
    f = LocalShardedFinder(seqs, [0] * R)
    f.enumerate(25)
    t = time.perf_counter(); got = f.enumerate(25); dt = time.perf_counter() - t
    assert got[0] == ref[0] and (got[1] == ref[1]).all() and (got[2] == ref[2]).all()
    st = f.stats()
    print(json.dumps({"ranks": R, "mode": "local virtual ranks on one GPU", "enumerate_call_s": dt,
                      "exchange_bytes_per_rank": [s["exchange_bytes"] for s in st],
                      "exchange_bytes_total": sum(s["exchange_bytes"] for s in st),
                      "bytes_per_strand_kmer": sum(s["exchange_bytes"] for s in st) / N,
                      "slice_table_ms": [round(s["kmer_table_ms"], 3) for s in st]}))
    f.close()
