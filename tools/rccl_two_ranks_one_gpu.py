#!/usr/bin/env python3
"""Self-test of the RCCL transport with TWO real ranks (processes) on ONE GPU: `python tools/rccl_two_ranks_one_gpu.py [nranks=2]`
(also reachable as `python bench.py --gpus 2 --dry-collectives`).

The test box has one MI355X, so the multi-rank exchanges of csrc/shard.hip (grouped ncclSend / ncclRecv + ncclAllGather) and of the
sharded rank doubling (csrc/longk.hip) have only ever carried self-sends through RCCL.  This harness starts `nranks` processes that all
open device 0, distributes the communicator id over a gloo group (127.0.0.1), attaches the library's own RcclComm and runs ONE job
through it: the hash-prefix sharded enumeration at k = 25, the sharded rank doubling at k = 100 and a stage with the read-only phases
shared out, each compared with the unsharded result of the same process.  RCCL normally refuses two ranks on one device
("Duplicate GPU detected"); the harness then prints {"skipped": ...} LOUDLY and exits 0 -- nothing is claimed.  One JSON line from rank 0."""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")                      # only to hand the 128-byte communicator id around
    from sibelia_amd import BlockFinder, workloads as W
    from sibelia_amd.api import comm_unique_id, COMM_ID_BYTES
    os.environ.setdefault("SBL_COMM_TIMEOUT_S", "60")
    seqs = W.gen_strains(L0=300_000, n=4, seed=9, inv_min=5000, inv_max=20000)
    one = BlockFinder(seqs, device=0)
    ref25, ref100 = one.enumerate(25), one.enumerate(100)
    refb = one.simplify_stage(25, 150, 4)
    refs = one.state()
    one.close()
    out = {"ranks": world, "device": "every rank on device 0", "rccl": None}
    bf = BlockFinder(seqs, device=0)
    try:
        t = torch.frombuffer(bytearray(comm_unique_id() if rank == 0 else bytes(COMM_ID_BYTES)), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        t0 = time.perf_counter()
        bf.attach_rccl(rank, world, bytes(t.numpy().tobytes()))
        a25 = bf.enumerate(25)
        a100 = bf.enumerate(100)
        b = bf.simplify_stage(25, 150, 4)
        s = bf.state()
        ok = (a25[0] == ref25[0] and np.array_equal(a25[1], ref25[1]) and np.array_equal(a25[2], ref25[2])
              and a100[0] == ref100[0] and np.array_equal(a100[1], ref100[1]) and np.array_equal(a100[2], ref100[2])
              and b == refb and s[0] == refs[0] and all(np.array_equal(x, y) for x, y in zip(s[1], refs[1])))
        st = bf.stats()
        out.update({"rccl": "ran", "identical_to_one_gpu": bool(ok), "seconds": round(time.perf_counter() - t0, 2),
                    "exchange_bytes_rank": int(st["exchange_bytes"]), "ro_ranks": int(st["ro_ranks"])})
    except Exception as e:      # noqa: BLE001 -- the whole point is to report what RCCL said
        out.update({"rccl": "refused", "skipped": "RCCL would not run %d ranks on one device: %s" % (world, str(e)[:300])})
    flags = [None] * world
    dist.all_gather_object(flags, out)
    if rank == 0:
        bad = [f for f in flags if f.get("rccl") == "ran" and not f.get("identical_to_one_gpu")]
        res = flags[0]
        res["all_ranks"] = [f.get("rccl") for f in flags]
        if any(f.get("rccl") == "refused" for f in flags):
            res["rccl"] = "refused"
            res["skipped"] = next(f["skipped"] for f in flags if f.get("rccl") == "refused")
            print("SKIPPED (nothing was tested): " + res["skipped"], file=sys.stderr, flush=True)
        print(json.dumps(res), flush=True)
        dist.destroy_process_group()
        sys.exit(1 if bad else 0)
    dist.destroy_process_group()


def main(nranks=2):
    if "RANK" in os.environ:
        return worker()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
                               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)])


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
