#!/usr/bin/env python3
"""bench.py -- k-mers/s of the BlockFinder stage (graph build + simplify) on MI355X.

A "step" is one PerformGraphSimplifications(k=25, D=150, maxIterations=4) over the workload
BASELINE.json's metric is quoted on: 8 E. coli-like strains (synthetic, sibelia_amd.workloads.gen_strains,
4.6 Mbp each, seed 1 -- real E. coli genomes are not available offline), k = 25.
Inputs are resident in HBM when the timed region starts (sbl_restore_state is a device-to-device copy
of the saved stage-boundary state and is inside the timed region).

  python bench.py --gpus N --steps K --warmup W
N > 1 (launched through torch.distributed.run, one rank per GPU over RCCL): north_star's configuration -- ONE job on all
ranks, the k-mer table of the enumeration sharded by hash prefix (RCCL all-to-all of 16-B k-mer records + all-gathers of
bifurcation codes and marks, csrc/shard.hip), the globally ordered simplification replicated and bit-identical on every GPU;
value = strand-k-mers of the one job / max time, "scaling": "strong".  The same line carries, under "replicas", the weak-scaling
configuration (every rank runs the whole job on its own strain set, no data-path collective), measured right after with the
same K / W; --replicas makes that one the headline instead.

The JSON line also carries
  roofline      the dominant kernel's algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak
  cpu_baseline  the UNMODIFIED reference (oracle/_ref/ref_dump, built by oracle/build_ref.sh; kind "reference") timed on
                rank 0 on the FULL workload of `value` (1 thread -- it has none more; ~4 - 8 min, its output hashed and compared
                with the GPU's) with the bounded sample (8 strains, shorter genomes, ~20 s) beside it; --no-cpu-full keeps the
                sample only; the bit-exact port (oracle/, kind "port") only where no reference build exists
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


# the translation unit of the round kernels' dominant one (k_commit, k_resume, k_chain) and every header it includes: what a PMC summary of
# those kernels depends on.  (Rounds 1 - 5 hashed every file under csrc/: a host-only change -- a default in simplify.hip -- made the
# committed summary "formally not of the timed build")
DIGEST_FILES = ["commit.hip", "simplify_walks.h", "simplify_device.h", "simplify_steps.h", "simplify_kernels.h", "bulge_txn.h", "kmer_kernels.h", "sbl_ctx.h", "sbl_common.h"]


def W_source_digest():
    """sha256 (first 16 hex) over the sources k_commit is compiled from: ties a PMC summary under profiles/ to the build it was taken on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sibelia_amd", "csrc")
    for f in DIGEST_FILES:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ---- BASELINE.json's other configurations in the driver's format: python bench.py --config 3|4|5 -------------------------------------
# (the default run is the headline configuration, configs[1]'s shape at 8 strains; these are single-GPU lines a reader can check
# without DESIGN.md: per-stage times, the SURVEY.md 8d roofline per stage -- with the k > 32 figure where k > 32 --, the reference
# beside them, the state the timed region left compared with the reference's own output)
FINE_STAGES = [(30, 150), (100, 500), (500, 1500)]          # reference src/util.cpp:76-87 (FineStageFile)


def alg_bytes_8d(N, instances, iterations, k):
    """algorithmic HBM bytes of one stage.  k <= 32: N x 24.125 + 12 x instances + iterations x N x 4;
    k > 32: 48.125 B per strand-k-mer (fingerprint-sized slots) + k / 4 B per verified occurrence of a bifurcation group."""
    if k <= 32:
        return N * 24.125 + 12.0 * instances + iterations * N * 4.0
    return N * 48.125 + 12.0 * instances + iterations * N * 4.0 + (k / 4.0) * instances


def run_config(a):
    import hashlib
    import struct
    import subprocess
    import tempfile
    import re
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no host compute path")
    torch.cuda.set_device(0)
    import __graft_entry__ as G
    G.build()
    from sibelia_amd import BlockFinder, workloads as W
    big = {v["name"]: v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "big_vectors.json")))["vectors"]}
    small = {v["name"]: v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "vectors.json")))["vectors"]}
    if a.config == 3:
        strains, L0 = (a.strains, a.L0)
        seqs = W.gen_strains(L0=L0, n=strains, seed=1)
        stages, iters = FINE_STAGES, 4
        what = "%d synthetic E. coli-like strains x %.1f Mbp (gen_strains seed 1), the reference's `-s fine` cascade (30,150) (100,500) (500,1500), maxIterations=4" % (strains, L0 / 1e6)
        fixture = small.get("synth/strains8_4600k_fine") if (strains, L0) == (8, 4_600_000) else None
        fx_cmd = "stage:500:1500:4"
    elif a.config == 4:
        strains, L0 = (62 if a.strains == 8 else a.strains, a.L0)
        seqs = W.gen_strains(L0=L0, n=strains, seed=1)
        stages, iters = [(a.k, a.D)], a.iters
        what = "%d synthetic E. coli-like strains x %.1f Mbp (gen_strains seed 1), k=%d D=%d maxIterations=%d, one full stage" % (strains, L0 / 1e6, a.k, a.D, a.iters)
        fixture, fx_cmd = None, None                            # (the reference needs > 30 h for 62 x 4.6 Mbp: pinned at 1/10 length, tests/test_gpu_parity.py)
    else:
        total, nrec = 900_000_000, 4
        seqs = W.longk_case(total, nrec)
        stages, iters = [(5000, 15000)], 4
        what = "%d Mbp of uniform random DNA in %d records with planted repeats (workloads.longk_case), k=5000 D=15000 maxIterations=4, one full stage" % (total // 1_000_000, nrec)
        fixture, fx_cmd = big.get("synth/random4x225M_k5000"), "stage:5000:15000:4"
    bf = BlockFinder(seqs, device=0)
    bf.save_state()

    def step(collect=None):
        bf.restore_state()
        last = 0
        for (k, D) in stages:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            last = bf.PerformGraphSimplifications(k, D, iters)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            if collect is not None:
                st = bf.stats()
                rec = collect.setdefault((k, D), {"ms": 0.0, "stats": st, "phase_ms": {}})
                rec["ms"] += 1e3 * dt1
                rec["stats"] = st
                for key, v in st.items():
                    if key.endswith("_ms"):
                        rec["phase_ms"][key] = rec["phase_ms"].get(key, 0.0) + v
        return last

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = {}
    for _ in range(a.steps):
        bulges = step(per)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_step = 1e3 * dt / a.steps
    # what was timed is what is checked
    vs, vp = bf.state_views()
    h = hashlib.sha256(struct.pack("<QI", bulges, len(vs)))
    for x, y in zip(vs, vp):
        h.update(struct.pack("<Q", len(x))); h.update(x); h.update(y)
    sha = h.hexdigest()
    match, fx_name = None, None
    if fixture is not None:
        want = [o for o in fixture["outputs"] if o["cmd"] == fx_cmd][0]
        match = bool(want["sha256"] == sha and want["bulges"] == bulges)
        fx_name = "%s %s (output of the unmodified reference, oracle/_ref)" % (fixture["name"], fx_cmd)
    stage_lines, Ntot, bytes_tot = [], 0.0, 0.0
    for (k, D) in stages:
        rec = per[(k, D)]
        st = rec["stats"]
        N = float(st["strand_kmers"])
        ms = rec["ms"] / a.steps
        alg = alg_bytes_8d(N, float(st["instances"]), float(st["iterations"]), k)
        ph = {kk: v / a.steps for kk, v in sorted(rec["phase_ms"].items())}
        if k > 32:
            path = int(st.get("longk_path", 0))
            dom = {1: "long-k enumeration (k_fp_*: window fingerprints -> bucketed LDS tables -> verification of every bifurcation group; %d occurrences verified)" % int(st.get("fp_verified", 0)),
                   3: "long-k enumeration (a fingerprint verification FAILED: exact rank doubling ran, k_lk_* + radix-sort passes)"}.get(
                       path, "long-k enumeration (k_lk_* + rocPRIM radix-sort passes: exact rank doubling)")
            dom_ms, dom_launches = ph.get("enumerate_ms", 0.0), 1
        else:
            cand = {"k_commit": ph.get("commit_ms", 0.0), "k_reserve": ph.get("reserve_ms", 0.0), "k_probe": ph.get("probe_ms", 0.0), "k_snapshot": ph.get("snapshot_ms", 0.0),
                    "enumeration (k_kmer_records .. k_scatter_members)": ph.get("enumerate_ms", 0.0)}
            dom = max(cand, key=lambda q: cand[q])
            dom_ms, dom_launches = cand[dom], (max(1, int(st["rounds"])) if dom in ("k_commit", "k_reserve", "k_probe") else 1)
        stage_lines.append({"k": k, "D": D, "ms": ms, "strand_kmers": N, "value": N / (ms * 1e-3), "bif_ids": st["bif_count"], "instances": st["instances"], "bulges": st["bulges"],
                            "iterations": st["iterations"], "rounds": st["rounds"], "replays": st["replays"], "device_bytes": int(st.get("device_bytes", 0)), "phase_ms": ph,
                            "roofline": {"bound": "hbm", "kernel": dom, "algorithmic_bytes_stage": alg, "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "dominant_ms": dom_ms, "dominant_launches": dom_launches,
                                         "dominant_avg_launch_ms": dom_ms / dom_launches,
                                         "model": "SURVEY.md 8d, k %s 32" % ("<=" if k <= 32 else ">")}})
        Ntot += N; bytes_tot += alg
    slow = max(stage_lines, key=lambda r: r["ms"])
    out = {"metric": "k-mers/sec in BlockFinder graph-build+simplify, BASELINE.json config %d" % a.config,
           "value": Ntot / (ms_step * 1e-3), "unit": "strand-k-mers/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": what, "baseline_config": a.config, "strand_kmers_per_step": Ntot, "stages": [[k, D] for k, D in stages], "parallelism": "1 GPU"},
           "stages": stage_lines,
           "roofline": {"bound": "hbm", "kernel": slow["roofline"]["kernel"] + " of the slowest stage (k=%d)" % slow["k"], "achieved": bytes_tot / (ms_step * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_tot / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_step": bytes_tot, "avg_launch_ms": slow["roofline"]["dominant_avg_launch_ms"],
                        "model": "SURVEY.md 8d summed over the stages of the step (per stage under `stages`); `achieved` = those bytes / the step's wall time; "
                                 "per-kernel times and HBM traffic of this command: profiles/r06_config%d_kernel_stats.csv / _pmc_summary.json" % a.config},
           "state_sha256": sha, "matches_reference_fixture": match, "reference_fixture": fx_name}
    # ---- the reference beside it
    cpu_model = "?"
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    host = "%s, %d logical cores on the host" % (cpu_model, os.cpu_count() or 0)
    ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if a.config == 3 and not a.no_cpu_baseline and os.path.exists(ref_dump):
        sample = W.gen_strains(L0=a.cpu_sample_L0, n=strains, seed=1)
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "in.fa")
            W.write_fasta(fa, sample)
            r = subprocess.run([ref_dump, fa, os.path.join(d, "o")] + ["stage:%d:%d:%d" % (k, D, iters) for k, D in stages], capture_output=True, text=True)
            secs = [float(x) for x in re.findall(r"seconds=([0-9.]+)", r.stderr)]
            if r.returncode == 0 and len(secs) == len(stages):
                Ns = sum(W.strand_kmers(sample, k) for k, _ in stages)
                out["cpu_baseline"] = {"value": Ns / sum(secs), "unit": "strand-k-mers/s", "cores": 1, "kind": "reference", "host": host,
                                       "sample": "the unmodified reference (oracle/_ref, 1 thread: it has no parallelism) through the same three stages on %d strains x %.2f Mbp from "
                                                 "the same generator (%d strand-k-mers over the stages; %s s per stage)" % (strains, a.cpu_sample_L0 / 1e6, Ns, " / ".join("%.1f" % x for x in secs))}
    elif a.config in (4, 5):
        v = big["synth/strains62_460k" if a.config == 4 else "synth/random4x225M_k5000"]
        o = [x for x in v["outputs"] if x["cmd"].startswith("stage:")][0]
        Nref = float(W.strand_kmers([b"x" * l for l in v["lengths"]], int(o["cmd"].split(":")[1])))
        out["cpu_baseline"] = {"value": Nref / o["reference_seconds"], "unit": "strand-k-mers/s", "cores": 1, "kind": "reference",
                               "host": "RECORDED in the build container (8 vCPU Intel Xeon @ 2.10 GHz, tests/golden/make_big_golden.py), not on this host",
                               "sample": "the unmodified reference's PerformGraphSimplifications on %s: %.0f s for %d strand-k-mers (tests/golden/big_vectors.json: reference_seconds)%s"
                                         % (v["name"], o["reference_seconds"], int(Nref),
                                            "; the reference is superlinear in the number of strains -- 62 x 460 kbp is 1/10 of this workload's genome length and took 3.2 h, the full size > 30 h" if a.config == 4 else
                                            " = this very workload at full size")}
        if a.config == 4 and not a.no_cpu_baseline and os.path.exists(ref_dump):
            # ... and the reference timed HERE, in the same run, on a bounded sample of the same shape (62 strains x 46 kbp: ~2 min of its one thread)
            sample = W.gen_strains(L0=46_000, n=62, seed=1)
            with tempfile.TemporaryDirectory() as d:
                fa = os.path.join(d, "in.fa")
                W.write_fasta(fa, sample)
                t1 = time.time()
                r = subprocess.run([ref_dump, fa, os.path.join(d, "o")] + ["stage:%d:%d:%d" % (k, D, iters) for k, D in stages], capture_output=True, text=True, timeout=1200)
                secs = [float(x) for x in re.findall(r"seconds=([0-9.]+)", r.stderr)]
                if r.returncode == 0 and len(secs) == len(stages):
                    Ns = sum(W.strand_kmers(sample, k) for k, _ in stages)
                    out["cpu_baseline_live_sample"] = {"value": Ns / sum(secs), "unit": "strand-k-mers/s", "cores": 1, "kind": "reference", "host": host,
                                                       "sample": "the unmodified reference (oracle/_ref, 1 thread) on 62 strains x 0.046 Mbp from the same generator, timed on this host in this run: "
                                                                 "%d strand-k-mers in %.1f s (wall %.1f s)" % (Ns, sum(secs), time.time() - t1)}
    print(json.dumps(out), flush=True)
    bf.close()



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--strains", type=int, default=8)
    ap.add_argument("--L0", type=int, default=4_600_000)
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--D", type=int, default=150)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--cpu-sample-L0", type=int, default=400_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-enum", action="store_true", help="one job on all ranks: hash-prefix sharded enumeration over RCCL (the default for --gpus > 1)")
    ap.add_argument("--replicas", action="store_true", help="--gpus > 1: make the replicas configuration (one independent job per GPU) the headline value")
    ap.add_argument("--check", action="store_true", help="compare the GPU result of the CPU sample with the oracle")
    ap.add_argument("--cpu-full", action="store_true", help="(default since round 4) time the unmodified reference on the FULL workload as well")
    ap.add_argument("--no-cpu-full", action="store_true", help="bounded CPU sample only (~20 s) instead of the reference on the full workload (~4 - 8 min, 1 thread)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 3, 4, 5],
                    help="one of BASELINE.json's other configurations instead of the headline one: 3 = 8 strains through the `-s fine` cascade, 4 = 62 strains at k = 25, 5 = 900 Mbp of random DNA at k = 5000 (single GPU)")
    ap.add_argument("--dry-collectives", action="store_true", help="self-test of the RCCL transport with --gpus real ranks (processes) that all open device 0 "
                    "(tools/rccl_two_ranks_one_gpu.py): runs, or reports LOUDLY that RCCL refuses two ranks on one device; no bench line")
    ap.add_argument("--require-sharded", action="store_true", help="--gpus > 1: exit non-zero instead of falling back to replicas when the sharded (strong-scaling) configuration cannot run")
    a = ap.parse_args()
    if a.dry_collectives:
        os.execvp(sys.executable, [sys.executable, os.path.join(ROOT, "tools", "rccl_two_ranks_one_gpu.py"), str(max(2, a.gpus))])
    if a.config:
        if a.gpus > 1:
            raise SystemExit("--config 3 / 4 / 5 are single-GPU lines (the multi-GPU configuration of the driver's contract is the default workload)")
        return run_config(a)

    # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (RCCL), so that the
    # multi-GPU line needs no wrapper (the driver's own launch sets WORLD_SIZE and lands below directly)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no host compute path")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import __graft_entry__ as G
    if rank == 0:
        G.build()
    if world > 1:
        dist.barrier()
    from sibelia_amd import BlockFinder, workloads as W, dist as D

    # N > 1: north_star's configuration is ONE job whose k-mer table is sharded by hash prefix over the GPUs (RCCL all-to-all);
    # the replicas configuration (every rank its own strain set, no data-path collective) is measured afterwards and reported
    # in the same line under "replicas" (or as the headline with --replicas)
    if world > 1 and not a.replicas:
        a.shard_enum = True
    os.environ.setdefault("SBL_COMM_TIMEOUT_S", "90")      # a rank that waits for a failed peer gives up (communicator aborted) instead of hanging the bench
    seqs = W.gen_strains(**D.rank_workload(0 if a.shard_enum else rank, a.strains, a.L0))
    N = W.strand_kmers(seqs, a.k)
    bf = BlockFinder(seqs, device=local)
    if a.window:
        bf.set_window(a.window)
    bf.save_state()

    def step():
        bf.restore_state()
        return bf.PerformGraphSimplifications(a.k, a.D, a.iters)

    # The sharded configuration is a collective over RCCL: attach + one trial stage are guarded, and the ranks agree on the outcome.
    # If any rank failed, every rank falls back to the replicas configuration and the line says why ("sharded_error").
    shard_error = None
    if a.shard_enum:
        try:
            if world > 1:
                D.attach(bf, device=torch.device("cuda", local))
            else:
                from sibelia_amd.api import comm_unique_id
                bf.attach_rccl(0, 1, comm_unique_id())
            if world > 1:
                step()
        except Exception as e:      # noqa: BLE001 -- reported in the JSON line
            shard_error = "rank %d: %s" % (rank, e)
        if world > 1:
            flag = torch.tensor([1 if shard_error else 0], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                errs = [None] * world
                dist.all_gather_object(errs, shard_error)
                shard_error = "; ".join(e for e in errs if e) or "a peer rank failed"
                try:
                    bf.detach()
                except Exception:      # noqa: BLE001
                    pass
                bf.close()
                if a.require_sharded:
                    raise SystemExit("sharded (strong-scaling) configuration failed and --require-sharded was given: " + shard_error)
                a.shard_enum = False
                seqs = W.gen_strains(**D.rank_workload(rank, a.strains, a.L0))
                N = W.strand_kmers(seqs, a.k)
                bf = BlockFinder(seqs, device=local)
                if a.window:
                    bf.set_window(a.window)
                bf.save_state()
        elif shard_error:
            raise SystemExit("sharded enumeration failed: " + shard_error)

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agg = {}
    for _ in range(a.steps):
        bulges = step()
        st = bf.stats()
        for key, v in st.items():
            if key.endswith("_ms") or key == "commit_event_launches":
                agg[key] = agg.get(key, 0.0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt, Ntot = D.aggregate(dt, float(N), device="cuda" if world > 1 else None)
    if a.shard_enum:
        Ntot = float(N)                      # one job, however many GPUs enumerate it

    # second configuration of a multi-GPU run: replicas (weak scaling), same K / W, same barrier + max-over-ranks timing
    replicas = None
    if world > 1 and a.shard_enum:
        seqs2 = W.gen_strains(**D.rank_workload(rank, a.strains, a.L0))
        N2 = W.strand_kmers(seqs2, a.k)
        bf2 = BlockFinder(seqs2, device=local)
        bf2.save_state()
        for _ in range(a.warmup + a.steps):
            if _ == a.warmup:
                dist.barrier(); torch.cuda.synchronize(); t2 = time.perf_counter()
            bf2.restore_state()
            bf2.PerformGraphSimplifications(a.k, a.D, a.iters)
        torch.cuda.synchronize(); dist.barrier()
        dt2, N2tot = D.aggregate(time.perf_counter() - t2, float(N2), device="cuda")
        replicas = {"value": N2tot / (dt2 / a.steps), "unit": "strand-k-mers/s", "ms_per_step": 1000.0 * dt2 / a.steps, "scaling": "weak",
                    "parallelism": "replicas (x%d): one independent job per GPU on its own strain set, no data-path collective" % world}
        bf2.close()

    # ---- what was timed is what is checked (outside the timed region): the state the last step left on the device, downloaded
    # once through the C ABI, serialised like the golden vectors (formats.state_bytes) and hashed; on the metric workload the
    # reference's own result is a committed fixture (tests/golden/vectors.json: synth/strains8_4600k, generated by oracle/_ref)
    verify = None
    if rank == 0:
        import hashlib, struct
        t1 = time.perf_counter()
        vs, vp = bf.state_views()
        t_dl_first = time.perf_counter() - t1
        h = hashlib.sha256(struct.pack("<QI", bulges, len(vs)))
        for x, y in zip(vs, vp):
            h.update(struct.pack("<Q", len(x))); h.update(x); h.update(y)
        verify = {"state_sha256": h.hexdigest(), "download_first_ms": 1e3 * t_dl_first}
        if (a.strains, a.L0, a.k, a.D, a.iters) == (8, 4_600_000, 25, 150, 4):
            try:
                fx = [v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "vectors.json")))["vectors"] if v["name"] == "synth/strains8_4600k"][0]
                want = [o for o in fx["outputs"] if o["cmd"] == "stage:25:150:4"][0]
                verify["reference_fixture"] = "tests/golden/vectors.json synth/strains8_4600k stage:25:150:4 (output of the unmodified reference, oracle/_ref)"
                verify["matches_reference_fixture"] = bool(want["sha256"] == verify["state_sha256"] and want["bulges"] == bulges)
            except Exception as e:      # noqa: BLE001
                verify["matches_reference_fixture"] = None
                verify["fixture_error"] = str(e)
        else:
            verify["matches_reference_fixture"] = None       # no reference fixture for a non-default workload

    if rank == 0:
        st = bf.stats()
        ms_step = 1000.0 * dt / a.steps
        # dominant kernel by accumulated HIP-event time
        per = {"k_kmer_table_build": agg.get("kmer_table_ms", 0.0) / a.steps,
               "k_snapshot": agg.get("snapshot_ms", 0.0) / a.steps,
               "k_reserve": agg.get("reserve_ms", 0.0) / a.steps,
               "k_commit": agg.get("commit_ms", 0.0) / a.steps,
               "k_probe": agg.get("probe_ms", 0.0) / a.steps}
        launches = {"k_kmer_table_build": 1, "k_snapshot": max(1, st["iterations"] + st["replays"]),
                    "k_reserve": max(1, st["rounds"]), "k_commit": max(1, st["rounds"]), "k_probe": max(1, st["rounds"])}
        # algorithmic HBM bytes per launch: table build = 32 B per base position + packed
        # sequence; snapshot = 4 B per strand-k-mer (one pass over the dense mark arrays);
        # reserve = 4 B x 2 strands x neighbourhood (3(D+k)+k elements), commit = 8 B x window (D+k), probe = 4 B x window,
        # each per instance of the ids the launch works on
        nbh = 3 * (a.D + a.k) + a.k
        per_id = st["instances"] / max(1, st["bif_count"])
        inst_per_launch = per_id * st["transactions"] / max(1, st["rounds"])          # reserve / commit work on real transactions
        probed_per_launch = per_id * st["executed"] / max(1, st["rounds"])           # the probe looks at every pending id
        alg = {"k_kmer_table_build": float(st["kmer_table_bytes"]), "k_snapshot": 4.0 * N,
               "k_reserve": 8.0 * nbh * inst_per_launch, "k_commit": 8.0 * (a.D + a.k) * inst_per_launch,
               "k_probe": 4.0 * (a.D + a.k) * probed_per_launch}
        dom = max(per, key=lambda kk: per[kk])
        dur_ms = per[dom] / launches[dom]                  # device wall-clock start stamps, every launch of the timed steps
        dur_ms_clock, timing = dur_ms, "device wall-clock start stamps written by the round kernels, every launch of the timed steps"
        ev_launches = agg.pop("commit_event_launches", 0.0)
        if dom == "k_commit" and ev_launches > 0:
            # HIP event pairs on the context's stream around every 4th launch of k_commit (which ones rotates from step to step): an
            # event pair around every round kernel cost 1.7 - 2.2 ms of the stage; the stamps beside them time every launch
            dur_ms = agg["commit_event_ms"] / ev_launches
            timing = ("HIP event pairs around %d of the %d launches of k_commit in the timed steps (every 4th, rotating); "
                      "avg_launch_ms_all_launches = device wall-clock start stamps of every launch" % (int(ev_launches), launches[dom] * a.steps))
        achieved_design = alg[dom] / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        # SURVEY.md 8d's figure (what roofline.achieved / frac use): bytes = N x 24.125 (table: 0.125 sequence + 16 build + 8 resolve)
        # + 12 x instances + iterations x N x 4 (one streaming pass over the dense marks per iteration).  The simplification share
        # belongs to the ordered rounds as a whole; the dominant kernel is charged all of it, spread over its launches (the same
        # accounting as the judge's in VERDICT.md) -- an upper bound of what that kernel alone achieves.
        sim_8d = float(st["iterations"]) * N * 4.0
        alg8d = {"k_kmer_table_build": N * 24.125, "k_snapshot": sim_8d / max(1, st["iterations"])}
        for kk in ("k_commit", "k_probe", "k_reserve"):
            alg8d[kk] = sim_8d / launches[kk]
        stage_8d = N * 24.125 + 12.0 * st["instances"] + sim_8d
        achieved = alg8d[dom] / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        # HBM traffic of that kernel per launch from the last committed PMC passes (tools/profile_summary.py), same workload only
        # (PMC counters need the profiler around the process: they are collected by tools/collect_profiles.sh in passes of their own and
        # summarised by tools/profile_summary.py, which records the digest of the kernel sources it ran on; traffic_source says which
        # file the figure comes from and whether that digest is the one of THIS build)
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and (a.strains, a.L0, a.k, a.D, a.iters) == (8, 4_600_000, 25, 150, 4):
            try:
                doc = json.load(open(pmc))
                traffic = doc["kernels"][dom]["hbm_bytes_per_launch_est"]
                here = W_source_digest()
                traffic_source = ("committed PMC passes: profiles/pmc_latest.json (round %s, FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, digest of commit.hip + the headers it includes %s; "
                                  "this build: %s = %s)" % (doc.get("round"), doc.get("source_digest", "unrecorded"), here,
                                                            "the same sources" if doc.get("source_digest") == here else "DIFFERENT sources: indicative only"))
            except Exception:
                traffic = None
        out = {
            "metric": "k-mers/sec in BlockFinder graph-build+simplify, 8xE.coli k=25",
            "value": Ntot / (dt / a.steps), "unit": "strand-k-mers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if a.shard_enum else ("weak-fallback" if shard_error else "weak"), "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%d synthetic E. coli-like strains x %.1f Mbp (gen_strains seed 1, 1%% SNP, indels, inversions), "
                                   "k=%d D=%d maxIterations=%d, one full stage" % (a.strains, a.L0 / 1e6, a.k, a.D, a.iters),
                       "strand_kmers_per_gpu": N,
                       "parallelism": ("1 job: enumeration sharded by k-mer hash prefix over %d GPU(s) (RCCL all-to-all), read-only simplification phases (snapshots, probes) shared out with all-gathered verdict bytes, commits replicated" % world)
                       if a.shard_enum else ("replicas (x%d), one job per GPU" % world if world > 1 else "1 GPU"),
                       "exchange_ms": st["exchange_ms"], "exchange_bytes_rank0": st["exchange_bytes"],
                       "bulges": bulges, "bif_ids": st["bif_count"], "instances": st["instances"],
                       "iterations": st["iterations"], "rounds": st["rounds"], "replays": st["replays"],
                       "device_bytes": int(st.get("device_bytes", 0)), "device_bytes_per_base": float(st.get("device_bytes", 0)) / max(1, a.strains * a.L0)},
            "phase_ms": {kk: agg[kk] / a.steps for kk in sorted(agg)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_ratio": (traffic / alg8d[dom]) if traffic else None,
                         "avg_launch_ms": dur_ms, "avg_launch_ms_all_launches": dur_ms_clock, "timing": timing,
                         "launches_per_step": launches[dom], "algorithmic_bytes_per_launch": alg8d[dom],
                         "algorithmic_model": "SURVEY.md 8d: N x 24.125 + 12 x instances + iterations x N x 4 = %.2f GB per stage; the simplification share "
                                              "(iterations x N x 4 B) spread over the launches of the dominant kernel (since round 3; rounds 1 and 2 priced `frac` "
                                              "with this design's own per-kernel model, now `frac_design_model`: not comparable)" % (stage_8d / 1e9),
                         "stage": {"bytes_8d": stage_8d, "achieved": stage_8d / (ms_step * 1e-3) / 1e9, "frac": stage_8d / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "frac_design_model": achieved_design / HBM_PEAK_GBS, "design_model_bytes_per_launch": alg[dom],
                         "design_model": "this design's per-kernel model (DESIGN.md 4: e.g. k_commit = 8 B x (D + k) per instance of the launch's transactions)",
                         "all_kernels_ms_per_step": per},
        }
        out["state_sha256"] = verify["state_sha256"]
        out["matches_reference_fixture"] = verify["matches_reference_fixture"]
        out["verify"] = verify
        if world > 1:
            out["rccl_ranks"] = world
            out["exchange_ms"] = st["exchange_ms"]
            out["exchange_bytes"] = st["exchange_bytes"]
            # read-only simplification phases (snapshots, probes) shared out over the ranks, verdict bytes all-gathered; commits replicated
            out["simplification_split"] = {"ranks_sharing_read_only_phases": st["ro_ranks"], "verdict_allgather_ms": st["verdict_ms"], "verdict_bytes_rank0": st["verdict_bytes"],
                                           "probe_ms_rank0": agg.get("probe_ms", 0.0) / a.steps, "snapshot_ms_rank0": agg.get("snapshot_ms", 0.0) / a.steps,
                                           "commit_ms_rank0 (replicated)": agg.get("commit_ms", 0.0) / a.steps, "reserve_ms_rank0 (replicated)": agg.get("reserve_ms", 0.0) / a.steps}
        if replicas is not None:
            out["replicas"] = replicas
        if shard_error:
            out["sharded_error"] = shard_error
        if world == 1:
            # PCIe-inclusive rate (never `value`): host buffers -> device (sbl_load: 1 B/base over PCIe; original positions and the
            # ambiguity scan are derived on the device) + one stage + the state back to the host (5 B/base)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            b2 = BlockFinder(seqs, device=local)
            t_load = time.perf_counter() - t1
            b2.PerformGraphSimplifications(a.k, a.D, a.iters)
            t_stage = time.perf_counter() - t1 - t_load
            b2.state_views()
            t_all = time.perf_counter() - t1
            b2.close()
            # steady-state download on the warm context of the timed loop: one bulk device-to-host copy of ch[] + op[] (5 B/base)
            # into the library's pinned staging buffer, chromosomes are slices of it (the first call also pins the buffer)
            bf.restore_state()
            bf.PerformGraphSimplifications(a.k, a.D, a.iters)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            bf.state_views()
            t_dl = time.perf_counter() - t1
            out["pcie_inclusive"] = {"value": N / (t_load + ms_step * 1e-3 + t_dl), "unit": "strand-k-mers/s", "load_ms": 1e3 * t_load, "stage_ms": ms_step,
                                     "download_ms": 1e3 * t_dl,
                                     "cold": {"value": N / t_all, "load_ms": 1e3 * t_load, "stage_ms": 1e3 * t_stage, "download_ms": 1e3 * (t_all - t_load - t_stage),
                                              "note": "cold context: first-call workspace allocations of the stage and pinning of the staging buffer included"},
                                     "note": "host buffers -> sbl_load (1 B/base up) + one warm stage + sbl_get_state (5 B/base down, pinned bulk copy)"}
        if not a.no_cpu_baseline and world == 1:
            # The reference's own CPU path beside the GPU number (north_star): oracle/_ref/ref_dump is the UNMODIFIED reference
            # compiled by oracle/build_ref.sh (it travels to the GPU box as a prebuilt binary).  The reference is single-threaded
            # and superlinear in the number of strains (8 x 4.6 Mbp take 500 s), so the sample keeps the 8 strains and shortens
            # the genomes (time is linear in genome length at a fixed strain count): ~20 s of reference time.
            import subprocess, tempfile, re
            from oracle.oracle import Oracle
            sample = W.gen_strains(L0=a.cpu_sample_L0, n=a.strains, seed=1)
            Ns = W.strand_kmers(sample, a.k)
            cpu_model = "?"
            try:
                cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except Exception:
                pass
            host = "%s, %d logical cores on the host" % (cpu_model, os.cpu_count() or 0)
            ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
            o = None
            if os.path.exists(ref_dump):
                with tempfile.TemporaryDirectory() as d:
                    fa = os.path.join(d, "in.fa")
                    W.write_fasta(fa, sample)
                    r = subprocess.run([ref_dump, fa, os.path.join(d, "o"), "stage:%d:%d:%d" % (a.k, a.D, a.iters)], capture_output=True, text=True)
                    m = re.search(r"bulges=(\d+) seconds=([0-9.]+)", r.stderr)
                    if r.returncode == 0 and m:
                        ob, cdt = int(m.group(1)), float(m.group(2))
                        out["cpu_baseline"] = {"value": Ns / cdt, "unit": "strand-k-mers/s", "cores": 1, "kind": "reference", "host": host,
                                               "sample": "the unmodified reference's BlockFinder::PerformGraphSimplifications(%d,%d,%d) (oracle/_ref, 1 thread: it has no "
                                                         "parallelism) on %d strains x %.2f Mbp from the same generator (%d strand-k-mers, %.1f s, %d bulges)"
                                                         % (a.k, a.D, a.iters, a.strains, a.cpu_sample_L0 / 1e6, Ns, cdt, ob)}
                    if "cpu_baseline" in out and not a.no_cpu_full:
                        # the same binary on the FULL workload of `value`, once, on THIS host (north_star: "the reference's own CPU path
                        # timed on the GPU box's host cores in the same run"): it becomes the baseline, the bounded sample stays beside it
                        fa2 = os.path.join(d, "full.fa")
                        W.write_fasta(fa2, seqs)
                        try:                                      # (bounded: a loaded host must not turn the bench line into a time-out of the whole run)
                            r2 = subprocess.run([ref_dump, fa2, os.path.join(d, "f"), "stage:%d:%d:%d" % (a.k, a.D, a.iters)], capture_output=True, text=True, timeout=1200)
                            m2 = re.search(r"bulges=(\d+) seconds=([0-9.]+)", r2.stderr)
                        except subprocess.TimeoutExpired:
                            r2, m2 = None, None
                            out["cpu_baseline"]["full"] = "not finished within 1200 s on this host: the bounded sample stands"
                        if r2 is not None and r2.returncode == 0 and m2:
                            fb, fdt = int(m2.group(1)), float(m2.group(2))
                            import hashlib
                            fsha = hashlib.sha256(open(os.path.join(d, "f.0.out"), "rb").read()).hexdigest()
                            smp = dict(out["cpu_baseline"])
                            out["cpu_baseline"] = {"value": N / fdt, "unit": "strand-k-mers/s", "cores": 1, "kind": "reference", "host": host,
                                                   "sample": "the FULL workload of `value` (%d strains x %.1f Mbp, %d strand-k-mers): the unmodified reference's "
                                                             "BlockFinder::PerformGraphSimplifications(%d,%d,%d), 1 thread (it has no parallelism), %.1f s, %d bulges"
                                                             % (a.strains, a.L0 / 1e6, N, a.k, a.D, a.iters, fdt, fb),
                                                   "full": {"seconds": fdt, "bulges": fb, "bulges_equal_gpu": fb == bulges, "state_sha256": fsha,
                                                            "state_equal_gpu": fsha == verify["state_sha256"],
                                                            "gpu_over_reference": (Ntot / (dt / a.steps)) / (N / fdt)},
                                                   "bounded_sample": {"value": smp["value"], "sample": smp["sample"], "sample_over_full_rate": smp["value"] / (N / fdt)}}
            if "cpu_baseline" not in out:                 # no reference build on this box: the bit-exact port (oracle/) instead
                o = Oracle(sample)
                t1 = time.perf_counter()
                ob = o.simplify_stage(a.k, a.D, a.iters)
                cdt = time.perf_counter() - t1
                out["cpu_baseline"] = {"value": Ns / cdt, "unit": "strand-k-mers/s", "cores": 1, "kind": "port", "host": host,
                                       "sample": "%d strains x %.2f Mbp from the same generator (%d strand-k-mers, %.1f s, %d bulges)"
                                                 % (a.strains, a.cpu_sample_L0 / 1e6, Ns, cdt, ob)}
            if a.check:
                if o is None:
                    o = Oracle(sample)
                    ob = o.simplify_stage(a.k, a.D, a.iters)
                g2 = BlockFinder(sample, device=local)
                gb = g2.PerformGraphSimplifications(a.k, a.D, a.iters)
                (sa, pa), (sb, pb) = g2.state(), o.state()
                import numpy as np
                ok = gb == ob and sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
                out["parity_check"] = "bit-exact vs oracle on the CPU sample" if ok else "MISMATCH"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
