"""HIP path vs the reference's golden vectors and vs the CPU oracle (needs an MI355X: -m gpu).

Every call goes through the C ABI of libsibelia_amd.so (sibelia_amd.api.BlockFinder).  Bit-exact:
bifurcation ids and instances, post-stage sequences, original positions, bulge counts, DOT text.
"""
import numpy as np
import os

import pytest

from tests import vectors as V

pytestmark = pytest.mark.gpu

VECS = V.load_vectors()


def _max_k(v):
    return max(int(o["cmd"].split(":")[1]) for o in v["outputs"])


def _bulges(v):
    return sum(o.get("bulges", 0) for o in v["outputs"])


# Low-complexity small cases with thousands of collapses on a few hundred bases are the dense-conflict regime: every element is a
# bifurcation, ids have thousands of instances, every transaction conflicts with every other.  Inputs of this size take the
# one-launch path (k_dense_stage: for iteration, for id, RemoveBulges(id) on one wave with lazy windows); ALL vectors run below.
# The same vectors up to 8000 collapses also run through the ordered rounds + serial chain (SBL_NO_DENSE_PATH=1), which is what a
# larger input of the same kind gets.
SUPPORTED = VECS
DENSE = [v for v in VECS if v["name"].startswith("small/") and 400 <= _bulges(v) < 8000]
HUGE = [v for v in VECS if v["name"].startswith("small/") and _bulges(v) >= 8000]
FAST = [v for v in SUPPORTED if not v["name"].startswith(("real/", "synth/strains2", "synth/strains8")) and v not in DENSE and v not in HUGE]
BIG = [v for v in SUPPORTED if v["name"].startswith(("real/", "synth/strains2"))]
assert len(FAST) + len(DENSE) + len(HUGE) + len(BIG) + 2 == len(VECS)      # (+ the two 8-strain vectors, replayed by their own tests below): nothing is left out


def _bf(seqs):
    from sibelia_amd import BlockFinder
    return BlockFinder(seqs, device=0)


@pytest.mark.parametrize("v", FAST, ids=[v["name"] for v in FAST])
def test_hip_matches_reference(v):
    V.replay(v, _bf)


@pytest.mark.parametrize("v", BIG, ids=[v["name"] for v in BIG])
def test_hip_matches_reference_genomes(v):
    V.replay(v, _bf)


@pytest.mark.parametrize("v", DENSE, ids=[v["name"] for v in DENSE])
def test_hip_matches_reference_dense_conflicts(v):
    V.replay(v, _bf)


@pytest.mark.parametrize("v", HUGE, ids=[v["name"] for v in HUGE])
def test_hip_matches_reference_densest(v):
    """every golden vector of the dense regime, small/078's second stage (k = 3, D = 144) included, inside a time bound that a
    return of the old cliff (13 - 94 s, > 10 min for 078) would break"""
    import time
    t0 = time.time()
    V.replay(v, _bf)
    # 1.4 - 4.5 s on an idle MI355X; the bound only has to catch the cliff, not a loaded box (SBL_STRICT_TIMING=1: the tight bound)
    bound = 30.0 if os.environ.get("SBL_STRICT_TIMING") else 240.0
    assert time.time() - t0 < bound, "dense vector took %.1f s" % (time.time() - t0)


@pytest.mark.parametrize("v", DENSE[::3], ids=[v["name"] for v in DENSE[::3]])
def test_dense_vectors_through_the_ordered_rounds(v, monkeypatch):
    """the general path (ordered rounds, serial chain with lazy windows) on every third dense vector: what an input of the same kind
    above the one-launch size limit runs through"""
    monkeypatch.setenv("SBL_NO_DENSE_PATH", "1")
    V.replay(v, _bf)


def test_one_launch_path_falls_back_when_a_pool_runs_out(monkeypatch):
    """k_dense_stage cannot grow a pool and replay: with a node pool that is too small it must stop, and the stage must come out of
    the ordered rounds (which can) with the reference's result all the same"""
    v = [x for x in VECS if x["name"] == "small/074"][0]
    monkeypatch.setenv("SBL_TEST_DENSE_NODE_SLACK", "64")
    seqs = V.vector_input(v)
    bf = _bf(seqs)
    try:
        outs = [o for o in v["outputs"]]
        for o in outs[:2]:                                   # enum + the first stage (thousands of collapses need far more than 64 new nodes)
            got = V.run_cmd(bf, o["cmd"])
            assert V.F.sha256(got) == o["sha256"], o["cmd"]
        st = bf.stats()
        assert st["rounds"] > 0, "the stage was expected to come out of the ordered rounds"
    finally:
        bf.close()


@pytest.mark.parametrize("checkpoints", [False, True])
def test_abandoned_optimistic_attempt_is_rerun_with_checkpoints(monkeypatch, checkpoints):
    """The first attempt at a stage takes no iteration checkpoints; when it would need a roll-back (here: the element pool is made too
    small for the insertions of the stage) it is abandoned and the stage runs again from its input with checkpoints, pool growth
    and iteration replays -- same result as the oracle, one progress sequence, the abandoned attempt counted as a replay."""
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=60_000, n=4, seed=21, inv_min=2000, inv_max=6000)
    monkeypatch.setenv("SBL_TEST_ELEM_SLACK", "64")
    monkeypatch.setenv("SBL_NO_DENSE_PATH", "1")
    if checkpoints:
        monkeypatch.setenv("SBL_CHECKPOINTS", "1")
    bf, orc = _bf(seqs), Oracle(seqs)
    calls = []
    try:
        assert bf.PerformGraphSimplifications(25, 150, 4, lambda p, s: calls.append((p, s))) == orc.simplify_stage(25, 150, 4)
        (sa, pa), (sb, pb) = bf.state(), orc.state()
        assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
        st = bf.stats()
        assert st["grow_replays"] >= 1 and st["replays"] >= st["grow_replays"] + (0 if checkpoints else 1)
        assert calls[0] == (0, 0) and calls[-1] == (50, 2) and [s for _, s in calls].count(0) == 1 and [s for _, s in calls].count(2) == 1
        runs = [p for p, s in calls if s == 1]
        assert runs == [min(i + 1, 50) for i in range(len(runs))]
    finally:
        bf.close()


@pytest.mark.parametrize("window", [1, 7, 100000])
def test_window_size_does_not_change_results(window):
    # the number of ids committed per ordered round is a pure performance knob
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=50_000, n=4, seed=5, inv_min=2000, inv_max=6000)
    bf, orc = _bf(seqs), Oracle(seqs)
    bf.set_window(window)
    assert bf.simplify_stage(25, 150, 4) == orc.simplify_stage(25, 150, 4)
    (sa, pa), (sb, pb) = bf.state(), orc.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))


def test_stage_cascade_matches_oracle_on_8_strains():
    # config-3 shaped input at reduced length: 8 strains, stage (25,150) then (30,150) then (16,120)
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=300_000, n=8, seed=21, inv_min=5000, inv_max=20000)
    bf, orc = _bf(seqs), Oracle(seqs)
    for k, d in ((25, 150), (30, 150), (16, 120)):
        assert bf.simplify_stage(k, d, 4) == orc.simplify_stage(k, d, 4)
        (sa, pa), (sb, pb) = bf.state(), orc.state()
        assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    a, b = bf.enumerate(25), orc.enumerate(25)
    assert a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all()


def test_progress_callback_contract():
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=20_000, n=2, seed=9, inv_min=500, inv_max=2000)
    bf = _bf(seqs)
    calls = []
    bf.PerformGraphSimplifications(25, 150, 4, lambda p, s: calls.append((p, s)))
    assert calls[0] == (0, 0) and calls[-1] == (50, 2)          # (0,start) ... (50,end), reference blockfinder.cpp:23-48
    runs = [p for p, s in calls if s == 1]
    assert runs == sorted(runs) and all(1 <= p <= 50 for p in runs)


def test_round_kernel_times_from_stamps_and_sampled_events_agree():
    """probe / reserve / commit times of a stage come from start stamps the round kernels write themselves (every launch); HIP event
    pairs are recorded around every 4th launch of the commit kernel only.  Both must be there and tell the same story."""
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=400_000, n=8, seed=4, inv_min=3000, inv_max=12000)
    bf = _bf(seqs)
    try:
        bf.save_state()
        ev_ms = ev_n = clock_ms = rounds = 0.0
        for _ in range(4):                                   # the sampled launches rotate from stage to stage
            bf.restore_state()
            bf.PerformGraphSimplifications(25, 150, 4)
            st = bf.stats()
            assert st["rounds"] > 8 and st["probe_ms"] > 0 and st["reserve_ms"] > 0 and st["commit_ms"] > 0
            assert st["probe_ms"] + st["reserve_ms"] + st["commit_ms"] < st["simplify_ms"]
            ev_ms += st["commit_event_ms"]; ev_n += st["commit_event_launches"]; clock_ms += st["commit_ms"]; rounds += st["rounds"]
        assert rounds / 4 - 16 <= ev_n <= rounds / 4 + 16    # every 4th launch of every iteration of four stages
        a, b = ev_ms / ev_n, clock_ms / rounds
        # same story: within 25 % on an idle box (SBL_STRICT_TIMING=1); by default only "the same order of magnitude", so that a
        # loaded or slower box cannot turn a measurement cross-check into a red suite
        tol = 0.25 if os.environ.get("SBL_STRICT_TIMING") else 1.0
        assert abs(a - b) < tol * b, (a, b)
    finally:
        bf.close()


def test_errors_cross_the_abi_as_codes():
    from sibelia_amd import SibeliaError
    bf = _bf([b"ACGTACGTAC"])
    with pytest.raises(SibeliaError):
        bf.enumerate(1)                                          # k < 2 is rejected (reference src/util.cpp:38-41)


def test_fine_cascade_on_8_strains_matches_oracle():
    # config 3 shape (8 strains, -s fine: (30,150)(100,500)(500,1500)) at reduced genome length; k > 32 uses the rank-doubling path
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=120_000, n=8, seed=33, inv_min=3000, inv_max=12000)
    bf, orc = _bf(seqs), Oracle(seqs)
    for k, d in ((30, 150), (100, 500), (500, 1500)):
        assert bf.simplify_stage(k, d, 4) == orc.simplify_stage(k, d, 4)
        (sa, pa), (sb, pb) = bf.state(), orc.state()
        assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    a, b = bf.enumerate(500), orc.enumerate(500)
    assert a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all()


def test_full_size_stage_is_independent_of_the_speculation_window():
    # BASELINE.json's metric workload at full size (8 strains x 4.6 Mbp, k=25, D=150): the result must not depend on how many
    # ids are speculated per round, must be reproducible, and must reproduce the counts the reference binary reports for
    # this input (1 120 044 ids, 7 089 276 instances, 334 284 bulges)
    import hashlib
    from sibelia_amd import workloads as W, formats as F
    gold = [v for v in VECS if v["name"] == "synth/strains8_4600k"][0]      # outputs of the unmodified reference on this input (499.8 s there)
    seqs = V.vector_input(gold)
    gold_stage = [o for o in gold["outputs"] if o["cmd"] == "stage:25:150:4"][0]
    digests = []
    for window in (16384, 3000, 16384):
        bf = _bf(seqs)
        bf.set_window(window)
        bulges = bf.simplify_stage(25, 150, 4)
        st = bf.stats()
        assert (bulges, st["bif_count"], st["instances"]) == (334284, 1120044, 7089276)
        s, p = bf.state()
        digests.append(hashlib.sha256(F.state_bytes(bulges, s, p)).hexdigest())
        bf.close()
    assert digests[0] == digests[1] == digests[2]
    assert digests[0] == gold_stage["sha256"], "post-stage state differs from the reference's (sequences + original positions, bit for bit)"


def test_full_size_enumeration_matches_reference():
    # E1/E2 on the metric workload: ids and instances sha256-identical to the reference's (1 120 044 ids / 7 089 276 instances)
    gold = [v for v in VECS if v["name"] == "synth/strains8_4600k"][0]
    o = [o for o in gold["outputs"] if o["cmd"] == "enum:25"][0]
    bf = _bf(V.vector_input(gold))
    got = V.run_cmd(bf, "enum:25")
    bf.close()
    assert len(got) == o["size"] and F_sha(got) == o["sha256"]


def F_sha(b):
    import hashlib
    return hashlib.sha256(b).hexdigest()


def test_many_strains_small_genomes_match_oracle():
    # config 4 shape (62 strains): ids with ~60-120 instances each
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=12_000, n=62, seed=4)
    bf, orc = _bf(seqs), Oracle(seqs)
    assert bf.simplify_stage(25, 150, 4) == orc.simplify_stage(25, 150, 4)
    (sa, pa), (sb, pb) = bf.state(), orc.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))


def test_cpp_class_surface_runs_on_the_gpu(tmp_path):
    # include/sibelia_amd/blockfinder.hpp (the reference's BlockFinder surface over the C ABI) from a plain g++ program:
    # hand/snp_k5 -- DOT text before and after the stage must be the reference's, byte for byte
    import os, subprocess
    from sibelia_amd.build import LIBDIR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    v = [x for x in VECS if x["name"] == "hand/snp_k5"][0]
    seqs = V.vector_input(v)
    src = tmp_path / "main.cpp"
    src.write_text('#include <fstream>\n#include "sibelia_amd/blockfinder.hpp"\n'
                   'struct Rec { std::string s; const std::string &GetSequence() const { return s; } };\n'
                   'int main(int, char **argv) { std::vector<Rec> v(%d);\n%s'
                   '  SyntenyFinderAMD::BlockFinder bf(v);\n'
                   '  { std::ofstream o(std::string(argv[1]) + ".0"); bf.SerializeCondensedGraph(5, o); }\n'
                   '  size_t calls = 0; size_t b = bf.PerformGraphSimplifications(5, 12, 4, [&](size_t, SyntenyFinderAMD::BlockFinder::State) { calls++; });\n'
                   '  { std::ofstream o(std::string(argv[1]) + ".1"); bf.SerializeCondensedGraph(5, o); }\n'
                   '  bool threw = false; try { bf.PerformGraphSimplifications(5, 12, 1, [](size_t, SyntenyFinderAMD::BlockFinder::State) { throw 7; }); } catch (int) { threw = true; }\n'
                   '  std::vector<SyntenyFinderAMD::BlockInstance> blk; bf.GenerateSyntenyBlocks(5, 4, 8, blk);\n'
                   '  { std::ofstream o(std::string(argv[1]) + ".blocks"); for (auto &x : blk) o << x.GetSignedBlockId() << " " << x.GetChrId() << " " << x.GetStart() << " " << x.GetEnd() << "\\n"; }\n'
                   '  std::string t0, t1, t2; bf.PostProcess(true, blk, t0, t1, t2, {"seq0", "seq1"});\n'
                   '  { std::ofstream o(std::string(argv[1]) + ".reports"); o << t0 << t1 << t2; }\n'
                   '  std::printf("bulges %%zu calls %%zu threw %%d\\n", b, calls, (int)threw); return 0; }\n'
                   % (len(seqs), "".join('  v[%d].s = "%s";\n' % (i, s.decode()) for i, s in enumerate(seqs))))
    exe = tmp_path / "main"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "include"), str(src), "-L", LIBDIR, "-lsibelia_amd",
                    "-Wl,-rpath," + LIBDIR, "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), str(tmp_path / "dot")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    stage = [o for o in v["outputs"] if o["cmd"].startswith("stage")][0]
    assert r.stdout.startswith("bulges %d calls " % stage["bulges"]) and r.stdout.strip().endswith("threw 1")
    dots = [o for o in v["outputs"] if o["cmd"] == "dot:5"]
    for i, o in enumerate(dots):
        got = open(str(tmp_path / ("dot.%d" % i)), "rb").read()
        assert F_sha(got) == o["sha256"], "DOT text %d differs from the reference" % i
    # GenerateSyntenyBlocks + PostProcess through the class = the same calls through the Python mirror (itself checked against the
    # reference's blocks: / write: fixtures) on the same sequence of operations
    py = _bf(seqs)
    py.list_edges(5); py.simplify_stage(5, 12, 4); py.list_edges(5); py.simplify_stage(5, 12, 1)
    blocks = py.generate_blocks(5, 4, 8)
    want = "".join("%d %d %d %d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert open(str(tmp_path / "dot.blocks")).read() == want
    _, texts = py.postprocess(["seq0", "seq1"], True)
    assert open(str(tmp_path / "dot.reports"), "rb").read() == b"".join(texts)


def test_config5_shape_matches_oracle():
    # config 5 shape (random DNA in 4 records, k = 5000, D = 15000: the rank-doubling long-k path) at a size the oracle
    # finishes in seconds; planted: a 40 kbp copy with one substitution (one bulge) and a 30 kbp exact duplicate
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.longk_case(8_000_000, 4)
    bf, orc = _bf(seqs), Oracle(seqs)
    a, b = bf.enumerate(5000), orc.enumerate(5000)
    assert a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all()
    assert bf.simplify_stage(5000, 15000, 4) == orc.simplify_stage(5000, 15000, 4) == 1
    (sa, pa), (sb, pb) = bf.state(), orc.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    a, b = bf.enumerate(5000), orc.enumerate(5000)
    assert a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all()


def test_config5_full_size_properties():
    # BASELINE.json config 5 at full size: 900 Mbp of random DNA in 4 records of 225 Mbp, k = 5000, D = 15000.  No CPU
    # implementation finishes this in test time; random DNA has no 5000-mer repeats of its own, so the result is fixed by
    # the planted segments alone and must equal what the oracle-checked 8 Mbp case gives: same bulge count, same ids and
    # instances before and after, sequences unchanged except for the one collapsed base, positions still the identity,
    # and the whole stage reproducible.
    import hashlib
    from sibelia_amd import workloads as W
    small = _bf(W.longk_case(8_000_000, 4))
    e0 = small.enumerate(5000)
    assert small.simplify_stage(5000, 15000, 4) == 1
    e1 = small.enumerate(5000)
    small.close()
    seqs = W.longk_case(900_000_000, 4)
    bf = _bf(seqs)
    bf.save_state()
    a0 = bf.enumerate(5000)
    assert (a0[0], len(a0[1]), len(a0[2])) == (e0[0], len(e0[1]), len(e0[2]))
    digests = []
    for _ in range(2):
        bf.restore_state()
        assert bf.simplify_stage(5000, 15000, 4) == 1
        st = bf.stats()
        assert st["strand_kmers"] == 2 * sum(len(s) - 5000 + 1 for s in seqs)
        s, p = bf.state()
        h = hashlib.sha256()
        for x, y in zip(s, p):
            h.update(x); h.update(np.ascontiguousarray(y).tobytes())
        digests.append(h.hexdigest())
    assert digests[0] == digests[1]
    a1 = bf.enumerate(5000)
    # (ids are lexicographic ranks: the terminal k-mers of the random records order differently in a different background;
    #  what is background-independent is the number of ids and how many instances each has)
    mult = lambda e: sorted(np.bincount(np.concatenate([e[1]["id"], e[2]["id"]]), minlength=e[0]).tolist())
    assert a1[0] == e1[0] and mult(a1) == mult(e1) and mult(a0) == mult(e0)
    assert [len(x) for x in s] == [len(x) for x in seqs]
    assert s[0][5000:45000] == s[1][1000:41000] and s[2][3000:33000] == s[3][7000:37000]        # the bulge is gone, the copies agree
    diff = [int(np.count_nonzero(np.frombuffer(x, np.uint8) != np.frombuffer(y, np.uint8))) for x, y in zip(s, seqs)]
    assert sum(diff) == 1 and diff[2] == diff[3] == 0
    assert all(y[0] == 0 and y[-1] == len(y) - 1 and bool((np.diff(y.astype(np.int64)) == 1).all()) for y in p)
    bf.close()


def test_config4_shape_full_size_is_reproducible(monkeypatch):
    # BASELINE.json config 4 shape at full size on one GPU: 62 strains x 4.6 Mbp (285 Mbp, 570 M strand-k-mers, ids with 124
    # instances), k = 25.  No CPU implementation finishes it in test time (the reference is superlinear in the number of
    # strains): the result must be reproducible and independent of the speculation window, and its first two strains are the
    # 2-strain workload whose reference result is a fixture -- the stage must at least leave their lengths within the indel budget.
    import hashlib
    from sibelia_amd import workloads as W, formats as F
    seqs = W.gen_strains(L0=4_600_000, n=62, seed=1)
    digests, counts = [], []
    monkeypatch.setenv("SBL_CHECK_DICTIONARY", "1")      # the reference's own invariant (IndexedSequence::Test) on the final graph of the full-size run
    for window in (0, 5000):
        bf = _bf(seqs)
        if window:
            bf.set_window(window)
        bulges = bf.simplify_stage(25, 150, 4)
        st = bf.stats()
        assert st["dict_mismatches"] == 0 and st["dict_checked"] > 500_000_000
        counts.append((bulges, st["bif_count"], st["instances"], st["replays"] - st["grow_replays"]))
        s, p = bf.state()
        digests.append(hashlib.sha256(F.state_bytes(bulges, s, p)).hexdigest())
        assert all(int(y.max()) < len(x0) and bool((np.diff(y.astype(np.int64)) >= 0).all()) for x0, y in zip(seqs, p))      # original positions stay monotone
        bf.close()
    assert digests[0] == digests[1] and counts[0][:3] == counts[1][:3]
    assert counts[0][0] > 2_000_000 and counts[0][1] > 1_000_000


# ---- full-size reference pins of configs 4 and 5 (tests/golden/big_vectors.json: sha256 + size + counts of what the UNMODIFIED reference
# wrote, generated once by tests/golden/gen/make_big_golden.py -- 30 min to hours of reference time and up to ~35 GB of host memory per
# input; the outputs themselves are gigabytes and are hashed as a stream on both sides)
def _big_vector(name):
    import json
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_vectors.json")
    if not os.path.exists(p):
        pytest.skip("tests/golden/big_vectors.json not generated")
    vs = [v for v in json.load(open(p))["vectors"] if v["name"] == name]
    if not vs:
        pytest.skip("no full-size reference fixture for " + name + " (make_big_golden.py " + name + ")")
    return vs[0]


def _big_input(v):
    from sibelia_amd import workloads as W
    spec = v["input"]
    seqs = W.gen_strains(**spec["args"]) if spec["kind"] == "gen_strains" else W.longk_case(spec["args"]["total"], spec["args"]["nrec"])
    assert W.input_digest(seqs) == v["input_sha256"], "input generator drifted for " + v["name"]
    return seqs


def _enum_digest(bf, k):
    import hashlib, struct
    bc, pos, neg = bf.enumerate(k)
    h = hashlib.sha256(struct.pack("<I", bc))
    size = 4
    for a in (pos, neg):
        cols = np.stack([a["id"], a["chr"], a["pos"]], axis=1).astype("<u4") if len(a) else np.zeros((0, 3), "<u4")
        h.update(struct.pack("<Q", len(a))); h.update(cols.tobytes())
        size += 8 + cols.nbytes
    return h.hexdigest(), size, bc


def _stage_digest(bf, k, D, it):
    import hashlib, struct
    bulges = bf.simplify_stage(k, D, it)
    vs, vp = bf.state_views()
    h = hashlib.sha256(struct.pack("<QI", bulges, len(vs)))
    size = 12
    for x, y in zip(vs, vp):
        h.update(struct.pack("<Q", len(x))); h.update(x); h.update(y)
        size += 8 + 5 * len(x)
    return h.hexdigest(), size, bulges


def _replay_big(name, check_dictionary=False, monkeypatch=None):
    v = _big_vector(name)
    seqs = _big_input(v)
    if check_dictionary:
        monkeypatch.setenv("SBL_CHECK_DICTIONARY", "1")
    bf = _bf(seqs)
    del seqs
    try:
        for o in v["outputs"]:
            p = o["cmd"].split(":")
            if p[0] == "enum":
                sha, size, bc = _enum_digest(bf, int(p[1]))
                assert (bc, size) == (o["bif_count"], o["size"]), "%s %s: %d ids / %d bytes, the reference has %d / %d" % (name, o["cmd"], bc, size, o["bif_count"], o["size"])
            else:
                sha, size, bulges = _stage_digest(bf, int(p[1]), int(p[2]), int(p[3]))
                assert (bulges, size) == (o["bulges"], o["size"]), "%s %s: %d bulges / %d bytes, the reference has %d / %d" % (name, o["cmd"], bulges, size, o["bulges"], o["size"])
                if check_dictionary:
                    st = bf.stats()
                    assert st["dict_mismatches"] == 0 and st["dict_checked"] == 2 * sum(max(0, len(x) - int(p[1]) + 1) for x in bf.state_views()[0])
            assert sha == o["sha256"], "%s: %s differs from the unmodified reference's output" % (name, o["cmd"])
    finally:
        bf.close()


def test_config4_62_strains_460k_matches_reference(monkeypatch):
    # BASELINE.json config 4 (62 strains, k = 25) at a tenth of the record length: enumeration and post-stage state sha256-identical to
    # the unmodified reference's (~30 min there), with the reference's own _DEBUG invariant checked on the final graph on the way
    _replay_big("synth/strains62_460k", check_dictionary=True, monkeypatch=monkeypatch)


def test_config4_full_size_matches_reference():
    # ... and at full size (62 x 4.6 Mbp: hours of reference time) when that fixture has been generated
    _replay_big("synth/strains62_4600k")


def test_config5_full_size_matches_reference():
    # BASELINE.json config 5 at full size: 4 x 225 Mbp of random DNA, k = 5000, D = 15000 -- enumeration before, the stage, enumeration
    # after, each sha256-identical to the unmodified reference's (int32 suffix array of 1.8 G suffixes, ~30 GB, about an hour)
    _replay_big("synth/random4x225M_k5000")


def test_dictionary_invariant_of_the_reference_holds_and_is_sharp(monkeypatch):
    """IndexedSequence::Test() (reference src/indexedsequence.cpp:74-103, _DEBUG builds only): after every collapse, at every window position
    the stored bifurcation id is what the dictionary of the initial marking says about the k-mer spelled there now.  With
    SBL_CHECK_DICTIONARY=1 the stage checks it on its final graph (k_dict_check): every window of the post-stage sequences is
    covered, none differs -- and ONE mark changed behind the stage's back (SBL_TEST_CORRUPT_MARK) is noticed."""
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=300_000, n=8, seed=12, inv_min=3000, inv_max=12000)
    monkeypatch.setenv("SBL_CHECK_DICTIONARY", "1")
    bf = _bf(seqs)
    try:
        bf.save_state()
        for k, D in ((25, 150), (16, 100), (32, 200)):
            bf.restore_state()
            assert bf.simplify_stage(k, D, 4) > 0
            st = bf.stats()
            assert st["dict_mismatches"] == 0
            assert st["dict_checked"] == 2 * sum(max(0, len(x) - k + 1) for x in bf.state()[0]) > 0
        monkeypatch.setenv("SBL_TEST_CORRUPT_MARK", "12345")
        bf.restore_state()
        bf.simplify_stage(25, 150, 4)
        assert 1 <= bf.stats()["dict_mismatches"] <= 2
    finally:
        bf.close()


def test_config3_full_size_fine_cascade_matches_reference():
    # BASELINE.json config 3 at full size: 8 strains x 4.6 Mbp through the whole `-s fine` cascade (30,150)(100,500)(500,1500),
    # then the enumeration at the last k, the synteny blocks and the three reports -- every output sha256-identical to what the
    # unmodified reference produced on this input (about 12 minutes there)
    gold = [v for v in VECS if v["name"] == "synth/strains8_4600k_fine"]
    if not gold:
        pytest.skip("fixture synth/strains8_4600k_fine not generated")
    V.replay(gold[0], _bf)


def test_bucket_overflow_and_buffer_growth_paths(monkeypatch):
    # the radix-bucketed table re-buckets with a longer prefix when a bucket holds more distinct k-mers than its LDS table, and
    # grows its classification buffers on demand: force both (2 bucket bits for 1.2 M positions, room for 16 pairs)
    from sibelia_amd import workloads as W
    from sibelia_amd.dist import LocalShardedFinder
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=300_000, n=4, seed=17, inv_min=3000, inv_max=12000)
    want = Oracle(seqs).enumerate(20)
    monkeypatch.setenv("SBL_TEST_BUCKET_BITS", "2")
    monkeypatch.setenv("SBL_TEST_MAXPAIRS", "16")
    for make in (_bf, lambda s: LocalShardedFinder(s, [0, 0, 0])):
        bf = make(seqs)
        got = bf.enumerate(20)
        assert got[0] == want[0] and (got[1] == want[1]).all() and (got[2] == want[2]).all()
        bf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,L0,k", [(11, 3, 20_000, 12), (12, 5, 8_000, 9), (13, 2, 40_000, 15), (14, 4, 6_000, 32), (15, 3, 5_000, 5)])
def test_small_block_index_equals_the_general_child_index(seed, n, L0, k):
    # GenerateSyntenyBlocks(k, k, k) on the raw graph (what -v / --allstages do): thousands of tiny candidate blocks, indexed by
    # k_tiny_enumerate (one launch each); SBL_NO_TINY_INDEX=1 sends them through the general child enumeration; the oracle is third
    import os
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=L0, n=n, seed=seed, snp=0.03, indel_every=300, inv_min=200, inv_max=800)
    if seed == 12:      # non-ACGT characters inside candidate blocks: the small-block kernel must decline and leave rand() to the general path
        seqs = [s[:1000] + b"N" + s[1001:3000] + b"RY" + s[3002:] for s in seqs]
    fa, fb, orc = _bf(seqs), _bf(seqs), Oracle(seqs)
    a = fa.generate_blocks(k, k, k)
    os.environ["SBL_NO_TINY_INDEX"] = "1"
    try:
        b = fb.generate_blocks(k, k, k)
    finally:
        del os.environ["SBL_NO_TINY_INDEX"]
    c = orc.generate_blocks(k, k, k)
    assert len(a) == len(b) == len(c) and len(a) > 0
    for f in ("id", "chr", "start", "end"):
        assert (a[f] == b[f]).all() and (a[f] == c[f]).all()
    # GlueStripes on these thousands of instances: the worklist (product) against the oracle's restatement of the reference's
    # rescan-per-merge procedure (oracle/output_oracle.cpp) -- blocks and the three report texts
    names = ["seq%d" % i for i in range(len(seqs))]
    ga, ta = fa.postprocess(names, True)
    gc, tc = orc.postprocess(c, names, True)
    assert len(ga) == len(gc) and len(ga) < len(a)
    for f in ("id", "chr", "start", "end"):
        assert (ga[f] == gc[f]).all()
    assert list(ta) == list(tc)


@pytest.mark.gpu
def test_positions_clamp_at_a_chromosome_end_in_a_later_stage():
    # Found by tools/stress.py (seed 110467): a collapse at the very end of a chromosome in the SECOND stage clamps the interpolated
    # positions to the position of the closing separator, which the reference sets to the record's CURRENT length when it rebuilds
    # the sequence for that stage (dnasequence.cpp:96) -- not to the original length.  The oracle agrees with the unmodified
    # reference on this input (sha256 of both stages' states compared with oracle/_ref/ref_dump when the case was found).
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=13939, n=8, seed=110467, snp=0.03, indel_every=1000, inv_min=139, inv_max=696)
    bf, orc = _bf(seqs), Oracle(seqs)
    for k, D in ((31, 124), (36, 174), (40, 250)):
        assert bf.simplify_stage(k, D, 4) == orc.simplify_stage(k, D, 4)
        (sa, pa), (sb, pb) = bf.state(), orc.state()
        assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb)), (k, D)


@pytest.mark.parametrize("k", [40, 100, 700])
def test_long_k_active_set_doubling_equals_the_full_doubling(k, monkeypatch):
    """k > 32: rank doubling over the still-active suffixes only (the default) against the plain doubling that sorts every suffix in
    every round (SBL_LONGK_NO_DISCARD=1) and against the oracle, on related strains (most suffixes stay active) with a random record
    and a short one appended (suffixes that settle early; a record shorter than k)"""
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=40_000, n=3, seed=77, inv_min=400, inv_max=2000) + W.random_dna(30_000, 1, seed=9) + [b"ACGTTGCA" * 4]
    a = _bf(seqs).enumerate(k)                                # default: the variant is chosen after the first round
    monkeypatch.setenv("SBL_LONGK_FORCE_ACTIVE", "1")
    f = _bf(seqs).enumerate(k)                                # active suffixes only, whatever their share
    monkeypatch.delenv("SBL_LONGK_FORCE_ACTIVE")
    monkeypatch.setenv("SBL_LONGK_NO_DISCARD", "1")
    b = _bf(seqs).enumerate(k)                                # every suffix in every round
    c = Oracle(seqs).enumerate(k)
    for x in (f, b, c):
        assert a[0] == x[0] and (a[1] == x[1]).all() and (a[2] == x[2]).all()
    assert a[0] > 0


@pytest.mark.parametrize("k", [40, 64])
def test_long_k_terminal_windows_listed_twice_change_nothing(k, monkeypatch):
    """k_lk_cand_keys (longk.hip) appends the first and the last window of both strands of every record to the candidate list without
    looking whether they are already there: a terminal window whose suffix is still ACTIVE (the k-mer occurs elsewhere) is listed twice,
    and in a record of exactly k characters first and last window are the same one -- three entries.  Harmless because a terminal
    window is always a bifurcation ('#' before or after it) and marks are written by value; this test pins that: records of exactly k
    characters whose k-mer (or its reverse complement) also lies inside longer records, active-set path forced, against the oracle."""
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    x = W.random_dna(k, 1, seed=5)[0]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rc = x.translate(comp)[::-1]
    a, b = W.random_dna(3000, 2, seed=6)
    seqs = [x, a[:1500] + x + a[1500:], x, b[:700] + rc + b[700:], rc, x + b[:900], a[:800] + x]
    want = Oracle(seqs).enumerate(k)
    assert want[0] > 0
    for env in (None, "SBL_LONGK_FORCE_ACTIVE", "SBL_LONGK_NO_DISCARD"):
        if env:
            monkeypatch.setenv(env, "1")
        got = _bf(seqs).enumerate(k)
        if env:
            monkeypatch.delenv(env)
        assert got[0] == want[0] and (got[1] == want[1]).all() and (got[2] == want[2]).all(), env


def test_long_k_stage_on_random_records_takes_the_active_set_path():
    """unrelated random records with one planted repeat: nearly every suffix is unique after 16 characters, the doubling continues over the
    few that are not -- result equal to the oracle's, state included"""
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    seqs = W.longk_case(400_000, 4, seed=3)
    bf, orc = _bf(seqs), Oracle(seqs)
    for k, D in ((200, 600), (5000, 15000)):
        assert bf.simplify_stage(k, D, 4) == orc.simplify_stage(k, D, 4)
        (sa, pa), (sb, pb) = bf.state(), orc.state()
        assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
