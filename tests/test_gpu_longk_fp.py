"""k > 32 through window fingerprints (sibelia_amd/csrc/longk_fp.hip, round 6; replaces the suffix-array group scan of
IndexedSequence::EnumerateBifurcationsSArrayInRAM, reference src/vertexenumeration.cpp:263-364, for long vertex sizes).

The path must return what the oracle (and the exact rank doubling, SBL_LONGK_DOUBLING=1) return -- ids, instances, order -- for every
k, including the ones that exercise its arithmetic: k below / at / above the tile of the block scans (1024) and its multiples, k = 33
(one symbol more than a packed word), records shorter than k, of exactly k, records that are each other's reverse complement,
palindromic k-mers (even k), and many short records.  With SBL_TEST_WEAK_FP=n the fingerprints keep n bits only: different k-mers
collide on purpose, the verification must notice, the stage falls back to the doubling and still returns the same result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _bf(seqs):
    from sibelia_amd import BlockFinder
    return BlockFinder(seqs, device=0)


def _same(a, b, what=""):
    assert a[0] == b[0], "bifurcation counts differ (%d / %d) %s" % (a[0], b[0], what)
    assert len(a[1]) == len(b[1]) and (a[1] == b[1]).all() and len(a[2]) == len(b[2]) and (a[2] == b[2]).all(), "instances differ " + what


def _enumerate(seqs, k, monkeypatch, doubling=False, weak=None):
    if doubling:
        monkeypatch.setenv("SBL_LONGK_DOUBLING", "1")
    if weak is not None:
        monkeypatch.setenv("SBL_TEST_WEAK_FP", str(weak))
    bf = _bf(seqs)
    try:
        r = bf.enumerate(k)
        path = int(bf.stats()["longk_path"])
    finally:
        bf.close()
        monkeypatch.delenv("SBL_LONGK_DOUBLING", raising=False)
        monkeypatch.delenv("SBL_TEST_WEAK_FP", raising=False)
    return r, path


def _mixed(seed=77):
    from sibelia_amd import workloads as W
    x = W.random_dna(700, 1, seed=seed + 1)[0]
    return (W.gen_strains(L0=40_000, n=3, seed=seed, inv_min=400, inv_max=2000) + W.random_dna(30_000, 1, seed=seed + 2)
            + [b"ACGTTGCA" * 4, x, x.translate(COMP)[::-1], x[:33], x[:34]])


@pytest.mark.parametrize("k", [33, 34, 40, 63, 64, 65, 100, 500, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 5000])
def test_fingerprint_enumeration_equals_oracle_and_doubling(k, monkeypatch):
    from oracle.oracle import Oracle
    seqs = _mixed()
    want = Oracle(seqs).enumerate(k)
    got, path = _enumerate(seqs, k, monkeypatch)
    assert path == 1, "the fingerprint path did not run (longk_path = %d)" % path
    _same(got, want, "fingerprints against the oracle, k = %d" % k)
    dbl, path = _enumerate(seqs, k, monkeypatch, doubling=True)
    assert path == 2
    _same(dbl, want, "doubling against the oracle, k = %d" % k)
    assert want[0] > 0


@pytest.mark.parametrize("k", [34, 40, 64])
def test_records_of_exactly_k_their_reverse_complements_and_palindromes(k, monkeypatch):
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    x = W.random_dna(k, 1, seed=5)[0]
    rc = x.translate(COMP)[::-1]
    half = W.random_dna(k // 2, 1, seed=8)[0]
    pal = half + half.translate(COMP)[::-1]                     # its own reverse complement (even k)
    a, b = W.random_dna(3000, 2, seed=6)
    seqs = [x, a[:1500] + x + a[1500:], x, b[:700] + rc + b[700:], rc, x + b[:900], a[:800] + x, pal, a[:300] + pal + b[:300], b[1000:1400] + pal + a[2000:2300], pal[:k - 1]]
    want = Oracle(seqs).enumerate(k)
    assert want[0] > 0
    got, path = _enumerate(seqs, k, monkeypatch)
    assert path == 1
    _same(got, want, "k = %d" % k)


@pytest.mark.parametrize("weak", [6, 14])
def test_colliding_fingerprints_are_caught_by_the_verification(weak, monkeypatch):
    """fingerprints cut to a few bits: groups of DIFFERENT k-mers merge, a merged bifurcation group fails its comparison with the
    representative, and the exact rank doubling takes over (longk_path = 3) -- same result"""
    from oracle.oracle import Oracle
    seqs = _mixed(seed=31)
    for k in (40, 300):
        want = Oracle(seqs).enumerate(k)
        got, path = _enumerate(seqs, k, monkeypatch, weak=weak)
        assert path == 3, "colliding fingerprints went unnoticed (longk_path = %d)" % path
        _same(got, want, "weak fingerprints, k = %d" % k)


@pytest.mark.parametrize("dense_marks", [False, True])
def test_many_short_records_and_a_cascade(monkeypatch, dense_marks):
    """180-record style input (records from 100 bp) through a long-k cascade: state after every stage against the oracle.  The ordered mark
    lists of a stage come from sorting the few member positions (the default where there are few) or from the scan of the dense mark
    arrays (SBL_FP_DENSE_MARKS=1): the same lists either way."""
    from oracle.oracle import Oracle
    if dense_marks:
        monkeypatch.setenv("SBL_FP_DENSE_MARKS", "1")
    from sibelia_amd import workloads as W
    base = W.gen_strains(L0=60_000, n=4, seed=19, snp=0.02, inv_min=500, inv_max=3000)
    rng = np.random.default_rng(4)
    seqs = []
    for s in base:
        p = 0
        while p < len(s):
            q = min(len(s), p + int(rng.integers(100, 9000)))
            seqs.append(s[p:q]); p = q
    bf, orc = _bf(seqs), Oracle(seqs)
    try:
        for k, D in ((30, 150), (100, 500), (500, 1500)):
            assert bf.simplify_stage(k, D, 4) == orc.simplify_stage(k, D, 4)
            if k > 32:
                assert int(bf.stats()["longk_path"]) == 1
            (sa, pa), (sb, pb) = bf.state(), orc.state()
            assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb)), (k, D)
    finally:
        bf.close()


@pytest.mark.parametrize("k", [40, 300, 1100])
def test_ranking_by_refinement_rounds_equals_ranking_by_counting(k, monkeypatch):
    """a few hundred bifurcation k-mers are ranked in one launch (k_fp_rank_small); SBL_FP_RANK_ROUNDS=1 sends the same set through the
    MSD refinement rounds that larger sets take"""
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=6_000, n=3, seed=5, snp=0.004, inv_min=200, inv_max=600)
    want = Oracle(seqs).enumerate(k)
    assert 0 < want[0] <= 1024
    a, path = _enumerate(seqs, k, monkeypatch)
    assert path == 1
    monkeypatch.setenv("SBL_FP_RANK_ROUNDS", "1")
    b, path = _enumerate(seqs, k, monkeypatch)
    monkeypatch.delenv("SBL_FP_RANK_ROUNDS")
    _same(a, want, "counting"); _same(b, want, "rounds")


def test_random_long_k_cases(monkeypatch):
    """randomised: strain sets with their own record splits, random k in 33 .. 1500"""
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    rng = np.random.default_rng(2026)
    for case in range(12):
        n, L0 = int(rng.integers(2, 7)), int(rng.integers(2_000, 30_000))
        k = int(rng.choice([33, 35, 48, 64, 96, 127, 128, 129, 200, 256, 511, 777, 1024, 1500]))
        seqs = W.gen_strains(L0=L0, n=n, seed=1000 + case, snp=float(rng.choice([0.002, 0.01, 0.05])), inv_min=max(50, L0 // 100), inv_max=max(200, L0 // 20))
        if rng.random() < 0.5:
            seqs = seqs + [seqs[0][: int(rng.integers(1, 2 * k))], seqs[-1].translate(COMP)[::-1][: int(rng.integers(k, 4 * k))]]
        want = Oracle(seqs).enumerate(k)
        got, path = _enumerate(seqs, k, monkeypatch)
        assert path == 1
        _same(got, want, "case %d n %d L0 %d k %d" % (case, n, L0, k))


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_fingerprint_table_sharded_over_virtual_ranks(nranks, monkeypatch):
    """One job on several GPUs: the fingerprint table is sharded by hash prefix like the k <= 32 table (one all-to-all of the 16-B
    records, owners classify and verify their buckets, representatives all-gathered and ranked everywhere, member marks all-gathered).
    Every rank returns the single-GPU result, says which path ran, and has sent something; colliding fingerprints send ALL ranks to the
    sharded rank doubling together."""
    from oracle.oracle import Oracle
    from sibelia_amd.dist import LocalShardedFinder
    seqs = _mixed(seed=5)
    for k in (33, 100, 1024, 3000):
        want = Oracle(seqs).enumerate(k)
        f = LocalShardedFinder(seqs, [0] * nranks)
        try:
            _same(f.enumerate(k), want, "%d ranks, k = %d" % (nranks, k))
            st = f.stats()
            assert all(int(x["longk_path"]) == 1 for x in st), [int(x["longk_path"]) for x in st]
            assert all(x["exchange_bytes"] > 0 for x in st)
        finally:
            f.close()
    monkeypatch.setenv("SBL_TEST_WEAK_FP", "8")
    f = LocalShardedFinder(seqs, [0] * nranks)
    try:
        _same(f.enumerate(100), Oracle(seqs).enumerate(100), "%d ranks, weak fingerprints" % nranks)
        assert all(int(x["longk_path"]) == 3 for x in f.stats())
    finally:
        f.close()


def test_sharded_long_k_cascade_matches_the_oracle(monkeypatch):
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    from sibelia_amd.dist import LocalShardedFinder
    seqs = W.gen_strains(L0=50_000, n=4, seed=23, snp=0.02, inv_min=500, inv_max=3000) + [b"ACGT" * 9]
    f, orc = LocalShardedFinder(seqs, [0, 0, 0]), Oracle(seqs)
    try:
        for k, D in ((30, 150), (100, 500), (500, 1500)):
            assert f.simplify_stage(k, D, 4) == orc.simplify_stage(k, D, 4)
            (sa, pa), (sb, pb) = f.state(), orc.state()
            assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb)), (k, D)
    finally:
        f.close()


def test_bucket_overflow_and_buffer_growth_paths_of_the_fingerprint_table(monkeypatch):
    """the table re-buckets with a longer prefix when a bucket holds more distinct fingerprints than its LDS table, and grows its pair
    buffer on demand: force both (2 bucket bits for 1.2 M windows, room for 16 pairs) -- one GPU and three virtual ranks (where the
    decision to re-bucket is collective)"""
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    from sibelia_amd.dist import LocalShardedFinder
    seqs = W.gen_strains(L0=300_000, n=4, seed=17, inv_min=3000, inv_max=12000)
    want = Oracle(seqs).enumerate(48)
    monkeypatch.setenv("SBL_TEST_BUCKET_BITS", "2")
    monkeypatch.setenv("SBL_TEST_MAXPAIRS", "16")
    for make in (_bf, lambda s: LocalShardedFinder(s, [0, 0, 0])):
        bf = make(seqs)
        try:
            _same(bf.enumerate(48), want, "forced re-bucketing / buffer growth")
            st = bf.stats()
            assert all(int(x["longk_path"]) == 1 for x in (st if isinstance(st, list) else [st]))
        finally:
            bf.close()


@pytest.mark.parametrize("k", [33, 64, 200])
def test_low_complexity_sequence_at_long_k(k, monkeypatch):
    """homopolymers and tandem repeats: ONE k-mer with thousands of occurrences (a bucket far larger than the member stage: the direct
    path of k_fp_classify), k-mers that are their own reverse complement over long stretches ((AT)n at even k), every window a bifurcation"""
    from oracle.oracle import Oracle
    from sibelia_amd import workloads as W
    rnd = W.random_dna(4000, 3, seed=41)
    seqs = [b"A" * 6000 + rnd[0] + b"T" * 3000, b"AT" * 2500 + rnd[1] + b"ACG" * 1500, rnd[2][:1500] + b"A" * 2500 + rnd[2][1500:] + b"CCGG" * 700, b"AT" * 40, b"A" * (k + 1)]
    want = Oracle(seqs).enumerate(k)
    got, path = _enumerate(seqs, k, monkeypatch)
    assert path == 1
    _same(got, want, "low complexity, k = %d" % k)
    assert want[0] > 0
