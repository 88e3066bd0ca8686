"""A bounded run of tools/stress.py inside the GPU suite: randomised strain sets (2 - 12 strains, 3 - 80 kbp, k 15 - 40, random D,
SNP and indel rates, sometimes a pinned commit window) through the HIP path and the oracle, every post-stage state compared bit for
bit.  Round 2's last real parity bug (the closing separator's position in later stages) was found by exactly this loop -- run by
hand; it now runs with every GPU test session: three-stage cascades, and the same through three virtual ranks (sharded enumeration).

Deterministic by default: fixed first seeds and a fixed NUMBER of cases (no wall-clock budget), so a red run reproduces from the commit
alone.  Exploratory sessions move the seeds with `SBL_STRESS_SEED=<offset>` (or `SBL_STRESS_SEED=date`) and lengthen the runs with
`SBL_STRESS_CASES=<n>`.  Every case -- seed, parameters, outcome -- is appended to gpurun_out/stress_seeds.txt, which gpurun pulls back;
a failing seed reproduces with `python tools/stress.py 1 <seed>` (STAGES=3 / SHARD=3 / N2=1)."""
import datetime
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _first_seed(base):
    off = os.environ.get("SBL_STRESS_SEED", "0")
    if off == "date":
        off = 1000 * (datetime.date.today().toordinal() % 1000)
    return base + int(off)


def _cases(default):
    return int(os.environ.get("SBL_STRESS_CASES", default))


def _run(name, **kw):
    import stress
    lines = []
    done, bad = stress.run(log=lines.append, **kw)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "stress_seeds.txt"), "a") as f:
            f.write("# %s first seed %d\n%s\n" % (name, kw["seed"], "\n".join(lines)))
    except OSError:
        pass
    assert done == kw["count"], "\n".join(lines)
    assert not bad, "mismatching seeds %s\n%s" % (bad, "\n".join(lines))


def test_three_stage_cascades_match_the_oracle():
    _run("three_stage", seed=_first_seed(2_000_000), count=_cases(12), stages3=True)


def test_sharded_enumeration_with_three_virtual_ranks_matches_the_oracle():
    _run("sharded3", seed=_first_seed(3_000_000), count=_cases(8), nshard=3)


def test_blocks_and_reports_after_random_stages_match_the_oracle():
    _run("blocks_reports", seed=_first_seed(4_000_000), count=_cases(8), n2=True)


def test_transactions_of_several_collapses_side_by_side_match_the_oracle(monkeypatch):
    """tools/stress.py PARKY=1: a dozen strains, k 15 - 20, 3 - 8 % SNPs -- the regime of seed 93194, where a parked transaction's private
    list sizes were read by a neighbour (round 6); bulge COUNTS are compared as well as states.  Parked after every collapse."""
    monkeypatch.setenv("PARKY", "1")
    monkeypatch.setenv("SBL_PARK", "1")
    _run("parky", seed=_first_seed(5_000_000), count=_cases(10))
