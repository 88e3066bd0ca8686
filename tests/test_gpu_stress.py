"""A bounded run of tools/stress.py inside the GPU suite: randomised strain sets (2 - 12 strains, 3 - 80 kbp, k 15 - 40, random D,
SNP and indel rates, sometimes a pinned commit window) through the HIP path and the oracle, every post-stage state compared bit for
bit.  Round 2's last real parity bug (the closing separator's position in later stages) was found by exactly this loop -- run by
hand; it now runs with every GPU test session: three-stage cascades, and the same through three virtual ranks (sharded enumeration).
The seeds move with the date so that successive sessions cover new cases; a failing seed is printed and reproduces with
`python tools/stress.py 1 <seed>` (STAGES=3 / SHARD=3)."""
import datetime
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _first_seed(base):
    d = datetime.date.today()
    return base + 1000 * (d.toordinal() % 1000)


def test_three_stage_cascades_match_the_oracle():
    import stress
    lines = []
    done, bad = stress.run(budget=45.0, seed=_first_seed(2_000_000), stages3=True, log=lines.append)
    assert done >= 5, "\n".join(lines)
    assert not bad, "mismatching seeds %s\n%s" % (bad, "\n".join(lines))


def test_sharded_enumeration_with_three_virtual_ranks_matches_the_oracle():
    import stress
    lines = []
    done, bad = stress.run(budget=30.0, seed=_first_seed(3_000_000), nshard=3, log=lines.append)
    assert done >= 3, "\n".join(lines)
    assert not bad, "mismatching seeds %s\n%s" % (bad, "\n".join(lines))


def test_blocks_and_reports_after_random_stages_match_the_oracle():
    import stress
    lines = []
    done, bad = stress.run(budget=30.0, seed=_first_seed(4_000_000), n2=True, log=lines.append)
    assert done >= 3, "\n".join(lines)
    assert not bad, "mismatching seeds %s\n%s" % (bad, "\n".join(lines))
