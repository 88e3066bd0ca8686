"""The reference's own PROGRAM over this library (drop-in boundary, end to end).

oracle/_ref/sibelia_dropin is the reference's src/sibelia.cpp -- command line, FASTA reader, post-processor, writers, all
unmodified reference code -- linked with integration/blockfinder_amd.cpp + libsibelia_amd.so instead of the reference's five
BlockFinder translation units (oracle/build_dropin.sh; the binary contains reference objects and is therefore built into
oracle/_ref/, never committed).  Every file it writes and its standard output (progress bars included) must be byte-identical
to what the unmodified reference program wrote for the same command line (tests/golden/dropin_cases.json, written by
tests/golden/gen/make_dropin_golden.py from oracle/_ref/sibelia_ref).
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "gen"))
from make_dropin_golden import run_case      # noqa: E402

DROPIN = os.path.join(ROOT, "oracle", "_ref", "sibelia_dropin")
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "dropin_cases.json")))["cases"]

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_program_over_the_library(case, tmp_path):
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/sibelia_dropin not built (oracle/build_dropin.sh needs /root/reference)")
    rc, stdout_sha, files, stdout, stderr = run_case(DROPIN, case["input"], case["args"], str(tmp_path))
    assert rc == case["returncode"], stderr.decode(errors="replace")[-2000:]
    assert sorted(files) == sorted(case["files"])
    wrong = [name for name in sorted(files) if files[name] != case["files"][name]]
    assert not wrong, "files differ from the reference program's: %s" % wrong
    assert stdout_sha == case["stdout_sha256"], stdout.decode(errors="replace")[-2000:]
