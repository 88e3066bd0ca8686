"""FASTA text -> HBM-resident state on the device (sbl_load_fasta) vs the reference's FASTAReader.

tests/golden/fasta_cases.json holds what the UNMODIFIED reference reader (src/fasta.cpp:23-104, through oracle/_ref/ref_dump)
makes of each text: records (description, upper-cased sequence) or its parse error, message and line number included."""
import base64
import gzip
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "fasta_cases.json")))["cases"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_fasta_load_matches_reference_reader(name, tmp_path):
    from sibelia_amd import BlockFinder, SibeliaError
    case = CASES[name]
    fa = str(tmp_path / "in.fa")
    open(fa, "wb").write(base64.b64decode(case["text_b64"]))
    if "error" in case:
        with pytest.raises(SibeliaError) as ei:
            BlockFinder.from_fasta(fa, device=0)
        assert case["error"].replace("<file>", fa) in str(ei.value)
        return
    bf = BlockFinder.from_fasta(fa, device=0)
    seqs, pos = bf.state()
    assert bf.record_names() == [r["name"] for r in case["records"]]
    assert [s.decode() for s in seqs] == [r["seq"] for r in case["records"]]
    assert all(np.array_equal(p, np.arange(len(s), dtype=np.uint32)) for s, p in zip(seqs, pos))      # originalPos_ = identity
    bf.close()


@pytest.mark.parametrize("data", ["Helicobacter_pylori.fa.gz", "Staphylococcus_aureus_pair.fa.gz"])
def test_fasta_load_of_real_inputs_equals_in_memory_load(data, tmp_path):
    # the two example inputs of the reference: loading the FILE on the device must give the state (and, through the ambiguity
    # list, the rand()-sanitised stage result) that loading the parsed records gives
    from sibelia_amd import BlockFinder, workloads as W
    src = os.path.join(HERE, "golden", "data", data)
    fa = str(tmp_path / "in.fa")
    open(fa, "wb").write(gzip.open(src, "rb").read())
    names, seqs = W.read_fasta(src)
    a, b = BlockFinder.from_fasta(fa, device=0), BlockFinder(seqs, device=0)
    assert a.record_names() == names
    (sa, pa), (sb, pb) = a.state(), b.state()
    assert sa == sb == [bytes(s) for s in seqs] and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    assert a.simplify_stage(25, 150, 4) == b.simplify_stage(25, 150, 4)
    (sa, pa), (sb, pb) = a.state(), b.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
