"""Pin of the golden fixtures to the UNMODIFIED reference, re-derived here when the reference build exists.

oracle/build_ref.sh compiles the reference's own sources in place (gcc/g++ directly, no CMake) into
oracle/_ref/ref_dump; this test re-runs a sample of tests/golden/vectors.json through that binary and
requires byte-identical outputs (sha256).  Together with tests/test_oracle_golden.py (oracle == fixtures on
all vectors) this closes the chain  reference -> fixtures -> oracle -> HIP path.
Skipped where oracle/_ref/ref_dump does not exist and cannot be built (no /root/reference)."""
import hashlib
import os
import subprocess

import pytest

from sibelia_amd import workloads as W
from tests import vectors as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")


def _have_ref():
    if os.path.exists(REF_DUMP):
        return True
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")], check=True, capture_output=True)
        return os.path.exists(REF_DUMP)
    return False


VECS = V.load_vectors()
# every hand case, every 12th small case, the H. pylori k=25 run (~10 s in all)
SAMPLE = [v for i, v in enumerate(VECS) if v["name"].startswith("hand/") or (v["name"].startswith("small/") and i % 12 == 0)
          or v["name"] == "real/hpylori_k25"]


@pytest.mark.parametrize("v", SAMPLE, ids=[v["name"] for v in SAMPLE])
def test_fixture_is_what_the_reference_build_produces(v, tmp_path):
    if not _have_ref():
        pytest.skip("no reference build (oracle/_ref/ref_dump) and no /root/reference to build it from")
    seqs = V.vector_input(v)
    fa = str(tmp_path / "in.fa")
    W.write_fasta(fa, seqs)
    cmds = [o["cmd"] for o in v["outputs"]]
    subprocess.run([REF_DUMP, fa, str(tmp_path / "o")] + cmds, check=True, stderr=subprocess.DEVNULL, timeout=120)
    for i, o in enumerate(v["outputs"]):
        b = open(str(tmp_path / ("o.%d.out" % i)), "rb").read()
        assert len(b) == o["size"] and hashlib.sha256(b).hexdigest() == o["sha256"], (v["name"], o["cmd"])


def test_reference_program_over_the_library_fails_loudly_without_a_gpu(tmp_path):
    """oracle/_ref/sibelia_dropin (tests/test_gpu_dropin.py) is the reference's main over libsibelia_amd.so: here, without a device,
    it must stop with the library's error through the reference's own catch block -- no CPU path behind the binding."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_dropin.py runs the program for real")
    dropin = os.path.join(ROOT, "oracle", "_ref", "sibelia_dropin")
    if not os.path.exists(dropin):
        if not os.path.isdir("/root/reference/src"):
            pytest.skip("no oracle/_ref/sibelia_dropin and no /root/reference to build it from")
        subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_dropin.sh")], check=True, capture_output=True)
    fa = tmp_path / "two.fa"
    fa.write_text(">a\n" + "ACGTTGCAAGGCTTAACCGGTTAGCATCGATCGGATCGATTAGC" * 40 + "\n>b\n" + "ACGTTGCAAGGCTTAACCGGTTAGCATCGATCGGATCGATTAGC" * 40 + "\n")
    p = subprocess.run([dropin, "-s", "loose", "-r", "-o", str(tmp_path / "out"), str(fa)], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "no usable HIP device" in (p.stdout + p.stderr)
    assert not (tmp_path / "out" / "blocks_coords.txt").exists()
