"""CPU-side checks of the drop-in boundary: the library builds, loads and exports every symbol that
include/sibelia_amd.h declares; without a GPU it fails loudly instead of computing on the host."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from sibelia_amd.build import LIB
    lib = ctypes.CDLL(LIB)
    hdr = open(os.path.join(ROOT, "include", "sibelia_amd.h")).read()
    declared = set(re.findall(r"\b(sbl_[a-z_]+)\s*\(", hdr))
    declared -= {"sbl_progress_fn"}
    assert len(declared) >= 12
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export " + name


def test_no_host_compute_path_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from sibelia_amd import BlockFinder, SibeliaError
    with pytest.raises(SibeliaError, match="no usable HIP device"):
        BlockFinder([b"ACGTACGT"])


def test_product_never_references_the_oracle():
    # the oracle is test infrastructure: nothing under sibelia_amd/ or include/ may import, link or load it
    for base in ("sibelia_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "sibelia_oracle" not in txt and "oracle.oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


def test_cpp_class_surface_compiles_and_fails_loudly_without_a_gpu(tmp_path):
    # include/sibelia_amd/blockfinder.hpp (the reference's BlockFinder surface over the C ABI) with plain g++
    import subprocess
    import torch
    import __graft_entry__ as g
    g.build()
    from sibelia_amd.build import LIBDIR
    src = tmp_path / "main.cpp"
    src.write_text('#include "sibelia_amd/blockfinder.hpp"\n'
                   'struct Rec { std::string s; const std::string &GetSequence() const { return s; } };\n'
                   'int main() { std::vector<Rec> v(1); v[0].s = "ACGTACGTTGCA";\n'
                   '  try { SyntenyFinderAMD::BlockFinder bf(v); std::printf("bulges %zu\\n", bf.PerformGraphSimplifications(3, 6, 2)); }\n'
                   '  catch (const std::exception &e) { std::printf("error: %s\\n", e.what()); return 3; } return 0; }\n')
    exe = tmp_path / "main"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", LIBDIR, "-lsibelia_amd",
                    "-Wl,-rpath," + LIBDIR, "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0 and r.stdout.startswith("bulges ")
    else:
        assert r.returncode == 3 and "no usable HIP device" in r.stdout


def test_glue_stripes_worklist_equals_the_oracle_on_random_block_lists():
    """sbl_glue_stripes is host bookkeeping (row N4): no device needed.  Random permutations built from runs of blocks that always
    travel together -- what GlueStripes exists to merge -- against the oracle's restatement of the reference's rescan-per-merge loop."""
    import numpy as np
    from oracle.oracle import Oracle
    from sibelia_amd import formats as F
    from sibelia_amd.api import glue_stripes
    rng = np.random.default_rng(5)
    shrunk = 0
    for case in range(40):
        nchr = int(rng.integers(1, 5))
        nsyn = int(rng.integers(1, 30))                       # "true" blocks, each cut into 1..4 stripes that stay adjacent
        stripes, nxt = [], 1
        for _ in range(nsyn):
            k = int(rng.integers(1, 5))
            stripes.append(list(range(nxt, nxt + k)))
            nxt += k
        rows = []
        for c in range(nchr):
            pos = 0
            for s in rng.permutation(nsyn)[: int(rng.integers(1, nsyn + 1))]:
                ids = stripes[int(s)]
                rev = bool(rng.integers(0, 2))
                if rng.random() < 0.15 and len(ids) > 1:      # an occurrence that lacks a stripe: those two must not glue
                    ids = ids[:-1]
                for i in (reversed(ids) if rev else ids):
                    ln = int(rng.integers(5, 50))
                    rows.append((-i if rev else i, c, pos, pos + ln))
                    pos += ln + int(rng.integers(0, 20))
        blocks = np.array(rows, dtype=F.BLOCK_DTYPE)
        seqs = [b"A" * 10_000] * nchr
        want, _ = Oracle(seqs).postprocess(blocks, ["s%d" % i for i in range(nchr)], True)
        got = glue_stripes(blocks, nchr)
        assert len(got) == len(want), case
        for f in ("id", "chr", "start", "end"):
            assert (got[f] == want[f]).all(), (case, f)
        shrunk += len(got) < len(blocks)
    assert shrunk >= 20
