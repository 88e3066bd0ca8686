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
