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
