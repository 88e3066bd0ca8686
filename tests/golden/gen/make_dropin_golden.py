#!/usr/bin/env python3
"""Fixtures for tests/test_gpu_dropin.py: the UNMODIFIED reference program (oracle/_ref/sibelia_ref, built by
oracle/build_dropin.sh from /root/reference) is run on the committed example inputs; the sha256 of every file it writes and of
its standard output go to tests/golden/dropin_cases.json.  Run in the build container (needs /root/reference): CPU only."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
REF = os.path.join(ROOT, "oracle", "_ref", "sibelia_ref")

# name, input (tests/golden/data/<input>.fa.gz, or "synth:L0:n:seed" = sibelia_amd.workloads.gen_strains written as FASTA),
# arguments between the program name and "-o out <input>.fa"
CASES = [
    ("hpylori_loose_inram_sequences", "Helicobacter_pylori", ["-s", "loose", "-r", "-q"]),
    ("hpylori_fine_allstages_graphs", "Helicobacter_pylori", ["-s", "fine", "--allstages", "-g", "-r", "-m", "2000"]),
    ("hpylori_far_hierarchy", "Helicobacter_pylori", ["-s", "far", "-v", "-r"]),
    ("saureus_loose_gff_sharedonly_tempfiles", "Staphylococcus_aureus_pair", ["-s", "loose", "--gff", "-a"]),
    # BASELINE.json config 3 (8 genomes x 4.6 Mbp, -s fine) through the reference's own main: ~12 min for the reference (--big)
    ("synth8_4600k_fine_config3", "synth:4600000:8:1", ["-s", "fine", "-r"]),
    # several ambiguous bases per record: WITHOUT -r the reference names its temporary files from the same rand() stream that replaces
    # them (src/platform.cpp:57), so the two modes produce different sequences from the second index on -- both must be reproduced
    ("ambig_fine_tempfiles_allstages", "ambig:60000:4:77:9", ["-s", "fine", "--allstages", "-g", "-m", "500"]),
    ("ambig_fine_inram_allstages", "ambig:60000:4:77:9", ["-s", "fine", "--allstages", "-g", "-m", "500", "-r"]),
    ("saureus_fine_nopostprocess_lastk", "Staphylococcus_aureus_pair", ["-s", "fine", "-r", "--nopostprocess", "--lastk", "200", "-m", "1000", "-i", "2"]),
]


def run_case(program, inp, args, workdir, env=None):
    """-> (returncode, sha256 of stdout, {relative path: [size, sha256]})"""
    if inp.startswith(("synth:", "ambig:")):
        sys.path.insert(0, ROOT)
        from sibelia_amd import workloads as W
        kind, L0, n, seed = inp.split(":")[:4]
        strains = W.gen_strains(L0=int(L0), n=int(n), seed=int(seed), inv_min=max(50, int(L0) // 100), inv_max=max(200, int(L0) // 20)) if kind == "ambig" \
            else W.gen_strains(L0=int(L0), n=int(n), seed=int(seed))
        if kind == "ambig":                                   # `count` ambiguity codes per strain at deterministic places
            import numpy as np
            rng = np.random.default_rng(int(seed) + 1)
            out_s = []
            for s in strains:
                g = bytearray(s)
                for pos in rng.choice(len(g), int(inp.split(":")[4]), replace=False):
                    g[int(pos)] = b"NRYKMSWBDHX"[int(rng.integers(0, 11))]
                out_s.append(bytes(g))
            strains = out_s
        fa = "synth.fa"
        with open(os.path.join(workdir, fa), "wb") as g:
            for i, s in enumerate(strains):
                b = s if isinstance(s, bytes) else s.encode()
                g.write(b">strain%d synthetic\n" % i)
                for o in range(0, len(b), 80):
                    g.write(b[o:o + 80] + b"\n")
    else:
        fa = inp + ".fa"
        with gzip.open(os.path.join(DATA, inp + ".fa.gz"), "rb") as f, open(os.path.join(workdir, fa), "wb") as g:
            shutil.copyfileobj(f, g)
    p = subprocess.run([program] + args + ["-o", "out", fa], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=3600)
    files = {}
    for d, _, names in os.walk(os.path.join(workdir, "out")):
        for n in names:
            path = os.path.join(d, n)
            b = open(path, "rb").read()
            files[os.path.relpath(path, os.path.join(workdir, "out"))] = [len(b), hashlib.sha256(b).hexdigest()]
    return p.returncode, hashlib.sha256(p.stdout).hexdigest(), files, p.stdout, p.stderr


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("build oracle/_ref/sibelia_ref first: bash oracle/build_dropin.sh")
    out = {"generator": "tests/golden/gen/make_dropin_golden.py", "program": "oracle/_ref/sibelia_ref (unmodified reference, oracle/build_dropin.sh)", "cases": []}
    path = os.path.join(ROOT, "tests", "golden", "dropin_cases.json")
    old = {c["name"]: c for c in json.load(open(path))["cases"]} if os.path.exists(path) else {}
    for name, inp, args in CASES:
        if inp.startswith("synth:") and "--big" not in sys.argv:      # keep what an earlier --big run wrote
            if name in old:
                out["cases"].append(old[name])
            continue
        if "--only-big" in sys.argv and not inp.startswith("synth:") and name in old:
            out["cases"].append(old[name])
            continue
        with tempfile.TemporaryDirectory() as wd:
            rc, so, files, stdout, stderr = run_case(REF, inp, args, wd)
        print(name, "rc", rc, len(files), "files", file=sys.stderr)
        out["cases"].append({"name": name, "input": inp, "args": args, "returncode": rc, "stdout_sha256": so, "stdout_bytes": len(stdout), "files": files})
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
