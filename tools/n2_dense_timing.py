#!/usr/bin/env python3
"""GenerateSyntenyBlocks at a small k on the raw graph (what -v / --allstages do before the first stage): python tools/n2_dense_timing.py [k] [input]"""
import gzip
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sibelia_amd import BlockFinder      # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 15
inp = sys.argv[2] if len(sys.argv) > 2 else "Helicobacter_pylori"
with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "in.fa")
    open(fa, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "data", inp + ".fa.gz")).read())
    bf = BlockFinder.from_fasta(fa, device=0)
t0 = time.time()
b = bf.generate_blocks(k, k, k)
print("GenerateSyntenyBlocks(%d, %d, %d): %.2f s, %d block instances" % (k, k, k, time.time() - t0, len(b)), flush=True)
t0 = time.time()
out, texts = bf.postprocess(None, True)
print("PostProcess (GlueStripes + report texts): %.2f s, %d block instances left" % (time.time() - t0, len(out)), flush=True)
