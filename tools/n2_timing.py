#!/usr/bin/env python3
"""Times the synteny stage (N2, sbl_generate_blocks) after the reference's parameter cascades: python tools/n2_timing.py"""
import gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sibelia_amd import BlockFinder, workloads as W
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names, hp = W.read_fasta(os.path.join(root, "tests", "golden", "data", "Helicobacter_pylori.fa.gz"))
cases = [("H. pylori -s loose", hp, [(30, 150), (100, 1000), (1000, 5000), (5000, 15000)], (5000, 30, 5000)),
         ("H. pylori -s fine", hp, [(30, 150), (100, 500), (500, 1500)], (500, 30, 500)),
         ("8 strains x 4.6 Mbp -s fine", W.gen_strains(L0=4_600_000, n=8, seed=1), [(30, 150), (100, 500), (500, 1500)], (500, 30, 500))]
for name, seqs, stages, (k, tk, ms) in cases:
    bf = BlockFinder(seqs, device=0)
    t = time.time()
    for kk, d in stages:
        bf.PerformGraphSimplifications(kk, d, 4)
    t_st = time.time() - t
    t = time.time()
    b = bf.GenerateSyntenyBlocks(k, tk, ms)
    t_b = time.time() - t
    t = time.time()
    b2 = bf.GenerateSyntenyBlocks(k, tk, ms)
    print("%-30s stages %.2f s   GenerateSyntenyBlocks(%d,%d,%d) %.2f s (again %.2f s)  %d block instances, %d blocks" % (name, t_st, k, tk, ms, t_b, time.time() - t, len(b), len(set(abs(int(x)) for x in b["id"]))), flush=True)
    bf.close()
