import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sibelia_amd import BlockFinder
from oracle.oracle import Oracle
seqs=[b"ACGTTGCAAGGCTTACGGATCCATGACCTGAATCGTTAGC", b"ACGTTGCAAGGCTAACGGATCCATGACCTGAATCGTTAGC"]
for k in (5,4,9):
    a=BlockFinder(seqs,device=0).enumerate(k); b=Oracle(seqs).enumerate(k)
    print(k, a[0], b[0], len(a[1]), len(b[1]), len(a[2]), len(b[2]))
    if a[0]!=b[0] or len(a[1])!=len(b[1]) or not (a[1]==b[1]).all():
        print(" gpu+", a[1][:12]); print(" orc+", b[1][:12])
