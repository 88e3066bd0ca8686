import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from sibelia_amd import BlockFinder, workloads as W
total = int(sys.argv[1]); nrec = int(sys.argv[2]); k = int(sys.argv[3]); D = int(sys.argv[4])
seqs = W.random_dna(total, nrec, seed=5)
# plant an exact duplicate of a 30 kbp segment of record 0 into record 1 so that there is something to find
seqs[1] = seqs[1][:1000] + seqs[0][5000:35000] + seqs[1][31000:]
t = time.time(); bf = BlockFinder(seqs, device=0); print('load %.1fs' % (time.time() - t), flush=True)
bf.save_state()
t = time.time(); b = bf.PerformGraphSimplifications(k, D, 4); dt0 = time.time() - t
print('first call (allocates the workspaces) %.2fs' % dt0, flush=True)
bf.restore_state()
t = time.time(); b = bf.PerformGraphSimplifications(k, D, 4); dt = time.time() - t
st = bf.stats()
print('k=%d D=%d total=%d bulges=%d ids=%d inst=%d time=%.2fs  %.1f M k-mers/s' % (k, D, total, b, st['bif_count'], st['instances'], dt, st['strand_kmers'] / dt / 1e6), st)
