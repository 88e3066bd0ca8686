"""The sharded k-mer table's layout and its one exchange, without a GPU (csrc/shard.hip).

run_enumeration_sharded takes its tile ranges, owner ranges and all-to-all byte tables from two device-free entry points of the
library (sbl_shard_layout, sbl_shard_exchange_plan).  Here they are checked for 1 .. 8 ranks against the defining formulas, for
consistency ACROSS ranks (what p sends q is what q expects from p), and then used for real: world-size-2 and -4 process groups over
gloo partition deterministic key sets by hash prefix, exchange them with point-to-point messages laid out by the plan -- the shape of
shard.hip's grouped ncclSend / ncclRecv -- and the owners end up with exactly the keys of their buckets."""
import os

import numpy as np
import pytest

from sibelia_amd import api as A


@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("bits", [4, 10, 17])
def test_owner_ranges_and_tile_ranges_partition(R, bits):
    nb = 1 << bits
    ntiles = 12345 + 7 * R
    tiles = []
    for r in range(R):
        fb, tr = A.shard_layout(R, r, bits, ntiles)
        assert fb[0] == 0 and fb[R] == nb and np.all(np.diff(fb.astype(np.int64)) >= 0)
        # owner(b) = (b * R) >> bits is the range that holds b: at every range boundary and at random buckets
        probe = np.unique(np.clip(np.concatenate([fb.astype(np.int64), fb.astype(np.int64) - 1, np.random.default_rng(R * 31 + bits).integers(0, nb, 200)]), 0, nb - 1))
        owner = (probe * R) >> bits
        assert np.all(fb[owner] <= probe) and np.all(probe < fb[owner + 1])
        tiles.append(tr)
    assert tiles[0][0] == 0 and tiles[-1][1] == ntiles and all(tiles[i][1] == tiles[i + 1][0] for i in range(R - 1))
    sizes = [b - a for a, b in tiles]
    assert max(sizes) - min(sizes) <= 1                                  # balanced by construction


@pytest.mark.parametrize("R", [2, 4, 8])
def test_exchange_plans_agree_across_ranks(R):
    rng = np.random.default_rng(100 + R)
    count = rng.integers(0, 5000, (R, R)).astype(np.uint64)
    count[rng.integers(0, R), rng.integers(0, R)] = 0                    # an empty message somewhere
    plans = []
    for r in range(R):
        send_at = np.concatenate([[0], np.cumsum(count[r])]).astype(np.uint32)
        plans.append(A.shard_exchange_plan(R, r, count, send_at, 8))
    for p in range(R):
        sb, so, rb, ro, nrecv = plans[p]
        assert nrecv == int(count[:, p].sum())
        assert np.array_equal(so, 8 * np.concatenate([[0], np.cumsum(count[p])[:-1]]))
        assert np.array_equal(ro, 8 * np.concatenate([[0], np.cumsum(count[:, p])[:-1]]))
        for q in range(R):
            assert sb[q] == plans[q][2][p] == 8 * count[p, q]               # what p sends q is what q expects from p
    with pytest.raises(A.SibeliaError):                                   # a gathered row that contradicts the rank's own partition is refused
        A.shard_exchange_plan(R, 0, count, np.zeros(R + 1, dtype=np.uint32), 8)


def _mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd); x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53); x ^= x >> np.uint64(33)
    return x


def _worker(rank, world, port, q, bits, ntiles, per_tile):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fb, (t0, t1) = A.shard_layout(world, rank, bits, ntiles)
    # phase A: the records of MY tiles (deterministic keys), partitioned by the low `bits` bits -- stable, as the radix sort is
    pos = np.arange(t0 * per_tile, t1 * per_tile, dtype=np.uint64)
    keys = _mix64(pos)
    bucket = (keys & np.uint64((1 << bits) - 1)).astype(np.int64)
    order = np.argsort(bucket, kind="stable")
    keys, bucket = keys[order], bucket[order]
    send_at = np.searchsorted(bucket, fb.astype(np.int64)).astype(np.uint32)       # boff[first bucket of every owner]
    # phase B: counts all-gathered, plan from the library, one message per peer
    mine = torch.from_numpy(np.diff(send_at.astype(np.int64)).astype(np.int64))
    rows = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(rows, mine)
    count = np.stack([r.numpy() for r in rows]).astype(np.uint64)
    sb, so, rb, ro, nrecv = A.shard_exchange_plan(world, rank, count, send_at, 8)
    send = torch.from_numpy(keys.view(np.uint8).copy())
    recv = torch.zeros(nrecv * 8, dtype=torch.uint8)
    reqs = []
    for i in range(world):                                                          # same peer order as RcclComm::alltoallv
        p, s = (rank + i) % world, (rank + world - i) % world
        if p == rank:
            recv[int(ro[p]):int(ro[p] + rb[p])] = send[int(so[p]):int(so[p] + sb[p])]
            continue
        if sb[p]:
            reqs.append(dist.isend(send[int(so[p]):int(so[p] + sb[p])].clone(), dst=p))
        if rb[s]:
            reqs.append(dist.irecv(recv[int(ro[s]):int(ro[s] + rb[s])], src=s))
    for r in reqs:
        r.wait()
    got = recv.numpy().view(np.uint64)
    gb = (got & np.uint64((1 << bits) - 1)).astype(np.int64)
    ok_owner = bool(np.all((gb >= fb[rank]) & (gb < fb[rank + 1])))
    # the segment received from p is exactly p's records for my buckets, in p's (bucket-sorted) order
    ok_seg = True
    for p in range(world):
        _, (a, b) = A.shard_layout(world, p, bits, ntiles)
        kp = _mix64(np.arange(a * per_tile, b * per_tile, dtype=np.uint64))
        bp = (kp & np.uint64((1 << bits) - 1)).astype(np.int64)
        kp = kp[np.argsort(bp, kind="stable")]
        bp = np.sort(bp, kind="stable")
        want = kp[(bp >= fb[rank]) & (bp < fb[rank + 1])]
        ok_seg = ok_seg and np.array_equal(got[int(ro[p]) // 8:int(ro[p] + rb[p]) // 8], want)
    q.put((rank, ok_owner, ok_seg, int(nrecv), int(np.bitwise_xor.reduce(got)) if len(got) else 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_hash_prefix_exchange_between_real_processes(world):
    import torch.multiprocessing as mp
    bits, ntiles, per_tile = 9, 37, 512
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 200 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, bits, ntiles, per_tile)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), res                            # every owner holds exactly its buckets, segment by segment
    assert sum(r[3] for r in res) == ntiles * per_tile                      # every record went to exactly one owner
    allk = _mix64(np.arange(ntiles * per_tile, dtype=np.uint64))
    x = 0
    for r in res:
        x ^= r[4]
    assert x == int(np.bitwise_xor.reduce(allk))                            # ... and it is the same multiset of keys
