"""The sharded rank doubling of the long-k enumeration (csrc/longk.hip, BASELINE.json config 5), without a GPU.

run_enumeration_longk_sharded takes its position slices, value / sorted-order bounds, owners and halo byte tables from four
device-free entry points of the library (sbl_longk_slices, sbl_longk_value_bounds, sbl_longk_owner, sbl_longk_halo_plan).  Here they
are checked against their defining formulas and for consistency ACROSS ranks, and then used for real: world-size-2 and -3 process
groups over gloo run the whole distributed algorithm -- rank8, doubling rounds over the active suffixes with the two routed exchanges,
halo fetches laid out by the plan, candidate windows, ids by rank order -- on numpy arrays, with point-to-point messages in the shape
of shard.hip's grouped ncclSend / ncclRecv, and every rank must end up with the marks the CPU oracle enumerates."""
import os

import numpy as np
import pytest

from sibelia_amd import api as A


@pytest.mark.parametrize("R", [1, 2, 3, 5, 8])
def test_slices_bounds_and_owners(R):
    for np_ in (7, 100, 12345, (1 << 31) - 17):
        P = A.longk_slices(R, np_)
        assert P[0] == 0 and P[R] == np_ and np.all(np.diff(P.astype(np.int64)) >= 0)
        assert max(np.diff(P.astype(np.int64))) - min(np.diff(P.astype(np.int64))) <= 1
        rng = np.random.default_rng(R + np_ % 1000)
        for x in np.unique(np.concatenate([P[:-1], np.maximum(P[1:].astype(np.int64) - 1, 0).astype(np.uint64), rng.integers(0, np_, 20).astype(np.uint64)])):
            o = A.longk_owner(P, int(x))
            assert P[o] <= x and (x < P[o + 1] or all(P[j] == P[o + 1] for j in range(o + 1, R + 1)))
    V = A.longk_value_bounds(R, 390624)
    assert V[0] == 0 and V[R] == 390625 and np.all(np.diff(V.astype(np.int64)) > 0 if R <= 390625 else True)
    assert [A.longk_owner(V, int(v)) for v in (0, 390624)] == [0, R - 1]


@pytest.mark.parametrize("R", [2, 3, 5, 8])
@pytest.mark.parametrize("H", [0, 1, 8, 64, 5000])
def test_halo_plans_agree_across_ranks(R, H):
    np_ = 1000 + 13 * R
    P = A.longk_slices(R, np_).astype(np.int64)
    plans = [A.longk_halo_plan(R, r, np_, H) for r in range(R)]
    for r in range(R):
        sb, so, rb, ro = plans[r]
        need = min(np_, P[r + 1] + H) - P[r + 1]
        assert int(rb.sum()) == 4 * need                                    # the halo is covered exactly once ...
        covered = np.zeros(need, dtype=np.int32)
        for p in range(R):
            assert rb[p] == plans[p][0][r]                                  # ... what p sends r is what r expects from p
            if rb[p]:
                a = int(ro[p]) // 4
                covered[a:a + int(rb[p]) // 4] += 1
                assert P[p] + int(plans[p][1][r]) // 4 == P[r + 1] + a      # ... and it is the same positions on both sides
        assert np.all(covered == 1)


# ------------------------------------------------------------------------------------------- the algorithm over a process group
def _super(seqs, k):
    """S = #c0#c1..#rc(c0)#rc(c1)..# as symbols 0 (separator) .. 4, padded with k zeros (longk.hip: k_lk_super)"""
    code = {65: 1, 67: 2, 71: 3, 84: 4}
    fw = [0]
    for s in seqs:
        fw += [code.get(c, 0) for c in s] + [0]
    rv = []
    for s in seqs:
        rv += [5 - code[c] if c in code else 0 for c in reversed(s)] + [0]
    return np.array(fw + rv + [0] * k, dtype=np.int64), len(fw)


class _Comm:
    def __init__(self, rank, world):
        self.rank, self.n = rank, world

    def allgather(self, x):
        import torch
        import torch.distributed as dist
        rows = [torch.zeros(1, dtype=torch.int64) for _ in range(self.n)]
        dist.all_gather(rows, torch.tensor([int(x)], dtype=torch.int64))
        return [int(r.item()) for r in rows]

    def alltoallv(self, send, sb, so, recv, rb, ro):
        """bytes; the peer order and the one-message-per-peer shape of RcclComm::alltoallv"""
        import torch
        import torch.distributed as dist
        s, r = torch.from_numpy(send.view(np.uint8)), torch.from_numpy(recv.view(np.uint8))
        reqs = []
        for i in range(self.n):
            p, q = (self.rank + i) % self.n, (self.rank + self.n - i) % self.n
            if p == self.rank:
                r[int(ro[p]):int(ro[p] + rb[p])] = s[int(so[p]):int(so[p] + sb[p])]
                continue
            if sb[p]:
                reqs.append(dist.isend(s[int(so[p]):int(so[p] + sb[p])].clone(), dst=p))
            if rb[q]:
                reqs.append(dist.irecv(r[int(ro[q]):int(ro[q] + rb[q])], src=q))
        for x in reqs:
            x.wait()

    def route(self, a, b, dest):
        """records (a[i], b[i]) to rank dest[i]: counts all-gathered, plan from the library (sbl_shard_exchange_plan)"""
        import torch
        import torch.distributed as dist
        order = np.argsort(dest, kind="stable")
        a, b, dest = a[order], b[order], dest[order]
        send_at = np.searchsorted(dest, np.arange(self.n + 1)).astype(np.uint32)
        rows = [torch.zeros(self.n, dtype=torch.int64) for _ in range(self.n)]
        dist.all_gather(rows, torch.from_numpy(np.diff(send_at.astype(np.int64))))
        count = np.stack([r.numpy() for r in rows]).astype(np.uint64)
        out = []
        for arr in (a, b):
            w = arr.dtype.itemsize
            sb, so, rb, ro, nrecv = A.shard_exchange_plan(self.n, self.rank, count, send_at, w)
            recv = np.zeros(nrecv, dtype=arr.dtype)
            self.alltoallv(np.ascontiguousarray(arr), sb, so, recv, rb, ro)
            out.append(recv)
        return out


def _owners(bounds, x):
    return (np.searchsorted(bounds.astype(np.int64), x, side="right") - 1).clip(0, len(bounds) - 2)


def _sharded_longk(comm, seqs, k):
    """longk.hip's run_enumeration_longk_sharded on numpy arrays; returns the (strand, element, id) marks of ALL ranks and the id count"""
    R, r = comm.n, comm.rank
    S, E = _super(seqs, k)
    n = 2 * E - 1
    np_ = n + k
    assert len(S) == np_
    P = A.longk_slices(R, np_)
    lo, hi = int(P[r]), int(P[r + 1])
    pad = np.concatenate([S, np.zeros(8, dtype=np.int64)])
    rk = np.zeros(hi - lo, dtype=np.int64)
    for t in range(8):
        rk = rk * 5 + pad[lo + t:hi + t]
    act = np.arange(lo, hi, dtype=np.int64)
    aflag = np.zeros(hi - lo, dtype=np.int64)
    V = A.longk_value_bounds(R, 390624)
    G = None
    h, first = 8, True
    rbp = int(np_).bit_length()

    def halo(H):
        sb, so, rb, ro = A.longk_halo_plan(R, r, np_, H)
        out = np.zeros(H + 1, dtype=np.uint32)
        comm.alltoallv(rk.astype(np.uint32), sb, so, out, rb, ro)
        return out.astype(np.int64)

    def second(i, off, hl):
        t = i + off
        res = np.zeros(len(i), dtype=np.int64)
        inside = t < hi
        res[inside] = rk[t[inside] - lo]
        beyond = (~inside) & (t < np_)
        res[beyond] = hl[t[beyond] - hi]
        return res

    na_all = np_
    while 2 * h <= k and na_all:
        rb = 19 if first else rbp
        hl = halo(h)
        keys = (rk[act - lo] << rb) | second(act, h, hl)
        dest = _owners(V if first else G, keys >> rb)
        for x, d in zip((keys >> rb)[:50], dest[:50]):                      # the library's owner function agrees with the vectorised one
            assert A.longk_owner(V if first else G, int(x)) == d
        rkeys, ridx = comm.route(keys.astype(np.uint64), act.astype(np.uint32), dest)
        m = len(rkeys)
        if first:
            G = np.concatenate([[0], np.cumsum(comm.allgather(m))]).astype(np.uint64)
        order = np.argsort(rkeys, kind="stable")
        sk, si = rkeys[order].astype(np.int64), ridx[order].astype(np.int64)
        j = np.arange(m)
        ghead = np.ones(m, dtype=bool); shead = np.ones(m, dtype=bool)
        if m:
            ghead[1:] = (sk[1:] >> rb) != (sk[:-1] >> rb)
            shead[1:] = sk[1:] != sk[:-1]
        gstart = np.maximum.accumulate(np.where(ghead, j, 0))
        sstart = np.maximum.accumulate(np.where(shead, j, 0))
        nr = int(G[r]) + sstart if first else (sk >> rb) + (sstart - gstart)
        nxt = np.concatenate([sstart[1:], [m]])
        single = (sstart == j) & (nxt == j + 1)
        val = nr | (np.where(single, 0, 1) << 31)
        bi, bv = comm.route(si.astype(np.uint32), val.astype(np.uint32), _owners(P, si))
        assert len(bi) == len(act)
        rk[bi.astype(np.int64) - lo] = bv.astype(np.int64) & 0x7FFFFFFF
        aflag[bi.astype(np.int64) - lo] = bv.astype(np.int64) >> 31
        act = lo + np.nonzero(aflag)[0]
        na_all = sum(comm.allgather(len(act)))
        first = False
        h *= 2
    while 2 * h <= k:
        h *= 2
    # candidate windows
    sep = np.nonzero(S[:E] == 0)[0]                                         # sepidx
    nchr = len(sep) - 1
    hl = halo(k - h)

    def valid_window(i):
        ok = np.zeros(len(i), dtype=bool)
        for x, p in enumerate(i):
            if p < E:
                c = np.searchsorted(sep, p, side="left") - 1
                ok[x] = c >= 0 and p > sep[c] and p + k <= sep[c + 1]
            elif p < n:
                q = p - E
                c = np.searchsorted(sep, q + 1, side="left") - 1
                ok[x] = (q - sep[c]) + k <= sep[c + 1] - sep[c] - 1
        return ok
    cand = [act[valid_window(act)]]
    for c in range(nchr):
        ln = sep[c + 1] - sep[c] - 1
        if ln < k:
            continue
        for which in range(4):
            if (which & 1) and ln == k:
                continue
            off = ln - k if which & 1 else 0
            i = E + sep[c] + off if which & 2 else sep[c] + 1 + off
            if lo <= i < hi and not aflag[i - lo]:
                cand.append(np.array([i], dtype=np.int64))
    cand = np.concatenate(cand) if cand else np.zeros(0, dtype=np.int64)
    keys = (rk[cand - lo] << rbp) | second(cand, k - h, hl)
    rkeys, ridx = comm.route(keys.astype(np.uint64), cand.astype(np.uint32), _owners(G, keys >> rbp))
    order = np.argsort(rkeys, kind="stable")
    sk, si = rkeys[order].astype(np.int64), ridx[order].astype(np.int64)
    marks = []
    nb = 0
    if len(sk):
        head = np.ones(len(sk), dtype=bool)
        head[1:] = sk[1:] != sk[:-1]
        gno = np.cumsum(head) - 1
        prevm = np.zeros(gno[-1] + 1, dtype=np.int64); nextm = np.zeros(gno[-1] + 1, dtype=np.int64)
        np.bitwise_or.at(prevm, gno, 1 << S[si - 1])
        np.bitwise_or.at(nextm, gno, 1 << S[si + k])
        bif = np.array([(p & 1) or (q & 1) or bin(p >> 1).count("1") > 1 or bin(q >> 1).count("1") > 1 for p, q in zip(prevm, nextm)], dtype=bool)
        gid = np.cumsum(bif) - bif
        nb = int(bif.sum())
        allb = comm.allgather(nb)
        off = sum(allb[:r])
        for j in np.nonzero(bif[gno])[0]:
            i = int(si[j])
            if i < E:
                marks.append((0, i, off + int(gid[gno[j]])))
            else:
                q = i - E
                c = np.searchsorted(sep, q + 1, side="left") - 1
                marks.append((1, int(sep[c + 1] - 1 - (q - sep[c])), off + int(gid[gno[j]])))
    else:
        allb = comm.allgather(0)
    return marks, sum(allb)


def _worker(rank, world, port, q, seed, k):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        marks, nb = _sharded_longk(_Comm(rank, world), _case(seed), k)
        q.put((rank, marks, nb, None))
    except BaseException as e:      # noqa: BLE001
        import traceback
        q.put((rank, [], 0, traceback.format_exc() + repr(e)))
    dist.barrier()
    dist.destroy_process_group()


def _case(seed):
    from sibelia_amd import workloads as W
    rng = np.random.default_rng(seed)
    seqs = W.gen_strains(L0=int(rng.integers(400, 900)), n=3, seed=seed, snp=0.02, indel_every=150, inv_min=40, inv_max=120)
    seqs.append(seqs[0][:70])                                                # a record of k .. 2k characters
    seqs.append(b"ACGTTGCA" * 4 + b"ACGTTGCAC")                              # ... and one of exactly k = 41
    return seqs


@pytest.mark.parametrize("world,seed,k", [(2, 11, 41), (3, 12, 64), (2, 13, 100)])
def test_sharded_rank_doubling_between_real_processes(world, seed, k):
    import torch.multiprocessing as mp
    from oracle.oracle import Oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + world + seed
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, seed, k)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(x[3] is None for x in res), [x[3] for x in res if x[3]]
    seqs = _case(seed)
    nb, pos, neg = Oracle(seqs).enumerate(k)
    assert all(x[2] == nb for x in res)                                      # every rank knows the total number of ids
    sep = np.concatenate([[0], np.cumsum([len(s) + 1 for s in seqs])])
    want = set()
    for strand, inst in ((0, pos), (1, neg)):
        for it in inst:
            c, p = int(it["chr"]), int(it["pos"])
            e = sep[c] + 1 + p if strand == 0 else sep[c] + 1 + (len(seqs[c]) - 1 - p)      # - strand: the element its k-mer STARTS at
            want.add((strand, int(e), int(it["id"])))
    got = set()
    for x in res:
        assert not (got & set(x[1]))                                         # a mark is produced by exactly one rank
        got |= set(x[1])
    assert got == want
