#!/usr/bin/env python3
"""Dependency depth of one iteration of SimplifyGraph under different conflict rules (analysis, CPU only).

The CPU oracle (ORC_TRACE) logs, for every RemoveBulges call of iteration 1 that has bulge groups, the instances it starts from and
the collapses it makes.  A transaction READS the D + k + 2 steps ahead of each of its instances and WRITES the target spans of its
collapses (k + dT + k elements from the target start).  With the transactions in id order, level(t) = 1 + max level of the earlier
transactions t conflicts with; the largest level is the number of ordered rounds an ideal scheduler needs.  Rules compared:
  footprint   what the product reserves today: two transactions conflict when the neighbourhoods of their instances
              ([a - (D+k+2), a + 2(D+k+2) + k]) overlap
  window      their read windows / write spans overlap in any way (read-read included)
  core        the exclusive claims of today without the ordering claims (read-read conflicts on whole cores included)
  core+target the exclusive cores of every instance, but ordering claims only around the instances that a collapse of the transaction
              actually rewrites (what a reservation could claim if the target were known beforehand)
  core+members ... ordering claims around the members of the transaction's bulge groups (known after AnyBulges, before any write; every
              source and target of the call is one of them)
  rw-core     reader / writer claims: windows are read (shared), the whole core of every collapse target is written (exclusive)
  rw          true dependencies only: a write of the earlier one meets a read or write of the later one, or vice versa
usage: python tools/dependency_depth.py [strains] [L0]      (default 8 x 460 kbp)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sibelia_amd import workloads as W
from oracle.oracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L0 = int(sys.argv[2]) if len(sys.argv) > 2 else 460_000
k, D = 25, 150
seqs = W.gen_strains(L0=L0, n=n, seed=1)
E = sum(len(s) for s in seqs) + len(seqs) + 1
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "trace.txt")
    os.environ["ORC_TRACE"] = path
    bulges = Oracle(seqs).simplify_stage(k, D, 4)
    del os.environ["ORC_TRACE"]
    txns, cur, it = [], None, 0
    for line in open(path):
        p = line.split()
        if p[0] == "ITER":
            it = int(p[1])
        elif it != 1:
            continue
        elif p[0] == "T":
            cur = {"id": int(p[1]), "inst": [], "col": [], "mem": set()}
            txns.append(cur)
        elif p[0] == "I":
            cur["inst"].append((int(p[1]), int(p[2])))
        elif p[0] == "M":
            cur["mem"].add(int(p[1]))
        elif p[0] == "C":
            cur["col"].append((int(p[1]), int(p[2]), int(p[3]), int(p[4])))
print("%d strains x %d bp: %d bulges in all, %d transactions with bulge groups in iteration 1 (%d collapses)" %
      (n, L0, bulges, len(txns), sum(len(t["col"]) for t in txns)))
cap = E + 64                                   # fresh slots (insertions) are clipped: they only exist after a collapse of the same region
win, back, fwd = D + k + 2, D + k + 2, 2 * (D + k + 2) + k


def span(a, strand, lo, hi):
    """slots of walk steps lo .. hi-1 from element a on its strand"""
    if a >= E:
        return None
    s, e = (a + lo, a + hi) if strand == 0 else (a - hi + 1, a - lo + 1)
    return max(0, s), min(cap, e)


def depth(rule):
    R = np.zeros(cap, np.int32)                # highest level of a transaction that reads the slot
    Wr = np.zeros(cap, np.int32)               # ... that writes it
    best = 0
    hist = {}
    for t in txns:
        reads = [x for x in (span(a, s, 0, win) for a, s in t["inst"]) if x]
        writes = [x for x in (span(a, s, 0, 2 * k + dT + 1) for a, s, dT, dS in t["col"]) if x]
        if rule == "footprint":
            reads = [x for x in (span(a, s, -back, fwd + 1) for a, s in t["inst"]) if x]
            writes = reads
        if rule == "core":                     # the exclusive claims alone (no ordering claims): every instance's core, read-read conflicts included
            reads = [x for x in (span(a, s, 0, D + 2 * k + 4) for a, s in t["inst"]) if x]
            writes = reads
        if rule == "core+target":              # today's exclusive cores, ordering claims only around the instances a collapse actually rewrites
            reads = [x for x in (span(a, s, 0, D + 2 * k + 4) for a, s in t["inst"]) if x] + [x for x in (span(a, s, -back, fwd + 1) for a, s, dT, dS in t["col"]) if x]
            writes = reads
        if rule == "core+members":             # ... ordering claims around the MEMBERS of the bulge groups: only they can be source or target, whatever happens
            reads = [x for x in (span(a, s, 0, D + 2 * k + 4) for a, s in t["inst"]) if x] + \
                    [x for x in (span(t["inst"][i][0], t["inst"][i][1], -back, fwd + 1) for i in sorted(t["mem"])) if x]
            writes = reads
        if rule == "rw-core":                  # reader / writer claims: every window read (shared), the whole CORE of a collapse's target written (exclusive)
            writes = [x for x in (span(a, s, 0, D + 2 * k + 4) for a, s, dT, dS in t["col"]) if x]
        lvl = 0
        for s, e in reads:
            lvl = max(lvl, int(Wr[s:e].max(initial=0)))
            if rule not in ("rw", "rw-core"):
                lvl = max(lvl, int(R[s:e].max(initial=0)))
        for s, e in writes:
            lvl = max(lvl, int(Wr[s:e].max(initial=0)), int(R[s:e].max(initial=0)))
        lvl += 1
        for s, e in reads:
            np.maximum(R[s:e], lvl, out=R[s:e])
        for s, e in writes:
            np.maximum(Wr[s:e], lvl, out=Wr[s:e])
        best = max(best, lvl)
        hist[lvl] = hist.get(lvl, 0) + 1
    return best, hist


def depth_asym():
    """what a reservation with member-only STAMPS gives: an earlier transaction marks its cores (all instances) and the neighbourhoods of
    its group members; a later one waits for everything marked inside the neighbourhoods of ALL its instances"""
    A = np.zeros(cap, np.int32)
    best, hist = 0, {}
    for t in txns:
        q = [x for x in (span(a, s, -back, fwd + 1) for a, s in t["inst"]) if x]
        lvl = 1 + max([int(A[s:e].max(initial=0)) for s, e in q] + [0])
        mark = [x for x in (span(a, s, 0, D + 2 * k + 4) for a, s in t["inst"]) if x] + \
               [x for x in (span(t["inst"][i][0], t["inst"][i][1], -back, fwd + 1) for i in sorted(t["mem"])) if x]
        for s, e in mark:
            np.maximum(A[s:e], lvl, out=A[s:e])
        best = max(best, lvl)
        hist[lvl] = hist.get(lvl, 0) + 1
    return best, hist


dmax, hist = depth_asym()
print("%-10s depth %4d" % ("member-stamps (asymmetric)", dmax))


def depth_asym_targets():
    """the same with stamps only around the instances a collapse actually rewrites (a plan phase would have to name them)"""
    A = np.zeros(cap, np.int32)
    best = 0
    for t in txns:
        q = [x for x in (span(a, s, -back, fwd + 1) for a, s in t["inst"]) if x]
        lvl = 1 + max([int(A[s:e].max(initial=0)) for s, e in q] + [0])
        mark = [x for x in (span(a, s, 0, D + 2 * k + 4) for a, s in t["inst"]) if x] + [x for x in (span(a, s, -back, fwd + 1) for a, s, dT, dS in t["col"]) if x]
        for s, e in mark:
            np.maximum(A[s:e], lvl, out=A[s:e])
        best = max(best, lvl)
    return best


print("%-10s depth %4d" % ("target-stamps (asymmetric)", depth_asym_targets()))
for rule in ("footprint", "core+members", "core+target", "core", "window", "rw-core", "rw"):
    dmax, hist = depth(rule)
    half = sorted(hist.items())
    acc, tot, p90 = 0, len(txns), 0
    for lvl, c in half:
        acc += c
        if acc >= 0.9 * tot and not p90:
            p90 = lvl
    print("%-10s depth %4d   (90 %% of the transactions are at level <= %d)" % (rule, dmax, p90))
