"""Deterministic synthetic genome workloads + minimal FASTA I/O.

Real E. coli genomes are not available offline, so BASELINE.json's E. coli configs
are exercised with `gen_strains` (uniform random ACGT
ancestor, 1 % SNPs, one indel per 2000 bp of length U[1,20], three inversions of
50-200 kbp per strain; numpy `default_rng(seed)`, draws in a fixed order so that the
n=2 set is a prefix of the n=8 set).  `small_case` produces the tiny randomised
genomes used for golden-vector parity tests (tests/golden/).

FASTA handling follows the reference reader's observable behaviour
(reference src/fasta.cpp:23-104): lines are trimmed, sequence letters upper-cased,
only `ACGTURYKMSWBDHWNX-` accepted, header = text up to the first blank.
"""
from __future__ import annotations

import gzip
import hashlib
from typing import List, Sequence, Tuple

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)
VALID_CHARS = b"ACGTURYKMSWBDHWNX-"


def gen_strains(L0: int = 4_600_000, n: int = 8, seed: int = 1, snp: float = 0.01,
                indel_every: int = 2000, inversions: int = 3,
                inv_min: int = 50_000, inv_max: int = 200_000) -> List[bytes]:
    """n mutated copies of one random ancestor (upper-case ASCII ACGT, one record each)."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, L0, dtype=np.uint8)
    out = []
    for _ in range(n):
        g = anc.copy()
        m = rng.random(L0) < snp
        g[m] = (g[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) % 4
        pieces = []
        p = 0
        for q in np.sort(rng.choice(L0, L0 // indel_every, replace=False)):
            q = int(q)
            pieces.append(g[p:q])
            if rng.random() < 0.5:
                pieces.append(rng.integers(0, 4, int(rng.integers(1, 21)), dtype=np.uint8))
                p = q
            else:
                p = min(L0, q + int(rng.integers(1, 21)))
        pieces.append(g[p:])
        g = np.concatenate(pieces)
        imax = inv_max if inv_max < len(g) // 2 else max(2, len(g) // 8)      # short genomes: scaled-down inversions
        imin = inv_min if inv_min < imax else max(1, imax // 4)
        for _ in range(inversions):
            a = int(rng.integers(0, len(g) - imax))
            b = a + int(rng.integers(imin, imax))
            g[a:b] = _COMP[g[a:b][::-1]]
        out.append(_ACGT[g].tobytes())
    return out


def cascade_case(seed: int) -> Tuple[List[bytes], List[Tuple[int, int]]]:
    """A small multi-strain input plus a 3- or 4-stage (k, D) cascade with growing k (the shape of the reference's parameter sets,
    src/util.cpp:52-87): state -- sequences AND original positions -- is carried across copy-backs, so later stages see what
    the earlier ones left (positions re-interpolated by collapses, closing separators stamped with the CURRENT record length,
    dnasequence.cpp:96).  Every strain additionally gets a substitution and a short indel within reach of each record end, so
    that collapses happen right at chromosome boundaries."""
    rng = np.random.default_rng(7_000_003 * (seed + 1))
    n = int(rng.integers(2, 9))
    L0 = int(rng.integers(3_000, 24_000))
    k = int(rng.choice([15, 16, 20, 25, 30, 31, 32]))
    D = int(rng.integers(3 * k, 8 * k))
    snp = float(rng.choice([0.005, 0.01, 0.03, 0.06]))
    seqs = gen_strains(L0=L0, n=n, seed=500_000 + seed, snp=snp, indel_every=int(rng.choice([150, 400, 1000, 2000])),
                       inv_min=max(50, L0 // 100), inv_max=max(200, L0 // 20))
    out = []
    for s in seqs:
        g = bytearray(s)
        for end in (0, 1):
            reach = k + int(rng.integers(1, D))              # distance of the edit from the record end
            if reach + 25 >= len(g):
                continue
            at = reach if end == 0 else len(g) - 1 - reach
            kind = int(rng.integers(0, 4))
            if kind == 0:
                continue
            if kind == 1:
                g[at] = b"ACGT"[(b"ACGT".index(g[at]) + int(rng.integers(1, 4))) % 4]
            elif kind == 2:
                ins = bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 12))))
                g[at:at] = ins
            else:
                del g[at:at + int(rng.integers(1, 12))]
        out.append(bytes(g))
    nst = 3 if rng.random() < 0.5 else 4
    stages, kk, dd = [], k, D
    for _ in range(nst):
        stages.append((kk, dd))
        kk = int(min(kk + int(rng.integers(3, 9)), 40)) if rng.random() < 0.7 else int(min(2 * kk + int(rng.integers(0, 20)), 120))
        dd = dd + int(rng.integers(20, 120)) if kk <= 40 else int(rng.integers(3 * kk, 6 * kk))
    return out, stages


def random_dna(total: int, nrec: int, seed: int) -> List[bytes]:
    """Uniform random ACGT split into nrec equal records (config 5 style input)."""
    rng = np.random.default_rng(seed)
    per = total // nrec
    return [_ACGT[rng.integers(0, 4, per, dtype=np.uint8)].tobytes() for _ in range(nrec)]


def longk_case(total: int, nrec: int, seed: int = 5) -> List[bytes]:
    """Config 5 shape: uniform random DNA in nrec records with planted structure, for k in the thousands.

    Random DNA has no repeated 5000-mers of its own, so everything a long-k stage finds comes from what is planted here
    (independently of the background size): a 40 kbp segment of record 0 copied into record 1 with ONE substitution in
    its middle (a bulge of k + 1 steps: collapsed when D > k + 1), and -- with four records or more -- a 30 kbp exact
    duplicate between records 2 and 3.  Needs records of at least 50 kbp."""
    seqs = [bytearray(s) for s in random_dna(total, nrec, seed)]
    assert nrec >= 2 and len(seqs[0]) >= 50_000
    seg = bytearray(seqs[0][5000:45000])
    seg[20000] = ord("ACGT"["ACGT".index(chr(seg[20000])) ^ 1])      # A<->C, G<->T
    seqs[1][1000:41000] = seg
    if nrec >= 4:
        seqs[3][7000:37000] = seqs[2][3000:33000]
    return [bytes(s) for s in seqs]


def small_case(seed: int) -> Tuple[List[bytes], int, int]:
    """A tiny randomised multi-record input plus (k, D) chosen to provoke many bulges.

    Mixes: low-complexity / repeated segments, SNPs, indels, inversions, duplicated
    records, records shorter than k, and (for some seeds) ambiguity codes that the
    reference replaces through glibc rand() (reference src/indexedsequence.cpp:31-37).
    """
    rng = np.random.default_rng(1_000_003 * (seed + 1))
    k = int(rng.choice([3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 16, 20, 25, 31, 32]))
    if seed % 11 == 7:
        k = int(rng.choice([33, 40, 64, 65, 100]))
    D = int(rng.integers(max(2, k // 2), 12 * k))
    nrec = int(rng.integers(1, 6))
    L0 = int(rng.integers(max(40, 3 * k), 2500))
    alpha = 4 if rng.random() < 0.7 else int(rng.integers(2, 4))
    anc = rng.integers(0, alpha, L0, dtype=np.uint8)
    # plant repeats
    for _ in range(int(rng.integers(0, 4))):
        if L0 > 4 * k + 10:
            ln = int(rng.integers(k, min(L0 // 3, 6 * k + 5)))
            a = int(rng.integers(0, L0 - ln))
            b = int(rng.integers(0, L0 - ln))
            seg = anc[a:a + ln].copy()
            if rng.random() < 0.4:
                seg = _COMP[seg[::-1]]
            anc[b:b + ln] = seg
    recs = []
    for r in range(nrec):
        g = anc.copy()
        rate = float(rng.choice([0.0, 0.005, 0.02, 0.05]))
        m = rng.random(L0) < rate
        g[m] = (g[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) % 4
        pieces = []
        p = 0
        nind = int(rng.integers(0, max(1, L0 // 150)))
        for q in np.sort(rng.choice(L0, nind, replace=False)) if nind else []:
            q = int(q)
            if q < p:
                continue
            pieces.append(g[p:q])
            if rng.random() < 0.5:
                pieces.append(rng.integers(0, 4, int(rng.integers(1, 2 * k + 2)), dtype=np.uint8))
                p = q
            else:
                p = min(L0, q + int(rng.integers(1, 2 * k + 2)))
        pieces.append(g[p:])
        g = np.concatenate(pieces)
        if rng.random() < 0.3 and len(g) > 3 * k:
            a = int(rng.integers(0, len(g) - 2 * k))
            b = a + int(rng.integers(k, len(g) - a))
            g[a:b] = _COMP[g[a:b][::-1]]
        if rng.random() < 0.15:
            g = g[: int(rng.integers(1, k + 2))]          # shorter than / around k
        s = bytearray(_ACGT[g].tobytes())
        if seed % 5 == 3 and len(s) > 4:
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(0, len(s)))] = int(rng.choice(list(b"NRYKMSWBDHX-U")))
        recs.append(bytes(s))
    return recs, k, D


# ----------------------------------------------------------------------------- FASTA

def read_fasta(path: str) -> Tuple[List[str], List[bytes]]:
    opener = gzip.open if path.endswith(".gz") else open
    names: List[str] = []
    seqs: List[bytearray] = []
    with opener(path, "rb") as f:
        for raw in f:
            line = raw.strip()
            if not line:
                continue
            if line[:1] == b">":
                hdr = line[1:].split(b" ")[0]
                if not hdr:
                    raise ValueError("empty header")
                names.append(hdr.decode())
                seqs.append(bytearray())
            else:
                up = line.upper()
                if up.translate(None, VALID_CHARS):
                    raise ValueError("illegal character in sequence")
                if not seqs:
                    raise ValueError("sequence before header")
                seqs[-1] += up
    for s in seqs:
        if not s:
            raise ValueError("empty sequence")
    return names, [bytes(s) for s in seqs]


def write_fasta(path: str, seqs: Sequence[bytes], names: Sequence[str] | None = None, width: int = 80) -> None:
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            nm = names[i] if names else "seq%d" % i
            f.write(b">" + nm.encode() + b"\n")
            for j in range(0, len(s), width):
                f.write(s[j:j + width] + b"\n")


def input_digest(seqs: Sequence[bytes]) -> str:
    h = hashlib.sha256()
    for s in seqs:
        h.update(len(s).to_bytes(8, "little"))
        h.update(s)
    return h.hexdigest()


def strand_kmers(seqs: Sequence[bytes], k: int) -> int:
    """N = 2 * sum(max(0, len - k + 1)): the metric's unit (SURVEY.md §8 notation)."""
    return 2 * sum(max(0, len(s) - k + 1) for s in seqs)
