#!/usr/bin/env python3
"""Regenerate tests/golden/vectors.json from the UNMODIFIED reference (build container only).

Test infrastructure.  Needs /root/reference and oracle/_ref/ref_dump (oracle/build_ref.sh compiles
the reference's sources in place with gcc/g++ directly — no CMake — together with the
oracle/ref_dump.cpp driver).  Nothing from the reference is copied into the repo: only inputs
(generated here, or the two example FASTA sets the reference ships) and the sha256 /
small literal outputs of the reference run on them are committed.

Each vector = an input + a command list for ref_dump; every command yields one output
file whose sha256 (and for tiny cases whose bytes) are stored.  Output formats:
  enum:K          u32 bif_count | per strand: u64 n, n x (u32 id, u32 chr, u32 pos) in (chr,pos) order
  stage:K:D:ITER  u64 bulges | u32 nchr | per chr: u64 len, len bytes, len x u32 original positions
  dot:K           text of BlockFinder::SerializeCondensedGraph(K)
  blocks:K:T:M:S  N2: BlockFinder::GenerateSyntenyBlocks(K, trimK=T, minSize=M, sharedOnly=S): u64 n, n x (i32 id, u32 chr, u64 start, u64 end)
  write:K:T:M:S:G N4: GenerateSyntenyBlocks, GlueStripes if G, then the texts of blocks_coords.txt / genomes_permutations.txt / coverage_report.txt
  graph:K         text of BlockFinder::SerializeGraph(K) (the uncondensed graph; only for inputs whose records all hold K + 1 characters: the
                  reference's SlidingWindow walks off the end of a shorter one)
  hash:K          H0: SlidingWindow hashes (src/hashing.h) of every K-mer of the current rawSeq_, per (strand, chr): u64 n, n x u64
"""
import base64, gzip, hashlib, json, os, shutil, struct, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
from sibelia_amd import workloads as W  # noqa: E402

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
REF = "/root/reference"


def run_case(seqs, cmds, keep_bytes=False, timeout=None):
    d = tempfile.mkdtemp(prefix="gold")
    try:
        fa = os.path.join(d, "in.fa")
        W.write_fasta(fa, seqs)
        subprocess.run([REF_DUMP, fa, os.path.join(d, "o")] + cmds, check=True,
                       stderr=subprocess.DEVNULL, timeout=timeout)
        outs = []
        for ci, c in enumerate(cmds):
            p = c.split(":")
            b = open(os.path.join(d, "o.%d.out" % ci), "rb").read()
            e = {"cmd": c, "sha256": hashlib.sha256(b).hexdigest(), "size": len(b)}
            if p[0] == "enum":
                e["bif_count"] = struct.unpack_from("<I", b, 0)[0]
                npos = struct.unpack_from("<Q", b, 4)[0]
                nneg = struct.unpack_from("<Q", b, 12 + 12 * npos)[0]
                e["instances"] = [npos, nneg]
            if p[0] == "stage":
                e["bulges"] = struct.unpack_from("<Q", b, 0)[0]
            if keep_bytes:
                e["b64"] = base64.b64encode(b).decode()
            outs.append(e)
        return outs
    finally:
        shutil.rmtree(d)


HAND = {
    # SURVEY.md Appendix A.1 / A.2 (SNP bulge; indel + N + reverse complement)
    "snp_k5": (["ACGTTGCAAGGCTTACGGATCCATGACCTGAATCGTTAGC", "ACGTTGCAAGGCTAACGGATCCATGACCTGAATCGTTAGC"],
               ["enum:5", "dot:5", "stage:5:12:4", "enum:5", "dot:5", "hash:5", "hash:1", "hash:40", "hash:41", "graph:5", "graph:39"]),
    "indel_N_rc_k5": (["ACGTTGCAAGGCTTACGGATCCATGACCTGAATCGTTAGC",
                       "ACGTTGCAAGGCTTATCACGGATCCATGACCTGAATCGTTAGC",
                       "GCTAACGATTCAGGTCATGGATCCGTNAGCCTTGCAACGT"],
                      ["dot:5", "stage:5:12:4", "enum:5", "dot:5", "blocks:5:4:8:0", "blocks:5:3:5:1"]),
    "palindrome_k4": (["AACGCGTTAGCTAGGATCCTTAATTAAGG", "AACGCGTTAGGTAGGATCCTTAATTCAGG"],
                      ["enum:4", "stage:4:9:4", "dot:4"]),
    "tandem_k3": (["ACGACGACGACGTTACGACGACG", "ACGACGACGTTTACGACGACGACG"], ["enum:3", "stage:3:8:4", "dot:3"]),
    "short_chr_k6": (["ACGTA", "ACGTAC", "ACGTACG", "TTACGTACGGA", "A"], ["enum:6", "stage:6:10:4", "dot:6", "hash:6", "hash:2"]),
    "identical_k5": (["ACGTTGCATGCCGTAAGCTTGGA"] * 3, ["enum:5", "stage:5:10:4", "dot:5", "blocks:5:4:6:0", "blocks:5:5:10:1", "write:5:4:6:0:1"]),
    "three_way_k4": (["TTGACCAGTACGGTCAATGCCATAGGCTAAGC", "TTGACCAGTTCGGTCAATGCGATAGGCTAAGC",
                      "TTGACCAGTGCGGTCAATGCTATAGGCTAAGC", "TTGACCAGTACGGTCAATGCCATAGGCTAAGC"],
                     ["enum:4", "stage:4:10:4", "dot:4"]),
    "ambig_codes_k4": (["ACGTNNACGTRYACGTKMACGT-ACGTXACGU", "NACGTTGCAACGTNACGTTGCAAN"],
                       ["hash:4", "graph:3", "blocks:4:3:6:0", "dot:4", "stage:4:8:4", "dot:4", "stage:6:12:2", "dot:6", "hash:33", "blocks:6:4:8:0"]),
    "single_base_iter1": (["ACGTTGCAAGGCTTACGGATCCATGACCTGAATCGTTAGC", "ACGTTGCAAGGCTAACGGATCCATGACCTGAATCGTTAGC"],
                          ["stage:5:12:1", "stage:5:3:4", "stage:2:4:4", "dot:2"]),
}


def cases(skipped):
    for name, (seqs, cmds) in HAND.items():
        yield "hand/" + name, {"kind": "literal", "seqs": seqs}, (lambda seqs=seqs: [x.encode() for x in seqs]), cmds, True, None
    for seed in range(240):
        if "small/%03d" % seed in skipped:
            continue
        seqs, k, D = W.small_case(seed)
        k2 = max(2, k // 2) if seed % 3 == 0 else min(2 * k, 130)
        cmds = ["enum:%d" % k, "stage:%d:%d:4" % (k, D), "dot:%d" % k]
        if seed % 2 == 0:
            cmds += ["stage:%d:%d:3" % (k2, 3 * D), "enum:%d" % k2]
        if seed % 5 == 0:                       # N2 on what the stages left (trimK <= k, small minimum block size)
            kb = k2 if seed % 2 == 0 else k
            cmds += ["blocks:%d:%d:%d:%d" % (kb, max(2, min(kb, 5 + seed % 7)), kb + seed % 13, seed % 3 == 0)]
            if seed % 10 == 0:
                cmds += ["write:%d:%d:%d:0:1" % (kb, max(2, min(kb, 5 + seed % 7)), kb + seed % 13)]
        yield "small/%03d" % seed, {"kind": "small_case", "seed": seed}, (lambda seqs=seqs: seqs), cmds, False, int(os.environ.get("GOLDEN_SMALL_TIMEOUT", "20"))
    # three- and four-stage cascades on small strain sets (k 15-40 first, collapses at record ends, indels near record boundaries):
    # the state carried across copy-backs -- positions re-interpolated by earlier collapses, closing separators restamped with
    # the current record lengths (dnasequence.cpp:96) -- is what later stages read; every intermediate state is an output
    for seed in range(48):
        seqs, stages = W.cascade_case(seed)
        cmds = ["stage:%d:%d:4" % st for st in stages] + ["dot:%d" % stages[-1][0]]
        if seed % 4 == 0:
            cmds += ["blocks:%d:%d:%d:0" % (stages[-1][0], min(stages[0][0], stages[-1][0]), 2 * stages[-1][0])]
        yield "cascade/%03d" % seed, {"kind": "cascade_case", "seed": seed}, (lambda seqs=seqs: seqs), cmds, False, 120
    # found by tools/stress.py in round 2 (a collapse at the very end of a chromosome in the SECOND stage): now a fixture of the reference
    kw = dict(L0=13939, n=8, seed=110467, snp=0.03, indel_every=1000, inv_min=139, inv_max=696)
    yield ("cascade/stress_110467", {"kind": "gen_strains", "args": kw}, (lambda kw=kw: W.gen_strains(**kw)),
           ["stage:31:124:4", "stage:36:174:4", "stage:40:250:4", "dot:40"], False, 300)
    hp = "Helicobacter_pylori.fa.gz"
    sa = "Staphylococcus_aureus_pair.fa.gz"
    data = os.path.join(ROOT, "tests", "golden", "data")
    rd = lambda f: (lambda: W.read_fasta(os.path.join(data, f))[1])
    yield "real/hpylori_k25", {"kind": "fasta", "file": hp}, rd(hp), ["enum:25", "stage:25:150:4", "enum:25", "dot:25", "hash:25"], False, None
    yield "real/hpylori_fine", {"kind": "fasta", "file": hp}, rd(hp), ["stage:30:150:4", "stage:100:500:4", "stage:500:1500:4", "dot:500", "blocks:500:30:500:0", "blocks:500:30:5000:1", "write:500:30:500:0:1", "write:500:30:500:0:0"], False, None
    yield ("real/hpylori_loose", {"kind": "fasta", "file": hp}, rd(hp),
           ["stage:30:150:4", "stage:100:1000:4", "stage:1000:5000:4", "stage:5000:15000:4", "enum:5000", "blocks:5000:30:5000:0", "write:5000:30:5000:0:1"], False, None)
    yield "real/saureus_k25", {"kind": "fasta", "file": sa}, rd(sa), ["enum:25", "stage:25:150:4", "enum:25", "blocks:25:25:200:0", "write:25:25:200:0:1"], False, None
    small_inv = dict(inv_min=2000, inv_max=9000)
    synth = [
        ("synth/strains4_100k", dict(L0=100_000, n=4, seed=7, **small_inv), ["enum:25", "stage:25:150:4", "enum:25", "dot:25"]),
        ("synth/strains4_100k_fine", dict(L0=100_000, n=4, seed=7, **small_inv),
         ["stage:30:150:4", "stage:100:500:4", "stage:500:1500:4", "enum:500", "blocks:500:30:500:0", "blocks:100:30:300:1", "write:500:30:500:0:1", "write:100:30:300:1:1", "graph:21"]),
        ("synth/strains3_400k_k16", dict(L0=400_000, n=3, seed=11, inv_min=5000, inv_max=20000), ["enum:16", "stage:16:120:4", "enum:31"]),
        ("synth/strains2_4600k", dict(L0=4_600_000, n=2, seed=1), ["enum:25", "stage:25:150:4"]),
        ("synth/strains8_4600k", dict(L0=4_600_000, n=8, seed=1), ["enum:25", "stage:25:150:4"]),
        # BASELINE.json config 3 at full size: 8 strains, -s fine cascade, then the synteny stage and the reports (~10 min of reference time)
        ("synth/strains8_4600k_fine", dict(L0=4_600_000, n=8, seed=1),
         ["stage:30:150:4", "stage:100:500:4", "stage:500:1500:4", "enum:500", "blocks:500:30:500:0", "write:500:30:500:0:1"]),
    ]
    for name, kw, cmds in synth:
        yield name, {"kind": "gen_strains", "args": kw}, (lambda kw=kw: W.gen_strains(**kw)), cmds, False, None


def write_real_inputs():
    """gzip the two example inputs the reference ships (data fixtures, deterministic gzip header)."""
    data = os.path.join(ROOT, "tests", "golden", "data")
    os.makedirs(data, exist_ok=True)
    hp = os.path.join(REF, "examples/Sibelia/Helicobacter_pylori/Helicobacter_pylori.fasta")
    sa = [os.path.join(REF, "examples/C-Sibelia/Staphylococcus_aureus", f) for f in ("NCTC8325.fasta", "RN4220.fasta")]
    hp_seqs = W.read_fasta(hp)[1]
    sa_seqs = []
    for f in sa:
        sa_seqs += W.read_fasta(f)[1]
    for nm, seqs in (("Helicobacter_pylori.fa.gz", hp_seqs), ("Staphylococcus_aureus_pair.fa.gz", sa_seqs)):
        tmp = os.path.join(data, nm[:-3])
        W.write_fasta(tmp, seqs)
        with open(tmp, "rb") as fi, gzip.GzipFile(os.path.join(data, nm), "wb", 9, mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
        os.remove(tmp)


def main():
    """usage: make_golden.py [--only PREFIX ...] [--big]   (without --big the 8-strain vector is not regenerated)"""
    out = os.path.join(ROOT, "tests", "golden", "vectors.json")
    old = json.load(open(out)) if os.path.exists(out) else {"vectors": [], "skipped": []}
    only = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--only"]
    skipped = set(old.get("skipped", []))
    byname = {v["name"]: v for v in old["vectors"]}
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "data", "Helicobacter_pylori.fa.gz")):
        write_real_inputs()
    order = []
    for name, spec, get, cmds, keep, tmo in cases(skipped):
        order.append(name)
        if only and not any(name.startswith(p) for p in only):
            continue
        if name.startswith("synth/strains8_4600k") and "--big" not in sys.argv:
            continue
        if not only and name in byname and [o["cmd"] for o in byname[name]["outputs"]] == cmds:
            continue                                   # already present with the same command list
        seqs = get()
        try:
            outs = run_case(seqs, cmds, keep_bytes=keep, timeout=tmo)
        except subprocess.TimeoutExpired:              # pathological low-complexity case: reference too slow
            print(name, "skipped (reference > %d s)" % tmo, flush=True)
            skipped.add(name)
            continue
        byname[name] = {"name": name, "input": spec, "input_sha256": W.input_digest(seqs), "outputs": outs}
        print(name, [(o["cmd"], o.get("bulges"), o.get("bif_count")) for o in outs], flush=True)
    vec = [byname[n] for n in order if n in byname]
    json.dump({"reference": "bioinf/Sibelia 3.0.7 (unmodified, sources compiled in place by oracle/build_ref.sh, g++ 11.4, -O3 -DNDEBUG)",
               "skipped": sorted(skipped), "vectors": vec}, open(out, "w"), indent=0, separators=(",", ":"))
    print("wrote", out, len(vec), "vectors")


if __name__ == "__main__":
    main()
