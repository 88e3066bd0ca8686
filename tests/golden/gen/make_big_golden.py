#!/usr/bin/env python3
"""Full-size reference fixtures for BASELINE.json configs 4 and 5 (build container only; test infrastructure).

The unmodified reference (oracle/_ref/ref_dump, built by oracle/build_ref.sh) is run ONCE on inputs no test could afford to
hand to a CPU implementation -- 62 strains (config 4) and 900 Mbp of random DNA at k = 5000 (config 5) -- and only the sha256,
the size and a few counts of each output are committed (tests/golden/big_vectors.json).  The outputs themselves are gigabytes:
they are hashed as a stream and deleted.  Formats as in make_golden.py (enum:K, stage:K:D:ITER).

usage: make_big_golden.py NAME [NAME ...]        (names of CASES below; existing entries of other names are kept)
"""
import hashlib, json, os, re, shutil, struct, subprocess, sys, tempfile, time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
from sibelia_amd import workloads as W  # noqa: E402

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
OUT = os.path.join(ROOT, "tests", "golden", "big_vectors.json")

CASES = {
    # config 4 at a tenth of the record length (the reference needs ~30 min) and at full size (hours)
    "synth/strains62_460k": ({"kind": "gen_strains", "args": dict(L0=460_000, n=62, seed=1)}, ["enum:25", "stage:25:150:4"]),
    "synth/strains62_4600k": ({"kind": "gen_strains", "args": dict(L0=4_600_000, n=62, seed=1)}, ["enum:25", "stage:25:150:4"]),
    # config 5 at full size: 4 x 225 Mbp of random DNA with the planted segments of W.longk_case, k = 5000, D = 15000
    "synth/random4x225M_k5000": ({"kind": "longk_case", "args": dict(total=900_000_000, nrec=4)},
                                 ["enum:5000", "stage:5000:15000:4", "enum:5000"]),
}


def make_input(spec):
    if spec["kind"] == "gen_strains":
        return W.gen_strains(**spec["args"])
    if spec["kind"] == "longk_case":
        return W.longk_case(spec["args"]["total"], spec["args"]["nrec"])
    raise ValueError(spec["kind"])


def file_digest(path):
    h = hashlib.sha256()
    n = 0
    with open(path, "rb") as f:
        head = f.read(20)
        h.update(head)
        n += len(head)
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
            n += len(b)
    return h.hexdigest(), n, head


def run(name):
    spec, cmds = CASES[name]
    seqs = make_input(spec)
    d = tempfile.mkdtemp(prefix="biggold", dir=os.environ.get("BIGGOLD_TMP", "/tmp"))
    try:
        fa = os.path.join(d, "in.fa")
        W.write_fasta(fa, seqs)
        digest = W.input_digest(seqs)
        lens = [len(s) for s in seqs]
        del seqs
        t0 = time.time()
        r = subprocess.run([REF_DUMP, fa, os.path.join(d, "o")] + cmds, check=True, stderr=subprocess.PIPE, text=True)
        outs = []
        for ci, c in enumerate(cmds):
            p = os.path.join(d, "o.%d.out" % ci)
            sha, size, head = file_digest(p)
            e = {"cmd": c, "sha256": sha, "size": size}
            if c.startswith("enum"):
                e["bif_count"] = struct.unpack_from("<I", head, 0)[0]
                e["npos"] = struct.unpack_from("<Q", head, 4)[0]
            if c.startswith("stage"):
                e["bulges"] = struct.unpack_from("<Q", head, 0)[0]
                m = re.search(r"stage k=%s .*seconds=([0-9.]+)" % c.split(":")[1], r.stderr)
                if m:
                    e["reference_seconds"] = float(m.group(1))
            os.remove(p)
            outs.append(e)
        return {"name": name, "input": spec, "input_sha256": digest, "lengths": lens, "outputs": outs,
                "reference_wall_seconds": round(time.time() - t0, 1)}
    finally:
        shutil.rmtree(d)


def main():
    old = json.load(open(OUT)) if os.path.exists(OUT) else {"vectors": []}
    by = {v["name"]: v for v in old["vectors"]}
    for name in sys.argv[1:]:
        v = run(name)
        # re-read: several generators may run side by side
        old = json.load(open(OUT)) if os.path.exists(OUT) else {"vectors": []}
        by = {x["name"]: x for x in old["vectors"]}
        by[name] = v
        json.dump({"reference": "bioinf/Sibelia 3.0.7 (unmodified, sources compiled in place by oracle/build_ref.sh, g++ 11.4, -O3 -DNDEBUG)",
                   "vectors": [by[n] for n in CASES if n in by]}, open(OUT, "w"), indent=1)
        print(name, v["outputs"], flush=True)


if __name__ == "__main__":
    main()
