#!/usr/bin/env python3
"""Golden vectors for the FASTA reader, produced by the UNMODIFIED reference's FASTAReader
(reference src/fasta.cpp:23-104) through oracle/_ref/ref_dump (build container only).

tests/golden/fasta_cases.json: per case the file text (base64) and either the parsed result (records: description,
sequence) or the reference's error message with the file name replaced by <file>."""
import base64, json, os, struct, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")

CASES = {
    "plain": b">a desc\nACGTACGT\nTTGA\n>b\nGGCC\n",
    "no_trailing_newline": b">a\nACGT\n>b\nTT",
    "crlf_and_blank_lines": b">a x y\r\nACGT\r\n\r\n  \r\nAC\r\n>b\r\n\r\nGG\r\n",
    "lowercase_and_ambiguity": b">chr1\nacgtnnRYkm\nswbdhwNX-u\n>chr2\nNNNN\n",
    "leading_and_trailing_blanks": b"   >a  \n  ACGT  \n\tTT\t\n>b\n A \n",
    "sequence_before_first_header": b"ACGT\nTT\n>a\nGG\n>b\nCC\n",
    "no_header_at_all": b"ACGT\nTTGA\n",
    "single_long_line": b">a\n" + b"ACGT" * 3000 + b"\n>b\n" + b"TGCA" * 2500,
    "many_short_records": b"".join(b">r%d\nAC%sGT\n" % (i, b"G" * (i % 7)) for i in range(300)),
    "header_with_pipes": b">gi|123|ref|NC_000001.1| Homo\nACGT\n",
    "err_empty_header": b">a\nACGT\n>\nTT\n",
    "err_empty_header_blank": b"> desc\nACGT\n",
    "err_empty_sequence_middle": b">a\n>b\nACGT\n",
    "err_empty_sequence_end": b">a\nACGT\n>b\n\n\n",
    "err_illegal_char": b">a\nACGT\nACZT\n",
    "err_illegal_inner_blank": b">a\nAC GT\n",
    "err_illegal_lowercase_reported_as_written": b">a\n\n\nACGT\nacjt\n",
    "err_two_errors_first_wins": b">a\nACGT\n>b\n>\nA!\n",
    "err_only_blank": b"\n\n  \n",
    "gt_inside_sequence_line": b">a\nAC>GT\n",
}


def main():
    out = {}
    for name, text in CASES.items():
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "in.fa")
            open(fa, "wb").write(text)
            r = subprocess.run([REF_DUMP, fa, os.path.join(d, "o"), "state"], capture_output=True)
            e = {"text_b64": base64.b64encode(text).decode()}
            if r.returncode == 4:
                e["error"] = open(os.path.join(d, "o.err")).read().replace(fa, "<file>")
            elif r.returncode == 0:
                b = open(os.path.join(d, "o.0.out"), "rb").read()
                names = open(os.path.join(d, "o.0.out.names")).read().split("\n")[:-1]
                nchr = struct.unpack_from("<I", b, 8)[0]
                off, recs = 12, []
                for c in range(nchr):
                    n = struct.unpack_from("<Q", b, off)[0]
                    off += 8
                    seq = b[off:off + n].decode()
                    off += n
                    pos = list(struct.unpack_from("<%dI" % n, b, off))
                    off += 4 * n
                    assert pos == list(range(n))
                    recs.append({"name": names[c], "seq": seq})
                e["records"] = recs
            else:
                raise SystemExit("%s: ref_dump rc %d: %s" % (name, r.returncode, r.stderr.decode()))
            out[name] = e
            print(name, e.get("error") or [(x["name"], len(x["seq"])) for x in e["records"]][:4])
    json.dump({"reference": "bioinf/Sibelia 3.0.7 FASTAReader (oracle/_ref/ref_dump state)", "cases": out},
              open(os.path.join(ROOT, "tests", "golden", "fasta_cases.json"), "w"), indent=0, separators=(",", ":"))


if __name__ == "__main__":
    main()
