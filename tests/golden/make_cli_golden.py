#!/usr/bin/env python3
"""Generate tests/golden/cli_*.json from the REAL reference command lines (build container only).

For a seeded FASTA database and primer files, runs oracle/_ref/compress_seq and
oracle/_ref/primer_match (the reference, compiled by oracle/Makefile) with a list of option sets
and stores: the inputs (FASTA text, primer file texts), the files compress_seq wrote (base64) and
primer_match's standard output per option set.  Data only -- no reference source.  Re-run:
    make -C oracle ref && python tests/golden/make_cli_golden.py
"""
import base64
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")

ONE_LINE = "%i %r %s %e %5 %3 %S %E %d %l %D %p %q %Q %t %T %U %A [%h|%H|%f] %| %^ %v %* %+ %%\\n"

# (name, primer source, extra options)
CASES = [
    ("k0_default", "P", ["-r"]),
    ("k1_default", "P", ["-r", "-k", "1"]),
    ("K1_default", "P", ["-r", "-K", "1"]),
    ("K2_oneline", "P", ["-r", "-K", "2", "-A", ONE_LINE]),
    ("k2_oneline", "P", ["-r", "-k", "2", "-A", ONE_LINE]),
    ("k1_oneline_fwd_only", "P", ["-k", "1", "-A", ONE_LINE]),
    ("k2_wrapped", "P", ["-r", "-k", "2", "-A", "%="]),
    ("k1_counts", "P", ["-r", "-k", "1", "-c"]),
    ("K1_counts_aggregate", "P", ["-r", "-K", "1", "-c", "-a"]),
    ("k1_counts_format_max", "P", ["-r", "-k", "1", "-C", "%i %p %q %r%R %c%+\\n", "-M", "1"]),
    ("k0_both", "P", ["-r", "-A", "%i %r %e\\n", "-C", "%i %r %c [%C]\\n"]),
    ("k1_start5", "P", ["-r", "-k", "1", "-s", "5", "-A", ONE_LINE]),
    ("k1_3prime4", "P", ["-r", "-k", "1", "-3", "4", "-A", ONE_LINE]),
    ("k2_5prime_inexact", "P", ["-r", "-k", "2", "-5", "~6", "-A", ONE_LINE]),
    ("k1_noindex", "P", ["-r", "-k", "1", "-I", "-A", "%i %r %s %e %S %E [%h]\\n"]),
    ("k1_fasta_primers", "F", ["-r", "-k", "1", "-A", "%P %i %r %e %d\\n"]),
    ("k1_sts_primers", "S", ["-k", "1", "-A", "%I %L %a [%O] %& %X %i %r %e %d\\n", "-C", "%I %L %i %r %c\\n"]),
    ("k0_inline_primers", "p", ["-r", "-A", "%i %r %s %e\\n"]),
    ("k1_report_interval", "P", ["-r", "-k", "1", "-R", "3", "-A", "%i %r %e %d\\n"]),
    ("k0_wildcards", "W", ["-r", "-w", "-A", ONE_LINE]),
    ("k0_wildcards_text_n", "W", ["-r", "-W", "-A", ONE_LINE]),
    ("k0_wildcards_counts", "W", ["-r", "-W", "-c"]),
    ("k1_wildcards", "W", ["-r", "-w", "-k", "1", "-A", ONE_LINE]),
    ("K2_wildcards_text_n", "W", ["-r", "-W", "-K", "2", "-A", ONE_LINE]),
    ("k2_wildcards_default", "W", ["-r", "-W", "-k", "2"]),
]


def build_inputs(seed):
    rng = np.random.default_rng(seed)
    ents = synth.make_entries(rng, 4, 700, n_runs=2, repeats=True, short=True)
    heads = ["chr%d synthetic entry %d len=%d" % (i + 1, i, len(s)) for i, s in enumerate(ents)]
    heads[1] = "chr2\ttab separated header"
    heads[2] = "nospaces"
    fasta = "".join(">%s\n%s" % (h, "".join(s[j:j + 60] + "\n" for j in range(0, len(s), 60))) for h, s in zip(heads, ents))
    text = "".join(ents[:4])
    pats = []
    for _ in range(14):
        L = int(rng.integers(16, 25))
        e = int(rng.integers(0, 4))
        a = int(rng.integers(0, len(ents[e]) - L))
        w = ents[e][a:a + L]
        if "N" in w:
            continue
        kind = int(rng.integers(0, 6))
        if kind == 1:
            w = synth.mutate(rng, w, nsub=1)
        elif kind == 2:
            w = synth.mutate(rng, w, nsub=2)
        elif kind == 3:
            w = synth.mutate(rng, w, nins=1)
        elif kind == 4:
            w = synth.mutate(rng, w, ndel=1)
        if int(rng.integers(0, 2)):
            w = synth.revcomp(w)
        pats.append(w)
    pats.append("".join(rng.choice(list("ACGT"), size=20).tolist()))      # no hit
    pats.append(pats[0])                                                   # duplicate primer
    pats.append("ACACACACACACACACACAC")                                     # tandem repeat
    assert text
    ptxt = "\n".join(pats) + "\n"
    pfa = "".join(">primer_%d some description\n%s\n" % (i + 1, p) for i, p in enumerate(pats))
    sts_lines = []
    for i in range(0, len(pats) - 1, 2):
        size = "150" if i % 4 == 0 else "100-%d" % (200 + i)
        sts_lines.append("STS%d\t%s\t%s\t%s\tACC%d\t%d\tALT%d\tHomo sapiens" % (i, pats[i], pats[i + 1], size, i, i % 23 + 1, i))
    # IUPAC primers for -w / -W: planted sites with some bases widened to ambiguity codes, one site
    # that covers an N run of the database (matches only with -W), one unrelated primer
    widen = {"A": "RMWN", "C": "YMSN", "G": "RKSN", "T": "YKWN"}
    wpats = []
    for _ in range(8):
        L = int(rng.integers(16, 23))
        e = int(rng.integers(0, 4))
        a = int(rng.integers(0, len(ents[e]) - L))
        w = list(ents[e][a:a + L])
        if "N" in w:
            continue
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(0, L))
            w[i] = str(rng.choice(list(widen.get(w[i], w[i]))))
        w = "".join(w)
        wpats.append(synth.revcomp_iupac(w) if int(rng.integers(0, 2)) else w)
    for e in range(4):
        i = ents[e].find("N")
        if i > 12 and i + 12 < len(ents[e]):
            site = ents[e][i - 10:i + 10]
            wpats.append("".join(c if c != "N" else "A" for c in site))
            wpats.append(site)                                              # pattern N against text N
            break
    wpats.append("ACGTRYKMACGTRYKMACGT")
    wtxt = "\n".join(wpats) + "\n"
    return fasta, ptxt, pfa, "\n".join(sts_lines) + "\n", pats, wtxt


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, check=False, **kw)


def main():
    for name, seed in [("cli_a", 11), ("cli_b", 12)]:
        fasta, ptxt, pfa, psts, pats, wtxt = build_inputs(seed)
        out = {"fasta": fasta, "primers_txt": ptxt, "primers_fasta": pfa, "primers_sts": psts, "primers_iupac": wtxt, "cases": {}, "db_files": {}}
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "db.fa")
            for variant, args in [("normalized", ["-n", "true"]), ("indexed", []), ("compressed", ["-z", "true"]), ("raw", None)]:
                sub = os.path.join(d, variant)
                os.mkdir(sub)
                fa = os.path.join(sub, "db.fa")
                with open(fa, "w") as f:
                    f.write(fasta)
                if args is None:                                    # the FASTA file itself (primer_match -D 1 / no database files)
                    continue
                r = run([os.path.join(REF, "compress_seq"), "-i", fa] + args)
                assert r.returncode == 0, r.stderr
                files = {}
                for ext in ("seq", "sqn", "tbl", "sqz", "tbz", "hdr", "idb"):
                    if os.path.exists(fa + "." + ext):
                        with open(fa + "." + ext, "rb") as f:
                            files[ext] = base64.b64encode(f.read()).decode()
                out["db_files"][variant] = files
            for src, text in (("P", ptxt), ("F", pfa), ("S", psts), ("W", wtxt)):
                with open(os.path.join(d, "primers." + src), "w") as f:
                    f.write(text)
            for cname, src, extra in CASES:
                for variant in ("normalized", "indexed", "compressed", "raw"):
                    fa = os.path.join(d, variant, "db.fa")
                    parg = ["-p", " ".join(pats[:5])] if src == "p" else ["-" + ("P" if src == "W" else src), os.path.join(d, "primers." + src)]
                    r = run([os.path.join(REF, "primer_match"), "-i", fa] + parg + extra)
                    assert r.returncode == 0, (cname, r.stderr[-500:])
                    out["cases"].setdefault(cname, {"primers": src, "options": extra})[variant] = r.stdout.decode("latin1")
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(out, f, indent=1)
        print(name, {k: len(v["normalized"].splitlines()) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
