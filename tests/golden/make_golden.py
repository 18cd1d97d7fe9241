#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (runs only in the build container).

Inputs are seeded synthetic FASTA databases and primer lists (tests/synth.py) plus the
reference's own data files db/pat.txt and db/test.seq (config 1 of BASELINE.json).  Outputs are
what the reference binaries built by oracle/Makefile into oracle/_ref/ print:

  * engine level  -- oracle/_ref/ref_harness: (end, id, value) triples at the
                     PatternMatch::find_patterns boundary (pattern_match.h:131), per -N engine;
  * CLI level     -- oracle/_ref/compress_seq -n true, then oracle/_ref/primer_match
                     -A '%i %r %s %e %S %E %d\\n': the lines a user sees.

The JSON files carry inputs and expected outputs only (data, no reference source).  Re-run:
    make -C oracle ref && python tests/golden/make_golden.py
"""
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
import refrun  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
HARNESS = os.path.join(REF, "ref_harness")

# (name, -N selector, k, indels)
ENGINE_CASES = [
    ("auto_k0", 0, 0, True), ("kt_list_k0", 1, 0, True), ("kt_dna_k0", 2, 0, True), ("shift_and_k0", 4, 0, True),
    ("auto_K1", 0, 1, False), ("auto_k1", 0, 1, True), ("auto_K2", 0, 2, False), ("auto_k2", 0, 2, True),
    ("filter_bitvec_K1", 5, 1, False), ("filter_bitvec_k1", 5, 1, True),
    ("exact_halves_sa_k1", 14, 1, True), ("exact_halves_kt_k2", 12, 2, True),
    ("inexact_raw_K1", 100, 1, False), ("inexact_raw_k1", 100, 1, True),
    ("inexact_raw_K2", 100, 2, False), ("inexact_raw_k2", 100, 2, True),
]


def cli_run(entries, patterns, k, indels, rc=True):
    """compress_seq -n true + primer_match with a machine-readable -A format."""
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fasta")
        with open(fa, "w") as f:
            for i, s in enumerate(entries):
                f.write(">e%d synthetic entry %d\n" % (i, i))
                for j in range(0, len(s), 60):
                    f.write(s[j:j + 60] + "\n")
        subprocess.run([os.path.join(REF, "compress_seq"), "-i", fa, "-n", "true"], check=True, capture_output=True)
        with open(fa + ".tbl", "rb") as f:
            tbl = f.read()
        with open(fa + ".sqn", "rb") as f:
            sqn = f.read()
        pf = os.path.join(d, "pat.txt")
        with open(pf, "w") as f:
            f.write("\n".join(patterns) + "\n")
        cmd = [os.path.join(REF, "primer_match"), "-i", fa, "-P", pf, "-A", "%i %r %s %e %S %E %d\n"]
        if k:
            cmd += ["-k" if indels else "-K", str(k)]
        if rc:
            cmd += ["-r"]
        out = subprocess.run(cmd, check=True, capture_output=True, text=True)
        lines = sorted(l for l in out.stdout.splitlines() if l.strip())
        return tbl, sqn, lines


def build_case(name, seed, n_entries, length, n_pat, L, **kw):
    rng = np.random.default_rng(seed)
    entries = synth.make_entries(rng, n_entries, length, n_runs=kw.get("n_runs", 2),
                                 repeats=kw.get("repeats", False), short=kw.get("short", True))
    pats = synth.make_patterns(rng, entries, n_pat, length=L, planted=kw.get("planted", 0.6),
                               minlen=kw.get("minlen"), indel_frac=kw.get("indel_frac", 0.3))
    if kw.get("boundary", True) and len(entries) >= 2:
        # a primer planted across an entry boundary: must never match (EOS inside the window)
        pats.append(entries[0][-(L // 2):] + entries[1][:L - L // 2])
    if kw.get("repeats", False):
        pats += ["AC" * 10, "A" * 20]
    table = synth.table_for(entries)
    raw = synth.stream(entries)
    codes = synth.normalize(raw, table)
    case = {"name": name, "seed": seed, "entries": entries, "patterns": pats,
            "table": table.decode("latin1"), "revcomp": True, "engine": {}, "cli": {}}
    for cname, sel, k, indels in ENGINE_CASES:
        hits = refrun.run_ref(HARNESS, codes, pats, table=table, sel=sel, k=k, indels=indels, rc=True, minka=37)
        case["engine"][cname] = {"sel": sel, "k": k, "indels": indels, "hits": hits}
    for cname, k, indels in [("k0", 0, True), ("K1", 1, False), ("k1", 1, True), ("K2", 2, False), ("k2", 2, True)]:
        tbl, sqn, lines = cli_run(entries, pats, k, indels)
        assert tbl == table, (tbl, table)
        assert sqn == codes.tobytes()
        case["cli"][cname] = {"k": k, "indels": indels, "format": "%i %r %s %e %S %E %d", "lines": lines}
    return case


def main():
    cases = [
        build_case("small_mixed", 20260101, 3, 4000, 120, 20),
        build_case("varlen_repeats", 20260102, 2, 1500, 60, 18, minlen=13, repeats=True, planted=0.8),
        build_case("dense_indels", 20260103, 4, 2500, 200, 20, indel_frac=0.8, planted=0.9, n_runs=4),
    ]
    for c in cases:
        with open(os.path.join(HERE, c["name"] + ".json"), "w") as f:
            json.dump(c, f, separators=(",", ":"))
        print(c["name"], {k: len(v["hits"]) for k, v in c["engine"].items()})

    # config 1 of BASELINE.json: the reference's own data files, exact match on the raw stream
    refdb = "/root/reference/db"
    with open(os.path.join(refdb, "test.seq"), "rb") as f:
        seq = f.read()
    with open(os.path.join(refdb, "pat.txt")) as f:
        pats = f.read().split()
    c1 = {"name": "config1_db_test_seq", "stream_latin1": seq.decode("latin1"), "patterns": pats, "engine": {}}
    for cname, sel in [("kt_list", 1), ("kt_jtable", 3), ("shift_and", 4)]:
        c1["engine"][cname] = {"sel": sel, "k": 0, "indels": True,
                               "hits": refrun.run_ref(HARNESS, seq, pats, sel=sel)}
    with open(os.path.join(HERE, "config1_db_test_seq.json"), "w") as f:
        json.dump(c1, f, separators=(",", ":"))
    print("config1", c1["engine"])


if __name__ == "__main__":
    main()
