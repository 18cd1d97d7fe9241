#!/usr/bin/env python3
"""Generate tests/golden/pcr_*.json from the REAL reference pcr_match (build container only).

A seeded FASTA database with planted amplicons (primer pairs at known distances, both strand
arrangements, some with edited primers, one pair spanning two entries, one beyond -M) is run
through oracle/_ref/compress_seq -n true and oracle/_ref/pcr_match with several option sets;
inputs and standard output are stored.  Data only -- no reference source.  Re-run:
    make -C oracle ref && python tests/golden/make_pcr_golden.py
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

REF = os.path.join(ROOT, "oracle", "_ref")

ONE = ("%i %r%R [%>s %>e %>S %>E %>d %>p %>q %>Q %>r%>R %>t %>T %>A] [%<s %<e %<S %<E %<d %<p %<q %<Q %<r%<R %<t %<T %<A] "
       "%l %>l %<l %N %H|%f %!>s|%!l|%!<e|%!>Q|%!<A %%\\n")

CASES = [
    ("sts_default", "S", []),
    ("sts_k1_default", "S", ["-k", "1"]),
    ("sts_K1_oneline", "S", ["-K", "1", "-A", ONE]),
    ("sts_k2_oneline", "S", ["-k", "2", "-A", ONE]),
    ("sts_k1_deviation", "S", ["-k", "1", "-d", "20", "-A", "%I %L %>L %<L %D %a %O %& %X %i %>s %<e %l\\n"]),
    ("sts_k1_unists", "S", ["-k", "1", "-A", "%0\\n"]),
    ("sts_k1_amplicon", "S", ["-k", "1", "-M", "400", "-A", "%i %r %l %@ %*\\n"]),
    ("pairs_r_k1", "P", ["-r", "-k", "1", "-A", ONE]),
    ("pairs_r_k1_allorient", "P", ["-r", "-a", "-k", "1", "-A", ONE]),
    ("pairs_r_k1_between", "P", ["-r", "-b", "-k", "1", "-m", "50", "-M", "600", "-A", ONE]),
    ("pairs_r_k1_window", "P", ["-r", "-k", "1", "-m", "200", "-M", "500", "-A", "%i %r %>s %<e %l\\n"]),
    ("pairs_norev_k0", "Q", ["-A", "%i %r %>s %<e %l %>r %<r\\n"]),
    ("pairs_r_k1_constraints", "P", ["-r", "-k", "1", "-3", "3", "-A", "%i %r %>s %<e %>d %<d\\n"]),
    ("fasta_pairs_k1", "F", ["-r", "-k", "1", "-A", "%>P|%<P %i %r %>s %<e\\n"]),
    ("pairs_r_k1_small_interval", "P", ["-r", "-k", "1", "-R", "2", "-A", "%i %r %>s %<e %l\\n"]),
]


def build_inputs(seed):
    rng = np.random.default_rng(seed)
    ents = [list(s) for s in synth.make_entries(rng, 3, 3000, n_runs=1)]
    pairs = []
    plan = [(0, 100, 180, 0, 0, False), (0, 900, 420, 1, 0, False), (1, 300, 260, 0, 1, True), (1, 1500, 800, 1, 1, False),
            (2, 200, 2400, 0, 0, False), (2, 1000, 150, 2, 0, True), (0, 2000, 330, 0, 2, False)]
    for e, a, amp, ef, er, minus in plan:
        Lf, Lr = int(rng.integers(18, 24)), int(rng.integers(18, 24))
        s = "".join(ents[e])
        fwd = s[a:a + Lf]
        rev_site = s[a + amp - Lr:a + amp]
        if "N" in fwd or "N" in rev_site:
            continue
        f = fwd if ef == 0 else synth.mutate(rng, fwd, nsub=1) if ef == 1 else synth.mutate(rng, fwd, ndel=1)
        r = rev_site if er == 0 else synth.mutate(rng, rev_site, nsub=1) if er == 1 else synth.mutate(rng, rev_site, nins=1)
        fp, rp = f, synth.revcomp(r)                       # primers as ordered: forward 5'->3', reverse on the other strand
        if minus:                                          # amplicon on the minus strand: swap roles
            fp, rp = synth.revcomp(r), f
        pairs.append((fp, rp, amp))
    # a pair whose ends lie in different entries (must not be reported)
    s0, s1 = "".join(ents[0]), "".join(ents[1])
    pairs.append((s0[2950:2970], synth.revcomp(s1[80:100]), 150))
    pairs.append(("".join(rng.choice(list("ACGT"), size=20).tolist()), "".join(rng.choice(list("ACGT"), size=20).tolist()), 100))
    entries = ["".join(e) for e in ents]
    fasta = "".join(">ctg%d description of contig %d\n%s" % (i + 1, i + 1, "".join(s[j:j + 70] + "\n" for j in range(0, len(s), 70)))
                    for i, s in enumerate(entries))
    sts = "".join("STS%d\t%s\t%s\t%s\tACC%d\t%d\tALT%d\tHomo sapiens\n" % (i + 1, f, r, ("%d" % amp) if i % 2 == 0 else "%d-%d" % (amp - 15, amp + 10), i + 1, i + 1, i + 1)
                  for i, (f, r, amp) in enumerate(pairs))
    ptxt = "".join("%s %s\n" % (f, r) for f, r, _ in pairs)
    qtxt = "".join("%s %s\n" % (f, synth.revcomp(r)) for f, r, _ in pairs)     # already oriented: no -r
    pfa = "".join(">pair%d_%s primer\n%s\n" % (i // 2 + 1, "fwd" if i % 2 == 0 else "rev", p)
                  for i, p in enumerate(x for f, r, _ in pairs for x in (f, r)))
    return fasta, sts, ptxt, qtxt, pfa


def main():
    for name, seed in [("pcr_a", 21), ("pcr_b", 22)]:
        fasta, sts, ptxt, qtxt, pfa = build_inputs(seed)
        out = {"fasta": fasta, "primers": {"S": sts, "P": ptxt, "Q": qtxt, "F": pfa}, "cases": {}}
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "db.fa")
            with open(fa, "w") as f:
                f.write(fasta)
            r = subprocess.run([os.path.join(REF, "compress_seq"), "-i", fa, "-n", "true"], capture_output=True)
            assert r.returncode == 0, r.stderr
            for k, text in out["primers"].items():
                with open(os.path.join(d, "primers." + k), "w") as f:
                    f.write(text)
            for cname, src, extra in CASES:
                flag = {"S": "-S", "P": "-P", "Q": "-P", "F": "-F"}[src]
                r = subprocess.run([os.path.join(REF, "pcr_match"), "-i", fa, flag, os.path.join(d, "primers." + src)] + extra, capture_output=True)
                assert r.returncode == 0, (cname, r.stderr[-500:])
                out["cases"][cname] = {"primers": src, "options": extra, "stdout": r.stdout.decode("latin1")}
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(out, f, indent=1)
        print(name, {k: len(v["stdout"].splitlines()) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
