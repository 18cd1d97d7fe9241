"""GPU: pm_primer_match (the reference's primer_match command line on the MI355X engine) against
the standard output of the real reference primer_match on the same database and primer files
(tests/golden/cli_*.json, made by tests/golden/make_cli_golden.py).  The database files are
written by pm_compress_seq.  Engines report hits of one position in different orders, so output
is compared as a sorted list of lines (the reference's own testscript.sh sorts before cmp)."""
import json
import os
import subprocess
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sequence-alignment-tools_amd", "host")
PM = os.path.join(HOST, "pm_primer_match")
CS = os.path.join(HOST, "pm_compress_seq")


def load(fixture):
    with open(os.path.join(ROOT, "tests", "golden", fixture + ".json")) as f:
        return json.load(f)


def prepare(g, d):
    for variant, args in (("normalized", ["-n", "true"]), ("indexed", [])):
        os.mkdir(os.path.join(d, variant))
        fa = os.path.join(d, variant, "db.fa")
        with open(fa, "w") as f:
            f.write(g["fasta"])
        r = subprocess.run([CS, "-i", fa] + args, capture_output=True)
        assert r.returncode == 0, r.stderr
    for src, key in (("P", "primers_txt"), ("F", "primers_fasta"), ("S", "primers_sts"), ("W", "primers_iupac")):
        with open(os.path.join(d, "primers." + src), "w") as f:
            f.write(g[key])


def run_case(g, d, case, variant, more=()):
    c = g["cases"][case]
    fa = os.path.join(d, variant, "db.fa")
    if c["primers"] == "p":
        parg = ["-p", " ".join(g["primers_txt"].split()[:5])]
    else:
        parg = ["-" + ("P" if c["primers"] == "W" else c["primers"]), os.path.join(d, "primers." + c["primers"])]
    r = subprocess.run([PM, "-i", fa] + parg + c["options"] + list(more), capture_output=True, timeout=300)
    assert r.returncode == 0, (case, r.stderr[-500:])
    return r.stdout.decode("latin1")


@pytest.mark.parametrize("fixture", ["cli_a", "cli_b"])
def test_output_matches_reference(fixture):
    assert os.path.exists(PM) and os.path.exists(CS), "run __graft_entry__.build()"
    g = load(fixture)
    with tempfile.TemporaryDirectory() as d:
        prepare(g, d)
        for case in g["cases"]:
            for variant in ("normalized", "indexed"):
                got = run_case(g, d, case, variant)
                want = g["cases"][case][variant]
                assert sorted(got.splitlines()) == sorted(want.splitlines()), (fixture, case, variant)
                assert len(got) == len(want)


def test_kernel_families_and_output_file():
    g = load("cli_a")
    with tempfile.TemporaryDirectory() as d:
        prepare(g, d)
        for case in ("K2_oneline", "k2_oneline", "k1_default", "k0_default"):
            want = sorted(g["cases"][case]["normalized"].splitlines())
            for more in (["-N", "16"], ["-N", "17"], ["-B"]):
                assert sorted(run_case(g, d, case, "normalized", more).splitlines()) == want, (case, more)
        # -o appends (primer_match.cc:160-166)
        outf = os.path.join(d, "out.txt")
        with open(outf, "w") as f:
            f.write("first line\n")
        assert run_case(g, d, "k1_counts", "normalized", ["-o", outf]) == ""
        with open(outf) as f:
            text = f.read()
        assert text.startswith("first line\n")
        assert sorted(text.splitlines()[1:]) == sorted(g["cases"]["k1_counts"]["normalized"].splitlines())


@pytest.mark.parametrize("fixture", ["cli_a", "cli_b"])
def test_compressed_and_raw_databases(fixture):
    """<db>.sqz + <db>.tbz (pm_compress_seq -z true; char_io.t:18-214: the normalized stream plus the end-of-sequence
    codes that fill the last buffer), picked by itself when there is no .sqn (select.t:74) or with -D 4; and the FASTA
    file itself (StreamedFastaFile, fasta_io.t:448-751) when there are no database files (select.t:152) or with -D 1.
    Expected output: the real reference's on the same files (goldens "compressed" / "raw")."""
    g = load(fixture)
    with tempfile.TemporaryDirectory() as d:
        prepare(g, d)
        for variant in ("compressed", "raw"):
            os.mkdir(os.path.join(d, variant))
            fa = os.path.join(d, variant, "db.fa")
            with open(fa, "w") as f:
                f.write(g["fasta"])
        fa = os.path.join(d, "compressed", "db.fa")
        r = subprocess.run([CS, "-i", fa, "-z", "true"], capture_output=True)
        assert r.returncode == 0, r.stderr
        assert os.path.exists(fa + ".sqz") and not os.path.exists(fa + ".sqn") and not os.path.exists(fa + ".seq")
        for case in g["cases"]:
            for variant, flag in (("compressed", "4"), ("raw", "1")):
                want = sorted(g["cases"][case][variant].splitlines())
                for more in ([], ["-D", flag]):
                    assert sorted(run_case(g, d, case, variant, more).splitlines()) == want, (fixture, case, variant, more)
        # -D 1 reads the FASTA file even when compress_seq's files lie next to it
        assert sorted(run_case(g, d, "k1_oneline_fwd_only", "normalized", ["-D", "1"]).splitlines()) == sorted(g["cases"]["k1_oneline_fwd_only"]["raw"].splitlines())


REF_PM = os.path.join(ROOT, "oracle", "_ref", "primer_match")


@pytest.mark.skipif(not os.path.exists(REF_PM), reason="reference primer_match not built (oracle/_ref)")
def test_raw_fasta_layouts_against_the_reference_binary():
    """The FASTA file itself as database (-D 1) in layouts the reference reads without a warning: CR LF line ends,
    lower-case bases (matched only with -u), no newline behind the last line, a header with tabs, a one-line entry;
    same lines as the real primer_match prints."""
    import numpy as np
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), size=n).tolist()) for n in (300, 240, 60, 180)]
    primers = [seqs[0][40:60], seqs[1][100:122], seqs[3][10:30], seqs[2][5:25]]
    layouts = {
        "crlf": "".join(">e%d some text\r\n%s" % (i, "".join(s[j:j + 60] + "\r\n" for j in range(0, len(s), 60))) for i, s in enumerate(seqs)),
        "lower_no_final_newline": "".join(">e%d\ttabbed\n%s" % (i, "".join(s[j:j + 60].lower() + "\n" for j in range(0, len(s), 60))) for i, s in enumerate(seqs))[:-1],
    }
    with tempfile.TemporaryDirectory() as d:
        pf = os.path.join(d, "p.txt")
        with open(pf, "w") as f:
            f.write("\n".join(primers) + "\n")
        for name, text in layouts.items():
            fa = os.path.join(d, name + ".fa")
            with open(fa, "wb") as f:
                f.write(text.encode())
            for opts in (["-k", "1", "-r"], ["-k", "1", "-r", "-u"], ["-K", "2", "-r", "-u", "-A", "%i %r %s %e %d [%h|%H]\\n"]):
                want = subprocess.run([REF_PM, "-i", fa, "-P", pf, "-D", "1"] + opts, capture_output=True, timeout=120)
                got = subprocess.run([PM, "-i", fa, "-P", pf, "-D", "1"] + opts, capture_output=True, timeout=120)
                assert want.returncode == 0 and got.returncode == 0, (name, opts, got.stderr[-300:])
                assert sorted(got.stdout.splitlines()) == sorted(want.stdout.splitlines()), (name, opts)
            assert want.stdout.strip(), name


def test_refusals():
    g = load("cli_a")
    with tempfile.TemporaryDirectory() as d:
        prepare(g, d)
        fa = os.path.join(d, "normalized", "db.fa")
        for bad in (["-T"], ["-k", ".1"], ["-D", "7"], ["-M", "3"], ["-a"]):
            r = subprocess.run([PM, "-i", fa, "-p", "ACGTACGTACGTACGT"] + bad, capture_output=True)
            assert r.returncode == 1 and r.stdout == b"", bad
