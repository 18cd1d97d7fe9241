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
def test_compressed_database(fixture):
    """<db>.sqz + <db>.tbz (pm_compress_seq -z true; char_io.t:18-214): the stream is the normalized one plus the
    end-of-sequence codes that fill the last buffer, so the output is the normalized database's (the real
    reference prints the same for -D 3 and -D 4).  Picked by itself when there is no .sqn (select.t:74), or -D 4."""
    g = load(fixture)
    with tempfile.TemporaryDirectory() as d:
        prepare(g, d)
        os.mkdir(os.path.join(d, "compressed"))
        fa = os.path.join(d, "compressed", "db.fa")
        with open(fa, "w") as f:
            f.write(g["fasta"])
        r = subprocess.run([CS, "-i", fa, "-z", "true"], capture_output=True)
        assert r.returncode == 0, r.stderr
        assert os.path.exists(fa + ".sqz") and not os.path.exists(fa + ".sqn") and not os.path.exists(fa + ".seq")
        for case in g["cases"]:
            want = sorted(g["cases"][case]["normalized"].splitlines())
            for more in ([], ["-D", "4"]):
                assert sorted(run_case(g, d, case, "compressed", more).splitlines()) == want, (fixture, case, more)


def test_refusals():
    g = load("cli_a")
    with tempfile.TemporaryDirectory() as d:
        prepare(g, d)
        fa = os.path.join(d, "normalized", "db.fa")
        for bad in (["-T"], ["-k", ".1"], ["-D", "1"], ["-M", "3"], ["-a"]):
            r = subprocess.run([PM, "-i", fa, "-p", "ACGTACGTACGTACGT"] + bad, capture_output=True)
            assert r.returncode == 1 and r.stdout == b"", bad
