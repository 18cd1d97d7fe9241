"""GPU: pm_pcr_match (the reference's pcr_match command line -- primer-pair search with the
amplicon-length join -- on the MI355X engine) against the standard output of the real reference
pcr_match on the same database and primer-pair files (tests/golden/pcr_*.json, made by
tests/golden/make_pcr_golden.py).  Compared as sorted lines: the order of hits that end at one
position is engine specific."""
import json
import os
import subprocess
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sequence-alignment-tools_amd", "host")
PCR = os.path.join(HOST, "pm_pcr_match")
CS = os.path.join(HOST, "pm_compress_seq")
FLAG = {"S": "-S", "P": "-P", "Q": "-P", "F": "-F"}


@pytest.mark.parametrize("fixture", ["pcr_a", "pcr_b"])
def test_output_matches_reference(fixture):
    assert os.path.exists(PCR) and os.path.exists(CS), "run __graft_entry__.build()"
    with open(os.path.join(ROOT, "tests", "golden", fixture + ".json")) as f:
        g = json.load(f)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            f.write(g["fasta"])
        r = subprocess.run([CS, "-i", fa, "-n", "true"], capture_output=True)
        assert r.returncode == 0, r.stderr
        for k, text in g["primers"].items():
            with open(os.path.join(d, "primers." + k), "w") as f:
                f.write(text)
        for case, c in g["cases"].items():
            for more in ([], ["-N", "16"]):
                r = subprocess.run([PCR, "-i", fa, FLAG[c["primers"]], os.path.join(d, "primers." + c["primers"])] + c["options"] + more,
                                   capture_output=True, timeout=300)
                assert r.returncode == 0, (case, r.stderr[-500:])
                got = r.stdout.decode("latin1")
                assert sorted(got.splitlines()) == sorted(c["stdout"].splitlines()), (fixture, case, more)
                assert len(got) == len(c["stdout"])


def test_odd_number_of_primers_is_refused():
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            f.write(">x\nACGTACGTACGTACGTACGTACGT\n")
        assert subprocess.run([CS, "-i", fa, "-n", "true"]).returncode == 0
        r = subprocess.run([PCR, "-i", fa, "-p", "ACGTACGTACGT ACGTACGTAC ACGTACGTAA"], capture_output=True)
        assert r.returncode == 1 and b"Odd number of primers" in r.stderr
