"""GPU: the C++ host class (GpuPatternMatch, PatternMatch-shaped) driven by pm_scan_cli with
primer_match's scan loop, against the committed reference outputs."""
import json
import os
import subprocess
import tempfile

import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "sequence-alignment-tools_amd", "host", "pm_scan_cli")


def run_cli(codes, table, pats, extra):
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "db.sqn"), "wb") as f:
            f.write(codes.tobytes())
        with open(os.path.join(d, "db.tbl"), "wb") as f:
            f.write(table)
        with open(os.path.join(d, "pat.txt"), "w") as f:
            f.write("\n".join(pats) + "\n")
        out = subprocess.run([CLI, "-n", "-r", "-i", os.path.join(d, "db"), "-P", os.path.join(d, "pat.txt")] + extra,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        return sorted(tuple(int(x) for x in l.split()) for l in out.stdout.splitlines()), out.stderr


@pytest.mark.parametrize("case", ["small_mixed", "varlen_repeats"])
def test_cli_matches_reference(case):
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    with open(os.path.join(ROOT, "tests", "golden", case + ".json")) as f:
        c = json.load(f)
    table = c["table"].encode("latin1")
    codes = synth.normalize(synth.stream(c["entries"]), table)
    for name, extra in [("auto_k0", []), ("auto_K1", ["-K", "1"]), ("auto_k1", ["-k", "1"]), ("auto_K2", ["-K", "2"]),
                        ("auto_k2", ["-k", "2"])]:
        want = [tuple(h) for h in c["engine"][name]["hits"]]
        for more in ([], ["-m", "7", "-c", "1500"], ["-N", "16"], ["-B", "-m", "50"]):
            got, _ = run_cli(codes, table, c["patterns"], extra + more)
            assert got == want, (case, name, more)


def test_cli_config1_raw_stream():
    with open(os.path.join(ROOT, "tests", "golden", "config1_db_test_seq.json")) as f:
        c = json.load(f)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "test.seq"), "wb") as f:
            f.write(c["stream_latin1"].encode("latin1"))
        with open(os.path.join(d, "pat.txt"), "w") as f:
            f.write("\n".join(c["patterns"]) + "\n")
        out = subprocess.run([CLI, "-i", os.path.join(d, "test.seq"), "-P", os.path.join(d, "pat.txt")],
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.split() == ["27", "10", "0"]
