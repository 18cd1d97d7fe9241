"""CPU: the oracle (oracle/pm_oracle.c) against the committed reference outputs in tests/golden/."""
import glob
import json
import os

import numpy as np
import pytest

import synth
from oracle import pmoracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "*.json")) if "config1" not in p and not os.path.basename(p).startswith(("cli_", "pcr_")))


def load(path):
    with open(path) as f:
        c = json.load(f)
    table = c["table"].encode("latin1")
    raw = synth.stream(c["entries"])
    codes = synth.normalize(raw, table)
    pats = c["patterns"]
    allp = pats + [O.reverse_comp(p) for p in pats]
    return c, O.Text(codes, table), allp


def test_config1_known_answer():
    """BASELINE config 1: db/pat.txt vs db/test.seq, exact -> primer 10 ends at 27."""
    with open(os.path.join(GOLD, "config1_db_test_seq.json")) as f:
        c = json.load(f)
    text = O.Text(np.frombuffer(c["stream_latin1"].encode("latin1"), dtype=np.uint8))
    for name, e in c["engine"].items():
        got = O.sorted_tuples(O.find_all(text, c["patterns"], engine=e["sel"], k=0))
        assert got == [tuple(h) for h in e["hits"]] == [(27, 10, 0)], name


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
def test_engine_hits_match_reference(path):
    c, text, allp = load(path)
    for name, e in c["engine"].items():
        sel = e["sel"]
        if sel == 0:
            sel = O.pick_engine(text, allp, e["k"], e["indels"])
        got = O.sorted_tuples(O.find_all(text, allp, engine=sel, k=e["k"], indels=e["indels"]))
        assert got == [tuple(h) for h in e["hits"]], (c["name"], name)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-5] for p in CASES])
def test_cli_lines_match_reference(path):
    """primer_match -A '%i %r %s %e %S %E %d' rebuilt from oracle hits + the CLI re-alignment."""
    c, text, allp = load(path)
    n = len(c["patterns"])
    starts = []                       # stream index of the first base of every entry
    pos = 1
    for s in c["entries"]:
        starts.append(pos)
        pos += len(s) + 1
    starts = np.array(starts)
    for name, e in c["cli"].items():
        hits = O.find_all(text, allp, engine=O.AUTO, k=e["k"], indels=e["indels"])
        lines = []
        for end, pid, _ in O.sorted_tuples(hits):
            rc, st, en, ed, _ = O.cli_align(text, allp[pid - 1], end, e["k"], e["indels"])
            assert ed <= e["k"], "bogus hit"
            base = starts[np.searchsorted(starts, st, side="right") - 1]
            ind = pid - n if pid > n else pid
            lines.append("%d %s %d %d %d %d %d" % (ind, "R" if pid > n else "F", st - base, en - base, st, en, ed))
        assert sorted(lines) == e["lines"], (c["name"], name)
