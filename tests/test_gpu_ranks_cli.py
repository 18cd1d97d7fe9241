"""GPU: the command lines as position-sharded multi-rank programs (`--ranks N`, SURVEY.md 8(e),
BASELINE config 5): one process per rank, every rank scans its shard (+ guard band) on the GPU,
the count exchange and the record gather run in C++ (host/pm_ranks.cc; RCCL send/recv behind
include/pm_gpu.h pm_comm_* when every rank has a GPU of its own -- on this one-GPU box the ranks
share the card and the records travel through the launcher's pipes), rank 0 pairs
(pcr_match.cc:948-1259) / re-aligns and prints.  The output must equal the single-rank output and
the goldens the real reference produced; amplicons and hits that straddle shard edges included.

Ordering contract (checked here byte for byte): one find_patterns call returns its hits in (end, id, k)
order; `--ranks N` covers the whole stream in one call, the single-rank program in calls of 2^30 stream
bytes -- on databases below that size the two print exactly the same bytes.  (Above it, a chain of
candidates deferred across a call's edge is reported by the later call, as the reference's own
filter_bitvec does, filter_bitvec.cc:118-121, so primer_match's line order may differ at those edges;
pcr_match sorts its hits, pcr_match.cc:958.)  Against the goldens the comparison is a multiset of
lines: the reference engines order hits that end at the same position differently (sortedvector.t:490-510
sorts by position only, unstably)."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sequence-alignment-tools_amd", "host")
PCR = os.path.join(HOST, "pm_pcr_match")
PM = os.path.join(HOST, "pm_primer_match")
CS = os.path.join(HOST, "pm_compress_seq")
REF_PCR = os.path.join(ROOT, "oracle", "_ref", "pcr_match")
FLAG = {"S": "-S", "P": "-P", "Q": "-P", "F": "-F"}


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, timeout=600, **kw)
    assert r.returncode == 0, (cmd, r.stderr[-800:])
    return r.stdout.decode("latin1")


def load(fixture):
    with open(os.path.join(ROOT, "tests", "golden", fixture + ".json")) as f:
        return json.load(f)


@pytest.mark.parametrize("fixture", ["pcr_a", "pcr_b"])
def test_pcr_match_ranks_equal_reference(fixture):
    g = load(fixture)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            f.write(g["fasta"])
        run([CS, "-i", fa, "-n", "true"])
        for k, text in g["primers"].items():
            with open(os.path.join(d, "primers." + k), "w") as f:
                f.write(text)
        n = os.path.getsize([os.path.join(d, x) for x in os.listdir(d) if x.endswith(".sqn")][0])
        for case, c in g["cases"].items():
            base = [PCR, "-i", fa, FLAG[c["primers"]], os.path.join(d, "primers." + c["primers"])] + c["options"]
            one = run(base)
            for ranks in (2, 5):
                got = run(base + ["--ranks", str(ranks)])
                assert got == one, (fixture, case, ranks)                 # byte for byte the single-rank output
                assert sorted(got.splitlines()) == sorted(c["stdout"].splitlines()), (fixture, case, ranks)
                assert len(got) == len(c["stdout"])
        # the fixture's 2400 bp amplicon in the third entry spans an edge of the five-rank split
        edges = [((n + 4) // 5) * r for r in range(1, 5)]
        assert any(2 * 3001 + 200 < e < 2 * 3001 + 2600 for e in edges)


@pytest.mark.parametrize("fixture", ["cli_a"])
def test_primer_match_ranks_equal_reference(fixture):
    g = load(fixture)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            f.write(g["fasta"])
        run([CS, "-i", fa, "-n", "true"])
        for src, key in (("P", "primers_txt"), ("F", "primers_fasta"), ("S", "primers_sts"), ("W", "primers_iupac")):
            with open(os.path.join(d, "primers." + src), "w") as f:
                f.write(g[key])
        for case, c in g["cases"].items():
            if c["primers"] == "p":
                parg = ["-p", " ".join(g["primers_txt"].split()[:5])]
            else:
                parg = ["-" + ("P" if c["primers"] == "W" else c["primers"]), os.path.join(d, "primers." + c["primers"])]
            for extra in (["--ranks", "3"], ["--ranks=2", "-N", "16"]):
                got = run([PM, "-i", fa] + parg + c["options"] + extra)
                one = run([PM, "-i", fa] + parg + c["options"] + extra[-2:] * (extra[-2] == "-N"))
                assert got == one, (fixture, case, extra)                 # byte for byte the single-rank output (same -N)
                want = c["normalized"]
                assert sorted(got.splitlines()) == sorted(want.splitlines()), (fixture, case, extra)
                assert len(got) == len(want)


def test_shards_larger_than_the_guard_band():
    """1.2 Mbp in four entries, three ranks: shards of 400 kbp with 64 KiB guard bands (a rank's GPU
    really holds a part of the stream only), planted primer pairs -- one amplicon across each shard
    edge -- and a tandem repeat at an edge; 1 rank == 3 ranks == 4 ranks for every option set, and the
    real reference agrees where it is present."""
    rng = np.random.default_rng(17)
    L = 300_000
    ents = ["".join(rng.choice(list("ACGT"), size=L).tolist()) for _ in range(4)]
    n = 4 * (L + 1) + 1
    edges = sorted({((n + 2) // 3) * r for r in (1, 2)} | {((n + 3) // 4) * r for r in (1, 2, 3)})
    pairs = []

    def plant(entry, a, amp, nsub_f=0, nsub_r=0):
        s = ents[entry]
        f, r = s[a:a + 20], s[a + amp - 22:a + amp]
        f = synth.mutate(rng, f, nsub=nsub_f) if nsub_f else f
        r = synth.mutate(rng, r, nsub=nsub_r) if nsub_r else r
        pairs.append((f, synth.revcomp(r)))

    for e in edges:                                   # stream index e lies in entry (e - 1) // (L + 1), offset (e - 1) % (L + 1)
        ent, off = (e - 1) // (L + 1), (e - 1) % (L + 1)
        if 600 < off < L - 600:
            plant(ent, off - 300, 700, 1, 0)          # amplicon across the edge
            plant(ent, off - 10, 400, 0, 1)           # forward primer itself across the edge
    plant(0, 1000, 250)
    plant(3, 200_000, 900, 1, 1)
    # a tandem repeat across the first edge: filter_bitvec clusters the whole run into one hit per pattern
    ent, off = (edges[0] - 1) // (L + 1), (edges[0] - 1) % (L + 1)
    rep = "ACGGT" * 40
    ents[ent] = ents[ent][:off - 100] + rep + ents[ent][off - 100 + len(rep):]
    pairs.append((("ACGGT" * 4), synth.revcomp(ents[ent][off + 300:off + 320])))
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            for i, s in enumerate(ents):
                f.write(">ctg%d\n" % (i + 1))
                f.write("".join(s[j:j + 80] + "\n" for j in range(0, len(s), 80)))
        run([CS, "-i", fa, "-n", "true"])
        pp = os.path.join(d, "pairs.txt")
        with open(pp, "w") as f:
            f.write("".join("%s %s\n" % p for p in pairs))
        fmt = "%i %r %>s %<e %l %>d %<d %H\\n"
        for opts in (["-k", "1"], ["-K", "2"], ["-k", "2"], []):
            base = [PCR, "-i", fa, "-P", pp, "-r", "-M", "1000", "-A", fmt] + opts
            one = run(base)
            assert one.strip(), opts
            for ranks in (3, 4):
                got = run(base + ["--ranks", str(ranks)])
                assert got == one, (opts, ranks)
            if os.path.exists(REF_PCR) and opts != ["-K", "2"]:        # the reference's -K 2 takes minutes at this size
                ref = run([REF_PCR, "-i", fa, "-P", pp, "-r", "-M", "1000", "-A", fmt] + opts)
                assert sorted(ref.splitlines()) == sorted(one.splitlines()), opts
        pf = os.path.join(d, "primers.txt")
        with open(pf, "w") as f:
            f.write("".join("%s\n%s\n" % p for p in pairs))
        for opts in (["-k", "1"], ["-K", "1"], ["-K", "2"], ["-k", "2"], []):
            base = [PM, "-i", fa, "-P", pf, "-r", "-A", "%i %r %s %e %d %H\\n"] + opts
            one = run(base)
            for ranks in (3, 4):
                assert run(base + ["--ranks", str(ranks)]) == one, (opts, ranks)


def test_primers_on_stdin_reach_every_rank():
    """`-P -` / `-S -` with --ranks: the primers are read before the ranks are forked (the rank processes would
    otherwise share one stdin: one drains it, or each builds tables from a different slice)."""
    g = load("cli_a")
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            f.write(g["fasta"])
        run([CS, "-i", fa, "-n", "true"])
        pf = os.path.join(d, "primers.txt")
        with open(pf, "w") as f:
            f.write(g["primers_txt"])
        fmt = "%i %r %s %e %d %H\\n"
        for opts in (["-k", "1"], ["-K", "2"]):
            want = run([PM, "-i", fa, "-P", pf, "-r", "-A", fmt] + opts)
            assert want.strip()
            for ranks in (2, 3):
                got = run([PM, "-i", fa, "-P", "-", "-r", "-A", fmt, "--ranks", str(ranks)] + opts, input=g["primers_txt"].encode())
                assert got == want, (opts, ranks)
    g = load("pcr_a")
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            f.write(g["fasta"])
        run([CS, "-i", fa, "-n", "true"])
        sts = os.path.join(d, "pairs.sts")
        with open(sts, "w") as f:
            f.write(g["primers"]["S"])
        want = run([PCR, "-i", fa, "-S", sts, "-k", "1"])
        got = run([PCR, "-i", fa, "-S", "-", "-k", "1", "--ranks", "2"], input=g["primers"]["S"].encode())
        assert want.strip() and got == want


def test_a_failing_rank_takes_the_run_down():
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fa")
        with open(fa, "w") as f:
            f.write(">x\n" + "ACGT" * 200 + "\n")
        run([CS, "-i", fa, "-n", "true"])
        # -k 2 with 11 exact bases of a 12-mer: fatal in every rank (select.cc:87-90), exit status 1, no output
        r = subprocess.run([PM, "-i", fa, "-p", "ACGTACGTACGT", "-k", "2", "-s", "11", "--ranks", "2"], capture_output=True, timeout=120)
        assert r.returncode == 1 and r.stdout == b""
