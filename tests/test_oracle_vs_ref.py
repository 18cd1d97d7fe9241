"""CPU, build container only: fuzz the oracle against the real reference engines (oracle/_ref).
Skipped where oracle/_ref/ref_harness does not exist."""
import numpy as np
import pytest

import refrun
import synth
from oracle import pmoracle as O

CONFIGS = [(0, 0, 1), (4, 0, 1), (2, 0, 1), (100, 1, 0), (100, 2, 0), (100, 2, 1), (5, 1, 1), (5, 2, 0), (5, 2, 1),
           (12, 1, 0), (12, 1, 1), (14, 1, 1), (12, 2, 1), (0, 1, 1), (0, 1, 0), (0, 2, 1), (0, 2, 0)]


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_against_reference(ref_harness, seed):
    rng = np.random.default_rng(1000 + seed)
    ents = synth.make_entries(rng, int(rng.integers(1, 4)), int(rng.integers(60, 500)),
                              n_runs=int(rng.integers(0, 3)), repeats=(seed % 2 == 0), short=(seed % 2 == 1))
    L = int(rng.integers(12, 25))
    pats = synth.make_patterns(rng, ents, int(rng.integers(4, 30)), length=L, planted=0.7,
                               minlen=(L - 4 if seed % 2 else None), indel_frac=0.5)
    table = synth.table_for(ents)
    raw = synth.stream(ents)
    codes = synth.normalize(raw, table)
    allp = pats + [synth.revcomp(p) for p in pats]
    for norm in (True, False):
        text = O.Text(codes, table) if norm else O.Text(np.frombuffer(raw, dtype=np.uint8))
        for sel, k, ind in CONFIGS:
            ref = refrun.run_ref(ref_harness, codes if norm else raw, pats, table=table if norm else None,
                                 sel=sel, k=k, indels=bool(ind), rc=True, minka=int(rng.integers(1, 50)))
            eng = O.pick_engine(text, allp, k, bool(ind)) if sel == 0 else sel
            got = O.sorted_tuples(O.find_all(text, allp, engine=eng, k=k, indels=bool(ind)))
            assert got == ref, (seed, norm, sel, k, ind)


@pytest.mark.parametrize("seed", range(2))
def test_fuzz_constraints_against_reference(ref_harness, seed):
    """exact_start_bases / exact_end_bases (-s/-e): exact_bases and the constrained verifies."""
    import os, subprocess, tempfile
    rng = np.random.default_rng(500 + seed)
    ents = synth.make_entries(rng, 3, int(rng.integers(200, 1200)), n_runs=2, repeats=(seed % 2 == 0))
    L = int(rng.integers(16, 24))
    pats = synth.make_patterns(rng, ents, int(rng.integers(5, 30)), length=L, planted=0.8, indel_frac=0.5)
    table = synth.table_for(ents)
    codes = synth.normalize(synth.stream(ents), table)
    allp = pats + [synth.revcomp(p) for p in pats]
    text = O.Text(codes, table)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "db.sqn"), "wb").write(codes.tobytes())
        open(os.path.join(d, "db.tbl"), "wb").write(table)
        open(os.path.join(d, "pat.txt"), "w").write("\n".join(pats) + "\n")
        for esb, eeb in [(8, 0), (0, 7), (6, 9), (3, 0)]:
            for sel, k, ind in [(0, 1, 1), (0, 2, 1), (0, 2, 0), (8, 1, 1), (10, 2, 0), (5, 2, 1), (12, 1, 1)]:
                cmd = [ref_harness, "-N", str(sel), "-n", "-r", "-i", os.path.join(d, "db"), "-P", os.path.join(d, "pat.txt"),
                       "-s", str(esb), "-e", str(eeb), "-k" if ind else "-K", str(k)]
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                assert out.returncode == 0, out.stderr[-300:]
                ref = sorted(tuple(int(x) for x in l.split()) for l in out.stdout.splitlines() if not l.startswith("#"))
                E, F = [esb] * len(allp), [eeb] * len(allp)
                eng = sel if sel else O.pick_engine(text, allp, k, bool(ind), E, F)
                got = O.sorted_tuples(O.find_all(text, allp, engine=eng, k=k, indels=bool(ind), esb=E, eeb=F))
                assert got == ref, (seed, esb, eeb, sel, k, ind)


@pytest.mark.parametrize("seed", range(2))
def test_fuzz_wildcards_against_reference(ref_harness, seed):
    """-w / -W: shift_and with IUPAC pattern classes (shift_and.cc:108-117)."""
    import os, subprocess, tempfile
    rng = np.random.default_rng(700 + seed)
    ents = synth.make_entries(rng, 3, int(rng.integers(300, 2000)), n_runs=4, repeats=(seed % 2 == 0))
    L = int(rng.integers(8, 14))
    pats = []
    for p in synth.make_patterns(rng, ents, int(rng.integers(5, 40)), length=L, planted=0.9, indel_frac=0.0, extras=False):
        p = list(p)
        for _ in range(int(rng.integers(0, 3))):
            p[int(rng.integers(0, len(p)))] = str(rng.choice(list("RYKMSWBDHVN")))
        pats.append("".join(p))
    table = synth.table_for(ents)
    codes = synth.normalize(synth.stream(ents), table)
    text = O.Text(codes, table)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "db.sqn"), "wb").write(codes.tobytes())
        open(os.path.join(d, "db.tbl"), "wb").write(table)
        open(os.path.join(d, "pat.txt"), "w").write("\n".join(pats) + "\n")
        for flag, tn in (("-w", False), ("-W", True)):
            out = subprocess.run([ref_harness, "-N", "4", flag, "-n", "-i", os.path.join(d, "db"), "-P", os.path.join(d, "pat.txt")],
                                 capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stderr[-300:]
            ref = sorted(tuple(int(x) for x in l.split()) for l in out.stdout.splitlines() if not l.startswith("#"))
            got = O.sorted_tuples(O.find_all(text, pats, engine=4, k=0, wildcards=True, text_n=tn))
            assert got == ref, (seed, flag)


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_wildcards_inexact_against_reference(ref_harness, seed):
    """-w / -W with k > 0: wildcard-equal cells in editdist_alignment (pattern_alignment.cc:317-319,
    461-475, 514-590) and global_align (primer_alignment.cc:151-154, 253-280); engines as
    pick_pattern_index chooses them (exact_halves over shift_and for k = 1, filter_bitvec for k = 2)
    and forced."""
    import os, subprocess, tempfile
    rng = np.random.default_rng(900 + seed)
    ents = synth.make_entries(rng, 3, int(rng.integers(400, 2500)), n_runs=4, repeats=(seed % 2 == 0))
    L = int(rng.integers(14, 22))
    pats = []
    for p in synth.make_patterns(rng, ents, int(rng.integers(8, 40)), length=L, planted=0.9, indel_frac=0.3, extras=False):
        p = list(p)
        for _ in range(int(rng.integers(0, 3))):
            p[int(rng.integers(0, len(p)))] = str(rng.choice(list("RYKMSWBDHVN")))
        pats.append("".join(p))
    table = synth.table_for(ents)
    codes = synth.normalize(synth.stream(ents), table)
    text = O.Text(codes, table)
    total = 0
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "db.sqn"), "wb").write(codes.tobytes())
        open(os.path.join(d, "db.tbl"), "wb").write(table)
        open(os.path.join(d, "pat.txt"), "w").write("\n".join(pats) + "\n")
        for flag, tn in (("-w", False), ("-W", True)):
            for k in (1, 2):
                for indels in (True, False):
                    for sel in (0, 5, 14, 100):
                        cmd = [ref_harness, "-N", str(sel), flag, "-k" if indels else "-K", str(k), "-n", "-i", os.path.join(d, "db"),
                               "-P", os.path.join(d, "pat.txt")]
                        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                        assert out.returncode == 0, (cmd, out.stderr[-300:])
                        ref = sorted(tuple(int(x) for x in l.split()) for l in out.stdout.splitlines() if not l.startswith("#"))
                        got = O.sorted_tuples(O.find_all(text, pats, engine=sel, k=k, indels=indels, wildcards=True, text_n=tn))
                        assert got == ref, (seed, flag, k, indels, sel, len(got), len(ref))
                        total += len(ref)
    assert total > 0


@pytest.mark.parametrize("norm", [True, False])
def test_stream_edges_against_reference(ref_harness, norm):
    """The quirks at the ends of the stream that round 3's GPU fixes rest on (tests/test_gpu_parity.py:
    test_extensions_read_past_the_end_of_the_stream, test_exact_bases_at_both_ends_of_the_stream,
    test_edit_plan_matches_that_end_with_the_stream): the reference's extensions read the mapped file's zero padding
    past the last character (mapFile.h:49-57) and report hits that end beyond the stream; a pattern whose first
    characters are deleted matches at the very start.  The oracle must say what the REAL reference says there."""
    import os, subprocess, tempfile
    table = b"ACGT\n"
    rng = np.random.default_rng(90)
    rnd = lambda n: "".join("ACGT"[c] for c in rng.integers(0, 4, n))
    body = rnd(1500)
    tail = "GATTACAGGCTTACCGTCAATGCCTGAAGTCCATGTTGCA"
    raw = ("\n" + body + tail).encode()
    pats = []
    for L in (14, 20, 21, 24, 32):
        len2 = L - L // 2
        for t in range(1, len2 + 2):
            p = tail[len(tail) - (L - t):] + "A" * t                 # hangs over the end by t characters
            pats += [p, p[:-1] + "C", "ACGT"[("ACGT".index(p[0]) + 2) % 4] + p[1:]]
            if t >= 2:
                pats.append(p[:-2] + "CG")
        head = body[:L]
        pats += [head[1:] + "A", head[2:] + "AC", head[:3] + head[4:] + "G"]       # first characters deleted at the start
        site = (body + tail)[-L:]
        pats += [site[:L - 3] + "C" + site[L - 3:], site[:L - 1] + "GG"]            # deleted pattern characters at the very end
    pats = list(dict.fromkeys(pats))
    data = synth.normalize(raw, table) if norm else np.frombuffer(raw, dtype=np.uint8)
    text = O.Text(data, table) if norm else O.Text(data)
    beyond = 0
    for sel, k, ind in [(12, 0, 0), (12, 1, 0), (12, 2, 0), (12, 1, 1), (12, 2, 1), (100, 2, 1), (100, 2, 0), (5, 2, 1), (5, 1, 1)]:
        use = [p for p in pats if len(p) >= 16] if (ind and sel == 12) else pats
        ref = refrun.run_ref(ref_harness, data, use, table=table if norm else None, sel=sel, k=k, indels=bool(ind))
        got = O.sorted_tuples(O.find_all(text, use, engine=sel, k=k, indels=bool(ind)))
        assert got == ref, (norm, sel, k, ind, sorted(set(ref) - set(got))[:4], sorted(set(got) - set(ref))[:4])
        beyond += len([h for h in ref if h[0] > len(raw)])
    assert beyond > (100 if norm else 20)                      # (on a raw stream every character past the end is a substitution)
    # exact_bases (-s / -e), prefix and suffix blocks
    use = [p for p in pats if len(p) >= 20]
    with tempfile.TemporaryDirectory() as d:
        if norm:
            open(os.path.join(d, "db.sqn"), "wb").write(data.tobytes())
            open(os.path.join(d, "db.tbl"), "wb").write(table)
        else:
            open(os.path.join(d, "db"), "wb").write(raw)
        open(os.path.join(d, "pat.txt"), "w").write("\n".join(use) + "\n")
        for esb, eeb in [(8, 0), (6, 3), (0, 7)]:
            for k, ind in [(1, 0), (2, 0), (1, 1), (2, 1)]:
                cmd = [ref_harness, "-N", "8", "-i", os.path.join(d, "db"), "-P", os.path.join(d, "pat.txt"), "-s", str(esb), "-e", str(eeb),
                       "-k" if ind else "-K", str(k)] + (["-n"] if norm else [])
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                assert out.returncode == 0, out.stderr[-300:]
                ref = sorted(tuple(int(x) for x in l.split()) for l in out.stdout.splitlines() if not l.startswith("#"))
                E, F = [esb] * len(use), [eeb] * len(use)
                got = O.sorted_tuples(O.find_all(text, use, engine=8, k=k, indels=bool(ind), esb=E, eeb=F))
                assert got == ref, (norm, esb, eeb, k, ind, sorted(set(ref) - set(got))[:4], sorted(set(got) - set(ref))[:4])
