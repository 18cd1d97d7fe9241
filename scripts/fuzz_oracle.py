"""CPU, build container only: fuzz the oracle (oracle/pm_oracle.c, test infrastructure) against the REAL reference
engines (oracle/_ref/ref_harness, the reference's sources compiled where they lie) on the adversarial streams of
tests/adversarial.py -- skewed composition, words of a small vocabulary, tandem repeats with drifting copies,
primers cut from the stream and edited -- at sizes the reference finishes in seconds.  The oracle is what the GPU
parity tests compare with; this is what pins it beyond the committed goldens.

    python scripts/fuzz_oracle.py [seconds] [first_seed]

Exit status 1 on the first difference."""
import os
import subprocess
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
if "torch" not in sys.modules:                                     # the generators of fuzz_families do not need it
    sys.modules["torch"] = types.ModuleType("torch")
import adversarial as F  # noqa: E402  (the generators; scripts/fuzz_families.py uses the same)
import refrun  # noqa: E402
from oracle import pmoracle as O  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
# (engine selector of ref_harness, k, indels)
CONFIGS = [(0, 0, 1), (4, 0, 1), (2, 0, 1), (100, 1, 0), (100, 2, 0), (100, 2, 1), (100, 1, 1), (5, 1, 1), (5, 2, 0), (5, 2, 1), (5, 1, 0),
           (12, 0, 0), (12, 1, 0), (12, 2, 0), (12, 1, 1), (14, 1, 1), (12, 2, 1), (0, 1, 1), (0, 1, 0), (0, 2, 1), (0, 2, 0)]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_end = time.time() + budget
    runs = 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        style = int(rng.integers(0, 4))
        n = int(rng.integers(200, 6000))
        s = F.make_stream(rng, n, style)
        lo, hi = (20, 20) if rng.integers(0, 2) else (int(rng.integers(8, 21)), int(rng.integers(21, 33)))
        pats = F.make_patterns(rng, s, int(rng.integers(1, 40)), lo, min(hi, n - 1), 2)
        # FUZZ_MINKA=random: many find_patterns calls of random size, on a stream that ends with an end-of-sequence
        # character as every compress_seq database does (see the note at run_ref below)
        many_calls = os.environ.get("FUZZ_MINKA") == "random"
        if many_calls:
            s = np.concatenate([s, np.array([4], dtype=np.uint8)])
        raw = np.frombuffer(b"ACGT\n", dtype=np.uint8)[s]
        norm = bool(rng.integers(0, 2))
        rc = bool(rng.integers(0, 2))
        allp = pats + [O.reverse_comp(p) for p in pats] if rc else pats
        data, table = (s, b"ACGT\n") if norm else (raw, None)
        text = O.Text(data, table) if norm else O.Text(data)
        for sel, k, ind in CONFIGS:
            try:
                # one find_patterns call: the reference's keyword tree keeps a look-ahead character across calls and drops it --
                # with the hits that end on it -- when a call returns right in front of the stream's last character
                # (keyword_tree.t:434: `if ((eof=cp.eof())) return false;`); which hits are lost depends on where the calls
                # happen to end, so the oracle (and the engines) state the answer of a single call.  A database written by
                # compress_seq ends with an end-of-sequence character: no hit ends there.
                ref = refrun.run_ref(HARNESS, data, pats, table=table, sel=sel, k=k, indels=bool(ind), rc=rc, minka=int(rng.integers(1, 50)) if many_calls else 1000000)
            except RuntimeError as e:                                  # the reference rejects the option set (e.g. k >= pattern length)
                continue
            eng = O.pick_engine(text, allp, k, bool(ind)) if sel == 0 else sel
            got = O.sorted_tuples(O.find_all(text, allp, engine=eng, k=k, indels=bool(ind)))
            runs += 1
            if got != ref:
                print("DIFFERENT seed %d style %d n %d norm %d rc %d engine %d k %d indels %d: reference %d hits, oracle %d" % (seed, style, n, norm, rc, sel, k, ind, len(ref), len(got)))
                print("  only reference:", sorted(set(ref) - set(got))[:6], " only oracle:", sorted(set(got) - set(ref))[:6])
                print("runs %d failures 1" % runs)
                sys.exit(1)
        # exact_start_bases / exact_end_bases (-s / -e, the same for every pattern): exact_bases and the constrained verifies
        esb, eeb = [(8, 0), (0, 7), (6, 9), (3, 0), (7, 6), (0, 0)][int(rng.integers(0, 6))]
        if (esb or eeb) and all(len(p) >= 14 for p in pats):
            with tempfile.TemporaryDirectory() as d:
                if norm:
                    open(os.path.join(d, "db.sqn"), "wb").write(data.tobytes())
                    open(os.path.join(d, "db.tbl"), "wb").write(table)
                else:
                    open(os.path.join(d, "db"), "wb").write(data.tobytes())
                open(os.path.join(d, "pat.txt"), "w").write("\n".join(pats) + "\n")
                for sel, k, ind in [(0, 1, 1), (0, 2, 1), (0, 2, 0), (8, 1, 1), (8, 2, 1), (8, 1, 0), (10, 2, 0), (5, 2, 1), (5, 2, 0), (12, 1, 1), (12, 2, 0)]:
                    cmd = [HARNESS, "-N", str(sel), "-m", str(int(rng.integers(1, 50))) if many_calls else "1000000", "-i", os.path.join(d, "db"), "-P", os.path.join(d, "pat.txt"), "-s", str(esb), "-e", str(eeb),
                           "-k" if ind else "-K", str(k)] + (["-n"] if norm else []) + (["-r"] if rc else [])
                    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    if out.returncode != 0:
                        continue
                    ref = sorted(tuple(int(x) for x in l.split()) for l in out.stdout.splitlines() if not l.startswith("#"))
                    E, Fz = [esb] * len(allp), [eeb] * len(allp)
                    eng = sel if sel else O.pick_engine(text, allp, k, bool(ind), E, Fz)
                    if eng < 0:
                        continue
                    got = O.sorted_tuples(O.find_all(text, allp, engine=eng, k=k, indels=bool(ind), esb=E, eeb=Fz))
                    runs += 1
                    if got != ref:
                        print("DIFFERENT seed %d style %d n %d norm %d rc %d zones %d/%d engine %d k %d indels %d: reference %d hits, oracle %d" % (seed, style, n, norm, rc, esb, eeb, sel, k, ind, len(ref), len(got)))
                        print("  only reference:", sorted(set(ref) - set(got))[:6], " only oracle:", sorted(set(got) - set(ref))[:6])
                        print("runs %d failures 1" % runs)
                        sys.exit(1)
        # IUPAC pattern classes (-w / -W) with N in the stream and in the primers: the automaton's masks leave a stream N out
        # of every class without -W (shift_and.cc:112) while the verify DPs take equal characters first
        # (pattern_alignment.cc:314-319) -- the pair on which engines and wrappers disagree, so the oracle must have it right
        if rng.integers(0, 2) == 0:
            sn = s.copy()
            sn[rng.integers(0, sn.size, int(rng.integers(1, 40)))] = 5
            wp = []
            for q in allp:
                q = list(q)
                for _ in range(int(rng.integers(0, 3))):
                    i = int(rng.integers(0, len(q)))
                    if q[i] in F.IUPAC:
                        q[i] = F.IUPAC[q[i]][int(rng.integers(0, 7))]
                wp.append("".join(q))
            datan, tablen = (sn, b"ACGT\nN") if norm else (np.frombuffer(b"ACGT\nN", dtype=np.uint8)[sn], None)
            textn = O.Text(datan, tablen)
            with tempfile.TemporaryDirectory() as d:
                if norm:
                    open(os.path.join(d, "db.sqn"), "wb").write(datan.tobytes())
                    open(os.path.join(d, "db.tbl"), "wb").write(tablen)
                else:
                    open(os.path.join(d, "db"), "wb").write(datan.tobytes())
                open(os.path.join(d, "pat.txt"), "w").write("\n".join(wp) + "\n")
                for sel, k, ind in [(0, 0, 1), (4, 0, 1), (0, 1, 1), (0, 1, 0), (0, 2, 1), (0, 2, 0), (5, 1, 0), (5, 2, 0), (5, 2, 1), (14, 1, 1), (14, 2, 0), (100, 2, 0), (100, 1, 1)]:
                    for flag, tn in (("-w", False), ("-W", True)):
                        cmd = [HARNESS, "-N", str(sel), "-m", "1000000", "-i", os.path.join(d, "db"), "-P", os.path.join(d, "pat.txt"), flag] + \
                              (["-k" if ind else "-K", str(k)] if k else []) + (["-n"] if norm else [])
                        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                        if out.returncode != 0:
                            continue
                        ref = sorted(tuple(int(x) for x in l.split()) for l in out.stdout.splitlines() if not l.startswith("#"))
                        try:
                            got = O.sorted_tuples(O.find_all(textn, wp, engine=sel, k=k, indels=bool(ind), wildcards=True, text_n=tn))
                        except RuntimeError:
                            continue
                        runs += 1
                        if got != ref:
                            print("DIFFERENT seed %d style %d n %d norm %d %s engine %d k %d indels %d: reference %d hits, oracle %d" % (seed, style, n, norm, flag, sel, k, ind, len(ref), len(got)))
                            print("  only reference:", sorted(set(ref) - set(got))[:6], " only oracle:", sorted(set(got) - set(ref))[:6])
                            print("runs %d failures 1" % runs)
                            sys.exit(1)
        if seed % 20 == 0:
            print("seed %d ok (%d engine runs so far)" % (seed, runs), flush=True)
        seed += 1
    print("runs %d failures 0" % runs)


if __name__ == "__main__":
    main()
