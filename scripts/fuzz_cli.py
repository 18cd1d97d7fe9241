"""Differential fuzz of the command lines on the GPU box: pm_primer_match / pm_pcr_match (MI355X engine) against the
real reference primer_match / pcr_match (oracle/_ref, built by oracle/Makefile in the build container; the binaries
travel with the snapshot) on the same seeded databases, primer files and option sets that tests/golden/make_*_golden.py
use for the committed fixtures -- with fresh seeds, so every round of this script is a new pair of fixtures.

    python scripts/fuzz_cli.py [seconds] [first_seed]

Output is compared as sorted lines (engines report the hits of one position in different orders; the reference's own
testscript.sh sorts before cmp).  Exit status 1 on the first difference."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_cli_golden as CLI  # noqa: E402
import make_pcr_golden as PCR  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
HOST = os.path.join(ROOT, "sequence-alignment-tools_amd", "host")


def run(cmd):
    return subprocess.run(cmd, capture_output=True, timeout=600)


def differ(name, a, b):
    la, lb = sorted(a.decode("latin1").splitlines()), sorted(b.decode("latin1").splitlines())
    if la == lb:
        return False
    sa, sb = set(la), set(lb)
    print("DIFFERENT %s: reference %d lines, ours %d" % (name, len(la), len(lb)))
    print("  only reference:", sorted(sa - sb)[:4])
    print("  only ours:", sorted(sb - sa)[:4])
    return True


def primer_round(seed, d):
    fasta, ptxt, pfa, psts, pats, wtxt = CLI.build_inputs(seed)
    n = 0
    for variant, args in (("normalized", ["-n", "true"]), ("indexed", []), ("compressed", ["-z", "true"])):
        sub = os.path.join(d, variant)
        os.mkdir(sub)
        fa = os.path.join(sub, "db.fa")
        with open(fa, "w") as f:
            f.write(fasta)
        r = run([os.path.join(HOST, "pm_compress_seq"), "-i", fa] + args)
        assert r.returncode == 0, r.stderr
    for src, text in (("P", ptxt), ("F", pfa), ("S", psts), ("W", wtxt)):
        with open(os.path.join(d, "primers." + src), "w") as f:
            f.write(text)
    for cname, src, extra in CLI.CASES:
        for variant in ("normalized", "indexed", "compressed"):
            fa = os.path.join(d, variant, "db.fa")
            parg = ["-p", " ".join(pats[:5])] if src == "p" else ["-" + ("P" if src == "W" else src), os.path.join(d, "primers." + src)]
            ref = run([os.path.join(REF, "primer_match"), "-i", fa] + parg + extra)
            if ref.returncode != 0:
                continue
            ours = run([os.path.join(HOST, "pm_primer_match"), "-i", fa] + parg + extra)
            if ours.returncode != 0:
                print("DIFFERENT primer_match seed %d %s %s: ours failed: %s" % (seed, cname, variant, ours.stderr[-300:]))
                return -1
            if differ("primer_match seed %d %s %s" % (seed, cname, variant), ref.stdout, ours.stdout):
                return -1
            n += 1
            # the reference's own main() on the GPU engine (oracle/_ref/primer_match_gpu: its primer_match.o linked with
            # class gpu_pattern_match, host/plugin/), on the uncompressed database forms
            plug = os.path.join(REF, "primer_match_gpu")
            if os.path.exists(plug) and variant != "compressed" and "-N" not in extra:
                mine = run([plug, "-i", fa] + parg + extra + ["-N", "17"])
                if mine.returncode != 0:
                    print("DIFFERENT primer_match_gpu seed %d %s %s: failed: %s" % (seed, cname, variant, mine.stderr[-300:]))
                    return -1
                if differ("primer_match_gpu seed %d %s %s" % (seed, cname, variant), ref.stdout, mine.stdout):
                    return -1
                n += 1
    return n


def pcr_round(seed, d):
    fasta, sts, ptxt, qtxt, pfa = PCR.build_inputs(seed)
    fa = os.path.join(d, "db.fa")
    with open(fa, "w") as f:
        f.write(fasta)
    r = run([os.path.join(HOST, "pm_compress_seq"), "-i", fa, "-n", "true"])
    assert r.returncode == 0, r.stderr
    for k, text in {"S": sts, "P": ptxt, "Q": qtxt, "F": pfa}.items():
        with open(os.path.join(d, "primers." + k), "w") as f:
            f.write(text)
    n = 0
    for cname, src, extra in PCR.CASES:
        flag = {"S": "-S", "P": "-P", "Q": "-P", "F": "-F"}[src]
        ref = run([os.path.join(REF, "pcr_match"), "-i", fa, flag, os.path.join(d, "primers." + src)] + extra)
        if ref.returncode != 0:
            continue
        for more in ([], ["--ranks", "2"]):
            ours = run([os.path.join(HOST, "pm_pcr_match"), "-i", fa, flag, os.path.join(d, "primers." + src)] + extra + more)
            if ours.returncode != 0:
                print("DIFFERENT pcr_match seed %d %s %s: ours failed: %s" % (seed, cname, more, ours.stderr[-300:]))
                return -1
            if differ("pcr_match seed %d %s %s" % (seed, cname, more), ref.stdout, ours.stdout):
                return -1
            n += 1
        plug = os.path.join(REF, "pcr_match_gpu")
        if os.path.exists(plug):
            mine = run([plug, "-i", fa, flag, os.path.join(d, "primers." + src)] + extra + ["-N", "17"])
            if mine.returncode != 0:
                print("DIFFERENT pcr_match_gpu seed %d %s: failed: %s" % (seed, cname, mine.stderr[-300:]))
                return -1
            if differ("pcr_match_gpu seed %d %s" % (seed, cname), ref.stdout, mine.stdout):
                return -1
            n += 1
    return n


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    t_end = time.time() + budget
    total = 0
    while time.time() < t_end:
        for fn in (primer_round, pcr_round):
            with tempfile.TemporaryDirectory() as d:
                try:
                    n = fn(seed, d)
                except (AssertionError, IndexError, ValueError) as e:        # the generator could not build this seed's inputs
                    print("seed %d %s: inputs not built (%s)" % (seed, fn.__name__, str(e)[:80]))
                    n = 0
            if n < 0:
                print("command lines compared %d, failures 1" % total)
                sys.exit(1)
            total += n
        print("seed %d ok (%d command lines so far)" % (seed, total), flush=True)
        seed += 1
    print("command lines compared %d, failures 0" % total)


if __name__ == "__main__":
    main()
