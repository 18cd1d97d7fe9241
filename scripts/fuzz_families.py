"""Long differential fuzz on the GPU box, two forms.

Default: the seed family (pair plan, edit plan, halves) against the bit-parallel family on streams made to be hard for
the seed kernels (tests/adversarial.py: low-complexity text built from a small vocabulary of words, tandem repeats,
primers cut from the stream and mutated), streams of 64 Ki .. 2 Mi characters and one case in twelve of 20 .. 400 Mbp.
The two families share no code before the final stage -- and everything from there on, so a bug in pm_finalize, the
host cluster / halves rules or the stream-edge records is invisible to this form.

--oracle: the library (its own choice of kernel family) against oracle/pm_oracle.c, the CPU restatement pinned to the real
reference, on the small cases of tests/adversarial.py (300 .. 8000 characters: what the oracle finishes in well under a
second) -- the open-ended form of tests/test_gpu_adversarial.py, which keeps 600 fixed seeds in the suite.

    python scripts/fuzz_families.py [seconds] [first_seed] [--oracle [--dense-bound RECORDS | --edge-cuts]]

Prints one line per case and a summary; exit status 1 on the first difference."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sat_amd  # noqa: E402
import adversarial as A  # noqa: E402
from adversarial import make_stream, make_patterns  # noqa: E402

TABLE = b"ACGT\n"
LUT = np.frombuffer(b"ACGT", dtype=np.uint8)


def sort3(end, pid, k):
    o = np.lexsort((k, pid, end))
    return end[o].astype(np.int64), pid[o].astype(np.int64), k[o].astype(np.int64)


def hits_of(pats, k, indels, kernel, sem, dev, cap, mode=0, rng=None, table=TABLE, zones=None, wild=False, host=None):
    """mode 0: find_all over the whole stream; 1: find_all in small chunks (resumable scans); 2: one scan + the
    device finalize (bench.py's single-rank path); 3: two position shards, each finalized on its own with a guard
    band (bench.py's multi-rank path).  Raises PmError(-2) where the library says the mode does not apply."""
    pm = sat_amd.PatternMatch(k=k, indels=indels, kernel=kernel, semantics=sem, wildcards=wild)
    for i, p in enumerate(pats):
        if zones is None:
            pm.add_pattern(p, i + 1)
        else:
            pm.add_pattern(p, i + 1, zones[i][0], zones[i][1])
    if host is not None:
        pm.init(host, table)                                           # the stream handed over in host memory (pm_init uploads it)
    else:
        pm.init_device(dev.data_ptr(), dev.numel(), table, keepalive=dev)
    pm.set_capacity(cap)
    n = dev.numel()
    try:
        if mode == 0:
            h = pm.find_all()
        elif mode == 1:
            h = pm.find_all(chunk=int(rng.integers(2000, 300000)))
        elif mode == 2:
            pm.reset()
            pm.scan_candidates(0, n, to_host=False)
            h = pm.finalize_device(n, last=True, sort=True)
        else:
            guard = 4096
            cut = int(rng.integers(n // 4, 3 * n // 4))
            parts = []
            for own_lo, own_hi in ((0, cut), (cut, n)):
                g_lo, g_hi = max(0, own_lo - guard), min(n, own_hi + guard)
                pm.reset()
                pm.scan_candidates(g_lo, g_hi, to_host=False)
                parts.append(pm.finalize_device(0, sort=True, owned=(own_lo, own_hi, g_lo, None if g_hi == n else g_hi)).copy())
            h = np.concatenate(parts)
        sel = pm.selected()
        desc = pm.describe()
    finally:
        pm.close()
    return sort3(h["end"], h["pid"], h["k"]) + (sel, desc)


def main_oracle(budget, seed):
    """small adversarial cases, library against oracle"""
    t_end = time.time() + budget
    cases = bad = skipped = 0
    fam = {sat_amd.KERNEL_SEED: 0, sat_amd.KERNEL_BITPAR: 0}
    while time.time() < t_end:
        c = A.small_case(seed)
        want = A.oracle_hits(c)
        note = ""
        try:
            if DENSE_BOUND:
                # pm_scan's piece-wise form (pm_api.cpp scan_range): every case through pm_scan -- three ranges handed out as
                # spans, or the case's small chunks -- with a bound so low that ranges are scanned in pieces
                for bound in (DENSE_BOUND, 16 * DENSE_BOUND, 1 << 30):
                    os.environ["PM_DENSE_BOUND"] = str(bound)
                    try:
                        st = {}
                        got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, mode=0, stats=st) if c["mode"] % 2 == 0 else A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, mode=1)
                        CUT[0] += st.get("range_splits", 0) > 0
                        break
                    except sat_amd.PmError as e2:                      # more records in 256 positions than the bound: not what is tested here
                        if e2.code != -2 or "smaller ranges" not in str(e2) or bound == 1 << 30:
                            raise
            elif EDGE_CUTS:
                # pm_scan in ranges of 1 .. 30 positions at both ends of the stream (the host-made records of the stream edges
                # belong to whatever range holds their end) and a few random cuts in between
                n = c["n"]
                r = np.random.default_rng(seed)
                cuts = list(np.cumsum(r.integers(1, 12, size=8))) + [n - int(x) for x in np.cumsum(r.integers(1, 12, size=8))] + [int(x) for x in r.integers(0, n + 1, size=3)]
                got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, cuts=cuts)
            else:
                got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO)
        except sat_amd.PmError as err:
            if want is None and err.code in (-6, -2):
                skipped += 1
                seed += 1
                continue
            if err.code != -2 or c["mode"] < 2:
                raise
            note = " (mode %d not available: %s; resumable scan instead)" % (c["mode"], str(err)[:60])
            got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, mode=1)
        cases += 1
        fam[c["selected"][1]] += 1
        same = want is not None and got == want
        print("%s: %d hits %s%s  %s" % (A.describe(c), len(got), "ok" if same else "DIFFERENT (oracle %s)" % (None if want is None else len(want)), note, c["kernel_desc"][:50]), flush=True)
        if not same:
            bad += 1
            sg, sw = set(got), set(want or [])
            print("  only GPU:", sorted(sg - sw)[:8], " only oracle:", sorted(sw - sg)[:8], flush=True)
            break
        seed += 1
    print("oracle form: cases %d (seed family %d, bit-parallel family %d), rejected by reference and library alike %d, failures %d, next seed %d" % (
        cases, fam[sat_amd.KERNEL_SEED], fam[sat_amd.KERNEL_BITPAR], skipped, bad, seed))
    if DENSE_BOUND:
        print("with PM_DENSE_BOUND=%d: three-range cases that were scanned in pieces: %d" % (DENSE_BOUND, CUT[0]))
    sys.exit(1 if bad else 0)


DENSE_BOUND = 0
EDGE_CUTS = False
CUT = [0]


def main():
    global DENSE_BOUND, EDGE_CUTS
    EDGE_CUTS = "--edge-cuts" in sys.argv[1:]
    argv = [a for a in sys.argv[1:] if a not in ("--oracle", "--edge-cuts")]
    if "--dense-bound" in argv:
        i = argv.index("--dense-bound")
        DENSE_BOUND = int(argv[i + 1])
        del argv[i:i + 2]
    budget = float(argv[0]) if len(argv) > 0 else 300.0
    seed = int(argv[1]) if len(argv) > 1 else 1000
    if "--oracle" in sys.argv[1:]:
        return main_oracle(budget, seed)
    t_end = time.time() + budget
    cases = bad = 0
    sems = [(sat_amd.SEM_AUTO, "auto"), (sat_amd.SEM_SHIFT_AND_INEXACT, "sai"), (sat_amd.SEM_FILTER_BITVEC, "fbv"), (sat_amd.SEM_EXACT_HALVES, "halves"),
            (sat_amd.SEM_EXACT_BASES, "bases")]
    IUPAC = {"A": "RWMDHVN", "C": "YSMBHVN", "G": "RSKBDVN", "T": "YWKBDHN"}
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        style = int(rng.integers(0, 4))
        n = int(rng.integers(1 << 16, 1 << 21))
        # one case in twelve: a long stream (the launch geometry large ranges get, long stretches without a key hit)
        big = bool(rng.integers(0, 12) == 0)
        if big:
            n = int(rng.integers(20_000_000, 400_000_000))
            g = torch.Generator(device="cuda")
            g.manual_seed(seed)
            dev_big = torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda", generator=g)
            for _ in range(int(rng.integers(0, 4))):
                dev_big[int(rng.integers(0, n))] = 4
            s = dev_big[: 1 << 20].cpu().numpy()                      # the primers are cut from the first Mi characters
            style = 0
        else:
            s = make_stream(rng, n, style)
        k = int(rng.integers(0, 3))
        indels = bool(rng.integers(0, 2)) and k > 0
        only20 = bool(rng.integers(0, 2))
        lo, hi = (20, 20) if only20 else (int(rng.integers(12, 21)), int(rng.integers(21, 33)))
        count = int(rng.integers(1, 300 if big else 1500))
        pats = make_patterns(rng, s, count, lo, hi, k)
        allp = pats + [sat_amd.reverse_comp(p) for p in pats]
        sem, sname = sems[int(rng.integers(0, len(sems)))]
        # exact zones (exact_start_bases / exact_end_bases) on a third of the cases, always for exact_bases
        zones = None
        if sem == sat_amd.SEM_EXACT_BASES or rng.integers(0, 3) == 0:
            zones = []
            for p in allp:
                e, f_ = (int(rng.integers(0, 9)), 0) if rng.integers(0, 2) else (0, int(rng.integers(0, 9)))
                if rng.integers(0, 4) == 0:
                    e, f_ = int(rng.integers(0, 7)), int(rng.integers(0, 7))
                if sem == sat_amd.SEM_EXACT_BASES and max(e, f_) < 6:
                    e = 6 + int(rng.integers(0, 4))
                zones.append((min(e, len(p)), min(f_, len(p))))
        # ambiguity codes in the primers (-w) on a sixth of the cases
        wild = bool(rng.integers(0, 6) == 0)
        if wild:
            wp = []
            for p in allp:
                q = list(p)
                for _ in range(int(rng.integers(0, 3))):
                    i = int(rng.integers(0, len(q)))
                    if q[i] in IUPAC:
                        q[i] = IUPAC[q[i]][int(rng.integers(0, 7))]
                wp.append("".join(q))
            allp = wp
        # a few characters that are no base (N) in the stream on a quarter of the cases
        with_n = bool(rng.integers(0, 4) == 0)
        for name, val in (("PM_SEED_CHUNK", [None, "16384", "65536", "524288"]), ("PM_PAIR_ROW", [None, None, "3", "6"]), ("PM_SEED_GROUP", [None, "1", "3"]),
                          ("PM_SEED_TILE", [None, None, "300", "1000"])):
            v = val[int(rng.integers(0, len(val)))]
            if v is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = v
        cap = [1 << 12, 1 << 18, 1 << 24][int(rng.integers(0, 3))]
        mode = int(rng.integers(0, 4))
        host = None
        raw = bool(rng.integers(0, 4) == 0)                            # the stream as bytes 'A','C','G','T','\n' with no table
        table = None if raw else (b"ACGT\nN" if with_n else TABLE)
        if big:
            with_n = False
            table = None if raw else TABLE
            dev = torch.from_numpy(np.frombuffer(b"ACGT\nN", dtype=np.uint8).copy()).cuda()[dev_big.long()] if raw else dev_big
            del dev_big
        else:
            if with_n:
                s = s.copy()
                s[rng.integers(0, n, int(rng.integers(1, 200)))] = 5
            if not raw and rng.integers(0, 8) == 0:                    # the end-of-entry character first in the table (code 0)
                table = b"\nACGT" + (b"N" if with_n else b"")
                s = np.array([1, 2, 3, 4, 0, 5], dtype=np.uint8)[s]
            dev = torch.from_numpy(np.frombuffer(b"ACGT\nN", dtype=np.uint8)[s] if raw else s).cuda()
            if rng.integers(0, 4) == 0:
                host = dev.cpu().numpy()
        t0 = time.time()
        try:
            b = hits_of(allp, k, indels, sat_amd.KERNEL_BITPAR, sem, dev, 1 << 24, 0, rng, table, zones, wild, host)
        except sat_amd.PmError as err:                                 # an option set the reference rejects as well (e.g. k too large for the zones)
            print("seed %d skipped: the bit-parallel family says %s" % (seed, str(err)[:100]), flush=True)
            seed += 1
            continue
        try:
            a = hits_of(allp, k, indels, sat_amd.KERNEL_SEED, sem, dev, cap, mode, rng, table, zones, wild, host)
        except sat_amd.PmError as err:
            if err.code == -2 or "chain" in str(err):                  # option set / mode the library does not cover: said loudly
                print("seed %d mode %d skipped (%s)" % (seed, mode, str(err)[:100]), flush=True)
                seed += 1
                continue
            raise
        same = a[0].size == b[0].size and (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
        cases += 1
        print("seed %d mode %d%s%s%s%s style %d n %d k %d indels %d L %d..%d patterns %d sem %s cap %d env %s: %d hits %s  %.1f s  %s" % (
            seed, mode, " raw" if raw else "", " zones" if zones else "", " wild" if wild else "", (" N" if with_n else "") + (" host" if host is not None else "") + (" eos0" if table and table[:1] == b"\n" else ""), style + (10 if big else 0), n, k, indels, lo, hi, 2 * count, sname, cap,
            ",".join("%s=%s" % (e[3:], os.environ[e]) for e in ("PM_SEED_CHUNK", "PM_PAIR_ROW", "PM_SEED_GROUP", "PM_SEED_TILE") if e in os.environ),
            a[0].size, "ok" if same else "DIFFERENT (bitpar %d)" % b[0].size, time.time() - t0, a[4][:60]), flush=True)
        if not same:
            bad += 1
            sa = set(zip(a[0].tolist(), a[1].tolist(), a[2].tolist()))
            sb = set(zip(b[0].tolist(), b[1].tolist(), b[2].tolist()))
            print("  only seed family:", sorted(sa - sb)[:8], " only bitpar:", sorted(sb - sa)[:8], flush=True)
            break
        seed += 1
    print("cases %d failures %d" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
