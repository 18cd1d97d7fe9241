"""Streams and pattern sets made to be hard for the seed kernels -- skewed composition, words of a small vocabulary,
tandem repeats with drifting copies, primers cut from the stream and edited -- so that most windows are key hits, lanes
run out of pending-hit bits, suspect queues flush all the time, record buffers overflow and same-pattern chains of
candidates grow long (filter_bitvec.cc:88-177, exact_halves.cc:120-197, exact_bases.cc:69-129,
shift_and_inexact.cc:249-352 are the rules that then matter).  Shared by tests/test_gpu_adversarial.py (fixed seeds,
GPU against the ORACLE), scripts/fuzz_families.py (open-ended runs on a GPU box) and bench.py --stream-style."""
import os

import numpy as np

import sat_amd

TABLE = b"ACGT\n"
LUT = np.frombuffer(b"ACGT", dtype=np.uint8)
STYLES = ("uniform", "skew", "vocab", "tandem")
IUPAC = {"A": "RWMDHVN", "C": "YSMBHVN", "G": "RSKBDVN", "T": "YWKBDHN"}
KNOBS = (("PM_SEED_CHUNK", [None, "16384", "65536", "524288"]), ("PM_PAIR_ROW", [None, None, "3", "6"]), ("PM_SEED_GROUP", [None, "1", "3"]),
         ("PM_SEED_TILE", [None, None, "300", "1000"]))


def make_stream(rng, n, style):
    """codes 0..3 (A,C,G,T), code 4 = end of an entry"""
    if style == 0:                                                 # uniform
        s = rng.integers(0, 4, n, dtype=np.uint8)
    elif style == 1:                                               # skewed composition
        s = rng.choice(4, size=n, p=[0.55, 0.05, 0.05, 0.35]).astype(np.uint8)
    elif style == 2:                                               # words of a small vocabulary, 2 % point mutations
        wl = int(rng.integers(5, 13))
        vocab = rng.integers(0, 4, (int(rng.integers(4, 200)), wl), dtype=np.uint8)
        s = vocab[rng.integers(0, vocab.shape[0], n // wl + 1)].reshape(-1)[:n].copy()
        m = rng.random(n) < 0.02
        s[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
    else:                                                          # tandem repeats of short units with drifting copies
        s = np.empty(n, dtype=np.uint8)
        at = 0
        while at < n:
            unit = rng.integers(0, 4, int(rng.integers(1, 40)), dtype=np.uint8)
            reps = int(rng.integers(1, 400))
            blk = np.tile(unit, reps)
            m = rng.random(blk.size) < 0.03
            blk[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
            blk = blk[: n - at]
            s[at:at + blk.size] = blk
            at += blk.size
    for _ in range(int(rng.integers(0, 4))):                       # entry ends
        s[int(rng.integers(0, n))] = 4
    return s


def make_patterns(rng, s, count, lo, hi, k):
    """primers cut from the stream, with 0 .. k+1 random edits each"""
    out = []
    n = s.size
    tries = 0
    while len(out) < count:
        tries += 1
        L = int(rng.integers(lo, hi + 1))
        a = int(rng.integers(0, max(1, n - L)))
        w = s[a:a + L]
        if w.size < L or (w > 3).any():
            if tries > 50 * count + 1000:                          # (a stream with hardly a clean window: random primers)
                out.append("".join("ACGT"[c] for c in rng.integers(0, 4, L)))
            continue
        p = LUT[w].tobytes().decode()
        for _ in range(int(rng.integers(0, k + 2))):               # 0 .. k+1 edits
            kind = int(rng.integers(0, 4))
            i = int(rng.integers(0, len(p)))
            if kind <= 1:
                p = p[:i] + "ACGT"[int(rng.integers(0, 4))] + p[i + 1:]
            elif kind == 2 and len(p) < hi:
                p = p[:i] + "ACGT"[int(rng.integers(0, 4))] + p[i:]
            elif kind == 3 and len(p) > lo:
                p = p[:i] + p[i + 1:]
        out.append(p)
    return out


SEMS = [(sat_amd.SEM_AUTO, "auto"), (sat_amd.SEM_SHIFT_AND_INEXACT, "sai"), (sat_amd.SEM_FILTER_BITVEC, "fbv"), (sat_amd.SEM_EXACT_HALVES, "halves"),
        (sat_amd.SEM_EXACT_BASES, "bases")]


def small_case(seed):
    """One adversarial case the oracle finishes in well under a second: a stream of 300 .. 8000 characters of one of the
    four styles, 1 .. 150 primers (x 2 strands) cut from it, one option set, one way through the library, one setting of
    the measurement knobs that shrink rows / chunks / groups / tiles.  Returns a dict; `stream` is what getnch() hands
    out (codes with `table`, or raw bytes with table None)."""
    rng = np.random.default_rng(seed)
    style = int(rng.integers(0, 4))
    n = int(rng.integers(300, 8001))
    s = make_stream(rng, n, style)
    k = int(rng.integers(0, 3))
    indels = bool(rng.integers(0, 2)) and k > 0
    only20 = bool(rng.integers(0, 2))
    lo, hi = (20, 20) if only20 else (int(rng.integers(12, 21)), int(rng.integers(21, 33)))
    count = int(rng.integers(1, 151))
    pats = make_patterns(rng, s, count, lo, hi, k)
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    sem, sname = SEMS[int(rng.integers(0, len(SEMS)))]
    zones = None
    if sem == sat_amd.SEM_EXACT_BASES or rng.integers(0, 3) == 0:    # exact zones on a third of the cases, always for exact_bases
        zones = []
        for p in allp:
            e, f_ = (int(rng.integers(0, 9)), 0) if rng.integers(0, 2) else (0, int(rng.integers(0, 9)))
            if rng.integers(0, 4) == 0:
                e, f_ = int(rng.integers(0, 7)), int(rng.integers(0, 7))
            if sem == sat_amd.SEM_EXACT_BASES and max(e, f_) < 6:
                e = 6 + int(rng.integers(0, 4))
            zones.append((min(e, len(p)), min(f_, len(p))))
    wild = bool(rng.integers(0, 6) == 0)                              # ambiguity codes in the primers (-w) on a sixth of the cases
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
    with_n = bool(rng.integers(0, 4) == 0)                            # a few N in the stream on a quarter of the cases
    env = {}
    for name, val in KNOBS:
        v = val[int(rng.integers(0, len(val)))]
        if v is not None:
            env[name] = v
    cap = [1 << 10, 1 << 14, 1 << 20][int(rng.integers(0, 3))]
    mode = int(rng.integers(0, 4))
    raw = bool(rng.integers(0, 4) == 0)                               # the stream as bytes 'A','C','G','T','\n','N', no table
    table = None if raw else (b"ACGT\nN" if with_n else TABLE)
    if with_n:
        s = s.copy()
        s[rng.integers(0, n, int(rng.integers(1, 40)))] = 5
    if not raw and rng.integers(0, 8) == 0:                           # the end-of-entry character first in the table (code 0)
        table = b"\nACGT" + (b"N" if with_n else b"")
        s = np.array([1, 2, 3, 4, 0, 5], dtype=np.uint8)[s]
    stream = np.frombuffer(b"ACGT\nN", dtype=np.uint8)[s].copy() if raw else s
    chunk = int(rng.integers(64, 3000))
    cut = int(rng.integers(n // 4, 3 * n // 4))
    return dict(seed=seed, style=style, n=n, k=k, indels=indels, lo=lo, hi=hi, patterns=allp, sem=sem, sname=sname, zones=zones, wild=wild,
                with_n=with_n, env=env, cap=cap, mode=mode, raw=raw, table=table, stream=stream, chunk=chunk, cut=cut, host=bool(rng.integers(0, 3) == 0))


def describe(c):
    return "seed %d mode %d%s%s%s%s%s style %s n %d k %d indels %d L %d..%d patterns %d sem %s cap %d env %s" % (
        c["seed"], c["mode"], " raw" if c["raw"] else "", " zones" if c["zones"] else "", " wild" if c["wild"] else "", " N" if c["with_n"] else "",
        " host" if c["host"] else "", STYLES[c["style"]], c["n"], c["k"], c["indels"], c["lo"], c["hi"], len(c["patterns"]), c["sname"], c["cap"],
        ",".join("%s=%s" % (e[3:], v) for e, v in sorted(c["env"].items())))


class knobs:
    """the library reads its measurement knobs from the environment once, in pm_create"""

    def __init__(self, env):
        self.env, self.old = env, {}

    def __enter__(self):
        for name, _ in KNOBS:
            self.old[name] = os.environ.pop(name, None)
        os.environ.update(self.env)

    def __exit__(self, *a):
        for name, _ in KNOBS:
            os.environ.pop(name, None)
            if self.old[name] is not None:
                os.environ[name] = self.old[name]


def gpu_hits(c, kernel=sat_amd.KERNEL_SEED, mode=None, guard=256, stats=None, cuts=None):
    """the case through the library; sorted (end, pid, k) tuples.  mode 0: find_all over the whole stream; 1: find_all in
    small consecutive ranges (resumable pm_scan); 2: one scan + the device finalize (bench.py's single-rank step); 3: two
    position shards, each finalized on its own with a guard band (bench.py's multi-rank step).  Raises PmError(-2) where
    the library says the mode does not apply to the option set."""
    import torch
    mode = c["mode"] if mode is None else mode
    with knobs(c["env"]):
        pm = sat_amd.PatternMatch(k=c["k"], indels=c["indels"], kernel=kernel, semantics=c["sem"], wildcards=c["wild"])
    try:
        for i, p in enumerate(c["patterns"]):
            z = c["zones"][i] if c["zones"] else (0, 0)
            pm.add_pattern(p, i + 1, z[0], z[1])
        n = c["n"]
        if c["host"]:
            pm.init(c["stream"], c["table"])
        else:
            dev = torch.from_numpy(c["stream"]).cuda()
            pm.init_device(dev.data_ptr(), n, c["table"], keepalive=dev)
        pm.set_capacity(c["cap"])
        if cuts is not None:                                              # pm_scan ranges that end at the given stream positions
            pm.reset()
            parts, pos = [], 0
            # (exact_halves / exact_bases extend a seed found inside the range to an end that may lie beyond it, as the reference's
            # wrappers do around their inner engine's position: exact_halves.cc:153-167; the other engines' ends are window ends)
            strict = pm.selected()[0] not in (sat_amd.SEM_EXACT_HALVES, sat_amd.SEM_EXACT_BASES)
            for e in sorted(set(min(max(int(x), 0), n) for x in cuts) | {n}):
                if e > pos:
                    parts.append(pm.scan_view(pos, e).copy())
                    assert not strict or parts[-1].size == 0 or (int(parts[-1]["end"].max()) <= e), "pm_scan: a hit beyond the scanned-to position"
                    pos = e
            h = np.concatenate(parts)
        elif mode == 0 and stats is not None:                               # three equal pm_scan ranges (the second and third guessed and scanned ahead), their spans as handed out
            pm.reset()
            parts, pos = [], 0
            for e in (n // 3, 2 * (n // 3), n):
                if e > pos:
                    parts.append(pm.scan_view(pos, e).copy())
                    t = list(zip(parts[-1]["end"].tolist(), parts[-1]["pid"].tolist(), parts[-1]["k"].tolist()))
                    assert t == sorted(t), "pm_scan_view: hits not in (end, pid, k) order"
                    pos = e
            h = np.concatenate(parts)
            stats.update(pm.scan_stats())
        elif mode == 0:
            h = pm.find_all()
        elif mode == 1:
            h = pm.find_all(chunk=c["chunk"])
        elif mode == 2:
            pm.reset()
            pm.scan_candidates(0, n, to_host=False)
            h = pm.finalize_device(n, last=True, sort=True, out=np.zeros(2 * pm.candidates_device()[1] + 1024, dtype=sat_amd.HIT_DTYPE))
        else:
            parts = []
            for own_lo, own_hi in ((0, c["cut"]), (c["cut"], n)):
                g_lo, g_hi = max(0, own_lo - guard), min(n, own_hi + guard)
                pm.reset()
                pm.scan_candidates(g_lo, g_hi, to_host=False)
                parts.append(pm.finalize_device(0, sort=True, owned=(own_lo, own_hi, g_lo, None if g_hi == n else g_hi),
                                                out=np.zeros(2 * pm.candidates_device()[1] + 1024, dtype=sat_amd.HIT_DTYPE)).copy())
            h = np.concatenate(parts)
        c["selected"], c["kernel_desc"] = pm.selected(), pm.describe()
        return sat_amd.sorted_tuples(h)
    finally:
        pm.close()


def oracle_hits(c):
    """the case through oracle/pm_oracle.c (the CPU restatement pinned to the real reference, tests/test_oracle_vs_ref.py);
    None when the reference rejects the option set (select.cc:87-90: edits >= inexact bases)."""
    from oracle import pmoracle as O
    text = O.Text(c["stream"], c["table"])
    eng = c["sem"]
    if c["wild"]:                                                     # with pattern classes the seed engine is shift_and (select.cc:101-104)
        eng = {sat_amd.SEM_EXACT_HALVES: O.EXACT_HALVES_SA, sat_amd.SEM_EXACT_BASES: O.EXACT_BASES_SA}.get(eng, eng)
    E = [z[0] for z in c["zones"]] if c["zones"] else None
    F = [z[1] for z in c["zones"]] if c["zones"] else None
    try:
        return O.sorted_tuples(O.find_all(text, c["patterns"], engine=eng, k=c["k"], indels=c["indels"], esb=E, eeb=F, wildcards=c["wild"]))
    except RuntimeError:
        return None
