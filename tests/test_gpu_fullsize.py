"""GPU: size-independent properties at sizes the oracle cannot reach (hundreds of Mbp, 10^5
patterns): planted primers are found with the right distance, the two kernel families agree,
position shards add up, a rescan is idempotent."""
import numpy as np
import pytest
import torch

import sat_amd

pytestmark = pytest.mark.gpu
TABLE = b"ACGT\n"
LUT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_db(n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    t = torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda", generator=g)
    t[0] = 4
    t[-1] = 4
    t[n // 3] = 4
    return t


def planted(host, rng, count, L, nsub):
    """(primer string, end position, substitutions) sampled from the stream"""
    out = []
    while len(out) < count:
        a = int(rng.integers(1, host.size - L - 1))
        w = host[a:a + L].copy()
        if (w > 3).any():
            continue
        pos = rng.choice(L, size=nsub, replace=False)
        for i in pos:
            w[i] = (w[i] + 1 + int(rng.integers(0, 3))) % 4
        out.append((LUT[w].tobytes().decode(), a + L, nsub))
    return out


def engine(pats, k, kernel, dev, sem=sat_amd.SEM_AUTO, indels=False):
    pm = sat_amd.PatternMatch(k=k, indels=indels, kernel=kernel, semantics=sem)
    for i, p in enumerate(pats):
        pm.add_pattern(p, i + 1)
    pm.init_device(dev.data_ptr(), dev.numel(), TABLE, keepalive=dev)
    return pm


@pytest.mark.parametrize("k", [0, 1, 2])
def test_planted_primers_found_at_scale(k):
    n, L, P = 1 << 28, 20, 50_000
    dev = make_db(n, 11 + k)
    host = dev[: 1 << 24].cpu().numpy()
    rng = np.random.default_rng(k)
    plant = [x for d in range(k + 1) for x in planted(host, rng, 300, L, d)]
    rnd = ["".join("ACGT"[c] for c in rng.integers(0, 4, L)) for _ in range(P - len(plant))]
    pats = [p for p, _, _ in plant] + rnd
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev)
    assert pm.selected()[1] == sat_amd.KERNEL_SEED
    pm.set_capacity(1 << 23)
    hits = pm.find_all()
    got = {(int(e), int(p)): int(d) for e, p, d in zip(hits["end"], hits["pid"], hits["k"])}
    for i, (_, end, d) in enumerate(plant):
        assert (end, i + 1) in got, ("planted primer not found", i, end, d)
        assert got[(end, i + 1)] <= d
    # idempotence + shard additivity of the device stage
    pm.reset()
    whole = np.sort(pm.scan_candidates(0, n), order=["end", "pid", "k"])
    cuts = [0, n // 5 + 123, n // 2 - 7, n]
    parts = np.concatenate([pm.scan_candidates(cuts[i], cuts[i + 1]) for i in range(3)])
    parts = np.sort(parts, order=["end", "pid", "k"])
    assert whole.size == parts.size and (whole["end"] == parts["end"]).all() and (whole["pid"] == parts["pid"]).all()
    again = np.sort(pm.scan_candidates(0, n), order=["end", "pid", "k"])
    assert (again["end"] == whole["end"]).all() and (again["k"] == whole["k"]).all()
    pm.close()


@pytest.mark.parametrize("k,indels", [(0, False), (1, False), (2, False)])
def test_seed_family_equals_bitpar_family(k, indels):
    """Cross-family equality on a stream both can afford (the bit-parallel kernel is ALU bound)."""
    n, L, P = 1 << 24, 20, 20_000
    dev = make_db(n, 5)
    host = dev.cpu().numpy()
    rng = np.random.default_rng(40 + k)
    plant = [x for d in range(3) for x in planted(host, rng, 200, L, d)]
    pats = [p for p, _, _ in plant] + ["".join("ACGT"[c] for c in rng.integers(0, 4, L)) for _ in range(P - len(plant))]
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    a = engine(allp, k, sat_amd.KERNEL_SEED, dev, indels=indels)
    b = engine(allp, k, sat_amd.KERNEL_BITPAR, dev, indels=indels)
    ha, hb = sat_amd.sorted_tuples(a.find_all()), sat_amd.sorted_tuples(b.find_all())
    assert ha == hb and len(ha) >= 200, (k, len(ha), len(hb))
    a.close()
    b.close()


# ---- BASELINE.json config sizes -------------------------------------------------------------------
def plant_edits(host, rng, count, L, nedit):
    """(primer, stream index of the site's first base, site length, edits): the primer is the site with
    `nedit` random edits (substitution / insertion / deletion)"""
    out = []
    while len(out) < count:
        a = int(rng.integers(1, host.size - L - 4))
        w = host[a:a + L]
        if (w > 3).any():
            continue
        s = LUT[w].tobytes().decode()
        p = s
        for _ in range(nedit):
            kind = int(rng.integers(0, 3))
            i = int(rng.integers(1, len(p) - 1))
            if kind == 0:
                p = p[:i] + "ACGT"[("ACGT".index(p[i]) + 1 + int(rng.integers(0, 3))) % 4] + p[i + 1:]
            elif kind == 1:
                p = p[:i] + "ACGT"[int(rng.integers(0, 4))] + p[i:]
            else:
                p = p[:i] + p[i + 1:]
        if len(p) < 20:
            continue
        out.append((p, a, L, nedit))
    return out


def random_primers(rng, count, L):
    a = rng.integers(0, 4, size=(count, L), dtype=np.uint8)
    return [LUT[r].tobytes().decode() for r in a]


def hit_index(hits):
    key = hits["pid"].astype(np.int64) << 40 | hits["end"]
    order = np.argsort(key)
    return key[order], hits["k"][order]


def found(key, kk, pid, end, tol, maxk):
    lo = np.searchsorted(key, (pid << 40) | max(0, end - tol))
    hi = np.searchsorted(key, (pid << 40) | (end + tol), side="right")
    return hi > lo and kk[lo:hi].min() <= maxk


@pytest.mark.parametrize("k", [0, 2])
def test_config_size_one_gbp_100k_primers(k):
    """BASELINE configs 2 and 3: 100k 20-mers (both strands) against 1 Gbp, exact and -K 2: every
    planted site is reported with its distance -- sites near the end of the stream included -- and on
    two 8 Mbp slices (one at the start, one across 2^29) the seed family's records equal the
    bit-parallel family's (the literal Shift-And automaton, shift_and.cc:208-255 /
    shift_and_inexact.cc:249-352)."""
    n, L, P = 1_000_000_000, 20, 100_000
    dev = make_db(n, 21 + k)
    rng = np.random.default_rng(300 + k)
    head, tail = dev[: 1 << 24].cpu().numpy(), dev[n - (1 << 22):].cpu().numpy()
    plant = [x for d in range(k + 1) for x in planted(head, rng, 200, L, d)]
    plant_tail = [(p, e + n - (1 << 22), d) for d in range(k + 1) for (p, e, d) in planted(tail, rng, 100, L, d)]
    plant += plant_tail
    pats = [p for p, _, _ in plant] + random_primers(rng, P - len(plant), L)
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev)
    assert pm.selected()[1] == sat_amd.KERNEL_SEED
    pm.set_capacity(1 << 24)
    hits = pm.find_all(chunk=1 << 30)
    key, kk = hit_index(hits)
    for i, (_, end, d) in enumerate(plant):
        assert found(key, kk, i + 1, end, 2 * k + 1 if k else 0, d), ("planted primer not found", i, end, d)
    # cross-family equality of the device stage on slices
    bp = engine(allp, k, sat_amd.KERNEL_BITPAR, dev)
    for lo in (0, (1 << 29) - (1 << 22)):
        a = np.sort(pm.scan_candidates(lo, lo + (1 << 23)), order=["end", "pid", "k"])
        b = np.sort(bp.scan_candidates(lo, lo + (1 << 23)), order=["end", "pid", "k"])
        assert a.size == b.size and (a["end"] == b["end"]).all() and (a["pid"] == b["pid"]).all() and (a["k"] == b["k"]).all(), (k, lo, a.size, b.size)
    pm.close()
    bp.close()


def test_one_million_primers_eight_tiles():
    """BASELINE config 4's pattern set: 1M 20-mers, both strands = 2M patterns = 8 pattern tiles of
    the seed family, -K 2, 256 Mbp: planted sites of primers from every tile are reported."""
    n, L, P = 1 << 28, 20, 1_000_000
    dev = make_db(n, 31)
    rng = np.random.default_rng(31)
    host = dev[: 1 << 24].cpu().numpy()
    plant = [x for d in range(3) for x in planted(host, rng, 240, L, d)]
    rnd = random_primers(rng, P - len(plant), L)
    # spread the planted primers over the whole id range, i.e. over all tiles
    step = P // len(plant)
    pats, where = list(rnd), {}
    for j, (p, e, d) in enumerate(plant):
        pats.insert(j * step, p)
    for j in range(len(plant)):
        where[j * step] = plant[j]
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, 2, sat_amd.KERNEL_AUTO, dev)
    assert pm.selected()[1] == sat_amd.KERNEL_SEED and "tiles=8" in pm.describe(), pm.describe()
    pm.set_capacity(1 << 24)
    hits = pm.find_all(chunk=1 << 30)
    key, kk = hit_index(hits)
    for idx, (_, end, d) in where.items():
        assert found(key, kk, idx + 1, end, 5, d), ("planted primer not found", idx, end, d)
    pm.close()


def test_edit_distance_plants_at_scale():
    """-k 2 (edits) on 256 Mbp x 50k primers: primers that are database sites with 0, 1 or 2 random
    edits (substitutions, insertions, deletions) are reported at their site with at most that many
    edits (filter_bitvec.cc:88-177 reports one hit per chain of candidates, the end may move by the
    edits)."""
    n, L, P, k = 1 << 28, 22, 50_000, 2
    dev = make_db(n, 41)
    rng = np.random.default_rng(41)
    host = dev[: 1 << 24].cpu().numpy()
    plant = [x for d in range(3) for x in plant_edits(host, rng, 300, L, d)]
    pats = [p for p, _, _, _ in plant] + random_primers(rng, P - len(plant), L)
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev, indels=True)
    assert pm.selected() == (sat_amd.SEM_FILTER_BITVEC, sat_amd.KERNEL_SEED)
    hits = pm.find_all(chunk=1 << 30)
    key, kk = hit_index(hits)
    for i, (_, a, sl, d) in enumerate(plant):
        assert found(key, kk, i + 1, a + sl, 2 * k + 1 + d, d), ("planted primer not found", i, a, d)
    pm.close()


@pytest.mark.parametrize("L", [20, 23])
def test_exact_halves_edits_at_scale(L):
    """-k 1 (exact_halves, exact_halves.cc:120-224) on 1 Gbp x 100k primers, the ranked plan (pm_half_scan +
    pm_half_verify): primers that are database sites with 0 or 1 random edit -- one half stays exact, the other
    is within one edit -- are reported at their site, at the start of the stream and at its far end; the
    round-1 form behind PM_HALF_SCAN=bloom reports the same hits on a 64 Mbp prefix."""
    n, P, k = 1 << 30, 100_000, 1
    dev = make_db(n, 61 + L)
    rng = np.random.default_rng(61 + L)
    head = dev[: 1 << 24].cpu().numpy()
    tail_at = n - (1 << 24)
    tail = dev[tail_at:].cpu().numpy()
    plant = [x for d in range(2) for x in plant_edits(head, rng, 200, L, d)]
    plant += [(p_, a + tail_at, sl, d) for d in range(2) for (p_, a, sl, d) in plant_edits(tail, rng, 200, L, d)]
    pats = [p_ for p_, _, _, _ in plant] + random_primers(rng, P - len(plant), L)
    allp = pats + [sat_amd.reverse_comp(p_) for p_ in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev, indels=True)
    assert pm.selected() == (sat_amd.SEM_EXACT_HALVES, sat_amd.KERNEL_SEED)
    assert "pm_half_scan" in pm.describe(), pm.describe()
    hits = pm.find_all(chunk=1 << 30)
    key, kk = hit_index(hits)
    for i, (_, a, sl, d) in enumerate(plant):
        assert found(key, kk, i + 1, a + sl, 2 * k + 1 + d, d), ("planted primer not found", i, a, d)
    pm.close()
    # the two forms of the seed stage agree (prefix of the stream, every hit)
    import os
    sub = dev[: 1 << 26]
    got = []
    for form in ("", "bloom"):
        if form:
            os.environ["PM_HALF_SCAN"] = form
        try:
            pm = engine(allp, k, sat_amd.KERNEL_AUTO, sub, indels=True)
            got.append(sat_amd.sorted_tuples(pm.find_all(chunk=1 << 30)))
            pm.close()
        finally:
            os.environ.pop("PM_HALF_SCAN", None)
    assert len(got[0]) > 100 and got[0] == got[1]


def test_stream_beyond_two_to_the_32_with_default_chunks():
    """4.3e9 stream bytes (positions need more than 32 bits), -K 2 with the chunk size the library
    picks by itself for large ranges (2 Mi positions per workgroup): sites planted at the far end of
    the stream, right at chunk edges and at the very start are reported; the count of the run equals
    the sum over three sub-ranges (the device stage is a pure function of the range)."""
    n, L, P, k = (1 << 32) + (1 << 26), 20, 20_000, 2
    dev = make_db(n, 51)
    rng = np.random.default_rng(51)
    regions = [0, (1 << 21) * 700 - 4096, (1 << 32) - (1 << 20), n - (1 << 21)]       # start, a chunk edge, across 2^32, the end
    plant = []
    for r in regions:
        w = dev[r:r + (1 << 21)].cpu().numpy() if r + (1 << 21) <= n else dev[r:].cpu().numpy()
        plant += [(p, e + r, d) for d in range(3) for (p, e, d) in planted(w, rng, 60, L, d)]
    # sites whose window straddles a 2 Mi chunk edge exactly
    edge = (1 << 21) * 1000
    w = dev[edge - 64:edge + 64].cpu().numpy()
    for off in (44, 50, 54, 60):
        s = w[off:off + L]
        if (s > 3).any():
            continue
        plant.append((LUT[s].tobytes().decode(), edge - 64 + off + L, 0))
    pats = [p for p, _, _ in plant] + random_primers(rng, P - len(plant), L)
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev)
    pm.set_capacity(1 << 24)
    whole = pm.scan_candidates(0, n)
    assert "chunk=2097152" in pm.describe(), pm.describe()
    key, kk = hit_index(whole)
    for i, (_, end, d) in enumerate(plant):
        assert found(key, kk, i + 1, end, 0, d), ("planted primer not found", i, end, d)
    cuts = [0, (1 << 31) + 12345, (1 << 32) + 77, n]
    total = sum(pm.scan_candidates(cuts[i], cuts[i + 1], to_host=False) for i in range(3))
    assert total == whole.size
    pm.close()
