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


def all_hits(pm, n, capacity):
    """every final hit of one pass over the whole stream, the way bench.py's step gets them: device
    stage, then the clustering / verify stage on the GPU (pm_finalize_device), as a numpy array"""
    pm.set_capacity(capacity)
    cnt = pm.scan_candidates(0, n, to_host=False)
    ptr, cnt2 = pm.candidates_device()
    assert cnt == cnt2
    return pm.finalize_device(n, last=True, sort=False, d_cands=ptr, n=cnt)


def test_config4_one_million_primers_three_gbp():
    """BASELINE config 4 on one GPU: 1M 20-mers, both strands = 2M patterns = 8 pattern tiles of the pair
    plan, -K 2, against 3 Gbp.  Sites planted with 0 / 1 / 2 substitutions for primers of EVERY tile --
    forward primers (tiles 0..3) and primers whose reverse complement is the site (tiles 4..7) -- at the
    start of the stream, beyond 2^31 and at its far end must all be reported with at most their planted
    distance (filter_bitvec.cc:88-177: one hit per chain, so the end may sit up to 2k+1 away); on two
    4 Mbp slices (one beyond 2^31) the records equal the bit-parallel family's, the literal
    shift_and_inexact automaton (shift_and_inexact.cc:249-352)."""
    n, L, P, k = 3_000_000_000, 20, 1_000_000, 2
    dev = make_db(n, 71)
    rng = np.random.default_rng(71)
    regions = [0, (1 << 31) + (1 << 27), n - (1 << 22)]
    plant = []                                                     # (primer, end, d, tile-side: 0 forward, 1 reverse complement)
    for r in regions:
        w = dev[r:r + (1 << 22)].cpu().numpy()
        for side in (0, 1):
            for d in range(3):
                for (p, e, dd) in planted(w, rng, 40, L, d):
                    plant.append((sat_amd.reverse_comp(p) if side else p, e + r, dd, side))
    assert max(e for _, e, _, _ in plant) > (1 << 31)
    rnd = random_primers(rng, P - len(plant), L)
    step = P // len(plant)                                           # spread over the whole id range: every forward tile, every reverse tile
    pats, where = list(rnd), {}
    for j, (p, e, d, side) in enumerate(plant):
        pats.insert(j * step, p)
    for j in range(len(plant)):
        where[j * step] = plant[j]
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev)
    assert pm.selected() == (sat_amd.SEM_FILTER_BITVEC, sat_amd.KERNEL_SEED)
    hits = all_hits(pm, n, 1 << 25)
    assert "pm_pair_scan" in pm.describe() and "tiles=8" in pm.describe(), pm.describe()
    key, kk = hit_index(hits)
    tiles_seen = set()
    for idx, (_, end, d, side) in where.items():
        pid = idx + 1 + (P if side else 0)
        assert found(key, kk, pid, end, 2 * k + 1, d), ("planted primer not found", idx, end, d, side)
        tiles_seen.add((pid - 1) // 250_000)
    assert tiles_seen == set(range(8)), tiles_seen
    # expected number of spurious hits: n * 2P * (1 + 60 + 1710) / 4^20 = 9.7e6; clustering removes few
    assert 8_000_000 < hits.size < 12_000_000, hits.size
    bp = engine(allp, k, sat_amd.KERNEL_BITPAR, dev)
    for lo in (1 << 20, (1 << 31) + (1 << 27)):
        a = np.sort(pm.scan_candidates(lo, lo + (1 << 22)), order=["end", "pid", "k"])
        b = np.sort(bp.scan_candidates(lo, lo + (1 << 22)), order=["end", "pid", "k"])
        assert a.size > 1000 and a.size == b.size and (a["end"] == b["end"]).all() and (a["pid"] == b["pid"]).all() and (a["k"] == b["k"]).all(), (lo, a.size, b.size)
    pm.close()
    bp.close()


def test_edit_distance_three_gbp_100k_primers():
    """-k 2 (edits; filter_bitvec over the k-error automaton's candidates, filter_bitvec.cc:88-177,
    shift_and_inexact.cc:249-352) at BASELINE size: 100k primers, both strands, 3 Gbp.  Primers that are
    database sites with 0, 1 or 2 random edits (substitution, insertion, deletion) are reported at their
    site with at most that many edits: sites at the very start of the stream, across an edge of the
    512 Ki-position chunks of the scan kernel (what 256 MiB ranges get), beyond 2^31 and at the far end; on a 2 Mbp slice beyond 2^31
    the deduplicated candidates equal the bit-parallel family's."""
    n, L, P, k = 3_000_000_000, 22, 100_000, 2
    dev = make_db(n, 81)
    rng = np.random.default_rng(81)
    plant = []
    for r in (0, (1 << 31) + (1 << 26), n - (1 << 22)):
        w = dev[r:r + (1 << 22)].cpu().numpy()
        plant += [(p, a + r, sl, d) for d in range(3) for (p, a, sl, d) in plant_edits(w, rng, 100, L, d)]
    edge = (1 << 19) * 3001                                        # a chunk edge of the edit-distance plan
    w = dev[edge - 64:edge + 64].cpu().numpy()
    for off in (30, 40, 44, 50, 56, 62):                           # sites that straddle or touch the edge
        site = w[off:off + L]
        if (site > 3).any():
            continue
        s_ = LUT[site].tobytes().decode()
        plant.append((s_, edge - 64 + off, L, 0))
        plant.append((s_[:7] + s_[8:], edge - 64 + off, L, 1))     # one deleted character
        plant.append((s_[:12] + "ACGT"[(("ACGT".index(s_[12])) + 1) % 4] + s_[13:], edge - 64 + off, L, 1))
    pats = [p for p, _, _, _ in plant] + random_primers(rng, P - len(plant), L)
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev, indels=True)
    assert pm.selected() == (sat_amd.SEM_FILTER_BITVEC, sat_amd.KERNEL_SEED)
    hits = all_hits(pm, n, 1 << 28)
    assert "pm_pair_edit_scan" in pm.describe() and "tests=14" in pm.describe(), pm.describe()      # (round 4: -k 2 on the pair geometry)
    key, kk = hit_index(hits)
    for i, (_, a, sl, d) in enumerate(plant):
        assert found(key, kk, i + 1, a + sl, 2 * k + 1 + d, d), ("planted primer not found", i, a, d)
    assert hits.size > 100_000, hits.size                           # (22-mers: ~3e5 chance hits within two edits per 3 Gbp x 200k patterns)
    bp = engine(allp, k, sat_amd.KERNEL_BITPAR, dev, indels=True)
    lo = (1 << 31) + (1 << 26)
    a = np.sort(pm.scan_candidates(lo, lo + (1 << 21)), order=["end", "pid", "k"])
    b = np.sort(bp.scan_candidates(lo, lo + (1 << 21)), order=["end", "pid", "k"])
    assert a.size > 300 and a.size == b.size and (a["end"] == b["end"]).all() and (a["pid"] == b["pid"]).all() and (a["k"] == b["k"]).all(), (a.size, b.size)
    pm.close()
    bp.close()


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


# ---- pm_scan (PatternMatch::find_patterns, pattern_match.h:131) = scan_candidates + finalize_device, sorted -----------
def _scan_all(pm, n, chunk, cap=None, view=False, lengths=None):
    """the whole stream through pm_scan / pm_scan_view in consecutive ranges; returns the concatenated records"""
    pm.reset()
    parts, pos, i = [], 0, 0
    buf = np.zeros(cap or 1, dtype=sat_amd.HIT_DTYPE)
    while pos < n:
        step = lengths[i % len(lengths)] if lengths else chunk
        end = min(n, pos + step)
        if view:
            parts.append(pm.scan_view(pos, end).copy())
        else:
            cnt, more = pm.scan(pos, end, buf)
            parts.append(buf[:cnt].copy())
            while more:
                cnt, more = pm.scan(end, end, buf)
                parts.append(buf[:cnt].copy())
        pos, i = end, i + 1
    return np.concatenate(parts) if parts else np.zeros(0, dtype=sat_amd.HIT_DTYPE)


@pytest.mark.parametrize("k,indels", [(2, False), (2, True), (0, False), (1, False), (1, True)])
def test_pm_scan_equals_candidates_plus_device_finalize_one_gbp(k, indels):
    """the reference-facing call must give what the benchmarked step gives -- the same hits, in
    (end, pid, k) order -- whatever the ranges: one call, the plugin's 256 MiB ranges through the zero-copy view,
    ranges of changing length (every guess of the pipelined next scan wrong) drained 70,000 records at a time."""
    n = (1 << 30) if not indels or k < 2 else (1 << 28)
    L, P = 20, 100_000 if not (indels and k == 2) else 20_000
    dev = make_db(n, 77 + k)
    host = dev[: 1 << 24].cpu().numpy()
    rng = np.random.default_rng(900 + 10 * k + int(indels))
    plant = [x for d in range(k + 1) for x in planted(host, rng, 200, L, d)]
    pats = [p for p, _, _ in plant] + ["".join("ACGT"[c] for c in rng.integers(0, 4, L)) for _ in range(P - len(plant))]
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_AUTO, dev, indels=indels)
    pm.set_capacity(1 << 24 if not (indels and k == 2) else 1 << 26)
    ncand = pm.scan_candidates(0, n, to_host=False)
    try:
        want = pm.finalize_device(n, last=True, sort=True, out=np.zeros(2 * ncand + 1024, dtype=sat_amd.HIT_DTYPE))
    except sat_amd.PmError as e:                                    # exact_halves in chunks etc.: the host stage
        assert e.code == -2
        pm.reset()
        want = pm.finalize(pm.copy_records(*pm.candidates_device()), n, last=True, sort=True)
    want = want.copy()
    assert want.size >= 200 and ncand >= want.size // 2
    order = np.lexsort((want["k"], want["pid"], want["end"]))
    assert (order == np.arange(want.size)).all()

    def same(got, what):
        assert got.size == want.size, (what, got.size, want.size)
        for f in ("end", "pid", "k"):
            assert (got[f] == want[f]).all(), (what, f)
        assert not got["aux"].any(), what

    same(_scan_all(pm, n, n, cap=1 << 24), "one call")
    same(_scan_all(pm, n, 1 << 28, view=True), "256 MiB ranges, view")
    same(_scan_all(pm, n, 1 << 28, view=True), "again (pm_reset in between)")
    same(_scan_all(pm, n, 0, cap=70_000, lengths=[(1 << 27) + 12345, (1 << 26) - 7, 1 << 28]), "ragged ranges, small buffer")
    # a scan nobody consumes, then direct stage calls on the same handle
    pm.reset()
    pm.scan_view(0, 1 << 26)
    assert pm.scan_candidates(0, n, to_host=False) == ncand
    pm.close()


@pytest.mark.parametrize("k,indels", [(2, True), (2, False), (1, True)])
def test_ranges_across_two_to_the_32_equal_the_oracle(k, indels):
    """Positions beyond 32 bits through every record format of the plans (the pair-edit plan's suspects carry
    pos | test << 40, its seeds pattern << 40 | pos; the final sort's key is end | pattern | k): a 4.3e9-byte stream,
    40,000 patterns; pm_scan over two ranges that meet at 2^32 + 77 must give, for the 200 patterns the oracle is given
    (filter_bitvec.cc:88-177 / exact_halves.cc:120-197 decide per pattern), exactly the oracle's hits on that window."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import pmoracle as O
    n, L = (1 << 32) + (1 << 26), 20
    dev = make_db(n, 91 + k)
    half = 1 << 19
    b, e = (1 << 32) - half, (1 << 32) + half
    margin = 256
    win = dev[b - margin:e + margin].cpu().numpy()
    rng = np.random.default_rng(500 + k)
    few = []
    for d in range(3):
        few += [p for p, _, _, _ in plant_edits(win[margin:-margin], rng, 30, L, d)] if indels else [p for p, _, _ in planted(win[margin:-margin], rng, 30, L, min(d, k))]
    few += random_primers(rng, 100 - len(few), L)
    few = [p[:32] for p in few]
    rest = random_primers(rng, 19_900, L)
    pats = few + rest
    allp = pats + [sat_amd.reverse_comp(p) for p in pats]
    pm = engine(allp, k, sat_amd.KERNEL_SEED, dev, indels=indels)
    if k == 2 and indels:
        assert "pm_pair_edit_scan" in pm.describe(), pm.describe()
    pm.reset()
    pos = 0
    while pos < b:                                                           # pm_scan's ranges are consecutive from the start of the stream
        nxt = min(b, pos + (1 << 30))
        pm.scan_view(pos, nxt)
        pos = nxt
    parts = [pm.scan_view(b, (1 << 32) + 77).copy(), pm.scan_view((1 << 32) + 77, e).copy()]
    got = np.concatenate(parts)
    P = len(pats)
    mine = {}                                                                # library pattern id -> oracle pattern id
    for i in range(100):
        mine[i + 1] = i + 1
        mine[P + i + 1] = 100 + i + 1
    lo, hi = b + 2 * L + 8, e - 2 * L - 8
    got = sorted((int(x["end"]), mine[int(x["pid"])], int(x["k"])) for x in got if int(x["pid"]) in mine and lo < int(x["end"]) <= hi)
    sub = few + [sat_amd.reverse_comp(p) for p in few]
    text = O.Text(win, TABLE)
    eng = O.pick_engine(text, sub, k, indels)
    want = sorted((int(end) + b - margin, int(pid), int(kk)) for end, pid, kk in O.sorted_tuples(O.find_all(text, sub, engine=eng, k=k, indels=indels))
                  if lo < int(end) + b - margin <= hi)
    assert len(want) >= 60, len(want)
    assert got == want, (len(got), len(want), [x for x in got if x not in want][:5], [x for x in want if x not in got][:5])
    pm.close()
