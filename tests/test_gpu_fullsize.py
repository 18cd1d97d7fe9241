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
