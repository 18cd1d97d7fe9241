"""GPU: completeness of the seed plans, exhaustively.

The seed-family kernels find candidates through necessary conditions -- field pairs that <= 2
substitutions leave intact (pm_pair.hip), displaced piece triples that <= 2 edits leave intact and
a greedy set cover over the (combo, displacement) pairs that are tested (edit_cover,
edits_plausible in pm_seed.hip), half seeds + a partner prefilter for exact_halves.  Random plants
exercise a sliver of the placements; here EVERY text reachable from a 20-, 24- and 32-mer by <= 2
substitutions, and every text reachable by <= 2 edits (substitution, insertion, deletion, in any
two positions), is put into the stream -- each between end-of-sequence characters, so that
nothing reaches across -- next to thousands of random decoy patterns that fill the filters, and the
HIP path must report exactly what the oracle reports (reference semantics:
shift_and_inexact.cc:249-352 candidates, filter_bitvec.cc:88-177 / exact_halves.cc:120-197 on top),
every variant at least once.  Bit-exact."""
import itertools

import numpy as np
import pytest

import sat_amd
import synth
from oracle import pmoracle as O

pytestmark = pytest.mark.gpu
TABLE = b"ACGT\n"


def substitution_variants(p, k):
    out = {p}
    for pos in itertools.combinations(range(len(p)), k):
        for repl in itertools.product("ACGT", repeat=k):
            if all(repl[j] != p[pos[j]] for j in range(k)):
                w = list(p)
                for j in range(k):
                    w[pos[j]] = repl[j]
                out.add("".join(w))
    return out


def one_edit(w):
    out = set()
    for i in range(len(w)):
        out.add(w[:i] + w[i + 1:])                                   # deletion
        for c in "ACGT":
            if c != w[i]:
                out.add(w[:i] + c + w[i + 1:])                       # substitution
    for i in range(len(w) + 1):
        for c in "ACGT":
            out.add(w[:i] + c + w[i:])                               # insertion
    return out


def edit_variants(p, k):
    level = {p}
    seen = {p}
    for _ in range(k):
        nxt = set()
        for w in level:
            nxt |= one_edit(w)
        level = nxt - seen
        seen |= nxt
    return seen


def stream_of(variants, rng):
    """variants between EOS characters, two to four random bases in front of each and up to three
    behind: the k-error automaton cannot delete a pattern's first characters right behind an
    end-of-sequence character (its rows are cleared there, shift_and_inexact.cc:293), it needs text
    to substitute -- with the padding every variant is within two edits of a text window"""
    parts = []
    for v in variants:
        pad_l = "".join(rng.choice(list("ACGT"), size=int(rng.integers(2, 5))).tolist())
        pad_r = "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 4))).tolist())
        parts.append(pad_l + v + pad_r)
    return parts


def run_gpu(codes, pats, k, indels, sem=sat_amd.SEM_AUTO, kernel=sat_amd.KERNEL_AUTO):
    pm = sat_amd.PatternMatch(k=k, indels=indels, semantics=sem, kernel=kernel)
    for i, p in enumerate(pats):
        pm.add_pattern(p, i + 1)
    pm.init(codes, TABLE)
    fam = pm.selected()[1]
    out = sat_amd.sorted_tuples(pm.find_all())
    pm.close()
    return out, fam


def entry_bounds(parts):
    """stream index range (first, one past last) of every entry in the .sqn layout of synth.stream"""
    b, at = [], 1
    for s in parts:
        b.append((at, at + len(s)))
        at += len(s) + 1
    return b


PATTERNS = {
    20: "ACGTTGCAAGCTTAGGCTCA",
    24: "GATTACAGGCTTAACCGTGTCAAT",
    32: "TTGACCGTAGGCATCGATCGGATCCTAGCAAT",
}


@pytest.mark.parametrize("L", [20, 24, 32])
def test_every_text_within_two_substitutions(L):
    rng = np.random.default_rng(L)
    p = PATTERNS[L]
    variants = sorted(set().union(*(substitution_variants(p, k) for k in (0, 1, 2))))
    assert len(variants) == 1 + 3 * L + 9 * L * (L - 1) // 2
    parts = stream_of(variants, rng)
    codes = synth.normalize(synth.stream(parts), TABLE)
    decoys = ["".join(rng.choice(list("ACGT"), size=L).tolist()) for _ in range(3000)]
    pats = [p] + decoys
    text = O.Text(codes, TABLE)
    for k, sem, osel in ((2, sat_amd.SEM_AUTO, 0), (1, sat_amd.SEM_AUTO, 0), (2, sat_amd.SEM_SHIFT_AND_INEXACT, 100),
                         (2, sat_amd.SEM_EXACT_HALVES, 12), (1, sat_amd.SEM_FILTER_BITVEC, 5)):
        eng = O.pick_engine(text, pats, k, False) if osel == 0 else osel
        want = O.sorted_tuples(O.find_all(text, pats, engine=eng, k=k, indels=False))
        got, fam = run_gpu(codes, pats, k, False, sem)
        assert fam == sat_amd.KERNEL_SEED
        assert got == want, (L, k, sem, len(got), len(want))
        if sem == sat_amd.SEM_SHIFT_AND_INEXACT:                         # raw candidates: every variant within k ends a hit of pattern 1
            ends = {e for e, pid, d in got if pid == 1}
            for (a, b), v in zip(entry_bounds(parts), variants):
                if sum(x != y for x, y in zip(v, p)) <= k:
                    assert any(a < e <= b for e in ends), (L, k, v)


@pytest.mark.parametrize("L", [20, 24, 32])
def test_every_text_within_two_edits(L):
    rng = np.random.default_rng(100 + L)
    p = PATTERNS[L]
    variants = sorted(edit_variants(p, 2))
    assert len(variants) > 5000
    parts = stream_of(variants, rng)
    codes = synth.normalize(synth.stream(parts), TABLE)
    decoys = ["".join(rng.choice(list("ACGT"), size=L).tolist()) for _ in range(1500)]
    pats = [p] + decoys
    text = O.Text(codes, TABLE)
    for k, sem, osel in ((2, sat_amd.SEM_AUTO, 0), (2, sat_amd.SEM_SHIFT_AND_INEXACT, 100), (1, sat_amd.SEM_FILTER_BITVEC, 5),
                         (1, sat_amd.SEM_AUTO, 0), (2, sat_amd.SEM_EXACT_HALVES, 12)):
        eng = O.pick_engine(text, pats, k, True) if osel == 0 else osel
        want = O.sorted_tuples(O.find_all(text, pats, engine=eng, k=k, indels=True))
        got, fam = run_gpu(codes, pats, k, True, sem)
        assert fam == sat_amd.KERNEL_SEED
        assert got == want, (L, k, sem, len(got), len(want))
        if sem == sat_amd.SEM_SHIFT_AND_INEXACT and k == 2:               # every variant is within two edits: one candidate each at least
            ends = np.array(sorted({e for e, pid, d in got if pid == 1}))
            for (a, b) in entry_bounds(parts):
                i = np.searchsorted(ends, a, side="right")
                assert i < ends.size and ends[i] <= b + 0, (L, a, b)
