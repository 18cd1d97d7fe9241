#!/usr/bin/env python3
"""Which (field pair, displacement) tests would an edit-distance plan on the PAIR geometry need?  

The last 20 pattern bases are four fields of five.  A text within <= k edits (substitution, insertion, deletion) of
the pattern leaves >= 2 fields clean for k = 2; two clean fields A < B appear in the text displaced by d = net
insertions between them.  A test (A, B, d) -- "field A here and field B 5 (B - A) + d bases further on are the pattern's" --
finds the alignment.  This script enumerates every placement of <= k edits, reduces it to its set of clean pairs with
their displacements, and computes the exact minimum set of tests that hits every scenario (brute force over subsets).

    python scripts/edit_pair_cover.py        ->  k = 1: 2 tests, k = 2: 14 tests
"""
import itertools

F, L = 4, 5
EDITS = [("s", i) for i in range(F * L)] + [("d", i) for i in range(F * L)] + [("i", i) for i in range(F * L + 1)]


def scenario(edits):
    """clean field pairs (a, b, displacement) of the pattern after `edits` (sub at base i / delete base i / insert before base i)"""
    dirty, events = set(), []
    for kind, i in edits:
        if kind == "s":
            dirty.add(i // L)
        elif kind == "d":
            dirty.add(i // L)
            events.append((i + 1, -1))          # every base behind the deleted one moves up
        else:
            if i % L:
                dirty.add(i // L)                # an insertion inside a field splits it; at a field boundary it only shifts
            events.append((i, +1))
    clean = [f for f in range(F) if f not in dirty]
    shift = lambda f: sum(d for pos, d in events if pos <= f * L)
    return frozenset((a, b, shift(b) - shift(a)) for a, b in itertools.combinations(clean, 2))


def min_cover(k):
    scen = {scenario(es) for n in range(k + 1) for es in itertools.combinations_with_replacement(EDITS, n)}
    assert all(scen), "a placement without a clean pair"
    tests = sorted(set().union(*scen))
    for r in range(1, len(tests) + 1):
        for sub in itertools.combinations(tests, r):
            s = set(sub)
            if all(x & s for x in scen):
                return len(scen), tests, sub
    raise AssertionError


if __name__ == "__main__":
    for k in (1, 2):
        n, tests, cover = min_cover(k)
        print("k = %d: %d distinct scenarios, %d tests can occur, minimum cover %d: %s" % (k, n, len(tests), len(cover), list(cover)))
