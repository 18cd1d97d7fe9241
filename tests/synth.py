"""Seeded synthetic inputs shared by the tests, the golden-vector generator and bench.py.

Layout of the stream follows compress_seq's .seq/.sqn (compress_seq.cc:438-665, 704-723):
a leading EOS, then every FASTA entry followed by one EOS; with `-n true` bytes are codes
A0 C1 G2 T3, then the other observed chars in ASCII order ('\\n' = 4, 'N' = 5 when present).
"""
import numpy as np

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(p):
    return "".join(COMP.get(c, c) for c in reversed(p))


IUPAC_COMP = dict(COMP, M="K", K="M", R="Y", Y="R", V="B", B="V", H="D", D="H", W="W", S="S", U="A")


def revcomp_iupac(p):
    return "".join(IUPAC_COMP.get(c, c) for c in reversed(p))


def make_entries(rng, n_entries, length, n_runs=0, repeats=False, short=False):
    """List of DNA strings: uniform ACGT, optional N runs, tandem repeats and a too-short entry."""
    ents = []
    for e in range(n_entries):
        s = rng.choice(list("ACGT"), size=length).tolist()
        for _ in range(n_runs):
            a = int(rng.integers(0, max(1, length - 8)))
            for i in range(a, min(length, a + int(rng.integers(1, 8)))):
                s[i] = "N"
        ents.append("".join(s))
    if repeats:
        ents.append("AC" * 20 + "G" + "A" * 30 + "T" + "ACG" * 12)
    if short:
        ents.append("ACGTACG")
    return ents


def table_for(entries):
    """The .tbl byte string compress_seq -n true would write for these entries."""
    obs = set("\n")
    for s in entries:
        obs.update(s)
    order = list("ACGT") + sorted(c for c in obs if c not in "ACGT")
    return "".join(c for c in order if c in obs or c in "ACGT").encode()


def stream(entries):
    """Raw .seq byte string."""
    return ("\n" + "".join(s + "\n" for s in entries)).encode()


def normalize(raw, table):
    inv = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(table):
        inv[c] = i
    return inv[np.frombuffer(raw, dtype=np.uint8)]


def mutate(rng, w, nsub=0, nins=0, ndel=0):
    w = list(w)
    for _ in range(nsub):
        i = int(rng.integers(0, len(w)))
        w[i] = rng.choice([c for c in "ACGT" if c != w[i]])
    for _ in range(nins):
        i = int(rng.integers(1, len(w)))
        w.insert(i, str(rng.choice(list("ACGT"))))
    for _ in range(ndel):
        if len(w) > 2:
            del w[int(rng.integers(1, len(w) - 1))]
    return "".join(w)


def make_patterns(rng, entries, n, length=20, planted=0.5, minlen=None, indel_frac=0.0, extras=True):
    """n primers: `planted` fraction sampled from the DB (pattern = DB window with 0..2 changes),
    the rest uniform random; plus (extras) a duplicate, a palindrome and one reverse-complement
    pair so the id bookkeeping is exercised."""
    pats = []
    joined = [s for s in entries if len(s) >= length + 2]
    for i in range(n):
        L = length if minlen is None else int(rng.integers(minlen, length + 1))
        if joined and rng.random() < planted:
            s = joined[int(rng.integers(0, len(joined)))]
            a = int(rng.integers(0, len(s) - L))
            w = s[a:a + L].replace("N", "A")
            if rng.random() < indel_frac:
                w = mutate(rng, w, nsub=int(rng.integers(0, 2)), nins=int(rng.integers(0, 2)), ndel=int(rng.integers(0, 2)))
            else:
                w = mutate(rng, w, nsub=int(rng.integers(0, 3)))
            if rng.random() < 0.5:
                w = revcomp(w)
            pats.append(w)
        else:
            pats.append("".join(rng.choice(list("ACGT"), size=L).tolist()))
    if extras and n >= 4:
        pats[1] = pats[0]                                   # duplicate
        half = "".join(rng.choice(list("ACGT"), size=length // 2).tolist())
        pats[2] = half + revcomp(half)                      # palindrome: equals its own revcomp
        pats[3] = revcomp(pats[0])                          # reverse-complement pair
    return pats
