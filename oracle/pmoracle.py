"""oracle/pmoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/libpm_oracle.so (the CPU restatement in pm_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product package
never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HIT_DTYPE = np.dtype([("end", "<i8"), ("pid", "<u4"), ("k", "u1"), ("pad", "u1", (3,))])

AUTO, KT_LIST, KT_DNA, KT_JTABLE, SHIFT_AND, FILTER_BITVEC = 0, 1, 2, 3, 4, 5
EXACT_BASES_KT, EXACT_BASES_SA, EXACT_HALVES_KT, EXACT_HALVES_SA = 8, 10, 12, 14
SHIFT_AND_INEXACT = 100


class _Text(C.Structure):
    _fields_ = [("codes", C.c_void_p), ("n", C.c_int64), ("size", C.c_int),
                ("ch", C.c_uint8 * 256), ("nch", C.c_int32 * 256)]


class _Config(C.Structure):
    _fields_ = [("engine", C.c_int), ("k", C.c_int), ("indels", C.c_int),
                ("wildcards", C.c_int), ("text_n", C.c_int), ("eos", C.c_uint8)]


class _Alignment(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("editdist", C.c_int32), ("value", C.c_int32)]


def build():
    """Compile the C restatement (and, when /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if os.path.exists("/root/reference/primer_match.cc"):
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpm_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.pmo_find_all.restype = C.c_int
        L.pmo_pick_engine.restype = C.c_int
        L.pmo_editdist_align.restype = C.c_int
        L.pmo_cli_align.restype = C.c_int
        L.pmo_time_find_all.restype = C.c_double
        _LIB = L
    return _LIB


class Text:
    """A CharacterProducer view: `codes` (uint8 array) + alphabet table (None = raw bytes)."""

    def __init__(self, codes, table=None):
        self.codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.table = None if table is None else bytes(table)
        self._t = _Text()
        if self.table is None:
            lib().pmo_text_raw(C.byref(self._t), self.codes.ctypes.data_as(C.c_void_p), C.c_int64(self.codes.size))
        else:
            tb = (C.c_uint8 * len(self.table)).from_buffer_copy(self.table)
            lib().pmo_text_normalized(C.byref(self._t), self.codes.ctypes.data_as(C.c_void_p),
                                      C.c_int64(self.codes.size), tb, C.c_int(len(self.table)))


def _pack(patterns):
    pats = [p.encode() if isinstance(p, str) else bytes(p) for p in patterns]
    off = np.zeros(len(pats) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(p) for p in pats])
    buf = b"".join(pats) + b"\0"
    return buf, off


def _cfg(engine, k, indels, eos, wildcards=False, text_n=False):
    c = _Config()
    c.engine, c.k, c.indels, c.wildcards, c.text_n, c.eos = engine, k, int(bool(indels)), int(bool(wildcards)), int(bool(text_n)), eos
    return c


def find_all(text, patterns, engine=AUTO, k=0, indels=True, eos=10, ids=None, esb=None, eeb=None, wildcards=False, text_n=False):
    """All hits of one engine over the whole text, as a structured array in emission order."""
    buf, off = _pack(patterns)
    cfg = _cfg(engine, k, indels, eos, wildcards, text_n)
    out = C.c_void_p()
    n = C.c_size_t()
    a = lambda x, dt: None if x is None else np.ascontiguousarray(x, dtype=dt)
    ids_a, esb_a, eeb_a = a(ids, np.uint32), a(esb, np.int32), a(eeb, np.int32)
    p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    rc = lib().pmo_find_all(C.byref(text._t), C.byref(cfg), buf, off.ctypes.data_as(C.c_void_p),
                            C.c_int(len(patterns)), p(ids_a), p(esb_a), p(eeb_a), C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("pmo_find_all failed: %d" % rc)
    if n.value == 0:
        res = np.zeros(0, dtype=HIT_DTYPE)
    else:
        res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n.value * HIT_DTYPE.itemsize,)).view(HIT_DTYPE).copy()
    lib().pmo_free(out)
    return res


def sorted_tuples(hits):
    """Canonical comparison form: sorted list of (end, pid, k)."""
    return sorted(zip(hits["end"].tolist(), hits["pid"].tolist(), hits["k"].tolist()))


def pick_engine(text, patterns, k, indels=True, esb=None, eeb=None):
    pl = np.array([len(p) for p in patterns], dtype=np.int32)
    a = lambda x: None if x is None else np.ascontiguousarray(x, dtype=np.int32).ctypes.data_as(C.c_void_p)
    return lib().pmo_pick_engine(C.byref(text._t), C.c_int(k), C.c_int(int(indels)), C.c_int(0),
                                 C.c_int(len(patterns)), pl.ctypes.data_as(C.c_void_p), a(esb), a(eeb))


def cli_align(text, pattern, end, k, indels=True, eos=10, esb=0, eeb=0, wildcards=False, text_n=False):
    cfg = _cfg(0, k, indels, eos, wildcards, text_n)
    al = _Alignment()
    pb = pattern.encode() if isinstance(pattern, str) else bytes(pattern)
    rc = lib().pmo_cli_align(C.byref(text._t), C.byref(cfg), pb, C.c_int(len(pb)), C.c_int(esb), C.c_int(eeb),
                             C.c_int64(end), C.byref(al))
    return rc, al.start, al.end, al.editdist, al.value


def time_find_all(text, patterns, engine=AUTO, k=0, indels=True, eos=10):
    buf, off = _pack(patterns)
    cfg = _cfg(engine, k, indels, eos)
    n = C.c_size_t()
    s = lib().pmo_time_find_all(C.byref(text._t), C.byref(cfg), buf, off.ctypes.data_as(C.c_void_p),
                                C.c_int(len(patterns)), C.byref(n))
    return s, n.value


def reverse_comp(p):
    """A,C,G,T reverse complement (util.cc:374; IUPAC codes beyond ACGT are not restated)."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N",
            "a": "t", "c": "g", "g": "c", "t": "a", "n": "n"}
    return "".join(comp.get(c, c) for c in reversed(p))
