"""CPU: the C-ABI library loads, exports every symbol include/pm_gpu.h declares, and its
GPU-free entry points (engine auto-selection, argument checks) behave."""
import ctypes
import os
import re

import numpy as np
import pytest

import sat_amd
from sat_amd import pattern_match as P
from oracle import pmoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include", "pm_gpu.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(sat_amd.library_path())
    syms = header_symbols()
    assert set(syms) == set(P.ABI_SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), s


def test_hit_record_layout():
    assert P.HIT_DTYPE.itemsize == 16 and O.HIT_DTYPE.itemsize == 16
    assert P.HIT_DTYPE.fields["pid"][1] == 8 and P.HIT_DTYPE.fields["k"][1] == 12


@pytest.mark.parametrize("k", [0, 1, 2, 3])
@pytest.mark.parametrize("L", [7, 11, 12, 20])
def test_pick_semantics_matches_oracle(k, L):
    """pm_pick_semantics (product) vs the oracle's restatement of select.cc:31-141."""
    pats = ["ACGT" * 8][0][:L], ("TTGACA" * 6)[:L + 3]
    fam = {1: 2, 2: 2, 3: 2, 4: 4, 5: 5, 7: 8, 8: 8, 9: 8, 10: 8, 11: 12, 12: 12, 13: 12, 14: 12, -1: -6}
    for table, raw in ((b"ACGT\n", False), (b"ACGT\nN", False), (None, True)):
        text = O.Text(np.zeros(4, dtype=np.uint8), table) if not raw else O.Text(np.zeros(4, dtype=np.uint8))
        for esb, eeb in ((None, None), ([8, 8], [0, 0]), ([0, 0], [6, 7]), ([3, 3], [0, 0])):
            want = O.pick_engine(text, list(pats), k, True, esb, eeb)
            got = sat_amd.pick_semantics(256 if raw else len(table), not raw, k, list(pats), esb, eeb)
            assert got == fam[want], (k, L, table, esb, eeb, want, got)


def test_create_rejects_bad_config():
    with pytest.raises(sat_amd.PmError):
        sat_amd.PatternMatch(k=-1)


def test_missing_gpu_fails_loudly():
    """No CPU fallback: without a device pm_init must return an error, not results."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pm = sat_amd.PatternMatch(k=0)
    pm.add_pattern("ACGTACGTACGT", 1)
    with pytest.raises(sat_amd.PmError):
        pm.init(np.zeros(64, dtype=np.uint8), table=b"ACGT\n")
