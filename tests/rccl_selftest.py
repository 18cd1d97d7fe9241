"""Run by tests/test_gpu_rccl.py in a process of its own (a hang in RCCL must not take the test
session with it): the C ABI's exchange step -- pm_comm_unique_id -> pm_comm_create(world = 1) ->
pm_comm_gather on real scan records -> pm_comm_destroy (csrc/pm_comm.cpp) -- on the one
GPU of this box.  With world = 1 rank 0's own records still travel through ncclSend / ncclRecv (to
itself, inside one group), so symbol resolution, communicator init, group start / end, the transfers
out of HBM and the device -> host tail all execute.  Prints "ok <records>"."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sat_amd  # noqa: E402
import synth  # noqa: E402


def main():
    rng = np.random.default_rng(123)
    ents = synth.make_entries(rng, 4, 250_000, n_runs=2, repeats=True, short=True)
    pats = synth.make_patterns(rng, ents, 4000, length=20, planted=0.5)
    allp = pats + [synth.revcomp(p) for p in pats]
    table = synth.table_for(ents)
    codes = synth.normalize(synth.stream(ents), table)
    dev = torch.from_numpy(codes).to("cuda:0")
    pm = sat_amd.PatternMatch(k=2, indels=False)
    for i, p in enumerate(allp):
        pm.add_pattern(p, i + 1)
    pm.init_device(dev.data_ptr(), dev.numel(), table, keepalive=dev)
    n = pm.scan_candidates(0, dev.numel(), to_host=False)
    ptr, n2 = pm.candidates_device()
    assert n == n2 and n > 1000, n
    want = pm.copy_records(ptr, n)

    L = sat_amd.load_library()
    L.pm_comm_last_error.restype = C.c_char_p
    L.pm_comm_last_error.argtypes = [C.c_void_p]
    L.pm_comm_destroy.restype = None
    L.pm_comm_destroy.argtypes = [C.c_void_p]
    L.pm_comm_unique_id.argtypes = [C.c_void_p]
    L.pm_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.pm_comm_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    uid = (C.c_uint8 * 128)()
    rc = L.pm_comm_unique_id(uid)
    assert rc == 0, (rc, L.pm_comm_last_error(None))
    assert any(uid), "unique id is all zero"
    comm = C.c_void_p()
    rc = L.pm_comm_create(0, 0, 1, uid, C.byref(comm))
    assert rc == 0, (rc, L.pm_comm_last_error(None))
    for rep in range(2):                                   # the second call reuses the landing buffer
        counts = (C.c_uint64 * 1)(n)
        got = np.zeros(n, dtype=sat_amd.HIT_DTYPE)
        rc = L.pm_comm_gather(comm, C.c_void_p(ptr), n, counts, got.ctypes.data_as(C.c_void_p))
        assert rc == 0, (rc, L.pm_comm_last_error(comm))
        assert got.tobytes() == want.tobytes(), "gathered records differ from pm_copy_records"
    # argument errors are reported, not executed
    bad = (C.c_uint64 * 1)(n + 1)
    assert L.pm_comm_gather(comm, C.c_void_p(ptr), n, bad, None) != 0
    L.pm_comm_destroy(comm)
    pm.close()
    print("ok %d" % n)


if __name__ == "__main__":
    main()
