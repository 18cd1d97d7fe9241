#!/usr/bin/env python3
"""A/B for VERDICT r03 item 3 (GPU box): what would the pair geometry cost as the first stage of the -k 2 plan?
Runs, on bench.py's 3 Gbp x 100k-primer workload: today's first stage (pm_edit_scan + pm_edits_verify, from a -k 2 handle)
and the measurement kernels pm_pair_edit_scan<1> / <2> (14 (field pair, displacement) tests per window on the -K 2 tables;
pm_measure_pair_edit_floor).  Writes one JSON object.    python scripts/edit_pair_floor.py [db_bases] > out.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import sat_amd  # noqa: E402

db = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
dev = torch.device("cuda", 0)
stream = bench.gen_stream(0, db, db, 24, 20260101, dev)
primers, _ = bench.make_primers(stream[: 1 << 26], 100_000, 20, 7)
allp = primers + [sat_amd.reverse_comp(p) for p in primers]
out = {"db_bases": db, "primers": 100_000, "code_sha": bench.code_sha()}


def handle(k, indels):
    pm = sat_amd.PatternMatch(k=k, indels=indels)
    for i, p in enumerate(allp):
        pm.add_pattern(p, i + 1)
    pm.init_device(stream.data_ptr(), stream.numel(), bench.TABLE, keepalive=stream)
    return pm


pm = handle(2, True)
pm.set_capacity(1 << 28)
for _ in range(3):
    pm.scan_async(0, db)
    n = pm.scan_wait()
out["today"] = {"kernels": pm.describe().split()[0], "scan_plus_verify_ms": pm.last_kernel_time()[0], "candidates": n,
                "seed_records": pm.scan_stats()["between_stages"]}
pm.close()
pm = handle(2, False)
pm.set_capacity(1 << 24)
for _ in range(2):
    pm.scan_async(0, db)
    pm.scan_wait()
out["K2_for_scale"] = {"scan_plus_verify_ms": pm.last_kernel_time()[0], "suspects": pm.scan_stats()["between_stages"]}
for mode, what in ((1, "14 tests, substitution compare of three patterns per slot (lower bound)"),
                   (2, "14 tests, five-shift edit test on two patterns per slot (cost model of the cheapest decision found)")):
    best = None
    for _ in range(3):
        ms, susp = pm.measure_pair_edit_floor(mode)
        best = ms if best is None else min(best, ms)
    out["pair_floor_%d" % mode] = {"what": what, "kernel_ms": best, "suspects": susp}
pm.close()
print(json.dumps(out, indent=1))
