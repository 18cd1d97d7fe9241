#!/usr/bin/env python3
"""pm_scan on hit-dense text at database size: a stream of bench.py's hard styles, every primer cut from it,
scanned in 256 MiB ranges; the library scans a range in pieces where its record lists would outgrow its bound
(pm_api.cpp scan_range).  Prints one JSON line: time, hits, halvings of the piece length; with --compare-bound B a second pass with that bound
(PM_DENSE_BOUND, read in pm_create) must give the same hits.
    python scripts/dense_split.py --style tandem --k 2 --indels 1 --db-bases 300000000"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import sat_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--style", default="tandem")
ap.add_argument("--k", type=int, default=2)
ap.add_argument("--indels", type=int, default=1)
ap.add_argument("--db-bases", type=int, default=300_000_000)
ap.add_argument("--primers", type=int, default=100_000)
ap.add_argument("--chunk", type=int, default=1 << 28)
ap.add_argument("--bound", type=int, default=0)
ap.add_argument("--compare-bound", type=int, default=0)
ap.add_argument("--passes", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda", 0)
stream = bench.gen_stream(0, args.db_bases, args.db_bases, 24, 20260101, dev, args.style)
primers, _ = bench.make_primers(stream[:min(stream.numel(), 1 << 26)], args.primers, 20, 7, 1.0)
allp = primers + [sat_amd.reverse_comp(p) for p in primers]


def one(bound):
    if bound:
        os.environ["PM_DENSE_BOUND"] = str(bound)
    else:
        os.environ.pop("PM_DENSE_BOUND", None)
    pm = sat_amd.PatternMatch(k=args.k, indels=bool(args.indels))
    for i, p in enumerate(allp):
        pm.add_pattern(p, i + 1)
    pm.init_device(stream.data_ptr(), stream.numel(), bench.TABLE, keepalive=stream)
    n = stream.numel()
    times = []
    for _ in range(max(1, args.passes)):                              # the first pass pins the landing buffer and sizes the lists; the later ones find them
        pm.reset()
        t0 = time.perf_counter()
        pos, parts, tick = 0, [], time.time()
        while pos < n:
            e = min(n, pos + args.chunk)
            parts.append(pm.scan_view(pos, e).copy())
            pos = e
            if time.time() - tick > 30:
                print("[dense_split] at %d of %d, %d hits" % (pos, n, sum(p.size for p in parts)), file=sys.stderr, flush=True)
                tick = time.time()
        times.append((time.perf_counter() - t0) * 1e3)
    ms = times
    st = pm.scan_stats()
    desc = pm.describe()
    pm.close()
    return np.concatenate(parts), ms, st, desc


hits, ms, st, desc = one(args.bound)
res = {"style": args.style, "k": args.k, "indels": args.indels, "db_bases": args.db_bases, "primers": args.primers, "chunk": args.chunk,
       "pm_scan_ms_per_pass": ms, "final_hits": int(hits.size), "range_cuts": st["range_splits"], "internal_rescans": st["internal_rescans"],
       "bound": args.bound or (1 << 30), "plan": desc[:120], "hbm_in_use_at_the_end_gb": torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9}
if args.compare_bound:
    h2, ms2, st2, _ = one(args.compare_bound)
    same = h2.size == hits.size and bool((h2["end"] == hits["end"]).all() and (h2["pid"] == hits["pid"]).all() and (h2["k"] == hits["k"]).all())
    res["compare"] = {"bound": args.compare_bound, "pm_scan_ms_per_pass": ms2, "range_cuts": st2["range_splits"], "same_hits": same}
print(json.dumps(res))
sys.exit(0 if (not args.compare_bound or res["compare"]["same_hits"]) else 1)
