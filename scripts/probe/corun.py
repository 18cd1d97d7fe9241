#!/usr/bin/env python3
"""Do small kernels on a second stream run beside the persistent pair scan kernel (104 VGPRs x 16 waves and 128 KiB of LDS
per CU), or wait for it?  Decides whether a second candidate buffer would hide pm_scan's finalize kernels behind the next
range's scan.  Usage (GPU box): python scripts/probe/corun.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import sat_amd

n, P, L = 1 << 30, 100_000, 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
dev = torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda", generator=g)
dev[0] = 4; dev[-1] = 4
rng = np.random.default_rng(3)
pats = ["".join("ACGT"[c] for c in rng.integers(0, 4, L)) for _ in range(P)]
allp = pats + [sat_amd.reverse_comp(p) for p in pats]
pm = sat_amd.PatternMatch(k=2, indels=False, kernel=sat_amd.KERNEL_SEED)
for i, p in enumerate(allp):
    pm.add_pattern(p, i + 1)
side = torch.cuda.Stream()
pm.init_device(dev.data_ptr(), n, b"ACGT\n", keepalive=dev)
pm.set_capacity(1 << 24)
pm.scan_candidates(0, n, to_host=False)                       # warm
keys = torch.randint(0, 1 << 40, (80_000,), dtype=torch.int64, device="cuda")


def small_work():
    """about what a finalize does: two sorts of 80k keys and a few elementwise kernels"""
    a = torch.sort(keys).values
    b = (a[1:] - a[:-1] > 5).nonzero()
    c = torch.sort(a ^ 12345).values
    return b.numel() + c.numel()


with torch.cuda.stream(side):
    small_work()
torch.cuda.synchronize()
for label, with_scan in (("alone", False), ("beside the scan", True), ("alone", False), ("beside the scan", True)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if with_scan:
        pm.scan_async(0, n)                                     # 1 Gbp: about 5 ms on the handle's own stream
        time.sleep(0.001)                                       # the scan is running
    with torch.cuda.stream(side):
        e0.record()
        small_work()
        e1.record()
    e1.synchronize()
    t1 = time.perf_counter()
    if with_scan:
        pm.scan_wait()
    t2 = time.perf_counter()
    print("%-16s small kernels: %.3f ms on their stream, host saw them done after %.3f ms; scan done after %.3f ms (kernel %.3f ms)"
          % (label, e0.elapsed_time(e1), (t1 - t0) * 1e3, (t2 - t0) * 1e3, pm.last_kernel_time()[0] if with_scan else 0.0))
