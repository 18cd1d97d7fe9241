"""Where a step of the position-sharded -K 2 path spends its time on one rank (a 375 Mbp shard = 3 Gbp over 8 ranks):
scan launch, scan wait, owned finalize (clusters on the GPU, hits left in HBM), wrap + copy to pinned host memory.
Run on the GPU box: python scripts/step_phases_owned.py [shard bases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sat_amd, bench
dev = torch.device("cuda", 0)
shard = int(sys.argv[1]) if len(sys.argv) > 1 else 375_000_000
total = shard * 2
G, H = bench.GUARD, bench.HALO
lo, hi = shard // 2, shard // 2 + shard                                # an inner shard: guard bands both sides
glo, ghi = lo - G - H, hi + G + H
stream = bench.gen_stream(glo, ghi, total, 24, 20260101, dev)
primers, _ = bench.make_primers(stream[:1 << 24], 100000, 20, 7)
allp = primers + [sat_amd.reverse_comp(p) for p in primers]
pm = sat_amd.PatternMatch(k=2, indels=False, device=0)
for i, p in enumerate(allp): pm.add_pattern(p, i + 1)
pm.init_device(stream.data_ptr(), stream.numel(), bench.TABLE, stream=torch.cuda.current_stream().cuda_stream, keepalive=stream)
pm.set_capacity(1 << 24)
begin, end = lo - glo, hi - glo
g_lo, g_hi = begin - G, end + G
pin = torch.empty((1 << 24) * 2, dtype=torch.int64, pin_memory=True)
acc = np.zeros(5)
N = 20
for it in range(N + 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pm.scan_async(g_lo, g_hi); t1 = time.perf_counter()
    n = pm.scan_wait(); t2 = time.perf_counter()
    ptr, cnt = pm.finalize_device(0, sort=False, owned=(begin, end, g_lo, g_hi), keep=True); t3 = time.perf_counter()
    if cnt:
        pin[:cnt * 2].copy_(torch.as_tensor(bench.CudaArray(ptr, cnt * 16), device=dev).view(torch.int64), non_blocking=True)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if it >= 3: acc += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]) * 1e3
acc /= N
print("shard %d: launch %.3f ms, wait %.3f ms (kernels %.3f), owned finalize %.3f ms, copy of %d hits %.3f ms, step %.3f ms" %
      (shard, acc[0], acc[1], pm.last_kernel_time()[0], acc[2], cnt, acc[3], acc[4]))
