import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sat_amd, bench
dev = torch.device("cuda", 0)
total = 1_000_000_000
stream = bench.gen_stream(0, total, total, 24, 20260101, dev)
primers = bench.make_primers(stream[:1 << 26], 100000, 20, 7)
allp = primers + [sat_amd.reverse_comp(p) for p in primers]
pm = sat_amd.PatternMatch(k=2, indels=False, device=0)
for i, p in enumerate(allp): pm.add_pattern(p, i + 1)
pm.init_device(stream.data_ptr(), stream.numel(), bench.TABLE, stream=torch.cuda.current_stream().cuda_stream, keepalive=stream)
pm.set_capacity(1 << 24)
pm.scan_async(0, total); pm.scan_wait()
ptr, cnt = pm.candidates_device()
rec = torch.as_tensor(bench.CudaArray(ptr, cnt * 16), device=dev).view(torch.int64).view(-1, 2).clone()
print("records per Gbp", cnt)
parts = []
for r in range(24):
    a = rec.clone(); a[:, 0] += r * total; parts.append(a)
allrec = torch.cat(parts).contiguous(); tot = allrec.shape[0]
torch.cuda.synchronize()
for name, buf in [("pageable", np.empty(1 << 25, dtype=sat_amd.HIT_DTYPE)),
                  ("pinned", torch.empty((1 << 25) * 16, dtype=torch.uint8, pin_memory=True).numpy().view(sat_amd.HIT_DTYPE))]:
    for it in range(3):
        t0 = time.perf_counter()
        out = pm.finalize_device(24 * total, last=True, sort=False, d_cands=allrec.data_ptr(), n=tot, out=buf)
        dt = time.perf_counter() - t0
    print(name, tot, "records ->", out.size, "hits: %.2f ms" % (dt * 1e3))
