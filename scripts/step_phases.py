import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sat_amd, bench
dev = torch.device("cuda", 0)
total = 3_000_000_000
stream = bench.gen_stream(0, total, total, 24, 20260101, dev)
primers = bench.make_primers(stream[:1 << 26], 100000, 20, 7)
allp = primers + [sat_amd.reverse_comp(p) for p in primers]
for k, indels in [(1, True), (1, False)]:
    pm = sat_amd.PatternMatch(k=k, indels=indels, device=0)
    for i, p in enumerate(allp): pm.add_pattern(p, i + 1)
    pm.init_device(stream.data_ptr(), stream.numel(), bench.TABLE, stream=torch.cuda.current_stream().cuda_stream, keepalive=stream)
    pm.set_capacity(1 << 24)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pm.scan_async(0, total); t1 = time.perf_counter()
        n = pm.scan_wait(); t2 = time.perf_counter()
        ptr, cnt = pm.candidates_device()
        cands = torch.as_tensor(bench.CudaArray(ptr, cnt * 16), device=dev).cpu().numpy().view(sat_amd.HIT_DTYPE); t3 = time.perf_counter()
        pm.reset(); hits = pm.finalize(cands, total, last=True, sort=False); t4 = time.perf_counter()
    print("k=%d indels=%s: launch %.2f ms, wait %.2f ms (kernel %.2f), D2H %d recs %.2f ms, host finalize %.2f ms -> %d hits" %
          (k, indels, (t1-t0)*1e3, (t2-t1)*1e3, pm.last_kernel_time()[0], cnt, (t3-t2)*1e3, (t4-t3)*1e3, hits.size))
    pm.close()
