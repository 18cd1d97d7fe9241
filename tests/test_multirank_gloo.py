"""CPU, world size 2 (gloo): the multi-GPU data path of bench.py -- position sharding with a
halo, count all_gather, padded record gather to rank 0, local->global index fix-up -- exercised
with the oracle standing in for the device stage (no GPU here).  The merged, finalized result must
equal a single-rank scan."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from oracle import pmoracle as O

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(11)
ents = synth.make_entries(rng, 4, 3000, n_runs=2, repeats=True, short=True)
pats = synth.make_patterns(rng, ents, 150, length=20, planted=0.7)
allp = pats + [synth.revcomp(p) for p in pats]
table = synth.table_for(ents)
codes = synth.normalize(synth.stream(ents), table)
total = codes.size
HALO = 64
shard = (total + world - 1) // world
lo, hi = rank * shard, min(total, (rank + 1) * shard)
glo = max(0, lo - HALO)
local = codes[glo:hi]
K = 2
# device stage stand-in: candidates of the local buffer, ownership filter begin < end <= end_r
c = O.find_all(O.Text(local, table), allp, engine=O.SHIFT_AND_INEXACT, k=K, indels=False)
begin, end = lo - glo, hi - glo
c = c[(c["end"] > begin) & (c["end"] <= end)]
if glo > 0:      # a rank that does not start at stream index 0 must not keep stream-start records
    pass
cnt = torch.tensor([c.size], dtype=torch.int64)
counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(counts, cnt)
mx = max(int(x.item()) for x in counts)
pad = torch.zeros(max(mx, 1) * 16, dtype=torch.uint8)
if c.size:
    pad[:c.size * 16] = torch.from_numpy(c.view(np.uint8).copy())
gathered = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
dist.gather(pad, gathered, dst=0)
if rank == 0:
    parts = []
    for r in range(world):
        a = gathered[r][:int(counts[r].item()) * 16].numpy().view(O.HIT_DTYPE).copy()
        a["end"] += max(0, r * shard - HALO)
        parts.append(a)
    merged = np.concatenate(parts)
    # reference answer: candidates of the whole stream
    whole = O.find_all(O.Text(codes, table), allp, engine=O.SHIFT_AND_INEXACT, k=K, indels=False)
    assert O.sorted_tuples(merged) == O.sorted_tuples(whole), "sharded candidates differ"
    print("OK", merged.size)
dist.destroy_process_group()
'''


# filter_bitvec: every rank clusters + verifies the chains whose hit it owns (its slice + a guard band
# + halo is all the text it needs) and only final hits are gathered, in stream order
WORKER_OWNED = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from oracle import pmoracle as O

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(12)
ents = synth.make_entries(rng, 4, 3000, n_runs=2, repeats=True, short=True)
pats = synth.make_patterns(rng, ents, 150, length=20, planted=0.7)
allp = pats + [synth.revcomp(p) for p in pats]
table = synth.table_for(ents)
codes = synth.normalize(synth.stream(ents), table)
total = codes.size
HALO, GUARD = 64, 512
shard = (total + world - 1) // world
lo, hi = rank * shard, min(total, (rank + 1) * shard)
glo, ghi = max(0, lo - GUARD - HALO), min(total, hi + GUARD + HALO)
for K, indels in ((2, False), (2, True), (1, True)):
    # device stage + owned finalize stand-in: the engine over the local buffer, hits kept by ownership
    h = O.find_all(O.Text(codes[glo:ghi], table), allp, engine=5, k=K, indels=indels)
    h = h[(h["end"] > lo - glo) & (h["end"] <= hi - glo)]
    cnt = torch.tensor([h.size], dtype=torch.int64)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, cnt)
    mx = max(int(x.item()) for x in counts)
    pad = torch.zeros(max(mx, 1) * 16, dtype=torch.uint8)
    if h.size:
        pad[:h.size * 16] = torch.from_numpy(h.view(np.uint8).copy())
    gathered = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, gathered, dst=0)
    if rank == 0:
        parts = []
        for r in range(world):
            a = gathered[r][:int(counts[r].item()) * 16].numpy().view(O.HIT_DTYPE).copy()
            a["end"] += max(0, r * shard - GUARD - HALO)
            parts.append(a)
        merged = np.concatenate(parts)
        whole = O.find_all(O.Text(codes, table), allp, engine=5, k=K, indels=indels)
        assert O.sorted_tuples(merged) == O.sorted_tuples(whole) and whole.size, "owned shards differ from the single scan"
        print("OK", K, indels, merged.size)
dist.destroy_process_group()
'''


def _run(tmp_path, worker):
    script = tmp_path / "worker.py"
    script.write_text("ROOT = %r\n" % ROOT + worker)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return out.stdout


def test_two_rank_owned_finalize_gather(tmp_path):
    assert _run(tmp_path, WORKER_OWNED).count("OK") == 3


def test_two_rank_shard_gather_merge(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text("ROOT = %r\n" % ROOT + WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "OK" in out.stdout
