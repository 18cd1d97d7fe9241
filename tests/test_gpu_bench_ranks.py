"""GPU: the multi-rank path that ships -- bench.py itself -- rehearsed with two ranks on one card.

SURVEY.md 8(e): the stream is sharded by position, every rank scans its shard (+ guard band), the
final hit records (filter_bitvec option sets: clustered and verified on the owning rank) or the
candidate records (the rest) are gathered to rank 0.  Here bench.py is launched exactly as the
driver launches it (`python -m torch.distributed.run ... bench.py --gpus 2`), with the collectives
on gloo because the box has one GPU (PM_BENCH_BACKEND=gloo; with RCCL the same code gathers device
buffers), and its final hit list must equal the single-rank run's, which in turn must hold every
planted primer site (bench.py checks that itself and fails otherwise).  The serial rule the shards
must reproduce: filter_bitvec.cc:88-177 (one hit per chain of candidates), exact_halves.cc:140-190.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = 50_000_000
PRIMERS = 20_000


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(ranks, k, indels, dump, extra=()):
    common = ["bench.py", "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--db-bases", str(DB), "--primers", str(PRIMERS),
              "--k", str(k), "--indels", str(indels), "--no-cpu", "--dump-hits", dump, *extra]     # default scaling: strong
    env = dict(os.environ, PM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ranks == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + common
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), np.load(dump)


@pytest.mark.parametrize("k,indels", [(2, 0), (2, 1), (0, 0), (1, 0), (1, 1)])
def test_two_ranks_equal_one_rank(tmp_path, k, indels):
    j1, h1 = run_bench(1, k, indels, str(tmp_path / "one.npy"))
    j2, h2 = run_bench(2, k, indels, str(tmp_path / "two.npy"))
    assert j1["ranks_seen"] == 1 and j2["ranks_seen"] == 2 and j2["n_gpus"] == 2
    assert j1["config"]["planted_found"] is not None and j2["config"]["planted_found"] == j1["config"]["planted_found"]
    assert h1.size > 0
    assert h1.size == h2.size and (h1 == h2).all(), (h1.size, h2.size)
    assert j2["config"]["final_hits"] == h2.size
    # BASELINE.json's metric is one database "at 1/2/4/8 GPUs": the default N > 1 line reports the SAME total stream,
    # split over the ranks, and says so; it also carries the exchange's cost and every rank's kernel time
    assert j1["scaling"] == j2["scaling"] == "strong"
    assert j1["config"]["db_bases_total"] == j2["config"]["db_bases_total"] == DB
    assert j2["config"]["db_bases_per_gpu"] == DB // 2 and j1["config"]["db_bases_per_gpu"] == DB
    assert "0.05 Gbp stream in total = 0.025 Gbp per GPU x 2 GPU(s)" in j2["config"]["workload"], j2["config"]["workload"]
    assert j2["exchange_ms"] is not None and j2["exchange_ms"] >= 0 and len(j2["kernel_ms_per_rank"]) == 2
    assert all(x > 0 for x in j2["kernel_ms_per_rank"])


def test_default_workload_is_the_three_gbp_database_in_total():
    """no flags but --gpus: the line would name 3 Gbp in total (argument defaults, read without running)"""
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--db-bases", type=int, default=3_000_000_000', src)
    assert re.search(r'"--scaling", choices=\["weak", "strong"\], default="strong"', src)


@pytest.mark.parametrize("k,indels", [(2, 0), (0, 0)])
def test_two_ranks_grow_and_rescan_after_overflow(tmp_path, k, indels):
    """a record buffer that is too small on every rank: the ranks agree to grow it and scan again
    (no rank is left waiting in the gather), and the result does not change"""
    j1, h1 = run_bench(1, k, indels, str(tmp_path / "one.npy"))
    j2, h2 = run_bench(2, k, indels, str(tmp_path / "two.npy"), extra=("--capacity", "64"))
    assert j2["config"]["rescans_after_overflow"] >= 1
    assert h1.size == h2.size and (h1 == h2).all()
    j3, h3 = run_bench(1, k, indels, str(tmp_path / "three.npy"), extra=("--capacity", "64"))
    assert j3["config"]["rescans_after_overflow"] >= 1 and (h1 == h3).all()


@pytest.mark.parametrize("k,indels", [(2, 0), (2, 1)])
def test_two_ranks_survive_a_chain_cut_by_the_guard_band(tmp_path, k, indels):
    """a homopolymer run of 200,000 A's across the shard edge (the guard band is 64 Ki) with the primer
    A x 20 in the set -- a chain of same-pattern candidates that no shard can decide on its own (filter_bitvec.cc:103-121:
    one hit per chain).  The owned finalize says so (PM_E_UNSUPPORTED), the ranks tell each other through the count
    exchange and take the gather-candidates form for that step, as GpuPatternMatch::sharded_scan does; the hits equal the
    one-rank run's."""
    extra = ("--plant-run", "200000", "--entries", "25")           # (25 entries: no end-of-sequence character at the shard edge, which would cut the run in two)
    j1, h1 = run_bench(1, k, indels, str(tmp_path / "one.npy"), extra=extra)
    j2, h2 = run_bench(2, k, indels, str(tmp_path / "two.npy"), extra=extra)
    assert j2["config"]["steps_with_a_cut_chain"] == 3 and j1["config"]["steps_with_a_cut_chain"] == 0     # warmup + 2 steps
    assert h1.size > 0 and h1.size == h2.size and (h1 == h2).all(), (h1.size, h2.size)
    run = h1[h1["pid"] == PRIMERS]                                  # the poly-A primer (the last forward primer)
    assert run.size == 1, run                                       # the whole run is ONE chain, one hit


@pytest.mark.parametrize("style", ["skew", "vocab", "tandem"])
def test_two_ranks_equal_one_rank_on_hard_streams(tmp_path, style):
    """bench.py --stream-style: low-complexity streams (tests/adversarial.py's generators at database size), primers cut
    from the stream; two position shards still give the one-rank hits (long chains, dense key hits, buffer growth)"""
    extra = ("--stream-style", style, "--db-bases", "20000000", "--primers", "2000")
    j1, h1 = run_bench(1, 2, 0, str(tmp_path / "one.npy"), extra=extra)
    j2, h2 = run_bench(2, 2, 0, str(tmp_path / "two.npy"), extra=extra)
    assert j1["config"]["stream_style"] == style
    assert h1.size > 0 and h1.size == h2.size and (h1 == h2).all(), (style, h1.size, h2.size)
