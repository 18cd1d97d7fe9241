"""GPU: the RCCL exchange of the C ABI (csrc/pm_comm.cpp) executes -- on the one GPU of this box with a
communicator of one rank, whose records still go through ncclSend / ncclRecv (the
reference has no counterpart, its scan is one serial pass, primer_match.cc:1118) -- and bench.py's
torch.distributed path on the `nccl` backend (= RCCL) gives the hits of the plain run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pm_comm_world_one_gather_equals_copy_records():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_selftest.py")], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode("latin1")[-1500:]
    assert r.stdout.decode().strip().splitlines()[-1].startswith("ok "), r.stdout          # (RCCL prints a version banner first)


@pytest.mark.parametrize("opts", [["--k", "2"], ["--k", "2", "--indels", "1"], ["--k", "0"]])
def test_bench_nccl_backend_world_one_equals_plain_run(tmp_path, opts):
    """bench.py with PM_BENCH_FORCE_DIST=1 under torch.distributed.run --nproc-per-node 1: process group
    on the nccl backend, the owned-finalize / gather code of the N > 1 path (all_gather of counts, records
    wrapped straight out of HBM, dist.gather, side-stream landing) with world == 1; the final hits must
    equal the plain single-process run's."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--db-bases", "50000000", "--primers", "20000", "--no-cpu"] + opts
    a, b = str(tmp_path / "plain.npy"), str(tmp_path / "dist.npy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--dump-hits", a], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode("latin1")[-1500:]
    plain = json.loads(r.stdout.decode().strip().splitlines()[-1])
    env2 = dict(env, PM_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py")] + common + ["--dump-hits", b],
                       capture_output=True, timeout=600, env=env2)
    assert r.returncode == 0, r.stderr.decode("latin1")[-1500:]
    dist = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert dist["config"]["exchange"]["backend"] == "nccl" and dist["config"]["exchange"]["forced_at_world_1"] is True
    assert dist["ranks_seen"] == 1
    ha, hb = np.load(a), np.load(b)
    assert ha.size > 500 and ha.tobytes() == hb.tobytes()
    assert plain["config"]["final_hits"] == dist["config"]["final_hits"]
