"""CPU: the command lines' rank launcher and host transport (host/pm_ranks.cc) -- fork N ranks,
all_gather of counts, broadcast, gather of hit records in rank order -- without a GPU.  (The record
path over RCCL is behind the C ABI, pm_comm_*; on a GPU box the multi-rank command lines are run by
tests/test_gpu_ranks_cli.py.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sequence-alignment-tools_amd", "host", "pm_ranks_selftest")


@pytest.mark.parametrize("ranks", [1, 2, 5])
def test_launcher_and_pipes(ranks):
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    r = subprocess.run([EXE, "--ranks", str(ranks)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok world=%d" % ranks), r.stdout


def test_ranks_from_the_environment_and_a_dying_rank():
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=60, env=dict(os.environ, PM_RANKS="3"))
    assert r.returncode == 0 and r.stdout.startswith("ok world=3"), (r.stdout, r.stderr)


def test_a_dead_rank_ends_the_others():
    """a rank that exits non-zero while its peers wait on a transfer that no pipe will end (the RCCL transport): the
    launcher signals the survivors and exits with the dead rank's status, at once -- it used to wait for rank 0 forever"""
    import time
    for dead in (0, 2):
        t0 = time.time()
        r = subprocess.run([EXE, "--ranks", "4", "--die", str(dead)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 3, (r.returncode, r.stderr)
        assert time.time() - t0 < 20
