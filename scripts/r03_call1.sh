#!/bin/bash
# round-3 GPU call 1: TCP gather probe, the new / changed tests, one headline bench line
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo skip probe
timeout -k 10 900 python -m pytest -x -q -m gpu "tests/test_gpu_rccl.py::test_bench_nccl_backend_world_one_equals_plain_run[opts2]" tests/test_gpu_parity.py::test_internal_suspect_buffer_overflow_rescans_inside_the_library \
   tests/test_gpu_ranks_cli.py "tests/test_gpu_bench_ranks.py::test_two_ranks_equal_one_rank" tests/test_gpu_fullsize.py::test_config4_one_million_primers_three_gbp \
   tests/test_gpu_fullsize.py::test_edit_distance_three_gbp_100k_primers tests/test_gpu_config5_pcr.py --durations=15 > gpurun_out/call1_tests.txt 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/call1_tests.txt
timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/call1_bench_K2.json 2> gpurun_out/call1_bench_K2.err; echo "bench rc=$?"
cat gpurun_out/call1_bench_K2.json | cut -c1-600
