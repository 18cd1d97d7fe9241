#!/bin/bash
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "value %.1f step %.2f kernel %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]), d["config"]["final_hits"], d["config"]["planted_found"])'
for lib in libpm_gpu.so libpm_gpu_b8.so libpm_gpu.so libpm_gpu_b8.so; do
PM_GPU_LIB=$PWD/sequence-alignment-tools_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --k 2 --indels 1 2>/dev/null | python -c "$P" "$lib"
done
PM_GPU_LIB=$PWD/sequence-alignment-tools_amd/csrc/libpm_gpu_b8.so timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_exhaustive.py tests/test_gpu_parity.py -k "edit or edits" 2>&1 | tail -2
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_exhaustive.py tests/test_gpu_parity.py -k "edit or edits" 2>&1 | tail -2
