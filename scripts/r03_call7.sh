#!/bin/bash
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step %.2f kernel %.2f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]), d["config"]["candidates"], d["config"]["planted_found"])'
for lib in libpm_gpu.so libpm_gpu_nr8.so libpm_gpu.so libpm_gpu_nr8.so; do
PM_GPU_LIB=$PWD/sequence-alignment-tools_amd/csrc/$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "$P" "$lib"
done
