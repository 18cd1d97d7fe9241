#!/bin/bash
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "value %.1f step %.2f kernel %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]), d["config"]["final_hits"], d["config"]["planted_found"])'
for o in "--k 2" "--k 1" "--k 0" "--k 1 --indels 1" "--k 2 --indels 1" "--k 2 --primers 1000000"; do
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu $o 2>/dev/null | python -c "$P" "$o"
done
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_bench_ranks.py tests/test_gpu_rccl.py 2>&1 | tail -3
