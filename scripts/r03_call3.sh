#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for dbg in 0 1 2 4 6; do
PM_SEED_DEBUG=$dbg PM_PAIR_ROW=16 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug $dbg', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['candidates'])"
done
