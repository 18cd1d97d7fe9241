#!/bin/bash
# last GPU call of round 3: the whole -m gpu suite on the final code, then the -k 2 bench line and kernel stats again (pm_dedup_unpack changed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r03 gpurun_out/lines_r03
timeout -k 10 900 python -m pytest -x -q -m gpu tests/ > gpurun_out/fullsuite.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/fullsuite.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/kt_k2_edits -- python bench.py --steps 3 --warmup 1 --no-cpu --k 2 --indels 1 > gpurun_out/prof_r03/kt_k2_edits.log 2>&1; echo "kernel-trace k2_edits rc=$?"
python bench.py --steps 3 --warmup 1 --k 2 --indels 1 > gpurun_out/lines_r03/bench_k2_edits.json 2> gpurun_out/lines_r03/bench_k2_edits.err && echo "k2 edits line done"
