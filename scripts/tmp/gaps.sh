#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaps && mkdir -p gpurun_out/gaps
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/gaps -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/gaps/log.txt 2>&1
echo rc=$?
ls gpurun_out/gaps/*/ | head
