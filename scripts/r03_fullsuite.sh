#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 scripts/probe/valu_rate > gpurun_out/valu_rate.txt 2>&1
timeout -k 10 1100 python -m pytest -x -q -m gpu tests/ --durations=25 > gpurun_out/fullsuite.txt 2>&1
echo "tests rc=$?"; tail -40 gpurun_out/fullsuite.txt
