#!/bin/bash
# round-3 profiles, stage 1: probes, kernel-trace stats of every bench option set, PMC passes (fabric traffic, pipe counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r03
for p in tcp_gather pipe_overlap valu_rate pack_rate; do timeout -k 10 120 scripts/probe/$p > gpurun_out/prof_r03/probe_$p.txt 2>&1; echo "probe $p rc=$?"; done
kt() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/kt_$name -- python bench.py --steps 3 --warmup 1 --no-cpu "$@" > gpurun_out/prof_r03/kt_$name.log 2>&1; echo "kernel-trace $name rc=$?"; }
kt K2 --k 2
kt K0 --k 0
kt K1 --k 1
kt k1_edits --k 1 --indels 1
kt k2_edits --k 2 --indels 1
kt K2_1M --k 2 --primers 1000000
bash scripts/pmc_traffic.sh r03 K2 pm_pair_scan --k 2 > gpurun_out/prof_r03/traffic_K2.log 2>&1; echo "traffic K2 rc=$?"
bash scripts/pmc_traffic.sh r03 K1 pm_pair_scan --k 1 > gpurun_out/prof_r03/traffic_K1.log 2>&1; echo "traffic K1 rc=$?"
bash scripts/pmc_issue.sh r03 K2 pm_pair_scan --k 2 > gpurun_out/prof_r03/issue_K2.log 2>&1; echo "issue K2 rc=$?"
bash scripts/pmc_issue.sh r03 K1 pm_pair_scan --k 1 > gpurun_out/prof_r03/issue_K1.log 2>&1; echo "issue K1 rc=$?"
bash scripts/pmc_issue.sh r03 k2_edits pm_edit_scan --k 2 --indels 1 > gpurun_out/prof_r03/issue_k2_edits.log 2>&1; echo "issue k2 edits rc=$?"
bash scripts/pmc_issue.sh r03 k0 pm_seed_scan --k 0 > gpurun_out/prof_r03/issue_k0.log 2>&1; echo "issue k0 rc=$?"
bash scripts/pmc_issue.sh r03 k1_edits pm_half_scan --k 1 --indels 1 > gpurun_out/prof_r03/issue_k1_edits.log 2>&1; echo "issue k1 edits rc=$?"
ls gpurun_out/*r03*json
