#!/bin/bash
# Everything profiles/ keeps about the edit-distance plan (-k 2, 3 Gbp x 100k primers): the bench line with the CPU
# baseline, the kernel trace, the stage switches, L2 hit/miss of the scan kernel (pm_edit_scan, and the round-1 first
# stage behind PM_EDIT_SCAN=bloom for comparison) and the SQ instruction-mix counters (1 Gbp).
# Usage (GPU box): bash scripts/profile_edits.sh r02
set -o pipefail
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lines_$tag
mkdir -p $out
python bench.py --steps 3 --warmup 1 --k 2 --indels 1 > $out/bench_k2_edits.json 2> $out/bench_k2_edits.err && echo "k2 edits line done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_k2e -- python bench.py --steps 2 --warmup 1 --k 2 --indels 1 --no-cpu > $out/kt_k2e.log 2>&1 && echo "trace done"
rm -f gpurun_out/edits_stages.txt
bash scripts/edits_stages.sh > /dev/null 2>&1
PM_EDIT_SCAN=bloom python bench.py --steps 2 --warmup 1 --k 2 --indels 1 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('k=2 edits PM_EDIT_SCAN=bloom (round-1 first stage) kernel_ms,candidates:', j['roofline']['kernel_ms'], j['config']['candidates'])" >> gpurun_out/edits_stages.txt
cp gpurun_out/edits_stages.txt gpurun_out/${tag}_stages_k2_edits.txt && echo "stages done"
{ bash scripts/pmc_l2_edit.sh pm_edit_scan; PM_EDIT_SCAN=bloom bash scripts/pmc_l2_edit.sh round1_first_stage; } > gpurun_out/${tag}_pmc_l2_k2_edits.txt 2>&1 && echo "l2 done"
bash scripts/pmc_sq.sh ${tag}k2e --k 2 --indels 1 > gpurun_out/${tag}_sq_k2_edits.txt 2>&1 && echo "sq done"
echo "all done"
