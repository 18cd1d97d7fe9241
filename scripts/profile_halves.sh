#!/bin/bash
# What profiles/ keeps about exact_halves -k 1 (3 Gbp x 100k primers): the bench line with the CPU baseline, the
# kernel trace, the stage switches (1 = test stage only, 2 = + compaction, 4 = + rank and slot load, 0 = all) and the
# round-1 form behind PM_HALF_SCAN=bloom.  Usage (GPU box): bash scripts/profile_halves.sh r02
set -o pipefail
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lines_$tag
mkdir -p $out
python bench.py --steps 5 --warmup 1 --k 1 --indels 1 > $out/bench_k1_edits.json 2> $out/bench_k1_edits.err && echo "k1 edits line done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_k1e -- python bench.py --steps 3 --warmup 1 --k 1 --indels 1 --no-cpu > $out/kt_k1e.log 2>&1 && echo "trace done"
f=gpurun_out/${tag}_stages_k1_edits.txt
rm -f $f
for dbg in 1 2 4 0; do
  PM_SEED_DEBUG=$dbg python bench.py --steps 2 --warmup 1 --k 1 --indels 1 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('k=1 edits (exact_halves) debug=$dbg kernel_ms,candidates:', j['roofline']['kernel_ms'], j['config']['candidates'])" >> $f
done
PM_HALF_SCAN=bloom python bench.py --steps 2 --warmup 1 --k 1 --indels 1 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('k=1 edits PM_HALF_SCAN=bloom (round-1 form) kernel_ms,candidates:', j['roofline']['kernel_ms'], j['config']['candidates'])" >> $f
cat $f
echo "all done"
