#!/bin/bash
# Where the pair-plan kernel's time goes: PM_SEED_DEBUG stage switches (pm_pair.hip), same workload.
#   12 = tests + direct-table loads that all read entry 0 + consume | 4 = real direct-table loads, no compaction
#   1 = + compaction, no drain | 2 = + drain without the exact-table load | 0 = all
out=gpurun_out/pair_stages.txt
: > $out
for k in ${KS:-1 2}; do
  for dbg in ${DBGS:-4 1 2 0}; do
    line=$(PM_SEED_DEBUG=$dbg python bench.py --steps 3 --warmup 1 --k $k --no-cpu --no-check --db-bases ${DB:-3000000000} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms'], j['config']['candidates'])" 2>/dev/null)
    echo "k=$k debug=$dbg kernel_ms,candidates: $line" | tee -a $out
  done
done
