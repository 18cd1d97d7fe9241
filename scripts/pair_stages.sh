#!/bin/bash
# Where the pair-plan kernels' time goes: PM_SEED_DEBUG stage switches (pm_pair.hip), same workload.
#   0 = all | 1 = suspects counted, not queued | 16 = pm_pair_verify writes no records | 32 = pm_pair_verify skips its resolve stage
# (read the kernels' own times from rocprofv3 --kernel-trace; kernel_ms brackets scan + verify)
out=gpurun_out/pair_stages.txt
: > $out
for k in ${KS:-1 2}; do
  for dbg in ${DBGS:-0 1 16 32}; do
    line=$(PM_SEED_DEBUG=$dbg python bench.py --steps 3 --warmup 1 --k $k --no-cpu --no-check --db-bases ${DB:-3000000000} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms'], j['config']['candidates'])" 2>/dev/null)
    echo "k=$k debug=$dbg kernel_ms,candidates: $line" | tee -a $out
  done
done
