#!/bin/bash
# stage switches of the edit-distance scan kernel (pm_edit_scan in pm_seed.hip): 1 = test + consume stages only, 2 = + compaction
# of the suspicious windows, 4 = + bucket loads without the q-gram test / records, 0 = all (with the verify kernel)
for dbg in 1 2 4 0; do
  line=$(PM_SEED_DEBUG=$dbg python bench.py --steps 2 --warmup 1 --k 2 --indels 1 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms'], j['config']['candidates'])" 2>/dev/null)
  echo "k=2 edits ${PM_EDIT_TABLE_LOG:+map 2^$PM_EDIT_TABLE_LOG }debug=$dbg kernel_ms,candidates: $line" | tee -a gpurun_out/edits_stages.txt
done
