#!/bin/bash
# pair-plan geometry sweep: positions per workgroup x chunks per (superchunk, combo) run; -K 2, 3 Gbp
out=gpurun_out/sweep_pair.txt
: > $out
for chunk in 524288 1048576 2097152 4194304; do
  for group in 128 256 512; do
    line=$(PM_SEED_CHUNK=$chunk PM_SEED_GROUP=$group python bench.py --steps 3 --warmup 1 --k ${K:-2} --no-cpu --no-check 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms'])" 2>/dev/null)
    echo "chunk=$chunk group=$group kernel_ms=$line" | tee -a $out
  done
done
