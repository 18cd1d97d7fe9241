#!/bin/bash
# A/B runs of the headline bench line: library variants (PM_GPU_LIB) x launch knobs.  Usage (GPU box): bash scripts/sweep_pair2.sh
cd $GRAFT_REPO_ROOT
L=sequence-alignment-tools_amd/csrc
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu --steps 5 --warmup 2 --scan-passes 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'value %.1f step %.2f kernel %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
}
run base X=1
run base_again X=1
for g in 128 512 1024 2048; do run group$g PM_SEED_GROUP=$g; done
run row11 PM_PAIR_ROW=11
run row11_g1024 PM_PAIR_ROW=11 PM_SEED_GROUP=1024
run chunk4M PM_SEED_CHUNK=4194304
run chunk1M PM_SEED_CHUNK=1048576
for v in $L/libpm_gpu_*.so; do
  n=$(basename $v .so)
  run $n PM_GPU_LIB=$PWD/$v PM_GPU_LIB_AB=1
  run ${n}_g1024 PM_GPU_LIB=$PWD/$v PM_GPU_LIB_AB=1 PM_SEED_GROUP=1024
done
