#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the default bench command, then FETCH_SIZE and
# WRITE_SIZE in their own PMC passes (MI355X_MICROARCH.md: TCC slots, no trace domains mixed in).
# Usage (on the GPU box, via gpurun): bash scripts/profile_round.sh r01
set -o pipefail
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out
for k in 2 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_K$k -- python bench.py --steps 3 --warmup 1 --k $k --no-cpu > $out/kt_K$k.log 2>&1
  echo "kernel-trace K$k done"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_K2 -- python bench.py --steps 1 --warmup 0 --k 2 --no-cpu > $out/pmc_fetch_K2.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_K2 -- python bench.py --steps 1 --warmup 0 --k 2 --no-cpu > $out/pmc_write_K2.log 2>&1
echo "write done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_K0 -- python bench.py --steps 1 --warmup 0 --k 0 --no-cpu > $out/pmc_fetch_K0.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2_K2 -- python bench.py --steps 1 --warmup 0 --k 2 --no-cpu > $out/pmc_l2_K2.log 2>&1
echo "all done"
