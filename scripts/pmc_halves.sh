#!/bin/bash
# PMC passes for the exact_halves -k 1 seed kernel (1 Gbp). Usage: bash scripts/pmc_halves.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for e in "$@"; do export "$e"; done
A="--db-bases 1000000000 --steps 1 --warmup 0 --no-cpu --k 1 --indels 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmch_${tag}_1 -- python bench.py $A > gpurun_out/pmch_${tag}_1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmch_${tag}_2 -- python bench.py $A > gpurun_out/pmch_${tag}_2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmch_${tag}_3 -- python bench.py $A > gpurun_out/pmch_${tag}_3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmch_${tag}_4 -- python bench.py $A > gpurun_out/pmch_${tag}_4.log 2>&1
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmch_${tag}_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "seed_scan" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print("${tag} %-24s %.4g"%(k,v))
PY
