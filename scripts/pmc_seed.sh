#!/bin/bash
# PMC passes for the seed kernel (1 Gbp, -K 2). Usage: bash scripts/pmc_seed.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for e in "$@"; do export "$e"; done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_1 -- python bench.py --db-bases 1000000000 --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_${tag}_1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_${tag}_2 -- python bench.py --db-bases 1000000000 --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_${tag}_2.log 2>&1
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "seed" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print("${tag} %-24s %.4g"%(k,v))
PY
