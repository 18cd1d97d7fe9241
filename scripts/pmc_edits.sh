#!/bin/bash
# PMC passes for the edit-distance plan's kernels (-k 2, 1 Gbp). Usage: bash scripts/pmc_edits.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for e in "$@"; do export "$e"; done
A="--db-bases 1000000000 --steps 1 --warmup 0 --no-cpu --k 2 --indels 1"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmce_${tag}_1 -- python bench.py $A > gpurun_out/pmce_${tag}_1.log 2>&1
echo "pass done rc=$?" >> gpurun_out/pmce_${tag}_progress.txt
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmce_${tag}_2 -- python bench.py $A > gpurun_out/pmce_${tag}_2.log 2>&1
echo "pass done rc=$?" >> gpurun_out/pmce_${tag}_progress.txt
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmce_${tag}_3 -- python bench.py $A > gpurun_out/pmce_${tag}_3.log 2>&1
echo "pass done rc=$?" >> gpurun_out/pmce_${tag}_progress.txt
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_FLAT --output-format csv -d gpurun_out/pmce_${tag}_4 -- python bench.py $A > gpurun_out/pmce_${tag}_4.log 2>&1
echo "pass done rc=$?" >> gpurun_out/pmce_${tag}_progress.txt
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmce_${tag}_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        for kn in ("seed_scan","edits_verify"):
            if kn in r["Kernel_Name"]: agg[(kn,r["Counter_Name"])]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print("${tag} %-14s %-24s %.4g"%(k[0],k[1],v))
PY
