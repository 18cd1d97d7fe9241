#!/bin/bash
# SQ instruction-mix / wait counters of the scan kernel for any bench option set (1 Gbp).
# Usage (GPU box): bash scripts/pmc_sq.sh <tag> [bench.py options...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
A="--db-bases 1000000000 --steps 1 --warmup 0 --no-cpu $*"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmcs_${tag}_1 -- python bench.py $A > gpurun_out/pmcs_${tag}_1.log 2>&1
echo "pass 1 rc=$?" >> gpurun_out/pmcs_${tag}_progress.txt
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmcs_${tag}_2 -- python bench.py $A > gpurun_out/pmcs_${tag}_2.log 2>&1
echo "pass 2 rc=$?" >> gpurun_out/pmcs_${tag}_progress.txt
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmcs_${tag}_3 -- python bench.py $A > gpurun_out/pmcs_${tag}_3.log 2>&1
echo "pass 3 rc=$?" >> gpurun_out/pmcs_${tag}_progress.txt
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmcs_${tag}_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if any(x in r["Kernel_Name"] for x in ("seed_scan", "pair_scan", "edit_scan", "half_scan")): agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print("${tag} %-24s %.4g"%(k,v))
PY
