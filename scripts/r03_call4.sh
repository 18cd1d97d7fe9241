#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--steps 1 --warmup 0 --no-cpu --no-check --k 2"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/c4_sq -- python bench.py $A > gpurun_out/c4_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/c4_l2 -- python bench.py $A > gpurun_out/c4_l2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/c4_m -- python bench.py $A > gpurun_out/c4_m.log 2>&1
python - <<'PY'
import csv,glob,collections
for kern in ("pm_pair_scan","pm_pair_verify"):
    agg=collections.defaultdict(float)
    for f in glob.glob("gpurun_out/c4_*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(kern, {k:"%.3g"%v for k,v in sorted(agg.items())})
PY
