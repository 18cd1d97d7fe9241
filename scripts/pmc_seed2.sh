#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for e in "$@"; do export "$e"; done
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$n -- python bench.py --db-bases 1000000000 --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_${tag}_$n.log 2>&1; }
run 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run 3 TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run 4 TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "seed" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print("${tag} %-32s %.4g"%(k,v))
PY
