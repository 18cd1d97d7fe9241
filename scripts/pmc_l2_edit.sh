#!/bin/bash
# L2 hit / miss and L1 -> L2 request counters of the edit-distance scan kernel (3 Gbp x 100k primers, -k 2).
# Usage (GPU box): bash scripts/pmc_l2_edit.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-edit}
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pmc_l2_$tag -- python bench.py --steps 1 --warmup 0 --k 2 --indels 1 --no-cpu --no-check > gpurun_out/pmc_l2_$tag.log 2>&1
python - <<PY
import csv,glob,collections
for f in glob.glob("gpurun_out/pmc_l2_$tag/*/*counter_collection.csv"):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "pm_edit_scan" in r["Kernel_Name"] or "seed_scan" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    print("$tag", dict(agg))
PY
