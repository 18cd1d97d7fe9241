#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 0 16 32; do
PM_SEED_DEBUG=$dbg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5_kt$dbg -- python bench.py --steps 3 --warmup 1 --no-cpu --no-check > gpurun_out/c5_kt.log 2>&1
python - $dbg <<'PY'
import csv,glob,sys
for f in glob.glob("gpurun_out/c5_kt%s/*/*kernel_stats.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if "pm_pair" in r["Name"]: print("debug", sys.argv[1], r["Name"][30:60], r["Calls"], r["AverageNs"])
PY
done
