#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_exhaustive.py > gpurun_out/call3_tests.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/call3_tests.txt
for cfg in "0 12" "0 14"; do set -- $cfg
PM_SEED_DEBUG=$1 PM_PAIR_ROW=$2 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug $1 row $2', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['candidates'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PM_PAIR_ROW=12 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/call3_kt -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/call3_kt.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/call3_kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "pm" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
