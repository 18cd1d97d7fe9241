#!/bin/bash
# round-3 GPU call 2: the new pair kernel -- parity, then bench with stage switches and row strides
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_exhaustive.py --durations=8 > gpurun_out/call2_tests.txt 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/call2_tests.txt
for row in 12 10 16 24; do
  PM_PAIR_ROW=$row timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu 2> gpurun_out/call2_bench_row$row.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('row $row', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['candidates'], d['config']['final_hits'], d['config']['planted_found'])" | tee -a gpurun_out/call2_bench.txt
done
PM_SEED_DEBUG=1 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug1 (no emit)', d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a gpurun_out/call2_bench.txt
timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu --k 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K1', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['planted_found'])" | tee -a gpurun_out/call2_bench.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/call2_kt -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/call2_kt.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/call2_kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "pm" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
