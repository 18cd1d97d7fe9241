#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step %.2f kernel %.2f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]), d["config"]["candidates"], d["config"]["planted_found"], d["config"]["kernel"][:70])'
for c in 524288 1048576 2097152; do
PM_SEED_CHUNK=$c timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu --k 1 2>/dev/null | python -c "$P" "K1 chunk $c"
done
for cfg in "262144 12" "349526 16" "349526 12" "524288 24" "524288 16" "1048576 33"; do set -- $cfg
PM_SEED_TILE=$1 PM_PAIR_ROW=$2 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu --primers 1000000 2>/dev/null | python -c "$P" "1M tile $1 row $2"
done
