#!/bin/bash
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "value %.1f step %.3f kernel %.3f exch %s %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("exchange_ms"), d.get("exchange_device_ms")), d["config"]["final_hits"])'
timeout -k 10 200 python scripts/step_phases_owned.py 375000000 2>&1 | grep -a "shard" | tail -1
for db in 3000000000 375000000; do
PM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu --db-bases $db 2> gpurun_out/dist1.err | tail -1 | python -c "$P" "dist1 $db"
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu --db-bases $db 2>/dev/null | python -c "$P" "plain $db"
done
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -4
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_bench_ranks.py tests/test_gpu_rccl.py tests/test_gpu_ranks_cli.py 2>&1 | tail -3
