#!/bin/bash
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "value %.1f step %.3f kernel %.3f exch %s %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("exchange_ms"), d.get("exchange_device_ms")), d["config"]["final_hits"])'
for db in 3000000000 375000000; do
PM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu --db-bases $db > gpurun_out/dist1.json 2> gpurun_out/dist1.err; tail -3 gpurun_out/dist1.err; tail -1 gpurun_out/dist1.json | python -c "$P" "dist1 $db"
done
