cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/kt_scan
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_scan/K2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu --scan-passes 3 > gpurun_out/kt_scan/K2.log 2>&1
echo rc=$?
f=$(ls gpurun_out/kt_scan/K2/*/*kernel_stats.csv | tail -1)
cut -d, -f1-4 $f | cut -c1-160 | head -30
