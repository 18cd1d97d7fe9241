#!/bin/bash
# The bench lines kept under profiles/ (with cpu_baseline), one per option set, plus the
# kernel-trace of the exact_halves -k 1 run.  Usage (GPU box): bash scripts/bench_lines.sh r01
set -o pipefail
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lines_$tag
mkdir -p $out
python bench.py --steps 5 --warmup 1 > $out/bench_K2.json 2> $out/bench_K2.err && echo "K2 done"
python bench.py --steps 5 --warmup 1 --k 0 > $out/bench_k0.json 2> $out/bench_k0.err && echo "k0 done"
python bench.py --steps 5 --warmup 1 --k 1 > $out/bench_K1.json 2> $out/bench_K1.err && echo "K1 done"
python bench.py --steps 5 --warmup 1 --k 1 --indels 1 > $out/bench_k1_edits.json 2> $out/bench_k1_edits.err && echo "k1 edits done"
python bench.py --steps 3 --warmup 1 --k 2 --indels 1 > $out/bench_k2_edits.json 2> $out/bench_k2_edits.err && echo "k2 edits done"
python bench.py --steps 2 --warmup 1 --primers 1000000 --no-cpu > $out/bench_K2_1M.json 2> $out/bench_K2_1M.err && echo "1M done"
# the bit-parallel family at config size (200k patterns): integer-ALU bound, alu_roofline in the line
python bench.py --steps 2 --warmup 1 --kernel bitpar --db-bases 16000000 --k 0 --no-cpu --no-check > $out/bench_bitpar_k0_200k.json 2> $out/bench_bitpar_k0_200k.err && echo "bitpar k0 done"
python bench.py --steps 2 --warmup 1 --kernel bitpar --db-bases 16000000 --k 2 --no-cpu --no-check > $out/bench_bitpar_K2_200k.json 2> $out/bench_bitpar_K2_200k.err && echo "bitpar K2 done"
python bench.py --steps 2 --warmup 1 --kernel bitpar --db-bases 16000000 --k 2 --indels 1 --no-cpu --no-check > $out/bench_bitpar_k2_200k.json 2> $out/bench_bitpar_k2_200k.err && echo "bitpar k2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_k1e -- python3 bench.py --steps 3 --warmup 1 --k 1 --indels 1 --no-cpu --scan-passes 0 > $out/kt_k1e.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_k2e -- python3 bench.py --steps 3 --warmup 1 --k 2 --indels 1 --no-cpu --scan-passes 0 > $out/kt_k2e.log 2>&1
echo "all done"
