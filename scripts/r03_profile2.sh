#!/bin/bash
# round-3 profiles, stage 2: the bench lines kept under profiles/ (with cpu_baseline, issue_roofline from profiles/issue_r03.json)
cd $GRAFT_REPO_ROOT
out=gpurun_out/lines_r03
mkdir -p $out
python bench.py --steps 5 --warmup 1 > $out/bench_K2.json 2> $out/bench_K2.err && echo "K2 done"
python bench.py --steps 5 --warmup 1 --k 0 > $out/bench_k0.json 2> $out/bench_k0.err && echo "k0 done"
python bench.py --steps 5 --warmup 1 --k 1 > $out/bench_K1.json 2> $out/bench_K1.err && echo "K1 done"
python bench.py --steps 5 --warmup 1 --k 1 --indels 1 > $out/bench_k1_edits.json 2> $out/bench_k1_edits.err && echo "k1 edits done"
python bench.py --steps 3 --warmup 1 --k 2 --indels 1 > $out/bench_k2_edits.json 2> $out/bench_k2_edits.err && echo "k2 edits done"
python bench.py --steps 2 --warmup 1 --primers 1000000 --no-cpu > $out/bench_K2_1M.json 2> $out/bench_K2_1M.err && echo "1M done"
PM_PAIR_ROW=12 python bench.py --steps 2 --warmup 1 --primers 1000000 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1M row 12', d['ms_per_step'], d['roofline']['kernel_ms'])"
python -c "import json; d=json.load(open('$out/bench_K2_1M.json')); print('1M auto', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['kernel'][:90])"
for f in K2 k0 K1 k1_edits k2_edits; do python -c "import json; d=json.load(open('$out/bench_$f.json')); print('$f', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'], d.get('issue_roofline',{}).get('binding'), d['cpu_baseline']['value'])"; done
