#!/bin/bash
# bench.py on the hard streams (--stream-style skew / vocab / tandem, primers cut from the stream), -K 2 and -k 2, with the
# pair kernel's block / round / key-hit counters (--pair-stats is a measurement build: the line WITHOUT it is the timing).
# Usage (GPU box): bash scripts/bench_hard.sh r04 3000000000 100000 [K2|k2e|both]
tag=$1; db=${2:-3000000000}; primers=${3:-100000}; which=${4:-both}
mkdir -p gpurun_out/hard_$tag
for style in uniform skew vocab tandem; do
  for opt in "K2 --k 2" "k2e --k 2 --indels 1"; do
    set -- $opt; name=$1; shift
    [ "$which" != "both" ] && [ "$which" != "$name" ] && continue
    for stats in "" "--pair-stats"; do
      [ -n "$stats" ] && [ "$name" != "K2" ] && continue
      out=gpurun_out/hard_$tag/bench_hard_${style}_${name}${stats:+_stats}
      timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu --scan-passes 0 --db-bases $db --primers $primers --landing $((1<<27)) \
        --stream-style $style "$@" $stats > $out.json 2> $out.err
      echo "$style $name $stats rc=$? $(python -c "
import json,sys
try:
    d=json.load(open('$out.json')); c=d['config']
    print('ms %.2f kernel %.2f value %.1f cands %d finals %d stats %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], c['candidates'], c['final_hits'], c['scan_stats']))
except Exception as e:
    print('no line', e)
")"
    done
  done
done
