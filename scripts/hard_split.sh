#!/bin/bash
# per-kernel split (rocprofv3 --kernel-trace --stats) of bench.py on the hard streams.  Usage (GPU box): bash scripts/hard_split.sh r04 300000000
tag=$1; db=${2:-300000000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hard_$tag
for style in skew vocab tandem; do
  out=$R/gpurun_out/hard_$tag/kt_$style
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-check --scan-passes 0 --db-bases $db --landing $((1<<27)) --stream-style $style > $out.json 2> $out.err
  echo "== $style"; python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("%-70s calls %5s avg_us %10.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
