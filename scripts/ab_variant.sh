#!/bin/bash
# A/B of a library variant (csrc/libpm_gpu<variant>.so, built with `make VARIANT=... EXTRA=...`) against the product build:
# bench lines for -K 2 and -k 2 and one --pmc pass of the L2 counters each.  Usage (GPU box): bash scripts/ab_variant.sh _nt
v=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$PWD/sequence-alignment-tools_amd/csrc/libpm_gpu$v.so
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'value %.1f step %.2f kernel %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for opt in "K2 --k 2" "k2e --k 2 --indels 1"; do
  set -- $opt; name=$1; shift
  for rep in 1 2; do
    python3 bench.py --no-cpu --steps 5 --warmup 2 --scan-passes 0 "$@" 2>/dev/null | line base_$name
    PM_GPU_LIB=$L PM_GPU_LIB_AB=1 python3 bench.py --no-cpu --steps 5 --warmup 2 --scan-passes 0 "$@" 2>/dev/null | line variant_$name
  done
  for which in base variant; do
    if [ $which = variant ]; then export PM_GPU_LIB=$L PM_GPU_LIB_AB=1; else unset PM_GPU_LIB PM_GPU_LIB_AB; fi
    timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/ab${v}_${which}_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-check --scan-passes 0 "$@" > /dev/null 2>&1
    python3 - "$which" "$name" "$v" <<'PY'
import csv, glob, sys
which, name, v = sys.argv[1:4]
tot = {}
for f in glob.glob("gpurun_out/ab%s_%s_%s/*/*counter_collection.csv" % (v, which, name)):
    for r in csv.DictReader(open(f)):
        if "pm_pair_scan" in r["Kernel_Name"] or "pm_pair_edit_scan" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(which, name, {k: "%.4g" % x for k, x in tot.items()})
PY
  done
  unset PM_GPU_LIB PM_GPU_LIB_AB
done
