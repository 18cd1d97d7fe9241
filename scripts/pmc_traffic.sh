#!/bin/bash
# Fabric traffic (FETCH_SIZE, WRITE_SIZE; separate --pmc passes as MI355X_MICROARCH.md prescribes) of the scan kernel of one
# bench option set, 3 Gbp x 100k primers, one launch.  Writes gpurun_out/<tag>_traffic_<name>.json; scripts/collect_profiles.py
# merges it into profiles/traffic_<tag>.json.  Usage (GPU box): bash scripts/pmc_traffic.sh r02 k2_edits pm_edit_scan --k 2 --indels 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; name=$2; kernel=$3; shift 3
A="--steps 1 --warmup 0 --no-cpu --no-check --scan-passes 0 $*"
# the same command without a profiler first: what the kernel takes on this box with this code (kernel_ms_at_profile)
python bench.py --steps 3 --warmup 1 --no-cpu --no-check --scan-passes 0 $* > gpurun_out/pmc_${tag}_${name}_plain.json 2> gpurun_out/pmc_${tag}_${name}_plain.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmct_${tag}_${name}_f -- python bench.py $A > gpurun_out/pmct_${tag}_${name}_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmct_${tag}_${name}_w -- python bench.py $A > gpurun_out/pmct_${tag}_${name}_w.log 2>&1
python - "$tag" "$name" "$kernel" "$@" <<'PY'
import csv, glob, json, sys
sys.path.insert(0, ".")
import bench
tag, name, kernel = sys.argv[1:4]
opts = sys.argv[4:]
def opt(flag, default):
    return int(opts[opts.index(flag) + 1]) if flag in opts else default
def sopt(flag, default):
    return opts[opts.index(flag) + 1] if flag in opts else default
try:
    plain_ms = json.load(open("gpurun_out/pmc_%s_%s_plain.json" % (tag, name)))["roofline"]["kernel_ms"]
except Exception:
    plain_ms = None
tie = {"stream_style": sopt("--stream-style", "uniform"), "code_sha": bench.code_sha(), "kernel_ms_at_profile": plain_ms}
tot = {}
for what, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    s = 0.0
    for f in glob.glob("gpurun_out/pmct_%s_%s_%s/*/*counter_collection.csv" % (tag, name, what)):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                s += float(r["Counter_Value"])
    tot[ctr] = s
e = {"k": opt("--k", 2), "indels": opt("--indels", 0), "db_bases": opt("--db-bases", 3000000000), "primers": opt("--primers", 100000), "kernel": kernel,
     "FETCH_SIZE_KiB": tot["FETCH_SIZE"], "WRITE_SIZE_KiB": tot["WRITE_SIZE"], "traffic_bytes": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024}
e.update(tie)
json.dump(e, open("gpurun_out/%s_traffic_%s.json" % (tag, name), "w"), indent=1)
print(e)
PY
