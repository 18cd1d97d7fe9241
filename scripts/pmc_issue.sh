#!/bin/bash
# Which on-chip pipe is how busy (SURVEY 8(d) "honest secondary bound"): VALU wave-instructions, LDS-array
# cycles and L1 -> L2 requests of the scan kernel of one bench option set, 3 Gbp x 100k primers, one
# launch; two PMC passes (8 SQ slots / 4 TCP-TCC slots), nothing but --pmc on the command line.
# Writes gpurun_out/<tag>_issue_<name>.json; scripts/collect_profiles.py merges these into
# profiles/issue_<tag>.json, which bench.py's issue_roofline reads.
# Usage (GPU box): bash scripts/pmc_issue.sh r03 K2 pm_pair_scan --k 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; name=$2; kernel=$3; shift 3
A="--steps 1 --warmup 0 --no-cpu --no-check --scan-passes 0 $*"
# the same command without a profiler first: what the kernel takes on this box with this code (kernel_ms_at_profile)
python bench.py --steps 3 --warmup 1 --no-cpu --no-check --scan-passes 0 $* > gpurun_out/pmc_${tag}_${name}_plain.json 2> gpurun_out/pmc_${tag}_${name}_plain.log
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmci_${tag}_${name}_sq -- python bench.py $A > gpurun_out/pmci_${tag}_${name}_sq.log 2>&1
echo "sq pass rc=$?"
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pmci_${tag}_${name}_l2 -- python bench.py $A > gpurun_out/pmci_${tag}_${name}_l2.log 2>&1
echo "l2 pass rc=$?"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmci_${tag}_${name}_m -- python bench.py $A > gpurun_out/pmci_${tag}_${name}_m.log 2>&1
echo "misc pass rc=$?"
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
e = {"k": opt("--k", 2), "indels": opt("--indels", 0), "db_bases": opt("--db-bases", 3000000000), "primers": opt("--primers", 100000), "kernel": kernel}
for what in ("sq", "l2", "m"):
    for f in glob.glob("gpurun_out/pmci_%s_%s_%s/*/*counter_collection.csv" % (tag, name, what)):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
e.update(tie)
json.dump(e, open("gpurun_out/%s_issue_%s.json" % (tag, name), "w"), indent=1)
print(e)
PY
