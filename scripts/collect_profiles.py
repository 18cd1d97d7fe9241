#!/usr/bin/env python3
"""Copy the summaries of scripts/profile_round.sh and scripts/bench_lines.sh from gpurun_out/ into
profiles/ (tracked) and rebuild profiles/traffic_<tag>.json from the PMC passes.
Usage (build container, after the two gpurun calls): python scripts/collect_profiles.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def head_of(code_sha):
    """the commit the counters belong to: HEAD when the working tree's kernel sources hash to what the GPU box hashed
    (bench.code_sha, recorded by scripts/pmc_*.sh), else unknown -- a counter file must never claim a commit it was not made on"""
    try:
        import bench
        if code_sha and bench.code_sha() == code_sha:
            h = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
            dirty = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "--", "sequence-alignment-tools_amd/csrc"]).returncode != 0
            return h + ("+uncommitted" if dirty else "")
    except Exception:
        pass
    return None


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)          # gpurun merges into gpurun_out/: older runs stay around
    return f[-1] if f else None


def product_rows(path):
    """rows of our kernels only (torch's generators and copies are not the product)"""
    with open(path) as f:
        r = list(csv.reader(f))
    keep = [r[0]] + [x for x in r[1:] if "pm::" in ",".join(x) or "pm_stream" in ",".join(x)]
    return keep


for name in ("K2", "K0", "K1", "k1_edits", "k2_edits", "K2_1M"):
    src = one(os.path.join(G, "prof_" + tag, "kt_" + name, "*", "*kernel_stats.csv"))
    if src:
        shutil.copy(src, os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, name)))
for name in ("tcp_gather", "pipe_overlap", "valu_rate", "pack_rate"):               # hardware probes (scripts/probe/*.hip)
    src = os.path.join(G, "prof_" + tag, "probe_%s.txt" % name)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(P, "%s_probe_%s.txt" % (tag, name)))
for name, dst in (("kt_k1e", "k1_edits"), ("kt_k2e", "k2_edits")):
    src = one(os.path.join(G, "lines_" + tag, name, "*", "*kernel_stats.csv"))
    if src:
        shutil.copy(src, os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, dst)))
for name in ("bench_K2", "bench_k0", "bench_K1", "bench_k1_edits", "bench_k2_edits", "bench_K2_1M", "bench_bitpar_k0_200k", "bench_bitpar_K2_200k", "bench_bitpar_k2_200k",
             "bench_K2_scanchunk26", "bench_K2_scanchunk28", "bench_K2_scanchunk30"):
    src = os.path.join(G, "lines_" + tag, name + ".json")
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(P, "%s_%s.json" % (tag, name)))

for name in ("edit_pair_floor", "cli_scale_3g"):
    src = os.path.join(G, "%s_%s.json" % (tag, name))
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(P, "%s_%s.json" % (tag, name)))
for name in ("stages_k2_edits", "pmc_l2_k2_edits", "sq_k2_edits", "stages_k1_edits", "sq_k1_edits"):
    src = os.path.join(G, "%s_%s.txt" % (tag, name))
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(P, "%s_%s.txt" % (tag, name)))

counters = {}
for name in ("pmc_fetch_K2", "pmc_write_K2", "pmc_fetch_K0", "pmc_l2_K2"):
    src = one(os.path.join(G, "prof_" + tag, name, "*", "*counter_collection.csv"))
    if not src:
        continue
    rows = product_rows(src)
    with open(os.path.join(P, "%s_%s.csv" % (tag, name)), "w", newline="") as f:
        csv.writer(f).writerows(rows)
    hdr = rows[0]
    kn, cn, cv = hdr.index("Kernel_Name"), hdr.index("Counter_Name"), hdr.index("Counter_Value")
    for x in rows[1:]:
        if "pm_seed_scan" in x[kn] or "pm_pair_scan" in x[kn] or "pm_pair_edit_scan" in x[kn]:
            counters[(name, x[cn])] = counters.get((name, x[cn]), 0.0) + float(x[cv])

entries = []
for k, fetch, write, kernel in ((2, "pmc_fetch_K2", "pmc_write_K2", "pm_pair_scan"), (0, "pmc_fetch_K0", None, "pm_seed_scan<20,2,false>")):
    fs = counters.get((fetch, "FETCH_SIZE"))
    if fs is None:
        continue
    ws = counters.get((write, "WRITE_SIZE"), 0.0) if write else 0.0
    entries.append({"k": k, "indels": 0, "db_bases": 3000000000, "primers": 100000, "kernel": kernel,
                    "FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": ws, "traffic_bytes": (2 * fs + ws) * 1024})
for f in sorted(glob.glob(os.path.join(G, "%s_traffic_*.json" % tag))):     # scripts/pmc_traffic.sh: one option set each
    with open(f) as fh:
        e = json.load(fh)
    e["head"] = head_of(e.get("code_sha"))
    if e.get("FETCH_SIZE_KiB"):
        entries = [x for x in entries if (x["k"], x["indels"], x["db_bases"], x["primers"]) != (e["k"], e["indels"], e["db_bases"], e["primers"])] + [e]
if entries:
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots); unit KiB; "
            "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the guide's gfx950 correction (FETCH_SIZE tallies 128-B requests at 64 B "
            "for wide coalesced streams). Fabric-side count: Infinity Cache hits are included.")
    with open(os.path.join(P, "traffic_%s.json" % tag), "w") as f:
        json.dump({"note": note, "entries": entries}, f, indent=1)
issue = []
for f in sorted(glob.glob(os.path.join(G, "%s_issue_*.json" % tag))):       # scripts/pmc_issue.sh: one option set each
    with open(f) as fh:
        e = json.load(fh)
    e["head"] = head_of(e.get("code_sha"))
    if e.get("SQ_INSTS_VALU") or e.get("TCP_TCC_READ_REQ_sum"):
        issue.append(e)
if issue:
    note = ("rocprofv3 --pmc, one launch of the scan kernel at 3 Gbp x 100k primers (scripts/pmc_issue.sh): SQ counters are sums over the chip "
            "(SQ_INSTS_* = wave-level instructions, SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT = LDS-array cycles summed over the CUs, "
            "SQ_WAVE_CYCLES / SQ_WAIT_* = quad-cycles summed over the waves); TCP_TCC_READ_REQ_sum = L1 -> L2 read requests, one 128-byte line each.")
    with open(os.path.join(P, "issue_%s.json" % tag), "w") as f:
        json.dump({"note": note, "entries": issue}, f, indent=1)
print({"%s/%s" % k: v for k, v in counters.items()})
