#!/usr/bin/env python3
"""Markdown tables for DESIGN.md section 7 / README from the committed profiles of a round.  python scripts/doc_tables.py r04"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def J(name):
    f = os.path.join(P, name)
    return json.load(open(f)) if os.path.exists(f) else None


def kstats(name):
    f = os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, name))
    out = []
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            if "pm::" in r["Name"] and "pack_stream" not in r["Name"] and "stream_read" not in r["Name"]:
                nm = r["Name"].replace("pm::(anonymous namespace)::", "").replace("void ", "")
                nm = nm.split("(")[0]
                out.append((nm, int(r["Calls"]), float(r["AverageNs"]) / 1e6))
    return out


rows = [("`-K 2`", "K2", "K2"), ("`-K 1`", "K1", "K1"), ("k = 0", "k0", "K0"), ("`-k 1`", "k1_edits", "k1_edits"), ("`-k 2`", "k2_edits", "k2_edits"), ("`-K 2`, 1M primers", "K2_1M", "K2_1M")]
tr = {(e["k"], e["indels"]): e for e in (J("traffic_%s.json" % tag) or {"entries": []})["entries"]}
print("| run | scan kernels, rocprofv3 avg per launch | step | `value`, Gbases/s | through `pm_scan` (1 GiB ranges) | `roofline.frac` of 8 TB/s | fabric traffic per launch | reference CPU, 1 thread / all cores, Gbases/s |")
print("|---|---|---|---|---|---|---|---|")
for label, b, k in rows:
    d = J("%s_bench_%s.json" % (tag, b))
    if not d:
        continue
    ks = [x for x in kstats(k) if x[2] > 0.2 and not x[0].startswith("pm_cluster") and "dedup" not in x[0]][:4]
    kk = " + ".join("`%s` %.2f ms%s" % (n, ms, "" if c <= 6 else " × %d" % (c // 4)) for n, c, ms in ks)
    cb = d.get("cpu_baseline") or {}
    ac = (cb.get("all_cores") or {}).get("value")
    key = {"K2": (2, 0), "K1": (1, 0), "k0": (0, 0), "k1_edits": (1, 1), "k2_edits": (2, 1)}.get(b)
    t = tr.get(key)
    print("| %s | %s | %.1f ms | **%.1f** | %.1f ms = %.0f | %.4f | %s | %s |" % (
        label, kk, d["ms_per_step"], d["value"], d["pm_scan_ms"], d["value_through_pm_scan"], d["roofline"]["frac"],
        "%.1f GB = %.1f× algorithmic" % (t["traffic_bytes"] / 1e9, t["traffic_bytes"] / d["roofline"]["algorithmic_bytes"]) if t else "--",
        "%.1e / %.1e" % (cb["value"], ac) if cb.get("value") and ac else "--"))
print()
print("| `pm_scan` range | 64 MiB | 256 MiB | 1 GiB (the plugin's) |")
print("|---|---|---|---|")
vals = [J("%s_bench_K2_scanchunk%d.json" % (tag, c)) for c in (26, 28, 30)]
if all(vals):
    print("| `-K 2`, 3 Gbp, ms per whole-stream pass | " + " | ".join("%.1f" % v["pm_scan_ms"] for v in vals) + " |")
print()
c = J("%s_cli_scale_3g.json" % tag)
if c:
    print("| command line (3 Gbp `.sqn`, 100k primers / pairs) | wall | phases |")
    print("|---|---|---|")
    for k2, v in c["runs"].items():
        print("| %s | %.2f s | %s |" % (k2, v["wall_s"], "; ".join(p.strip() for p in v.get("phases", [])[2:4])))
print()
print("| stream (300 Mbp unless noted, 100k primers cut from it) | `-K 2` step | candidates -> final hits | suspects | rounds per block / key-hit rate | `-k 2` step |")
print("|---|---|---|---|---|---|")
for st in ("uniform", "skew", "vocab", "tandem"):
    a = J("%s_bench_hard_%s_K2_300m.json" % (tag, st)); s = J("%s_bench_hard_%s_K2_300m_pairstats.json" % (tag, st)); e = J("%s_bench_hard_%s_k2e_300m.json" % (tag, st))
    if not a:
        continue
    ss = (s or a)["config"]["scan_stats"]
    print("| %s | %.1f ms | %.3g -> %.3g | %.3g | %s | %s |" % (st, a["ms_per_step"], a["config"]["candidates"], a["config"]["final_hits"], a["config"]["scan_stats"]["between_stages"],
          "%.1f / %.2f" % (ss["rounds_per_block"], ss["key_hit_rate"]) if "rounds_per_block" in ss else "--", "%.0f ms (%.3g hits)" % (e["ms_per_step"], e["config"]["final_hits"]) if e else "out of memory / time"))
for st in ("uniform", "skew", "tandem"):
    a = J("%s_bench_hard_%s_K2_3g.json" % (tag, st))
    if a:
        print("| %s, 3 Gbp | %.1f ms | %.3g -> %.3g | %.3g | -- | -- |" % (st, a["ms_per_step"], a["config"]["candidates"], a["config"]["final_hits"], a["config"]["scan_stats"]["between_stages"]))
