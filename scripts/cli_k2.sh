#!/bin/bash
# pm_primer_match -k 2 -r -c on a 3 Gbp synthetic database, 100k primers (GPU box): bash scripts/cli_k2.sh
set -e
python - <<'PY'
import sys, os; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/scripts")
import numpy as np, cli_scale as cs
rng = np.random.default_rng(20260101)
codes = rng.integers(0, 4, size=3_000_000_000, dtype=np.uint8)
cs.write_db("/tmp/k2db", codes, 24)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
head = lut[codes[:1 << 26]].tobytes()
prim = []
for i in range(100000):
    if i % 10 == 0:
        a = int(rng.integers(0, len(head) - 20)); prim.append(cs.mutate(rng, head[a:a + 20], i // 10 % 3))
    else:
        prim.append(lut[rng.integers(0, 4, size=20)].tobytes())
open("/tmp/k2prim.txt", "wb").write(b"\n".join(prim) + b"\n")
PY
H=$GRAFT_REPO_ROOT/sequence-alignment-tools_amd/host
$H/pm_primer_match -i /tmp/k2db -P /tmp/k2prim.txt -k 2 -r -c -v > /tmp/k2.out 2> /tmp/k2.err || true
tail -n 8 /tmp/k2.err; wc -l /tmp/k2.out
