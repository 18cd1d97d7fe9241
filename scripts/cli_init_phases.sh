#!/bin/bash
# PM_DEBUG phase lines of pm_init for a 3 Gbp database through pm_primer_match (GPU box).
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, "scripts")
import cli_scale as C
rng = np.random.default_rng(1)
codes = rng.integers(0, 4, size=3_000_000_000, dtype=np.uint8)
C.write_db("/dev/shm/pmdb", codes, 24)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
with open("/dev/shm/pmprimers.txt", "wb") as f:
    for i in range(100000):
        f.write(lut[rng.integers(0, 4, size=20)].tobytes() + b"\n")
PY
for i in 1; do PM_DEBUG=1 sequence-alignment-tools_amd/host/pm_primer_match -i /dev/shm/pmdb -P /dev/shm/pmprimers.txt -K 2 -r -c -v 2>&1 >/dev/null | grep -v "^$" ; done
rm -f /dev/shm/pmdb.* /dev/shm/pmprimers.txt
