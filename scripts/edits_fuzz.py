"""Extended fuzz of the edit-distance seed plan against the oracle (GPU box): python scripts/edits_fuzz.py [nseeds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for seed in range(4, 4 + n):
    try:
        T.test_edit_distance_seed_plan(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAIL", str(e)[:300])
print("seeds", n, "failures", bad)
