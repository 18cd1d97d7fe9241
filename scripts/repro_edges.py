import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import adversarial as A, sat_amd
for seed in map(int, sys.argv[1:]):
    c = A.small_case(seed)
    want = A.oracle_hits(c)
    n = c["n"]
    cuts = [1, 3, 7, 19, 33, 60, n // 2, n - 61, n - 30, n - 5, n - 3, n - 2, n - 1]
    try:
        got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, cuts=cuts)
    except Exception as e:
        print(seed, A.describe(c), "ERROR", str(e)[:200]); continue
    sg, sw = set(got), set(want)
    print(seed, A.describe(c), c["kernel_desc"][:60], "n", n, "hits", len(got), "oracle", len(want), "only GPU", sorted(sg - sw)[:8], "only oracle", sorted(sw - sg)[:8], "dups", len(got) - len(sg))
