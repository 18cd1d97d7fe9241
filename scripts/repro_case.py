import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import adversarial as A, sat_amd
seed = int(sys.argv[1])
c = A.small_case(seed)
want = A.oracle_hits(c)
print(A.describe(c), "oracle", len(want))
for bound in (300, 4800, 1 << 30):
    os.environ["PM_DENSE_BOUND"] = str(bound)
    os.environ["PM_DEBUG"] = "1"
    st = {}
    try:
        got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, mode=0, stats=st)
    except sat_amd.PmError as e:
        print("bound", bound, "error", e); continue
    print("bound", bound, "hits", len(got), "cuts", st.get("range_splits"), "missing", sorted(set(want) - set(got))[:5], "extra", sorted(set(got) - set(want))[:5])
os.environ.pop("PM_DENSE_BOUND")
got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, mode=0)
print("find_all", len(got), sorted(set(want) - set(got))[:5])
got = A.gpu_hits(c, kernel=sat_amd.KERNEL_AUTO, mode=1)
print("chunks", c["chunk"], len(got), sorted(set(want) - set(got))[:5])
