"""Run oracle/_ref/ref_harness (the real reference engines) on in-memory inputs.  Test-only."""
import os
import subprocess
import tempfile


def run_ref(harness, raw_or_codes, patterns, table=None, sel=0, k=0, indels=True, rc=False, minka=1000):
    """Returns the reference engine's hits as a sorted list of (end, id, value)."""
    with tempfile.TemporaryDirectory() as d:
        db = os.path.join(d, "db")
        if table is None:
            with open(db, "wb") as f:
                f.write(bytes(raw_or_codes))
        else:
            with open(db + ".sqn", "wb") as f:
                f.write(bytes(raw_or_codes))
            with open(db + ".tbl", "wb") as f:
                f.write(bytes(table))
        pf = os.path.join(d, "pat.txt")
        with open(pf, "w") as f:
            f.write("\n".join(patterns) + "\n")
        cmd = [harness, "-N", str(sel), "-m", str(minka), "-i", db, "-P", pf]
        if k:
            cmd += ["-k" if indels else "-K", str(k)]
        if rc:
            cmd += ["-r"]
        if table is not None:
            cmd += ["-n"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            raise RuntimeError("ref_harness failed (%d): %s" % (out.returncode, out.stderr[-2000:]))
        hits = []
        for line in out.stdout.splitlines():
            if line.startswith("#"):
                continue
            a, b, c = line.split()
            hits.append((int(a), int(b), int(c)))
        return sorted(hits)
