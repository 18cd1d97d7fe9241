#!/usr/bin/env python3
"""Static instruction histogram of a gfx950 kernel's basic blocks: per block the wave-level instruction
counts by issue class, so that the VALU bound of a kernel becomes ONE computed number -- sum over classes of count x measured
time per instruction (scripts/probe/valu_rate.hip -> profiles/r03_probe_valu_rate.txt) -- instead of a full-rate .. half-rate
bracket.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S csrc/pm_pair.hip -o /tmp/pm_pair.s
    python scripts/isa_histogram.py /tmp/pm_pair.s pm_pair_scan [--blocks]

Issue classes (time per wave64 instruction and SIMD with 4 waves per SIMD, the occupancy of these kernels; the probe
reports cycles at a nominal 2.4 GHz, i.e. times):
  full   VOP1 / VOP2 encodings of v_and, v_or, v_xor, v_add, v_sub(rev), v_lshrrev, v_ashrrev, v_mov, v_not with VGPR /
         inline-constant / literal operands                                              2.45 cyc = 1.02 ns
  half   every other VALU instruction: VOP3 forms (v_alignbit, v_bfe, v_bcnt, v_and_or, v_or3, v_lshl_or, v_perm ...),
         left shifts, v_mul_u32_u24, v_min/max, v_ffbl, DPP / SDWA forms, v_readlane / v_cndmask / v_cmp, and any of the
         `full` opcodes with an SGPR operand                                              4.27 cyc = 1.78 ns
  lds / vmem / salu / other: counted, not priced here (their pipes are separate)."""
import collections
import re
import sys

FULL_OPS = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
            "v_not_b32", "v_add_co_u32", "v_addc_co_u32"}
T_FULL_NS, T_HALF_NS = 2.45 / 2.4, 4.27 / 2.4


def classify(op, operands):
    if op.startswith("v_"):
        base = op
        for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
            if base.endswith(suf):
                base = base[: -len(suf)]
        if op.endswith(("_dpp", "_sdwa", "_e64")):
            return "half"
        if base in FULL_OPS and not re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc|\bexec|\bm0\b", operands.split(",", 1)[1] if "," in operands else ""):
            return "full"
        return "half"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def blocks_of(path, kernel, with_loops=False):
    """[(label, [(op, operands)])] of the kernel whose mangled name contains `kernel`; with_loops: a third element
    (header label, depth) from the compiler's loop comments ("in Loop: Header=BB1_32 Depth=2" / "This Loop Header: Depth=2")"""
    out, cur, inside = [], None, False
    pending = None
    for line in open(path):
        line = line.rstrip("\n")
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = kernel in m.group(1) and not m.group(1).endswith(".kd")
            if inside:
                cur = [m.group(1), [], (None, 0)]
                out.append(cur)
            continue
        if not inside:
            continue
        if re.match(r"^\s*\.Lfunc_end", line):
            inside = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", line)
        if m:
            cur = [m.group(1), [], (None, 0)]
            out.append(cur)
            line = m.group(2)
        m = re.match(r"^; %bb\.(\d+):(.*)$", line)
        if m:                                                       # a fall-through block without a label of its own
            cur = ["%bb." + m.group(1), [], (None, 0)]
            out.append(cur)
            line = m.group(2)
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", line)
        if m:
            cur[2] = (m.group(1), int(m.group(2)))
            continue
        m = re.search(r"This Loop Header: Depth=(\d+)", line)
        if m:
            cur[2] = (cur[0].lstrip(".L"), int(m.group(1)))
            continue
        m = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", line)
        if m and cur[2][0] is None:
            cur[2] = ("parent:" + m.group(1), int(m.group(2)))     # refined by the "This Loop Header" line that follows
            continue
        m = re.match(r"^\s+([a-z]\w+)\s*(.*?)\s*(;.*)?$", line)
        if m and not m.group(1).startswith("."):
            cur[1].append((m.group(1), m.group(2)))
    return out if with_loops else [(b[0], b[1]) for b in out]


def summarize(ins):
    c = collections.Counter()
    for op, operands in ins:
        c[classify(op, operands)] += 1
    return c


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    bl = blocks_of(path, kernel)
    if "--blocks" in sys.argv:
        for label, ins in bl:
            c = summarize(ins)
            ops = collections.Counter(op for op, _ in ins)
            sig = "dsr32=%d gld2=%d gld1=%d dsw=%d atom=%d" % (ops["ds_read_b32"], ops["global_load_dwordx2"], ops["global_load_dword"],
                                                                sum(v for k, v in ops.items() if k.startswith("ds_write")), sum(v for k, v in ops.items() if "atomic" in k))
            br = [o for op, o in ins if op.startswith("s_cbranch") or op == "s_branch"]
            print("%-12s n=%4d full=%3d half=%3d lds=%3d vmem=%2d salu=%3d  %s  -> %s" % (label, len(ins), c["full"], c["half"], c["lds"], c["vmem"], c["salu"], sig, ",".join(br)))
        return
    if "--pair" in sys.argv:
        return pair_report(path, kernel)
    tot = collections.Counter()
    for _, ins in bl:
        tot += summarize(ins)
    print(dict(tot))


def pair_report(path, kernel):
    """pm_pair_scan: per field pair (one main loop each: the depth-1 loop that holds pass A, 16 ds_read_b32 in one block) the
    instructions of one block of 1024 positions -- the loop's own blocks -- and of one pair of rounds -- the blocks of the inner
    do-while (depth 2), a third of them per trip (three register-ring phases) -- without the suspect path (blocks with
    v_mbcnt / ds_write / atomics and the loops below depth 2).  Then the whole launch at R rounds per block."""
    args = dict(a.split("=") for a in sys.argv if a.startswith("--") and "=" in a)
    R = float(args.get("--rounds", 4.216))                         # rounds per block and wave (bench.py --pair-stats)
    bases = float(args.get("--bases", 3e9))
    kms = float(args.get("--kernel-ms", 14.6))
    bl = blocks_of(path, kernel, with_loops=True)
    # loop nest: header label -> parent header
    parent = {}
    for i, (label, ins, (hdr, depth)) in enumerate(bl):
        pass
    # depth-1 headers that own a pass A block
    def top_of(idx):
        """the depth-1 header a block belongs to: walk back to the nearest block that IS a depth-1 header"""
        for j in range(idx, -1, -1):
            lab, _, (hdr, depth) = bl[j]
            if depth == 1 and hdr == lab.lstrip(".L"):
                return lab
        return None
    mains = {}
    for i, (label, ins, (hdr, depth)) in enumerate(bl):
        ops = collections.Counter(op for op, _ in ins)
        if ops["ds_read_b32"] >= 16 and depth == 1:
            mains[top_of(i)] = None
    rows = []
    for top in mains:
        start = [i for i, b in enumerate(bl) if b[0] == top][0]
        per_block, per_pair = collections.Counter(), collections.Counter()
        i = start
        while i < len(bl):
            label, ins, (hdr, depth) = bl[i]
            if i > start and depth == 0:
                break
            if i > start and depth == 1 and hdr == label.lstrip(".L"):
                break                                               # the next depth-1 loop
            ops = collections.Counter(op for op, _ in ins)
            rare = any(k.startswith(("v_mbcnt", "ds_write", "global_atomic", "global_store")) for k in ops)
            c = summarize(ins)
            if depth == 1:
                per_block += c
            elif depth == 2 and not rare:
                per_pair += c
            i += 1
        rows.append((top, per_block, per_pair))
    print("# pm_pair_scan, static counts per wave (wave-level instructions), suspect path left out; R = %.3f rounds per block" % R)
    print("%-12s | %-44s | %-44s" % ("main loop", "one block of 1024 positions (pass A + overhead)", "one PAIR of rounds (mean of the three phases)"))
    tot = collections.Counter()
    for top, b, p in rows:
        pp = {k: v / 3.0 for k, v in p.items()}
        print("%-12s | full %3d half %3d lds %2d vmem %d salu %3d        | full %5.1f half %5.1f lds %3.1f vmem %3.1f salu %5.1f" % (
            top, b["full"], b["half"], b["lds"], b["vmem"], b["salu"], pp.get("full", 0), pp.get("half", 0), pp.get("lds", 0), pp.get("vmem", 0), pp.get("salu", 0)))
        for k in ("full", "half", "lds", "vmem", "salu"):
            tot[k] += (b[k] + R / 2.0 * pp.get(k, 0)) / len(rows)
    wave_blocks = bases / 1024.0 * len(rows)
    print("per block and wave at R rounds (mean over the %d field pairs): full %.1f half %.1f  (VALU %.1f)  lds %.1f vmem %.1f salu %.1f" % (
        len(rows), tot["full"], tot["half"], tot["full"] + tot["half"], tot["lds"], tot["vmem"], tot["salu"]))
    valu = (tot["full"] + tot["half"]) * wave_blocks
    t_ns = (tot["full"] * T_FULL_NS + tot["half"] * T_HALF_NS) * wave_blocks / 1024.0      # per SIMD: 256 CUs x 4
    print("launch of %.3g bases x %d field pairs = %.4g wave-blocks: %.3g VALU wave-instructions (compare SQ_INSTS_VALU), %.0f %% of them half rate" % (
        bases, len(rows), wave_blocks, valu, 100.0 * tot["half"] / (tot["full"] + tot["half"])))
    print("VALU issue time per SIMD at the probe's rates (full %.2f ns, half %.2f ns per wave-instruction): %.2f ms = %.3f of the kernel's %.2f ms" % (
        T_FULL_NS, T_HALF_NS, t_ns * 1e-6, t_ns * 1e-6 / kms, kms))


if __name__ == "__main__":
    main()
