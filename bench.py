#!/usr/bin/env python3
"""bench.py -- primer_match scan throughput on MI355X.

Metric (BASELINE.json): Gbases/s scanned, 100k x 20-mer primers (both strands, so 200k patterns),
k <= 2 mismatches, synthetic DNA database; one "step" = one full pass of the hot path over the
database: scan kernel(s) -> candidate records -> (N>1: RCCL gather to rank 0) -> host
cluster/verify on rank 0 -> final hits resident on the host of rank 0.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The database is sharded by stream position across ranks: rank r holds and scans
stream range [r*S, (r+1)*S) plus a 256-byte halo on either side; every rank holds all patterns.
Default scaling is strong: the --db-bases database (3 Gbp, BASELINE.json's "3 Gbp DB at 1/2/4/8 GPUs")
is split over the ranks; --scaling weak scans --db-bases per GPU instead.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sat_amd  # noqa: E402

TABLE = b"ACGT\n"           # compress_seq -n true codes: A0 C1 G2 T3 EOS4 (compress_seq.cc:704-719)
EOS = 4
HALO = 256
GUARD = 1 << 16          # guard band of a shard: chains of candidates shorter than this are decided locally
BLOCK = 1 << 24             # generation granule: block b of the global stream is seeded with (seed, b)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 32-bit integer lane-ops/s: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz


STYLES = ("uniform", "skew", "vocab", "tandem")


def gen_block(b, seed, device, style):
    """block b (BLOCK bases) of the global stream.  uniform: i.i.d. A/C/G/T (BASELINE.json's synthetic database).  The other
    three are the generators of tests/adversarial.py (what scripts/fuzz_families.py found seven parity bugs with) at
    database size, with the parameters at the hard end that still give an output a host can take:
    skew   = i.i.d. with composition A .40 C .10 G .10 T .40;
    vocab  = words of one vocabulary of 200 twelve-base words (the same for every block), 2 % point mutations;
    tandem = tandem repeats: units of 1..39 random bases, 1..399 copies each, 3 % of the bases drifted."""
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000003 + b)
    if style == "uniform":
        return torch.randint(0, 4, (BLOCK,), dtype=torch.uint8, device=device, generator=g)
    if style == "skew":
        u = torch.rand(BLOCK, device=device, generator=g)
        edges = torch.tensor([0.40, 0.50, 0.60], device=device)
        return torch.bucketize(u, edges).to(torch.uint8)
    if style == "vocab":
        gv = torch.Generator(device=device)
        gv.manual_seed(seed * 7919 + 17)
        wl, nv = 12, 200
        vocab = torch.randint(0, 4, (nv, wl), dtype=torch.uint8, device=device, generator=gv)
        idx = torch.randint(0, nv, (BLOCK // wl + 1,), device=device, generator=g)
        blk = vocab[idx].reshape(-1)[:BLOCK].clone()
    else:                                                          # tandem
        nseg = BLOCK // 20 + 64                                    # more segments than the block can hold (>= 1 base each)
        ulen = torch.randint(1, 40, (nseg,), device=device, generator=g)
        reps = torch.randint(1, 400, (nseg,), device=device, generator=g)
        ends = torch.cumsum(ulen * reps, 0)
        units = torch.randint(0, 4, (nseg, 40), dtype=torch.uint8, device=device, generator=g)
        pos = torch.arange(BLOCK, device=device)
        seg = torch.searchsorted(ends, pos, right=True)
        start = torch.where(seg > 0, ends[(seg - 1).clamp(min=0)], torch.zeros_like(pos))
        blk = units[seg, (pos - start) % ulen[seg]]
    m = torch.rand(BLOCK, device=device, generator=g) < (0.02 if style == "vocab" else 0.03)
    blk[m] = torch.randint(0, 4, (int(m.sum().item()),), dtype=torch.uint8, device=device, generator=g)
    return blk


def gen_stream(lo, hi, total, entries, seed, device, style="uniform", run=None):
    """Codes of global stream range [lo,hi): A/C/G/T of the given style with an EOS at index 0 and after each of
    `entries` equal-length entries (layout of compress_seq's .sqn).  run = (start, length): that many A's from `start`
    on (a homopolymer run -- a centromeric satellite in the small -- for the cut-chain test of the sharded step)."""
    out = torch.empty(hi - lo, dtype=torch.uint8, device=device)
    b0, b1 = lo // BLOCK, (hi - 1) // BLOCK
    for b in range(b0, b1 + 1):
        blk = gen_block(b, seed, device, style)
        s, e = max(lo, b * BLOCK), min(hi, (b + 1) * BLOCK)
        out[s - lo:e - lo] = blk[s - b * BLOCK:e - b * BLOCK]
    if run:
        a, z = max(lo, run[0]), min(hi, run[0] + run[1])
        if z > a:
            out[a - lo:z - lo] = 0
    elen = (total - 1) // entries
    eos_pos = torch.arange(0, entries + 1, device=device, dtype=torch.int64) * elen
    eos_pos[-1] = total - 1
    m = (eos_pos >= lo) & (eos_pos < hi)
    out[(eos_pos[m] - lo)] = EOS
    return out


def make_primers(stream0, n_primers, L, seed, from_stream=0.1):
    """90 % i.i.d. random 20-mers, 10 % sampled from the database and given 0/1/2 substitutions
    (SURVEY 8d); from_stream = 1.0: every primer cut from the database (the hard streams of --stream-style).  Returns (list of str, planted) with planted = [(primer index, stream index of the
    site's first base, Hamming distance of the primer to the site)]."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    pri = rng.integers(0, 4, size=(n_primers, L), dtype=np.uint8)
    n_pl = int(n_primers * from_stream)
    host = stream0.cpu().numpy()
    placed = 0
    planted = []
    while placed < n_pl:
        a = int(rng.integers(1, host.size - L - 1))
        w = host[a:a + L]
        if (w > 3).any():
            continue
        w = w.copy()
        for _ in range(int(rng.integers(0, 3))):
            i = int(rng.integers(0, L))
            w[i] = (w[i] + 1 + int(rng.integers(0, 3))) % 4
        pri[placed] = w
        planted.append((placed, a, int((w != host[a:a + L]).sum())))
        placed += 1
    return [lut[r].tobytes().decode() for r in pri], planted


class CudaArray:
    """Expose a raw HBM pointer to torch through __cuda_array_interface__ (for the RCCL gather)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def host_cores():
    """CPUs this process can really use: its affinity mask, capped by the cgroup's CPU quota (a GPU box
    hands a job a share of the host, not the host)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(args, stream0, primers_fwd, log):
    """The reference CPU path on a bounded sample of the same workload (BASELINE.md
    "CPU-baseline plan"): the real `primer_match -c` (oracle/_ref/, the reference compiled where it
    lies by oracle/Makefile; its own compress_seq makes the database files), the same primers, both
    strands, automatic engine, 1 thread (the reference has no parallelism).  Scan time is separated
    from the pattern-index build by timing the same command on a 2 kb database as well; the "all host
    cores" figure is one reference process per core this process may run on, each on its own slice."""
    k, indels = args.k, bool(args.indels)
    sample = args.cpu_sample
    if sample <= 0:
        sample = {0: 40_000_000, 1: 20_000_000}.get(k, 100_000)      # ~10-30 s of single-thread CPU work
    sample = min(sample, stream0.numel())
    ref = os.path.join(ROOT, "oracle", "_ref", "primer_match")
    cseq = os.path.join(ROOT, "oracle", "_ref", "compress_seq")
    lut = np.frombuffer(b"ACGT\n", dtype=np.uint8)

    def write_db(path, codes):
        """FASTA of a slice of the stream (entries end at its EOS codes) -> compress_seq -n true"""
        text = lut[codes]
        with open(path, "wb") as f:
            at = 0
            for e, piece in enumerate(text.tobytes().split(b"\n")):
                if piece:
                    f.write(b">e%d\n" % e)
                    f.write(piece)
                    f.write(b"\n")
        r = subprocess.run([cseq, "-i", path, "-n", "true"], capture_output=True)
        return r.returncode == 0

    def command(db, pat):
        cmd = [ref, "-i", db, "-P", pat, "-r", "-c"]
        if k:
            cmd += ["-k" if indels else "-K", str(k)]
        return cmd

    if not (os.path.exists(ref) and os.path.exists(cseq)):
        from oracle import pmoracle as O
        codes = stream0[:sample].cpu().numpy()
        allp = primers_fwd + [sat_amd.reverse_comp(p) for p in primers_fwd]
        text = O.Text(codes, TABLE)
        dt, _ = O.time_find_all(text, allp, engine=O.pick_engine(text, allp, k, indels), k=k, indels=indels)
        return {"value": sample / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
                "sample": "first %d bases of the rank-0 shard, all %d primers x 2 strands, scan only %.1f s" % (sample, len(primers_fwd), dt)}
    ncores = host_cores()
    if args.cpu_procs > 0:
        ncores = min(ncores, args.cpu_procs)
    with tempfile.TemporaryDirectory() as d:
        pat = os.path.join(d, "pat.txt")
        with open(pat, "w") as f:
            f.write("\n".join(primers_fwd) + "\n")
        if not (write_db(os.path.join(d, "tiny.fa"), stream0[:2048].cpu().numpy()) and write_db(os.path.join(d, "db.fa"), stream0[:sample].cpu().numpy())):
            log("cpu_baseline: compress_seq failed")
            return None
        log("cpu_baseline: reference primer_match on 2 kb (pattern index build) ...")
        t0 = time.time()
        r0 = subprocess.run(command(os.path.join(d, "tiny.fa"), pat), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        t_build = time.time() - t0
        log("cpu_baseline: %.1f s; now on %d bases ..." % (t_build, sample))
        t0 = time.time()
        r1 = subprocess.run(command(os.path.join(d, "db.fa"), pat), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        t_all = time.time() - t0
        log("cpu_baseline: %.1f s; %d processes next" % (t_all, ncores))
        if r0.returncode or r1.returncode:
            log("cpu_baseline: primer_match failed: " + (r1.stderr or r0.stderr).decode("latin1")[-300:])
            return None
        t_scan = max(t_all - t_build, 1e-3)
        res = {"value": sample / t_scan / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "reference",
               "sample": "reference primer_match -c on the first %d bases of the rank-0 shard, all %d primers x 2 strands: scan %.1f s "
                         "(whole process %.1f s minus %.1f s for the same command on 2 kb = pattern index build)"
                         % (sample, len(primers_fwd), t_scan, t_all, t_build),
               "value_including_index_build": sample / t_all / 1e9}
        # one reference process per core on disjoint slices of the stream (same sample size each)
        if ncores > 1 and not args.no_cpu_all:
            avail = stream0.numel()
            from concurrent.futures import ThreadPoolExecutor
            slices = [stream0[min(c * sample, max(0, avail - sample)):][:sample].cpu().numpy() for c in range(ncores)]
            with ThreadPoolExecutor(max_workers=min(ncores, 16)) as ex:
                ok = all(ex.map(lambda c: write_db(os.path.join(d, "db%d.fa" % c), slices[c]), range(ncores)))
            if ok:
                t0 = time.time()
                procs = [subprocess.Popen(command(os.path.join(d, "db%d.fa" % c), pat), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for c in range(ncores)]
                ok = all(pr.wait() == 0 for pr in procs)
                dta = time.time() - t0
                if ok:
                    res["all_cores"] = {"value": ncores * sample / max(dta - t_build, 1e-3) / 1e9, "unit": "Gbases/s", "cores": ncores,
                                        "sample": "%d reference processes (every core this process may run on), %d bases each, wall %.1f s incl. %.1f s index build each"
                                                  % (ncores, sample, dta, t_build)}
    return res


def baseline_metric():
    """BASELINE.json's metric string, verbatim."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "Gbases/s scanned, 100k\u00d720-mer primers k\u22642, 3 Gbp DB at 1/2/4/8 GPUs"


def code_sha():
    """sha256 over the product's kernel and library sources (csrc/*.hip, *.cpp, *.h): what a counter file was collected on.
    The GPU box has no .git, so this -- not the commit -- is what bench.py can compare at run time; scripts/collect_profiles.py
    adds the commit (`head`) when it copies the counters into profiles/."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "sequence-alignment-tools_amd", "csrc", "*.*"))):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def profiled_entry(pattern, args, shard):
    """the newest committed counter entry (profiles/<pattern>) of this exact workload, with the file it came from"""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        with open(f) as fh:
            for e in json.load(fh).get("entries", []):
                if (e["k"], e["indels"], e["db_bases"], e["primers"]) == (args.k, args.indels, shard, args.primers) and \
                        e.get("stream_style", "uniform") == args.stream_style:
                    best = (e, "profiles/" + os.path.basename(f))
    return best


def staleness(e, kms):
    """Were these counters collected on the code that runs now, and does the kernel still take what it took then?"""
    sha = code_sha()
    at = e.get("kernel_ms_at_profile")
    drift = None if not at or not (kms > 0) else abs(kms - at) / at
    return {"head": e.get("head"), "code_sha_at_profile": e.get("code_sha"), "code_sha_now": sha, "kernel_ms_at_profile": at,
            "stale": bool(e.get("code_sha") != sha or drift is None or drift > 0.10)}


def measured_traffic(args, shard):
    """HBM/fabric bytes per launch of the scan kernel from the committed PMC passes
    (profiles/traffic_r*.json, produced by scripts/pmc_traffic.sh), or None when this exact
    workload was not profiled."""
    best = profiled_entry("traffic_r*.json", args, shard)
    return (None, None, None) if best is None else (best[0]["traffic_bytes"], best[1], best[0])


def issue_roofline(args, shard, kms):
    """SURVEY 8(d) "honest secondary bound" for the seed family: the kernels are not HBM bound; which on-chip
    pipe is how busy, from the committed PMC passes of this exact workload (profiles/issue_r*.json, made by
    scripts/pmc_issue.sh) and THIS run's kernel time, at the 2.4 GHz peak clock (the chip runs ~2.05 GHz under
    these kernels, so every fraction is a lower bound):
      valu  = VALU wave-instructions x cycles per instruction / (256 CUs x 4 SIMDs x kernel cycles).  gfx950 issues
              v_and/v_or/v_xor/v_add/v_sub/v_lshrrev/v_mov at one wave64 instruction per ~2 cycles and SIMD, but every
              three-operand (VOP3) form, v_bcnt, v_bfe, v_alignbit, v_min/max, v_mul_u32_u24, left shifts, SDWA / DPP
              forms and anything with an SGPR operand at one per ~4 (profiles/r03_probe_valu_rate.txt): both bounds given;
              these kernels are mostly made of the second kind
      lds   = LDS-array active cycles (incl. bank conflicts), summed over the CUs / (256 x kernel cycles)
      l1_l2 = L1 -> L2 read requests (one 128-byte line each) x 2 cycles / (256 CUs x kernel cycles): the L2 -> L1
              return path moves 64 B per clock and CU (269 G lines/s for the chip, profiles/r03_probe_tcp_gather.txt)."""
    best = profiled_entry("issue_r*.json", args, shard)
    if best is None or not (kms > 0):
        return None
    e, src = best
    cyc = kms * 1e-3 * 2.4e9
    out = {"source": src, "kernel": e.get("kernel"), "clock_ghz": 2.4, "kernel_ms": kms,
           "note": "counters of the scan kernel alone; kernel_ms (HIP events) also holds the verify kernel behind it"}
    fr = {}
    if e.get("SQ_INSTS_VALU"):
        out["valu"] = {"wave_instructions": e["SQ_INSTS_VALU"], "frac_if_all_full_rate": e["SQ_INSTS_VALU"] * 2 / (256 * 4 * cyc),
                       "frac_if_all_half_rate": e["SQ_INSTS_VALU"] * 4 / (256 * 4 * cyc)}
        fr["valu"] = out["valu"]["frac_if_all_half_rate"]
        # one number instead of the bracket where the kernel's instruction mix has been counted (scripts/isa_histogram.py ->
        # profiles/r04_isa_hist_pair_scan.txt): wave-instructions x the probe's TIME per instruction (1.02 ns full rate, 1.78 ns
        # half rate per SIMD at four waves per SIMD: profiles/r03_probe_valu_rate.txt) -- times, so no clock assumption
        half = {"pm_pair_scan": 0.57, "pm_pair_edit_scan": 0.64}.get(e.get("kernel"))
        if half is not None:
            t_ns = (1.0 - half) * (2.45 / 2.4) + half * (4.27 / 2.4)
            out["valu"]["half_rate_share_static"] = half
            out["valu"]["frac_at_measured_mix"] = e["SQ_INSTS_VALU"] * t_ns * 1e-9 / (256 * 4) / (kms * 1e-3)
            fr["valu"] = out["valu"]["frac_at_measured_mix"]
    if e.get("SQ_LDS_IDX_ACTIVE"):
        out["lds"] = {"active_cycles": e["SQ_LDS_IDX_ACTIVE"], "bank_conflict_cycles": e.get("SQ_LDS_BANK_CONFLICT"),
                      "frac": e["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)}
        fr["lds"] = out["lds"]["frac"]
    if e.get("TCP_TCC_READ_REQ_sum"):
        out["l1_l2"] = {"read_requests": e["TCP_TCC_READ_REQ_sum"], "l2_misses": e.get("TCC_MISS_sum"),
                        "frac": e["TCP_TCC_READ_REQ_sum"] * 2 / (256 * cyc),
                        # the probe's price of one distinct 128-byte line of a wave-level gather: 0.95 CU-ns (r03_probe_tcp_gather.txt)
                        "frac_at_probe_rate": e["TCP_TCC_READ_REQ_sum"] * 0.95e-9 / 256 / (kms * 1e-3)}
        fr["l1_l2"] = out["l1_l2"]["frac_at_probe_rate"]
    if fr:
        out["binding"] = max(fr, key=fr.get)
    out.update(staleness(e, kms))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--db-bases", type=int, default=3_000_000_000, help="stream bytes in total (strong, the default) or per GPU (weak)")
    ap.add_argument("--entries", type=int, default=24)
    ap.add_argument("--primers", type=int, default=100_000)
    ap.add_argument("--length", type=int, default=20)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--indels", type=int, default=0, help="0: -K (mismatches), 1: -k (edits)")
    ap.add_argument("--kernel", choices=["auto", "bitpar", "seed"], default="auto")
    ap.add_argument("--odd", type=int, default=0, help="replace this many primers by ones with an N in the middle (they go to the bit-parallel residue engine)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--cpu-sample", type=int, default=0, help="bases for the CPU baseline (0 = auto, -1 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-all", action="store_true", help="skip the one-reference-process-per-core figure")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes for the all-cores figure (0 = every host core)")
    ap.add_argument("--no-check", action="store_true", help="do not require the planted primer sites in the result (kernel stage measurements with PM_SEED_DEBUG)")
    ap.add_argument("--dump-hits", default="", help="rank 0 writes the final hits of the last step (global stream indices, sorted) to this .npy file")
    ap.add_argument("--stream-style", choices=list(STYLES), default="uniform", help="composition of the synthetic stream (gen_block)")
    ap.add_argument("--primer-source", choices=["auto", "mixed", "stream"], default="auto",
                    help="mixed: 10 %% of the primers cut from the stream (SURVEY 8d); stream: all of them; auto: mixed on the uniform stream, stream on the others")
    ap.add_argument("--landing", type=int, default=1 << 24, help="records the host landing zones hold (hit-dense streams need more)")
    ap.add_argument("--pair-stats", action="store_true", help="count blocks, rounds and key hits in the pair kernel (PM_SEED_DEBUG bit 5; a measurement build of the same kernel)")
    ap.add_argument("--plant-run", type=int, default=0, help="A x this many across the middle of the stream, and the primer A x length with it")
    ap.add_argument("--scan-passes", type=int, default=3, help="timed whole-stream passes through pm_scan itself after the timed region (0 = skip; single GPU only)")
    ap.add_argument("--scan-chunk", type=int, default=1 << 30, help="stream bytes per pm_scan range (the compiled plugin default: 1 GiB)")
    ap.add_argument("--scan-cap", type=int, default=1 << 20, help="records the caller takes per pm_scan call")
    ap.add_argument("--capacity", type=int, default=0, help="initial record capacity (0 = default; small values exercise the grow-and-rescan path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("PM_BENCH_BACKEND", "nccl")   # "gloo": rehearse N>1 on a 1-GPU box (collectives via host)
    local = local % ndev if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # PM_BENCH_FORCE_DIST=1: take the N > 1 code (process group, count exchange, owned finalize, gather,
    # side-stream landing) with a world of one -- the only way to execute the RCCL path on a 1-GPU box
    force_dist = world == 1 and os.environ.get("PM_BENCH_FORCE_DIST", "") == "1"
    use_dist = world > 1 or force_dist
    dist = None
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    cdev = dev if backend == "nccl" else torch.device("cpu")       # where collective buffers live
    log = (lambda s: print("[bench] " + s, file=sys.stderr, flush=True)) if rank == 0 else (lambda s: None)

    # ---- synthetic inputs (untimed) ----------------------------------------------------------
    total = args.db_bases * world if args.scaling == "weak" else args.db_bases
    shard = (total + world - 1) // world
    lo, hi = rank * shard, min(total, (rank + 1) * shard)
    # a shard holds its own bytes, a guard band either side (filter_bitvec chains that straddle a shard
    # edge are decided by the shard that owns the hit, pm_finalize_device_owned) and a halo of text
    # the windows / seed extensions of the outermost candidates read
    glo = max(0, lo - GUARD - HALO)
    ghi = min(total, hi + GUARD + HALO)
    run = None
    if args.plant_run > 0:                                          # a homopolymer run across the middle of the stream (= a shard edge of 2, 4, 8 ranks)
        run = (total // 2 - args.plant_run // 2, args.plant_run)
    stream = gen_stream(glo, ghi, total, args.entries * (world if args.scaling == "weak" else 1), 20260101, dev, args.stream_style, run)
    n_bases_total = total - (args.entries * (world if args.scaling == "weak" else 1) + 1)
    planted = []
    if rank == 0:
        # sampled from a prefix every launch geometry holds on rank 0 (so that 1 and N ranks of a
        # --scaling strong run search the same primers)
        src = {"mixed": 0.1, "stream": 1.0, "auto": 0.1 if args.stream_style == "uniform" else 1.0}[args.primer_source]
        primers, planted = make_primers(stream[:min(stream.numel(), 1 << 26, max(total // 8, 1 << 16))], args.primers, args.length, 7, src)
        if args.plant_run > 0:
            primers[-1] = "A" * args.length                          # its candidates chain along the whole run
    else:
        primers = None
    if use_dist:
        box = [primers]
        dist.broadcast_object_list(box, src=0, device=cdev)
        primers = box[0]
    if args.odd:
        primers = [p[:len(p) // 2] + "N" + p[len(p) // 2 + 1:] if i < args.odd else p for i, p in enumerate(primers)]
    allp = primers + [sat_amd.reverse_comp(p) for p in primers]
    kern = {"auto": sat_amd.KERNEL_AUTO, "bitpar": sat_amd.KERNEL_BITPAR, "seed": sat_amd.KERNEL_SEED}[args.kernel]
    if args.pair_stats:                                             # the library reads its measurement knobs once, in pm_create
        os.environ["PM_SEED_DEBUG"] = str(int(os.environ.get("PM_SEED_DEBUG", "0")) | 32)
    pm = sat_amd.PatternMatch(k=args.k, indels=bool(args.indels), kernel=kern, device=local)
    for i, p in enumerate(allp):
        pm.add_pattern(p, i + 1)
    t0 = time.time()
    pm.init_device(stream.data_ptr(), stream.numel(), TABLE, stream=torch.cuda.current_stream().cuda_stream, keepalive=stream)
    # record buffer: the edit-distance seed plan reports a candidate through several seeds before its dedup
    pm.set_capacity(args.capacity if args.capacity > 0 else (1 << 28 if (args.indels and args.k >= 2) else 1 << 24))
    log("index build %.2f s; semantics=%s kernel=%s" % (time.time() - t0, *pm.selected()))
    begin, end = lo - glo, hi - glo

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = []
    exch_host_ms = []                                               # per step: host wall time inside the exchange calls (counts + gather)
    exch_events = []                                                # nccl: (start, end) events on the current stream around the record gather
    final_hits = [0]
    cand_count = [0]
    rescans = [0]
    cut_steps = [0]                                                 # steps that fell back to the gather-candidates form (cut chain)
    last_final = [None]                                             # rank 0: final hits of the last step (host array or pinned int64 pairs)

    # host landing zone for final hits (pinned: the copy out of HBM is part of every step)
    out_pin = torch.empty(args.landing * 16, dtype=torch.uint8, pin_memory=True)
    out_buf = out_pin.numpy().view(sat_amd.HIT_DTYPE)
    # filter_bitvec: every rank clusters and verifies what it owns; only final hits travel
    own_path = use_dist and pm.selected()[0] == sat_amd.SEM_FILTER_BITVEC
    g_lo = 0 if glo == 0 else begin - GUARD
    g_hi = stream.numel() if ghi == total else end + GUARD
    land_stream = torch.cuda.Stream(device=dev) if rank == 0 else None
    one_pin = torch.empty(args.landing * 2, dtype=torch.int64, pin_memory=True) if rank == 0 and not use_dist else None
    landed = [None, False]                                          # single rank: event of the copy in flight; does it read the candidate buffer itself?
    all_pin = torch.empty(args.landing * 2 * (world if own_path else 1), dtype=torch.int64, pin_memory=True) if rank == 0 and own_path else None
    dev_final = [True]                                              # GPU clustering available for this option set?

    def scan(lo_, hi_):
        """Device stage.  Returns (count, 0), or (0, needed capacity) when the record buffer was too
        small (the caller grows it and scans again -- on every rank, so that no rank waits in a
        collective for one that raised)."""
        pm.scan_async(lo_, hi_)
        try:
            ncand = pm.scan_wait()
        except sat_amd.PmError as e:
            if e.code != sat_amd.PM_E_OVERFLOW:
                raise
            return 0, max(int(e.required), 1)
        ms, _ = pm.last_kernel_time()
        kernel_ms.append(ms)
        return ncand, 0

    def grow(need):
        rescans[0] += 1
        pm.set_capacity(int(need * 1.25) + 1024)

    # exchange buffers live across steps (allocating and zeroing them per step cost more host time than the collectives):
    # my count, everybody's counts, my padded records, and on rank 0 one landing buffer per rank
    xb = {"mine": None, "all": None, "pad": None, "land": None, "cap": 0}
    if use_dist:
        xb["mine"] = torch.zeros(1, dtype=torch.int64, device=cdev)
        xb["all"] = torch.zeros(world, dtype=torch.int64, device=cdev)

    CUT = "cut"

    def exchange_counts(cnt, need, cut=False):
        """all_gather of the ranks' record counts; a rank whose scan overflowed sends -1, a rank whose owned finalize met
        a chain of candidates cut by its guard band sends -2.  Returns the counts; None when some rank has to scan again
        (then every rank does); CUT when some rank cannot decide its shard locally (then every rank takes the
        gather-candidates form for this step, as GpuPatternMatch::sharded_scan does, host/gpu_pattern_match.cc)."""
        xb["mine"].fill_(-1 if need else (-2 if cut else cnt))
        dist.all_gather_into_tensor(xb["all"], xb["mine"])
        cl = xb["all"].tolist()                                         # one host sync for all counts
        if -1 in cl:
            if need:
                grow(need)
            return None
        if -2 in cl:
            return CUT
        return cl

    def timed_gather(ptr, cnt, cl, tx0):
        """padded gather of every rank's records to rank 0 (a record = two int64 words; every rank sends max(counts)
        records, the receiver reads counts[r] of them): straight out of HBM with RCCL, through host memory only in the
        gloo rehearsal (its transport is the host).  Books the exchange's host time (from tx0, the start of the count
        exchange) and, on the nccl backend, brackets the transfer with events on the current stream."""
        mx = max(max(cl), 1)
        if mx > xb["cap"]:
            xb["cap"] = mx + mx // 4 + 1024
            xb["pad"] = torch.zeros(xb["cap"] * 2, dtype=torch.int64, device=cdev)
            xb["land"] = [torch.zeros(xb["cap"] * 2, dtype=torch.int64, device=cdev) for _ in range(world)] if rank == 0 else None
        cur = torch.cuda.current_stream()
        if rank == 0 and land_stream is not None and cdev.type == "cuda":
            cur.wait_stream(land_stream)                                # the previous step's landing still reads the buffers this gather writes
        ev = None
        if cdev.type == "cuda":
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        pad = xb["pad"][:mx * 2]
        if cnt:
            if cdev.type == "cuda":
                pad[:cnt * 2].copy_(torch.as_tensor(CudaArray(ptr, cnt * 16), device=dev).view(torch.int64))
            else:
                pad[:cnt * 2] = torch.from_numpy(pm.copy_records(ptr, cnt).view(np.int64).reshape(-1))
        gathered = [g[:mx * 2] for g in xb["land"]] if rank == 0 else None
        dist.gather(pad, gathered, dst=0)
        if ev:
            ev[1].record()
            exch_events.append(ev)
        exch_host_ms.append((time.perf_counter() - tx0) * 1e3)
        return gathered

    def finalize_rank0(ptr, cnt, scanned_to):
        """records in HBM -> final hits on the host of rank 0"""
        if dev_final[0]:
            try:
                return pm.finalize_device(scanned_to, last=True, sort=False, d_cands=ptr, n=cnt, out=out_buf)
            except sat_amd.PmError as e:
                if e.code != -2:
                    raise
                dev_final[0] = False
        if use_dist and pm.selected()[1] == sat_amd.KERNEL_BITPAR and pm.selected()[0] in (sat_amd.SEM_EXACT_HALVES, sat_amd.SEM_EXACT_BASES, sat_amd.SEM_FILTER_BITVEC) and args.indels:
            raise SystemExit("bench.py --gpus>1: this option set verifies on the host with stream text, which rank 0 does not hold (DESIGN.md 5)")
        cands = pm.copy_records(ptr, cnt)
        pm.reset()
        return pm.finalize(cands, scanned_to, last=True, sort=False)

    def step_owned():
        while True:
            ncand, need = scan(g_lo, g_hi)
            ptr, cnt, cut = 0, 0, False
            if not need:
                # sort, clustering (and for -k the cluster DPs) on this rank's GPU; the final hits stay in HBM
                try:
                    ptr, cnt = pm.finalize_device(0, sort=False, owned=(begin, end, g_lo, None if ghi == total else g_hi), keep=True)
                except sat_amd.PmError as e:
                    if e.code != -2:
                        raise
                    cut = True                                          # a same-pattern chain runs from the guard edge into the owned range
            # the path's one exchange: final hit records to rank 0 over xGMI
            tx0 = time.perf_counter()
            cl = exchange_counts(cnt, need, cut)
            if cl is CUT:
                cut_steps[0] += 1
                return step_gather()
            if cl is not None:
                break
        gathered = timed_gather(ptr, cnt, cl, tx0)
        if rank == 0:
            # landing on the host: index fix-up and copy into pinned memory on a side stream, so that it
            # overlaps the next step's scan (the timed region ends with a device-wide synchronize)
            cur = torch.cuda.current_stream()
            land = land_stream if cdev.type == "cuda" else cur
            land.wait_stream(cur)
            with torch.cuda.stream(land):
                at = 0
                for r in range(world):
                    a = gathered[r][:cl[r] * 2].view(-1, 2)
                    a[:, 0] += max(0, r * shard - GUARD - HALO)        # local -> global stream index
                    all_pin[at:at + cl[r] * 2].copy_(a.reshape(-1), non_blocking=True)   # shards are in stream order
                    at += cl[r] * 2
            if cdev.type != "cuda":
                torch.cuda.synchronize()
            cand_count[0] = at // 2
            final_hits[0] = at // 2
            last_final[0] = all_pin[:at]
        return ncand

    def step():
        if own_path:
            return step_owned()
        if not use_dist:
            cur = torch.cuda.current_stream()
            if landed[0] is not None and landed[1]:
                cur.wait_event(landed[0])                             # the copy in flight reads the record buffer this scan writes
            while True:
                ncand, need = scan(begin, end)
                if not need:
                    break
                grow(need)
            ptr, cnt = pm.candidates_device()
            cand_count[0] = cnt
            if dev_final[0] and cnt >= (1 << 16):                     # (a few thousand records: the plain copy is cheaper than the stream hand-over)
                # final hits stay in HBM (pm_final_hits_device); their copy into pinned host memory runs on a side
                # stream beside the NEXT step's scan (the timed region ends with a device-wide synchronize, so the
                # last copy is inside it)
                try:
                    if landed[0] is not None:
                        cur.wait_event(landed[0])                     # the finalize stage reuses the buffer the copy in flight reads
                    fptr, fcnt = pm.finalize_device(end, last=True, sort=False, d_cands=ptr, n=cnt, keep=True)
                    if fcnt * 2 > one_pin.numel():
                        raise SystemExit("bench.py: more final hits than the host landing zone holds")
                    land_stream.wait_stream(cur)
                    with torch.cuda.stream(land_stream):
                        if fcnt:
                            one_pin[:fcnt * 2].copy_(torch.as_tensor(CudaArray(fptr, fcnt * 16), device=dev).view(torch.int64), non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(land_stream)
                    landed[0], landed[1] = ev, fptr == ptr
                    last_final[0] = one_pin[:fcnt * 2]
                    final_hits[0] = fcnt
                    return ncand
                except sat_amd.PmError as e:
                    if e.code != -2:
                        raise
                    dev_final[0] = False
            last_final[0] = finalize_rank0(ptr, cnt, end)
            final_hits[0] = last_final[0].size
            return ncand
        return step_gather()

    merge = [None]

    def merge_handle():
        """rank 0, first use: a host-stage handle over the WHOLE stream (pm_init_host: pattern tables, no upload), for the
        option sets whose verify reads stream text -- the counterpart of the merge handle of the command lines
        (host/gpu_pattern_match.cc).  The synthetic stream is generated once more, block by block, into host memory."""
        if merge[0] is None:
            ent = args.entries * (world if args.scaling == "weak" else 1)
            host = np.empty(total, dtype=np.uint8)
            for b0 in range(0, total, BLOCK * 8):
                b1 = min(total, b0 + BLOCK * 8)
                host[b0:b1] = gen_stream(b0, b1, total, ent, 20260101, dev, args.stream_style, run).cpu().numpy()
            m = sat_amd.PatternMatch(k=args.k, indels=bool(args.indels), kernel=kern, device=local)
            for i, p in enumerate(allp):
                m.add_pattern(p, i + 1)
            m.init_host(host, TABLE)
            merge[0] = m
        return merge[0]

    def step_gather():
        """candidate records of every rank's own range -> rank 0, which clusters / verifies them in stream order"""
        while True:
            ncand, need = scan(begin, end)
            ptr, cnt = (0, 0) if need else pm.candidates_device()
            # the path's one real exchange: variable-length hit records to rank 0 over xGMI
            tx0 = time.perf_counter()
            cl = exchange_counts(cnt, need)
            if cl is not None:
                break
        gathered = timed_gather(ptr, cnt, cl, tx0)
        if rank == 0:
            parts = []
            for r in range(world):
                a = gathered[r][:cl[r] * 2].view(-1, 2).to(dev)
                a[:, 0] += max(0, r * shard - GUARD - HALO)          # local -> global stream index
                parts.append(a)
            allrec = torch.cat(parts).contiguous()
            torch.cuda.current_stream().synchronize()
            tot = allrec.shape[0]
            cand_count[0] = tot
            scanned = int(total)
            if own_path and args.indels:
                # filter_bitvec -k: the cluster DPs read the text around global positions, which no single GPU holds
                m = merge_handle()
                m.reset()
                last_final[0] = m.finalize(allrec.cpu().numpy().view(sat_amd.HIT_DTYPE).reshape(-1), scanned, last=True, sort=False)
            else:
                last_final[0] = finalize_rank0(allrec.data_ptr(), tot, scanned)
            final_hits[0] = last_final[0].size
        return ncand

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- the reference-facing call: PatternMatch::find_patterns = pm_scan (pattern_match.h:131, primer_match.cc:1118) ----
    # One whole-stream pass as the compiled plugin makes it (host/plugin/gpu_pattern_match.cc): pm_reset, then consecutive
    # ranges of --scan-chunk stream bytes, every range drained into the CALLER's (pageable) array, hits sorted by (end, pid).
    # Wall clock from pm_reset to the last record resident in that array; not the headline `value` (which times the
    # sharded step the multi-GPU contract describes), reported beside it.
    desc = pm.describe()                                            # (the timed steps' launch geometry, not the pm_scan leg's ranges)
    scan_stats = pm.scan_stats()
    pm_scan_ms, pm_scan_hits, scan_hits_arr = None, None, None
    if not use_dist and args.scan_passes > 0:
        torch.cuda.synchronize()
        caller = np.zeros(args.landing, dtype=sat_amd.HIT_DTYPE)
        times = []
        for _ in range(args.scan_passes + 1):                       # first pass untimed (buffers grow to their working size)
            pm.reset()
            t0s = time.perf_counter()
            got, pos = 0, begin
            while pos < end:
                e2 = min(end, pos + args.scan_chunk)
                cnt, more = pm.scan(pos, e2, caller[got:got + args.scan_cap])
                got += cnt
                while more:
                    cnt, more = pm.scan(e2, e2, caller[got:got + args.scan_cap])
                    got += cnt
                pos = e2
            times.append((time.perf_counter() - t0s) * 1e3)
        pm_scan_ms, pm_scan_hits = float(np.mean(times[1:])), got
        scan_hits_arr = caller[:got]
        pm.reset()

    found_planted = None
    if rank == 0 and last_final[0] is not None:
        torch.cuda.synchronize()
        lf = last_final[0]
        fin = (lf.numpy().view(sat_amd.HIT_DTYPE) if isinstance(lf, torch.Tensor) else lf).copy()
        fin = fin[np.lexsort((fin["k"], fin["pid"], fin["end"]))]
        if args.dump_hits:
            np.save(args.dump_hits, fin)
        if scan_hits_arr is not None:
            # the same hits, already in (end, pid, k) order, through pm_scan as through the timed step
            same = scan_hits_arr.size == fin.size and bool((scan_hits_arr["end"] == fin["end"]).all() and (scan_hits_arr["pid"] == fin["pid"]).all()
                                                         and (scan_hits_arr["k"] == fin["k"]).all())
            if not same:
                raise SystemExit("bench.py: pm_scan returned %d hits that differ from the %d of the timed step (or are not sorted)" % (scan_hits_arr.size, fin.size))
        # every planted site within reach (distance <= k) must be reported: primer i (forward strand,
        # id i+1) with at most its planted distance, at the site's end -- filter_bitvec reports one hit
        # per chain of candidates and exact_halves drops hits within 2k of the last kept one, so the
        # reported end may sit up to 2k+1 from the planted one
        if planted and not args.odd and not args.no_check:
            tol = 2 * args.k + 1
            key = fin["pid"].astype(np.int64) << 40 | fin["end"]
            order = np.argsort(key)
            key, kk = key[order], fin["k"][order]
            want = [(i, a, d) for (i, a, d) in planted if d <= args.k]
            found_planted = 0
            if args.stream_style != "uniform" or args.plant_run:
                tol = 1 << 38                                         # repeats: a chain of candidates has ONE hit, anywhere along the chain (filter_bitvec.cc:103-135)
            for (i, a, d) in want:
                e = a + args.length
                lo_i = np.searchsorted(key, ((i + 1) << 40) | max(0, e - tol))
                hi_i = np.searchsorted(key, ((i + 1) << 40) | min(e + tol, (1 << 40) - 1), side="right")
                if hi_i > lo_i and kk[lo_i:hi_i].min() <= d:
                    found_planted += 1
            planted = want
            if found_planted != len(planted):
                raise SystemExit("bench.py: only %d of %d planted primer sites were reported" % (found_planted, len(planted)))
    # per-rank scan kernel time and the exchange's device time (outside the timed region)
    kms_mine = float(np.mean(kernel_ms[args.warmup:])) if len(kernel_ms) > args.warmup else float("nan")
    kms_all, exch_dev_ms = [kms_mine], None
    if use_dist:
        torch.cuda.synchronize()
        box = [None] * world
        dist.all_gather_object(box, kms_mine)
        kms_all = [float(x) for x in box]
        evs = exch_events[args.warmup:]
        if evs:
            exch_dev_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if scan_stats.get("blocks"):
        # per block of 1024 positions and wave: rounds of the pair kernel's second pass, key hits per lane; key-hit rate per window test
        scan_stats["rounds_per_block"] = scan_stats["rounds"] / scan_stats["blocks"]
        scan_stats["key_hit_rate"] = scan_stats["key_hits"] / (scan_stats["blocks"] * 1024.0)
    else:
        for kx in ("blocks", "rounds", "key_hits"):
            scan_stats.pop(kx, None)
    if rank == 0:
        traffic, traffic_source, traffic_entry = measured_traffic(args, shard)
        ms_per_step = dt / args.steps * 1e3
        value = n_bases_total / (dt / args.steps) / 1e9
        kms = float(np.mean(kernel_ms[args.warmup:])) if len(kernel_ms) > args.warmup else float("nan")
        shard_bytes = end - begin
        alg_bytes = shard_bytes + 16 * cand_count[0] / max(world, 1)
        achieved = alg_bytes / (kms * 1e-3) / 1e9
        res = {
            "metric": baseline_metric(),
            "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "ranks_seen": dist.get_world_size() if use_dist else 1, "pack_ms": pm.pack_time(),
            # the one-off 2-bit re-encoding of the stream (untimed, at init) charged to a single cold pass
            "value_including_pack": n_bases_total / (dt / args.steps + pm.pack_time() * 1e-3) / 1e9,
            "pm_scan_ms": pm_scan_ms, "value_through_pm_scan": None if not pm_scan_ms else n_bases_total / (pm_scan_ms * 1e-3) / 1e9,
            "pm_scan": None if not pm_scan_ms else {"hits": pm_scan_hits, "chunk_bytes": args.scan_chunk, "records_per_call": args.scan_cap, "passes": args.scan_passes,
                        "what": "pm_reset + consecutive pm_scan ranges over the whole stream, every range drained into the caller's pageable array, "
                                "hits sorted by (end, pid); equal to the timed step's hits (checked)"},
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%d x %d-mer primers, both strands (%d patterns), %s %d, %s Gbp stream in total = %s Gbp per GPU x %d GPU(s), %d entries"
                                   % (args.primers, args.length, len(allp), "-k" if args.indels else "-K", args.k,
                                      ("%.3g" % (total / 1e9)), ("%.3g" % (shard / 1e9)), world, args.entries * (world if args.scaling == "weak" else 1)),
                       "db_bases_total": total, "db_bases_per_gpu": shard,
                       "semantics": pm.selected()[0], "kernel_family": pm.selected()[1], "kernel": desc,
                       "final_hits": final_hits[0], "candidates": cand_count[0],
                       "planted_found": None if found_planted is None else "%d of %d" % (found_planted, len(planted)),
                       "rescans_after_overflow": rescans[0], "steps_with_a_cut_chain": cut_steps[0],
                       "stream_style": args.stream_style, "primer_source": args.primer_source,
                       "scan_stats": scan_stats,
                       "stream": "1 B/base resident in HBM before the timed region; the handle's init (untimed, with the pattern tables) "
                                 "also derives its 2-bit form, which the seed kernels' first stage reads"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": kms, "algorithmic_bytes": alg_bytes},
        }
        if traffic_entry is not None:
            st = staleness(traffic_entry, kms)
            res["roofline"].update({"traffic_head": st["head"], "traffic_code_sha": st["code_sha_at_profile"], "code_sha_now": st["code_sha_now"],
                                    "traffic_kernel_ms_at_profile": st["kernel_ms_at_profile"], "traffic_stale": st["stale"]})
        if use_dist:
            res["config"]["exchange"] = {"backend": backend, "forced_at_world_1": bool(force_dist),
                                         "form": "owned finalize on every rank, final hits gathered" if own_path else "candidate records gathered, rank 0 finalizes"}
            res["exchange_ms"] = float(np.mean(exch_host_ms[args.warmup:])) if len(exch_host_ms) > args.warmup else None
            res["exchange_device_ms"] = exch_dev_ms
            res["kernel_ms_per_rank"] = kms_all
        if pm.selected()[1] == sat_amd.KERNEL_SEED:
            ir = issue_roofline(args, shard, kms)
            if ir:
                res["issue_roofline"] = ir
        # the streaming-read rate this box sustains (same 16-byte loads, nothing else to do)
        try:
            mb = min(stream.numel(), 1 << 31) // 16 * 16
            if mb >= (1 << 29):                                      # well past the 256 MiB Infinity Cache
                res["roofline"]["measured_read_peak"] = sat_amd.measure_stream_read(stream.data_ptr(), mb, reps=5)
                res["roofline"]["frac_of_measured"] = achieved / res["roofline"]["measured_read_peak"]
        except Exception as e:                                       # measurement only: never fails the bench
            log("measure_stream_read: %s" % e)
        if pm.selected()[1] == sat_amd.KERNEL_BITPAR:
            # the bit-parallel family is integer-ALU bound.  One 32-bit state word per lane,
            # tile and row is updated per stream byte with ~c vector ops (4 exact; 6 per row with edits,
            # 4 with mismatches only); peak = 256 CUs x 128 lanes x ~2.4 GHz 32-bit integer ops.
            import re
            m = re.search(r"tiles=(\d+) lanes_per_tile=64 words_per_lane=(\d+)", desc)
            if m:
                words = int(m.group(1)) * 64 * int(m.group(2))
                c = 4.0 if args.k == 0 else (6.0 if args.indels else 4.0) * (args.k + 1)
                ops = shard_bytes * words * c
                peak = 256 * 4 * 32 * 2.4e9 / 1e12                      # 4 SIMD-32 per CU (MI355X_MICROARCH.md: wave64 v_fma 2 cyc)
                res["alu_roofline"] = {"achieved": ops / (kms * 1e-3) / 1e12, "peak": peak, "unit": "T u32-op/s",
                                       "frac": ops / (kms * 1e-3) / 1e12 / peak,
                                       "model": "%d state words x %.0f ops per stream byte" % (words, c)}
        if not args.no_cpu and args.cpu_sample >= 0 and world == 1:
            cb = cpu_baseline(args, stream, primers, log)
            if cb:
                res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    pm.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
