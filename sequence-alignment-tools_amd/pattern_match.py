"""ctypes binding of include/pm_gpu.h and a PatternMatch-shaped host class."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = None

HIT_DTYPE = np.dtype([("end", "<i8"), ("pid", "<u4"), ("k", "u1"), ("aux", "u1", (3,))])

SEM_AUTO, SEM_KEYWORD_TREE, SEM_SHIFT_AND, SEM_FILTER_BITVEC = 0, 2, 4, 5
SEM_EXACT_BASES, SEM_EXACT_HALVES, SEM_SHIFT_AND_INEXACT = 8, 12, 100
KERNEL_AUTO, KERNEL_BITPAR, KERNEL_SEED = 0, 16, 17
PM_E_OVERFLOW = -5

# every entry point include/pm_gpu.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "pm_create", "pm_add_pattern", "pm_init", "pm_init_device", "pm_scan", "pm_scan_view", "pm_scan_candidates",
    "pm_scan_candidates_async", "pm_scan_wait", "pm_candidates_device", "pm_set_capacity", "pm_finalize",
    "pm_finalize_device", "pm_finalize_device_owned", "pm_align_hits", "pm_align_hits_text",
    "pm_reset", "pm_destroy", "pm_last_error", "pm_selected_semantics", "pm_selected_kernel", "pm_describe",
    "pm_last_kernel_time", "pm_pick_semantics", "pm_measure_stream_read",
    "pm_final_hits_device", "pm_copy_records", "pm_pack_time", "pm_init_host", "pm_scan_stats", "pm_measure_pair_edit_floor", "pm_prepare_device",
    "pm_comm_unique_id", "pm_comm_create", "pm_comm_gather", "pm_comm_destroy", "pm_comm_last_error",
]


class PmError(RuntimeError):
    def __init__(self, code, msg, required=0):
        super().__init__("pm_gpu error %d: %s" % (code, msg))
        self.code = code
        self.required = required          # PM_E_OVERFLOW: the record count the buffer must hold


class _Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("semantics", C.c_int32), ("kernel", C.c_int32), ("k", C.c_int32),
                ("indels", C.c_int32), ("wildcards", C.c_int32), ("text_n", C.c_int32), ("eos", C.c_int32),
                ("device", C.c_int32), ("reserved", C.c_int32 * 7)]


def library_path():
    """csrc/libpm_gpu.so.  A/B measurement builds of the same sources (csrc/Makefile VARIANT=...) are loaded only when the
    caller says so twice: PM_GPU_LIB names a libpm_gpu*.so inside csrc/ AND PM_GPU_LIB_AB=1 is set (scripts/sweep_pair2.sh);
    a stray environment variable cannot point the product at another file."""
    alt = os.environ.get("PM_GPU_LIB")
    if alt and os.environ.get("PM_GPU_LIB_AB") == "1":
        alt = os.path.abspath(alt)
        if os.path.dirname(alt) == _CSRC and os.path.basename(alt).startswith("libpm_gpu") and alt.endswith(".so"):
            return alt
        raise PmError(-1, "PM_GPU_LIB must name a libpm_gpu*.so inside %s" % _CSRC)
    return os.path.join(_CSRC, "libpm_gpu.so")


def build_library(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", _CSRC])
    return library_path()


def load_library():
    """Load csrc/libpm_gpu.so.  No fallback: a missing library is an error.

    Processes that also use PyTorch on the GPU should import torch first: torch ships its own HIP
    runtime, and whichever runtime is loaded first serves the process."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise PmError(-4, "HIP extension %s is not built (run __graft_entry__.build())" % path)
        L = C.CDLL(path)
        L.pm_last_error.restype = C.c_char_p
        L.pm_last_error.argtypes = [C.c_void_p]
        L.pm_destroy.restype = None
        L.pm_destroy.argtypes = [C.c_void_p]
        for name in ABI_SYMBOLS:
            f = getattr(L, name)
            if name not in ("pm_last_error", "pm_destroy", "pm_comm_destroy", "pm_comm_last_error"):
                f.restype = C.c_int
        L.pm_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
        L.pm_add_pattern.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint64, C.c_int32, C.c_int32]
        L.pm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.pm_init_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.pm_init_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        L.pm_scan.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.pm_scan_view.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.pm_scan_candidates.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.pm_scan_candidates_async.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.pm_scan_wait.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.pm_candidates_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.pm_set_capacity.argtypes = [C.c_void_p, C.c_size_t]
        L.pm_finalize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.pm_finalize_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.pm_finalize_device_owned.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                               C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.pm_align_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.pm_align_hits_text.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.pm_reset.argtypes = [C.c_void_p]
        L.pm_selected_semantics.argtypes = [C.c_void_p]
        L.pm_selected_kernel.argtypes = [C.c_void_p]
        L.pm_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.pm_last_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.pm_pick_semantics.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pm_measure_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
        L.pm_final_hits_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.pm_copy_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.pm_scan_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pm_measure_pair_edit_floor.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        L.pm_pack_time.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _LIB = L
    return _LIB


def pick_semantics(alphabet_size, acgt_normalized, k, patterns, esb=None, eeb=None, wildcards=False):
    """pick_pattern_index's automatic engine choice (reference select.cc:101-141)."""
    L = load_library()
    pl = np.array([len(p) for p in patterns], dtype=np.int32)
    a = lambda x: None if x is None else np.ascontiguousarray(x, dtype=np.int32).ctypes.data_as(C.c_void_p)
    return L.pm_pick_semantics(alphabet_size, int(acgt_normalized), k, int(wildcards), len(patterns),
                               pl.ctypes.data_as(C.c_void_p), a(esb), a(eeb))


def measure_stream_read(d_ptr, nbytes, reps=5, stream=0):
    """Streaming-read rate (GB/s) of this box over an HBM buffer (pm_measure_stream_read)."""
    g = C.c_float()
    rc = load_library().pm_measure_stream_read(C.c_void_p(d_ptr), C.c_size_t(nbytes), int(reps), C.c_void_p(stream), C.byref(g))
    if rc:
        raise PmError(rc, "pm_measure_stream_read failed")
    return g.value


def reverse_comp(p):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "a": "t", "c": "g", "g": "c", "t": "a", "n": "n"}
    return "".join(comp.get(c, c) for c in reversed(p))


def sorted_tuples(hits):
    return sorted(zip(hits["end"].tolist(), hits["pid"].tolist(), hits["k"].tolist()))


class PatternMatch:
    """Host-side mirror of the reference's PatternMatch (pattern_match.h:84-156) over the GPU engine.

    k / indels / eos follow pick_pattern_index's arguments (select.cc:19-30): indels=True is the
    CLI's -k (edits), False is -K (substitutions only).  `semantics` forces a reference engine
    (-N), SEM_AUTO reproduces the automatic choice; `kernel` picks the GPU kernel family.
    """

    def __init__(self, k=0, indels=True, eos="\n", semantics=SEM_AUTO, kernel=KERNEL_AUTO, device=0,
                 wildcards=False, text_n=False):
        self._L = load_library()
        cfg = _Config()
        cfg.abi_version, cfg.semantics, cfg.kernel, cfg.k = 1, semantics, kernel, k
        cfg.indels, cfg.wildcards, cfg.text_n = int(bool(indels)), int(bool(wildcards)), int(bool(text_n))
        cfg.eos = ord(eos) if isinstance(eos, str) else int(eos)
        cfg.device = device
        self._h = C.c_void_p()
        rc = self._L.pm_create(C.byref(cfg), C.byref(self._h))
        if rc:
            raise PmError(rc, (self._L.pm_last_error(None) or b"").decode())
        self._n = 0
        self._pos = 0
        self._keep = None

    def _check(self, rc):
        if rc:
            raise PmError(rc, (self._L.pm_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.pm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- PatternMatch::add_pattern (pattern_match.h:116) --------------------------------------
    def add_pattern(self, pat, id=0, exact_start_bases=0, exact_end_bases=0):
        b = pat.encode() if isinstance(pat, str) else bytes(pat)
        self._check(self._L.pm_add_pattern(self._h, b, len(b), id, exact_start_bases, exact_end_bases))
        return id

    # -- PatternMatch::init (pattern_match.h:130) -----------------------------------------------
    def init(self, text, table=None):
        """`text`: uint8 numpy array (host stream bytes as getnch() returns them).
        `table`: cp.ch(0..size-1) for a normalized stream, None for a raw one."""
        arr = np.ascontiguousarray(text, dtype=np.uint8)
        self._keep = arr                      # pm_init borrows the host pointer
        tb = None if table is None else (C.c_uint8 * len(table)).from_buffer_copy(bytes(table))
        self._check(self._L.pm_init(self._h, arr.ctypes.data_as(C.c_void_p), arr.size, tb, 0 if table is None else len(table)))
        self._n, self._pos = arr.size, 0

    def init_host(self, text, table=None):
        """pm_init_host: a handle that runs the HOST stage only (finalize, align_hits) over the whole stream -- the merge
        rank of a position-sharded scan."""
        arr = np.ascontiguousarray(text, dtype=np.uint8)
        self._keep = arr
        tb = None if table is None else (C.c_uint8 * len(table)).from_buffer_copy(bytes(table))
        self._check(self._L.pm_init_host(self._h, arr.ctypes.data_as(C.c_void_p), arr.size, tb, 0 if table is None else len(table)))
        self._n, self._pos = arr.size, 0

    def init_device(self, data_ptr, n, table=None, stream=None, keepalive=None):
        """Stream already resident in HBM (e.g. a torch uint8 tensor's data_ptr())."""
        self._keep = keepalive
        tb = None if table is None else (C.c_uint8 * len(table)).from_buffer_copy(bytes(table))
        self._check(self._L.pm_init_device(self._h, C.c_void_p(data_ptr), n, tb, 0 if table is None else len(table),
                                           C.c_void_p(stream or 0)))
        self._n, self._pos = n, 0

    # -- PatternMatch::find_patterns (pattern_match.h:131) -------------------------------------
    def find_patterns(self, hits, minka=1, chunk=1 << 26):
        """Scan on from the current stream position until >= minka hits were produced or the end
        of the stream; appends (end, pid, k) tuples to `hits`; returns True if any were appended
        (the reference's `more`).  pos() afterwards is the scanned-to position."""
        got = 0
        buf = np.zeros(1 << 16, dtype=HIT_DTYPE)
        n_out, more = C.c_size_t(), C.c_int()
        while True:
            begin = self._pos
            end = min(self._n, begin + chunk)
            self._check(self._L.pm_scan(self._h, begin, end, buf.ctypes.data_as(C.c_void_p), buf.size,
                                        C.byref(n_out), C.byref(more)))
            self._pos = end
            while True:
                for i in range(n_out.value):
                    hits.append((int(buf["end"][i]), int(buf["pid"][i]), int(buf["k"][i])))
                got += n_out.value
                if not more.value:
                    break
                self._check(self._L.pm_scan(self._h, end, end, buf.ctypes.data_as(C.c_void_p), buf.size,
                                            C.byref(n_out), C.byref(more)))
            if got >= minka or self._pos >= self._n:
                return got > 0

    def find_all(self, chunk=1 << 26):
        """All hits over the whole stream as a structured array sorted by (end, pid)."""
        self.reset()
        hits = []
        while self.find_patterns(hits, minka=1 << 62, chunk=chunk):
            if self._pos >= self._n:
                break
        out = np.zeros(len(hits), dtype=HIT_DTYPE)
        if hits:
            a = np.array(hits, dtype=np.int64)
            out["end"], out["pid"], out["k"] = a[:, 0], a[:, 1], a[:, 2]
        return out

    def scan(self, begin, end, out):
        """pm_scan itself (PatternMatch::find_patterns over [begin,end), pattern_match.h:131): final hits, sorted by
        (end, pid), into the caller's array `out`; returns (count, more).  more: call again with begin == end."""
        n_out, more = C.c_size_t(), C.c_int()
        self._check(self._L.pm_scan(self._h, begin, end, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n_out), C.byref(more)))
        return n_out.value, more.value

    def scan_view(self, begin, end):
        """pm_scan_view: the range's final hits as a numpy view of the handle's own buffer (valid until the next call)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.pm_scan_view(self._h, begin, end, C.byref(p), C.byref(n)))
        if not n.value:
            return np.zeros(0, dtype=HIT_DTYPE)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value * 16,)).view(HIT_DTYPE)

    def pos(self):
        return self._pos

    # -- PatternMatch::reset (pattern_match.h:134) ----------------------------------------------
    def reset(self):
        self._check(self._L.pm_reset(self._h))
        self._pos = 0

    # -- device / host stages separately (multi-GPU path, bench) -------------------------------
    def set_capacity(self, n):
        self._check(self._L.pm_set_capacity(self._h, n))

    def scan_candidates(self, begin, end, to_host=True):
        """Device stage over (begin, end]; the record buffer grows (and the range is scanned again)
        when it was too small.  Returns the records (host array) or, with to_host=False, their count."""
        n_out = C.c_size_t()
        rc = self._L.pm_scan_candidates(self._h, begin, end, None, 0, C.byref(n_out))
        while rc == PM_E_OVERFLOW:
            self.set_capacity(int(n_out.value * 1.25) + 1024)
            rc = self._L.pm_scan_candidates(self._h, begin, end, None, 0, C.byref(n_out))
        self._check(rc)
        if not to_host:
            return n_out.value
        ptr, n = self.candidates_device()
        return self.copy_records(ptr, n)

    def copy_records(self, d_ptr, n):
        """n 16-byte records from HBM (pm_copy_records)."""
        out = np.zeros(n, dtype=HIT_DTYPE)
        if n:
            self._check(self._L.pm_copy_records(self._h, C.c_void_p(d_ptr), n, out.ctypes.data_as(C.c_void_p)))
        return out

    def scan_async(self, begin, end):
        self._check(self._L.pm_scan_candidates_async(self._h, begin, end))

    def scan_wait(self):
        """Raises PmError(PM_E_OVERFLOW) with .required set when the record buffer was too small:
        set_capacity(required * 1.25) and scan the range again."""
        n_out = C.c_size_t()
        rc = self._L.pm_scan_wait(self._h, C.byref(n_out))
        if rc == PM_E_OVERFLOW:
            raise PmError(rc, (self._L.pm_last_error(self._h) or b"").decode(), required=n_out.value)
        self._check(rc)
        return n_out.value

    def candidates_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.pm_candidates_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def finalize(self, cands, scanned_to, last=True, sort=True):
        cands = np.ascontiguousarray(cands, dtype=HIT_DTYPE)
        out = np.empty(max(cands.size, 1), dtype=HIT_DTYPE)
        n_out = C.c_size_t()
        self._check(self._L.pm_finalize(self._h, cands.ctypes.data_as(C.c_void_p), cands.size, scanned_to,
                                        (1 if last else 0) | (2 if sort else 0),
                                        out.ctypes.data_as(C.c_void_p), out.size, C.byref(n_out)))
        return out[:n_out.value]

    def finalize_device(self, scanned_to, last=True, sort=True, d_cands=None, n=0, out=None, owned=None, keep=False):
        """GPU clustering of the records of the last scan (or of `d_cands`, a device pointer);
        returns the final hits (host array).  Raises PmError(-2) where only the host stage applies.
        owned=(own_lo, own_hi, guard_lo, guard_hi): this call is one shard of a position-sharded
        scan (pm_finalize_device_owned); guard_hi=None declares the true end of the stream.
        keep=True: the hits stay in HBM; returns (device pointer, count) (pm_final_hits_device)."""
        if keep:
            optr, ocap, sort = C.c_void_p(0), 0, False
        else:
            if out is None:
                cap = max(n if d_cands else self.candidates_device()[1], 1) + 1024
                out = np.empty(cap, dtype=HIT_DTYPE)
            optr, ocap = out.ctypes.data_as(C.c_void_p), out.size
        n_out = C.c_size_t()
        if owned is not None:
            own_lo, own_hi, guard_lo, guard_hi = owned
            self._check(self._L.pm_finalize_device_owned(self._h, C.c_void_p(d_cands or 0), n, own_lo, own_hi, guard_lo,
                                                         (1 << 63) - 1 if guard_hi is None else guard_hi, 2 if sort else 0,
                                                         optr, ocap, C.byref(n_out)))
        else:
            self._check(self._L.pm_finalize_device(self._h, C.c_void_p(d_cands or 0), n, scanned_to,
                                                   (1 if last else 0) | (2 if sort else 0), optr, ocap, C.byref(n_out)))
        if keep:
            p, cnt = C.c_void_p(), C.c_size_t()
            self._check(self._L.pm_final_hits_device(self._h, C.byref(p), C.byref(cnt)))
            return p.value or 0, cnt.value
        return out[:n_out.value]

    def align_hits(self, hits):
        """primer_match's per-hit re-alignment; returns a structured array (start, end, editdist, value)."""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        out = np.zeros(hits.size, dtype=np.dtype([("start", "<i8"), ("end", "<i8"), ("editdist", "<i4"), ("value", "<i4")]))
        if hits.size:
            self._check(self._L.pm_align_hits(self._h, hits.ctypes.data_as(C.c_void_p), hits.size, out.ctypes.data_as(C.c_void_p)))
        return out

    def selected(self):
        return self._L.pm_selected_semantics(self._h), self._L.pm_selected_kernel(self._h)

    def describe(self):
        buf = C.create_string_buffer(512)
        self._check(self._L.pm_describe(self._h, buf, 512))
        return buf.value.decode()

    def scan_stats(self):
        """pm_scan_stats: counters of the last scan (see include/pm_gpu.h)"""
        v = (C.c_uint64 * 8)()
        self._check(self._L.pm_scan_stats(self._h, v, 8))
        names = ("candidates", "between_stages", "internal_rescans", "blocks", "rounds", "key_hits", "range_splits")
        return {k: int(v[i]) for i, k in enumerate(names)}

    def measure_pair_edit_floor(self, mode):
        """pm_measure_pair_edit_floor: (kernel ms, suspect records)"""
        ms, n = C.c_float(), C.c_uint64()
        self._check(self._L.pm_measure_pair_edit_floor(self._h, mode, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def pack_time(self):
        ms = C.c_float()
        self._check(self._L.pm_pack_time(self._h, C.byref(ms)))
        return ms.value

    def last_kernel_time(self):
        ms, n = C.c_float(), C.c_int()
        self._check(self._L.pm_last_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
