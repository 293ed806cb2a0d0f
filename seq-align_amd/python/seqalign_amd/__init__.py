"""ctypes driver for seq-align_amd/lib/libseqalign_hip.so (tests + bench only).

The product is the C-ABI shared library (include/seqalign_hip.h and the mirrored
reference headers next to it).  Python is not on the hot path: this module only
loads the .so, mirrors the C structs and moves pointers around.  torch is used
by DeviceBatch for device memory and streams (plumbing).

There is no fallback of any kind: if the library is missing, `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import workloads  # noqa: F401

PKG_ROOT = Path(__file__).resolve().parents[2]          # seq-align_amd/
REPO_ROOT = PKG_ROOT.parent
import os as _os
# SEQALIGN_LIB: tuning experiments point this at an alternative build of the same library
LIB_PATH = Path(_os.environ.get("SEQALIGN_LIB") or (PKG_ROOT / "lib" / "libseqalign_hip.so"))

OK, E_NO_DEVICE, E_HIP, E_ARG, E_NOMEM, E_UNKNOWN_PAIR, E_DOMAIN, E_TRACEBACK, E_TOO_LARGE = range(9)
KERNEL_AUTO, KERNEL_WAVEFRONT, KERNEL_ROWSCAN, KERNEL_STREAM, KERNEL_STRIPS, KERNEL_WGSTREAM = 0, 1, 2, 3, 4, 5
KERNEL_NAMES = {KERNEL_AUTO: "auto", KERNEL_WAVEFRONT: "wavefront", KERNEL_ROWSCAN: "rowscan",
                KERNEL_STREAM: "stream", KERNEL_STRIPS: "strips", KERNEL_WGSTREAM: "wgstream"}
STATUS_OK = 0xFFFFFFFFFFFFFFFF


class Scoring(C.Structure):
    """scoring_t -- include/alignment_scoring.h (reference alignment_scoring.h:19-40)."""
    _fields_ = [
        ("gap_open", C.c_int), ("gap_extend", C.c_int),
        ("no_start_gap_penalty", C.c_bool), ("no_end_gap_penalty", C.c_bool),
        ("no_gaps_in_a", C.c_bool), ("no_gaps_in_b", C.c_bool),
        ("no_mismatches", C.c_bool), ("use_match_mismatch", C.c_bool),
        ("match", C.c_int), ("mismatch", C.c_int),
        ("case_sensitive", C.c_bool),
        ("wildcards", C.c_uint32 * 8), ("swap_set", (C.c_uint32 * 8) * 256),
        ("wildscores", C.c_int * 256), ("swap_scores", (C.c_int * 256) * 256),
        ("min_penalty", C.c_int), ("max_penalty", C.c_int),
    ]


class BatchDesc(C.Structure):
    """seqalign_batch_t"""
    _fields_ = [("n_pairs", C.c_uint64), ("arena", C.c_void_p), ("arena_bytes", C.c_uint64),
                ("off_a", C.c_void_p), ("len_a", C.c_void_p), ("off_b", C.c_void_p), ("len_b", C.c_void_p)]


class DevBatchDesc(C.Structure):
    """seqalign_dev_batch_t"""
    _fields_ = [("n_pairs", C.c_uint64), ("arena", C.c_void_p), ("off_a", C.c_void_p),
                ("len_a", C.c_void_p), ("off_b", C.c_void_p), ("len_b", C.c_void_p),
                ("mat_off", C.c_void_p), ("match_scores", C.c_void_p), ("gap_a_scores", C.c_void_p),
                ("gap_b_scores", C.c_void_p), ("status", C.c_void_p),
                ("max_len_a", C.c_uint32), ("max_len_b", C.c_uint32)]


class SwReduceDesc(C.Structure):
    """seqalign_sw_reduce_t"""
    _fields_ = [("n_pairs", C.c_uint64), ("len_a", C.c_void_p), ("len_b", C.c_void_p),
                ("mat_off", C.c_void_p), ("match_scores", C.c_void_p), ("min_score", C.c_int32),
                ("best_score", C.c_void_p), ("best_index", C.c_void_p), ("cand_count", C.c_void_p),
                ("cand_off", C.c_void_p), ("cand_cap", C.c_void_p), ("cand_index", C.c_void_p),
                ("cand_score", C.c_void_p)]


class SwHit(C.Structure):
    """seqalign_sw_hit_t"""
    _fields_ = [("pair", C.c_uint64), ("score", C.c_int32), ("pos_a", C.c_uint32), ("pos_b", C.c_uint32),
                ("len_a", C.c_uint32), ("len_b", C.c_uint32), ("length", C.c_uint32), ("str_off", C.c_uint64)]


class SeqAlignError(RuntimeError):
    def __init__(self, code: int, where: str):
        l = lib()
        super().__init__(f"{where}: {l.seqalign_strerror(code).decode()} [{code}] {l.seqalign_last_error().decode()}")
        self.code = code


_lib = None


def lib():
    """The product library.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FileNotFoundError(
                f"{LIB_PATH} missing: build it with `make -C {PKG_ROOT}` "
                "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU path")
        # One HIP runtime per process: torch wheels bundle their own libamdhip64, and a
        # process that loads /opt/rocm's copy first (through our DT_NEEDED) and torch's
        # second ends up with two runtimes, the second of which sees no device.  When
        # torch is installed, let it load first; our library then binds to the same
        # runtime by SONAME.  (The C tools and non-Python callers use /opt/rocm's.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(str(LIB_PATH))
        l.seqalign_strerror.restype = C.c_char_p
        l.seqalign_last_error.restype = C.c_char_p
        l.seqalign_kernel_kind_name.restype = C.c_char_p
        l.needleman_wunsch_new.restype = C.c_void_p
        l.smith_waterman_new.restype = C.c_void_p
        l.alignment_create.restype = C.c_void_p
        l.smith_waterman_get_aligner.restype = C.c_void_p
        _lib = l
    return _lib


def _check(code: int, where: str):
    if code != OK:
        raise SeqAlignError(code, where)


def _ptr(arr) -> C.c_void_p:
    return C.c_void_p(arr.ctypes.data) if arr is not None else C.c_void_p(0)


def make_scoring(spec: dict) -> Scoring:
    """Build a scoring_t through OUR host library (scoring_init & friends).
    Same JSON-able spec as tests/orclib.build_scoring."""
    l = lib()
    sc = Scoring()
    C.memset(C.byref(sc), 0, C.sizeof(sc))
    if "preset" in spec:
        getattr(l, "scoring_system_" + spec["preset"])(C.byref(sc))
    else:
        l.scoring_init(C.byref(sc), *[C.c_int(int(v)) for v in spec["init"][:4]],
                       *[C.c_bool(bool(v)) for v in spec["init"][4:]])
    for ch, s in spec.get("wildcards", []):
        l.scoring_add_wildcard(C.byref(sc), C.c_char(ch.encode()), C.c_int(s))
    for a, b, s in spec.get("mutations", []):
        l.scoring_add_mutation(C.byref(sc), C.c_char(a.encode()), C.c_char(b.encode()), C.c_int(s))
    if "use_match_mismatch" in spec:
        sc.use_match_mismatch = bool(spec["use_match_mismatch"])
    for k, v in spec.get("flags", {}).items():
        setattr(sc, k, bool(v))
    return sc


def batch_desc(batch: "workloads.Batch") -> BatchDesc:
    return BatchDesc(batch.n_pairs, batch.arena.ctypes.data, batch.arena.nbytes,
                     batch.off_a.ctypes.data, batch.len_a.ctypes.data,
                     batch.off_b.ctypes.data, batch.len_b.ctypes.data)


OPTION_DEFAULTS = {"kernel": "auto", "cpl": 0, "wpb": 0, "lds_pad": 0, "traceback": "device", "trace_kernel": "auto",
                   "sweep_mode": "auto", "sweep_strip": 0, "sweep_cpl": 0, "sweep_trace": 0, "sweep_dirs": 1, "nw_dirs": 1, "pack16": 1, "quad": 0, "walk_overlap": 0, "timing": 0, "chunk_bytes": 0,
                   "subbatches": 0, "arena_scan_gib": 160, "arena_quality": 1.045, "arena_keep_gib": 16, "upload_slices": 0, "arena_free_pct": 60, "nw_moves": 1, "zero_copy": "auto", "sweep_ev": 1, "reduce_depth": 0, "async_lanes": 0, "walk_group": 0, "dirs_local": 1, "walk_stage": 1, "walk_tile": 0}


K_MAX = 32


class CallInfo(C.Structure):
    """seqalign_call_info_t (include/seqalign_hip.h)."""
    _fields_ = [("launches", C.c_uint32 * K_MAX), ("items", C.c_uint64 * K_MAX)]


class ArenaInfo(C.Structure):
    _fields_ = [("quality", C.c_float), ("target", C.c_float), ("vmm", C.c_int32), ("chunk_mib", C.c_uint32),
                ("depth_gib", C.c_float), ("scanned_gib", C.c_float), ("depth_a_gib", C.c_float), ("tries", C.c_uint32),
                ("second_walk_from", C.c_uint32), ("try_quality", C.c_float * 96), ("try_depth_gib", C.c_float * 96),
                ("kept_gib", C.c_float), ("seconds", C.c_float)]

    def as_dict(self):
        n = int(self.tries)
        return {"quality": round(float(self.quality), 3), "target": round(float(self.target), 3), "vmm": bool(self.vmm),
                "chunk_mib": int(self.chunk_mib), "depth_gib": round(float(self.depth_gib), 1),
                "scanned_gib": round(float(self.scanned_gib), 1), "depth_a_gib": round(float(self.depth_a_gib), 1),
                "tries": n, "second_walk_from": int(self.second_walk_from), "kept_gib": round(float(self.kept_gib), 1), "seconds": round(float(self.seconds), 3),
                "try_quality": [round(float(self.try_quality[i]), 3) for i in range(n)],
                "try_depth_gib": [round(float(self.try_depth_gib[i]), 1) for i in range(n)]}


class Context:
    """seqalign_ctx_t* for one device."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p(0)
        _check(lib().seqalign_ctx_create(C.c_int(device), C.byref(self._h)), "seqalign_ctx_create")
        self.device = device

    def close(self):
        if self._h:
            lib().seqalign_ctx_destroy(self._h)
            self._h = C.c_void_p(0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def pool_trim(self, keep_bytes=None) -> int:
        """seqalign_pool_trim: release the device's chunk pool down to keep_bytes (None: only ask); returns what it holds."""
        held = C.c_uint64(0)
        keep = C.c_uint64(0xFFFFFFFFFFFFFFFF if keep_bytes is None else int(keep_bytes))
        _check(lib().seqalign_pool_trim(self._h, keep, C.byref(held)), "seqalign_pool_trim")
        return int(held.value)

    # ---- options (seqalign_ctx_set_option: key = SEQALIGN_<KEY> in lower case) -----------------
    def set_option(self, key: str, value) -> None:
        if isinstance(value, (bool, np.bool_)):
            value = int(value)          # str(True) is "True": the library takes 1 / 0 (and true / false, on / off, yes / no)
        _check(lib().seqalign_ctx_set_option(self._h, key.encode(), str(value).encode()), f"seqalign_ctx_set_option({key}={value})")

    def get_option(self, key: str) -> str:
        """The value in force, as text (what set_option would take)."""
        buf = C.create_string_buffer(64)
        _check(lib().seqalign_ctx_get_option(self._h, key.encode(), buf, C.c_size_t(64)), f"seqalign_ctx_get_option({key})")
        return buf.value.decode()

    def options(self, **kv):
        """Context manager: set options on THIS context, put back on exit what was in force before -- whatever set it
        (SEQALIGN_* at context creation, an earlier set_option): `with ctx.options(traceback="host"): ...`."""
        ctx = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.before = {k: ctx.get_option(k) for k in kv}
                for k, v in kv.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self_inner, *exc):
                for k, v in self_inner.before.items():
                    ctx.set_option(k, v)
        return _Scope()

    def last_call(self) -> dict:
        """seqalign_ctx_last_call_info as {kind name: (launches, items)} for the kinds the last call launched."""
        info = CallInfo()
        _check(lib().seqalign_ctx_last_call_info(self._h, C.byref(info)), "seqalign_ctx_last_call_info")
        out = {}
        for k in range(K_MAX):
            if info.launches[k]:
                name = lib().seqalign_kernel_kind_name(C.c_int(k))
                out[name.decode() if name else f"kind{k}"] = (int(info.launches[k]), int(info.items[k]))
        return out

    # ---- scoring -----------------------------------------------------------
    def upload_scoring(self, scoring: Scoring, is_sw: int) -> C.c_void_p:
        h = C.c_void_p(0)
        _check(lib().seqalign_scoring_upload(self._h, C.byref(scoring), C.c_int(is_sw), C.byref(h)),
               "seqalign_scoring_upload")
        return h

    def release_scoring(self, h):
        lib().seqalign_scoring_release(self._h, h)

    # ---- host-level --------------------------------------------------------
    def _handles(self, peers):
        """(ctx array, count) for the *_multi entry points: this context plus `peers`."""
        ctxs = [self] + list(peers)
        return (C.c_void_p * len(ctxs))(*[c._h for c in ctxs]), C.c_int(len(ctxs))

    def fill_batch(self, batch, scoring: Scoring, is_sw: int, check: bool = True, out=None, peers=None):
        """H2D -> GPU fill -> D2H.  Returns (M, A, B, mat_off, status) numpy.
        out=(M, A, B) re-uses caller-owned (already touched) arrays."""
        cells = batch.matrix_cells()
        mat_off = np.zeros(batch.n_pairs, np.uint64)
        if batch.n_pairs:
            mat_off[1:] = np.cumsum(cells)[:-1]
        total = int(cells.sum())
        if out is not None:
            M, A, B = out
            assert M.size >= total and A.size >= total and B.size >= total
        else:
            M = np.empty(total, np.int32); A = np.empty(total, np.int32); B = np.empty(total, np.int32)
        status = np.zeros(batch.n_pairs, np.uint64)
        d = batch_desc(batch)
        if peers:
            hs, nh = self._handles(peers)
            rc = lib().seqalign_fill_batch_multi(hs, nh, C.byref(d), C.byref(scoring), C.c_int(is_sw), _ptr(mat_off),
                                                 _ptr(M), _ptr(A), _ptr(B), _ptr(status))
        else:
            rc = lib().seqalign_fill_batch(self._h, C.byref(d), C.byref(scoring), C.c_int(is_sw), _ptr(mat_off),
                                           _ptr(M), _ptr(A), _ptr(B), _ptr(status))
        if check:
            _check(rc, "seqalign_fill_batch")
        return (M, A, B, mat_off, status) if check else (rc, M, A, B, mat_off, status)

    def nw_batch(self, batch, scoring: Scoring, raw: bool = False, peers=None):
        """seqalign_nw_batch.  raw=True returns the C-side arrays (str_off, out_a,
        out_b, out_len, out_score) without building Python tuples per pair."""
        n = batch.n_pairs
        cache = getattr(self, "_nw_buffers", None)
        if raw and cache is not None and cache[0] is batch:
            str_off, out_a, out_b, out_len, out_score = cache[1]          # timing loops: same batch, same buffers
        else:
            caps = batch.len_a.astype(np.uint64) + batch.len_b.astype(np.uint64) + np.uint64(1)
            str_off = np.zeros(n, np.uint64)
            if n:
                str_off[1:] = np.cumsum(caps)[:-1]
            total = int(caps.sum()) + 1
            out_a, out_b = np.zeros(total, np.uint8), np.zeros(total, np.uint8)
            out_len, out_score = np.zeros(n, np.uint32), np.zeros(n, np.int32)
            if raw:
                self._nw_buffers = (batch, (str_off, out_a, out_b, out_len, out_score))
        d = batch_desc(batch)
        if peers:
            hs, nh = self._handles(peers)
            _check(lib().seqalign_nw_batch_multi(hs, nh, C.byref(d), C.byref(scoring), _ptr(str_off), _ptr(out_a),
                                                 _ptr(out_b), _ptr(out_len), _ptr(out_score)), "seqalign_nw_batch_multi")
        else:
            _check(lib().seqalign_nw_batch(self._h, C.byref(d), C.byref(scoring), _ptr(str_off), _ptr(out_a),
                                           _ptr(out_b), _ptr(out_len), _ptr(out_score)), "seqalign_nw_batch")
        if raw:
            return str_off, out_a, out_b, out_len, out_score
        res = []
        for p in range(n):
            o, ln = int(str_off[p]), int(out_len[p])
            res.append((int(out_score[p]), out_a[o:o + ln].tobytes(), out_b[o:o + ln].tobytes()))
        return res

    def sw_batch(self, batch, scoring: Scoring, min_score, max_hits: int = 1 << 20, hit_cap: int | None = None,
                 raw: bool = False, peers=None):
        """seqalign_sw_batch.  raw=True returns (n_hits, hits array, out_a, out_b)
        without building Python dicts per hit."""
        n = batch.n_pairs
        ms = np.full(n, min_score, np.int32) if np.isscalar(min_score) else np.asarray(min_score, np.int32)
        hit_cap = hit_cap or max(1024, 64 * n)
        str_cap = int(hit_cap * (int(batch.len_a.max(initial=0)) + int(batch.len_b.max(initial=0)) + 2))
        str_cap = min(str_cap, 1 << 30)
        cache = getattr(self, "_sw_buffers", None)
        if raw and cache is not None and cache[0] is batch and cache[1] == (hit_cap, str_cap):
            hits, out_a, out_b = cache[2]            # timing loops: same batch, same (already touched) buffers
        else:
            hits = (SwHit * hit_cap)()
            out_a, out_b = np.zeros(str_cap, np.uint8), np.zeros(str_cap, np.uint8)
            if raw:
                self._sw_buffers = (batch, (hit_cap, str_cap), (hits, out_a, out_b))
        n_hits = C.c_uint64(0)
        d = batch_desc(batch)
        if peers:
            hs, nh = self._handles(peers)
            _check(lib().seqalign_sw_batch_multi(hs, nh, C.byref(d), C.byref(scoring), _ptr(ms),
                                                 C.c_uint32(min(max_hits, 0xFFFFFFFF)), hits, C.c_uint64(hit_cap),
                                                 C.byref(n_hits), _ptr(out_a), _ptr(out_b), C.c_uint64(str_cap)),
                   "seqalign_sw_batch_multi")
        else:
            _check(lib().seqalign_sw_batch(self._h, C.byref(d), C.byref(scoring), _ptr(ms), C.c_uint32(min(max_hits, 0xFFFFFFFF)),
                                           hits, C.c_uint64(hit_cap), C.byref(n_hits), _ptr(out_a), _ptr(out_b),
                                           C.c_uint64(str_cap)), "seqalign_sw_batch")
        if raw:
            return n_hits.value, hits, out_a, out_b
        per_pair = [[] for _ in range(n)]
        for k in range(n_hits.value):
            h = hits[k]
            per_pair[h.pair].append(dict(score=h.score, pos_a=h.pos_a, pos_b=h.pos_b, len_a=h.len_a, len_b=h.len_b,
                                         a=out_a[h.str_off:h.str_off + h.length].tobytes().decode(),
                                         b=out_b[h.str_off:h.str_off + h.length].tobytes().decode()))
        return per_pair

    # ---- asynchronous host-level calls (seqalign_*_batch_submit / seqalign_job_wait) ---------------
    def nw_buffers(self, batch):
        """Output buffers of one seqalign_nw_batch call on `batch`: (str_off, out_a, out_b, out_len, out_score)."""
        n = batch.n_pairs
        caps = batch.len_a.astype(np.uint64) + batch.len_b.astype(np.uint64) + np.uint64(1)
        str_off = np.zeros(n, np.uint64)
        if n:
            str_off[1:] = np.cumsum(caps)[:-1]
        total = int(caps.sum()) + 1
        return str_off, np.zeros(total, np.uint8), np.zeros(total, np.uint8), np.zeros(n, np.uint32), np.zeros(n, np.int32)

    def nw_batch_submit(self, batch, scoring: Scoring, buffers=None) -> "Job":
        buffers = buffers or self.nw_buffers(batch)
        str_off, out_a, out_b, out_len, out_score = buffers
        job = Job(self, "nw", batch, scoring, buffers)
        job.desc = batch_desc(batch)
        _check(lib().seqalign_nw_batch_submit(self._h, C.byref(job.desc), C.byref(scoring), _ptr(str_off), _ptr(out_a), _ptr(out_b),
                                              _ptr(out_len), _ptr(out_score), C.byref(job.handle)), "seqalign_nw_batch_submit")
        return job

    def sw_batch_submit(self, batch, scoring: Scoring, min_score, max_hits: int = 1 << 20, hit_cap: int | None = None, buffers=None) -> "Job":
        n = batch.n_pairs
        ms = np.full(n, min_score, np.int32) if np.isscalar(min_score) else np.asarray(min_score, np.int32)
        hit_cap = hit_cap or max(1024, 64 * n)
        str_cap = min(int(hit_cap * (int(batch.len_a.max(initial=0)) + int(batch.len_b.max(initial=0)) + 2)), 1 << 30)
        buffers = buffers or ((SwHit * hit_cap)(), np.zeros(str_cap, np.uint8), np.zeros(str_cap, np.uint8))
        hits, out_a, out_b = buffers
        job = Job(self, "sw", batch, scoring, buffers)
        job.desc, job.ms, job.n_hits = batch_desc(batch), ms, C.c_uint64(0)
        _check(lib().seqalign_sw_batch_submit(self._h, C.byref(job.desc), C.byref(scoring), _ptr(ms), C.c_uint32(min(max_hits, 0xFFFFFFFF)),
                                              hits, C.c_uint64(hit_cap), C.byref(job.n_hits), _ptr(out_a), _ptr(out_b), C.c_uint64(len(out_a)),
                                              C.byref(job.handle)), "seqalign_sw_batch_submit")
        return job

    # ---- CIGAR as the batch calls' output (seqalign_*_batch_cigar) -------------------------------
    def nw_batch_cigar(self, batch, scoring: Scoring, fmt: int = 1, slot: int | None = None, raw: bool = False):
        """seqalign_nw_batch_cigar -> [(score, cigar bytes)] (raw: the C-side arrays).  slot: bytes per pair (default: the worst
        case 2 (len_a + len_b) + 2)."""
        n = batch.n_pairs
        cache = getattr(self, "_cg_buffers", None)
        if raw and cache is not None and cache[0] is batch and cache[1] == slot:
            off, out, out_len, out_score = cache[2]
        else:
            caps = (np.full(n, slot, np.uint64) if slot is not None else
                    2 * (batch.len_a.astype(np.uint64) + batch.len_b.astype(np.uint64)) + np.uint64(2))
            off = np.zeros(n + 1, np.uint64)
            off[1:] = np.cumsum(caps)
            out = np.zeros(int(off[n]) + 1, np.uint8)
            out_len, out_score = np.zeros(n, np.uint32), np.zeros(n, np.int32)
            if raw:
                self._cg_buffers = (batch, slot, (off, out, out_len, out_score))
        d = batch_desc(batch)
        _check(lib().seqalign_nw_batch_cigar(self._h, C.byref(d), C.byref(scoring), C.c_int(fmt), _ptr(off), _ptr(out), _ptr(out_len),
                                             _ptr(out_score)), "seqalign_nw_batch_cigar")
        if raw:
            return off, out, out_len, out_score
        return [(int(out_score[p]), out[int(off[p]):int(off[p]) + int(out_len[p])].tobytes()) for p in range(n)]

    def sw_batch_cigar(self, batch, scoring: Scoring, min_score, max_hits: int = 1 << 20, fmt: int = 1, hit_cap: int | None = None,
                       cigar_cap: int | None = None, raw: bool = False):
        """seqalign_sw_batch_cigar -> per pair a list of dict(score, pos_a, pos_b, len_a, len_b, length, cigar)."""
        n = batch.n_pairs
        ms = np.full(n, min_score, np.int32) if np.isscalar(min_score) else np.asarray(min_score, np.int32)
        hit_cap = hit_cap or max(1024, 64 * n)
        cigar_cap = cigar_cap or min(int(hit_cap * 2 * (int(batch.len_a.max(initial=0)) + int(batch.len_b.max(initial=0)) + 2)), 1 << 30)
        cache = getattr(self, "_swcg_buffers", None)
        if raw and cache is not None and cache[0] is batch and cache[1] == (hit_cap, cigar_cap):
            hits, out = cache[2]
        else:
            hits, out = (SwHit * hit_cap)(), np.zeros(cigar_cap, np.uint8)
            if raw:
                self._swcg_buffers = (batch, (hit_cap, cigar_cap), (hits, out))
        n_hits = C.c_uint64(0)
        d = batch_desc(batch)
        rc = lib().seqalign_sw_batch_cigar(self._h, C.byref(d), C.byref(scoring), _ptr(ms), C.c_uint32(min(max_hits, 0xFFFFFFFF)),
                                           C.c_int(fmt), hits, C.c_uint64(hit_cap), C.byref(n_hits), _ptr(out), C.c_uint64(cigar_cap))
        if raw:
            return rc, n_hits.value, hits, out
        _check(rc, "seqalign_sw_batch_cigar")
        per_pair = [[] for _ in range(n)]
        for k in range(n_hits.value):
            h = hits[k]
            end = int(h.str_off)
            while out[end]:
                end += 1
            per_pair[h.pair].append(dict(score=h.score, pos_a=h.pos_a, pos_b=h.pos_b, len_a=h.len_a, len_b=h.len_b, length=h.length,
                                         cigar=out[h.str_off:end].tobytes().decode()))
        return per_pair

    def dpp_probe(self, fill: int = -7):
        out = np.zeros(64, np.int32)
        _check(lib().sa_dpp_probe(self._h, C.c_int32(fill), _ptr(out)), "sa_dpp_probe")
        return out


class Job:
    """A submitted host-level call (seqalign_job_t*).  Keeps everything the C side borrows alive until wait()."""

    def __init__(self, ctx, kind, batch, scoring, buffers):
        self.ctx, self.kind, self.batch, self.scoring, self.buffers = ctx, kind, batch, scoring, buffers
        self.handle = C.c_void_p(0)

    def done(self) -> bool:
        return bool(lib().seqalign_job_done(self.handle)) if self.handle else True

    def wait(self, raw: bool = False):
        """seqalign_job_wait; returns what the synchronous wrapper would have (raw: the buffers as they are)."""
        h, self.handle = self.handle, C.c_void_p(0)
        _check(lib().seqalign_job_wait(h), f"seqalign_job_wait({self.kind})")
        if raw:
            return self.buffers if self.kind == "nw" else (self.n_hits.value, *self.buffers)
        if self.kind == "nw":
            str_off, out_a, out_b, out_len, out_score = self.buffers
            return [(int(out_score[p]), out_a[int(str_off[p]):int(str_off[p]) + int(out_len[p])].tobytes(),
                     out_b[int(str_off[p]):int(str_off[p]) + int(out_len[p])].tobytes()) for p in range(self.batch.n_pairs)]
        hits, out_a, out_b = self.buffers
        per_pair = [[] for _ in range(self.batch.n_pairs)]
        for k in range(self.n_hits.value):
            h = hits[k]
            per_pair[h.pair].append(dict(score=h.score, pos_a=h.pos_a, pos_b=h.pos_b, len_a=h.len_a, len_b=h.len_b,
                                         a=out_a[h.str_off:h.str_off + h.length].tobytes().decode(),
                                         b=out_b[h.str_off:h.str_off + h.length].tobytes().decode()))
        return per_pair


_ARENA_CTX = {}


def _arena_context(device: int) -> "Context":
    """One long-lived context per device for arena allocations of DeviceBatches
    created without one."""
    if device not in _ARENA_CTX:
        _ARENA_CTX[device] = Context(device)
    return _ARENA_CTX[device]


class _RawDeviceInts:
    """int32 device memory owned by the library, exposed to torch (zero copy)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


class DeviceBatch:
    """A batch resident in HBM (torch tensors own the memory) + its three output
    arenas.  `pad_cells` aligns each pair's first cell (32 cells = 128 B)."""

    def __init__(self, batch, device: int = 0, pad_cells: int = 32, placement: str = "spread", ctx=None):
        import torch
        self.torch = torch
        self.host = batch
        dev = torch.device("cuda", device)
        self.device = device
        cells = batch.matrix_cells()
        padded = (cells + pad_cells - 1) // pad_cells * pad_cells
        mat_off = np.zeros(batch.n_pairs, np.int64)
        if batch.n_pairs:
            mat_off[1:] = np.cumsum(padded)[:-1]
        self.mat_off_host = mat_off.astype(np.uint64)
        self.cells_host = cells
        self.total_cells = int(padded.sum())
        t = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else
                                       (a.view(np.int32) if a.dtype == np.uint32 else a)).to(dev)
        self.arena = t(batch.arena)
        self.off_a, self.len_a = t(batch.off_a), t(batch.len_a)
        self.off_b, self.len_b = t(batch.off_b), t(batch.len_b)
        self.mat_off = t(self.mat_off_host)
        stride = (self.total_cells + 1023) // 1024 * 1024
        self.placement_quality = -1.0
        self.placement_info = None
        if placement == "packed":
            # one allocation, three 4 KiB-aligned arenas back to back (the stream kernel
            # wants the arenas congruent mod 4 KiB; torch's allocator only promises 512 B)
            stride += int(_os.environ.get("SEQALIGN_ARENA_SKEW", "0")) // 256 * 256   # layout experiments
            self._arena3 = torch.empty(3 * stride + 1024, dtype=torch.int32, device=dev)
            skew = (-(self._arena3.data_ptr() // 4)) % 1024
            self.M = self._arena3[skew:skew + self.total_cells]
            self.A = self._arena3[skew + stride:skew + stride + self.total_cells]
            self.B = self._arena3[skew + 2 * stride:skew + 2 * stride + self.total_cells]
        else:
            # the library's allocator: arenas spread over HBM (seqalign_arenas_alloc)
            self._ctx = ctx if ctx is not None else _arena_context(device)
            ptrs = (C.c_void_p * 3)()
            q = C.c_float(-1.0)
            _check(lib().seqalign_arenas_alloc(self._ctx._h, C.c_uint64(4 * stride), ptrs, C.byref(q)),
                   "seqalign_arenas_alloc")
            self._arena_ptrs = ptrs
            self.placement_quality = float(q.value)
            info = ArenaInfo()
            _check(lib().seqalign_arenas_info(self._ctx._h, ptrs, C.byref(info)), "seqalign_arenas_info")
            self.placement_info = info.as_dict()
            self.M, self.A, self.B = (torch.as_tensor(_RawDeviceInts(ptrs[k], self.total_cells), device=dev)
                                      for k in range(3))
        self.status = torch.zeros(batch.n_pairs, dtype=torch.int64, device=dev)
        # launches go to a real (non-null) torch stream: a NULL stream handle means
        # "the context's own stream" in the C ABI, and torch events only see torch streams
        # (the context's own stream when there is one: a stream of our own would take one more of the runtime's hardware
        # queues, and the host-level calls measured beside it get slower -- DESIGN.md 3.5c)
        if ctx is not None and hasattr(lib(), "seqalign_ctx_stream"):
            lib().seqalign_ctx_stream.restype = C.c_void_p
            self.stream = torch.cuda.ExternalStream(int(lib().seqalign_ctx_stream(ctx._h)), device=dev)
        else:
            self.stream = torch.cuda.Stream(dev)
        self.desc = DevBatchDesc(batch.n_pairs, self.arena.data_ptr(), self.off_a.data_ptr(),
                                 self.len_a.data_ptr(), self.off_b.data_ptr(), self.len_b.data_ptr(),
                                 self.mat_off.data_ptr(), self.M.data_ptr(), self.A.data_ptr(),
                                 self.B.data_ptr(), self.status.data_ptr(),
                                 int(batch.len_a.max(initial=0)), int(batch.len_b.max(initial=0)))

    def __del__(self):
        ptrs = getattr(self, "_arena_ptrs", None)
        if ptrs is not None and getattr(self._ctx, "_h", None):
            self.M = self.A = self.B = None
            try:
                self.torch.cuda.synchronize(self.device)
                lib().seqalign_arenas_free(self._ctx._h, ptrs)
            except Exception:
                pass
            self._arena_ptrs = None

    def fill(self, ctx: Context, dev_scoring, kernel: int = KERNEL_AUTO, order_after_current: bool = True):
        """Enqueue THE HOT PATH on self.stream (no host sync).  By default the launch
        is ordered after work already queued on torch's current stream (e.g. the
        tests' poison fill)."""
        if order_after_current:
            self.stream.wait_stream(self.torch.cuda.current_stream(self.device))
        _check(lib().seqalign_fill_batch_device(ctx._h, dev_scoring, C.byref(self.desc), C.c_int(kernel),
                                                C.c_void_p(self.stream.cuda_stream)), "seqalign_fill_batch_device")

    def time_fill_ms(self, ctx: Context, dev_scoring, kernel: int, repeats: int):
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))
        st = self.stream.cuda_stream
        ms = (C.c_float * repeats)()
        _check(lib().seqalign_time_fill_ms(ctx._h, dev_scoring, C.byref(self.desc), C.c_int(kernel),
                                           C.c_void_p(st), C.c_int(repeats), ms), "seqalign_time_fill_ms")
        return list(ms)

    def algorithmic_bytes(self) -> int:
        """3*4*(len_a+1)*(len_b+1) written + (len_a+len_b) read per pair."""
        return int(12 * self.cells_host.sum() + self.host.len_a.astype(np.int64).sum()
                   + self.host.len_b.astype(np.int64).sum())

    def pair_matrices(self, p: int):
        o, n = int(self.mat_off_host[p]), int(self.cells_host[p])
        return tuple(x[o:o + n].cpu().numpy() for x in (self.M, self.A, self.B))

    def sw_reduce_launch(self, ctx: Context, min_score: int):
        """Enqueue the SW reduction (best cell + count, no compaction) on self.stream."""
        torch = self.torch
        if not hasattr(self, "_red"):
            n, dev = self.host.n_pairs, self.M.device
            self._red = (torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev),
                         torch.zeros(n, dtype=torch.int32, device=dev))
        bs, bi, cnt = self._red
        r = SwReduceDesc(self.host.n_pairs, self.len_a.data_ptr(), self.len_b.data_ptr(), self.mat_off.data_ptr(),
                         self.M.data_ptr(), min_score, bs.data_ptr(), bi.data_ptr(), cnt.data_ptr(), 0, 0, 0, 0)
        _check(lib().seqalign_sw_reduce_device(ctx._h, C.byref(r), C.c_void_p(self.stream.cuda_stream)),
               "seqalign_sw_reduce_device")

    def sw_reduce(self, ctx: Context, min_score: int, with_candidates: bool = True):
        """Device SW reduction; returns (best_score, best_index, counts, cand lists)."""
        torch = self.torch
        n = self.host.n_pairs
        dev = self.M.device
        best_s = torch.zeros(n, dtype=torch.int32, device=dev)
        best_i = torch.zeros(n, dtype=torch.int64, device=dev)
        count = torch.zeros(n, dtype=torch.int32, device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        st = self.stream.cuda_stream
        r = SwReduceDesc(n, self.len_a.data_ptr(), self.len_b.data_ptr(), self.mat_off.data_ptr(),
                         self.M.data_ptr(), min_score, best_s.data_ptr(), best_i.data_ptr(),
                         count.data_ptr(), 0, 0, 0, 0)
        _check(lib().seqalign_sw_reduce_device(ctx._h, C.byref(r), C.c_void_p(st)), "seqalign_sw_reduce_device")
        torch.cuda.synchronize(self.device)
        cands = None
        if with_candidates:
            cap = count.cpu().numpy().astype(np.uint32)
            off = np.zeros(n, np.uint64)
            if n:
                off[1:] = np.cumsum(cap.astype(np.uint64))[:-1]
            total = int(cap.sum())
            d_off = torch.from_numpy(off.view(np.int64)).to(dev)
            d_cap = torch.from_numpy(cap.view(np.int32)).to(dev)
            c_idx = torch.zeros(total + 1, dtype=torch.int32, device=dev)
            c_sc = torch.zeros(total + 1, dtype=torch.int32, device=dev)
            r.cand_off, r.cand_cap = d_off.data_ptr(), d_cap.data_ptr()
            r.cand_index, r.cand_score = c_idx.data_ptr(), c_sc.data_ptr()
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            _check(lib().seqalign_sw_reduce_device(ctx._h, C.byref(r), C.c_void_p(st)), "seqalign_sw_reduce_device")
            torch.cuda.synchronize(self.device)
            ci, cs = c_idx.cpu().numpy().view(np.uint32), c_sc.cpu().numpy()
            cands = [(ci[int(off[p]):int(off[p]) + int(cap[p])], cs[int(off[p]):int(off[p]) + int(cap[p])])
                     for p in range(n)]
        return best_s.cpu().numpy(), best_i.cpu().numpy(), count.cpu().numpy(), cands


EXPORTED_SYMBOLS = [
    # include/seqalign_hip.h
    "seqalign_strerror", "seqalign_last_error", "seqalign_device_count", "seqalign_ctx_create",
    "seqalign_ctx_destroy", "seqalign_ctx_device", "seqalign_scoring_upload", "seqalign_scoring_release",
    "seqalign_fill_batch_device", "seqalign_sw_reduce_device", "seqalign_nw_traceback_device", "seqalign_sw_traceback_device", "seqalign_fill_batch", "seqalign_nw_batch",
    "seqalign_sw_batch", "seqalign_time_fill_ms", "seqalign_arenas_alloc", "seqalign_arenas_free",
    "seqalign_arenas_info", "seqalign_pool_trim", "seqalign_ctx_set_option", "seqalign_ctx_get_option", "seqalign_ctx_last_call_info",
    "seqalign_kernel_kind_name", "seqalign_host_legs_nw", "seqalign_ctx_stream",
    "seqalign_fill_batch_multi", "seqalign_nw_batch_multi", "seqalign_sw_batch_multi", "seqalign_cigar",
    "seqalign_nw_batch_submit", "seqalign_sw_batch_submit", "seqalign_nw_batch_cigar_submit", "seqalign_job_wait", "seqalign_job_done",
    "seqalign_nw_batch_cigar", "seqalign_sw_batch_cigar", "seqalign_nw_batch_cigar_multi", "seqalign_sw_batch_cigar_multi",
    # include/seqalign_io.h
    "seqalign_scoring_load_matrix", "seqalign_scoring_load_pairs", "seqalign_reader_open", "seqalign_reader_close",
    "seqalign_reader_next",
    # include/seqalign_compat.h (scoring)
    "scoring_init", "scoring_add_wildcard", "scoring_add_mutation", "scoring_add_mutations", "scoring_print",
    "scoring_lookup", "scoring_system_PAM30", "scoring_system_PAM70", "scoring_system_BLOSUM80",
    "scoring_system_BLOSUM62", "scoring_system_DNA_hybridization", "scoring_system_default", "blosum62",
    # include/alignment.h
    "align_col_mismatch", "align_col_indel", "align_col_context", "align_col_stop", "aligner_align",
    "aligner_destroy", "alignment_create", "alignment_ensure_capacity", "alignment_free",
    "alignment_reverse_move", "alignment_print_matrices", "alignment_colour_print_against",
    "alignment_print_spacer",
    # include/needleman_wunsch.h, include/smith_waterman.h
    "needleman_wunsch_new", "needleman_wunsch_free", "needleman_wunsch_align", "needleman_wunsch_align2",
    "smith_waterman_new", "smith_waterman_free", "smith_waterman_get_aligner", "smith_waterman_align",
    "smith_waterman_align2", "smith_waterman_fetch", "sort_match_indices",
]
