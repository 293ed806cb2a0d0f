"""ctypes access to the CHECKERS: oracle/liboracle.so (our CPU restatement) and,
when it was built in the authoring container, oracle/_ref/libseqalign_ref.so (the
real reference compiled from its own sources).  Test infrastructure only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"

MATCH, GAP_A, GAP_B = 0, 1, 2


class Scoring(C.Structure):
    """scoring_t layout -- reference src/alignment_scoring.h:19-40."""
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


class Alignment(C.Structure):
    """alignment_t layout -- reference src/alignment.h:33-40."""
    _fields_ = [
        ("result_a", C.c_void_p), ("result_b", C.c_void_p),
        ("capacity", C.c_size_t), ("length", C.c_size_t),
        ("pos_a", C.c_size_t), ("pos_b", C.c_size_t),
        ("len_a", C.c_size_t), ("len_b", C.c_size_t),
        ("score", C.c_int),
    ]


class Aligner(C.Structure):
    """aligner_t layout -- reference src/alignment.h:23-30."""
    _fields_ = [
        ("scoring", C.c_void_p), ("seq_a", C.c_char_p), ("seq_b", C.c_char_p),
        ("score_width", C.c_size_t), ("score_height", C.c_size_t),
        ("match_scores", C.POINTER(C.c_int)), ("gap_a_scores", C.POINTER(C.c_int)),
        ("gap_b_scores", C.POINTER(C.c_int)), ("capacity", C.c_size_t),
    ]


class OrcHit(C.Structure):
    _fields_ = [("score", C.c_int32), ("pos_a", C.c_uint64), ("pos_b", C.c_uint64),
                ("len_a", C.c_uint64), ("len_b", C.c_uint64), ("length", C.c_uint64),
                ("str_off", C.c_uint64)]


def scoring_defined_bytes(sc: Scoring) -> bytes:
    """The fields scoring_init defines (wildscores/swap_scores are left
    uninitialised upstream, so only entries whose bit is set are meaningful)."""
    head = bytes(C.string_at(C.addressof(sc), Scoring.wildcards.offset))
    wc = np.frombuffer(sc.wildcards, dtype=np.uint32)
    ss = np.frombuffer(sc.swap_set, dtype=np.uint32).reshape(256, 8)
    ws = np.frombuffer(sc.wildscores, dtype=np.int32)
    sw = np.frombuffer(sc.swap_scores, dtype=np.int32).reshape(256, 256)
    bits_w = ((wc[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(256).astype(bool)
    bits_s = ((ss[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(256, 256).astype(bool)
    # mask out padding bytes inside the head (bool fields are followed by padding)
    pads = bytearray(head)
    for lo, hi in ((14, 16), (25, 28)):
        pads[lo:hi] = b"\0" * (hi - lo)
    return (bytes(pads) + wc.tobytes() + ss.tobytes()
            + np.where(bits_w, ws, 0).astype(np.int32).tobytes()
            + np.where(bits_s, sw, 0).astype(np.int32).tobytes()
            + C.string_at(C.addressof(sc) + Scoring.min_penalty.offset, 8))


# --------------------------------------------------------------------- loading

def _ensure_oracle_built() -> Path:
    so = ORACLE_DIR / "liboracle.so"
    src = ORACLE_DIR / "seqalign_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(ORACLE_DIR), "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


_oracle = None
_ref = None
_ref_tried = False


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(str(_ensure_oracle_built()))
        lib.orc_sizeof_scoring.restype = C.c_size_t
        assert lib.orc_sizeof_scoring() == C.sizeof(Scoring)
        lib.orc_fnv1a64.restype = C.c_uint64
        lib.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        lib.orc_time_fill_batch.restype = C.c_double
        _oracle = lib
    return _oracle


def ref():
    """The compiled reference, or None when oracle/_ref was never built."""
    global _ref, _ref_tried
    if not _ref_tried:
        _ref_tried = True
        so = ORACLE_DIR / "_ref" / "libseqalign_ref.so"
        if so.exists():
            _ref = C.CDLL(str(so))
    return _ref


# ------------------------------------------------------------ scoring builders

PRESETS = ("default", "BLOSUM62", "BLOSUM80", "PAM30", "PAM70", "DNA_hybridization")


def build_scoring(spec: dict, flavour: str = "oracle", lib=None) -> Scoring:
    """Build a scoring_t from a JSON-able spec.

    spec = {"preset": name} or {"init": [match, mismatch, gap_open, gap_extend,
    no_start, no_end, no_gaps_a, no_gaps_b, no_mismatches, case_sensitive]},
    plus optional "wildcards": [[char, score]...], "mutations": [[a, b, score]...],
    "use_match_mismatch": 0/1 and "flags": {field: 0/1} set after the builder.

    flavour: "oracle" (orc_* builders; presets unavailable -> needs lib with
    scoring_system_*), "ref" (reference functions) or "product" (our host lib).
    """
    sc = Scoring()
    C.memset(C.byref(sc), 0, C.sizeof(sc))
    if flavour == "oracle":
        lib = oracle()
        init, add_w, add_m = lib.orc_scoring_init, lib.orc_scoring_add_wildcard, lib.orc_scoring_add_mutation
    else:
        if lib is None:
            lib = ref()
        init, add_w, add_m = lib.scoring_init, lib.scoring_add_wildcard, lib.scoring_add_mutation
    if "preset" in spec:
        if flavour == "oracle":
            raise ValueError("presets are table data; build them with ref/product")
        getattr(lib, "scoring_system_" + spec["preset"])(C.byref(sc))
    else:
        init(C.byref(sc), *[C.c_int(int(v)) for v in spec["init"]])
    for ch, s in spec.get("wildcards", []):
        add_w(C.byref(sc), C.c_char(ch.encode()), C.c_int(s))
    for a, b, s in spec.get("mutations", []):
        add_m(C.byref(sc), C.c_char(a.encode()), C.c_char(b.encode()), C.c_int(s))
    if "use_match_mismatch" in spec:
        sc.use_match_mismatch = bool(spec["use_match_mismatch"])
    for k, v in spec.get("flags", {}).items():
        setattr(sc, k, bool(v))
    return sc


# ------------------------------------------------------------- oracle wrappers

def _buf(b: bytes):
    return C.create_string_buffer(b, len(b) + 1)


def oracle_fill(sc: Scoring, a: bytes, b: bytes, is_sw: int):
    lib = oracle()
    cells = (len(a) + 1) * (len(b) + 1)
    M = np.empty(cells, np.int32); A = np.empty(cells, np.int32); B = np.empty(cells, np.int32)
    rc = lib.orc_fill(C.byref(sc), _buf(a), C.c_size_t(len(a)), _buf(b), C.c_size_t(len(b)),
                      C.c_int(is_sw), M.ctypes.data_as(C.c_void_p),
                      A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p))
    return rc, M, A, B


def oracle_nw_traceback(sc: Scoring, a: bytes, b: bytes, M, A, B):
    lib = oracle()
    n = len(a) + len(b) + 1
    ra, rb = C.create_string_buffer(n), C.create_string_buffer(n)
    ln, score = C.c_size_t(0), C.c_int32(0)
    rc = lib.orc_nw_traceback(C.byref(sc), _buf(a), C.c_size_t(len(a)), _buf(b), C.c_size_t(len(b)),
                              M.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p),
                              B.ctypes.data_as(C.c_void_p), ra, rb, C.byref(ln), C.byref(score))
    return rc, score.value, ra.value, rb.value


def oracle_nw(sc: Scoring, a: bytes, b: bytes):
    rc, M, A, B = oracle_fill(sc, a, b, 0)
    if rc:
        return rc, None, None, None
    return oracle_nw_traceback(sc, a, b, M, A, B)


def oracle_sw_hits(sc: Scoring, a: bytes, b: bytes, M, A, B, min_score: int, max_hits: int = 1 << 30):
    lib = oracle()
    cap_hits = min(max_hits, M.size) + 1
    hits = (OrcHit * cap_hits)()
    str_cap = 2 * M.size + 16
    sa, sb = C.create_string_buffer(str_cap), C.create_string_buffer(str_cap)
    n = C.c_size_t(0)
    rc = lib.orc_sw_hits(C.byref(sc), _buf(a), C.c_size_t(len(a)), _buf(b), C.c_size_t(len(b)),
                         M.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p),
                         B.ctypes.data_as(C.c_void_p), C.c_int32(min_score),
                         C.c_size_t(min(max_hits, M.size)), hits, C.byref(n), sa, sb,
                         C.c_size_t(str_cap))
    out = []
    for k in range(n.value):
        h = hits[k]
        out.append(dict(score=h.score, pos_a=h.pos_a, pos_b=h.pos_b, len_a=h.len_a,
                        len_b=h.len_b,
                        a=sa.raw[h.str_off:h.str_off + h.length].decode(),
                        b=sb.raw[h.str_off:h.str_off + h.length].decode()))
    return rc, out


def oracle_sw(sc: Scoring, a: bytes, b: bytes, min_score: int, max_hits: int = 1 << 30):
    rc, M, A, B = oracle_fill(sc, a, b, 1)
    if rc:
        return rc, []
    return oracle_sw_hits(sc, a, b, M, A, B, min_score, max_hits)


def fnv(arr: np.ndarray) -> int:
    arr = np.ascontiguousarray(arr)
    return int(oracle().orc_fnv1a64(arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))


# ---------------------------------------------------------- reference wrappers

def ref_fill(sc: Scoring, a: bytes, b: bytes, is_sw: int):
    """aligner_align of the REAL reference (src/alignment.c:170-193)."""
    lib = ref()
    al = Aligner()
    C.memset(C.byref(al), 0, C.sizeof(al))
    ba, bb = _buf(a), _buf(b)
    lib.aligner_align(C.byref(al), ba, bb, C.c_size_t(len(a)), C.c_size_t(len(b)),
                      C.byref(sc), C.c_char(bytes([is_sw])))
    cells = (len(a) + 1) * (len(b) + 1)
    M = np.ctypeslib.as_array(al.match_scores, (cells,)).copy()
    A = np.ctypeslib.as_array(al.gap_a_scores, (cells,)).copy()
    B = np.ctypeslib.as_array(al.gap_b_scores, (cells,)).copy()
    lib.aligner_destroy(C.byref(al))
    return M, A, B


def ref_nw(sc: Scoring, a: bytes, b: bytes):
    """needleman_wunsch_align2 of the REAL reference (src/needleman_wunsch.c:34)."""
    lib = ref()
    lib.needleman_wunsch_new.restype = C.c_void_p
    lib.alignment_create.restype = C.c_void_p
    nw = C.c_void_p(lib.needleman_wunsch_new())
    res = C.c_void_p(lib.alignment_create(C.c_size_t(256)))
    ba, bb = _buf(a), _buf(b)
    lib.needleman_wunsch_align2(ba, bb, C.c_size_t(len(a)), C.c_size_t(len(b)),
                                C.byref(sc), nw, res)
    r = Alignment.from_address(res.value)
    out = (r.score, C.string_at(r.result_a), C.string_at(r.result_b))
    lib.alignment_free(res)
    lib.needleman_wunsch_free(nw)
    return out


def ref_lookup(sc: Scoring, a: int, b: int):
    lib = ref()
    s, m = C.c_int(0), C.c_bool(False)
    lib.scoring_lookup(C.byref(sc), C.c_char(bytes([a])), C.c_char(bytes([b])), C.byref(s), C.byref(m))
    return s.value, int(m.value)


def ref_sw_hits(sc: Scoring, a: bytes, b: bytes, min_score: int, max_hits: int = 1 << 30):
    """The reference's SW hit list with EVERY piece of arithmetic done by the compiled reference (oracle/_ref): the fill is its
    aligner_align(is_sw = 1) (src/alignment.c:170-193), every step of every walk its alignment_reverse_move (:244-350).  Restated
    here -- smith_waterman.c itself cannot be compiled (it needs the un-vendored sort_r) -- are only the two pieces that carry no
    arithmetic: the candidate ORDER (smith_waterman.c:152-161 collects every pos with M > 0 in ascending pos; the comparator :71-86
    orders by score descending, then column pos % W ascending, and returns 0 beyond that -- a stable sort leaves ascending pos,
    SURVEY A.3-4) and the VISITED MASK of smith_waterman_fetch / _follow_hit (:165-277: a candidate already visited is skipped; a
    walk marks each cell it stands on, is abandoned -- its marks stay -- at a cell somebody marked before, and is a hit when it
    reaches score 0).  Fresh mask per pair.  Hits come out in descending score, so "score >= min_score" (the
    command line's --minscore, sw_cmdline.c:214-217) is a prefix: candidates below it are never needed.
    An enumeration independent of oracle/seqalign_oracle.c's orc_sw_hits: the two must agree on every hit."""
    lib = ref()
    al = Aligner()
    C.memset(C.byref(al), 0, C.sizeof(al))
    ba, bb = _buf(a), _buf(b)
    lib.aligner_align(C.byref(al), ba, bb, C.c_size_t(len(a)), C.c_size_t(len(b)), C.byref(sc), C.c_char(b"\1"))
    W_ = len(a) + 1
    M = np.ctypeslib.as_array(al.match_scores, (W_ * (len(b) + 1),))
    cand = np.flatnonzero(M >= max(int(min_score), 1))
    order = cand[np.lexsort((cand, cand % W_, -M[cand].astype(np.int64)))]
    visited = np.zeros(M.size, bool)
    hits = []
    mat, score, cx, cy, idx = C.c_int(0), C.c_int(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    for pos in order.tolist():
        if len(hits) >= max_hits:
            break
        if visited[pos]:
            continue
        x0, y0 = pos % W_, pos // W_
        mat.value, score.value, cx.value, cy.value, idx.value = MATCH, int(M[pos]), x0, y0, pos
        ra, rb, alive = [], [], True
        while True:
            if visited[idx.value]:
                alive = False
                break
            visited[idx.value] = True
            if score.value == 0:
                break
            ra.append("-" if mat.value == GAP_A else chr(a[cx.value - 1]))
            rb.append("-" if mat.value == GAP_B else chr(b[cy.value - 1]))
            lib.alignment_reverse_move(C.byref(mat), C.byref(score), C.byref(cx), C.byref(cy), C.byref(idx), C.byref(al))
        if alive:
            hits.append(dict(score=int(M[pos]), pos_a=cx.value, pos_b=cy.value, len_a=x0 - cx.value, len_b=y0 - cy.value,
                             a="".join(reversed(ra)), b="".join(reversed(rb))))
    lib.aligner_destroy(C.byref(al))
    return hits
