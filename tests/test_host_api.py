"""CPU-only checks of the product's host layer: the C-ABI library loads and
exports every declared symbol, the scoring builders / presets equal the
reference's (via committed digests), and the scoring flatten is exact.
No compute entry point is called here (no GPU in this tier).
"""
import ctypes as C
import itertools
import json
import re
from pathlib import Path

import numpy as np
import pytest

import orclib as O
import seqalign_amd as S

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"


def test_library_exports_every_declared_symbol():
    lib = S.lib()
    missing = [s for s in S.EXPORTED_SYMBOLS if not hasattr(lib, s)]
    assert not missing
    # every function declared in include/*.h is in EXPORTED_SYMBOLS
    declared = set()
    for h in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        text = re.sub(r"#define[^\n]*(\\\n[^\n]*)*", "", text)
        for m in re.finditer(r"\b([a-z_0-9A-Z]+)\s*\([^;{]*\)\s*;", text):
            declared.add(m.group(1))
    declared -= {"defined", "sizeof"}
    assert declared <= set(S.EXPORTED_SYMBOLS), declared - set(S.EXPORTED_SYMBOLS)


def test_struct_sizes_match_reference_layout():
    assert C.sizeof(S.Scoring) == 271428
    assert C.sizeof(O.Aligner) == 72 and C.sizeof(O.Alignment) == 72


def test_no_device_is_reported_not_faked():
    """No GPU in the CPU tier: context creation must FAIL, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p(0)
    rc = S.lib().seqalign_ctx_create(C.c_int(0), C.byref(h))
    assert rc == S.E_NO_DEVICE and not h.value
    assert S.lib().seqalign_device_count() == 0


@pytest.mark.parametrize("name", O.PRESETS)
def test_presets_equal_reference(name):
    entry = json.loads((GOLD / "presets.json").read_text())[name]
    sc = S.make_scoring({"preset": name})
    raw = O.Scoring.from_buffer_copy(bytes(sc))
    digest = O.fnv(np.frombuffer(O.scoring_defined_bytes(raw), np.uint8))
    assert f"{digest:016x}" == entry["digest"]
    for x, y, want_s, want_m in entry["lookup"]:
        s, m = C.c_int(0), C.c_bool(False)
        S.lib().scoring_lookup(C.byref(sc), C.c_char(x.encode()), C.c_char(y.encode()), C.byref(s), C.byref(m))
        assert (s.value, int(m.value)) == (want_s, want_m)


def test_blosum62_export_matches_preset():
    tbl = (C.c_int * 576).in_dll(S.lib(), "blosum62")
    sc = S.make_scoring({"preset": "BLOSUM62"})
    letters = "arndcqeghilkmfpstwyvbzx*"
    for i, x in enumerate(letters):
        for j, y in enumerate(letters):
            assert tbl[j * 24 + i] == sc.swap_scores[ord(x)][ord(y)]


def test_builders_equal_oracle_builders():
    for flags in itertools.product([0, 1], repeat=6):
        spec = {"init": [2, -3, -5, -2, *flags], "wildcards": [["N", 0], ["x", -1]],
                "mutations": [["a", "c", -3], ["c", "a", -1], ["G", "T", 4]]}
        ours = O.Scoring.from_buffer_copy(bytes(S.make_scoring(spec)))
        assert O.scoring_defined_bytes(ours) == O.scoring_defined_bytes(O.build_scoring(spec, "oracle"))


class Flat(C.Structure):
    """sa_flat_scoring_t (seq-align_amd/host/sa_internal.h)"""
    _fields_ = [("gap_open", C.c_int32), ("open1", C.c_int32), ("ext", C.c_int32), ("floor", C.c_int32),
                ("gen_eq", C.c_int32), ("gen_ne", C.c_int32), ("flags", C.c_uint32), ("n_classes", C.c_uint32),
                ("code", C.c_uint16 * 256), ("table", C.POINTER(C.c_int32))]


BLOCKED, UNKNOWN = -2**31, -2**31 + 1


def flat_lookup(f: Flat, a: int, b: int) -> int:
    """What the kernels compute from the flattened form (sa_fill_common.hpp)."""
    ca, cb = f.code[a], f.code[b]
    fa, fb, ka, kb = ca & 0xFF, cb & 0xFF, ca >> 8, cb >> 8
    K = f.n_classes
    if K <= 1:
        return f.gen_eq if fa == fb else f.gen_ne
    s = f.table[ka * K + kb]
    if (ka | kb) == 0 and fa != fb:
        s = f.gen_ne
    return s


@pytest.mark.parametrize("spec", [
    {"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]},
    {"init": [1, -2, -4, -1, 0, 0, 0, 0, 1, 0], "wildcards": [["N", 0], ["R", -1]]},
    {"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 1], "wildcards": [["N", 0]],
     "mutations": [["a", "c", -3], ["c", "a", 2], ["A", "C", 1], ["n", "a", -4]]},
    {"init": [1, -2, -4, -1, 0, 0, 0, 0, 1, 1], "mutations": [["a", "c", -3], ["G", "T", 2]]},
    {"preset": "BLOSUM62"}, {"preset": "PAM30"}, {"preset": "DNA_hybridization"},
    {"init": [3, -1, -2, -1, 0, 0, 0, 0, 0, 0], "use_match_mismatch": 0, "mutations": [["a", "a", 2], ["a", "c", -1]]},
], ids=lambda s: json.dumps(s)[:60])
def test_flatten_is_exact_for_every_char_pair(spec):
    """flat(code, table) == scoring_lookup for all 128x128 ASCII pairs."""
    sc = S.make_scoring(spec)
    osc = O.Scoring.from_buffer_copy(bytes(sc))
    f = Flat()
    rc = S.lib().sa_flatten_scoring(C.byref(sc), C.c_int(1), C.byref(f))
    assert rc == 0
    lk = O.oracle().orc_scoring_lookup
    s, m = C.c_int(0), C.c_int(0)
    for a in range(128):
        for b in range(128):
            rc = lk(C.byref(osc), C.c_char(bytes([a])), C.c_char(bytes([b])), C.byref(s), C.byref(m))
            want = UNKNOWN if rc else (BLOCKED if (osc.no_mismatches and not m.value) else s.value)
            assert flat_lookup(f, a, b) == want, (chr(a), chr(b))
    S.lib().sa_flat_scoring_free(C.byref(f))


def test_flatten_rejects_nw_scorings_outside_the_parity_domain():
    """gap penalties below -|min_penalty| overflow the reference."""
    sc = S.make_scoring({"init": [1, -2, -4, -1, 0, 0, 1, 1, 0, 0]})   # both no_gaps: min_penalty=-2, open1=-5
    f = Flat()
    assert S.lib().sa_flatten_scoring(C.byref(sc), C.c_int(0), C.byref(f)) == S.E_DOMAIN
    assert S.lib().sa_flatten_scoring(C.byref(sc), C.c_int(1), C.byref(f)) == 0   # SW: floor 0, defined
    S.lib().sa_flat_scoring_free(C.byref(f))


def test_host_traceback_matches_oracle_on_oracle_matrices():
    """sa_nw_traceback (product, host C) on matrices from the oracle fill: checks the
    host consumer independently of the GPU."""
    from seqalign_amd import workloads as W

    class View(C.Structure):
        _fields_ = [("sc", C.c_void_p), ("a", C.c_char_p), ("b", C.c_char_p), ("len_a", C.c_size_t),
                    ("len_b", C.c_size_t), ("M", C.c_void_p), ("A", C.c_void_p), ("B", C.c_void_p)]
    for flags in itertools.product([0, 1], repeat=5):
        if flags[2] and flags[3]:
            continue
        spec = {"init": [1, -2, -4, -1, *flags, 0]}
        sc, osc = S.make_scoring(spec), O.build_scoring(spec, "oracle")
        batch = W.ragged(6, seed=sum(flags) + 40, max_len=30)
        for p in range(batch.n_pairs):
            a, b = batch.seq_a(p), batch.seq_b(p)
            rc, M, A, B = O.oracle_fill(osc, a, b, 0)
            want = O.oracle_nw_traceback(osc, a, b, M, A, B)
            v = View(C.addressof(sc), a, b, len(a), len(b), M.ctypes.data, A.ctypes.data, B.ctypes.data)
            ra, rb = C.create_string_buffer(len(a) + len(b) + 1), C.create_string_buffer(len(a) + len(b) + 1)
            n, score = C.c_size_t(0), C.c_int32(0)
            rc = S.lib().sa_nw_traceback(C.byref(v), ra, rb, C.byref(n), C.byref(score))
            assert (rc, score.value, ra.value, rb.value) == want


def test_cigar_of_reference_alignments():
    """north_star asks for identical CIGAR / alignment strings; the reference has no CIGAR, so it is the derived format
    of the two gapped strings (seqalign_cigar).  Checked on the golden NW alignments of the compiled reference against a
    Python run-length encoding, plain and extended ops, and on the error paths."""
    lib = S.lib()
    lib.seqalign_cigar.restype = C.c_size_t

    def rle(a, b, extended, fold):
        ops = []
        for x, y in zip(a, b):
            op = "D" if x == "-" else "I" if y == "-" else "M" if not extended else \
                ("=" if (x.lower() == y.lower() if fold else x == y) else "X")
            if ops and ops[-1][1] == op:
                ops[-1][0] += 1
            else:
                ops.append([1, op])
        return "".join(f"{n}{o}" for n, o in ops)

    cfg = json.loads((GOLD / "configs.json").read_text())
    n = 0
    for name in ("C2_related", "C5"):
        for g in cfg[name]["pairs"][:32]:
            a, b = g["result_a"], g["result_b"]
            for ext, fold in ((0, 0), (1, 0), (1, 1)):
                out = C.create_string_buffer(4 * len(a) + 8)
                ln = lib.seqalign_cigar(a.encode(), b.encode(), C.c_size_t(len(a)), ext, fold, out, C.c_size_t(len(out)))
                assert ln == len(out.value) and out.value.decode() == rle(a, b, ext, fold), (name, a, b)
                n += 1
    assert n == 192
    out = C.create_string_buffer(64)
    assert lib.seqalign_cigar(b"AC-T", b"ACGT", C.c_size_t(4), 0, 0, out, C.c_size_t(64)) == 6 and out.value == b"2M1D1M"
    assert lib.seqalign_cigar(b"acgt", b"ACGA", C.c_size_t(4), 1, 1, out, C.c_size_t(64)) == 4 and out.value == b"3=1X"
    assert lib.seqalign_cigar(b"", b"", C.c_size_t(0), 0, 0, out, C.c_size_t(64)) == 0 and out.value == b""
    assert lib.seqalign_cigar(b"A-", b"A-", C.c_size_t(2), 0, 0, out, C.c_size_t(64)) == C.c_size_t(-1).value
    assert lib.seqalign_cigar(b"ACGT", b"ACGT", C.c_size_t(4), 0, 0, out, C.c_size_t(2)) == C.c_size_t(-1).value
