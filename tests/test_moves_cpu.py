"""CPU tier: the host half of the "moves instead of strings" path (seq-align_amd/host/sa_moves.c).

The device walkers on direction bytes send home two bit planes per alignment ("gap in a", "gap in b" per walked
column); the host expands them against the sequences into the reference's pair of gapped strings
(reference src/needleman_wunsch.c:82-145, src/smith_waterman.c:187-255).  Here the planes are derived from the ORACLE's
alignments exactly as a walk would emit them, expanded by the product's host code (SIMD and scalar loops), and compared
with the oracle's strings.  No device involved.
"""
import ctypes as C
import random

import numpy as np
import pytest

import orclib as O
import seqalign_amd as S


def planes_from_alignment(ra: bytes, rb: bytes, la: int, lb: int, stop_at_border: bool):
    """What a backwards walk over the alignment (ra, rb) emits: per walked column one bit per plane, right-aligned in
    n_words words; a global walk stops when x == 0 or y == 0 (the reference pads the rest, needleman_wunsch.c:117-132)."""
    n_words = (la + lb + 31) >> 5
    bits_a = np.zeros(32 * n_words, np.uint8)
    bits_b = np.zeros(32 * n_words, np.uint8)
    x, y, k = la, lb, 0
    for col in range(len(ra) - 1, -1, -1):
        if stop_at_border and (x == 0 or y == 0):
            break
        ga, gb = ra[col:col + 1] == b"-", rb[col:col + 1] == b"-"
        bits_a[32 * n_words - 1 - k] = ga
        bits_b[32 * n_words - 1 - k] = gb
        x -= not ga
        y -= not gb
        k += 1
    pack = lambda bits: np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)
    return pack(bits_a), pack(bits_b), n_words, k


def expand_nw(lib, a, b, pa, pb, n_words, n_moves):
    oa = C.create_string_buffer(len(a) + len(b) + 1)
    ob = C.create_string_buffer(len(a) + len(b) + 1)
    n = C.c_uint32(0)
    # garbage in the words the walk did not reach must not matter
    rc = lib.sa_expand_nw_moves(a, C.c_uint32(len(a)), b, C.c_uint32(len(b)), pa.ctypes.data_as(C.c_void_p),
                                pb.ctypes.data_as(C.c_void_p), C.c_uint32(n_words), C.c_uint32(n_moves), oa, ob, C.byref(n))
    return rc, oa.value, ob.value, n.value


@pytest.mark.parametrize("scalar", [0, 1])
def test_nw_moves_expand_to_the_oracle_strings(scalar):
    lib = S.lib()
    lib.sa_moves_force_scalar(C.c_int(scalar))
    rng = random.Random(11 + scalar)
    specs = [{"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, {"init": [2, -3, -1, -1, 0, 0, 0, 0, 0, 0]},
             {"init": [1, -1, 0, -2, 0, 0, 0, 0, 0, 0]}]
    try:
        for trial in range(400):
            osc = O.build_scoring(specs[trial % 3], "oracle")
            la = rng.choice([0, 1, 2, 31, 32, 33, 63, 64, 65, 100, 150, 151, 300, rng.randrange(0, 520)])
            lb = rng.choice([0, 1, 5, 64, 127, 128, 150, 400, rng.randrange(0, 700)])
            a = bytes(rng.choice(b"ACGT") for _ in range(la))
            if trial % 2:   # related pair: long runs of matches, a few gaps
                b = bytearray(a)
                for _ in range(rng.randrange(0, 6)):
                    if b and rng.random() < 0.5:
                        at = rng.randrange(len(b)); del b[at:at + rng.randrange(1, 9)]
                    else:
                        at = rng.randrange(len(b) + 1); b[at:at] = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(1, 9)))
                b = bytes(b)
                lb = len(b)
            else:
                b = bytes(rng.choice(b"ACGT") for _ in range(lb))
            rc, score, ra, rb = O.oracle_nw(osc, a, b)
            assert rc == 0
            pa, pb, n_words, n_moves = planes_from_alignment(ra, rb, la, lb, True)
            if n_words:   # poison what the walk did not write
                first = 32 * n_words - n_moves
                for plane in (pa, pb):
                    for w in range(first // 32):
                        plane[w] = 0xDEADBEEF
                    if first % 32:
                        plane[first // 32] |= np.uint32((1 << (first % 32)) - 1)
            rc, ga, gb, n = expand_nw(lib, a, b, pa, pb, n_words, n_moves)
            assert rc == 0 and (ga, gb, n) == (ra, rb, len(ra)), (trial, la, lb)
    finally:
        lib.sa_moves_force_scalar(C.c_int(0))


def test_nw_moves_reject_planes_that_are_no_walk():
    lib = S.lib()
    a, b = b"ACGT", b"AC"
    n_words = 1
    pa = np.zeros(n_words, np.uint32)
    pb = np.zeros(n_words, np.uint32)
    rc, *_ = expand_nw(lib, a, b, pa, pb, n_words, 3)      # three MATCH moves over a 2-character seq_b
    assert rc == S.E_TRACEBACK
    rc, *_ = expand_nw(lib, a, b, pa, pb, n_words, 40)     # more moves than the slot holds
    assert rc == S.E_TRACEBACK
    # a column with BOTH planes set (stale or corrupt words): no walk has one, and expanding it would overrun len_a + len_b
    pa[0] = pb[0] = np.uint32(1 << 31)
    rc, *_ = expand_nw(lib, a, b, pa, pb, n_words, 2)
    assert rc == S.E_TRACEBACK
    oa, ob, pos = C.create_string_buffer(8), C.create_string_buffer(8), (C.c_uint32 * 4)()
    assert lib.sa_expand_sw_moves(a, b, C.c_uint32(4), C.c_uint32(2), pa.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p),
                                  C.c_uint32(1), C.c_uint32(2), oa, ob, pos) == S.E_TRACEBACK
    # a walk that stopped inside the matrix (both sequences have characters left): not a global alignment
    pa[0] = pb[0] = 0
    rc, *_ = expand_nw(lib, a, b, pa, pb, n_words, 1)
    assert rc == S.E_TRACEBACK


@pytest.mark.parametrize("scalar", [0, 1])
def test_sw_moves_expand_to_the_oracle_hits(scalar):
    lib = S.lib()
    lib.sa_moves_force_scalar(C.c_int(scalar))
    rng = random.Random(5 + scalar)
    osc = O.build_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    try:
        done = 0
        for trial in range(200):
            lb = rng.randrange(40, 400)
            b = bytes(rng.choice(b"ACGT") for _ in range(lb))
            o = rng.randrange(0, lb - 30)
            a = bytearray(b[o:o + rng.randrange(30, 160)])
            for _ in range(rng.randrange(0, 5)):
                at = rng.randrange(len(a))
                if rng.random() < 0.5: del a[at:at + rng.randrange(1, 4)]
                else: a[at:at] = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(1, 4)))
            a = bytes(a)
            rc, hits = O.oracle_sw(osc, a, b, 10, 3)
            assert rc == 0
            for h in hits:
                pos_a, pos_b, len_a, len_b = h["pos_a"], h["pos_b"], h["len_a"], h["len_b"]
                ra, rb = h["a"].encode(), h["b"].encode()
                pa, pb, n_words, n_moves = planes_from_alignment(ra, rb, len(a), len(b), False)
                oa = C.create_string_buffer(len(a) + len(b) + 1)
                ob = C.create_string_buffer(len(a) + len(b) + 1)
                pos = (C.c_uint32 * 4)()
                rc = lib.sa_expand_sw_moves(a, b, C.c_uint32(pos_a + len_a), C.c_uint32(pos_b + len_b),
                                            pa.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p), C.c_uint32(n_words),
                                            C.c_uint32(n_moves), oa, ob, pos)
                assert rc == 0 and (oa.value, ob.value, list(pos)) == (ra, rb, [pos_a, pos_b, len_a, len_b])
                done += 1
        assert done > 100
    finally:
        lib.sa_moves_force_scalar(C.c_int(0))


# ------------------------------------------------------------------ CIGAR straight from the planes ---

def cigar_of_strings(lib, ra: bytes, rb: bytes, extended: int, fold: int) -> bytes:
    """seqalign_cigar (host/sa_alignment.c) over the two gapped strings: the definition the plane encoder must reproduce."""
    lib.seqalign_cigar.restype = C.c_size_t
    out = C.create_string_buffer(2 * len(ra) + 8)
    n = lib.seqalign_cigar(ra, rb, C.c_size_t(len(ra)), C.c_int(extended), C.c_int(fold), out, C.c_size_t(len(out)))
    assert n != C.c_size_t(-1).value
    return out.value


def py_cigar(ra: bytes, rb: bytes, extended: int, fold: int) -> bytes:
    """The same definition once more, in Python (independent of both C encoders): seq_a = query, seq_b = reference."""
    ops = []
    for x, y in zip(ra, rb):
        if x == 45: op = "D"
        elif y == 45: op = "I"
        elif not extended: op = "M"
        else: op = "=" if (bytes([x]).lower() == bytes([y]).lower() if fold else x == y) else "X"
        if ops and ops[-1][0] == op: ops[-1][1] += 1
        else: ops.append([op, 1])
    return "".join(f"{n}{op}" for op, n in ops).encode()


def test_nw_cigar_from_planes_equals_cigar_of_the_strings():
    lib = S.lib()
    rng = random.Random(77)
    specs = [{"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, {"init": [2, -3, -1, -1, 0, 0, 0, 0, 0, 0]}, {"init": [1, -1, 0, -2, 0, 0, 0, 0, 0, 0]}]
    for trial in range(300):
        osc = O.build_scoring(specs[trial % 3], "oracle")
        la = rng.choice([0, 1, 2, 31, 32, 33, 63, 64, 65, 100, 150, 151, 300, rng.randrange(0, 400)])
        a = bytes(rng.choice(b"ACGTacgt") for _ in range(la))
        if trial % 2:
            b = bytearray(a.upper() if trial % 4 == 1 else a)
            for _ in range(rng.randrange(0, 6)):
                if b and rng.random() < 0.5:
                    at = rng.randrange(len(b)); del b[at:at + rng.randrange(1, 9)]
                else:
                    at = rng.randrange(len(b) + 1); b[at:at] = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(1, 9)))
            b = bytes(b)
        else:
            b = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 1, 5, 64, 127, 128, 150, rng.randrange(0, 500)])))
        rc, score, ra, rb = O.oracle_nw(osc, a, b)
        assert rc == 0
        pa, pb, n_words, n_moves = planes_from_alignment(ra, rb, len(a), len(b), True)
        if n_words:   # poison what the walk did not write
            first = 32 * n_words - n_moves
            for plane in (pa, pb):
                for w in range(first // 32):
                    plane[w] = 0xDEADBEEF
                if first % 32:
                    plane[first // 32] |= np.uint32((1 << (first % 32)) - 1)
        for fmt, fold in ((1, 0), (2, 0), (2, 1)):
            want = cigar_of_strings(lib, ra, rb, fmt == 2, fold)
            assert want == py_cigar(ra, rb, fmt == 2, fold)
            n, cols = C.c_uint32(0), C.c_uint32(0)
            args = (a, C.c_uint32(len(a)), b, C.c_uint32(len(b)), pa.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p),
                    C.c_uint32(n_words), C.c_uint32(n_moves), C.c_int(fmt), C.c_int(fold))
            # length only, exact capacity, one byte short
            assert lib.sa_cigar_nw_moves(*args, None, C.c_uint64(0), C.byref(n), C.byref(cols)) == 0
            assert (n.value, cols.value) == (len(want), len(ra)), (trial, fmt)
            out = C.create_string_buffer(b"\xff" * (len(want) + 9), len(want) + 9)
            assert lib.sa_cigar_nw_moves(*args, out, C.c_uint64(len(want) + 1), C.byref(n), C.byref(cols)) == 0
            assert out.raw[:len(want) + 1] == want + b"\0" and out.raw[len(want) + 1:] == b"\xff" * 8, (trial, fmt, want)
            if want:
                out = C.create_string_buffer(b"\xff" * (len(want) + 9), len(want) + 9)
                assert lib.sa_cigar_nw_moves(*args, out, C.c_uint64(len(want)), C.byref(n), C.byref(cols)) == S.E_NOMEM
                assert out.raw[len(want):] == b"\xff" * 9      # never past the capacity it was given


def test_sw_cigar_from_planes_equals_cigar_of_the_hits():
    lib = S.lib()
    rng = random.Random(78)
    osc = O.build_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}, "oracle")
    done = 0
    for trial in range(150):
        lb = rng.randrange(40, 400)
        b = bytes(rng.choice(b"ACGT") for _ in range(lb))
        o = rng.randrange(0, lb - 30)
        a = bytearray(b[o:o + rng.randrange(30, 200)])
        for _ in range(rng.randrange(0, 6)):
            at = rng.randrange(len(a))
            if rng.random() < 0.5: del a[at:at + rng.randrange(1, 4)]
            else: a[at:at] = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(1, 4)))
        a = bytes(a).lower() if trial % 3 == 0 else bytes(a)
        rc, hits = O.oracle_sw(osc, a, b, 10, 3)
        assert rc == 0
        for h in hits:
            ra, rb = h["a"].encode(), h["b"].encode()
            pa, pb, n_words, n_moves = planes_from_alignment(ra, rb, len(a), len(b), False)
            for fmt, fold in ((1, 0), (2, 0), (2, 1)):
                want = py_cigar(ra, rb, fmt == 2, fold)
                assert want == cigar_of_strings(lib, ra, rb, fmt == 2, fold)
                out, pos, n = C.create_string_buffer(len(want) + 1), (C.c_uint32 * 4)(), C.c_uint32(0)
                rc = lib.sa_cigar_sw_moves(a, b, C.c_uint32(h["pos_a"] + h["len_a"]), C.c_uint32(h["pos_b"] + h["len_b"]),
                                           pa.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p), C.c_uint32(n_words),
                                           C.c_uint32(n_moves), C.c_int(fmt), C.c_int(fold), out, C.c_uint64(len(want) + 1), pos, C.byref(n))
                assert rc == 0 and out.value == want and n.value == len(want)
                assert list(pos) == [h["pos_a"], h["pos_b"], h["len_a"], h["len_b"]]
                done += 1
    assert done > 300
