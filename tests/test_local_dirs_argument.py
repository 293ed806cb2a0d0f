"""The LOCAL form of the direction byte (seq-align_amd/csrc/sa_kernels.h: SA_LD_*), checked on the CPU against the oracle.

The older direction byte answers, per state, "in which state does a walk that leaves this cell arrive" -- for MATCH and
GAP_A a fact about a NEIGHBOUR cell, which the fills carry over and convert into two-bit codes.  The local form stores
each cell's OWN five comparisons (GA, BM, CA, FA, FB; Smith-Waterman: three "this state's score is 0" bits) and the
tile walkers (sa_traceback.hip: local_depart / local_arrive) read the state a walk arrives in from the byte of the cell it arrives
at.  Here both halves are restated in a few lines of Python on the oracle's matrices -- the byte exactly as
nw_dirs_x1_wave<.., LOCAL> / sw_best_x2_wave<.., LOCAL> define it, the walk exactly as the walkers take it -- and the
resulting alignments are compared with the oracle's own tracebacks (alignment_reverse_move, alignment.c:244-350,
pinned against the compiled reference): same strings, for random plain scorings and random / related / repetitive
pairs.  No GPU involved: this pins the ARGUMENT; tests/test_gpu_*.py pin the kernels."""
import numpy as np
import pytest

import orclib as O

MATCH, GAP_A, GAP_B = 0, 1, 2
GA, BM, CA, FA, FB, END0 = 1, 2, 4, 8, 16, 32


def local_bytes(M, A, B, la, lb, open1, ext, is_sw):
    """One byte per cell, from the three matrices: the comparisons the fills make while they compute the cell."""
    W, H = la + 1, lb + 1
    M, A, B = (m.reshape(H, W).astype(np.int64) for m in (M, A, B))
    D = np.zeros((H, W), dtype=np.uint8)
    for y in range(1, H):
        for x in range(1, W):
            m, a, b = int(M[y, x]), int(A[y, x]), int(B[y, x])
            v = 0
            if a >= max(m, b):
                v |= GA
            if b >= m:
                v |= BM
            if int(A[y - 1, x]) + ext == a:
                v |= CA
            if int(A[y, x - 1]) + open1 == b:
                v |= FA
            if int(B[y, x - 1]) + ext == b:
                v |= FB
            if is_sw:
                v |= (END0 if m == 0 else 0) | (END0 << 1 if a == 0 else 0) | (END0 << 2 if b == 0 else 0)
            D[y, x] = v
    if is_sw:
        D[0, :] = 7 * END0
        D[:, 0] = 7 * END0
    return D


def local_depart(st, cur):
    """sa_traceback.hip: local_depart -- leaving `cur`'s cell in state st: which bits of the arrival cell's byte decide, what is decided."""
    from_b = (0x64 >> (2 * ((cur >> 3) & 3))) & 3
    ca = bool(cur & CA)
    am = 3 if st == MATCH else (2 if st == GAP_A and not ca else 0)
    fx = from_b if st == GAP_B else (GAP_A if st == GAP_A and ca else 0)
    return am, fx


def local_arrive(cur, am, fx):
    """sa_traceback.hip: local_arrive."""
    return (0x64 >> (2 * ((cur & am) | fx))) & 3


def walk(D, a, b, x, y, st, is_sw):
    """The tile walkers' walk on local bytes; returns the two gapped strings (NW: padded as needleman_wunsch.c:117-132) and the end."""
    ra, rb = [], []
    am, fx = 0, st
    while True:
        if not is_sw and (x == 0 or y == 0):
            break
        cur = int(D[y, x])
        st = local_arrive(cur, am, fx)
        if is_sw and (cur >> (5 + st)) & 1:
            break
        if st == MATCH:
            ra.append(a[x - 1]); rb.append(b[y - 1]); x -= 1; y -= 1
        elif st == GAP_A:
            ra.append(ord("-")); rb.append(b[y - 1]); y -= 1
        else:
            ra.append(a[x - 1]); rb.append(ord("-")); x -= 1
        am, fx = local_depart(st, cur)
    if not is_sw:
        while x > 0:
            ra.append(a[x - 1]); rb.append(ord("-")); x -= 1
        while y > 0:
            ra.append(ord("-")); rb.append(b[y - 1]); y -= 1
    return bytes(reversed(ra)), bytes(reversed(rb)), x, y


def random_pair(rng, kind, la, lb, alphabet=b"ACGT"):
    a = bytes(rng.choice(list(alphabet), size=la).tolist())
    if kind == "random":
        b = bytes(rng.choice(list(alphabet), size=lb).tolist())
    elif kind == "related":
        out = []
        for ch in a:
            r = rng.random()
            if r < 0.08:
                continue
            out.append(int(rng.choice(list(alphabet))) if r < 0.16 else ch)
            if rng.random() < 0.06:
                out.extend(rng.choice(list(alphabet), size=int(rng.integers(1, 4))).tolist())
        b = bytes(out[:max(lb, 1)]) or b"A"
    else:   # repeats: ties everywhere
        unit = bytes(rng.choice(list(alphabet), size=int(rng.integers(1, 4))).tolist())
        a = (unit * (la // len(unit) + 1))[:la]
        b = (unit * (lb // len(unit) + 1))[:lb]
    return a, b


def plain_scoring(rng):
    match = int(rng.integers(1, 6))
    mismatch = -int(rng.integers(0, 6))
    gap_open = -int(rng.integers(0, 8))
    gap_extend = -int(rng.integers(0, 4))
    return match, mismatch, gap_open, gap_extend


@pytest.mark.parametrize("seed", range(6))
def test_nw_walks_on_local_bytes_equal_the_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    for trial in range(20):
        match, mismatch, gap_open, gap_extend = plain_scoring(rng)
        sc = O.build_scoring({"init": [match, mismatch, gap_open, gap_extend, 0, 0, 0, 0, 0, 0]}, "oracle")
        kind = ("random", "related", "repeats")[trial % 3]
        a, b = random_pair(rng, kind, int(rng.integers(1, 40)), int(rng.integers(1, 40)))
        rc, M, A, B = O.oracle_fill(sc, a, b, 0)
        assert rc == 0
        rc, score, ra, rb = O.oracle_nw(sc, a, b)
        assert rc == 0
        la, lb = len(a), len(b)
        D = local_bytes(M, A, B, la, lb, gap_open + gap_extend, gap_extend, False)
        W = la + 1
        m, ga, gb = int(M[lb * W + la]), int(A[lb * W + la]), int(B[lb * W + la])
        st, s = MATCH, m                      # needleman_wunsch.c:53-66 as the fills report it: GAP_A >= GAP_B >= MATCH on ties
        if gb >= s:
            st, s = GAP_B, gb
        if ga >= s:
            st, s = GAP_A, ga
        assert s == score
        wa, wb, _, _ = walk(D, a, b, la, lb, st, False)
        assert (wa, wb) == (ra, rb), (seed, trial, kind, (match, mismatch, gap_open, gap_extend), a, b)


@pytest.mark.parametrize("seed", range(6))
def test_sw_best_hit_walks_on_local_bytes_equal_the_oracle(seed):
    rng = np.random.default_rng(200 + seed)
    for trial in range(20):
        match, mismatch, gap_open, gap_extend = plain_scoring(rng)
        sc = O.build_scoring({"init": [match, mismatch, gap_open, gap_extend, 0, 0, 0, 0, 0, 0]}, "oracle")
        kind = ("random", "related", "repeats")[trial % 3]
        a, b = random_pair(rng, kind, int(rng.integers(1, 40)), int(rng.integers(1, 60)))
        rc, M, A, B = O.oracle_fill(sc, a, b, 1)
        assert rc == 0
        rc, hits = O.oracle_sw_hits(sc, a, b, M, A, B, 1, 1)
        assert rc == 0
        la, lb = len(a), len(b)
        W = la + 1
        D = local_bytes(M, A, B, la, lb, gap_open + gap_extend, gap_extend, True)
        if not hits:
            assert int(M.max()) <= 0
            continue
        h = hits[0]
        # the best cell as the fills report it: highest score, then lowest column, then lowest row (smith_waterman.c:71-86)
        Mm = M.reshape(lb + 1, W)
        best = int(Mm.max())
        ys, xs = np.nonzero(Mm == best)
        x = int(xs.min()); y = int(ys[xs == x].min())
        wa, wb, ex, ey = walk(D, a, b, x, y, MATCH, True)
        assert h["score"] == best and (wa.decode(), wb.decode()) == (h["a"], h["b"]), (seed, trial, kind, a, b, h)
        assert (ex, ey) == (h["pos_a"], h["pos_b"])


AMINO = b"ARNDCQEGHILKMFPSTWYV"


@pytest.mark.parametrize("seed", range(3))
def test_walks_on_local_bytes_equal_the_oracle_with_a_substitution_table(seed):
    """BLOSUM62 (BASELINE configs[3]'s scoring; alignment_scoring.c:349-360): the decisions never look at the substitution score itself,
    only at the matrices it produced -- global alignments and best local hits of protein pairs, related and random."""
    rng = np.random.default_rng(300 + seed)
    import seqalign_amd as S    # (the preset comes from the product's builder -- byte-equal to the compiled reference's: tests/test_host_api.py)
    sc = O.Scoring.from_buffer_copy(bytes(S.make_scoring({"preset": "BLOSUM62"})))
    gap_open, gap_extend = int(sc.gap_open), int(sc.gap_extend)
    for trial in range(12):
        kind = ("random", "related")[trial % 2]
        a, b = random_pair(rng, kind, int(rng.integers(1, 45)), int(rng.integers(1, 45)), alphabet=AMINO)
        la, lb = len(a), len(b)
        W = la + 1
        rc, M, A, B = O.oracle_fill(sc, a, b, 0)
        assert rc == 0
        rc, score, ra, rb = O.oracle_nw(sc, a, b)
        assert rc == 0
        D = local_bytes(M, A, B, la, lb, gap_open + gap_extend, gap_extend, False)
        m, ga, gb = int(M[lb * W + la]), int(A[lb * W + la]), int(B[lb * W + la])
        st, s = MATCH, m
        if gb >= s:
            st, s = GAP_B, gb
        if ga >= s:
            st, s = GAP_A, ga
        wa, wb, _, _ = walk(D, a, b, la, lb, st, False)
        assert s == score and (wa, wb) == (ra, rb), (seed, trial, kind, a, b)
        rc, M, A, B = O.oracle_fill(sc, a, b, 1)
        assert rc == 0
        rc, hits = O.oracle_sw_hits(sc, a, b, M, A, B, 1, 1)
        assert rc == 0
        if not hits:
            continue
        D = local_bytes(M, A, B, la, lb, gap_open + gap_extend, gap_extend, True)
        Mm = M.reshape(lb + 1, W)
        best = int(Mm.max())
        ys, xs = np.nonzero(Mm == best)
        x = int(xs.min()); y = int(ys[xs == x].min())
        wa, wb, ex, ey = walk(D, a, b, x, y, MATCH, True)
        h = hits[0]
        assert h["score"] == best and (wa.decode(), wb.decode()) == (h["a"], h["b"]) and (ex, ey) == (h["pos_a"], h["pos_b"]), (seed, trial, kind, a, b, h)

