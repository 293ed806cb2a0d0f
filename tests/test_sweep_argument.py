"""The multi-hit path's central claim, checked on the CPU against the sequential procedure.

smith_waterman.c:165-277 visits candidates in (score desc, column asc, index asc) order, walks each back marking
cells, and abandons a walk at a marked cell.  sa_sw_sweep.hip never runs that procedure: it says a cell is won by the
lowest-ranked walk that ever arrives at it, that a walk moves on exactly when it won the cell it stands on, and that
the winners whose state has score 0 are the hits -- so one reverse pass over the cells settles everything.  Here that
pass is restated in a few lines of Python on the oracle's matrices and compared with the oracle's own sequential
enumeration (oracle/seqalign_oracle.c, pinned against the reference's tests): same hits, same order, for random
scorings, related / random / repetitive pairs, low thresholds (many hits, ties) and every max_hits cut."""
import numpy as np
import pytest

import orclib as O

MATCH, GAP_A, GAP_B = 0, 1, 2


def sweep_hits(M, A, B, a, b, match, mismatch, gap_open, gap_extend, min_score):
    """Every hit of the pair as (score, end column, end row), in the reference's order."""
    Wd, H = len(a) + 1, len(b) + 1
    S = (M.reshape(H, Wd), A.reshape(H, Wd), B.reshape(H, Wd))
    go, ge = gap_open + gap_extend, gap_extend

    def pred_state(x, y, st):
        """alignment_reverse_move (alignment.c:244-350): the matrix the walk comes from, GAP_A before GAP_B before MATCH."""
        s = int(S[st][y, x])
        if st == MATCH:
            sub = match if a[x - 1] == b[y - 1] else mismatch
            px, py, va, vb = x - 1, y - 1, sub, sub
        elif st == GAP_A:
            px, py, va, vb = x, y - 1, ge, go
        else:
            px, py, va, vb = x - 1, y, go, ge
        if int(S[GAP_A][py, px]) + va == s:
            return GAP_A
        if int(S[GAP_B][py, px]) + vb == s:
            return GAP_B
        return MATCH

    # key: ascending = (score desc, column asc, row asc)
    NONE = (1 << 62, 0)
    win = {}          # cell -> (key, state) of its winner
    hits = []
    for y in range(H - 1, -1, -1):
        for x in range(Wd - 1, -1, -1):
            best = NONE
            if S[MATCH][y, x] >= max(min_score, 1):
                best = ((-int(S[MATCH][y, x]), x, y), MATCH)
            # the winners of the three cells a backward move can come from, if they move here
            for dx, dy, via in ((1, 1, MATCH), (0, 1, GAP_A), (1, 0, GAP_B)):
                src = win.get((x + dx, y + dy))
                if src is None or src[1] != via or int(S[via][y + dy, x + dx]) == 0:
                    continue                      # nobody there, it stands in another state, or its walk ended there
                arrives = (src[0], pred_state(x + dx, y + dy, via))
                if best is NONE or arrives[0] < best[0]:
                    best = arrives
            if best is NONE:
                continue
            win[(x, y)] = best
            if int(S[best[1]][y, x]) == 0:        # a winner whose state has score 0: a hit
                hits.append(best[0])
    hits.sort()
    return [(-k[0], k[1], k[2]) for k in hits]


@pytest.mark.parametrize("seed", range(6))
def test_reverse_sweep_equals_the_sequential_procedure(seed):
    rng = np.random.default_rng(1000 + seed)
    checked = 0
    for trial in range(40):
        match, mismatch = int(rng.integers(1, 4)), -int(rng.integers(0, 4))
        gap_open, gap_extend = -int(rng.integers(0, 5)), -int(rng.integers(0, 3))
        sc = O.build_scoring({"init": [match, mismatch, gap_open, gap_extend, 0, 0, 0, 0, 0, 0]}, "oracle")
        alpha = b"ACGT"[: int(rng.integers(2, 5))]
        la, lb = int(rng.integers(1, 34)), int(rng.integers(1, 34))
        a = bytes(rng.choice(list(alpha), la).tolist())
        kind = trial % 3
        if kind == 0:
            b = bytes(rng.choice(list(alpha), lb).tolist())
        elif kind == 1:                                   # related: a piece of a inside b
            cut = int(rng.integers(0, la))
            b = (bytes(rng.choice(list(alpha), 3).tolist()) + a[cut:] + bytes(rng.choice(list(alpha), lb).tolist()))[: max(lb, 4)]
        else:                                             # tandem repeats: ties, many hits
            unit = bytes(rng.choice(list(alpha), int(rng.integers(1, 5))).tolist())
            a, b = (unit * 12)[:la], (unit * 12)[:lb]
        thr = int(rng.integers(1, 6)) * match
        rc, M, A, B = O.oracle_fill(sc, a, b, 1)
        assert rc == 0
        mine = sweep_hits(M, A, B, a, b, match, mismatch, gap_open, gap_extend, thr)
        for max_hits in (1, 3, 1 << 30):
            rc, want = O.oracle_sw_hits(sc, a, b, M, A, B, thr, max_hits)
            assert rc == 0
            got = mine[:max_hits]
            assert [(h["score"], h["pos_a"] + h["len_a"], h["pos_b"] + h["len_b"]) for h in want] == got, (seed, trial, a, b, thr, max_hits)
            checked += len(want)
    assert checked > 50
