"""GPU tier: WHICH kernels a call launched (seqalign_ctx_last_call_info), and the packed int16 fills at the edge of
their admission bound.

Every form of the fill / sweep / walk gives the same results (that is what test_gpu_parity.py checks), so equal results
cannot tell whether the packed two-pairs-per-wave kernels (sa_fill_dirs_x2.hip) ran or silently declined.  These tests
ask the library what it launched -- and run the packed kernels where their scores come closest to leaving int16:
the arithmetic the halves must reproduce is the reference's plain-int recurrence, src/alignment.c:101-155.
"""
import itertools

import numpy as np
import pytest

import orclib as O
import seqalign_amd as S
from seqalign_amd import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with S.Context(0) as c:
        yield c


@pytest.fixture
def opts(ctx):
    changed = {}

    def set_(**kv):
        for k, v in kv.items():
            changed.setdefault(k, ctx.get_option(k))
            ctx.set_option(k, v)
    yield set_
    for k, v in changed.items():
        ctx.set_option(k, v)


def osc_of(sc):
    return O.Scoring.from_buffer_copy(bytes(sc))


def uniform(n, la, lb, seed, alpha=b"ACGT", related=0.5):
    rng = W.Rng(seed)
    al = np.frombuffer(alpha, np.uint8)
    a = al[rng.below(len(al), n * la).astype(np.int64)].reshape(n, la)
    b = al[rng.below(len(al), n * lb).astype(np.int64)].reshape(n, lb)
    k = min(la, lb)
    if k:
        keep = rng.unit(n * k).reshape(n, k) < 0.8
        m = int(n * related)
        b[:m, :k] = np.where(keep[:m], a[:m, :k], b[:m, :k])
    return W._fixed_batch(a, b)


# ------------------------------------------------------------------ which kernels ran ---

def test_sw_best_hit_fill_takes_rows_up_to_1024_columns(ctx, opts):
    """Round 5: the packed best-hit fill (direction bytes + the best cell, no matrices) has no sweep behind it and takes rows up
    to 1 024 columns like the NW fill; the multi-hit fill takes such rows only from 128 pairs up (next test)."""
    opts(pack16=2)
    sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    osc = osc_of(sc)
    for la, best_packed in ((700, True), (1023, True), (1024, False)):
        batch = uniform(8, la, 200, seed=700 + la)
        one = ctx.sw_batch(batch, sc, 30, max_hits=1)
        assert ("fill_sw_best_x2" in ctx.last_call()) == best_packed, (la, ctx.last_call())
        many = ctx.sw_batch(batch, sc, 30, max_hits=3)
        assert "fill_sw_dirs_x2" not in ctx.last_call() and "fill_sw_dirs" not in ctx.last_call(), ctx.last_call()
        for p in range(batch.n_pairs):
            rc, want = O.oracle_sw(osc, batch.seq_a(p), batch.seq_b(p), 30, 3)
            assert rc == 0 and many[p] == want and one[p] == want[:1], (la, p)


@pytest.mark.parametrize("pack16", [2, 0])
@pytest.mark.parametrize("la,lb,wide_keys", [(512, 150, False), (700, 200, False), (767, 90, False), (768, 130, False), (1023, 200, False),
                                             (640, 2500, True), (1000, 2100, True),
                                             # keys of 31 bits in the layout's fields, < 2^30 in mixed radix (sw_sweep_dirs_ev_kernel<.., true>)
                                             (700, 1000, True), (600, 1000, True), (500, 2100, True), (767, 900, True)])
def test_sw_multi_hit_direction_path_takes_rows_up_to_1024_columns(ctx, opts, pack16, la, lb, wide_keys):
    """Round 5: match_scores + direction bytes and the one-word sweep behind them (sw_sweep_dirs_ev_kernel, 12 / 16 columns per lane)
    for rows of 513 .. 1 024 columns -- offered from 128 pairs up, or with sweep_mode = pair (here: few pairs, so that the oracle
    finishes); with 32-bit and with 64-bit keys; through the one-trip call (max_hits <= 8) and the three-trip one; against the
    three-matrix path (sweep_ev = 0 does not take wide rows) and the oracle (smith_waterman.c:137-277)."""
    sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    osc = osc_of(sc)
    n = 6 if wide_keys else 12
    batch = uniform(n, la, lb, seed=1300 + la)
    opts(pack16=pack16, sweep_mode="pair")
    few = ctx.sw_batch(batch, sc, 24, max_hits=3)
    launched = ctx.last_call()
    assert ("fill_sw_dirs_x2" if pack16 else "fill_sw_dirs") in launched and "sweep_dirs" in launched, launched
    many = ctx.sw_batch(batch, sc, 24, max_hits=40)
    assert "sweep_dirs" in ctx.last_call(), ctx.last_call()
    opts(sweep_ev=0)
    ref = ctx.sw_batch(batch, sc, 24, max_hits=40)
    assert ("sweep_dirs" in ctx.last_call()) == (la + 1 <= 512), ctx.last_call()
    assert many == ref
    for p in range(batch.n_pairs):
        rc, want = O.oracle_sw(osc, batch.seq_a(p), batch.seq_b(p), 24, 40)
        assert rc == 0 and many[p] == want and few[p] == want[:3], (la, lb, p, len(want))


@pytest.mark.parametrize("la,n_min", [(700, 128), (767, 128), (768, 640), (900, 640)])
def test_sw_multi_hit_wide_rows_by_batch_size(ctx, la, n_min):
    """The same path as the default from 128 pairs up for rows of 513 .. 768 columns, from 640 for 769 .. 1 024 (below that few wide
    pairs go to the strip sweep as before: tools/sw_wide_few.py); hit lists against the oracle."""
    sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    osc = osc_of(sc)
    batch = uniform(n_min, la, 60, seed=77 + la)
    res = ctx.sw_batch(batch, sc, 30, max_hits=4)
    launched = ctx.last_call()
    assert "fill_sw_dirs" in launched and "sweep_dirs" in launched, launched   # (packed two per wave from 1 025 pairs up)
    for p in range(0, n_min, 3):
        rc, want = O.oracle_sw(osc, batch.seq_a(p), batch.seq_b(p), 30, 4)
        assert rc == 0 and res[p] == want, p
    small = uniform(n_min - 1, la, 60, seed=77 + la)
    ctx.sw_batch(small, sc, 30, max_hits=4)
    assert "sweep_dirs" not in ctx.last_call(), ctx.last_call()


@pytest.mark.parametrize("kind,la,lb,n_min", [("blosum", 60, 50, 128), ("dna", 100, 40, 1536), ("dna", 600, 30, 1024)])
def test_sw_best_hit_packed_fill_by_batch_size(ctx, kind, la, lb, n_min):
    """From how many pairs of one shape seqalign_sw_batch(max_hits = 1) takes the packed best-hit fill (sa_batch_sw.hip: it competes with
    three matrices, not with a one-pair direction fill): 128 on table scorings, 1 536 on match / mismatch, 1 024 there for rows over 512
    columns (tools/sw_best_few.py); either way the oracle's best hit (smith_waterman.c:137-277)."""
    if kind == "blosum":
        sc = S.make_scoring({"preset": "BLOSUM62"})
        batch = uniform(n_min, la, lb, seed=31, alpha=b"ARNDCQEGHILKMFPSTWYV")
    else:
        sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
        batch = uniform(n_min, la, lb, seed=32 + la)
    osc = osc_of(sc)
    res = ctx.sw_batch(batch, sc, 12, max_hits=1)
    assert "fill_sw_best_x2" in ctx.last_call() or "fill_sw_best_x4" in ctx.last_call(), ctx.last_call()
    for p in range(0, n_min, 7):
        rc, want = O.oracle_sw(osc, batch.seq_a(p), batch.seq_b(p), 12, 1)
        assert rc == 0 and res[p] == want, p
    fewer = W.from_pairs([(batch.seq_a(p), batch.seq_b(p)) for p in range(n_min - 1)])
    res2 = ctx.sw_batch(fewer, sc, 12, max_hits=1)
    assert "fill_sw_best_x2" not in ctx.last_call() and "fill_sw_best_x4" not in ctx.last_call(), ctx.last_call()
    assert res2 == res[:n_min - 1]


def test_packed_fills_from_1025_pairs_uniform_2048_ragged(ctx):
    """seqalign_nw_batch: one shape -- two pairs per wave from the 1 025th pair on (every one-pair wave has a SIMD to itself up to 1 024);
    ragged -- bucketed from 2 048 (tools/pack_by_batch_size.py, ragged_bench.py); the oracle's strings for a sample."""
    sc = S.make_scoring({"preset": "default"})
    osc = osc_of(sc)
    for n, packed in ((1024, False), (1025, True)):
        batch = uniform(n, 90, 40, seed=5 + n)
        res = ctx.nw_batch(batch, sc)
        assert any(k in ctx.last_call() for k in ("fill_nw_dirs_x2", "fill_nw_dirs_x4")) == packed, (n, ctx.last_call())
        for p in range(0, n, 41):
            rc, s_, ra, rb = O.oracle_nw(osc, batch.seq_a(p), batch.seq_b(p))
            assert rc == 0 and res[p] == (s_, ra, rb), (n, p)
    for n, packed in ((2047, False), (2048, True)):
        base = uniform(n, 90, 40, seed=9)
        batch = W.from_pairs([(base.seq_a(p)[:60 + p % 31], base.seq_b(p)[:20 + p % 21]) for p in range(n)])
        res = ctx.nw_batch(batch, sc)
        assert any(k in ctx.last_call() for k in ("fill_nw_dirs_x2", "fill_nw_dirs_x4")) == packed, (n, ctx.last_call())
        for p in range(0, n, 97):
            rc, s_, ra, rb = O.oracle_nw(osc, batch.seq_a(p), batch.seq_b(p))
            assert rc == 0 and res[p] == (s_, ra, rb), (n, p)


@pytest.mark.parametrize("la,n,expect_dirs", [(512, 6, True), (700, 6, True), (767, 6, True), (768, 6, False), (1023, 6, False),
                                              (768, 384, True), (1023, 384, True), (1023, 383, False), (1024, 384, False), (1500, 6, False)])
def test_nw_direction_fill_takes_rows_up_to_1024_columns(ctx, la, n, expect_dirs):
    """Round 5: seqalign_nw_batch writes one byte of directions per cell for rows up to 1 024 columns (len_a <= 1 023; 12 / 16
    columns per lane, one pair per wave; rows over 768 columns from 384 pairs up: fewer are done sooner by the three-matrix fills,
    which put several waves on a pair) -- beyond that the three matrices; either way the oracle's strings
    (needleman_wunsch.c:34-146)."""
    sc = S.make_scoring({"preset": "default"})
    batch = uniform(n, la, 300 if n < 100 else 24, seed=900 + la)
    res = ctx.nw_batch(batch, sc)
    launched = ctx.last_call()
    assert any(k.startswith("fill_nw_dirs") for k in launched) == expect_dirs, launched
    osc = osc_of(sc)
    for p in range(batch.n_pairs):
        rc, s_, ra, rb = O.oracle_nw(osc, batch.seq_a(p), batch.seq_b(p))
        assert rc == 0 and res[p] == (s_, ra, rb), (la, p)


def test_nw_batch_reports_its_kernels(ctx, opts):
    sc = S.make_scoring({"preset": "default"})
    batch = uniform(64, 150, 150, 1)
    opts(pack16=2)
    ctx.nw_batch(batch, sc)
    assert ctx.last_call() == {"fill_nw_dirs_x2": (1, 64), "walk_moves_tile": (1, 64)}
    opts(pack16=1)          # 64 pairs: below the size from which packing pays
    ctx.nw_batch(batch, sc)
    assert ctx.last_call() == {"fill_nw_dirs": (1, 64), "walk_moves_tile": (1, 64)}
    opts(pack16=2, trace_kernel="lane")
    ctx.nw_batch(batch, sc)
    assert ctx.last_call() == {"fill_nw_dirs_x2": (1, 64), "walk_moves_lane": (1, 64)}
    opts(trace_kernel="auto", nw_moves=0)     # strings home instead of moves
    ctx.nw_batch(batch, sc)
    assert ctx.last_call() == {"fill_nw_dirs_x2": (1, 64), "walk_dirs_tile": (1, 64)}
    opts(nw_moves=1, nw_dirs=0)               # three matrices
    ctx.nw_batch(batch, sc)
    assert ctx.last_call() == {"fill_stream": (1, 64), "walk_wave": (1, 64)}
    opts(nw_dirs=1, subbatches=3)
    ctx.nw_batch(batch, sc)
    got = ctx.last_call()
    assert got["fill_nw_dirs_x2"][1] == 64 and got["fill_nw_dirs_x2"][0] in (2, 3) and got["walk_moves_tile"][1] == 64
    # a scoring with flags is outside the direction fills' domain
    opts(subbatches=0)
    ctx.nw_batch(batch, S.make_scoring({"init": [1, -2, -4, -1, 1, 1, 0, 0, 0, 0]}))
    assert set(ctx.last_call()) == {"fill_stream", "walk_wave"}
    # at a size where the packed kernel is the library's own choice
    big = uniform(2304, 100, 120, 2)
    opts(pack16=1)
    ctx.nw_batch(big, sc)
    assert ctx.last_call() == {"fill_nw_dirs_x2": (1, 2304), "walk_moves_tile": (1, 2304)}


def test_mostly_one_shape_reports_both_kinds_of_waves(ctx, opts):
    sc = S.make_scoring({"preset": "default"})
    rng = W.Rng(77)
    pairs = []
    for k in range(300):
        la, lb = (100, 100) if k % 5 else (int(rng.below(150, 1)[0]), int(rng.below(150, 1)[0]))
        pairs.append((bytes(b"ACGT"[i] for i in rng.below(4, la)), bytes(b"ACGT"[i] for i in rng.below(4, lb))))
    batch = W.from_pairs(pairs)
    n_modal = sum(1 for a, b in pairs if (len(a), len(b)) == (100, 100))
    opts(pack16=2)
    ctx.nw_batch(batch, sc)
    got = ctx.last_call()
    assert got["fill_nw_dirs_x2"][1] == n_modal and got["fill_nw_dirs"][1] == 300 - n_modal
    opts(pack16=0)
    ctx.nw_batch(batch, sc)
    assert ctx.last_call()["fill_nw_dirs"] == (1, 300) and "fill_nw_dirs_x2" not in ctx.last_call()


def test_sw_batch_reports_its_kernels(ctx, opts):
    sc = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    batch = uniform(48, 120, 200, 3)
    opts(pack16=2)
    ctx.sw_batch(batch, sc, 20, max_hits=4)
    got = ctx.last_call()
    assert got["fill_sw_dirs_x2"] == (1, 48) and "fill_sw_dirs" not in got and "fill_stream" not in got
    # the hit walks are launched before the counts are known: max_hits slots per pair, moves home
    assert got["sweep_dirs"] == (1, 48) and got["walk_moves_tile"] == (1, 4 * 48)
    opts(nw_moves=0)          # three trips: counts -> walker lists -> strings
    ctx.sw_batch(batch, sc, 20, max_hits=4)
    got = ctx.last_call()
    assert got["sweep_dirs"] == (1, 48) and got["walk_dirs_tile"][0] == 1 and got["walk_dirs_tile"][1] <= 4 * 48
    opts(nw_moves=1)
    ctx.sw_batch(batch, sc, 20, max_hits=1)
    got = ctx.last_call()
    assert got["fill_sw_best_x2"] == (1, 48) and "sw_reduce" not in got and got["walk_moves_tile"] == (1, 48)
    opts(pack16=0)
    ctx.sw_batch(batch, sc, 20, max_hits=4)
    got = ctx.last_call()
    assert got["fill_sw_dirs"] == (1, 48) and "fill_sw_dirs_x2" not in got
    opts(sweep_dirs=0)
    ctx.sw_batch(batch, sc, 20, max_hits=4)
    got = ctx.last_call()
    assert "fill_stream" in got and "sweep_regs" in got and not any("dirs" in k for k in got)


def gap_rich(n, la, lb, seed, alpha=b"ACGT"):
    """n pairs of one shape: seq_b = seq_a with ~8 % deletions, ~8 % insertions, ~8 % substitutions, cut / padded to lb"""
    rng = W.Rng(seed)
    pairs = []
    for _ in range(n):
        a = bytes(alpha[v] for v in rng.below(len(alpha), la))
        u = rng.unit(3 * la + 3).reshape(-1, 3)
        r = rng.below(len(alpha), 2 * la + lb + 2)
        b = bytearray()
        for i, ch in enumerate(a):
            if u[i][0] < 0.08:
                continue
            if u[i][1] < 0.08:
                b.append(alpha[r[2 * i]])
            b.append(alpha[r[2 * i + 1]] if u[i][2] < 0.08 else ch)
        b = (bytes(b) + bytes(alpha[v] for v in r[2 * la:]))[:lb]
        pairs.append((a, b))
    return W.from_pairs(pairs), pairs


@pytest.mark.parametrize("la,lb", [(150, 150), (31, 40), (32, 33), (95, 120), (96, 64), (159, 70), (160, 35), (191, 150), (1, 1), (5, 0)])
def test_four_pairs_per_wave(ctx, opts, la, lb):
    """The packed fills with FOUR pairs per wave (sa_fill_dirs_x2.hip, LANES = 32: lanes 0-31 one couple of pairs in the 16-bit
    halves, lanes 32-63 another) -- seqalign_nw_batch and the SW best hit, gap-rich pairs, pair counts that leave the last wave
    with one, two or three pairs, shapes at the edges of the columns-per-lane classes (32 k and 32 k - 1 columns).  Strings and
    hits equal the two-pairs-per-wave form's and the oracle's (the arithmetic: src/alignment.c:101-155), and the kernel that
    ran is the one asked for."""
    sc_nw, sc_sw = S.make_scoring({"preset": "default"}), S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    o_nw, o_sw = osc_of(sc_nw), osc_of(sc_sw)
    for n in (1, 2, 3, 4, 5, 7, 42):
        batch, pairs = gap_rich(n, la, lb, 1000 * la + lb + n)
        opts(pack16=2, quad=2)
        nw4 = ctx.nw_batch(batch, sc_nw)
        assert ctx.last_call()["fill_nw_dirs_x4"] == (1, n) and "fill_nw_dirs_x2" not in ctx.last_call()
        sw4 = ctx.sw_batch(batch, sc_sw, 8, max_hits=1)
        assert ctx.last_call()["fill_sw_best_x4"] == (1, n) and "fill_sw_best_x2" not in ctx.last_call()
        opts(quad=1)
        nw2 = ctx.nw_batch(batch, sc_nw)
        assert ctx.last_call()["fill_nw_dirs_x2"] == (1, n) and "fill_nw_dirs_x4" not in ctx.last_call()
        sw2 = ctx.sw_batch(batch, sc_sw, 8, max_hits=1)
        assert "fill_sw_best_x2" in ctx.last_call()
        assert nw4 == nw2 and sw4 == sw2
        for p, (a, b) in enumerate(pairs):
            rc, s_, ra, rb = O.oracle_nw(o_nw, a, b)
            assert rc == 0 and nw4[p] == (s_, ra, rb), (n, p)
            rc, want = O.oracle_sw(o_sw, a, b, 8, 1)
            assert rc == 0 and sw4[p] == want, (n, p)


def test_four_pairs_per_wave_is_chosen_by_size_and_shape(ctx, opts):
    """quad = 0 (the default): NW and the SW best hit from 4 097 pairs of one shape (whole rounds + a short rest two per wave in the same grid), rows up to 192 columns; ragged
    chunks (a pair list) and the multi-hit fill stay two per wave; a substitution table (BLOSUM62) goes four per wave too."""
    sc_nw, sc_sw = S.make_scoring({"preset": "default"}), S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    opts(pack16=1, quad=0)
    ctx.nw_batch(uniform(4096, 60, 50, 1), sc_nw)            # exactly one round of four-per-wave waves: two per wave
    assert "fill_nw_dirs_x2" in ctx.last_call() and "fill_nw_dirs_x4" not in ctx.last_call()
    big = uniform(8192, 60, 50, 2)
    got = ctx.nw_batch(big, sc_nw)
    assert ctx.last_call()["fill_nw_dirs_x4"] == (1, 8192)
    opts(quad=1)
    assert got == ctx.nw_batch(big, sc_nw)
    opts(quad=0)
    # a last round of four-per-wave waves that would be less than half full goes two per wave, in the same grid (gap-rich pairs)
    mixed, pairs = gap_rich(4096 + 101, 70, 64, 77)
    got = ctx.nw_batch(mixed, sc_nw)
    info = ctx.last_call()
    assert info["fill_nw_dirs_x4"] == (1, 4096) and info["fill_nw_dirs_x2"] == (1, 101), info
    opts(quad=1)
    assert got == ctx.nw_batch(mixed, sc_nw)
    o_nw = osc_of(sc_nw)
    for p in list(range(0, 4096, 257)) + list(range(4090, 4197)):
        rc, s_, ra, rb = O.oracle_nw(o_nw, *pairs[p])
        assert rc == 0 and got[p] == (s_, ra, rb), p
    opts(quad=0)
    # ... the same with a substitution table (BLOSUM62: the table in LDS behind rings of two sizes) and a scoring with gap_extend 0
    for spec, alpha in (({"preset": "BLOSUM62"}, b"ARNDCQEGHILKMFPSTWYV"), ({"init": [3, -2, -5, 0, 0, 0, 0, 0, 0, 0]}, b"ACGT")):
        scx = S.make_scoring(spec)
        mixed, pairs = gap_rich(4096 + 37, 45, 52, 78, alpha=alpha)
        got = ctx.nw_batch(mixed, scx)
        info = ctx.last_call()
        assert info["fill_nw_dirs_x4"] == (1, 4096) and info["fill_nw_dirs_x2"] == (1, 37), info
        ox = osc_of(scx)
        for p in list(range(0, 4096, 311)) + list(range(4092, 4133)):
            rc, s_, ra, rb = O.oracle_nw(ox, *pairs[p])
            assert rc == 0 and got[p] == (s_, ra, rb), (spec, p)
    ctx.nw_batch(uniform(4096 + 2049, 60, 50, 8), sc_nw)     # more than half a round left: four per wave throughout
    assert ctx.last_call()["fill_nw_dirs_x4"] == (1, 4096 + 2049) and "fill_nw_dirs_x2" not in ctx.last_call()
    ctx.nw_batch(uniform(8192, 192, 20, 3), sc_nw)           # 193 columns: seven per lane of a span -- two pairs per wave
    assert "fill_nw_dirs_x2" in ctx.last_call() and "fill_nw_dirs_x4" not in ctx.last_call()
    sw = uniform(16384, 40, 30, 4)
    got = ctx.sw_batch(sw, sc_sw, 10, max_hits=1)
    assert ctx.last_call()["fill_sw_best_x4"] == (1, 16384)
    opts(quad=1)
    assert got == ctx.sw_batch(sw, sc_sw, 10, max_hits=1)
    # the best-hit fill's mixed grid (round 6): whole rounds of four-per-wave waves + a rest of less than half a round two per wave,
    # from 4 097 pairs on -- BASELINE configs[2]'s 10 000 pairs are 8 192 + 1 808 -- with match / mismatch and with a table
    opts(quad=0, pack16=1)
    for spec, alpha, thr in (({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}, b"ACGT", 10), ({"preset": "BLOSUM62"}, b"ARNDCQEGHILKMFPSTWYV", 20)):
        scx = S.make_scoring(spec)
        mixed, pairs = gap_rich(8192 + 213, 38, 61, 79, alpha=alpha)
        got = ctx.sw_batch(mixed, scx, thr, max_hits=1)
        info = ctx.last_call()
        assert info["fill_sw_best_x4"] == (1, 8192) and info["fill_sw_best_x2"] == (1, 213), info
        opts(quad=1)
        assert got == ctx.sw_batch(mixed, scx, thr, max_hits=1)
        assert "fill_sw_best_x4" not in ctx.last_call()
        opts(quad=0)
        ox = osc_of(scx)
        for p in list(range(0, 8192, 397)) + list(range(8186, 8405)):
            rc, want = O.oracle_sw(ox, *pairs[p], thr, 1)
            assert rc == 0 and got[p] == want, (spec, p)
    ctx.sw_batch(uniform(8192 + 2049, 40, 30, 9), sc_sw, 10, max_hits=1)   # more than half a round left: four per wave throughout
    assert ctx.last_call()["fill_sw_best_x4"] == (1, 8192 + 2049) and "fill_sw_best_x2" not in ctx.last_call()
    ctx.sw_batch(uniform(4096, 40, 30, 10), sc_sw, 10, max_hits=1)         # one whole round and nothing more: two per wave
    assert "fill_sw_best_x2" in ctx.last_call() and "fill_sw_best_x4" not in ctx.last_call()
    small, pairs = gap_rich(4096 + 300, 38, 61, 80)
    got = ctx.sw_batch(small, sc_sw, 10, max_hits=1)
    assert ctx.last_call()["fill_sw_best_x4"] == (1, 4096) and ctx.last_call()["fill_sw_best_x2"] == (1, 300)
    for p in list(range(0, 4096, 509)) + list(range(4090, 4396, 7)):
        rc, want = O.oracle_sw(osc_of(sc_sw), *pairs[p], 10, 1)
        assert rc == 0 and got[p] == want, p
    opts(quad=2, pack16=2)
    ctx.sw_batch(uniform(64, 40, 30, 5), sc_sw, 10, max_hits=4)      # the multi-hit fill has no such form
    assert "fill_sw_dirs_x2" in ctx.last_call()
    prot = uniform(67, 90, 110, 6, alpha=b"ARNDCQEGHILKMFPSTWYV", related=1.0)
    bl = S.make_scoring({"preset": "BLOSUM62"})
    four = (ctx.nw_batch(prot, bl), ctx.last_call())
    four_sw = (ctx.sw_batch(prot, bl, 15, max_hits=1), ctx.last_call())
    assert "fill_nw_dirs_x4" in four[1] and "fill_sw_best_x4" in four_sw[1]
    o_bl = osc_of(bl)
    for p in range(prot.n_pairs):
        rc, s_, ra, rb = O.oracle_nw(o_bl, prot.seq_a(p), prot.seq_b(p))
        assert rc == 0 and four[0][p] == (s_, ra, rb), p
        rc, want = O.oracle_sw(o_bl, prot.seq_a(p), prot.seq_b(p), 15, 1)
        assert rc == 0 and four_sw[0][p] == want, p


def test_ragged_batches_take_the_packed_fills(ctx, opts):
    """"for variable-length batches, sort / bucket by W x H first" -- chunks of reads of many lengths: the pairs of
    equal shape are paired up on the host and go two per wave (NW: the others one per wave in the same grid; SW: a pair without
    a partner has a wave to itself).  Results equal the one-pair-per-wave path's and the oracle's; the packed kernels ran."""
    rng = W.Rng(515)
    rnd = lambda n: bytes(b"ACGT"[i] for i in rng.below(4, n)) if n else b""
    pairs = []
    for k in range(360):
        la, lb = int(40 + rng.below(12, 1)[0]), int(70 + rng.below(9, 1)[0])       # 12 x 9 shapes, ~3 pairs of each
        a = rnd(la)
        b = (rnd(lb // 3) + a[5:la - 5] + rnd(lb))[:lb] if k % 2 else rnd(lb)       # planted / unrelated
        pairs.append((a, b))
    pairs += [(b"", b"ACGT"), (rnd(51), b"")]
    batch = W.from_pairs(pairs)
    shapes = {}
    for a, b in pairs:
        shapes[(len(a), len(b))] = shapes.get((len(a), len(b)), 0) + 1
    n_single = sum(v % 2 for v in shapes.values())
    sc_nw, sc_sw = S.make_scoring({"preset": "default"}), S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    opts(pack16=2)
    nw = ctx.nw_batch(batch, sc_nw)
    info = ctx.last_call()
    assert info["fill_nw_dirs_x2"][1] == len(pairs) - n_single and info["fill_nw_dirs"][1] == n_single, info
    sw4 = ctx.sw_batch(batch, sc_sw, 12, max_hits=4, hit_cap=1 << 16)
    info = ctx.last_call()
    assert info["fill_sw_dirs_x2"] == (1, 2 * ((len(pairs) - n_single) // 2 + n_single)) and "fill_sw_dirs" not in info, info
    sw1 = ctx.sw_batch(batch, sc_sw, 12, max_hits=1, hit_cap=1 << 16)
    info = ctx.last_call()
    assert "fill_sw_best_x2" in info and "fill_stream" not in info, info
    opts(pack16=0)
    assert nw == ctx.nw_batch(batch, sc_nw)
    assert sw4 == ctx.sw_batch(batch, sc_sw, 12, max_hits=4, hit_cap=1 << 16)
    assert sw1 == ctx.sw_batch(batch, sc_sw, 12, max_hits=1, hit_cap=1 << 16)
    o_nw, o_sw = osc_of(sc_nw), osc_of(sc_sw)
    for p, (a, b) in enumerate(pairs):
        rc, s_, ra, rb = O.oracle_nw(o_nw, a, b)
        assert rc == 0 and nw[p] == (s_, ra, rb), p
        rc, want = O.oracle_sw(o_sw, a, b, 12, 4)
        assert rc == 0 and sw4[p] == want and sw1[p] == want[:1], p


def test_arena_placement_when_the_device_is_half_full(ctx, opts):
    """seqalign_arenas_alloc with 150 GB of the device held by somebody else (the three matrices of src/alignment.c:183-190 for
    a C2-sized batch): the walk looks around in at most 60 % of what is FREE, says what it did, and still finds memory
    where the three arenas do not all disturb each other (probe ratio ~0.80 when they do; >= 0.95 -- or a walk that used its
    whole budget -- asked here, measured values in profiles/r04/r04_placement_pressure.txt); everything comes back afterwards."""
    import ctypes as C
    import torch
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(0)[0]
    if free0 < 200 * 10**9:
        pytest.skip("needs an (almost) empty 288 GB device")
    hog = torch.empty(150 * 10**9, dtype=torch.uint8, device="cuda")
    free1 = torch.cuda.mem_get_info(0)[0]
    nbytes = 4 * 10000 * 22801 // 4096 * 4096 + 4096
    for scan in (160, 24):
        opts(arena_scan_gib=scan)
        ptrs, q = (C.c_void_p * 3)(), C.c_float(-1)
        assert S.lib().seqalign_arenas_alloc(ctx._h, C.c_uint64(nbytes), ptrs, C.byref(q)) == 0
        info = S.ArenaInfo()
        assert S.lib().seqalign_arenas_info(ctx._h, ptrs, C.byref(info)) == 0
        d = info.as_dict()
        assert d["vmm"] and d["scanned_gib"] * 2**30 <= 0.6 * free1 + 3 * nbytes + (1 << 30), d
        assert d["scanned_gib"] <= scan + 3 * nbytes / 2**30 + 1, d
        # a placement below "another class" (~1.0; all three arenas in one class: ~0.80) only after the walk has used what it may:
        # which memory the driver hands out differs from box to box (profiles/r04/r04_placement_pressure.txt: 0.996 / 1.051)
        budget_gib = min(scan + 3 * nbytes / 2**30, 0.6 * free1 / 2**30)
        assert q.value >= 0.95 or d["scanned_gib"] >= budget_gib - 4.5, d
        assert q.value > 0.6, d
        assert S.lib().seqalign_arenas_free(ctx._h, ptrs) == 0
    del hog
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ctx.pool_trim(0)          # (what the walks left in the process's chunk pool)
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (256 << 20)


def test_scratch_comes_from_the_chunk_pool(opts):
    """Round 5: the chunks a placement walk created and did not use stay with the process (<= arena_keep_gib), and a context's
    large scratch buffers -- here seqalign_nw_batch's direction bytes for 30 000 pairs, 0.7 GB -- are mapped from them instead of
    from freshly released VRAM the driver has to clear first (round 4: the first call after a full walk waited 3.7 s).
    Results are the oracle's whatever memory holds the bytes; the pool shrinks by what the buffers took, gets it back when the
    context goes, and is emptied with the device's last context."""
    import ctypes as C
    import torch
    torch.cuda.empty_cache()
    sc = S.make_scoring({"preset": "default"})
    osc = osc_of(sc)
    batch = W.dna_nw_150(30000, seed=11)
    with S.Context(0) as keeper:              # keeps the pool alive while `c` comes and goes
        keeper.pool_trim(0)                   # (whatever earlier tests of this process left; also resets the pool's cap)
        keeper.set_option("arena_keep_gib", 4)
        keeper.set_option("arena_scan_gib", 12)
        keeper.set_option("arena_quality", 1.49)      # unreachable: the walk uses its whole 12 GiB
        ptrs, q = (C.c_void_p * 3)(), C.c_float(-1)
        assert S.lib().seqalign_arenas_alloc(keeper._h, C.c_uint64(600 << 20), ptrs, C.byref(q)) == 0
        info = S.ArenaInfo()
        assert S.lib().seqalign_arenas_info(keeper._h, ptrs, C.byref(info)) == 0
        held0 = keeper.pool_trim()
        assert held0 == 4 << 30 and abs(info.as_dict()["kept_gib"] - 4.0) < 0.01
        with S.Context(0) as c:
            res = c.nw_batch(batch, sc)
            assert any(k.startswith("fill_nw_dirs") for k in c.last_call())
            held1 = c.pool_trim()
            assert held1 <= held0 - (512 << 20)            # at least the direction bytes came from the pool
            for p in range(0, batch.n_pairs, 997):
                rc, s_, ra, rb = O.oracle_nw(osc, batch.seq_a(p), batch.seq_b(p))
                assert rc == 0 and res[p] == (s_, ra, rb), p
        assert keeper.pool_trim() == held0                 # the buffers' chunks are back
        assert S.lib().seqalign_arenas_free(keeper._h, ptrs) == 0
        assert keeper.pool_trim() == held0                 # full: the arenas' own chunks went to the driver
        assert keeper.pool_trim(1 << 30) == 1 << 30
        # (the pool is emptied when the device's LAST context goes; this process holds others -- the module's fixture, the
        # DeviceBatch arena context -- so here it is emptied by hand)
        assert keeper.pool_trim(0) == 0


@pytest.mark.parametrize("walker", ["wave", "lane", "wave-1", "wave-4", "wave-8"])
def test_long_walks_cross_move_blocks(ctx, opts, walker):
    """Walks of several thousand columns: the wave-per-walk walker keeps 64 words (2 048 columns) of each plane in its lanes and
    stores whole blocks; here walks of up to ~5 000 columns cross two and more block boundaries, end exactly on one (2 048 and
    4 096 walked columns: identical sequences) and one column past it.  needleman_wunsch.c:82-145 via the oracle."""
    # ("wave": one wave per walk, 64 words of each plane in its lanes; "wave-4": four walks per wave in lockstep, 16 words per walk --
    #  blocks of 512 columns; "wave-8": eight per wave, blocks of 256 columns)
    walker, group = (walker.split("-") + ["0"])[:2]
    opts(trace_kernel=walker, pack16=0, walk_group=int(group))
    rng = W.Rng(808)
    rnd = lambda n: bytes(b"ACGT"[i] for i in rng.below(4, n)) if n else b""
    sc = S.make_scoring({"preset": "default"})
    osc = osc_of(sc)
    pairs = []
    for la, lb in ((300, 4700), (511, 2048), (64, 2047), (2, 4100)):
        a = rnd(la)
        pairs.append((a, (a * (lb // max(la, 1) + 1))[:lb]))     # b = a repeated: long runs of matches and long gaps
        pairs.append((a, rnd(lb)))
    same = rnd(2048)
    pairs += [(same[:511], same[:511]), (rnd(100), b""), (b"", rnd(3000))]
    batch = W.from_pairs(pairs)
    got = ctx.nw_batch(batch, sc)
    assert "walk_moves_" + ("tile" if walker == "wave" else "lane") in ctx.last_call()
    for p, (a, b) in enumerate(pairs):
        rc, s_, ra, rb = O.oracle_nw(osc, a, b)
        assert rc == 0 and got[p] == (s_, ra, rb), (walker, p, len(a), len(b))
    # local hits that long: a read against a window that contains it several times over
    sw = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    osw = osc_of(sw)
    a = rnd(500)
    long_pairs = [(a, rnd(300) + a + rnd(200) + a[:400] + rnd(100)), (a, a), (a[:70], rnd(2500) + a[:70])]
    lb = W.from_pairs(long_pairs)
    for max_hits in (1, 3):
        hits = ctx.sw_batch(lb, sw, 40, max_hits=max_hits, hit_cap=64)
        for p, (x, y) in enumerate(long_pairs):
            rc, want = O.oracle_sw(osw, x, y, 40, max_hits)
            assert rc == 0 and hits[p] == want, (walker, max_hits, p)


def test_options_are_parsed_strictly_and_put_back(ctx):
    """seqalign_ctx_set_option: a number is an integer and nothing else, a switch takes 1 / 0, true / false, on / off, yes / no --
    garbage is refused (SEQALIGN_E_ARG, nothing changes), never read as 0; Context.options() puts back what was in force before,
    whatever set it (seqalign_ctx_get_option), not the library's defaults."""
    for key, bad in (("cpl", "abc"), ("cpl", "3x"), ("cpl", ""), ("subbatches", "1e3"), ("nw_dirs", "maybe"), ("pack16", "7"),
                     ("arena_quality", "fast"), ("zero_copy", "5"), ("no_such_option", "1")):
        before = None if key == "no_such_option" else ctx.get_option(key)
        with pytest.raises(S.SeqAlignError) as err:
            ctx.set_option(key, bad)
        assert err.value.code == S.E_ARG
        if before is not None:
            assert ctx.get_option(key) == before
    for text, want in (("yes", "1"), ("off", "0"), ("TRUE", "1"), ("no", "0"), (True, "1"), (False, "0"), (1, "1")):
        ctx.set_option("walk_overlap", text)
        assert ctx.get_option("walk_overlap") == want
    ctx.set_option("walk_overlap", 1)
    ctx.set_option("subbatches", 5)                 # "configured earlier" (as SEQALIGN_SUBBATCHES=5 at context creation would)
    ctx.set_option("zero_copy", 2)
    with ctx.options(subbatches=2, zero_copy="auto", traceback="host"):
        assert (ctx.get_option("subbatches"), ctx.get_option("zero_copy"), ctx.get_option("traceback")) == ("2", "auto", "host")
    assert (ctx.get_option("subbatches"), ctx.get_option("zero_copy"), ctx.get_option("traceback")) == ("5", "2", "device")
    ctx.set_option("subbatches", 0)
    ctx.set_option("zero_copy", "auto")


def test_device_level_calls_report_too(ctx):
    sc = S.make_scoring({"preset": "default"})
    batch = uniform(32, 90, 90, 4)
    h = ctx.upload_scoring(sc, 0)
    db = S.DeviceBatch(batch, 0)
    for kernel, name in ((S.KERNEL_STREAM, "fill_stream"), (S.KERNEL_ROWSCAN, "fill_rowscan"), (S.KERNEL_WAVEFRONT, "fill_wavefront")):
        db.fill(ctx, h, kernel)
        assert ctx.last_call() == {name: (1, 32)}
    ctx.release_scoring(h)


# --------------------------------------------------------- the int16 bound at its edge ---

def x2_bound(la, lb, match, mismatch, gap_open, gap_extend):
    """sa_x2_scores_fit (csrc/sa_fill_dirs_x2.hip): every score of a la x lb pair stays inside int16 when this is <= 30 000."""
    pen = max(abs(match), abs(mismatch), abs(gap_open + gap_extend), abs(gap_extend), abs(gap_open) + abs(gap_extend))
    return (la + lb + 2) * pen + (la + 1) * abs(gap_extend)


def x2_bound_nw(la, lb, match, mismatch, gap_open, gap_extend):
    """sa_domain_nw_x2_scores_fit (csrc/sa_kernels.h): the packed NW fills keep their values de-trended by (column + row) x gap_extend."""
    pen = max(abs(match), abs(mismatch), abs(gap_open + gap_extend), abs(gap_extend), abs(gap_open) + abs(gap_extend))
    return (la + lb + 2) * (pen + abs(gap_extend))


def edge_cases():
    """(la, lb, scoring numbers, bound) just inside and just outside the bound, at the widest admitted row and at read size."""
    out = []
    for la, lb in ((511, 300), (150, 150), (300, 300), (64, 511)):
        for ext in (-1, -7):
            inside = outside = None
            for bound_of in (x2_bound, x2_bound_nw):     # the edge of the SW fills' bound, and of the NW fills'
                inside = outside = None
                for pen in range(8, 200):
                    b = bound_of(la, lb, pen, -pen, -(pen + ext) if pen + ext > 0 else 0, ext)
                    if b <= 30000 and b >= 29400:
                        inside = (pen, b)
                    if b > 30000 and outside is None:
                        outside = (pen, b)
                if inside and (la, lb, inside[0], ext) not in [(x[0], x[1], x[2], x[3]) for x in out]:
                    out.append((la, lb, inside[0], ext, inside[1], True))
                if outside and (la, lb, outside[0], ext) not in [(x[0], x[1], x[2], x[3]) for x in out]:
                    out.append((la, lb, outside[0], ext, outside[1], False))
    return out


def worst_case_pairs(la, lb, seed):
    """Identical sequences (the highest scores), nothing in common (the lowest: all mismatches or all gaps), one long gap,
    and random ones -- all of ONE shape."""
    rng = W.Rng(seed)
    rnd = lambda n: bytes(b"ACGT"[i] for i in rng.below(4, n)) if n else b""
    base = rnd(max(la, lb))
    pairs = [(base[:la], base[:lb]), (b"A" * la, b"C" * lb), (b"A" * la, b"A" * lb), (base[:la], base[::-1][:lb]),
             (rnd(la), rnd(lb)), (b"AC" * (la // 2) + b"A" * (la % 2), b"CA" * (lb // 2) + b"C" * (lb % 2)),
             ((b"A" * (la // 2) + base)[:la], (base + b"A" * lb)[:lb])]
    return pairs


@pytest.mark.parametrize("la,lb,pen,ext,bound,packs", edge_cases(), ids=lambda v: str(v))
def test_packed_fills_at_the_edge_of_int16(ctx, opts, la, lb, pen, ext, bound, packs):
    """Scorings whose admission bound evaluates to just under / just over 30 000 on worst-case inputs at that shape: packed
    (and equal to the oracle) / NOT packed (the 32-bit kernels take the chunk, same results).  Three score profiles per
    bound: match-heavy, mismatch-heavy, gap-heavy."""
    go = -(pen + ext) if pen + ext > 0 else 0            # gap_open + gap_extend = -pen: the first gap character costs `pen`
    assert (bound <= 30000) == packs and bound in (x2_bound(la, lb, pen, -pen, go, ext), x2_bound_nw(la, lb, pen, -pen, go, ext))
    profiles = [(pen, -pen, go, ext), (max(1, pen // 3), -pen, go, ext), (pen, -max(1, pen // 2), min(0, go // 4), ext)]
    opts(pack16=2)
    for match, mismatch, gap_open, gap_extend in profiles:
        spec = {"init": [match, mismatch, gap_open, gap_extend, 0, 0, 0, 0, 0, 0], "wildcards": []}
        fits = x2_bound(la, lb, match, mismatch, gap_open, gap_extend) <= 30000
        fits_nw = x2_bound_nw(la, lb, match, mismatch, gap_open, gap_extend) <= 30000
        sc = S.make_scoring(spec)
        osc = osc_of(sc)
        pairs = worst_case_pairs(la, lb, 31 * la + lb + pen)
        batch = W.from_pairs(pairs)
        got = ctx.nw_batch(batch, sc)
        info = ctx.last_call()
        assert ("fill_nw_dirs_x2" in info) == fits_nw, (spec, info)
        if not fits_nw:
            assert info.get("fill_nw_dirs", (0, 0))[1] == len(pairs), (spec, info)
        for p, (a, b) in enumerate(pairs):
            rc, s_, ra, rb = O.oracle_nw(osc, a, b)
            assert rc == 0 and got[p] == (s_, ra, rb), (spec, "nw", p)
        for max_hits in (1, 3):
            thr = max(1, match * 2)
            got_sw = ctx.sw_batch(batch, sc, thr, max_hits=max_hits, hit_cap=1 << 16)
            info = ctx.last_call()
            assert (("fill_sw_best_x2" if max_hits == 1 else "fill_sw_dirs_x2") in info) == fits, (spec, max_hits, info)
            for p, (a, b) in enumerate(pairs):
                rc, want = O.oracle_sw(osc, a, b, thr, max_hits)
                assert rc == 0 and got_sw[p] == want, (spec, "sw", max_hits, p)


# ------------------------------------------------------------------ one pair of more than 2^30 cells ---

def test_sweep_reads_the_right_rows_beyond_2_30_cells(ctx):
    """the one-word sweep formed a cell's byte offset as a 32-bit `cell * 4`, which wraps from cell 2^30 on --
    a 500 x 2.2 M pair (1.1 G cells: admitted, the limit is 2^31) read match_scores of the wrong rows for its last 57 000 rows and
    lost the hits there.  The row's base is now part of the 64-bit scalar address.  One such pair with the read planted three
    times -- once in the first rows, twice BEHIND cell 2^30 -- against the oracle run on windows of seq_b around the copies (the
    whole pair would need 13 GB of host matrices; with min_score 200 nothing outside the copies is a candidate)."""
    la, lb = 500, 2_200_000
    assert (la + 1) * (lb + 1) > 1 << 30 and (la + 1) * (lb + 1) < 1 << 31
    rng = W.Rng(6001)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    a = acgt[rng.below(4, la).astype(np.int64)].copy()
    b = acgt[rng.below(4, lb).astype(np.int64)].copy()
    starts = (100_000, 2_160_000, 2_190_000)        # (2^30 / 501 = row 2 143 237)
    for k, s in enumerate(starts):
        copy = a.copy()
        sub = rng.unit(la) < 0.03 * (k + 1)           # 3 %, 6 %, 9 % substitutions: three different scores
        copy[sub] = acgt[(np.searchsorted(acgt, copy[sub]) + 1 + rng.below(3, int(sub.sum())).astype(np.int64)) % 4]
        b[s:s + la] = copy
    batch = W._fixed_batch(a.reshape(1, la), b.reshape(1, lb))
    spec = {"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]}
    sc, osc = S.make_scoring(spec), O.build_scoring(spec, "oracle")
    got = ctx.sw_batch(batch, sc, 200, max_hits=64, hit_cap=4096)[0]
    ran = ctx.last_call()
    assert "sweep_dirs" in ran and "fill_sw_dirs" in ran, ran     # the direction-byte path: the one the offset was wrong in
    want = []
    for s in starts:
        lo, hi = s - 2000, s + la + 2000
        rc, hits = O.oracle_sw(osc, a.tobytes(), b[lo:hi].tobytes(), 200, 64)
        assert rc == 0 and hits
        for h in hits:
            assert h["pos_b"] > 500 and h["pos_b"] + h["len_b"] < hi - lo - 500     # the walk stayed inside the window
            want.append(dict(h, pos_b=h["pos_b"] + lo))
    # the reference's hit order (smith_waterman.c:71-86): score desc, end column asc, end cell index asc
    want.sort(key=lambda h: (-h["score"], h["pos_a"] + h["len_a"], h["pos_b"] + h["len_b"]))
    assert len({h["score"] for h in want[:3]}) == 3
    assert got == want, (len(got), len(want), [h["score"] for h in got[:5]], [h["score"] for h in want[:5]])
    assert sum(h["pos_b"] > 2_143_237 for h in got) >= 2


def test_large_chunks_take_tile_walks_on_the_local_byte(ctx, opts):
    """Round 6: with the direction byte's local form the tile walks are level with or ahead of the lane walkers at every batch size, so
    chunks of 24 576 pairs and more (round 4's crossover) take them too; dirs_local = 0 brings back the older byte and, from that size on,
    the lane walkers -- same alignments, same best hits, and a sample of both against the oracle."""
    sc_nw, sc_sw = S.make_scoring({"preset": "default"}), S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    batch, pairs = gap_rich(24576 + 900, 38, 45, 81)
    opts(dirs_local=1)
    nw1 = ctx.nw_batch(batch, sc_nw)
    assert "walk_moves_tile" in ctx.last_call() and "walk_moves_lane" not in ctx.last_call(), ctx.last_call()
    sw1 = ctx.sw_batch(batch, sc_sw, 10, max_hits=1)
    assert "walk_moves_tile" in ctx.last_call() and "walk_moves_lane" not in ctx.last_call(), ctx.last_call()
    opts(dirs_local=0)
    nw0 = ctx.nw_batch(batch, sc_nw)
    assert "walk_moves_lane" in ctx.last_call() and "walk_moves_tile" not in ctx.last_call(), ctx.last_call()
    sw0 = ctx.sw_batch(batch, sc_sw, 10, max_hits=1)
    assert "walk_moves_lane" in ctx.last_call(), ctx.last_call()
    assert nw1 == nw0 and sw1 == sw0
    o_nw, o_sw = osc_of(sc_nw), osc_of(sc_sw)
    for p in list(range(0, len(pairs), 997)) + list(range(24570, 24590)):
        rc, s_, ra, rb = O.oracle_nw(o_nw, *pairs[p])
        assert rc == 0 and nw1[p] == (s_, ra, rb), p
        rc, want = O.oracle_sw(o_sw, *pairs[p], 10, 1)
        assert rc == 0 and sw1[p] == want, p


# ------------------------------------------------------------------ direction bytes in blocks of 8 x 16 cells (round 6) ---

@pytest.mark.parametrize("form", ["x1", "x2", "x4", "mixed"])
def test_blocked_direction_bytes_at_the_blocks_edges(ctx, opts, form):
    """Round 6: where only walkers read them (seqalign_nw_batch, seqalign_sw_batch best hit; rows <= 512 columns) the direction bytes lie
    in blocks of 8 rows x 16 columns (csrc/sa_kernels.h).  Shapes whose rows / columns end exactly on, one short of and one past a block
    edge, the narrowest and the widest blocked rows (len_a + 1 = 512) and the first row-major one (513), through the one-pair fill, the
    packed fills with two and four pairs per wave and the mixed grid, every tile-walker form and the lane walker -- global alignments and
    best local hits against the oracle (needleman_wunsch.c:53-145, smith_waterman.c:165-277)."""
    rng = W.Rng(4242)
    rnd = lambda n: bytes(b"ACGT"[i] for i in rng.below(4, n)) if n else b""
    las = (0, 1, 14, 15, 16, 17, 30, 31, 32, 47, 63, 64, 150, 191, 255, 300, 510, 511, 512)
    lbs = (0, 1, 6, 7, 8, 9, 15, 16, 62, 63, 64, 65, 150, 301)
    sc = S.make_scoring({"preset": "default"})
    sw = S.make_scoring({"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]})
    osc, osw = osc_of(sc), osc_of(sw)
    if form == "x1":
        opts(pack16=0)
    elif form == "x2":
        opts(pack16=2, quad=1)
    elif form == "x4":
        opts(pack16=2, quad=2)
    else:
        opts(pack16=2)
    for la in las:
        if form == "x4" and la + 1 > 192:
            continue
        shapes = [(la, lb) for lb in lbs]
        pairs = []
        for la_, lb_ in shapes:
            a = rnd(la_)
            b = (a[: lb_] if lb_ <= la_ else a + rnd(lb_ - la_))          # related: long diagonal runs, then a tail of gaps
            for _ in range(6 if form != "x1" else 2):                      # several pairs of each shape: the packed fills pair them up
                pairs.append((a, bytes(b)))
                b = bytearray(b)
                if b: b[int(rng.below(len(b), 1)[0])] = b"ACGT"[int(rng.below(4, 1)[0])]
                b = bytes(b)
        if form in ("x2", "x4"):            # uniform chunks: one shape per call
            groups = {}
            for a, b in pairs: groups.setdefault((len(a), len(b)), []).append((a, b))
            batches = list(groups.values())
        else:
            batches = [pairs]
        # (dirs_local: the byte's LOCAL form -- a cell's own comparisons, the tile walkers resolve the state they arrive in,
        #  csrc/sa_kernels.h -- is what tile walks get by default; 0: the older form through the same fills and walkers)
        for walker, grp, local in (("wave", 0, 1), ("wave", 1, 1), ("wave", 8, 1), ("lane", 0, 1), ("wave", 0, 0), ("wave", 1, 0), ("wave", 0, 64), ("wave", 0, 32)):
            # (local = 32 / 64: the local form with the walker's tile edge forced, option walk_tile)
            # (the 64-byte-tile form also with the moves leaving as two pieces per walk instead of one run per wave: option walk_stage)
            opts(trace_kernel=walker, walk_group=grp, dirs_local=min(local, 1), walk_tile=local if local > 1 else 0, walk_stage=0 if local == 64 else 1)
            for bp in batches[:: (3 if (walker, grp, local) != ("wave", 0, 1) else 1)]:
                batch = W.from_pairs(bp)
                got = ctx.nw_batch(batch, sc)
                best = ctx.sw_batch(batch, sw, 4, max_hits=1)
                for p, (a, b) in enumerate(bp):
                    rc, s_, ra, rb = O.oracle_nw(osc, a, b)
                    assert rc == 0 and got[p] == (s_, ra, rb), (form, walker, grp, local, la, len(b), p)
                    rc, want = O.oracle_sw(osw, a, b, 4, 1)
                    assert rc == 0 and best[p] == want, (form, walker, grp, local, la, len(b), p)
