// sa_sw_sweep.hip -- Smith-Waterman multi-hit enumeration as ONE reverse sweep over the score matrices
// (SURVEY 8f-2).
//
// Reference semantics (src/smith_waterman.c:137-277): candidates (cells with match_scores >= min_score) are
// visited in (score desc, column asc, index asc) order; a candidate that is already marked is skipped, else it
// is walked back (alignment_reverse_move, alignment.c:244-350) marking every cell it stands on, the walk is
// abandoned -- its marks stay -- when it meets a marked cell, and a walk that reaches a score of 0 is a hit.
// A 150x1000 pair has ~15 000 candidates, a 300x300 BLOSUM62 pair ~50 000, nearly all in the plume of one real
// hit.  The procedure is sequential by definition; this file computes its result without running it.
//
// Give every candidate its rank in that order (the key below: ascending key = the reference's order).  Then:
//   * a cell is marked by the LOWEST-ranked walk that ever arrives at it: that walk gets there first in the
//     sequential order, nothing else can have marked the cell before (marks are only made by arrivals), and every
//     later arrival finds it marked and stops;
//   * so a walk arrives at the next cell of its route exactly when it WON the cell it stands on, and the arrivals
//     at a cell are: its own candidacy (if it is a candidate) and the winners of the three cells a backward move
//     can come from -- (x+1,y+1) if that winner stands there in MATCH, (x,y+1) in GAP_A, (x+1,y) in GAP_B;
//   * a winner whose state has score 0 is a hit (smith_waterman.c:187-199 marks the cell, then stops).
// The winner of a cell therefore depends only on cells below / right of it: one sweep over the rows from the
// bottom of the candidates' box upwards settles every cell, yields every hit of the pair (whatever max_hits is:
// the first max_hits in key order are the reference's), and touches each matrix value once -- no sorting of
// candidates, no visited bitmap, no iteration over walks.  (Checked against the sequential procedure by
// tests/test_gpu_parity.py and tools/fuzz_e2e.py; the argument in full: DESIGN.md 3.6.)
//
// One WAVE per pair; lane l owns columns l, l+64, ... (coalesced row loads, 12 B per cell, each row loaded once and
// kept for the row above it, the next row in flight while this one is worked on).  Per row and column slot the
// wave carries one 64-bit record: key << 4 | state at the NEXT cell << 2 | state here, of the walk that won the
// cell and moves on.  Arrivals from the row below are one DPP shift; arrivals along the row (GAP_B moves) make a
// right-to-left dependency, which is resolved by iterating the row until nothing changes -- walks rarely move
// sideways for more than a cell or two, so that is one or two rounds.  The predecessor of the winner's state is the
// traceback's own decision (reverse_move_t; for plain scorings the same three equality tests on 32-bit values).
// Rows without candidates and without live walks cost their loads and a ballot.
// Pairs wider than SA_SWEEP_SEGMENT columns are swept in column segments, right to left within a row, with the
// records of the last two rows in HBM (SaSweepParams::rows) and segments nothing can reach skipped.
//
// Bound: HBM reads of the box's rows (12 B per cell) for many pairs; the latency of one row's work x the rows of
// the box for few pairs.
#include <algorithm>

#include "sa_trace_common.hpp"

namespace sa {

typedef unsigned long long rec_t;
constexpr rec_t kNone = ~0ull;           // no walk (its state bits read 3: never a valid state)

// lane l <- lane l+1; lane 63 <- `last`.  DPP ctrl 0x130 = wave_shl:1 (GFX9 family)
__device__ __forceinline__ int wave_shl1(int src, int last) {
  return __builtin_amdgcn_update_dpp(last, src, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ rec_t rec_from_right(rec_t v, rec_t last) {
  const uint32_t lo = (uint32_t)wave_shl1((int)(uint32_t)v, (int)(uint32_t)last);
  const uint32_t hi = (uint32_t)wave_shl1((int)(uint32_t)(v >> 32), (int)(uint32_t)(last >> 32));
  return ((rec_t)hi << 32) | lo;
}
__device__ __forceinline__ rec_t rec_lane(rec_t v, int lane_uniform) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane_uniform);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane_uniform);
  return ((rec_t)hi << 32) | lo;
}
__device__ __forceinline__ rec_t rec_min(rec_t a, rec_t b) { return a < b ? a : b; }
// an out record (walk leaving a cell) as the arrival it makes if it left in state `dir`: the key with the state
// it arrives in, or kNone
__device__ __forceinline__ rec_t arrival_of(rec_t out, uint32_t dir) {
  return ((uint32_t)out & 3u) == dir ? ((out & ~0xfull) | (((uint32_t)out >> 2) & 3u)) : kNone;
}

// reverse_move_t's view of the one predecessor cell a decision needs
struct RegAccess {
  int pm, pa, pb, ca, cb;
  __device__ __forceinline__ int code_a(uint32_t) const { return ca; }
  __device__ __forceinline__ int code_b(uint32_t) const { return cb; }
  __device__ __forceinline__ void cell(uint32_t, uint32_t, int &m, int &a, int &b) const { m = pm; a = pa; b = pb; }
};

template <int CPL, bool MULTI>
__global__ void __launch_bounds__(kWave) sw_sweep_kernel(const SaSweepParams p) {
  const int lane = threadIdx.x;
  const uint32_t pair = blockIdx.x;
  if (p.cand_count[pair] == 0) {
    if (lane == 0) { p.hit_count[pair] = 0; p.status[pair] = 0; }
    return;
  }
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const int32_t *__restrict__ Mg = p.M + mo, *__restrict__ Ag = p.A + mo, *__restrict__ Bg = p.B + mo;
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair], *__restrict__ sb_ = p.arena + p.off_b[pair];
  rec_t *hit_keys = p.hit_keys + mo;
  const uint32_t rmin = p.cand_box[4ull * pair], rmax = p.cand_box[4ull * pair + 1], cmin = p.cand_box[4ull * pair + 2],
                 cmax = p.cand_box[4ull * pair + 3];
  const int thr = max(p.min_score[pair], 1);
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  // plain scorings (no free / forbidden gaps, no sentinel scores): the three decisions of alignment_reverse_move
  // (alignment.c:311-327: GAP_A, then GAP_B, then MATCH) on 32-bit values -- SW scores are >= 0 and far from the
  // int range, so this is the 64-bit code's result; everything else goes through reverse_move_t
  const bool plain = !(p.flags & (SA_F_NO_START_GAP | SA_F_NO_END_GAP | SA_F_NO_GAPS_A | SA_F_NO_GAPS_B | SA_F_NO_MISMATCH |
                                  SA_F_HAS_SENTINEL));
  const uint32_t cshift = p.layout.row_bits, sshift = p.layout.row_bits + p.layout.col_bits;
  const int cap = p.layout.cap;
  constexpr uint32_t kSegW = kWave * CPL;
  const uint32_t seg_hi = MULTI ? cmax / kSegW : 0u, seg_cmin = MULTI ? cmin / kSegW : 0u;
  rec_t *rows = MULTI ? p.rows + p.row_off[pair] : nullptr;   // [2][W]

  int m[CPL], a[CPL], b[CPL], pm[CPL], pa[CPL], pb[CPL];   // this segment of row y / of row y - 1
  rec_t orec[CPL];                                          // walks leaving the cells of row y + 1
  int ca[CPL];                                              // codes of seq_a[x - 1] (one segment: loaded once)
  uint32_t n_hits = 0;                                      // wave-uniform
  uint32_t err = 0;                                         // per lane: error of the lowest walk that met one,
  rec_t err_key = kNone;                                    // and that walk
  int chunk_code = 0;                                       // lane t: code of seq_b[y - 1] for the row t below the chunk's top

  auto load_row = [&](uint32_t y, uint32_t x0, int (&dm)[CPL], int (&da)[CPL], int (&db)[CPL]) __attribute__((always_inline)) {
    const uint32_t at0 = y * W + x0 + lane;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const bool in = x0 + c * kWave + lane < W;
      dm[c] = in ? Mg[at0 + c * kWave] : 0; da[c] = in ? Ag[at0 + c * kWave] : 0; db[c] = in ? Bg[at0 + c * kWave] : 0;
    }
  };

  // One segment of one row.  left[6]: match / gap_a / gap_b of the column left of the segment on row y and on
  // row y - 1 (wave-uniform); r_prev / r_cur: the records right of the segment on row y + 1 / row y.  orec comes in
  // as row y + 1's records and leaves as row y's.  Returns whether any walk leaves this segment of the row.
  auto sweep_segment = [&](uint32_t y, uint32_t x0, const int (&left)[6], rec_t r_prev, rec_t r_cur, int code_b) __attribute__((always_inline)) -> bool {
    // ---- arrivals from below and the cell's own candidacy
    rec_t base[CPL];
    unsigned long long any = 0;
#pragma unroll
    for (int c = CPL - 1; c >= 0; --c) {
      const uint32_t x = x0 + c * kWave + lane;
      const rec_t right = (c == CPL - 1) ? r_prev : rec_lane(orec[c + 1 < CPL ? c + 1 : c], 0);
      const rec_t diag = arrival_of(rec_from_right(orec[c], right), MAT_MATCH);
      const rec_t vert = arrival_of(orec[c], MAT_GAP_A);
      const rec_t own = (m[c] >= thr && x < W)
                            ? ((((rec_t)(uint32_t)(cap - m[c]) << sshift) | ((rec_t)x << cshift) | y) << 4)
                            : kNone;
      base[c] = rec_min(own, rec_min(diag, vert));
      any |= __ballot(base[c] != kNone);
    }
    if (any == 0 && r_cur == kNone) {   // nothing arrives in this segment (a walk entering from the right would)
#pragma unroll
      for (int c = 0; c < CPL; ++c) orec[c] = kNone;
      return false;
    }
    // neighbours to the left: (x-1, y-1) for MATCH, (x-1, y) for GAP_B
    int dm_[CPL], da_[CPL], db_[CPL], lm_[CPL], la_[CPL], lb_[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      dm_[c] = wave_shr1(pm[c], c ? read_lane(pm[c ? c - 1 : 0], 63) : left[3]);
      da_[c] = wave_shr1(pa[c], c ? read_lane(pa[c ? c - 1 : 0], 63) : left[4]);
      db_[c] = wave_shr1(pb[c], c ? read_lane(pb[c ? c - 1 : 0], 63) : left[5]);
      lm_[c] = wave_shr1(m[c], c ? read_lane(m[c ? c - 1 : 0], 63) : left[0]);
      la_[c] = wave_shr1(a[c], c ? read_lane(a[c ? c - 1 : 0], 63) : left[1]);
      lb_[c] = wave_shr1(b[c], c ? read_lane(b[c ? c - 1 : 0], 63) : left[2]);
    }
    // the walk `w` that wins cell c: where does it go?  -> its out record (kNone + term: score 0, a hit)
    uint32_t term = 0;                      // bit c: the winner of slot c ends here
    auto decide = [&](int c, rec_t w) __attribute__((always_inline)) -> rec_t {
      const uint32_t st = (uint32_t)w & 3u, x = x0 + c * kWave + lane;
      const int s = st == MAT_MATCH ? m[c] : st == MAT_GAP_A ? a[c] : b[c];
      term &= ~(1u << c);
      if (s <= 0) { term |= 1u << c; return kNone; }
      const int qm = st == MAT_MATCH ? dm_[c] : st == MAT_GAP_A ? pm[c] : lm_[c];
      const int qa = st == MAT_MATCH ? da_[c] : st == MAT_GAP_A ? pa[c] : la_[c];
      const int qb = st == MAT_MATCH ? db_[c] : st == MAT_GAP_A ? pb[c] : lb_[c];
      int code_a = 0;
      if (st == MAT_MATCH) {
        if constexpr (MULTI) code_a = p.code[sa_[x - 1]];
        else code_a = ca[c];
      }
      uint32_t ns;
      if (plain) {
        int va = k.open1, vb = k.open1, vm = k.open1;
        if (st == MAT_MATCH) {
          va = vb = vm = (k.K <= 1) ? ((code_a & 0xff) == (code_b & 0xff) ? k.gen_eq : k.gen_ne)
                                    : subst_score<SA_SUBST_GLOBAL>(code_a & 0xff, (code_a >> 8) * k.K, code_b, k.table,
                                                                   k.gen_eq, k.gen_ne);
        } else if (st == MAT_GAP_A) va = k.ext;
        else vb = k.ext;
        ns = (qa + va == s) ? 1u : (qb + vb == s) ? 2u : 0u;
        if (ns == 0u && qm + vm != s) { if (!err || w < err_key) { err = 7; err_key = w; } }
      } else {
        RegAccess acc{qm, qa, qb, code_a, code_b};
        uint32_t qx = x, qy = y;
        int pmx = (int)st, ps = s;
        const uint32_t e = reverse_move_t(acc, k, la, lb, qx, qy, pmx, ps);
        if (e && (!err || w < err_key)) { err = e; err_key = w; }
        ns = (uint32_t)pmx;
      }
      return (w & ~0xfull) | (ns << 2) | st;
    };
    rec_t w[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      w[c] = base[c];
      orec[c] = kNone;
      if (w[c] != kNone) orec[c] = decide(c, w[c]);
    }
    // ---- arrivals along the row: iterate until the row is stable (its fixed point is unique: the rightmost cell
    // has no such arrival, and every cell is a function of the one to its right)
    for (;;) {
      bool changed = false;
#pragma unroll
      for (int c = CPL - 1; c >= 0; --c) {
        const rec_t right = (c == CPL - 1) ? r_cur : rec_lane(orec[c + 1 < CPL ? c + 1 : c], 0);
        const rec_t nw = rec_min(base[c], arrival_of(rec_from_right(orec[c], right), MAT_GAP_B));
        if (nw != w[c]) {   // (also when a walk that seemed to arrive does not: the cell to the right changed hands)
          w[c] = nw;
          term &= ~(1u << c);
          orec[c] = kNone;
          if (nw != kNone) orec[c] = decide(c, nw);
          changed = true;
        }
      }
      if (!__any(changed)) break;
    }
    // ---- hits: winners whose state has score 0
    bool out_live = false;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const bool hit = w[c] != kNone && ((term >> c) & 1u);
      const unsigned long long bal = __ballot(hit);
      if (bal) {
        const uint32_t pos = n_hits + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (hit) hit_keys[pos] = w[c] >> 4;
        n_hits += (uint32_t)__popcll(bal);
      }
      out_live |= orec[c] != kNone;
    }
    return __any(out_live);
  };

  uint32_t y = rmax;
  if constexpr (!MULTI) {
    // ------------------------------------------------------------------ the whole row in one segment
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t x = c * kWave + lane;
      ca[c] = (x >= 1 && x <= la) ? (int)p.code[sa_[x - 1]] : 0;
      orec[c] = kNone;
    }
    const int none[6] = {0, 0, 0, 0, 0, 0};   // column 0 is a border column: its states never move left
    int nm[CPL], na[CPL], nb[CPL];
    load_row(y, 0, m, a, b);
    if (y > 0) load_row(y - 1, 0, pm, pa, pb);
    for (;; --y) {
      if (y >= 2) load_row(y - 2, 0, nm, na, nb);   // in flight while this row is worked on
      const int q = (int)((rmax - y) & (kWave - 1));
      if (q == 0) {   // every 64 rows: lane t fetches seq_b's code for row y - t
        chunk_code = (y >= 1u + lane) ? (int)p.code[sb_[y - lane - 1]] : 0;
      }
      const bool live = sweep_segment(y, 0, none, kNone, kNone, read_lane(chunk_code, q));
      if (y == 0 || (!live && y <= rmin)) break;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        m[c] = pm[c]; a[c] = pa[c]; b[c] = pb[c];
        pm[c] = nm[c]; pa[c] = na[c]; pb[c] = nb[c];
      }
    }
  } else {
    // ------------------------------------------------------------------ column segments, records in HBM
    uint32_t prev_lo = seg_hi + 1;        // lowest segment whose records row y + 1 wrote (none yet)
    uint32_t prev_live_lo = seg_hi + 1;   // lowest segment of row y + 1 with a walk leaving it
    for (;; --y) {
      const int q = (int)((rmax - y) & (kWave - 1));
      if (q == 0) chunk_code = (y >= 1u + lane) ? (int)p.code[sb_[y - lane - 1]] : 0;
      const int code_b = read_lane(chunk_code, q);
      rec_t *cur_rows = rows + (size_t)(y & 1u) * W;
      const rec_t *prev_rows = rows + (size_t)((y + 1u) & 1u) * W;
      const bool box_row = y >= rmin;     // (y <= rmax always)
      rec_t r_cur = kNone;
      uint32_t lo = seg_hi + 1, live_lo = seg_hi + 1;
      bool row_live = false;
      for (uint32_t s = seg_hi;; --s) {
        const uint32_t x0 = s * kSegW;
        load_row(y, x0, m, a, b);
        if (y > 0) load_row(y - 1, x0, pm, pa, pb);
        int left[6] = {0, 0, 0, 0, 0, 0};
        if (x0 > 0) {
          const uint32_t at = y * W + x0 - 1;
          left[0] = Mg[at]; left[1] = Ag[at]; left[2] = Bg[at];
          if (y > 0) { left[3] = Mg[at - W]; left[4] = Ag[at - W]; left[5] = Bg[at - W]; }
        }
        const bool have_prev = s >= prev_lo;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const uint32_t x = x0 + c * kWave + lane;
          orec[c] = (have_prev && x < W) ? __hip_atomic_load(prev_rows + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kNone;
        }
        rec_t r_prev = kNone;
        if (s + 1 >= prev_lo && s + 1 <= seg_hi && x0 + kSegW < W)
          r_prev = __hip_atomic_load(prev_rows + x0 + kSegW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool live = sweep_segment(y, x0, left, r_prev, r_cur, code_b);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const uint32_t x = x0 + c * kWave + lane;
          if (x < W) cur_rows[x] = orec[c];
        }
        lo = s;
        if (live) { live_lo = s; row_live = true; }
        r_cur = rec_lane(orec[0], 0);
        if (s == 0) break;
        // is anything left of here reachable?  candidates, walks from the row below (a diagonal move crosses one
        // segment border at most), the walk leaving this segment's first column
        const bool more = (box_row && s - 1 >= seg_cmin) || s >= prev_live_lo || r_cur != kNone;
        if (!more) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this row's records are in L2 before the next row reads them
      prev_lo = lo; prev_live_lo = live_lo;
      if (y == 0 || (!row_live && y <= rmin)) break;
    }
  }

  // ---- the hits in key order (= the reference's order).  Up to 64: ranked here, one per lane.
  rec_t first_err = err ? err_key : kNone;   // the lowest erroring walk of the wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) first_err = rec_min(first_err, (rec_t)__shfl_xor(first_err, o));
  const unsigned long long err_lanes = __ballot(err != 0 && err_key == first_err);
  uint32_t status = err_lanes ? (uint32_t)__builtin_amdgcn_readlane((int)err, __builtin_ctzll(err_lanes)) : 0u;
  if (n_hits > 1) {
    if (n_hits <= (uint32_t)kWave) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const rec_t key = lane < (int)n_hits ? __hip_atomic_load(hit_keys + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kNone;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n_hits; ++j) rank += rec_lane(key, (int)j) < key;
      if (lane < (int)n_hits) hit_keys[rank] = key;
    } else {
      status |= SA_SWEEP_UNSORTED;
    }
  }
  if (lane == 0) {
    p.hit_count[pair] = n_hits;
    p.status[pair] = status;
    p.err_key[pair] = first_err >> 4;
  }
}

// ---- one traceback per wanted hit (smith_waterman.c:217-255).  One LANE per hit: the walk is a chain of dependent
// loads; the hits of a batch overlap each other's.
__global__ void __launch_bounds__(kWave) sw_hit_traceback_kernel(const SaHitTraceParams p) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= p.n_walkers) return;
  const uint32_t pair = p.walker_pair[w];
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const PairView v{p.arena + p.off_a[pair], p.arena + p.off_b[pair], p.M + mo, p.A + mo, p.B + mo, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  const unsigned long long key = p.hit_keys[mo + p.walker_rank[w]];
  const uint32_t end_y = (uint32_t)key & ((1u << p.layout.row_bits) - 1u);
  const uint32_t end_x = (uint32_t)(key >> p.layout.row_bits) & ((1u << p.layout.col_bits) - 1u);
  const int end_score = p.layout.cap - (int)(uint32_t)(key >> (p.layout.row_bits + p.layout.col_bits));
  char *oa = p.out_a + p.walker_str[w], *ob = p.out_b + p.walker_str[w];
  uint32_t x = end_x, y = end_y, head = la + lb, e = 0;
  int matrix = MAT_MATCH, score = end_score;
  while (score > 0) {
    --head;
    oa[head] = (matrix == MAT_GAP_A) ? '-' : (char)v.seq_a[x - 1];
    ob[head] = (matrix == MAT_GAP_B) ? '-' : (char)v.seq_b[y - 1];
    if ((e = reverse_move(v, k, x, y, matrix, score))) break;
  }
  SaDevHit h;   // smith_waterman.c:249-255
  h.score = end_score; h.pos_a = x; h.pos_b = y; h.len_a = end_x - x; h.len_b = end_y - y;
  h.length = la + lb - head; h.str_off = head;
  p.hits[w] = h;
  p.trace_status[w] = e;
}

// every hit's strings packed back to back for one D2H each: one wave per hit
__global__ void __launch_bounds__(256) gather_hits_kernel(const char *src_a, const char *src_b, const uint64_t *walker_str,
                                                          const SaDevHit *hits, const uint64_t *dst_off, char *dst_a,
                                                          char *dst_b, uint32_t n_walkers) {
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_walkers) return;
  const int lane = threadIdx.x & 63;
  const uint32_t n = hits[w].length;
  const char *sa_ = src_a + walker_str[w] + hits[w].str_off, *sb_ = src_b + walker_str[w] + hits[w].str_off;
  char *da = dst_a + dst_off[w], *db = dst_b + dst_off[w];
  for (uint32_t i = lane; i < n; i += 64) { da[i] = sa_[i]; db[i] = sb_[i]; }
}

template <int CPL>
static void launch_sweep(const SaSweepParams &p, hipStream_t stream) {
  hipLaunchKernelGGL((sw_sweep_kernel<CPL, false>), dim3(p.n_pairs), dim3(kWave), 0, stream, p);
}

}  // namespace sa

hipError_t sa_launch_sw_sweep(const SaSweepParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  uint32_t need = (p.max_len_a + 1 + sa::kWave - 1) / sa::kWave;   // columns per lane for the widest pair
  if (const char *env = getenv("SEQALIGN_SWEEP_SEGMENTS")) {       // tests: segments of 64 * v columns for every pair
    const int v = atoi(env);
    if ((v == 2 || v == 3) && p.rows) {
      if (v == 2) hipLaunchKernelGGL((sa::sw_sweep_kernel<2, true>), dim3(p.n_pairs), dim3(sa::kWave), 0, stream, p);
      else hipLaunchKernelGGL((sa::sw_sweep_kernel<3, true>), dim3(p.n_pairs), dim3(sa::kWave), 0, stream, p);
      return hipGetLastError();
    }
  }
  // (one column per lane is not instantiated: the compiler keeps its 1-element arrays in scratch)
  if (need <= 2) sa::launch_sweep<2>(p, stream);
  else if (need <= 3) sa::launch_sweep<3>(p, stream);
  else if (need <= 4) sa::launch_sweep<4>(p, stream);
  else if (need <= 5) sa::launch_sweep<5>(p, stream);
  else if (need <= 6) sa::launch_sweep<6>(p, stream);
  else if (need <= SA_SWEEP_SEGMENT / sa::kWave) sa::launch_sweep<SA_SWEEP_SEGMENT / sa::kWave>(p, stream);
  else hipLaunchKernelGGL((sa::sw_sweep_kernel<SA_SWEEP_SEGMENT / sa::kWave, true>), dim3(p.n_pairs), dim3(sa::kWave), 0, stream, p);
  return hipGetLastError();
}

hipError_t sa_launch_sw_hit_traceback(const SaHitTraceParams &p, hipStream_t stream) {
  if (p.n_walkers == 0) return hipSuccess;
  hipLaunchKernelGGL(sa::sw_hit_traceback_kernel, dim3((p.n_walkers + sa::kWave - 1) / sa::kWave), dim3(sa::kWave), 0, stream, p);
  return hipGetLastError();
}

hipError_t sa_launch_gather_hits(const char *src_a, const char *src_b, const uint64_t *walker_str, const SaDevHit *hits,
                                 const uint64_t *dst_off, char *dst_a, char *dst_b, uint32_t n_walkers,
                                 hipStream_t stream) {
  if (n_walkers == 0) return hipSuccess;
  hipLaunchKernelGGL(sa::gather_hits_kernel, dim3((n_walkers + 3) / 4), dim3(256), 0, stream, src_a, src_b, walker_str, hits,
                     dst_off, dst_a, dst_b, n_walkers);
  return hipGetLastError();
}
