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
// Per cell, the key of the walk that won it and four bits: where that walk goes (up-left / up / left / nowhere)
// and the state it arrives in.  Lane l owns CPL consecutive columns of a SEGMENT of 64 * CPL columns; match / gap_a /
// gap_b of a row are one wide load per lane and matrix.  Arrivals from the row below are register moves and a DPP
// shift at the lane border.  Arrivals along the row (GAP_B moves) make a right-to-left dependency: each lane resolves
// its own columns in order, and the lanes iterate until no lane's incoming walk changes (the row's fixed point is
// unique: the rightmost cell has no such arrival and every cell is a function of the one to its right).  Where a
// walk goes from a cell is worked out for all three states of every cell of a segment at once: the traceback's own
// decision (reverse_move_t; for plain scorings the same three equality tests on 32-bit values).  Keys are 32 bits
// wide when row, column and score fit 31 bits, else 64.  Three ways to lay a pair over waves:
//   * rows up to 512 columns: one wave per pair, the segment is the whole row and never moves; the winners stay in
//     registers, every row is loaded once (12 B per cell), the next row is in flight while this one is worked on;
//   * wider rows, many pairs: one wave per pair, the winners of two rows in LDS by column, and only the stretch of
//     a row anything can arrive in is worked on -- the candidate columns the fill reports per row
//     (SaFillParams::cand_rows) and the columns walks left the row below from -- in segments counted from its right
//     end (usually one);
//   * few wide pairs: one wave per 128- or 256-column strip, the strips of a pair a pipeline from right to left
//     (the first column's winners handed to the strip on the left through HBM).
//
// Bound: VALU work per cell (~100 instructions) when there are enough pairs; the ~2 us of dependent work per row
// when there are not.
#include <algorithm>

#include "sa_trace_common.hpp"

namespace sa {

// lane l <- lane l+1; lane 63 <- `last`.  DPP ctrl 0x130 = wave_shl:1 (GFX9 family)
__device__ __forceinline__ uint32_t wave_shl1(uint32_t src, uint32_t last) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)src, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned long long wave_shl1(unsigned long long v, unsigned long long last) {
  const uint32_t lo = wave_shl1((uint32_t)v, (uint32_t)last), hi = wave_shl1((uint32_t)(v >> 32), (uint32_t)(last >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int lane_uniform) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_uniform);
}
__device__ __forceinline__ unsigned long long lane_value(unsigned long long v, int lane_uniform) {
  return ((unsigned long long)lane_value((uint32_t)(v >> 32), lane_uniform) << 32) | lane_value((uint32_t)v, lane_uniform);
}

// reverse_move_t's view of the one predecessor cell a decision needs
struct RegAccess {
  int pm, pa, pb, ca, cb;
  __device__ __forceinline__ int code_a(uint32_t) const { return ca; }
  __device__ __forceinline__ int code_b(uint32_t) const { return cb; }
  __device__ __forceinline__ void cell(uint32_t, uint32_t, int &m, int &a, int &b) const { m = pm; a = pa; b = pb; }
};

// N consecutive ints from a 4-byte aligned address (rows start anywhere: the reference layout has pitch len_a + 1)
template <int N>
__device__ __forceinline__ void load_run(const int32_t *src, int (&v)[N], int first = 0) {
  if constexpr (N == 1) {
    v[first] = src[0];
  } else if constexpr (N == 2) {
    const v2i_u q = *reinterpret_cast<const v2i_u *>(src);
    v[first] = q.x; v[first + 1] = q.y;
  } else if constexpr (N == 3) {
    const v3i_u q = *reinterpret_cast<const v3i_u *>(src);
    v[first] = q.x; v[first + 1] = q.y; v[first + 2] = q.z;
  } else {
    const v4i_u q = *reinterpret_cast<const v4i_u *>(src);
    v[first] = q.x; v[first + 1] = q.y; v[first + 2] = q.z; v[first + 3] = q.w;
    if constexpr (N > 4) {
      int rest[N - 4];
      load_run<N - 4>(src + 4, rest);
#pragma unroll
      for (int k = 0; k < N - 4; ++k) v[first + 4 + k] = rest[k];
    }
  }
}

// what a cell's winner does next, 4 bits: bits 0-1 = where it goes (MAT_MATCH: up-left, MAT_GAP_A: up, MAT_GAP_B:
// left, kStay: nowhere -- no winner, or the walk ends here), bits 2-3 = the state it arrives in
constexpr uint32_t kStay = 3u;

// ROWS: how a pair is laid over waves, and where the winners of the row below live --
//   SA_ROWS_REG    one wave, the segment is the whole row (up to 512 columns): they stay in registers;
//   SA_ROWS_LDS    one wave, segments that follow the walks: by column in LDS (up to SA_SWEEP_LDS_COLUMNS columns);
//   SA_ROWS_STRIP  one wave per strip of 64 * CPL columns of the pair, the strips of a pair a pipeline from right to left:
//                  in registers, with the first column's winners handed to the strip on the left through HBM.
enum { SA_ROWS_REG = 0, SA_ROWS_LDS = 1, SA_ROWS_STRIP = 2 };

// (occupancy: the row loop is half latency -- a row's loads, its dependent passes -- so a wave more per SIMD is worth
// a few spilled registers on the cold paths: C3 2.91 -> 2.39 ms with 5 instead of 4; 6 loses again)
// PLAIN: the scoring has no free / forbidden gaps and no sentinel scores (the launcher decides): only the 32-bit
// decision code is compiled in, which is what every BASELINE config runs
template <int CPL, typename KeyT, int ROWS, bool PLAIN>
__global__ void __launch_bounds__(kWave, (ROWS == SA_ROWS_REG ? (CPL <= 2 ? 6 : CPL == 3 ? 5 : CPL == 4 ? 4 : CPL == 5 ? 3 : 1) : 1)) sw_sweep_kernel(const SaSweepParams p, const uint32_t table_ints, const uint32_t code_ints) {
  constexpr KeyT kNone = ~(KeyT)0;         // no walk
  const int lane = threadIdx.x;
  uint32_t pair = blockIdx.x, strip = 0;
  if constexpr (ROWS == SA_ROWS_STRIP) {
    // A workgroup draws a TICKET when it starts running (sa_fill_strips.hip has the argument in full): ticket =
    // (group of 8 pairs, strip counted from the RIGHT, pair in group), so the strip a wave waits for always holds a
    // lower ticket -- it is resident or done, whatever order the hardware dispatches workgroups in.
    uint32_t ticket = 0;
    if (lane == 0) ticket = atomicAdd(p.strip_progress + 2ull * gridDim.x, 1u);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    const uint32_t gs = ticket >> 3;
    pair = (gs / p.strips_per_pair) * 8 + (ticket & 7u);
    strip = p.strips_per_pair - 1 - gs % p.strips_per_pair;
    if (pair >= p.n_pairs) return;
    if (p.cand_count[pair] == 0) return;   // (hit_count / status / err_key are initialised by the host)
  } else {
    if (p.cand_count[pair] == 0) {
      if (lane == 0) { p.hit_count[pair] = 0; p.status[pair] = 0; }
      return;
    }
  }
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const int32_t *__restrict__ Mg = p.M + mo, *__restrict__ Ag = p.A + mo, *__restrict__ Bg = p.B + mo;
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair], *__restrict__ sb_ = p.arena + p.off_b[pair];
  // the pair's part of the scratch arena: its rows' candidate columns (from the fill), then room for its hits' keys
  const uint32_t *cand_rows = reinterpret_cast<const uint32_t *>(p.hit_keys + p.hit_off[pair]);
  unsigned long long *hit_keys = p.hit_keys + p.hit_off[pair] + lb + 1;
  const uint32_t hit_cap = (uint32_t)min(p.hit_off[pair + 1] - p.hit_off[pair] - (lb + 1), (uint64_t)0xffffffffu);
  const uint32_t rmin = p.cand_box[4ull * pair], rmax = p.cand_box[4ull * pair + 1];
  const int thr = max(p.min_score[pair], 1);
  // LDS: the substitution table (up to SA_LDS_TABLE_MAX_K classes: one lookup per cell and row), then the records
  extern __shared__ __attribute__((aligned(8))) int32_t lds_words[];
  const int32_t *table = p.table;
  if (table_ints) {
    for (uint32_t i = lane; i < p.K * p.K; i += kWave) lds_words[i] = p.table[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // one wave: program order; the fence pins the compiler
    __builtin_amdgcn_s_waitcnt(0);
    table = lds_words;
  }
  // the codes of seq_a, by column (column x holds seq_a[x - 1]; code_ints = 0: pairs too wide, read from HBM each time)
  uint16_t *col_code = reinterpret_cast<uint16_t *>(lds_words + table_ints);
  if (ROWS == SA_ROWS_LDS && code_ints) {
    for (uint32_t x = lane; x < W; x += kWave) col_code[x] = x >= 1 ? p.code[sa_[x - 1]] : (uint16_t)0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
  }
  // records of two rows: winner's key and what it does next, per column
  KeyT *rk = nullptr;
  uint32_t *rz = nullptr;
  if constexpr (ROWS == SA_ROWS_LDS) {
    rk = reinterpret_cast<KeyT *>(lds_words + table_ints + code_ints);
    rz = reinterpret_cast<uint32_t *>(rk + 2ull * p.lds_columns);
  }
  const uint32_t row_pitch = p.lds_columns;
  const TraceConsts k{p.code, table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  // plain scorings (no free / forbidden gaps, no sentinel scores): the three decisions of alignment_reverse_move
  // (alignment.c:311-327: GAP_A, then GAP_B, then MATCH) on 32-bit values -- SW scores are >= 0 and far from the
  // int range, so this is the 64-bit code's result; everything else goes through reverse_move_t
  constexpr bool plain = PLAIN;
  const uint32_t cshift = p.layout.row_bits, sshift = p.layout.row_bits + p.layout.col_bits;
  const int cap = p.layout.cap;
  constexpr int kSegW = kWave * CPL;

  int m[CPL], a[CPL], b[CPL], pm[CPL], pa[CPL], pb[CPL];   // this segment of row y / of row y - 1
  KeyT wk[CPL];                                             // row y + 1 coming in, row y going out: the cells' winners
  uint32_t wz[CPL];                                         // and what they do next (kStay: nothing leaves the cell)
  int ca[CPL];                                              // codes of seq_a[x - 1]
  int thr_c[CPL];                                           // the pair's min_score; INT_MAX for columns that do not exist
  uint32_t n_hits = 0;                                      // wave-uniform
  bool overflow = false;                                    // wave-uniform: the hit list ran out of room
  uint32_t err = 0;                                         // per lane: error of the lowest walk that met one,
  KeyT err_key = kNone;                                     // and that walk
  int chunk_code = 0;                                       // lane t: code of seq_b[y - 1] for the row t below the chunk's top
  uint2 chunk_range = make_uint2(0xffffffffu, 0u);          // lane t: that row's candidate columns
  int live_lo = INT32_MAX, live_hi = -1;                    // columns of this row a walk leaves (wave-uniform)
  unsigned long long tr_rows = 0, tr_active = 0, tr_rounds = 0, tr_cycles = 0, tr_a = 0, tr_b = 0, tr_c = 0;   // development aid (p.trace)
  const unsigned long long t_start = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;

  auto load_row = [&](uint32_t y, int x0, int (&dm)[CPL], int (&da)[CPL], int (&db)[CPL]) __attribute__((always_inline)) {
    const int xl = x0 + lane * CPL;
    const uint32_t at = y * W + (uint32_t)xl;
    if (y < lb && x0 >= 0) {   // (wave-uniform) a lane's run may reach past the row's end: that is the next row, still
                               // inside the pair's matrix
#pragma unroll
      for (int c = 0; c < CPL; ++c) { dm[c] = 0; da[c] = 0; db[c] = 0; }
      // (what a straddling lane reads past the row's end is not used: thr_c below)
      if ((uint32_t)xl < W) { load_run<CPL>(Mg + at, dm); load_run<CPL>(Ag + at, da); load_run<CPL>(Bg + at, db); }
    } else {                   // the last row, or a segment that straddles column 0: cell by cell
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const bool in = (uint32_t)(xl + c) < W;   // (negative columns wrap to huge values)
        dm[c] = in ? Mg[at + c] : 0; da[c] = in ? Ag[at + c] : 0; db[c] = in ? Bg[at + c] : 0;
      }
    }
  };
  auto rec_key = [&](uint32_t row, uint32_t col) __attribute__((always_inline)) -> KeyT { return rk[row * row_pitch + col]; };
  auto rec_next = [&](uint32_t row, uint32_t col) __attribute__((always_inline)) -> uint32_t { return rz[row * row_pitch + col]; };

  // One segment of one row: columns x0 .. x0 + 64 * CPL - 1 (x0 may be negative: those cells do not exist).
  // left[6]: match / gap_a / gap_b of column x0 - 1 on row y and on row y - 1 (wave-uniform); rp_k / rp_z: winner
  // and what-next of the cell right of the segment on row y + 1; rc_k / rc_z: of the cell right of it on row y.
  // wk / wz come in as row y + 1's and leave as row y's; live_lo / live_hi take in the columns a walk leaves.
  auto sweep_segment = [&](uint32_t y, int x0, const int (&left)[6], KeyT rp_k, uint32_t rp_z, KeyT rc_k, uint32_t rc_z,
                           int code_b) __attribute__((always_inline)) {
    const int xl = x0 + lane * CPL;
    // ---- arrivals from below and the cell's own candidacy
    KeyT bk[CPL];
    uint32_t bs[CPL];   // state the best arrival from below / the candidate stands in
    {
      const KeyT dk_edge = wave_shl1(wk[0], rp_k);          // the cell right of my last column, on row y + 1
      const uint32_t dz_edge = wave_shl1(wz[0], rp_z);
      bool any = false;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const KeyT dk = c + 1 < CPL ? wk[c + 1 < CPL ? c + 1 : c] : dk_edge;
        const uint32_t dz = c + 1 < CPL ? wz[c + 1 < CPL ? c + 1 : c] : dz_edge;
        KeyT best = (m[c] >= thr_c[c]) ? (KeyT)((((unsigned long long)(uint32_t)(cap - m[c]) << sshift) |
                                            ((unsigned long long)(uint32_t)(xl + c) << cshift) | y))
                                  : kNone;
        uint32_t st = MAT_MATCH;
        if ((dz & 3u) == MAT_MATCH && dk < best) { best = dk; st = dz >> 2; }
        if ((wz[c] & 3u) == MAT_GAP_A && wk[c] < best) { best = wk[c]; st = wz[c] >> 2; }
        bk[c] = best; bs[c] = st;
        any |= best != kNone;
      }
      if (!__any(any) && (rc_z & 3u) != MAT_GAP_B) {   // nothing arrives in this segment
#pragma unroll
        for (int c = 0; c < CPL; ++c) { wk[c] = kNone; wz[c] = kStay; }
        return;
      }
    }
    const unsigned long long t_seg = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    ++tr_active;
    // ---- where a walk standing on a cell goes, for each of its three states: 2 bits per state = the predecessor's
    // matrix, or 3 = this state's score is 0 (the walk ends here: a hit).  For every cell of the segment at once (no
    // divergence).  bad (scorings that are not plain): bit s = state s has no predecessor that explains its score
    // (bit 3 + s: its pair of characters has no score).
    uint32_t dir[CPL], bad[CPL];
    {
      const int e_pm = wave_shr1(pm[CPL - 1], left[3]), e_pa = wave_shr1(pa[CPL - 1], left[4]), e_pb = wave_shr1(pb[CPL - 1], left[5]);
      const int e_m = wave_shr1(m[CPL - 1], left[0]), e_a = wave_shr1(a[CPL - 1], left[1]), e_b = wave_shr1(b[CPL - 1], left[2]);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int cl = c ? c - 1 : 0;
        const int s3[3] = {m[c], a[c], b[c]};
        const int qm[3] = {c ? pm[cl] : e_pm, pm[c], c ? m[cl] : e_m};   // MATCH <- (x-1, y-1), GAP_A <- (x, y-1), GAP_B <- (x-1, y)
        const int qa[3] = {c ? pa[cl] : e_pa, pa[c], c ? a[cl] : e_a};
        const int qb[3] = {c ? pb[cl] : e_pb, pb[c], c ? b[cl] : e_b};
        (void)qm;
        dir[c] = 0x3fu; bad[c] = 0;
        if constexpr (plain) {
          const int sub = (k.K <= 1) ? ((ca[c] & 0xff) == (code_b & 0xff) ? k.gen_eq : k.gen_ne)
                                     : subst_score<SA_SUBST_LDS>(ca[c] & 0xff, (ca[c] >> 8) * k.K, code_b, k.table, k.gen_eq, k.gen_ne);
          const int va[3] = {sub, k.ext, k.open1}, vb[3] = {sub, k.open1, k.ext};
#pragma unroll
          for (int st = 0; st < 3; ++st) {
            const uint32_t f = (qa[st] + va[st] == s3[st]) ? 1u : (qb[st] + vb[st] == s3[st]) ? 2u : 0u;
            if (s3[st] > 0) dir[c] = (dir[c] & ~(3u << (2 * st))) | (f << (2 * st));
            // (f = 0 without qm + cost == s, the "program error" of alignment.c:329-345, cannot happen here: these
            // are the fill's own values, and a positive plain score IS one of its three candidates)
          }
        } else {
#pragma unroll
          for (int st = 0; st < 3; ++st) {
            if (s3[st] > 0) {
              RegAccess acc{qm[st], qa[st], qb[st], ca[c], code_b};
              uint32_t qx = (uint32_t)(xl + c), qy = y;
              int pmx = st, ps = s3[st];
              const uint32_t e = reverse_move_t(acc, k, la, lb, qx, qy, pmx, ps);
              if (e) bad[c] |= (e == 5u ? 8u : 1u) << st;
              dir[c] = (dir[c] & ~(3u << (2 * st))) | ((uint32_t)pmx << (2 * st));
            }
          }
        }
      }
    }
    // ---- arrivals along the row: my columns right to left, then again while some lane's incoming walk changes
    KeyT in_k = kNone;          // the walk entering my last column from the right, and its state
    uint32_t in_s = 0;
    uint32_t ws[CPL];           // state the winner stands in
    for (;;) {
      ++tr_rounds;
      KeyT hk = in_k;
      uint32_t hs = in_s;
#pragma unroll
      for (int c = CPL - 1; c >= 0; --c) {
        const bool side = hk < bk[c];
        const KeyT win = side ? hk : bk[c];
        const uint32_t st = side ? hs : bs[c];
        const uint32_t f = (dir[c] >> (2u * st)) & 3u;
        wk[c] = win; ws[c] = st;
        const bool leaves = win != kNone && f != 3u;
        wz[c] = leaves ? ((f << 2) | st) : kStay;
        hk = (leaves && st == MAT_GAP_B) ? win : kNone;   // what enters the column to the left
        hs = f;
      }
      // my first column's sideways walk is the left lane's incoming one; lane 63 takes the segment's right neighbour
      const KeyT nk = wave_shl1((wz[0] & 3u) == MAT_GAP_B ? wk[0] : kNone, (rc_z & 3u) == MAT_GAP_B ? rc_k : kNone);
      const uint32_t ns = wave_shl1(wz[0] >> 2, rc_z >> 2);
      const bool changed = nk != in_k || (nk != kNone && ns != in_s);
      in_k = nk; in_s = ns;
      if (!__any(changed)) break;
    }
    // a winner standing in a state that cannot be explained: the error of alignment_reverse_move (alignment.c:329-345)
    if constexpr (!plain) {
      bool any_bad = false;
#pragma unroll
      for (int c = 0; c < CPL; ++c) any_bad |= wk[c] != kNone && ((bad[c] >> ws[c]) & 9u) != 0u;
      if (__any(any_bad)) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          if (wk[c] != kNone && ((bad[c] >> ws[c]) & 9u) && (!err || wk[c] < err_key)) { err = ((bad[c] >> ws[c]) & 8u) ? 5u : 7u; err_key = wk[c]; }
      }
    }
    // ---- hits: winners whose state has score 0
    bool out_live = false;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const bool hit = wk[c] != kNone && ((dir[c] >> (2u * ws[c])) & 3u) == 3u;
      const unsigned long long bal = __ballot(hit);
      if (bal) {
        uint32_t first = n_hits;
        if constexpr (ROWS == SA_ROWS_STRIP) {   // the strips of a pair share its hit list
          if (lane == 0) first = atomicAdd(p.hit_count + pair, (uint32_t)__popcll(bal));
          first = __builtin_amdgcn_readfirstlane(first);
        }
        const uint32_t pos = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (hit && pos < hit_cap) hit_keys[pos] = (unsigned long long)wk[c];
        n_hits += (uint32_t)__popcll(bal);
        if (first + (uint32_t)__popcll(bal) > hit_cap) overflow = true;   // (cannot happen: SaSweepParams::hit_off)
      }
      out_live |= wz[c] != kStay;
    }
    const unsigned long long live = __ballot(out_live);
    if (live) {   // (at lane granularity: a superset is fine)
      live_lo = min(live_lo, x0 + (int)__builtin_ctzll(live) * CPL);
      live_hi = max(live_hi, x0 + (63 - (int)__builtin_clzll(live)) * CPL + CPL - 1);
    }
    if (p.trace) tr_cycles += __builtin_amdgcn_s_memtime() - t_seg;
  };

  if constexpr (ROWS == SA_ROWS_REG) {
    // ------------------------------------------------------------------ the whole row in one segment that never
    // moves: the winners stay in registers, every row is loaded once and kept for the row above it, the next row
    // is in flight while this one is worked on
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t x = lane * CPL + c;
      ca[c] = (x >= 1 && x <= la) ? (int)p.code[sa_[x - 1]] : 0;
      thr_c[c] = x < W ? thr : INT32_MAX;
      wk[c] = kNone; wz[c] = kStay;
    }
    const int none[6] = {0, 0, 0, 0, 0, 0};   // column 0 is a border column: its states never move left
    int nm[CPL], na[CPL], nb[CPL];
    uint32_t y = rmax;
    load_row(y, 0, m, a, b);
    if (y > 0) load_row(y - 1, 0, pm, pa, pb);
    for (;; --y) {
      if (y >= 2) load_row(y - 2, 0, nm, na, nb);
      const int q = (int)((rmax - y) & (kWave - 1));
      if (q == 0) chunk_code = (y >= 1u + lane) ? (int)p.code[sb_[y - lane - 1]] : 0;   // every 64 rows: lane t, row y - t
      live_hi = -1;
      sweep_segment(y, 0, none, kNone, kStay, kNone, kStay, read_lane(chunk_code, q));
      ++tr_rows;
      if (y == 0 || (live_hi < 0 && y <= rmin)) break;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        m[c] = pm[c]; a[c] = pa[c]; b[c] = pb[c];
        pm[c] = nm[c]; pa[c] = na[c]; pb[c] = nb[c];
      }
    }
  } else if constexpr (ROWS == SA_ROWS_STRIP) {
    // ------------------------------------------------------------------ one strip of a wide pair
    // Strip s works on columns [s * 64 * CPL, (s + 1) * 64 * CPL) of every row, bottom to top, like the one-wave
    // form above.  All it needs from outside is what enters from the right: the winner of the strip's right
    // neighbour column on this row and on the row below -- the strip to the right publishes its first column's
    // winners row by row (16 B per row) and, every strip_interval rows, how far it has got; this strip waits for
    // that before it starts a chunk of as many rows, so it runs that many rows behind.  A strip ends when nothing is alive in it, no candidate
    // lies above, and the strip to its right has ended; the ends ripple leftwards.
    const uint32_t s_hi = p.cand_box[4ull * pair + 3] / (uint32_t)kSegW;   // the strip holding the highest candidate column
    if (strip > s_hi) return;                                               // nothing ever happens right of it
    const int x0 = (int)(strip * (uint32_t)kSegW);
    const uint32_t S = p.strips_per_pair, ivl = p.strip_interval;   // ivl: rows between two publications (16 or 64)
    uint32_t *prog = p.strip_progress + 2ull * ((uint64_t)pair * S);        // [strip][rows done | 1 + rows done at the end]
    unsigned long long *bnd = p.bnd + 2ull * p.row_off[pair] * S;           // [row counted from rmax][strip][key | what-next]
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t x = (uint32_t)x0 + lane * CPL + c;
      ca[c] = (x >= 1 && x <= la) ? (int)p.code[sa_[x - 1]] : 0;
      thr_c[c] = x < W ? thr : INT32_MAX;
      wk[c] = kNone; wz[c] = kStay;
    }
    auto load_left = [&](uint32_t y, int (&d)[3]) __attribute__((always_inline)) {
      d[0] = d[1] = d[2] = 0;
      if (x0 > 0) { const uint32_t at = y * W + (uint32_t)x0 - 1; d[0] = Mg[at]; d[1] = Ag[at]; d[2] = Bg[at]; }
    };
    int nm[CPL], na[CPL], nb[CPL], lc[3], lp[3], ln[3] = {0, 0, 0};
    lp[0] = lp[1] = lp[2] = 0;
    uint32_t y = rmax, r = 0;                   // r: rows done, counted from rmax
    load_row(y, x0, m, a, b); load_left(y, lc);
    if (y > 0) { load_row(y - 1, x0, pm, pa, pb); load_left(y - 1, lp); }
    bool right_over = strip == s_hi;            // nothing (more) will come from the right
    bool right_ended = strip == s_hi;           // the strip to the right has published its end
    uint32_t avail = 0;                         // rows the strip to the right has published
    unsigned long long chunk_k = ~0ull, chunk_z = kStay;   // lane t: its first column's winner on row r0 + t
    KeyT rp_k = kNone;
    uint32_t rp_z = kStay;
    for (;; --y, ++r) {
      const int q = (int)(r & (kWave - 1));
      const int qb = (int)(r & (ivl - 1u));    // row inside the chunk of boundary winners
      if (q == 0) chunk_code = (y >= 1u + lane) ? (int)p.code[sb_[y - lane - 1]] : 0;
      if (qb == 0) {
        if (!right_over) {
          const uint32_t need = min(r + ivl, rmax + 1u);
          uint32_t done_rows, ended;
          // (the strip to my right holds a lower ticket: it is resident or done, see above)
          for (;;) {
            done_rows = __hip_atomic_load(prog + 2 * (strip + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ended = __hip_atomic_load(prog + 2 * (strip + 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done_rows >= need || ended) break;
            __builtin_amdgcn_s_sleep(8);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the loads below see those rows' winners
          avail = ended ? ended - 1 : done_rows;
          right_ended = ended != 0;
          const uint32_t rr = r + (uint32_t)lane;
          chunk_k = ~0ull; chunk_z = kStay;
          if (rr < avail && (uint32_t)lane < ivl) {
            const unsigned long long *src = bnd + 2ull * ((uint64_t)rr * S + strip + 1);
            chunk_k = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chunk_z = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (right_ended && avail <= r) right_over = true;
        }
      }
      KeyT rc_k = kNone;
      uint32_t rc_z = kStay;
      if (!right_over && r < avail) { rc_k = (KeyT)lane_value(chunk_k, qb); rc_z = (uint32_t)lane_value(chunk_z, qb); }
      if (y >= 2) { load_row(y - 2, x0, nm, na, nb); load_left(y - 2, ln); }
      const int left[6] = {lc[0], lc[1], lc[2], lp[0], lp[1], lp[2]};
      live_hi = -1;
      sweep_segment(y, x0, left, rp_k, rp_z, rc_k, rc_z, read_lane(chunk_code, q));
      ++tr_rows;
      if (strip > 0 && lane == 0) {   // my first column's winner on this row, for the strip to the left
        unsigned long long *dst = bnd + 2ull * ((uint64_t)r * S + strip);
        dst[0] = wk[0] == kNone ? ~0ull : (unsigned long long)wk[0];
        dst[1] = wz[0];
      }
      const bool last = y == 0 || (live_hi < 0 && y <= rmin && (right_over || (right_ended && r + 1 >= avail)));
      if (strip > 0 && (qb == (int)ivl - 1 || last)) {   // publish: rows up to this one are written
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) {
          __hip_atomic_store(prog + 2 * strip, r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (last) __hip_atomic_store(prog + 2 * strip + 1, r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (last) break;
      rp_k = rc_k; rp_z = rc_z;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        m[c] = pm[c]; a[c] = pa[c]; b[c] = pb[c];
        pm[c] = nm[c]; pa[c] = na[c]; pb[c] = nb[c];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) { lc[i] = lp[i]; lp[i] = ln[i]; }
    }
  } else {
    // ------------------------------------------------------------------ segments that follow the walks
  // the scores a segment needs: rows y and y - 1 at columns x0 .. x0 + 64 * CPL - 1, and the column left of them
  auto load_segment = [&](uint32_t y, int x0, int (&dm)[CPL], int (&da)[CPL], int (&db)[CPL], int (&dpm)[CPL], int (&dpa)[CPL],
                          int (&dpb)[CPL], int (&dleft)[6]) __attribute__((always_inline)) {
    load_row(y, x0, dm, da, db);
    if (y > 0) load_row(y - 1, x0, dpm, dpa, dpb);
#pragma unroll
    for (int i = 0; i < 6; ++i) dleft[i] = 0;
    if (x0 > 0) {
      const uint32_t at = y * W + (uint32_t)x0 - 1;
      dleft[0] = Mg[at]; dleft[1] = Ag[at]; dleft[2] = Bg[at];
      if (y > 0) { dleft[3] = Mg[at - W]; dleft[4] = Ag[at - W]; dleft[5] = Bg[at - W]; }
    }
  };
  // candidate columns of a row (lo > hi: none), from the chunk of 63 rows the lanes hold
  uint32_t chunk_top = 0;
  auto row_range = [&](uint32_t y, int &c_lo, int &c_hi) __attribute__((always_inline)) {
    const int idx = (int)(chunk_top - y);
    const uint32_t lo_ = (uint32_t)read_lane((int)chunk_range.x, idx), hi_ = (uint32_t)read_lane((int)chunk_range.y, idx);
    const bool some = lo_ <= hi_ && y >= rmin;
    c_lo = some ? (int)lo_ : INT32_MAX; c_hi = some ? (int)hi_ : -1;
  };

  int prev_w_lo = 1, prev_w_hi = 0;              // columns whose records row y + 1 wrote (none yet)
  int prev_live_lo = INT32_MAX, prev_live_hi = -1;
  // The first segment of a row ends at column `top`: the highest column anything can arrive in -- the row's own
  // candidates, and the columns walks left the row below from.  That is known only when the row below is done, so the
  // loads go out a row ahead for an upper bound of it (the candidates of this row and of the row below, the walks that
  // left the row below that one); too far right only costs a little of the segment's width.
  int top, n_top = -1;
  int nm[CPL], na[CPL], nb[CPL], npm[CPL], npa[CPL], npb[CPL], nleft[6], left[6];
  {
    chunk_top = rmax;
    chunk_code = (rmax >= 1u + lane) ? (int)p.code[sb_[rmax - lane - 1]] : 0;
    chunk_range = (rmax >= (uint32_t)lane) ? *reinterpret_cast<const uint2 *>(cand_rows + 2ull * (rmax - lane)) : make_uint2(0xffffffffu, 0u);
    int c_lo, c_hi;
    row_range(rmax, c_lo, c_hi);
    top = min(c_hi, (int)W - 1);
    load_segment(rmax, top - kSegW + 1, m, a, b, pm, pa, pb, left);
  }
  for (uint32_t y = rmax;; --y) {
    if (chunk_top - y >= (uint32_t)kWave - 1) {   // lane t: seq_b's code and the candidate columns of row y - t (rows y and y - 1 are needed)
      chunk_top = y;
      chunk_code = (y >= 1u + lane) ? (int)p.code[sb_[y - lane - 1]] : 0;
      chunk_range = (y >= (uint32_t)lane) ? *reinterpret_cast<const uint2 *>(cand_rows + 2ull * (y - lane)) : make_uint2(0xffffffffu, 0u);
    }
    const int code_b = read_lane(chunk_code, (int)(chunk_top - y));
    const unsigned long long tp0 = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    int c_lo, c_hi;
    row_range(y, c_lo, c_hi);
    ++tr_rows;
    if (y > 0) {   // the next row's first segment: loads in flight while this row is worked on
      int n_lo, n_hi;
      row_range(y - 1, n_lo, n_hi);
      n_top = min(max(max(n_hi, c_hi), prev_live_hi), (int)W - 1);
      if (n_top >= 0 && n_top == top) {   // the same columns: row y - 1 is here already, as this row's row above
#pragma unroll
        for (int c = 0; c < CPL; ++c) { nm[c] = pm[c]; na[c] = pa[c]; nb[c] = pb[c]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { nleft[i] = left[3 + i]; nleft[3 + i] = 0; }
        if (y > 1) {
          const int x0 = n_top - kSegW + 1;
          load_row(y - 2, x0, npm, npa, npb);
          if (x0 > 0) {
            const uint32_t at = (y - 2) * W + (uint32_t)x0 - 1;
            nleft[3] = Mg[at]; nleft[4] = Ag[at]; nleft[5] = Bg[at];
          }
        }
      } else if (n_top >= 0) {
        load_segment(y - 1, n_top - kSegW + 1, nm, na, nb, npm, npa, npb, nleft);
      }
    }
    const unsigned long long tp1 = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    // the stretch of this row anything can arrive in: its candidates, and up / up-left of where walks left row y + 1
    const int lo = max(min(c_lo, prev_live_lo == INT32_MAX ? INT32_MAX : prev_live_lo - 1), 0);
    const uint32_t cur = y & 1u, prv = cur ^ 1u;
    int w_lo = 1, w_hi = 0;
    live_lo = INT32_MAX; live_hi = -1;
    if (top >= 0 && (c_hi >= 0 || prev_live_hi >= 0)) {
      w_hi = top;
      KeyT rc_k = kNone;
      uint32_t rc_z = kStay;
      for (int x_top = top;;) {
        const int x0 = x_top - kSegW + 1, xl = x0 + lane * CPL;
        if (x_top != top) load_segment(y, x0, m, a, b, pm, pa, pb, left);   // (a second segment: the walks spread out)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int x = xl + c;
          ca[c] = (x >= 1 && x <= (int)la) ? (code_ints ? (int)col_code[x] : (int)p.code[sa_[x - 1]]) : 0;
          thr_c[c] = (uint32_t)x < W ? thr : INT32_MAX;
          const bool have = x >= prev_w_lo && x <= prev_w_hi;
          wk[c] = have ? rec_key(prv, (uint32_t)x) : kNone;
          wz[c] = have ? rec_next(prv, (uint32_t)x) : kStay;
        }
        KeyT rp_k = kNone;
        uint32_t rp_z = kStay;
        if (x_top + 1 >= prev_w_lo && x_top + 1 <= prev_w_hi) { rp_k = rec_key(prv, (uint32_t)x_top + 1); rp_z = rec_next(prv, (uint32_t)x_top + 1); }
        sweep_segment(y, x0, left, rp_k, rp_z, rc_k, rc_z, code_b);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int x = xl + c;
          if (x >= 0 && x <= x_top) { rk[cur * row_pitch + (uint32_t)x] = wk[c]; rz[cur * row_pitch + (uint32_t)x] = wz[c]; }
        }
        w_lo = max(x0, 0);
        if (x0 <= 0) break;
        rc_k = lane_value(wk[0], 0); rc_z = lane_value(wz[0], 0);
        x_top = x0 - 1;
        // is anything left of here reachable?  candidates and arrivals from below (lo), the walk leaving this
        // segment's first column sideways
        if (x_top < lo && (rc_z & 3u) != MAT_GAP_B) break;
      }
    }
    const unsigned long long tp2 = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    // this row's records are written before the next row reads them (one wave: LDS in program order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    prev_w_lo = w_lo; prev_w_hi = w_hi; prev_live_lo = live_lo; prev_live_hi = live_hi;
    if (y == 0 || (live_hi < 0 && y <= rmin)) break;
    top = n_top;
#pragma unroll
    for (int c = 0; c < CPL; ++c) { m[c] = nm[c]; a[c] = na[c]; b[c] = nb[c]; pm[c] = npm[c]; pa[c] = npa[c]; pb[c] = npb[c]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) left[i] = nleft[i];
    if (p.trace) { const unsigned long long tp3 = __builtin_amdgcn_s_memtime(); tr_a += tp1 - tp0; tr_b += tp2 - tp1; tr_c += tp3 - tp2; }
  }

  }

  // ---- the hits in key order (= the reference's order).  Up to 64: ranked here, one per lane.
  unsigned long long first_err = err ? (unsigned long long)err_key : ~0ull;   // the lowest erroring walk of the wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(first_err, o);
    first_err = other < first_err ? other : first_err;
  }
  const unsigned long long err_lanes = __ballot(err != 0 && (unsigned long long)err_key == first_err);
  uint32_t status = err_lanes ? (uint32_t)__builtin_amdgcn_readlane((int)err, __builtin_ctzll(err_lanes)) : 0u;
  if (overflow) status |= SA_SWEEP_OVERFLOW;
  if constexpr (ROWS == SA_ROWS_STRIP) {   // the strips of a pair report into the same words
    // the error of the LOWEST-ranked erroring walk over all strips (what the reference would have met first): key and
    // code travel in one word -- key << 1 | (code == 7) -- through one atomicMin (keys are at most 63 bits wide);
    // sw_order_hits_kernel takes them apart again
    if (lane == 0 && err_lanes) atomicMin(p.err_key + pair, (first_err << 1) | ((status & 7u) == 7u ? 1ull : 0ull));
    if (lane == 0 && overflow) atomicOr(p.status + pair, SA_SWEEP_OVERFLOW);
    return;
  }
  if (n_hits > 1 && !overflow) {
    if (n_hits <= (uint32_t)kWave) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned long long key = lane < (int)n_hits ? __hip_atomic_load(hit_keys + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n_hits; ++j) rank += lane_value(key, (int)j) < key;
      if (lane < (int)n_hits) hit_keys[rank] = key;
    } else {
      status |= SA_SWEEP_UNSORTED;
    }
  }
  if (lane == 0) {
    p.hit_count[pair] = n_hits;
    p.status[pair] = status;
    p.err_key[pair] = first_err;
    if (p.trace) {
      unsigned long long *t = p.trace + 8ull * pair;
      t[0] = __builtin_amdgcn_s_memtime() - t_start; t[1] = tr_rows; t[2] = tr_active; t[3] = tr_rounds; t[4] = tr_cycles; t[5] = tr_a; t[6] = tr_b; t[7] = tr_c;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same sweep over what sa_fill_dirs.hip leaves behind: match_scores (candidacy, keys) and one byte of directions
// per cell -- where a walk goes from a cell in each of its three states was decided by the fill, with the operands
// still in its registers.  What is left here is the part that IS the sweep: arrivals from the row below, the row's
// right-to-left passes, hits.  5 B per cell loaded instead of 12, two row registers per column instead of nine,
// no decision code at all.  Rows up to 512 columns (the whole row in one segment, winners in registers), plain
// scorings; one wave per pair.
template <int CPL, typename KeyT>
__global__ void __launch_bounds__(kWave, (CPL <= 3 ? 8 : CPL == 4 ? 6 : CPL <= 6 ? 5 : 4)) sw_sweep_dirs_kernel(const SaSweepParams p) {
  constexpr KeyT kNone = ~(KeyT)0;
  const int lane = threadIdx.x;
  const uint32_t pair = blockIdx.x;
  if (p.cand_count[pair] == 0) {
    if (lane == 0) { p.hit_count[pair] = 0; p.status[pair] = 0; p.err_key[pair] = ~0ull; }
    return;
  }
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const int32_t *__restrict__ Mg = p.M + mo;
  const uint8_t *__restrict__ Dg = p.dirs + mo;
  unsigned long long *hit_keys = p.hit_keys + p.hit_off[pair] + lb + 1;
  const uint32_t hit_cap = (uint32_t)min(p.hit_off[pair + 1] - p.hit_off[pair] - (lb + 1), (uint64_t)0xffffffffu);
  const uint32_t rmin = p.cand_box[4ull * pair], rmax = p.cand_box[4ull * pair + 1];
  const int thr = max(p.min_score[pair], 1);
  const uint32_t cshift = p.layout.row_bits, sshift = p.layout.row_bits + p.layout.col_bits;
  const int cap = p.layout.cap;

  int m[CPL], nm[CPL], thr_c[CPL];
  uint32_t d[CPL], nd[CPL];
  KeyT wk[CPL];
  uint32_t wz[CPL];
  uint32_t n_hits = 0;
  bool overflow = false;
  const int xl = lane * CPL;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    thr_c[c] = (uint32_t)(xl + c) < W ? thr : INT32_MAX;
    wk[c] = kNone; wz[c] = kStay;
  }
  auto load_row = [&](uint32_t y, int (&dm)[CPL], uint32_t (&dd)[CPL]) __attribute__((always_inline)) {
    const uint32_t at = y * W + (uint32_t)xl;
#pragma unroll
    for (int c = 0; c < CPL; ++c) { dm[c] = 0; dd[c] = 0x3fu; }
    if (y < lb) {   // a lane's run may reach past the row's end: that is the next row, still inside the pair's matrix
      if ((uint32_t)xl < W) {
        load_run<CPL>(Mg + at, dm);
#pragma unroll
        for (int c = 0; c < CPL; ++c) dd[c] = Dg[at + c];
      }
    } else {        // the last row: cell by cell
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        if ((uint32_t)(xl + c) < W) { dm[c] = Mg[at + c]; dd[c] = Dg[at + c]; }
    }
  };

  uint32_t y = rmax;
  load_row(y, m, d);
  for (;; --y) {
    if (y >= 1) load_row(y - 1, nm, nd);
    bool live = false;
    // ---- arrivals from below and the cell's own candidacy (sw_sweep_kernel::sweep_segment, same rules)
    KeyT bk[CPL];
    uint32_t bs[CPL];
    bool any = false;
    {
      const KeyT dk_edge = wave_shl1(wk[0], kNone);
      const uint32_t dz_edge = wave_shl1(wz[0], kStay);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const KeyT dk = c + 1 < CPL ? wk[c + 1 < CPL ? c + 1 : c] : dk_edge;
        const uint32_t dz = c + 1 < CPL ? wz[c + 1 < CPL ? c + 1 : c] : dz_edge;
        KeyT best = (m[c] >= thr_c[c]) ? (KeyT)((((unsigned long long)(uint32_t)(cap - m[c]) << sshift) |
                                            ((unsigned long long)(uint32_t)(xl + c) << cshift) | y))
                                  : kNone;
        uint32_t st = MAT_MATCH;
        if ((dz & 3u) == MAT_MATCH && dk < best) { best = dk; st = dz >> 2; }
        if ((wz[c] & 3u) == MAT_GAP_A && wk[c] < best) { best = wk[c]; st = wz[c] >> 2; }
        bk[c] = best; bs[c] = st;
        any |= best != kNone;
      }
    }
    if (!__any(any)) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) { wk[c] = kNone; wz[c] = kStay; }
    } else {
      // ---- arrivals along the row: my columns right to left, then again while some lane's incoming walk changes
      KeyT in_k = kNone;
      uint32_t in_s = 0;
      uint32_t ws[CPL];
      for (;;) {
        KeyT hk = in_k;
        uint32_t hs = in_s;
#pragma unroll
        for (int c = CPL - 1; c >= 0; --c) {
          const bool side = hk < bk[c];
          const KeyT win = side ? hk : bk[c];
          const uint32_t st = side ? hs : bs[c];
          const uint32_t f = (d[c] >> (2u * st)) & 3u;
          wk[c] = win; ws[c] = st;
          const bool leaves = win != kNone && f != 3u;
          wz[c] = leaves ? ((f << 2) | st) : kStay;
          hk = (leaves && st == MAT_GAP_B) ? win : kNone;
          hs = f;
        }
        const KeyT nk = wave_shl1((wz[0] & 3u) == MAT_GAP_B ? wk[0] : kNone, kNone);
        const uint32_t ns = wave_shl1(wz[0] >> 2, 0u);
        const bool changed = nk != in_k || (nk != kNone && ns != in_s);
        in_k = nk; in_s = ns;
        if (!__any(changed)) break;
      }
      // ---- hits: winners whose state has score 0
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const bool hit = wk[c] != kNone && ((d[c] >> (2u * ws[c])) & 3u) == 3u;
        const unsigned long long bal = __ballot(hit);
        if (bal) {
          const uint32_t pos = n_hits + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          if (hit && pos < hit_cap) hit_keys[pos] = (unsigned long long)wk[c];
          n_hits += (uint32_t)__popcll(bal);
          if (n_hits > hit_cap) overflow = true;   // (cannot happen: SaSweepParams::hit_off)
        }
        live |= wz[c] != kStay;
      }
    }
    if (y == 0 || (!__any(live) && y <= rmin)) break;
#pragma unroll
    for (int c = 0; c < CPL; ++c) { m[c] = nm[c]; d[c] = nd[c]; }
  }

  // ---- the hits in key order (= the reference's order).  Up to 64: ranked here, one per lane.
  uint32_t status = overflow ? SA_SWEEP_OVERFLOW : 0u;
  if (n_hits > 1 && !overflow) {
    if (n_hits <= (uint32_t)kWave) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned long long key = lane < (int)n_hits ? __hip_atomic_load(hit_keys + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n_hits; ++j) rank += lane_value(key, (int)j) < key;
      if (lane < (int)n_hits) hit_keys[rank] = key;
    } else {
      status |= SA_SWEEP_UNSORTED;
    }
  }
  if (lane == 0) {
    p.hit_count[pair] = n_hits;
    p.status[pair] = status;
    p.err_key[pair] = ~0ull;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same sweep, second form (round 4): a walk travels as ONE sortable word, ev = key << 2 | the state it arrives in.
// The reference's rule "the lowest-ranked arrival wins the cell" (a cell is marked by the first walk, in hit
// order, that reaches it -- smith_waterman.c:165-277 run sequentially) is then a plain minimum: keys are unique, so the two
// state bits never decide it, and the winner's state comes along in the word instead of through a pair of selects per
// arrival.  What a cell sends on is routed when it is produced -- up-left (the winner stood in MATCH), up (GAP_A) or left
// (GAP_B: inside the row) -- so the next row's arrivals are one v_min3_u32 per cell:
//     arrivals(c, y) = min3(own candidacy, up-left word of (c + 1, y + 1), up word of (c, y + 1))
// ~34 instead of ~50 vector instructions per cell slot (the first form: compare + and + two selects per arrival and state).
// Same hits, same order, same keys in the pair's list as sw_sweep_dirs_kernel (every sweep test runs both: option sweep_ev).
//
// Round 5: the rows' loads as a real pipeline.  Round 4's loop asked for row y - 1 before it worked on row y, but the compiler
// had put `s_waitcnt vmcnt(0)` right behind the request (the loads sat in branches -- last row / lane beyond the row's end --
// whose results merged at a join, and the first row's loads were still pending on loop entry): every row paid a full memory
// latency, SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.56, VALU busy 0.42 (profiles/r05/r05a_sweep_pmc_C3_before.json).  Now
//   * the loads are BRANCHLESS: a lane's run starts at min(its first column, W - 1) of the row -- always inside the pair's
//     matrix; what it reads beyond the row's end is the next row (or, on the pair's last row, the <= CPL * 4 bytes of match_scores / CPL + 3 bytes of directions behind the pair (60 / 18 at 16 columns per lane):
//     the next pair, padding, or the arenas' slack -- sa_host::reserve_arenas adds >= 4 KiB; the first buffer's cells beyond
//     column W - 1 are zeroed once, below).  Cells beyond the row's end can never win: no candidacy (thr_c) and no arrivals;
//   * NB row buffers rotate through an NB-times unrolled row loop: while row y is worked on, the loads of rows y - 1 ..
//     y - NB + 1 are in flight, and the only wait is the in-order vmcnt for the oldest of them -- written by hand (SweepRow:
//     inline-asm loads + `s_waitcnt vmcnt((NB - 1) x loads per row)`), because the compiler's own counting gave up at the
//     loop headers (a first version with plain loads still waited vmcnt(0) / vmcnt(1) where 6 was right);
//   * a lane's direction bytes stay one word; the two bits of the winner's state come out of it with one v_bfe_u32.
// One row of a lane's cells in flight: CPL match_scores (runs of 4 + a tail of 1..3) and the run's direction bytes as words,
// requested with inline-asm loads the compiler does not track and claimed with an explicit in-order `s_waitcnt vmcnt(N)`
// (below: why).  Between request and claim the registers are live (the claim takes them as "+v" operands) and nobody reads them.
template <int T> struct SweepTail { typedef int type; };
template <> struct SweepTail<2> { typedef int type __attribute__((ext_vector_type(2))); };
template <> struct SweepTail<3> { typedef int type __attribute__((ext_vector_type(3))); };
template <int CPL>
struct SweepRow {
  static constexpr int NV4 = CPL / 4, TAIL = CPL % 4, QW = (CPL + 3) / 4;
  static constexpr int kLoads = NV4 + (TAIL ? 1 : 0) + QW;   // VMEM instructions per row
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  v4i_t m4[NV4 ? NV4 : 1];
  typename SweepTail<TAIL>::type mt;
  uint32_t q[QW];
  // cell_off: the run's first cell, counted from the pair's first (bytes of directions = cells; match_scores: x 4).
  // ONE asm statement per row: the compiler does not look into it, so (a) the outputs are early-clobber -- they must not share a
  // register with an address a later load of the statement still reads -- and (b) the statement opens with `s_nop 4`: when the
  // SGPRs holding Mg / Dg were spilled to a VGPR, the compiler reloads them with v_readlane right in front of the statement and
  // cannot see that a VMEM instruction reads them within the five wait states gfx9 requires after a VALU write of an SGPR (a
  // first version without the nop faulted on exactly that, eight columns per lane: a stale high word of Mg).
  //  The ROW's base goes into the scalar address (row and W are wave-uniform: 64-bit SALU arithmetic), the lane's
  // column into the 32-bit VGPR offset: a pair of >= 2^30 cells (rows <= 1 024 columns, e.g. 500 x 2.2 M) used to wrap
  // `cell_off * 4` and read match_scores of the wrong rows.  The VGPR offsets are now < 4 * 1 024 + 64 whatever the pair's size.
  __device__ __forceinline__ void request(uint32_t row, uint32_t W, uint32_t xc, const int32_t *Mp, const uint8_t *Dp) {
    const uint64_t row_cells = (uint64_t)__builtin_amdgcn_readfirstlane(row) * (uint64_t)W;
    const int32_t *Mg = Mp + row_cells;
    const uint8_t *Dg = Dp + row_cells;
    const uint32_t cell_off = xc, mo = xc * 4u;
    if constexpr (NV4 == 0 && TAIL == 2)
      asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %4\n\tglobal_load_dword %1, %3, %5"
                   : "=&v"(mt), "=&v"(q[0]) : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else if constexpr (NV4 == 0 && TAIL == 3)
      asm volatile("s_nop 4\n\tglobal_load_dwordx3 %0, %2, %4\n\tglobal_load_dword %1, %3, %5"
                   : "=&v"(mt), "=&v"(q[0]) : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else if constexpr (NV4 == 1 && TAIL == 0)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %4\n\tglobal_load_dword %1, %3, %5"
                   : "=&v"(m4[0]), "=&v"(q[0]) : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else if constexpr (NV4 == 1 && TAIL == 1)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dword %1, %4, %6 offset:16\n\t"
                   "global_load_dword %2, %5, %7\n\tglobal_load_dword %3, %5, %7 offset:4"
                   : "=&v"(m4[0]), "=&v"(mt), "=&v"(q[0]), "=&v"(q[1]) : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else if constexpr (NV4 == 1 && TAIL == 2)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx2 %1, %4, %6 offset:16\n\t"
                   "global_load_dword %2, %5, %7\n\tglobal_load_dword %3, %5, %7 offset:4"
                   : "=&v"(m4[0]), "=&v"(mt), "=&v"(q[0]), "=&v"(q[1]) : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else if constexpr (NV4 == 2 && TAIL == 0)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %4, %6 offset:16\n\t"
                   "global_load_dword %2, %5, %7\n\tglobal_load_dword %3, %5, %7 offset:4"
                   : "=&v"(m4[0]), "=&v"(m4[1]), "=&v"(q[0]), "=&v"(q[1]) : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else if constexpr (NV4 == 3 && TAIL == 0)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %6, %8\n\tglobal_load_dwordx4 %1, %6, %8 offset:16\n\t"
                   "global_load_dwordx4 %2, %6, %8 offset:32\n\t"
                   "global_load_dword %3, %7, %9\n\tglobal_load_dword %4, %7, %9 offset:4\n\tglobal_load_dword %5, %7, %9 offset:8"
                   : "=&v"(m4[0]), "=&v"(m4[1]), "=&v"(m4[2]), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2])
                   : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else if constexpr (NV4 == 4 && TAIL == 0)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %8, %10\n\tglobal_load_dwordx4 %1, %8, %10 offset:16\n\t"
                   "global_load_dwordx4 %2, %8, %10 offset:32\n\tglobal_load_dwordx4 %3, %8, %10 offset:48\n\t"
                   "global_load_dword %4, %9, %11\n\tglobal_load_dword %5, %9, %11 offset:4\n\t"
                   "global_load_dword %6, %9, %11 offset:8\n\tglobal_load_dword %7, %9, %11 offset:12"
                   : "=&v"(m4[0]), "=&v"(m4[1]), "=&v"(m4[2]), "=&v"(m4[3]), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
                   : "v"(mo), "v"(cell_off), "s"(Mg), "s"(Dg));
    else
      static_assert(NV4 == 0 && TAIL == 2, "columns per lane: 2, 3, 4, 5, 6, 8, 12 or 16");
  }
  // the row is here once at most YOUNGER of the wave's VMEM instructions are outstanding (loads return in order)
  template <int YOUNGER>
  __device__ __forceinline__ void claim() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(YOUNGER) : "memory");
#pragma unroll
    for (int k = 0; k < NV4; ++k) asm volatile("" : "+v"(m4[k]));
    if constexpr (TAIL != 0) asm volatile("" : "+v"(mt));
#pragma unroll
    for (int k = 0; k < QW; ++k) asm volatile("" : "+v"(q[k]));
  }
  __device__ __forceinline__ int m(int c) const {
    if (c < 4 * NV4) return m4[c >> 2][c & 3];
    if constexpr (TAIL == 1) return mt; else if constexpr (TAIL != 0) return mt[c & 3]; else return 0;
  }
  __device__ __forceinline__ void set_m(int c, int v) {
    if (c < 4 * NV4) m4[c >> 2][c & 3] = v;
    else if constexpr (TAIL == 1) mt = v; else if constexpr (TAIL != 0) mt[c & 3] = v;
  }
};

// waves per SIMD the kernel is compiled for (its register budget): the row buffers must stay in registers -- a spilled buffer is a
// buffer stored while its load is still on its way (tools/check_inflight_loads.py fails the build on that)
constexpr int sweep_ev_waves(int cpl, bool wide_keys) {
  return wide_keys ? (cpl <= 2 ? 8 : cpl == 3 ? 6 : cpl == 4 ? 5 : cpl <= 6 ? 3 : 2)
                   : (cpl <= 3 ? 8 : cpl == 4 ? 6 : cpl <= 6 ? 5 : cpl <= 12 ? 4 : 3);
}
// MIXED (32-bit words only): the key inside the word is not the layout's bit fields but the same three numbers in mixed radix,
// ((cap - score) x W + column) x H + row with W, H = the chunk's longest sequences + 1 -- same order, up to three bits shorter,
// which is what puts reads of 600-700 bp against 1 000-column windows (31 bits of fields) into a 32-bit word: one v_min3_u32 per
// cell instead of two 64-bit compare-and-selects.  Hits (rare) are turned back into layout keys where they are stored.
template <int CPL, typename EvT, bool MIXED = false>
__global__ void __launch_bounds__(kWave, sweep_ev_waves(CPL, sizeof(EvT) > 4)) sw_sweep_dirs_ev_kernel(const SaSweepParams p) {
  static_assert(!MIXED || sizeof(EvT) == 4, "the mixed-radix word is the 32-bit one");
  constexpr EvT kNone = ~(EvT)0;
  constexpr int NB = CPL <= 4 ? 4 : CPL <= 8 ? 3 : 2;   // row buffers (12 / 16 columns per lane: a row's work outlasts a load)
  typedef SweepRow<CPL> Row;
  constexpr int kYounger = (NB - 1) * Row::kLoads;   // VMEM instructions behind a row's own when its turn comes
  const int lane = threadIdx.x;
  const uint32_t pair = blockIdx.x;
  if (p.cand_count[pair] == 0) {
    if (lane == 0) { p.hit_count[pair] = 0; p.status[pair] = 0; p.err_key[pair] = ~0ull; }
    return;
  }
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const int32_t *__restrict__ Mg = p.M + mo;
  const uint8_t *__restrict__ Dg = p.dirs + mo;
  unsigned long long *hit_keys = p.hit_keys + p.hit_off[pair] + lb + 1;
  const uint32_t hit_cap = (uint32_t)min(p.hit_off[pair + 1] - p.hit_off[pair] - (lb + 1), (uint64_t)0xffffffffu);
  const uint32_t rmin = p.cand_box[4ull * pair], rmax = p.cand_box[4ull * pair + 1];
  const int thr = max(p.min_score[pair], 1);
  const uint32_t cshift = p.layout.row_bits + 2u, sshift = p.layout.row_bits + p.layout.col_bits + 2u;   // (of the ev word)
  const uint32_t mix_h = p.max_len_b + 1u, mix_wh = (p.max_len_a + 1u) * mix_h;   // (MIXED: the radices; mix_wh * 4 < 2^24)
  const int cap = p.layout.cap;

  int thr_c[CPL];
  Row rb[NB];
  EvT out_diag[CPL], out_up[CPL];          // what the cells of the row below send up-left / up (kNone: nothing)
  EvT col_row[CPL];                        // column << cshift | row << 2 of my cells on the current row
  uint32_t n_hits = 0;
  bool overflow = false;
  const int xl = lane * CPL;
  const uint32_t xc = min((uint32_t)xl, W - 1u);   // where my run starts in memory (lanes beyond the row: its last cell)
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    thr_c[c] = (uint32_t)(xl + c) < W ? thr : INT32_MAX;
    out_diag[c] = kNone; out_up[c] = kNone;
    if constexpr (MIXED) col_row[c] = (EvT)((((uint32_t)(xl + c) * mix_h) + rmax) << 2);
    else col_row[c] = ((EvT)(uint32_t)(xl + c) << cshift) | ((EvT)rmax << 2);
  }
  uint32_t y = rmax;
#pragma unroll
  for (int b = 0; b < NB; ++b) rb[b].request(y >= (uint32_t)b ? y - (uint32_t)b : 0u, W, xc, Mg, Dg);
  // rmax may be the pair's last row, and what lies behind that is not the pair's: the first row's cells beyond column W - 1 = 0
  rb[0].template claim<kYounger>();
#pragma unroll
  for (int c = 0; c < CPL; ++c) rb[0].set_m(c, (uint32_t)(xl + c) < W ? rb[0].m(c) : 0);

  bool done = false;
  while (!done) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      Row &row = rb[b];
      row.template claim<kYounger>();
      int m[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) m[c] = row.m(c);
      bool live = false;
      // ---- arrivals from below and the cell's own candidacy: one minimum
      EvT arr[CPL];
      bool any = false;
      {
        const EvT diag_edge = wave_shl1(out_diag[0], kNone);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          EvT own_key;
          if constexpr (MIXED) own_key = (EvT)(__umul24((uint32_t)(cap - m[c]), mix_wh << 2) + (uint32_t)col_row[c]);
          else own_key = ((EvT)(uint32_t)(cap - m[c]) << sshift) | col_row[c];
          const EvT own = (m[c] >= thr_c[c]) ? own_key : kNone;   // (arrives in MATCH: 0)
          const EvT dg = c + 1 < CPL ? out_diag[c + 1 < CPL ? c + 1 : c] : diag_edge;
          arr[c] = min(own, min(dg, out_up[c]));
          any |= arr[c] != kNone;
        }
      }
      if (!__any(any)) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) { out_diag[c] = kNone; out_up[c] = kNone; }
      } else {
        // ---- arrivals along the row: my columns right to left, then again while some lane's incoming walk changes
        EvT in_ev = kNone, win[CPL];
        uint32_t fdir[CPL];
        for (;;) {
          EvT h = in_ev;
#pragma unroll
          for (int c = CPL - 1; c >= 0; --c) {
            const EvT w = min(h, arr[c]);
            const uint32_t st = (uint32_t)w & 3u;
            const uint32_t f = __builtin_amdgcn_ubfe(row.q[c >> 2], 2u * st + 8u * (uint32_t)(c & 3), 2u);
            win[c] = w; fdir[c] = f;
            // it goes on to the left iff it stands in GAP_B and its state's score is not 0 (f == 3: the walk ends here)
            h = (w != kNone && st == MAT_GAP_B && f != 3u) ? ((w & ~(EvT)3) | f) : kNone;
          }
          const EvT nxt = wave_shl1(h, kNone);
          const bool changed = nxt != in_ev;
          in_ev = nxt;
          if (!__any(changed)) break;
        }
        // ---- what every cell sends up-left / up; hits = winners whose state has score 0 (rare: one test for the whole row)
        bool any_hit = false;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const EvT w = win[c];
          const uint32_t st = (uint32_t)w & 3u, f = fdir[c];
          const bool has = w != kNone, leaves = has && f != 3u;
          const EvT on = (w & ~(EvT)3) | f;
          out_diag[c] = (leaves && st == MAT_MATCH) ? on : kNone;
          out_up[c] = (leaves && st == MAT_GAP_A) ? on : kNone;
          live |= leaves;
          any_hit |= has && f == 3u;
        }
        if (__any(any_hit)) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const bool hit = win[c] != kNone && fdir[c] == 3u;
            const unsigned long long bal = __ballot(hit);
            if (bal) {
              const uint32_t pos = n_hits + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
              if (hit && pos < hit_cap) {
                if constexpr (MIXED) {   // back to the layout's fields (the walkers and the host read those)
                  const uint32_t k = (uint32_t)(win[c] >> 2), s = k / mix_wh, rem = k - s * mix_wh, col = rem / mix_h, row = rem - col * mix_h;
                  hit_keys[pos] = ((unsigned long long)s << (sshift - 2u)) | ((unsigned long long)col << (cshift - 2u)) | row;
                } else {
                  hit_keys[pos] = (unsigned long long)(win[c] >> 2);
                }
              }
              n_hits += (uint32_t)__popcll(bal);
              if (n_hits > hit_cap) overflow = true;   // (cannot happen: SaSweepParams::hit_off)
            }
          }
          // stores count in vmcnt too and need not return in order with loads: drain, so that the claims' counts hold
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      if (y == 0 || (!__any(live) && y <= rmin)) { done = true; break; }
      row.request(y >= (uint32_t)NB ? y - (uint32_t)NB : 0u, W, xc, Mg, Dg);   // this buffer's next row: NB rows up (above the first: row 0 again, never used)
      --y;
#pragma unroll
      for (int c = 0; c < CPL; ++c) col_row[c] -= 4;   // (the row field: y - 1)
    }
  }

  // rows requested and never worked on are still on their way INTO registers: nothing below may start before they are in
#pragma unroll
  for (int b = 0; b < NB; ++b) rb[b].template claim<0>();
  // ---- the hits in key order (= the reference's order).  Up to 64: ranked here, one per lane.
  uint32_t status = overflow ? SA_SWEEP_OVERFLOW : 0u;
  if (n_hits > 1 && !overflow) {
    if (n_hits <= (uint32_t)kWave) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned long long key = lane < (int)n_hits ? __hip_atomic_load(hit_keys + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n_hits; ++j) rank += lane_value(key, (int)j) < key;
      if (lane < (int)n_hits) hit_keys[rank] = key;
    } else {
      status |= SA_SWEEP_UNSORTED;
    }
  }
  if (lane == 0) {
    p.hit_count[pair] = n_hits;
    p.status[pair] = status;
    p.err_key[pair] = ~0ull;
  }
}

// The strips of a pair append to its hit list in no particular order: one wave per pair ranks up to 64 hits
// afterwards (as the one-wave-per-pair forms do themselves); longer lists are flagged for the host.
__global__ void __launch_bounds__(kWave) sw_order_hits_kernel(const SaSweepParams p) {
  const int lane = threadIdx.x;
  const uint32_t pair = blockIdx.x, n_hits = p.hit_count[pair];
  {  // the strips' merged error: key << 1 | (code == 7) -> err_key, status
    const unsigned long long ek = p.err_key[pair];
    if (ek != ~0ull && lane == 0) {
      p.status[pair] |= (ek & 1ull) ? 7u : 5u;
      p.err_key[pair] = ek >> 1;
    }
  }
  if (n_hits <= 1 || (p.status[pair] & SA_SWEEP_OVERFLOW)) return;
  if (n_hits > (uint32_t)kWave) {
    if (lane == 0) p.status[pair] |= SA_SWEEP_UNSORTED;
    return;
  }
  unsigned long long *hit_keys = p.hit_keys + p.hit_off[pair] + p.len_b[pair] + 1;
  const unsigned long long key = lane < (int)n_hits ? hit_keys[lane] : ~0ull;
  uint32_t rank = 0;
  for (uint32_t j = 0; j < n_hits; ++j) rank += lane_value(key, (int)j) < key;
  if (lane < (int)n_hits) hit_keys[rank] = key;
}

// every hit's strings packed back to back for one D2H each: one wave per hit
__global__ void __launch_bounds__(256) gather_hits_kernel(const char *src_a, const char *src_b, const uint64_t *walker_str,
                                                          const uint32_t *head, const uint32_t *len, const uint64_t *dst_off,
                                                          char *dst_a, char *dst_b, uint32_t n_walkers) {
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_walkers) return;
  const int lane = threadIdx.x & 63;
  const uint32_t n = len[w];
  const char *sa_ = src_a + walker_str[w] + head[w], *sb_ = src_b + walker_str[w] + head[w];
  char *da = dst_a + dst_off[w], *db = dst_b + dst_off[w];
  for (uint32_t i = lane; i < n; i += 64) { da[i] = sa_[i]; db[i] = sb_[i]; }
}

template <int CPL, int ROWS>
static void launch_sweep(const SaSweepParams &p, hipStream_t stream) {
  const uint32_t table_ints = (p.K > 1 && p.K <= SA_LDS_TABLE_MAX_K) ? ((p.K * p.K + 1u) & ~1u) : 0u;
  // uint16 per column
  const uint32_t code_ints = (ROWS == SA_ROWS_LDS && p.max_len_a + 1 <= 16384u) ? (((p.max_len_a + 2u) / 2u + 1u) & ~1u) : 0u;
  // 32-bit keys when they fit with the all-ones value to spare
  const bool key32 = p.layout.row_bits + p.layout.col_bits + p.layout.score_bits <= 31;
  const dim3 grid(ROWS == SA_ROWS_STRIP ? sa_sweep_strip_blocks(p.n_pairs, p.max_len_a, kWave * CPL) : p.n_pairs), block(kWave);
  size_t lds = ((size_t)table_ints + code_ints) * 4;
  if (ROWS == SA_ROWS_LDS) lds += (size_t)2 * p.lds_columns * ((key32 ? 4 : 8) + 4);
  const bool plain = !(p.flags & (SA_F_NO_START_GAP | SA_F_NO_END_GAP | SA_F_NO_GAPS_A | SA_F_NO_GAPS_B | SA_F_NO_MISMATCH |
                                  SA_F_HAS_SENTINEL));
  if (plain) {
    if (key32) hipLaunchKernelGGL((sw_sweep_kernel<CPL, uint32_t, ROWS, true>), grid, block, lds, stream, p, table_ints, code_ints);
    else hipLaunchKernelGGL((sw_sweep_kernel<CPL, unsigned long long, ROWS, true>), grid, block, lds, stream, p, table_ints, code_ints);
  } else {
    if (key32) hipLaunchKernelGGL((sw_sweep_kernel<CPL, uint32_t, ROWS, false>), grid, block, lds, stream, p, table_ints, code_ints);
    else hipLaunchKernelGGL((sw_sweep_kernel<CPL, unsigned long long, ROWS, false>), grid, block, lds, stream, p, table_ints, code_ints);
  }
}

}  // namespace sa

uint32_t sa_sweep_strips_per_pair(uint32_t max_len_a, uint32_t strip_columns) { return (max_len_a + strip_columns) / strip_columns; }
uint32_t sa_sweep_strip_blocks(uint32_t n_pairs, uint32_t max_len_a, uint32_t strip_columns) {
  return ((n_pairs + 7u) / 8u) * 8u * sa_sweep_strips_per_pair(max_len_a, strip_columns);
}

namespace sa {
// the mixed-radix 32-bit word (sw_sweep_dirs_ev_kernel<.., true>): every key (cap - score, column, row) of the chunk below 2^30 - 1
// (all ones = "no walk"), and the score's factor small enough for the 24-bit multiplier
static bool sweep_mixed_fits(const SaSweepParams &p) {
  if (!p.max_len_b || p.layout.cap <= 0) return false;
  const uint64_t wh = ((uint64_t)p.max_len_a + 1) * ((uint64_t)p.max_len_b + 1);
  return wh * 4 < ((uint64_t)1 << 24) && (uint64_t)p.layout.cap < ((uint64_t)1 << 24) && (uint64_t)p.layout.cap * wh < ((uint64_t)1 << 30) - 1;
}
template <int CPL>
static void launch_sweep_dirs(const SaSweepParams &p, hipStream_t stream) {
  const uint32_t bits = p.layout.row_bits + p.layout.col_bits + p.layout.score_bits;
  if (p.tune_ev) {   // key << 2 | state in one word (62-bit keys at most: seqalign_sw_batch's layouts are <= 63 bits, the host path takes the rest)
    if (bits + 2 <= 32) { hipLaunchKernelGGL((sw_sweep_dirs_ev_kernel<CPL, uint32_t>), dim3(p.n_pairs), dim3(kWave), 0, stream, p); return; }
    if (sweep_mixed_fits(p)) { hipLaunchKernelGGL((sw_sweep_dirs_ev_kernel<CPL, uint32_t, true>), dim3(p.n_pairs), dim3(kWave), 0, stream, p); return; }
    if (bits + 2 <= 64) { hipLaunchKernelGGL((sw_sweep_dirs_ev_kernel<CPL, unsigned long long>), dim3(p.n_pairs), dim3(kWave), 0, stream, p); return; }
  }
  const bool key32 = bits <= 31;
  if (key32) hipLaunchKernelGGL((sw_sweep_dirs_kernel<CPL, uint32_t>), dim3(p.n_pairs), dim3(kWave), 0, stream, p);
  else hipLaunchKernelGGL((sw_sweep_dirs_kernel<CPL, unsigned long long>), dim3(p.n_pairs), dim3(kWave), 0, stream, p);
}
template <int CPL>
static void launch_sweep_dirs_wide(const SaSweepParams &p, hipStream_t stream) {
  const uint32_t bits = p.layout.row_bits + p.layout.col_bits + p.layout.score_bits;
  if (bits + 2 <= 32) hipLaunchKernelGGL((sw_sweep_dirs_ev_kernel<CPL, uint32_t>), dim3(p.n_pairs), dim3(kWave), 0, stream, p);
  else if (sweep_mixed_fits(p)) hipLaunchKernelGGL((sw_sweep_dirs_ev_kernel<CPL, uint32_t, true>), dim3(p.n_pairs), dim3(kWave), 0, stream, p);
  else hipLaunchKernelGGL((sw_sweep_dirs_ev_kernel<CPL, unsigned long long>), dim3(p.n_pairs), dim3(kWave), 0, stream, p);
}
}  // namespace sa

hipError_t sa_launch_sw_sweep(const SaSweepParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  if (p.dirs) {   // behind sa_fill_dirs.hip: match_scores + a byte of directions per cell (rows up to 1 024 columns)
    const uint32_t need = (p.max_len_a + 1 + sa::kWave - 1) / sa::kWave;
    if (need > 16 || p.strip_progress) return hipErrorInvalidValue;
    if (need > 8 && !(p.tune_ev && p.layout.row_bits + p.layout.col_bits + p.layout.score_bits + 2 <= 64)) return hipErrorInvalidValue;
    sa_record_launch(SEQALIGN_K_SWEEP_DIRS, p.n_pairs);
    if (need <= 2) sa::launch_sweep_dirs<2>(p, stream);
    else if (need <= 3) sa::launch_sweep_dirs<3>(p, stream);
    else if (need <= 4) sa::launch_sweep_dirs<4>(p, stream);
    else if (need <= 5) sa::launch_sweep_dirs<5>(p, stream);
    else if (need <= 6) sa::launch_sweep_dirs<6>(p, stream);
    else if (need <= 8) sa::launch_sweep_dirs<8>(p, stream);
    else if (need <= 12) sa::launch_sweep_dirs_wide<12>(p, stream);   // (rows of 513 .. 1 024 columns: the ev form only)
    else sa::launch_sweep_dirs_wide<16>(p, stream);
    return hipGetLastError();
  }
  // Up to 512 columns a segment holds the whole row and the winners stay in registers (short sequences: the walks
  // spread over most of the row anyway).  Beyond that: many pairs -- one wave per pair, segments of 256 columns that
  // follow the walks, the winners of two rows in LDS; few pairs, or rows too wide for LDS -- one wave per strip of
  // 64 / 128 / 256 columns (the caller decides: strip_progress != NULL, strip_columns).  the option sweep_cpl = 1, 2, 4 forces the LDS form
  // with segments of 64 * that many columns (tests, experiments).
  const uint32_t need = (p.max_len_a + 1 + sa::kWave - 1) / sa::kWave;   // columns per lane for the widest row
  int forced = (int)p.tune_cpl;
  if (forced != 1 && forced != 2 && forced != 4) forced = 0;
  sa_record_launch(p.strip_progress ? SEQALIGN_K_SWEEP_STRIPS : (!forced && need <= 8) ? SEQALIGN_K_SWEEP_REGS : SEQALIGN_K_SWEEP_LDS, p.n_pairs);
  if (p.strip_progress) {
    if (p.strip_columns == 64) sa::launch_sweep<1, sa::SA_ROWS_STRIP>(p, stream);
    else if (p.strip_columns == 128) sa::launch_sweep<2, sa::SA_ROWS_STRIP>(p, stream);
    else if (p.strip_columns == 256) sa::launch_sweep<4, sa::SA_ROWS_STRIP>(p, stream);
    else return hipErrorInvalidValue;
    hipLaunchKernelGGL(sa::sw_order_hits_kernel, dim3(p.n_pairs), dim3(sa::kWave), 0, stream, p);
  } else if (!forced && need <= 8) {
    // (one column per lane is not instantiated: no pair is that narrow in practice)
    if (need <= 2) sa::launch_sweep<2, sa::SA_ROWS_REG>(p, stream);
    else if (need <= 3) sa::launch_sweep<3, sa::SA_ROWS_REG>(p, stream);
    else if (need <= 4) sa::launch_sweep<4, sa::SA_ROWS_REG>(p, stream);
    else if (need <= 5) sa::launch_sweep<5, sa::SA_ROWS_REG>(p, stream);
    else if (need <= 6) sa::launch_sweep<6, sa::SA_ROWS_REG>(p, stream);
    else sa::launch_sweep<8, sa::SA_ROWS_REG>(p, stream);
  } else {
    const int cpl = forced ? forced : 4;
    if (!p.lds_columns) return hipErrorInvalidValue;
    if (cpl == 1) sa::launch_sweep<1, sa::SA_ROWS_LDS>(p, stream);
    else if (cpl == 2) sa::launch_sweep<2, sa::SA_ROWS_LDS>(p, stream);
    else sa::launch_sweep<4, sa::SA_ROWS_LDS>(p, stream);
  }
  return hipGetLastError();
}

hipError_t sa_launch_gather_hits(const char *src_a, const char *src_b, const uint64_t *walker_str, const uint32_t *head,
                                 const uint32_t *len, const uint64_t *dst_off, char *dst_a, char *dst_b, uint32_t n_walkers,
                                 hipStream_t stream) {
  if (n_walkers == 0) return hipSuccess;
  hipLaunchKernelGGL(sa::gather_hits_kernel, dim3((n_walkers + 3) / 4), dim3(256), 0, stream, src_a, src_b, walker_str, head,
                     len, dst_off, dst_a, dst_b, n_walkers);
  return hipGetLastError();
}
