// sa_sw_enum_window.hip -- Smith-Waterman multi-hit enumeration, one WORKGROUP per pair, everything the
// procedure touches staged in LDS.
//
// Reference semantics (src/smith_waterman.c:137-277): candidates (cells with match_scores >= min_score) are
// visited in (score desc, column asc, index asc) order; a candidate that is already marked is skipped, else it
// is walked back (alignment_reverse_move, alignment.c:244-350) to score 0 marking every cell, and the walk is
// abandoned -- its marks stay -- when it meets a marked cell; a completed walk is a hit.  A 150x1000 pair has
// ~15 000 candidates >= --minscore, a 300x300 BLOSUM62 pair ~50 000, nearly all of them in the plume of the
// one real hit: ~half are already marked when their turn comes, the others walk 1-3 cells into a marked one.
// The procedure is sequential by definition; what follows makes it parallel without changing its result.
//
// 1. The route of a walk does not depend on the marks -- only where it stops does.  So the predecessor of every
//    state is worked out ONCE, for all cells of a window around the candidates, by a streaming kernel at full
//    occupancy (sw_direction_kernel: one thread per cell, the decision code is the traceback's own,
//    sa_trace_common.hpp; 12 B per cell read, 1 B written): one byte per cell, 2 bits per state (predecessor
//    matrix, or 3 = "this state's score is 0": the walk ends here).  The enumeration workgroup copies its pair's
//    bytes into LDS and adds a visited bit; after that a walk step is ONE LDS byte read -- no match_scores / gap
//    scores / sequence / table access, no dependent HBM round trip.
// 2. A pool of T candidates in flight, one per thread, handed out in order (rank = position in the sorted list); a
//    thread whose candidate is done takes the next one.  Every iteration:
//      claim   every unfinished thread walks from where it stands until a marked cell / the end of the walk,
//              and atomicMin's its rank into a claim table slot (hash of the cell) for every cell it passes;
//      commit  it walks the same cells again and marks them for as long as the claim slot still holds ITS rank;
//              at the first cell claimed by a lower rank it stops and resumes there in the next iteration.
//    Why this is the sequential result: a thread's claimed path is a superset of its true path (it can only be
//    cut short by marks of lower ranks, and those are only ever committed when true, by induction), so cells no
//    lower rank claims are cells no lower rank will ever mark -- the sequential walk would find them unmarked
//    and mark them: committing them now is exact.  Marks of a higher rank never land on a lower rank's path
//    (the lower rank holds the claim).  The lowest unfinished rank always wins all its claims, so every
//    iteration makes progress; a walk is a hit when it has committed its way to a score-0 state.  A hash
//    collision only delays a commit by an iteration.  (C3 / C4 pairs: ~35 / ~90 iterations for 15 000 / 50 000
//    candidates with T = 1 024; checked against the sequential procedure by the tests and tools/fuzz_e2e.py.)
// 3. Completed hits are collected with their rank; the walk stops when the pool is empty or when max_hits hits lie
//    below every rank still in flight.  The first max_hits in rank order are written out, one thread per hit.
//
// The window is the candidates' bounding box (reported by the fill) extended up/left by a margin for the part of
// a hit that lies below min_score (up to 64 + 4 * min_score / best score per move cells, as LDS allows).  A walk
// that leaves it: the pair is flagged SA_ENUM_FALLBACK and run again with the largest window LDS can hold.  A box
// that fits no window, a traceback error to report, a second escape: SA_ENUM_GENERIC, the generic kernel
// (sa_sw_enum.hip) takes the pair -- exact, only slower.
#include <algorithm>

#include "sa_trace_common.hpp"

namespace sa {

constexpr uint32_t kVis = 0x40u;            // visited bit of a window byte
constexpr uint32_t kEsc = 0x80u;            // sentinel: outside the window (a walk that gets here leaves it)
constexpr size_t kWindowLdsLimit = 160u * 1024u - 2048u;   // dynamic LDS per workgroup (CDNA4: 160 KiB per CU; ~4 KiB are static)

// Byte offset of pair `pair`'s direction bytes: 4-byte aligned, regions never overlap.  A window is stored
// PADDED with one sentinel row above and one sentinel column to the left (kEsc bytes), so it has at most
// (W+1)(H+1) <= 2*W*H + 2 bytes for a pair of W*H cells.
__device__ __forceinline__ uint64_t dir_offset(uint64_t mat_off, uint32_t pair) {
  return (2ull * mat_off + 8ull * pair) & ~3ull;
}

struct Window {
  uint32_t r0, c0, Ww, Hw;   // first row / column of the matrix inside the window; real cells per row / rows
  bool ok;                   // stored padded: (Ww + 1) x (Hw + 1) bytes, cell (wx, wy) at (wy + 1) * (Ww + 1) + wx + 1
};

// the candidates' box extended up / left by the margin (shrunk until the window fits window_bytes); wave-uniform
__device__ __forceinline__ Window window_of(const SaEnumParams &p, uint32_t pair) {
  const uint32_t *box = p.cand_box + 4ull * pair;
  const uint32_t rmin = box[0], rmax = box[1], cmin = box[2], cmax = box[3];
  const uint64_t budget = p.window_bytes;
  auto area = [&](uint32_t m) {
    const uint32_t r0 = rmin > m ? rmin - m : 0, c0 = cmin > m ? cmin - m : 0;
    return (uint64_t)(rmax - r0 + 2) * (cmax - c0 + 2);   // with the sentinel row and column
  };
  Window w{0, 0, 0, 0, false};
  if (area(0) > budget) return w;
  const uint32_t thr = (uint32_t)max(p.min_score[pair], 1);
  // (the host picks the LDS class with a margin of 16 + thr / best_step; whatever room the class leaves is used)
  const uint32_t want = p.retry ? 0xffffffffu : 64u + 4u * ((thr + p.best_step - 1u) / p.best_step);
  uint32_t lo = 0, hi = min(want, max(rmin, cmin));
  while (lo < hi) {   // largest margin <= want with area <= budget
    const uint32_t mid = lo + (hi - lo + 1) / 2;
    if (area(mid) <= budget) lo = mid; else hi = mid - 1;
  }
  w.r0 = rmin > lo ? rmin - lo : 0;
  w.c0 = cmin > lo ? cmin - lo : 0;
  w.Ww = cmax - w.c0 + 1;
  w.Hw = rmax - w.r0 + 1;
  w.ok = true;
  return w;
}

// ---- 1. predecessor bits of every state of every pair's window.
// One workgroup per band of R window rows: the band's match / gap_a / gap_b values (plus the row above and the
// column to the left, where the predecessors live) and the sequences' codes are staged in LDS with coalesced
// loads -- every value is fetched from HBM once, 12 B per cell -- and the three decisions per cell are the
// traceback's own code (reverse_move_t) reading from the tile.  Bound: HBM reads of the window.
constexpr int kDirThreads = 256;
constexpr uint32_t kDirTileBytes = 24u * 1024u;
constexpr uint32_t kDirBandsPerBlock = 8;

struct TileAccess {
  const int32_t *tm, *ta, *tb;   // (rows) x TW planes: tile row 0 = matrix row yb, tile column 0 = matrix column xb
  const uint16_t *ca, *cb;       // codes of seq_a[x - 1] at [x - xb], of seq_b[y - 1] at [y - yb]
  int TW, xb, yb;
  __device__ __forceinline__ int code_a(uint32_t i) const { return ca[(int)i + 1 - xb]; }
  __device__ __forceinline__ int code_b(uint32_t j) const { return cb[(int)j + 1 - yb]; }
  __device__ __forceinline__ void cell(uint32_t x, uint32_t y, int &m, int &a, int &b) const {
    const int at = ((int)y - yb) * TW + ((int)x - xb);
    m = tm[at]; a = ta[at]; b = tb[at];
  }
};

__global__ void __launch_bounds__(kDirThreads) sw_direction_kernel(const SaEnumParams p, const uint32_t bands_per_pair,
                                                                   const uint32_t R, const uint32_t pair0) {
  extern __shared__ int32_t tile_lds[];
  const uint32_t slot = pair0 + blockIdx.x / bands_per_pair, band = blockIdx.x % bands_per_pair;
  const uint32_t pair = p.pair_list ? p.pair_list[slot] : slot;
  if (p.cand_count[pair] == 0) return;
  const Window w = window_of(p, pair);
  if (!w.ok) return;                                   // the enumeration kernel flags the pair
  const uint32_t Wp = w.Ww + 1, Hp = w.Hw + 1;
  if (band * kDirBandsPerBlock * R >= Hp) return;
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const int32_t *Mg = p.M + mo, *Ag = p.A + mo, *Bg = p.B + mo;
  const uint8_t *sa_ = p.arena + p.off_a[pair], *sb_ = p.arena + p.off_b[pair];
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  bool bad = false;
  // a workgroup takes kDirBandsPerBlock consecutive bands: the set-up above (dependent loads) is paid once
  for (uint32_t sub = 0; sub < kDirBandsPerBlock; ++sub) {
  const uint32_t py0 = (band * kDirBandsPerBlock + sub) * R;
  if (py0 >= Hp) break;
  const uint32_t rows = min(R, Hp - py0);              // padded rows [py0, py0 + rows) are mine
  if (sub) __syncthreads();                            // the previous band's tile is no longer read
  // tile: matrix rows yb .. yb + rows (one more than mine: the row above), columns xb .. xb + Wp - 1 (tile column
  // = padded window column: the sentinel column's place holds the real column left of the window)
  const int TW = (int)Wp, xb = (int)w.c0 - 1, yb = (int)(w.r0 + py0) - 2;
  const uint32_t plane = (rows + 1) * Wp;
  int32_t *tm = tile_lds, *ta = tm + plane, *tb = ta + plane;
  uint16_t *ca = reinterpret_cast<uint16_t *>(tb + plane), *cb = ca + Wp;
  // (idx -> (row, column) without a division per element: one at the start, then steps of kDirThreads)
  const uint32_t step_r = kDirThreads / Wp, step_c = kDirThreads - step_r * Wp;
  for (uint32_t ty = threadIdx.x / Wp, tx = threadIdx.x - ty * Wp, idx = threadIdx.x; idx < plane; idx += kDirThreads) {
    const int x = xb + (int)tx, y = yb + (int)ty;
    const bool inside = x >= 0 && y >= 0;
    const uint32_t at = inside ? (uint32_t)y * W + (uint32_t)x : 0u;
    tm[idx] = inside ? Mg[at] : 0; ta[idx] = inside ? Ag[at] : 0; tb[idx] = inside ? Bg[at] : 0;
    tx += step_c; ty += step_r;
    if (tx >= Wp) { tx -= Wp; ++ty; }
  }
  for (uint32_t tx = threadIdx.x; tx < Wp; tx += kDirThreads) { const int x = xb + (int)tx; ca[tx] = x >= 1 ? p.code[sa_[x - 1]] : 0; }
  for (uint32_t ty = threadIdx.x; ty <= rows; ty += kDirThreads) { const int y = yb + (int)ty; cb[ty] = y >= 1 ? p.code[sb_[y - 1]] : 0; }
  __syncthreads();

  TileAccess acc{tm, ta, tb, ca, cb, TW, xb, yb};
  uint8_t *dir = p.dir + dir_offset(mo, pair) + (uint64_t)py0 * Wp;
  // plain scorings (no free / forbidden gaps, no sentinel scores): the three decisions of alignment_reverse_move
  // (alignment.c:311-327: GAP_A, then GAP_B, then MATCH) written out on 32-bit values -- SW scores are >= 0 and far
  // from the int range, so this is the 64-bit code's result; everything else goes through reverse_move_t
  const bool plain = !(p.flags & (SA_F_NO_START_GAP | SA_F_NO_END_GAP | SA_F_NO_GAPS_A | SA_F_NO_GAPS_B | SA_F_NO_MISMATCH |
                                  SA_F_HAS_SENTINEL));
  for (uint32_t pr = threadIdx.x / Wp, px = threadIdx.x - pr * Wp, idx = threadIdx.x; idx < rows * Wp;
       idx += kDirThreads, px += step_c, pr += step_r) {
    if (px >= Wp) { px -= Wp; ++pr; }
    const uint32_t py = py0 + pr;
    uint32_t byte = kEsc;                                // sentinel row / column
    if (py != 0 && px != 0) {
      const uint32_t x = w.c0 + px - 1, y = w.r0 + py - 1;
      byte = 0x3f;                                       // border cells: every SW border score is 0
      if (x > 0 && y > 0) {
        const uint32_t at = (pr + 1) * Wp + px;          // (x, y) in the tile
        const int s[3] = {tm[at], ta[at], tb[at]};
        if (plain) {
          if (s[0] > 0) {                                // MATCH <- (x-1, y-1), every move costs the substitution score
            const int code_a = ca[px], code_b = cb[pr + 1];
            const int sub = (k.K <= 1) ? ((code_a & 0xff) == (code_b & 0xff) ? k.gen_eq : k.gen_ne)
                                       : subst_score<SA_SUBST_GLOBAL>(code_a & 0xff, (code_a >> 8) * k.K, code_b, k.table,
                                                                      k.gen_eq, k.gen_ne);
            const uint32_t d = at - Wp - 1;
            const uint32_t f = (ta[d] + sub == s[0]) ? 1u : (tb[d] + sub == s[0]) ? 2u : 0u;
            if (f == 0u && tm[d] + sub != s[0]) bad = true;
            byte = (byte & ~3u) | f;
          }
          if (s[1] > 0) {                                // GAP_A <- (x, y-1): extend from gap_a, open from the others
            const uint32_t d = at - Wp;
            const uint32_t f = (ta[d] + k.ext == s[1]) ? 1u : (tb[d] + k.open1 == s[1]) ? 2u : 0u;
            if (f == 0u && tm[d] + k.open1 != s[1]) bad = true;
            byte = (byte & ~(3u << 2)) | (f << 2);
          }
          if (s[2] > 0) {                                // GAP_B <- (x-1, y): extend from gap_b, open from the others
            const uint32_t d = at - 1;
            const uint32_t f = (ta[d] + k.open1 == s[2]) ? 1u : (tb[d] + k.ext == s[2]) ? 2u : 0u;
            if (f == 0u && tm[d] + k.open1 != s[2]) bad = true;
            byte = (byte & ~(3u << 4)) | (f << 4);
          }
        } else
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          if (s[m] > 0) {
            uint32_t qx = x, qy = y;
            int pm = m, ps = s[m];
            if (reverse_move_t(acc, k, la, lb, qx, qy, pm, ps)) bad = true;   // the generic kernel reports the error
            byte = (byte & ~(3u << (2 * m))) | ((uint32_t)pm << (2 * m));
          }
        }
      }
    }
    dir[idx] = (uint8_t)byte;
  }
  }
  if (bad) p.enum_status[pair] = SA_ENUM_GENERIC;
}

constexpr int kMaxPoolHits = 64;            // hits a pair may complete before the first max_hits are known
constexpr int kInlineSteps = 3;             // cells a thread walks itself before handing a long walk to the queue

// One walk step back from window cell `at` in state m2 (= 2 * matrix) whose byte is `byte`: the predecessor's
// matrix field, or 3 when this state's score is 0.
__device__ __forceinline__ uint32_t dir_field(uint32_t byte, uint32_t m2) { return (byte >> m2) & 3u; }

template <int T, typename KeyT>
__global__ void __launch_bounds__(T) sw_enumerate_window_kernel(const SaEnumParams p) {
  extern __shared__ uint32_t lds32[];
  __shared__ uint32_t s_flag;              // != 0: hand the pair to the generic kernel
  __shared__ uint32_t s_next;              // next candidate (rank) to hand to a free thread
  __shared__ uint32_t s_nhits;             // completed hits so far (any order)
  __shared__ uint32_t s_min_active;        // lowest rank still being walked (only maintained once s_nhits >= max_hits)
  __shared__ uint32_t s_hit_rank[kMaxPoolHits], s_hit_steps[kMaxPoolHits], s_hit_off[kMaxPoolHits];
  __shared__ uint32_t s_emit;
  __shared__ uint32_t s_nq[2];             // long walks queued in the claim / commit phase
  const uint32_t pair = p.pair_list ? p.pair_list[blockIdx.x] : blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const int lane = tid & 63;

  const uint32_t n_cand = p.cand_count[pair];
  if (n_cand == 0) {
    if (tid == 0) { p.hit_count[pair] = 0; p.str_used[pair] = 0; p.enum_status[pair] = 0; }
    return;
  }
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const unsigned long long t_start = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;

  // LDS: claim table | queue of long walks (3 words each, one slot per thread) | ring of 2T prefetched keys | window
  const uint32_t nclaim = 1u << p.claim_bits;
  uint32_t *claim = lds32;
  uint32_t *s_q0 = lds32 + nclaim, *s_q1 = s_q0 + T, *s_q2 = s_q1 + T;
  constexpr uint32_t kQueueCap = T;
  KeyT *ring = reinterpret_cast<KeyT *>(s_q2 + T);
  uint32_t *win32 = s_q2 + T + 2 * T * (sizeof(KeyT) / 4);
  uint8_t *win = reinterpret_cast<uint8_t *>(win32);
  const Window w = window_of(p, pair);
  if (!w.ok || p.enum_status[pair] == SA_ENUM_GENERIC) {   // box too large for LDS / flagged by the direction kernel
    if (tid == 0) p.enum_status[pair] = SA_ENUM_GENERIC;
    return;
  }
  const uint32_t r0 = w.r0, c0 = w.c0, Wp = w.Ww + 1, wbytes = Wp * (w.Hw + 1);
  const PairView v{p.arena + p.off_a[pair], p.arena + p.off_b[pair], p.M + mo, p.A + mo, p.B + mo, la, lb, W};
  const KeyT *keys = static_cast<const KeyT *>(p.keys) + mo;
  if (tid == 0) { s_flag = 0; s_next = 0; s_nhits = 0; s_min_active = 0xffffffffu; s_emit = 0; s_nq[0] = s_nq[1] = 0; }
  // the window's direction bytes -> LDS (visited bits clear: a fresh mask per pair); first 2T keys
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.dir + dir_offset(mo, pair));
    for (uint32_t i = tid; i < (wbytes + 3) / 4; i += T) win32[i] = src[i];
    for (uint32_t i = tid; i < 2 * T && i < n_cand; i += T) ring[i] = keys[i];
  }
  uint32_t loaded = min(n_cand, 2u * T);    // keys [0, loaded) have been put into the ring (slot = rank % 2T)
  uint32_t nx = 0;                          // s_next as every thread last saw it at a point where nobody changes it
  __syncthreads();
  const unsigned long long t_loaded = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;

  const int min_score = p.min_score[pair];
  const uint32_t cshift = p.layout.row_bits, sshift = p.layout.row_bits + p.layout.col_bits;
  const uint32_t rmask = (1u << p.layout.row_bits) - 1u, cmask = (1u << p.layout.col_bits) - 1u;
  const uint32_t cb = p.claim_bits, hmask = nclaim - 1u;
  auto slot_of = [&](uint32_t cell) { return (cell ^ (cell >> cb)) & hmask; };
  // one step back from a cell in state m2 (= 2 * matrix): MATCH up-left, GAP_A up, GAP_B left.  The three
  // distances sit in one 64-bit constant, 16 bits each, selected with a shift: no branch, no table in memory
  const unsigned long long deltas = (unsigned long long)(Wp + 1u) | ((unsigned long long)Wp << 16) | (1ull << 32);
  auto back = [&](uint32_t m2) { return (uint32_t)(deltas >> (m2 * 8u)) & 0xffffu; };

  // Claim from cell `cat` (unmarked, its byte given), state cm2, for at most `budget` cells.
  // Returns 0 = ran into a marked cell, 1 = reached a score-0 state, 2 = budget used up (cat / cm2 / byte = the
  // next cell to claim), 3 = left the window.
  // The loop is written for its dependent chain: byte -> field -> address -> next byte.  The next cell's byte is
  // requested before this cell's claim goes out (LDS operations of a wave complete in order), and nothing in
  // the body branches except the exits.
  auto claim_walk = [&](uint32_t r, uint32_t &cat, uint32_t &cm2, uint32_t &byte, uint32_t &n, uint32_t budget) -> int {
    for (uint32_t s = 0;;) {
      const uint32_t d = dir_field(byte, cm2), here = cat;
      cat = (d == 3u) ? cat : cat - back(cm2);
      const uint32_t nbyte = win[cat];
      atomicMin(&claim[slot_of(here)], r);
      ++n;
      if (d == 3u) return 1;
      cm2 = 2u * d;
      byte = nbyte;
      if (byte & (kVis | kEsc)) return (byte & kEsc) ? 3 : 0;
      if (++s == budget) return 2;
    }
  };
  // Mark the cells of my claimed path from `cat` on, while the claim slot still holds my rank, for at most
  // `budget` cells; `left` = claimed cells not yet committed.  Returns 0 = a lower rank holds the next cell
  // (resume there), 1 = all claimed cells committed, 2 = budget used up.  (The cell's byte and its claim slot
  // are read together: one LDS latency per step.)
  auto commit_walk = [&](uint32_t r, uint32_t &cat, uint32_t &cm2, uint32_t &left, uint32_t budget) -> int {
    uint32_t byte = win[cat], owner = claim[slot_of(cat)];
    for (uint32_t s = 0;;) {
      if (owner != r) return 0;
      atomicOr(&win32[cat >> 2], kVis << (8u * (cat & 3u)));
      if (--left == 0) return 1;
      cat -= back(cm2);
      cm2 = 2u * dir_field(byte, cm2);
      byte = win[cat];
      owner = claim[slot_of(cat)];
      if (++s == budget) return 2;
    }
  };

  // ---- 2. a pool of T walks in flight; a thread whose candidate is done takes the next one
  const uint32_t inline_steps = p.inline_steps ? p.inline_steps : (uint32_t)kInlineSteps;
  bool active = false;                      // this thread holds an unfinished candidate
  uint32_t rank = 0, at = 0, m2 = 0, done_len = 0;
  uint32_t n_iter = 0;
  bool stop_early = false;
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, tp = t_loaded;   // trace: cycles per phase (thread 0's clock)
#define SA_PHASE(k) do { if (p.trace) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[k] += now_ - tp; tp = now_; } } while (0)
  for (;;) {
    ++n_iter;
    // Top up the key ring one iteration ahead of its use (the load lands while this iteration runs): this
    // iteration hands out at most T keys from [nx, nx + T), the next one needs up to nx + 2T in the ring.
    const uint32_t top_up = min(n_cand, nx + 2u * T) - loaded;      // uniform; <= T
    KeyT pre = 0;
    if (tid < top_up) pre = keys[loaded + tid];
    for (uint32_t i = tid; i < nclaim; i += T) claim[i] = 0xffffffffu;
    {   // hand out candidates in rank order: one LDS atomic per wave
      const unsigned long long free_lanes = __ballot(!active);
      uint32_t base = 0;
      if (free_lanes) {
        if (lane == __builtin_ctzll(free_lanes)) base = atomicAdd(&s_next, (uint32_t)__popcll(free_lanes));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(free_lanes));
      }
      if (!active) {
        const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(free_lanes >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)free_lanes, 0u));
        if (idx < n_cand) {
          const KeyT key = ring[idx % (2u * T)];
          const int cscore = p.layout.cap - (int)(uint32_t)(key >> sshift);
          if (cscore >= min_score) {          // (always: the keys were emitted with this threshold)
            rank = idx; active = true; done_len = 0; m2 = 0;
            at = (((uint32_t)key & rmask) - r0 + 1u) * Wp + (((uint32_t)(key >> cshift) & cmask) - c0 + 1u);
          }
        }
      }
    }
    __syncthreads();
    SA_PHASE(0);

    // ---- claim: from where I stand to a marked cell / the end of the walk.  kInlineSteps cells here; a walk
    // that is longer goes into a queue and is finished by the first threads of the workgroup, so that the few
    // long walks of an iteration keep one or two waves busy instead of one lane in every wave.
    uint32_t plen = 0, qslot = 0xffffffffu;
    int cres = 0;                             // claim_walk result
    if (active) {
      uint32_t cat = at, cm2 = m2, byte = win[cat];
      if (byte & kVis) {
        active = false;                       // already marked (smith_waterman.c:269), or the walk ran into a mark
      } else {
        cres = claim_walk(rank, cat, cm2, byte, plen, inline_steps);
        if (cres == 2) {
          qslot = atomicAdd(&s_nq[0], 1u);
          if (qslot < (uint32_t)kQueueCap) { s_q0[qslot] = cat | (cm2 << 24); s_q1[qslot] = rank; s_q2[qslot] = plen; }
          else cres = claim_walk(rank, cat, cm2, byte, plen, 0xffffffffu);   // queue full: finish it here
        }
      }
    }
    __syncthreads();
    SA_PHASE(1);
    {
      const uint32_t nq = min(s_nq[0], (uint32_t)kQueueCap);
      if (tid < nq) {
        uint32_t cat = s_q0[tid] & 0xffffffu, cm2 = s_q0[tid] >> 24, n = s_q2[tid], byte = win[cat];
        const int r = claim_walk(s_q1[tid], cat, cm2, byte, n, 0xffffffffu);
        s_q2[tid] = n;
        s_q0[tid] = (uint32_t)r;
      }
      if (nq) __syncthreads();                // uniform: nq is the same for every thread
      if (qslot < (uint32_t)kQueueCap) { plen = s_q2[qslot]; cres = (int)s_q0[qslot]; }
    }
    if (active && cres == 3) atomicOr(&s_flag, 1u);
    __syncthreads();
    nx = s_next;                              // stable from here until the next hand-out
    if (tid == 0) s_nq[0] = 0;
    SA_PHASE(2);

    // ---- commit the prefix nobody of lower rank claims (same split: kInlineSteps here, long ones queued)
    uint32_t left_cells = plen;
    int mres = 0;
    qslot = 0xffffffffu;
    if (active) {
      mres = commit_walk(rank, at, m2, left_cells, inline_steps);
      if (mres == 2) {
        qslot = atomicAdd(&s_nq[1], 1u);
        if (qslot < (uint32_t)kQueueCap) { s_q0[qslot] = at | (m2 << 24); s_q1[qslot] = rank; s_q2[qslot] = left_cells; }
        else mres = commit_walk(rank, at, m2, left_cells, 0xffffffffu);
      }
    }
    __syncthreads();
    SA_PHASE(3);
    {
      const uint32_t nq = min(s_nq[1], (uint32_t)kQueueCap);
      if (tid < nq) {
        uint32_t cat = s_q0[tid] & 0xffffffu, cm2 = s_q0[tid] >> 24, left = s_q2[tid];
        const int r = commit_walk(s_q1[tid], cat, cm2, left, 0xffffffffu);
        s_q0[tid] = cat | (cm2 << 24);
        s_q2[tid] = left;
        s_q1[tid] = (uint32_t)r;
      }
      if (nq) __syncthreads();
      if (qslot < (uint32_t)kQueueCap) {
        at = s_q0[qslot] & 0xffffffu; m2 = s_q0[qslot] >> 24; left_cells = s_q2[qslot]; mres = (int)s_q1[qslot];
      }
    }
    if (active) {
      done_len += plen - left_cells;
      if (mres == 1) {                        // walked to the end of the claimed path: this candidate is done
        active = false;
        if (cres == 1) {                      // ... in a score-0 state: a hit (smith_waterman.c:217-255)
          const uint32_t slot = atomicAdd(&s_nhits, 1u);
          if (slot < (uint32_t)kMaxPoolHits) { s_hit_rank[slot] = rank; s_hit_steps[slot] = done_len - 1; }
          else atomicOr(&s_flag, 2u);
        }
      }
    }
    if (tid < top_up) ring[(loaded + tid) % (2u * T)] = pre;
    loaded += top_up;
    if (stop_early && active) {               // only once max_hits hits exist: who is still walking?
      uint32_t r = rank;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) r = min(r, (uint32_t)__shfl_xor((int)r, o));
      if (lane == __builtin_ctzll(__ballot(true))) atomicMin(&s_min_active, r);
    }
    const int left = __syncthreads_count(active);
    if (tid == 0) s_nq[1] = 0;
    SA_PHASE(4);
    if (s_flag) {                             // a walk left the window (retry with the largest) / too many hits in flight
      if (tid == 0) p.enum_status[pair] = (s_flag & 2u) || p.retry ? SA_ENUM_GENERIC : SA_ENUM_FALLBACK;
      return;
    }
    if (left == 0 && nx >= n_cand) break;
    if (s_nhits >= p.max_hits) {              // enough hits -- are the first max_hits of them final?
      if (stop_early) {
        const uint32_t horizon = min(s_min_active, nx);       // every rank below this is finished
        uint32_t below = 0;
        for (uint32_t h = 0; h < min(s_nhits, (uint32_t)kMaxPoolHits); ++h) below += s_hit_rank[h] < horizon;
        if (below >= p.max_hits) break;
      }
      stop_early = true;
      __syncthreads();
      if (tid == 0) s_min_active = 0xffffffffu;
    }
  }
  const unsigned long long t_walked = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;

  // ---- 3. the first max_hits hits in rank order: thread 0 orders them, thread k writes hit k
  __syncthreads();
  if (tid == 0) {
    const uint32_t nh = min(s_nhits, (uint32_t)kMaxPoolHits);
    for (uint32_t a = 1; a < nh; ++a) {       // insertion sort by rank (a handful of entries)
      const uint32_t r = s_hit_rank[a], st = s_hit_steps[a];
      uint32_t b = a;
      for (; b > 0 && s_hit_rank[b - 1] > r; --b) { s_hit_rank[b] = s_hit_rank[b - 1]; s_hit_steps[b] = s_hit_steps[b - 1]; }
      s_hit_rank[b] = r; s_hit_steps[b] = st;
    }
    const uint32_t ne = min(nh, p.max_hits);
    uint32_t off = 0;
    for (uint32_t a = 0; a < ne; ++a) { s_hit_off[a] = off; off += s_hit_steps[a]; }
    s_emit = ne;
    p.hit_count[pair] = ne;
    p.str_used[pair] = off;
    // stopped at max_hits with candidates left after the last reported hit?
    p.enum_status[pair] = (ne == p.max_hits && s_hit_rank[ne - 1] + 1 < n_cand) ? SA_ENUM_STOPPED_AT_MAX : 0u;
  }
  __syncthreads();
  if (tid < s_emit) {
    char *oa = p.out_a + p.str_off[pair];
    char *ob = p.out_b + p.str_off[pair];
    const KeyT key = keys[s_hit_rank[tid]];
    const uint32_t col = (uint32_t)(key >> cshift) & cmask, row = (uint32_t)key & rmask;
    const uint32_t steps = s_hit_steps[tid], soff = s_hit_off[tid];
    // replay from the candidate cell, writing the columns right to left
    uint32_t x = col, y = row, a2 = (row - r0 + 1u) * Wp + (col - c0 + 1u), hm2 = 0;   // hm2: 2 * matrix
    for (uint32_t k2 = steps; k2-- > 0;) {
      oa[soff + k2] = (hm2 == 2u) ? '-' : (char)v.seq_a[x - 1];   // 2 * GAP_A
      ob[soff + k2] = (hm2 == 4u) ? '-' : (char)v.seq_b[y - 1];   // 2 * GAP_B
      const uint32_t d = dir_field(win[a2], hm2);
      if (hm2 != 2u) --x;
      if (hm2 != 4u) --y;
      a2 -= back(hm2);
      hm2 = 2u * d;
    }
    SaDevHit h;
    h.score = p.layout.cap - (int)(uint32_t)(key >> sshift);
    h.pos_a = x; h.pos_b = y; h.len_a = col - x; h.len_b = row - y; h.length = steps; h.str_off = soff;
    (p.hits + (uint64_t)pair * p.max_hits)[tid] = h;
  }
  if (tid == 0 && p.trace) {
    unsigned long long *t = p.trace + 16ull * pair;
    t[0] = t_loaded - t_start; t[1] = t_walked - t_loaded; t[2] = n_iter; t[3] = 1;
    for (int k2 = 0; k2 < 5; ++k2) t[4 + k2] = ph[k2];
  }
}

template <int T>
static hipError_t launch_window(const SaEnumParams &p, size_t lds, hipStream_t stream) {
  hipError_t e;
  const uint32_t n = p.pair_list ? p.n_list : p.n_pairs;
  if (p.layout.key64) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_enumerate_window_kernel<T, unsigned long long>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sw_enumerate_window_kernel<T, unsigned long long>), dim3(n), dim3(T), lds, stream, p);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_enumerate_window_kernel<T, uint32_t>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sw_enumerate_window_kernel<T, uint32_t>), dim3(n), dim3(T), lds, stream, p);
  }
  return hipGetLastError();
}

}  // namespace sa

size_t sa_enum_window_lds_limit() { return sa::kWindowLdsLimit; }

// LDS = claim table + long-walk queue (12 B per thread) + ring of 2T keys + window.  Three workgroups per CU for small windows, two for medium ones,
// one for the rest; the last class trades threads and claim slots for the largest window that fits at all.
int sa_enum_classes(uint32_t key64, SaEnumClass out[4]) {
  const size_t kb = key64 ? 8 : 4, cu = 160u * 1024u, fixed = 2048;   // static LDS + slack per workgroup
  auto window = [&](size_t per_cu, uint32_t threads, uint32_t claim_bits) {
    const size_t total = std::min<size_t>(cu / per_cu - fixed, sa::kWindowLdsLimit);
    return (uint32_t)((total - ((size_t)4 << claim_bits) - 12 * threads - 2 * threads * kb) & ~(size_t)15u);
  };
  out[0] = SaEnumClass{256, 11, window(3, 256, 11)};
  out[1] = SaEnumClass{512, 11, window(2, 512, 11)};
  out[2] = SaEnumClass{1024, 12, window(1, 1024, 12)};
  out[3] = SaEnumClass{256, 10, window(1, 256, 10)};
  return 4;
}

static SaEnumParams with_env_overrides(const SaEnumParams &p_in) {
  SaEnumParams p = p_in;
  if (const char *env = getenv("SEQALIGN_ENUM_INLINE")) p.inline_steps = (uint32_t)std::max(1, atoi(env));   // tuning experiments
  if (const char *env = getenv("SEQALIGN_ENUM_WINDOW_BYTES")) {   // tests: a window too small for the walks -> fallback path
    const long v = atol(env);
    if (v >= 16) p.window_bytes = (uint32_t)std::min<long>(v, (long)p.window_bytes);
  }
  return p;
}

// Direction bytes of every window of one class: bands of R rows (bands beyond a pair's window return at once).
// Needs the fill's matrices and the candidates' boxes, NOT the sorted keys: the caller may run it next to the sort.
hipError_t sa_launch_sw_direction(const SaEnumParams &p_in, hipStream_t stream) {
  const SaEnumParams p = with_env_overrides(p_in);
  const uint32_t n = p.pair_list ? p.n_list : p.n_pairs;
  if (n == 0) return hipSuccess;
  const uint32_t wp_max = std::min<uint32_t>(p.max_len_a + 2u, p.window_bytes / 2u);
  const uint32_t hp_max = std::min<uint32_t>(p.max_len_b + 2u, p.window_bytes / 2u);
  const uint32_t R = std::max<uint32_t>(2u, std::min<uint32_t>(32u, sa::kDirTileBytes / (12u * wp_max + 2u)) ) - 1u;
  const size_t tile_lds = (size_t)12 * (R + 1) * wp_max + 2 * (wp_max + R + 2) + 16;
  const uint32_t bpp = (hp_max + R * sa::kDirBandsPerBlock - 1) / (R * sa::kDirBandsPerBlock);
  const uint32_t pairs_per_launch = std::max<uint32_t>(1u, 0x7fffffffu / bpp);
  if (tile_lds > 48u * 1024u) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa::sw_direction_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds);
    if (ea != hipSuccess) return ea;
  }
  for (uint32_t pair0 = 0; pair0 < n; pair0 += pairs_per_launch) {
    const uint32_t np = std::min(pairs_per_launch, n - pair0);
    hipLaunchKernelGGL(sa::sw_direction_kernel, dim3(np * bpp), dim3(sa::kDirThreads), tile_lds, stream, p, bpp, R, pair0);
  }
  return hipGetLastError();
}

// The enumeration of one class of pairs (p.pair_list / p.n_list, or all pairs); their direction bytes must be there
// (sa_launch_sw_direction with the same parameters).  p.threads / p.claim_bits / p.window_bytes: sa_enum_classes.
hipError_t sa_launch_sw_enumerate_window(const SaEnumParams &p_in, hipStream_t stream) {
  const SaEnumParams p = with_env_overrides(p_in);
  const uint32_t n = p.pair_list ? p.n_list : p.n_pairs;
  if (n == 0) return hipSuccess;
  const size_t lds = ((size_t)4 << p.claim_bits) + (size_t)12 * p.threads + (size_t)2 * p.threads * (p.layout.key64 ? 8 : 4) +
                     ((p.window_bytes + 15u) & ~(size_t)15u);
  if (p.threads == 256) return sa::launch_window<256>(p, lds, stream);
  if (p.threads == 512) return sa::launch_window<512>(p, lds, stream);
  return sa::launch_window<1024>(p, lds, stream);
}
