// sa_reduce.hip -- Smith-Waterman local-maxima reduction / candidate compaction.
//
// Replaces the post-fill scan of the reference (src/smith_waterman.c:152-156:
// "collect every index with match_scores > 0", followed by a sort of ~80 % of
// all cells, :159-161) with one streaming pass over match_scores on the device:
//   * best cell per pair in the reference's hit order
//     (score desc, column asc -- smith_waterman.c:81-85 -- then index asc),
//   * every cell with score >= min_score (and > 0), compacted in ascending
//     index order, so the host only sorts the handful of cells that can still
//     become a reported hit (sw_cmdline.c:214-217 stops at the first hit below
//     min_score).
// One wave per pair: the candidate counter lives in a wave-uniform register, no
// atomics.  HBM-read bound: 4 B per cell.
#include <algorithm>
#include <type_traits>

#include "sa_fill_common.hpp"

namespace sa {

// best cell so far; the column (idx % W, a ~25-instruction division) is only worked out
// when two cells tie on the score
struct Best {
  int score;
  unsigned idx;
};
typedef int v4i_al __attribute__((ext_vector_type(4)));   // 16-byte aligned: the reduction's block loads

__device__ __forceinline__ bool better(int s, unsigned idx, const Best &b, unsigned W) {
  if (s != b.score) return s > b.score;
  const unsigned col = idx % W, bcol = b.idx % W;
  if (col != bcol) return col < bcol;
  return idx < b.idx;
}

// COMPACT = false: best cell + number of cells >= min_score only (no positions: no ballots)
template <bool COMPACT, int kUnroll>
__global__ void __launch_bounds__(kWave *kWavesPerBlock)
sw_reduce_kernel(const SaReduceParams p) {
  const int lane = threadIdx.x & (kWave - 1);
  // (wave-uniform, and SAID so: the pair's sizes and addresses then live in scalar registers)
  const uint32_t pair = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
  if (pair >= p.n_pairs) return;

  const uint32_t W = p.len_a[pair] + 1, H = p.len_b[pair] + 1;
  const uint32_t cells = W * H;
  const int32_t *__restrict__ M = p.M + p.mat_off[pair];
  const int min_score = max(p.min_score, 1);           // candidates need M > 0
  const uint32_t cap = p.cand_cap ? p.cand_cap[pair] : 0;
  uint32_t *cidx = p.cand_index ? p.cand_index + p.cand_off[pair] : nullptr;
  int32_t *cscore = p.cand_score ? p.cand_score + p.cand_off[pair] : nullptr;

  Best best{0, 0};                                      // cell 0 holds score 0
  uint32_t count = 0;                                   // COMPACT: wave-uniform; else per lane

  // kUnroll independent 1 KiB loads per wave per step, and the NEXT step's loads are issued
  // before this step's values are examined (software pipeline): with one wave per pair
  // (4 k - 10 k waves) anything less leaves HBM latency exposed (C4: 3.5 TB/s with a
  // load-wait-compute loop).
  // Every load of the wave is a whole ALIGNED 1 KiB block (64 lanes x 16 B), as the fill's stores are: a pair's matrix starts
  // wherever the pair before it ended (4-byte granularity), and a 1 KiB load that starts mid-line touches 17 lines of 64 B
  // instead of 16 -- so the stream starts at the 1 KiB boundary at or below the pair's first cell (`skew` cells earlier) and
  // the cells before the first / behind the last are masked out of the first and last block.
  // (kUnroll: 4; 8 is the option reduce_depth's experiment)
  constexpr uint32_t kStep = kWave * 4 * kUnroll;
  const uint32_t skew = (uint32_t)((reinterpret_cast<uintptr_t>(M) >> 2) & 255u);
  const int32_t *__restrict__ Mal = M - skew;            // 1 KiB aligned
  const uint32_t span = cells + skew;                    // block-space index v <-> cell v - skew
  auto load_step = [&](uint32_t base, int (&v)[kUnroll][4]) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t v0 = base + u * (kWave * 4) + lane * 4;
      if (v0 >= skew && v0 + 4 <= span) {
        const v4i_al q = __builtin_nontemporal_load(reinterpret_cast<const v4i_al *>(Mal + v0));
        v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = (v0 + k >= skew && v0 + k < span) ? Mal[v0 + k] : 0;
      }
    }
  };
  // The best cell in the reference's hit order -- score descending, then column ascending, then index ascending
  // (smith_waterman.c:71-86) -- as ONE 32-bit maximum per lane: key = score << col_bits | (2^col_bits - 1 - column), strictly
  // greater wins (a lane meets its cells in ascending index, so the first of equal keys stays).  The column of a lane's
  // vector follows from the previous one without a division (the stride between a lane's vectors is fixed: 256 cells, its
  // residue mod W is computed once per pair), and a vector whose largest score cannot beat the lane's key -- nearly every
  // vector once the lanes have seen a good cell -- costs its max and one compare.  (Until round 4 a tie on the SCORE went
  // through a division per cell, and with 64 lanes x 4 cells a tie somewhere in the wave is the rule, not the exception, on
  // a matrix of small scores: the kernel was bound by that branch, 0.72-0.76 of the HBM peak where the same access pattern
  // without arithmetic reads 0.85, tools/probes/read_probe.hip.)  Scores too large for the key's score field, or rows of
  // 2^20 columns and more, take the division (never on sequence data: the field holds scores up to 2^(31 - col_bits)).
  const uint32_t col_bits = 32u - (uint32_t)__builtin_clz(W | 1u);                 // columns 0 .. W - 1 fit
  const bool keyed = col_bits <= 20 && W >= 4;                                   // (W < 4: a vector wraps the row more than once)
  const uint32_t col_mask = (1u << col_bits) - 1u;
  const int score_lim = keyed ? (int)(0x7fffffffu >> col_bits) : 0;                // scores below this fit the key
  const uint32_t r256 = 256u % W, r1024 = kStep % W;                               // (wave-uniform: scalar divisions, once per pair)
  // column of this lane's first vector (block-space index lane * 4, cell lane * 4 - skew): cells before the pair's first are
  // masked, so any residue will do for them -- start from the cell index taken mod W in the unsigned wrap-free way
  uint32_t col_step0 = (uint32_t)(((uint64_t)lane * 4 + (uint64_t)W * 256u - skew) % W);   // (lane * 4 - skew) mod W; W * 256 >= skew
  uint32_t bkey = 0;                                                               // score 0, the highest column: nothing
  uint32_t bidx = 0;
  bool slow_seen = false;
  auto update_vector = [&](const int (&q)[4], uint32_t i0, uint32_t c0) __attribute__((always_inline)) {
    const int m4 = max(max(q[0], q[1]), max(q[2], q[3]));
    if (__builtin_expect(!keyed || __any(m4 >= score_lim), 0)) {   // a score the key cannot hold (or rows too wide for it): the division
      slow_seen = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (q[k] > best.score) best = Best{q[k], i0 + k};
        else if (q[k] == best.score && q[k] > 0 && better(q[k], i0 + k, best, W)) best = Best{q[k], i0 + k};
      }
      return;
    }
    // can any of the four beat the lane's key?  (its best possible key: the largest score in column 0)
    if (!__any((((uint32_t)m4 << col_bits) | col_mask) > bkey)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t c = c0 + k;
      c = c >= W ? c - W : c;
      const uint32_t key = ((uint32_t)q[k] << col_bits) | (col_mask - c);
      const bool wins = key > bkey && q[k] > 0;
      bkey = wins ? key : bkey;
      bidx = wins ? i0 + k : bidx;
    }
  };
  if constexpr (COMPACT) {
    int nxt[kUnroll][4];
    load_step(0, nxt);
    for (uint32_t base = 0; base < span; base += kStep) {
      int v[kUnroll][4];
  #pragma unroll
      for (int u = 0; u < kUnroll; ++u)
  #pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = nxt[u][k];
      if (base + kStep < span) load_step(base + kStep, nxt);
      uint32_t cu = col_step0;
  #pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const uint32_t i0 = base + u * (kWave * 4) + lane * 4 - skew;   // the cell (wraps below 0 for masked cells: their value is 0)
        uint32_t mine = 0;
  #pragma unroll
        for (int k = 0; k < 4; ++k) mine += (v[u][k] >= min_score);
        update_vector(v[u], i0, cu);
        cu += r256; cu = cu >= W ? cu - W : cu;
        if constexpr (!COMPACT) { count += mine; continue; }
        // exclusive prefix of `mine` over lanes (mine <= 4: three ballots of its bits)
        const unsigned long long b0 = __ballot(mine & 1), b1 = __ballot(mine & 2), b2 = __ballot(mine & 4);
        if ((b0 | b1 | b2) == 0) continue;                // wave-uniform: no candidate in this KiB
        const unsigned long long lt = (1ull << lane) - 1ull;
        uint32_t pos = count + __popcll(b0 & lt) + 2 * __popcll(b1 & lt) + 4 * __popcll(b2 & lt);
        count += __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
        if (mine) {
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (v[u][k] >= min_score) {
              if (pos < cap) {
                cidx[pos] = i0 + k;
                if (cscore) cscore[pos] = v[u][k];
              }
              ++pos;
            }
          }
        }
      }
      col_step0 += r1024; col_step0 = col_step0 >= W ? col_step0 - W : col_step0;
    }
  } else {
    // Round 5: the stream as an explicit three-deep pipeline.  The loop above asks for step s + 1 before it works on step s
    // (one step = kUnroll KiB in flight per wave, and none while the wave waits), its loads sit in branches (whole block /
    // edge cells), and it hands `nxt` to `v` through 16 register moves behind a vmcnt(0): 125 VGPRs, 0.80 (C3) / 0.725 (C4: 4 000
    // waves, four per SIMD) of the HBM peak where the bare access pattern reads 0.85.  Here every load is a whole aligned 1 KiB
    // block, unconditionally -- a 1 KiB-aligned block that holds one of the pair's cells lies in the same 4 KiB page as that
    // cell, so it is mapped whoever owns the rest of it; blocks past the pair's last are redirected to the last one and
    // masked out by index -- issued by inline asm into NB rotating buffers and claimed with an in-order `s_waitcnt
    // vmcnt((NB - 1) * kUnroll)`: two steps are always in flight while one is worked on (the compiler's own counting waits
    // vmcnt(0) at every loop header, see sa_sw_sweep.hip).
    constexpr int NB = 3;
    typedef int v4i_r __attribute__((ext_vector_type(4)));
    v4i_r buf[NB][kUnroll];
    const uint32_t last_blk = (span - 1u) & ~255u;       // block-space index of the pair's last 1 KiB block
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // (one asm statement per step: early-clobber outputs, and `s_nop 4` first -- a block address the compiler had to reload
    // from a spilled SGPR with v_readlane would otherwise be read by the load inside the five wait states gfx9 asks for after a
    // VALU write of an SGPR, and the compiler does not look into the statement: sa_sw_sweep.hip, SweepRow::request)
    auto request = [&](uint32_t base, v4i_r (&b)[kUnroll]) __attribute__((always_inline)) {
      static_assert(kUnroll == 4 || kUnroll == 8, "1 KiB blocks per step");
#pragma unroll
      for (int u = 0; u < kUnroll; u += 4) {
        const int32_t *b0 = Mal + min(base + (uint32_t)u * 256u, last_blk);   // (wave-uniform: scalar address arithmetic)
        const int32_t *b1 = Mal + min(base + (uint32_t)(u + 1) * 256u, last_blk);
        const int32_t *b2 = Mal + min(base + (uint32_t)(u + 2) * 256u, last_blk);
        const int32_t *b3 = Mal + min(base + (uint32_t)(u + 3) * 256u, last_blk);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5 nt\n\tglobal_load_dwordx4 %1, %4, %6 nt\n\t"
                     "global_load_dwordx4 %2, %4, %7 nt\n\tglobal_load_dwordx4 %3, %4, %8 nt"
                     : "=&v"(b[u]), "=&v"(b[u + 1]), "=&v"(b[u + 2]), "=&v"(b[u + 3])
                     : "v"(lane16), "s"(b0), "s"(b1), "s"(b2), "s"(b3));
      }
    };
    auto claim = [&](v4i_r (&b)[kUnroll], auto younger) __attribute__((always_inline)) {
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(decltype(younger)::value) : "memory");
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) asm volatile("" : "+v"(b[u]));
    };
    auto work = [&](uint32_t base, const v4i_r (&b)[kUnroll]) __attribute__((always_inline)) {
      const bool edge = base < skew || base + kStep > span;   // (wave-uniform) the step holds cells that are not the pair's
      int q[kUnroll][4];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        q[u][0] = b[u].x; q[u][1] = b[u].y; q[u][2] = b[u].z; q[u][3] = b[u].w;
        if (edge) {
          const uint32_t v0 = base + u * (kWave * 4) + lane * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) q[u][k] = (v0 + k >= skew && v0 + k < span) ? q[u][k] : 0;
        }
      }
      // cells >= min_score, and the step's largest score: straight-line
      int top = q[0][0];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) { count += (q[u][k] >= min_score); top = max(top, q[u][k]); }
      // can any of the step's cells beat its lane's key (its best possible key: the largest score in column 0), or does a score
      // not fit the key?  ONE test per step -- nearly always "no" once the lanes have seen a good cell -- where the first form
      // asked twice per vector
      const bool look = !keyed || top >= score_lim || (((uint32_t)top << col_bits) | col_mask) > bkey;
      if (__any(look)) {
        uint32_t cu = col_step0;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          update_vector(q[u], base + u * (kWave * 4) + lane * 4 - skew, cu);
          cu += r256; cu = cu >= W ? cu - W : cu;
        }
      }
      col_step0 += r1024; col_step0 = col_step0 >= W ? col_step0 - W : col_step0;
    };
#pragma unroll
    for (int b = 0; b < NB; ++b) request((uint32_t)b * kStep, buf[b]);
    uint32_t base = 0;
    bool done = false;
    while (!done) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        claim(buf[b], std::integral_constant<int, (NB - 1) * kUnroll>());
        work(base, buf[b]);
        base += kStep;
        if (base >= span) { done = true; break; }
        request(base + (NB - 1) * kStep, buf[b]);   // (beyond the pair's last block: that block again, never looked at)
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) claim(buf[b], std::integral_constant<int, 0>());   // nothing may still be on its way into registers
  }
  // the lane's keyed best against what the division path may have found (both in hit order)
  if (bkey != 0) {
    const Best kb{(int)(bkey >> col_bits), bidx};
    if (!slow_seen || better(kb.score, kb.idx, best, W)) best = kb;
  }

  // wave reduction of the best cell
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const Best other{__shfl_xor(best.score, o), (unsigned)__shfl_xor((int)best.idx, o)};
    if (better(other.score, other.idx, best, W)) best = other;
    if constexpr (!COMPACT) count += __shfl_xor((int)count, o);
  }
  if (lane == 0) {
    p.best_score[pair] = best.score;
    p.best_index[pair] = best.idx;
    if (p.cand_count) p.cand_count[pair] = count;
  }
}

// ---- the best cell only, FEW LONG pairs: a pair's cells in slices, one wave per slice.
// One wave per pair streams a 10 000 x 10 000 pair's 400 MB alone (100 ms); here the waves of a pair merge their best
// cells with one 64-bit atomicMin on a key whose order is the reference's hit order:
//     (INT_MAX - score) << 32 | column << row_bits | row        (column and row share 32 bits: cells < 2^31)
__global__ void __launch_bounds__(256) sw_best_init_kernel(uint64_t *best_index, uint32_t n_pairs) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair < n_pairs) best_index[pair] = ~0ull;
}

__global__ void __launch_bounds__(kWave) sw_best_slices_kernel(const SaReduceParams p, const uint32_t slices) {
  const int lane = threadIdx.x;
  const uint32_t pair = blockIdx.x;
  const uint32_t W = p.len_a[pair] + 1, H = p.len_b[pair] + 1, cells = W * H;
  constexpr uint32_t kStep = kWave * 4;
  const uint32_t per = ((cells + slices - 1) / slices + kStep - 1) / kStep * kStep;
  const uint32_t lo = blockIdx.y * per, hi = min(cells, lo + per);
  if (lo >= cells) return;
  const int32_t *__restrict__ M = p.M + p.mat_off[pair];
  Best best{0, 0};
  for (uint32_t base = lo; base < hi; base += kStep) {
    const uint32_t i0 = base + lane * 4;
    int v[4];
    if (i0 + 4 <= cells) {
      const v4i_u q = __builtin_nontemporal_load(reinterpret_cast<const v4i_u *>(M + i0));
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (i0 + k < cells) ? M[i0 + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (v[k] > best.score) best = Best{v[k], i0 + k};
      else if (v[k] == best.score && v[k] > 0 && better(v[k], i0 + k, best, W)) best = Best{v[k], i0 + k};
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const Best other{__shfl_xor(best.score, o), (unsigned)__shfl_xor((int)best.idx, o)};
    if (better(other.score, other.idx, best, W)) best = other;
  }
  if (lane == 0 && best.score > 0) {
    const uint32_t row = best.idx / W, col = best.idx - row * W, row_bits = 32u - (uint32_t)__builtin_clz(H | 1u);
    const unsigned long long key = ((unsigned long long)(uint32_t)(INT32_MAX - best.score) << 32) | ((unsigned long long)col << row_bits) | row;
    atomicMin(reinterpret_cast<unsigned long long *>(p.best_index + pair), key);
  }
}

__global__ void __launch_bounds__(256) sw_best_finish_kernel(const SaReduceParams p) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= p.n_pairs) return;
  const unsigned long long key = p.best_index[pair];
  int score = 0;
  uint64_t index = 0;
  if (key != ~0ull) {
    const uint32_t W = p.len_a[pair] + 1, H = p.len_b[pair] + 1, row_bits = 32u - (uint32_t)__builtin_clz(H | 1u);
    const uint32_t low = (uint32_t)key, row = low & ((1u << row_bits) - 1u), col = row_bits < 32u ? low >> row_bits : 0u;
    score = INT32_MAX - (int)(uint32_t)(key >> 32);
    index = (uint64_t)row * W + col;
  }
  p.best_score[pair] = score;
  p.best_index[pair] = index;
}

// The candidates' count, bounding box and per-row column ranges (SaFillParams::cand_*) from a match_scores matrix
// that is already in HBM -- for fills that cannot report them while the values are in registers (anything but the
// stream kernel: rows over 1 023 columns).  One wave per block of 64 rows of a pair, row by row: 4 B per cell
// read; a row's lowest / highest candidate column from two ballots; the pair's count and box with a few atomics per
// wave.
constexpr uint32_t kBoxRows = 64;

__global__ void __launch_bounds__(256) sw_box_init_kernel(const SaCandBox c, uint32_t n_pairs) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= n_pairs) return;
  c.cand_count[pair] = 0;
  uint32_t *box = c.cand_box + 4ull * pair;
  box[0] = 0xffffffffu; box[1] = 0u; box[2] = 0xffffffffu; box[3] = 0u;
}

__global__ void __launch_bounds__(kWave) sw_box_kernel(const SaReduceParams p, const SaCandBox c, const uint32_t rows_per_block) {
  const int lane = threadIdx.x;
  const uint32_t pair = blockIdx.x;
  const uint32_t W = p.len_a[pair] + 1, H = p.len_b[pair] + 1;
  const uint32_t row0 = blockIdx.y * rows_per_block;
  if (row0 >= H) return;
  const int32_t *__restrict__ M = p.M + p.mat_off[pair];
  uint32_t *rows = c.cand_rows + 2ull * c.hit_off[pair];
  const int thr = max(c.cand_min[pair], 1);
  uint32_t count = 0, rmin = 0xffffffffu, rmax = 0, cmin = 0xffffffffu, cmax = 0;   // wave-uniform
  for (uint32_t r = row0; r < min(row0 + rows_per_block, H); ++r) {
    const int32_t *row = M + (size_t)r * W;
    uint32_t lo = 0xffffffffu, hi = 0;
    for (uint32_t base = 0; base < W; base += kWave * 4) {
      const uint32_t x = base + lane * 4;
      int v[4] = {0, 0, 0, 0};
      if (x + 4 <= W) {
        const v4i_u q = __builtin_nontemporal_load(reinterpret_cast<const v4i_u *>(row + x));
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (x + k < W) ? row[x + k] : 0;
      }
      const uint32_t bits = (v[0] >= thr) | ((v[1] >= thr) << 1) | ((v[2] >= thr) << 2) | ((v[3] >= thr) << 3);
      const unsigned long long any = __ballot(bits != 0);
      if (any) {   // columns grow with the lane: the row's lowest candidate is in the lowest such lane
        const int first = __builtin_ctzll(any), last = 63 - __builtin_clzll(any);
        const uint32_t bf = (uint32_t)__builtin_amdgcn_readlane((int)bits, first), bl = (uint32_t)__builtin_amdgcn_readlane((int)bits, last);
        lo = min(lo, base + first * 4 + (uint32_t)__builtin_ctz(bf));
        hi = max(hi, base + last * 4 + (31u - (uint32_t)__builtin_clz(bl)));
        // (the count: 4 ballots would do; a per-lane popcount and one reduction at the end is cheaper)
      }
      count += (uint32_t)__popc(bits);   // per lane here, reduced below
    }
    if (lane == 0) { rows[2 * r] = lo; rows[2 * r + 1] = hi; }
    if (lo <= hi) { rmin = min(rmin, r); rmax = r; cmin = min(cmin, lo); cmax = max(cmax, hi); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) count += (uint32_t)__shfl_xor((int)count, o);
  if (lane == 0 && count) {
    atomicAdd(&c.cand_count[pair], count);
    uint32_t *box = c.cand_box + 4ull * pair;
    atomicMin(&box[0], rmin); atomicMax(&box[1], rmax); atomicMin(&box[2], cmin); atomicMax(&box[3], cmax);
  }
}

}  // namespace sa

hipError_t sa_launch_sw_box(const SaReduceParams &p, const SaCandBox &c, uint32_t max_len_b, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  sa_record_launch(SEQALIGN_K_SW_BOX, p.n_pairs);
  hipLaunchKernelGGL(sa::sw_box_init_kernel, dim3((p.n_pairs + 255) / 256), dim3(256), 0, stream, c, p.n_pairs);
  const uint32_t rows_per_block = std::max<uint32_t>(sa::kBoxRows, (max_len_b + 65535u) / 65535u);   // (grid.y limit)
  const uint32_t row_blocks = (max_len_b + rows_per_block) / rows_per_block;
  for (uint32_t first = 0; first < p.n_pairs; first += 32768) {   // (grid.x, grid.y limits)
    SaReduceParams q = p;
    SaCandBox d = c;
    const uint32_t cnt = p.n_pairs - first < 32768u ? p.n_pairs - first : 32768u;
    q.len_a += first; q.len_b += first; q.mat_off += first; q.n_pairs = cnt;
    d.cand_count += first; d.cand_box += 4ull * first; d.cand_min += first; d.hit_off += first;
    hipLaunchKernelGGL(sa::sw_box_kernel, dim3(cnt, row_blocks), dim3(sa::kWave), 0, stream, q, d, rows_per_block);
  }
  return hipGetLastError();
}

hipError_t sa_launch_sw_reduce(const SaReduceParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  sa_record_launch(SEQALIGN_K_SW_REDUCE, p.n_pairs);
  if (p.slices > 1 && !p.cand_count && !p.cand_cap && p.n_pairs <= 32768u) {   // few long pairs, best cell only
    hipLaunchKernelGGL(sa::sw_best_init_kernel, dim3((p.n_pairs + 255) / 256), dim3(256), 0, stream, p.best_index, p.n_pairs);
    hipLaunchKernelGGL(sa::sw_best_slices_kernel, dim3(p.n_pairs, p.slices), dim3(sa::kWave), 0, stream, p, p.slices);
    hipLaunchKernelGGL(sa::sw_best_finish_kernel, dim3((p.n_pairs + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
  }
  const dim3 grid((p.n_pairs + sa::kWavesPerBlock - 1) / sa::kWavesPerBlock),
      block(sa::kWave * sa::kWavesPerBlock);
  // (8 KiB per wave and step only on request: it loses on both configurations -- C3 0.807 against 0.814, C4's 4 000 waves 0.726
  // against 0.771, tools/reduce_bench.py)
  const bool deep = p.tune_depth == 8;
  if (p.cand_cap) hipLaunchKernelGGL((sa::sw_reduce_kernel<true, 4>), grid, block, 0, stream, p);
  else if (deep) hipLaunchKernelGGL((sa::sw_reduce_kernel<false, 8>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((sa::sw_reduce_kernel<false, 4>), grid, block, 0, stream, p);
  return hipGetLastError();
}
