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
#include "sa_fill_common.hpp"

namespace sa {

// best cell so far; the column (idx % W, a ~25-instruction division) is only worked out
// when two cells tie on the score
struct Best {
  int score;
  unsigned idx;
};

__device__ __forceinline__ bool better(int s, unsigned idx, const Best &b, unsigned W) {
  if (s != b.score) return s > b.score;
  const unsigned col = idx % W, bcol = b.idx % W;
  if (col != bcol) return col < bcol;
  return idx < b.idx;
}

// COMPACT = false: best cell + number of cells >= min_score only (no positions: no ballots)
template <bool COMPACT>
__global__ void __launch_bounds__(kWave *kWavesPerBlock)
sw_reduce_kernel(const SaReduceParams p) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t pair = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
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
  // load-wait-compute loop)
  constexpr int kUnroll = 4;
  constexpr uint32_t kStep = kWave * 4 * kUnroll;
  auto load_step = [&](uint32_t base, int (&v)[kUnroll][4]) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t i0 = base + u * (kWave * 4) + lane * 4;
      if (i0 + 4 <= cells) {
        const v4i_u q = __builtin_nontemporal_load(reinterpret_cast<const v4i_u *>(M + i0));
        v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = (i0 + k < cells) ? M[i0 + k] : 0;
      }
    }
  };
  int nxt[kUnroll][4];
  load_step(0, nxt);
  for (uint32_t base = 0; base < cells; base += kStep) {
    int v[kUnroll][4];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = nxt[u][k];
    if (base + kStep < cells) load_step(base + kStep, nxt);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t i0 = base + u * (kWave * 4) + lane * 4;
      uint32_t mine = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // within a lane indices ascend, so a strictly higher score always wins; only a tie
        // (rare) needs the columns
        if (v[u][k] > best.score) best = Best{v[u][k], i0 + k};
        else if (v[u][k] == best.score && v[u][k] > 0 && better(v[u][k], i0 + k, best, W)) best = Best{v[u][k], i0 + k};
        mine += (v[u][k] >= min_score);
      }
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

// The candidates' count, bounding box and per-row column ranges (SaFillParams::cand_*) from a match_scores matrix
// that is already in HBM -- for fills that cannot report them while the values are in registers (anything but the
// stream kernel).  Same streaming loop as above: 4 B per cell read, one wave per pair; a candidate updates its row's
// range with two atomics (the lanes of a step straddle rows).
__global__ void __launch_bounds__(kWave *kWavesPerBlock)
sw_box_kernel(const SaReduceParams p, const SaCandBox c) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t pair = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (pair >= p.n_pairs) return;
  const uint32_t W = p.len_a[pair] + 1, H = p.len_b[pair] + 1;
  const uint32_t cells = W * H;
  const int32_t *__restrict__ M = p.M + p.mat_off[pair];
  uint32_t *rows = sa_cand_rows(c.cand_rows, p.mat_off[pair], W, H - 1);
  for (uint32_t r = lane; r < H; r += kWave) { rows[2 * r] = 0xffffffffu; rows[2 * r + 1] = 0u; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // (one wave: the atomics below follow in program order)
  const int thr = max(c.cand_min[pair], 1);
  uint32_t count = 0;                                                    // per lane, reduced at the end
  uint32_t rmin = 0xffffffffu, rmax = 0, cmin = 0xffffffffu, cmax = 0;   // per lane, reduced at the end
  constexpr uint32_t kStep = kWave * 4;
  for (uint32_t base = 0; base < cells; base += kStep) {
    const uint32_t i0 = base + lane * 4;
    int v[4];
    if (i0 + 4 <= cells) {
      const v4i_u q = __builtin_nontemporal_load(reinterpret_cast<const v4i_u *>(M + i0));
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (i0 + k < cells) ? M[i0 + k] : 0;
    }
    if (v[0] >= thr || v[1] >= thr || v[2] >= thr || v[3] >= thr) {
      uint32_t row = i0 / W, col = i0 - row * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (v[k] >= thr) {
          ++count;
          rmin = min(rmin, row); rmax = max(rmax, row); cmin = min(cmin, col); cmax = max(cmax, col);
          atomicMin(&rows[2 * row], col);
          atomicMax(&rows[2 * row + 1], col);
        }
        if (++col == W) { col = 0; ++row; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    count += (uint32_t)__shfl_xor((int)count, o);
    rmin = min(rmin, (uint32_t)__shfl_xor((int)rmin, o)); rmax = max(rmax, (uint32_t)__shfl_xor((int)rmax, o));
    cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, o)); cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, o));
  }
  if (lane == 0) {
    c.cand_count[pair] = count;
    uint32_t *box = c.cand_box + 4ull * pair;
    box[0] = rmin; box[1] = rmax; box[2] = cmin; box[3] = cmax;
  }
}

}  // namespace sa

hipError_t sa_launch_sw_box(const SaReduceParams &p, const SaCandBox &c, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  const dim3 grid((p.n_pairs + sa::kWavesPerBlock - 1) / sa::kWavesPerBlock), block(sa::kWave * sa::kWavesPerBlock);
  hipLaunchKernelGGL(sa::sw_box_kernel, grid, block, 0, stream, p, c);
  return hipGetLastError();
}

hipError_t sa_launch_sw_reduce(const SaReduceParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  const dim3 grid((p.n_pairs + sa::kWavesPerBlock - 1) / sa::kWavesPerBlock),
      block(sa::kWave * sa::kWavesPerBlock);
  if (p.cand_cap) hipLaunchKernelGGL(sa::sw_reduce_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(sa::sw_reduce_kernel<false>, grid, block, 0, stream, p);
  return hipGetLastError();
}
