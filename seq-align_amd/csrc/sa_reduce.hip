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

struct Best {
  int score;
  unsigned col;
  unsigned idx;
};

__device__ __forceinline__ bool better(int s, unsigned col, unsigned idx, const Best &b) {
  if (s != b.score) return s > b.score;
  if (col != b.col) return col < b.col;
  return idx < b.idx;
}

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
  uint64_t *ckey = p.cand_key ? p.cand_key + p.cand_off[pair] : nullptr;

  Best best{0, 0, 0};                                   // cell 0 holds score 0
  uint32_t count = 0;                                   // wave-uniform

  // 4 independent 1 KiB loads per wave in flight per step: with one wave per pair
  // (4 k - 10 k waves) a single load per step leaves HBM latency exposed
  constexpr int kUnroll = 4;
  constexpr uint32_t kStep = kWave * 4 * kUnroll;
  for (uint32_t base = 0; base < cells; base += kStep) {
    int v[kUnroll][4];
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
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t i0 = base + u * (kWave * 4) + lane * 4;
      uint32_t mine = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (v[u][k] >= best.score && v[u][k] > 0) {
          const unsigned idx = i0 + k, col = idx % W;
          if (better(v[u][k], col, idx, best)) best = Best{v[u][k], col, idx};
        }
        mine += (v[u][k] >= min_score);
      }
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
              if (ckey) ckey[pos] = ((uint64_t)(uint32_t)(p.key_cap - v[u][k]) << p.key_shift) | ((i0 + k) % W);
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
    Best other{__shfl_xor(best.score, o), (unsigned)__shfl_xor((int)best.col, o),
               (unsigned)__shfl_xor((int)best.idx, o)};
    if (better(other.score, other.col, other.idx, best)) best = other;
  }
  if (lane == 0) {
    p.best_score[pair] = best.score;
    p.best_index[pair] = best.idx;
    if (p.cand_count) p.cand_count[pair] = count;
  }
}

}  // namespace sa

hipError_t sa_launch_sw_reduce(const SaReduceParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  const dim3 grid((p.n_pairs + sa::kWavesPerBlock - 1) / sa::kWavesPerBlock),
      block(sa::kWave * sa::kWavesPerBlock);
  hipLaunchKernelGGL(sa::sw_reduce_kernel, grid, block, 0, stream, p);
  return hipGetLastError();
}
