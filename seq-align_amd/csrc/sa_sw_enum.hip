// sa_sw_enum.hip -- Smith-Waterman multi-hit enumeration on the device
// (SURVEY 8f-2).
//
// Reference semantics (src/smith_waterman.c:137-277): every cell with
// match_scores > 0 is a candidate; candidates are visited in the order
// (score desc, column asc [, index asc]); a candidate that is already marked is
// skipped, otherwise it is walked back to score 0 marking every cell on the way,
// and the walk is abandoned (its marks stay) as soon as it meets a marked cell; a
// completed walk is a hit.  The CLI stops at the first hit below min_score and
// after max_hits (sw_cmdline.c:214-217).
//
// Here: sa_reduce.hip compacts the cells >= min_score (ascending index) together
// with the sort key (INT_MAX - score) << 32 | column; a stable segmented radix sort
// (hipCUB -- not the hot path) orders each pair's candidates; then ONE LANE per
// pair runs the sequential enumeration against a per-pair visited bitmap in HBM
// (fresh = zeroed per call) and writes its hits' strings
// left-aligned into the pair's slot.  Latency-bound by construction (the reference
// algorithm is sequential per pair); the parallelism is across pairs.
#include <hipcub/hipcub.hpp>

#include "sa_trace_common.hpp"

namespace sa {

__global__ void __launch_bounds__(64) sw_enumerate_kernel(const SaEnumParams p) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= p.n_pairs) return;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const PairView v{p.arena + p.off_a[pair], p.arena + p.off_b[pair], p.M + mo, p.A + mo, p.B + mo, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  uint32_t *seen = p.mask + p.mask_off[pair];
  const uint64_t *keys = p.sorted_key + p.cand_off[pair];
  const uint32_t *cells = p.sorted_index + p.cand_off[pair];
  const uint32_t n_cand = p.cand_count[pair];
  const int min_score = p.min_score[pair];
  char *oa = p.out_a + p.str_off[pair];
  char *ob = p.out_b + p.str_off[pair];
  SaDevHit *hits = p.hits + (uint64_t)pair * p.max_hits;

  uint32_t emitted = 0, used = 0, err = 0, kpos = 0;
  bool exhausted = true;
  for (; kpos < n_cand; ++kpos) {
    if (emitted >= p.max_hits) { exhausted = false; break; }
    const int cscore = INT32_MAX - (int)(keys[kpos] >> 32);
    if (cscore < min_score) break;                     // sorted: nothing later qualifies
    const uint32_t end = cells[kpos];
    if ((seen[end >> 5] >> (end & 31)) & 1u) continue; // smith_waterman.c:269

    // pass 1 (:187-199): walk to score 0, marking; abandon on a marked cell
    uint32_t x = end % W, y = end / W, steps = 0;
    int matrix = MAT_MATCH, score = cscore;
    bool clash = false;
    for (;; ++steps) {
      const uint32_t at = y * W + x;
      const uint32_t word = seen[at >> 5], bit = 1u << (at & 31);
      if (word & bit) { clash = true; break; }
      seen[at >> 5] = word | bit;
      if (score == 0) break;
      if ((err = reverse_move(v, k, x, y, matrix, score))) break;
    }
    if (err) break;
    if (clash) continue;

    // pass 2 (:217-244): replay, writing the columns right to left
    x = end % W; y = end / W; matrix = MAT_MATCH; score = cscore;
    for (uint32_t w = steps; score > 0;) {
      --w;
      oa[used + w] = (matrix == MAT_GAP_A) ? '-' : (char)v.seq_a[x - 1];
      ob[used + w] = (matrix == MAT_GAP_B) ? '-' : (char)v.seq_b[y - 1];
      if ((err = reverse_move(v, k, x, y, matrix, score))) break;
    }
    if (err) break;
    SaDevHit h;                                         // smith_waterman.c:249-255
    h.score = cscore; h.pos_a = x; h.pos_b = y;
    h.len_a = end % W - x; h.len_b = end / W - y; h.length = steps; h.str_off = used;
    hits[emitted++] = h;
    used += steps;
  }
  p.hit_count[pair] = emitted;
  p.str_used[pair] = used;
  p.enum_status[pair] = err ? err : (exhausted ? 0u : 0x80000000u);   // top bit: stopped at max_hits
}

// strings of all pairs packed back to back for one D2H: one wave per pair
__global__ void __launch_bounds__(256) gather_strings_kernel(const char *src_a, const char *src_b,
                                                             const uint64_t *str_off, const uint32_t *used,
                                                             const uint64_t *dst_off, char *dst_a, char *dst_b,
                                                             uint32_t n_pairs) {
  const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x & 63;
  const uint32_t n = used[pair];
  const char *sa_ = src_a + str_off[pair], *sb_ = src_b + str_off[pair];
  char *da = dst_a + dst_off[pair], *db = dst_b + dst_off[pair];
  for (uint32_t i = lane; i < n; i += 64) { da[i] = sa_[i]; db[i] = sb_[i]; }
}

}  // namespace sa

hipError_t sa_sort_candidates(void *tmp, size_t *tmp_bytes, const uint64_t *key_in, uint64_t *key_out,
                              const uint32_t *idx_in, uint32_t *idx_out, uint64_t total, uint32_t n_pairs,
                              const uint64_t *seg_off /* n_pairs + 1 */, hipStream_t stream) {
  // stable LSD radix sort: equal (score, column) keep the compaction's ascending cell index
  return hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, *tmp_bytes, key_in, key_out, idx_in, idx_out,
                                                     (int)total, (int)n_pairs, seg_off, seg_off + 1, 0, 64, stream);
}

hipError_t sa_launch_sw_enumerate(const SaEnumParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(sa::sw_enumerate_kernel, dim3((p.n_pairs + 63) / 64), dim3(64), 0, stream, p);
  return hipGetLastError();
}

hipError_t sa_launch_gather_strings(const char *src_a, const char *src_b, const uint64_t *str_off,
                                    const uint32_t *used, const uint64_t *dst_off, char *dst_a, char *dst_b,
                                    uint32_t n_pairs, hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(sa::gather_strings_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, stream, src_a, src_b,
                     str_off, used, dst_off, dst_a, dst_b, n_pairs);
  return hipGetLastError();
}
