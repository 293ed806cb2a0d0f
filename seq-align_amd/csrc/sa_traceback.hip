// sa_traceback.hip -- traceback on the device: global NW, and the
// best local SW hit.
//
// The reference re-derives each predecessor from the three score matrices with
// equality tests, priority GAP_A, GAP_B, MATCH (src/alignment.c:244-350), after
// choosing the end matrix at the bottom-right cell (src/needleman_wunsch.c:53-66).
// This kernel does exactly that on the matrices the fill kernel left in HBM, so
// an end-to-end batch returns O(len_a+len_b) characters per pair over PCIe
// instead of 12 B per DP cell (273 KB per 150x150 pair).
//
// One LANE per pair (64 pairs per wave): the walk is a chain of dependent loads
// (~len_a+len_b steps), so the only parallelism is across pairs.  Each lane
// writes its alignment right-to-left into its slot and reports where it starts;
// the host left-aligns while copying out of the staging buffer.
#include "sa_trace_common.hpp"

namespace sa {

// SW: start_index != nullptr -> local alignment ending at that match_scores cell
// (smith_waterman.c:165-258 on a fresh mask: the first fetched hit always
// succeeds), walked until the score reaches 0.
template <bool SW>
__global__ void __launch_bounds__(64) traceback_kernel(const SaTraceParams p) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= p.n_pairs) return;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair];
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const int32_t *__restrict__ Mg = p.M + mo;
  const int32_t *__restrict__ Ag = p.A + mo;
  const int32_t *__restrict__ Bg = p.B + mo;
  const uint32_t W = la + 1;
  char *oa = p.out_a + p.str_off[pair];
  char *ob = p.out_b + p.str_off[pair];

  const PairView v{sa_, sb_, Mg, Ag, Bg, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};

  int matrix = MAT_MATCH;
  int score;
  uint32_t x, y, head = la + lb, err = 0;
  if constexpr (SW) {
    const uint32_t end = (uint32_t)p.start_index[pair];
    x = end % W; y = end / W;
    score = Mg[end];
  } else {
    // end cell: ties resolve GAP_A > GAP_B > MATCH (needleman_wunsch.c:53-66)
    const uint32_t corner = W * (lb + 1) - 1;
    x = la; y = lb;
    score = Mg[corner];
    { const int b = Bg[corner]; if (b >= score) { matrix = MAT_GAP_B; score = b; } }
    { const int a = Ag[corner]; if (a >= score) { matrix = MAT_GAP_A; score = a; } }
  }
  p.out_score[pair] = score;
  const uint32_t end_x = x, end_y = y;

  while (SW ? (score > 0) : (x > 0 && y > 0)) {
    const uint8_t ca = sa_[x - 1], cb = sb_[y - 1];
    --head;
    oa[head] = (matrix == MAT_GAP_A) ? '-' : (char)ca;
    ob[head] = (matrix == MAT_GAP_B) ? '-' : (char)cb;
    if ((err = reverse_move(v, k, x, y, matrix, score))) break;
  }
  if constexpr (!SW) {
    if (!err) {
      for (; y > 0; --y) { --head; oa[head] = '-'; ob[head] = (char)sb_[y - 1]; }   // needleman_wunsch.c:117-123
      for (; x > 0; --x) { --head; oa[head] = (char)sa_[x - 1]; ob[head] = '-'; }   // :126-132
    }
  } else {
    // smith_waterman.c:251-255: start position and consumed lengths
    p.out_pos[4 * pair + 0] = x;
    p.out_pos[4 * pair + 1] = y;
    p.out_pos[4 * pair + 2] = end_x - x;
    p.out_pos[4 * pair + 3] = end_y - y;
  }
  (void)end_x; (void)end_y;
  p.out_head[pair] = head;
  p.out_len[pair] = la + lb - head;
  p.trace_status[pair] = err;
}

}  // namespace sa

hipError_t sa_launch_nw_traceback(const SaTraceParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  const dim3 grid((p.n_pairs + 63) / 64), block(64);   // one wave per workgroup: spread over all CUs
  if (p.start_index) hipLaunchKernelGGL(sa::traceback_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(sa::traceback_kernel<false>, grid, block, 0, stream, p);
  return hipGetLastError();
}
