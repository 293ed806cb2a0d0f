// sa_fill_stream.hip -- the LDS-ring stream writer (sa_fill_stream.hpp), plain and best-cell instantiations;
// the candidate-emitting instantiations are compiled in sa_fill_stream_cand.hip.
#include "sa_fill_stream.hpp"

hipError_t sa_launch_fill_stream_cand(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream);   // sa_fill_stream_cand.hip

bool sa_stream_kernel_applicable(const SaFillParams &p, uint32_t max_len_a) {
  if (max_len_a + 1 > 16 * sa::kWave) return false;                // a row must fit one wave (CPL <= 16)
  const uintptr_t m = (uintptr_t)p.M, a = (uintptr_t)p.A, b = (uintptr_t)p.B;
  return ((m ^ a) & 4095) == 0 && ((m ^ b) & 4095) == 0;          // arenas congruent mod 4 KiB
}

bool sa_stream_kernel_reports_best(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b) {
  return p.best_score && p.best_index && (p.flags & SA_F_IS_SW) && sa_stream_kernel_applicable(p, max_len_a) &&
         max_len_b < (1u << sa::kBestRowBits) && max_len_a + 1 < (1u << (32 - sa::kBestRowBits));
}

bool sa_stream_kernel_emits_candidates(const SaFillParams &p, uint32_t max_len_a) {
  return p.cand_count && p.cand_box && p.cand_rows && p.cand_rows_off && p.cand_min && (p.flags & SA_F_IS_SW) &&
         sa_stream_kernel_applicable(p, max_len_a);
}

hipError_t sa_launch_fill_stream(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  if (sa_stream_kernel_emits_candidates(p, max_len_a)) return sa_launch_fill_stream_cand(p, max_len_a, stream);
  // best-cell reporting: SW only (NW has no local maxima to report)
  if (p.best_score && p.best_index && (p.flags & SA_F_IS_SW)) return sa::launch_stream_mode<sa::SA_STREAM_BEST>(p, max_len_a, stream);
  return sa::launch_stream_mode<sa::SA_STREAM_PLAIN>(p, max_len_a, stream);
}
