// sa_fill_stream_cand.hip -- the stream fill kernel with candidate emission (SA_STREAM_CAND): its own
// translation unit so that the two halves of the template family compile in parallel.
#include "sa_fill_stream.hpp"

hipError_t sa_launch_fill_stream_cand(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  return sa::launch_stream_mode<sa::SA_STREAM_CAND>(p, max_len_a, stream);
}
