// sa_sort.hip -- per-pair sort of the Smith-Waterman candidate keys (hand-written; replaces the
// segmented radix sort of a library).
//
// The reference sorts every cell with match_scores > 0 by (score desc, column asc), stable from an
// index-ascending list (src/smith_waterman.c:152-161, comparator :71-86).  Here a
// candidate IS its sort key
//     key = (cap - score) << (row_bits + col_bits) | column << row_bits | row
// so ascending key order is (score desc, column asc, row asc) = the reference's order with the
// defined tie-break, keys are unique, and no payload travels with them.
//
// The keys arrive in row-major order (the fill emits them row by row, sa_fill_stream.hpp), i.e.
// ALREADY sorted by the row field.  A stable LSD radix sort therefore only has to process the column
// field and then the score field: two or three counting-sort passes of <= 10 bits instead of a
// full-width sort.  One 256-thread workgroup per pair; the pair's keys ping-pong between two global
// buffers (tens to a few hundred KiB per pair: L2 traffic), histograms and bases live in LDS.
//
// A pass, stable by construction:
//   1. wave w owns the contiguous quarter w of the list; it histograms its quarter into hist[w][*];
//   2. one exclusive scan over (digit, wave) turns the histograms into base[w][digit] -- where the
//      first key of that digit from wave w goes;
//   3. each wave walks its quarter IN ORDER, 64 keys per step: lanes with equal digits find each other
//      with one ballot per digit bit, a key's slot is base + (number of equal-digit lanes below it),
//      and the highest lane of each group advances the base.  Only wave w touches base[w][*], and LDS
//      operations of one wave execute in order, so no barrier is needed inside the walk.
#include "sa_fill_common.hpp"

namespace sa {

constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / kWave;
constexpr int kSortMaxBits = 10;

__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

template <typename KeyT>
__global__ void __launch_bounds__(kSortThreads) sort_keys_kernel(const SaSortParams p) {
  __shared__ uint32_t hist[kSortWaves << kSortMaxBits];   // [wave][digit]: counts, then bases
  __shared__ uint32_t wave_total[kSortWaves];
  const uint32_t pair = blockIdx.x;
  const uint32_t n = p.cand_count[pair];
  if (n < 2) {   // nothing to order; but an odd number of passes must still leave the key in `tmp`
    if (n == 1 && (p.n_passes & 1) && threadIdx.x == 0)
      static_cast<KeyT *>(p.tmp)[p.mat_off[pair]] = static_cast<const KeyT *>(p.keys)[p.mat_off[pair]];
    return;
  }
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid >> 6;
  KeyT *src = static_cast<KeyT *>(p.keys) + p.mat_off[pair];
  KeyT *dst = static_cast<KeyT *>(p.tmp) + p.mat_off[pair];
  // wave w's quarter: [beg, end), a multiple of 64 long except for the last
  const uint32_t q = ((n + kSortThreads - 1) / kSortThreads) * kWave;
  const uint32_t beg = min(n, (uint32_t)w * q), end = min(n, beg + q);

  for (uint32_t pass = 0; pass < p.n_passes; ++pass) {
    const uint32_t shift = p.shift[pass], bits = p.bits[pass], nb = 1u << bits, dmask = nb - 1;
    for (uint32_t i = tid; i < kSortWaves * nb; i += kSortThreads) hist[i] = 0;
    __syncthreads();
    uint32_t *mine = hist + w * nb;
    for (uint32_t i = beg + lane; i < end; i += kWave) atomicAdd(&mine[(uint32_t)(src[i] >> shift) & dmask], 1u);
    __syncthreads();

    // exclusive scan in (digit, wave) order: thread t owns digits [t*per, (t+1)*per)
    const uint32_t per = (nb + kSortThreads - 1) / kSortThreads;
    const uint32_t d0 = tid * per, d1 = min(nb, d0 + per);
    uint32_t sum = 0;
    for (uint32_t d = d0; d < d1; ++d)
#pragma unroll
      for (int u = 0; u < kSortWaves; ++u) sum += hist[u * nb + d];
    const uint32_t incl = wave_incl_scan_add(sum, lane);
    if (lane == kWave - 1) wave_total[w] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int u = 0; u < w; ++u) base += wave_total[u];
    for (uint32_t d = d0; d < d1; ++d)
#pragma unroll
      for (int u = 0; u < kSortWaves; ++u) {
        const uint32_t c = hist[u * nb + d];
        hist[u * nb + d] = base;
        base += c;
      }
    __syncthreads();

    // stable scatter of my quarter
    for (uint32_t tile = beg; tile < end; tile += kWave) {
      const uint32_t i = tile + lane;
      const bool valid = i < end;
      const KeyT key = valid ? src[i] : (KeyT)0;
      const uint32_t d = (uint32_t)(key >> shift) & dmask;
      unsigned long long peers = __ballot(valid);
      for (uint32_t b = 0; b < bits; ++b) {
        const bool one = (d >> b) & 1u;
        const unsigned long long bal = __ballot(one);
        peers &= one ? bal : ~bal;
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
      const uint32_t cnt = (uint32_t)__popcll(peers);
      const uint32_t at = mine[d];
      if (valid) {
        dst[at + rank] = key;
        if (rank == cnt - 1) mine[d] = at + cnt;   // after every lane's read above: one wave, LDS in order
      }
    }
    __syncthreads();   // also orders this pass's global writes before the next pass's reads (same workgroup)
    KeyT *t = src; src = dst; dst = t;
  }
}

}  // namespace sa

// passes over the column field, then the score field; each field split into equal digits of <= 10 bits
void sa_sort_plan(const SaKeyLayout &l, SaSortParams *p) {
  p->n_passes = 0;
  p->key64 = l.key64;
  const uint32_t field_lo[2] = {l.row_bits, l.row_bits + l.col_bits}, field_bits[2] = {l.col_bits, l.score_bits};
  for (int f = 0; f < 2; ++f) {
    const uint32_t bits = field_bits[f];
    if (!bits) continue;
    const uint32_t passes = (bits + sa::kSortMaxBits - 1) / sa::kSortMaxBits, each = (bits + passes - 1) / passes;
    for (uint32_t k = 0, done = 0; k < passes; ++k) {
      const uint32_t b = bits - done < each ? bits - done : each;
      p->shift[p->n_passes] = (uint8_t)(field_lo[f] + done);
      p->bits[p->n_passes] = (uint8_t)b;
      ++p->n_passes;
      done += b;
    }
  }
}

hipError_t sa_launch_sort_keys(const SaSortParams &p, hipStream_t stream) {
  if (p.n_pairs == 0 || p.n_passes == 0) return hipSuccess;
  if (p.key64) hipLaunchKernelGGL(sa::sort_keys_kernel<unsigned long long>, dim3(p.n_pairs), dim3(sa::kSortThreads), 0, stream, p);
  else hipLaunchKernelGGL(sa::sort_keys_kernel<uint32_t>, dim3(p.n_pairs), dim3(sa::kSortThreads), 0, stream, p);
  return hipGetLastError();
}
