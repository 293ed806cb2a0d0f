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
// A pass, stable by construction, and with global stores that are runs of equal-digit keys:
//   1. histogram of the whole list -> running global base per digit (exclusive scan);
//   2. the list is taken in tiles of 4 096 keys, in order.  Wave w owns the contiguous quarter w of the tile and
//      keeps its 16 x 64 keys in registers; per-wave histograms + one exclusive scan over (digit, wave) give
//      every (wave, digit) its place INSIDE the tile; each wave places its keys into an LDS copy of the tile,
//      64 at a time in order: lanes with equal digits find each other with one ballot per digit bit, a key's slot
//      is base + (number of equal-digit lanes below it), the highest lane of a group advances the base (only
//      wave w touches base[w][*], and LDS operations of one wave execute in order: no barrier inside the walk);
//   3. the LDS tile is now grouped by digit: thread e copies tile[e] to global_base[digit] + (e - start of the
//      digit's run in the tile) -- consecutive threads, consecutive addresses -- and the global bases advance.
//      (Scattering straight from registers wrote 4-byte pieces into 500+ places per pair; with thousands of pairs in
//      flight that is far more than L2 holds, every piece became a 64-byte read-modify-write in HBM:
//      profiles/r02/swenum_*: 3.2 ms for 155 M keys, 7.2 ms for 223 M.)
#include "sa_fill_common.hpp"

namespace sa {

constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / kWave;
constexpr int kSortMaxBits = 10;
constexpr int kSortPerLane = 16;                                   // keys per lane per tile
constexpr int kSortTile = kSortThreads * kSortPerLane;             // 4 096 keys
constexpr int kSortQuarter = kSortTile / kSortWaves;               // 1 024 keys per wave

__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan of `nb` counters in LDS, in place, by the whole workgroup; counts[nb] is left untouched
__device__ __forceinline__ void block_excl_scan(uint32_t *counts, uint32_t nb, uint32_t *wave_total) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid >> 6;
  const uint32_t per = (nb + kSortThreads - 1) / kSortThreads;
  const uint32_t d0 = min(nb, tid * per), d1 = min(nb, d0 + per);
  uint32_t sum = 0;
  for (uint32_t d = d0; d < d1; ++d) sum += counts[d];
  const uint32_t incl = wave_incl_scan_add(sum, lane);
  if (lane == kWave - 1) wave_total[w] = incl;
  __syncthreads();
  uint32_t base = incl - sum;
  for (int u = 0; u < w; ++u) base += wave_total[u];
  for (uint32_t d = d0; d < d1; ++d) {
    const uint32_t c = counts[d];
    counts[d] = base;
    base += c;
  }
  __syncthreads();
}

template <typename KeyT>
__global__ void __launch_bounds__(kSortThreads) sort_keys_kernel(const SaSortParams p) {
  constexpr uint32_t NBmax = 1u << kSortMaxBits;
  __shared__ KeyT tile[kSortTile];
  __shared__ uint32_t hist[kSortWaves * NBmax];   // [wave][digit]: counts, then places inside the tile
  __shared__ uint32_t gbase[NBmax];               // where the next key of a digit goes in the output
  __shared__ uint32_t tstart[NBmax + 1];          // start of a digit's run inside the tile
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

  for (uint32_t pass = 0; pass < p.n_passes; ++pass) {
    const uint32_t shift = p.shift[pass], bits = p.bits[pass], nb = 1u << bits, dmask = nb - 1;
    // ---- 1. global histogram -> gbase
    for (uint32_t i = tid; i < nb; i += kSortThreads) gbase[i] = 0;
    __syncthreads();
    for (uint32_t i0 = tid; i0 < n; i0 += kSortThreads * 8) {      // 8 independent loads in flight per lane
      KeyT k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) k[u] = (i0 + u * kSortThreads < n) ? src[i0 + u * kSortThreads] : (KeyT)0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * kSortThreads < n) atomicAdd(&gbase[(uint32_t)(k[u] >> shift) & dmask], 1u);
    }
    __syncthreads();
    block_excl_scan(gbase, nb, wave_total);

    // ---- 2./3. tiles
    uint32_t *mine = hist + w * nb;
    for (uint32_t t0 = 0; t0 < n; t0 += kSortTile) {
      const uint32_t tn = min((uint32_t)kSortTile, n - t0);
      const uint32_t qb = t0 + w * kSortQuarter;                   // my wave's quarter starts here
      for (uint32_t i = tid; i < kSortWaves * nb; i += kSortThreads) hist[i] = 0;
      KeyT k[kSortPerLane];
#pragma unroll
      for (int u = 0; u < kSortPerLane; ++u) {
        const uint32_t i = qb + u * kWave + lane;
        k[u] = (i < t0 + tn) ? src[i] : (KeyT)0;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kSortPerLane; ++u)
        if (qb + u * kWave + lane < t0 + tn) atomicAdd(&mine[(uint32_t)(k[u] >> shift) & dmask], 1u);
      __syncthreads();
      {   // exclusive scan in (digit, wave) order: places inside the tile; tstart[d] = where digit d's run begins
        const uint32_t per = (nb + kSortThreads - 1) / kSortThreads;
        const uint32_t d0 = min(nb, tid * per), d1 = min(nb, d0 + per);
        uint32_t sum = 0;
        for (uint32_t d = d0; d < d1; ++d)
#pragma unroll
          for (int u = 0; u < kSortWaves; ++u) sum += hist[u * nb + d];
        const uint32_t incl = wave_incl_scan_add(sum, lane);
        if (lane == kWave - 1) wave_total[w] = incl;
        __syncthreads();
        uint32_t base = incl - sum;
        for (int u = 0; u < w; ++u) base += wave_total[u];
        for (uint32_t d = d0; d < d1; ++d) {
          tstart[d] = base;
#pragma unroll
          for (int u = 0; u < kSortWaves; ++u) {
            const uint32_t c = hist[u * nb + d];
            hist[u * nb + d] = base;
            base += c;
          }
        }
        if (tid == 0) tstart[nb] = tn;
        __syncthreads();
      }
      // my quarter into the LDS tile, 64 keys at a time, in order
#pragma unroll
      for (int u = 0; u < kSortPerLane; ++u) {
        if (qb + u * kWave >= t0 + tn) break;                      // wave-uniform
        const bool valid = qb + u * kWave + lane < t0 + tn;
        const uint32_t d = (uint32_t)(k[u] >> shift) & dmask;
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
          tile[at + rank] = k[u];
          if (rank == cnt - 1) mine[d] = at + cnt;                 // after every lane's read above: one wave, LDS in order
        }
      }
      __syncthreads();
      // the tile, grouped by digit, to its places in the output: runs of consecutive addresses
      for (uint32_t e = tid; e < tn; e += kSortThreads) {
        const KeyT key = tile[e];
        const uint32_t d = (uint32_t)(key >> shift) & dmask;
        dst[gbase[d] + (e - tstart[d])] = key;
      }
      __syncthreads();
      for (uint32_t d = tid; d < nb; d += kSortThreads) gbase[d] += tstart[d + 1] - tstart[d];
      __syncthreads();
    }
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
