// sa_fill_stream.hip -- row-sweep fill writing through an LDS ring: every global
// store is an aligned, fully coalesced 1 KiB block.
//
// Replaces alignment_fill_matrices (reference src/alignment.c:28-168) for
// len_a <= 511; same arithmetic as sa_fill_rowscan.hip (sa_rowsweep.hpp).
//
// Why.  The reference layout is dense: pitch len_a+1 ints, so a pair's matrix is
// ONE contiguous run of (len_a+1)*(len_b+1) ints and a row-major sweep produces
// it strictly in order -- but rows start at arbitrary 4-byte offsets (604 B pitch
// at 150 columns).  Stored straight from registers (sa_fill_rowscan.hip) those
// rows cost one TA cycle per lane (TCP_TOTAL_WRITE = 64 per store instruction,
// 53 B per L2 request, waves parked in store issue 87 % of the time;
// profiles/r01_rowscan_c3.json) and the kernel ran at 2.9 TB/s while a plain
// fill_ reaches 6.1 TB/s on the same box.  So the wave treats each matrix as a
// byte stream: rows are appended to a per-wave LDS ring (ds_write_b32), and
// whenever 256 ints are complete they leave as ONE global_store_dwordx4 per
// lane, 64 lanes x 16 B, 1 KiB-aligned -- the same access pattern as a memset.
// Only the first and last block of a pair are partial (predicated dwords).
//
// Column 0 (the border column) is owned by lane 0 like any other column
// (RowSweep COL0 mode), so a row is exactly len_a+1 consecutive stream cells
// held by consecutive lanes and nothing is fed in from the left.
//
// LDS per wave: 3 rings x R ints (R = 512 for len_a <= 255, else 1024) = 6 / 12
// KiB; one wave per pair, 4 pairs per workgroup, no barrier after the table load.
#include "sa_rowsweep.hpp"

namespace sa {

constexpr int kBlockInts = 256;   // flush unit: 64 lanes x dwordx4 = 1 KiB

template <int R>
struct StreamOut {
  int32_t *ring;        // this wave's rings: M at 0, A at R, B at 2R (ints)
  int32_t *g0[3];       // matrix base minus a0 ints: g0 + v is 1 KiB aligned when v % 256 == 0
  uint32_t a0, vend;    // virtual range of the pair: [a0, vend)
  uint32_t wv, rv;      // virtual write / flush positions (rv % 256 == 0)

  __device__ __forceinline__ void flush_block(int lane) {
    const uint32_t ro = rv & (R - 1);
    const bool inside = (rv >= a0) && (rv + kBlockInts <= vend);   // wave-uniform
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const v4i_u q = *reinterpret_cast<const v4i_u *>(ring + m * R + ro + 4 * lane);   // ds_read_b128
      int32_t *dst = g0[m] + rv + 4 * lane;
      if (inside) {
        typedef int v4i_a __attribute__((ext_vector_type(4)));
        // aligned global_store_dwordx4 ... nt: write-once stream (see store_vec)
        __builtin_nontemporal_store(v4i_a{q.x, q.y, q.z, q.w}, reinterpret_cast<v4i_a *>(dst));
      } else {
        const uint32_t e = rv + 4 * lane;
        if (e + 0 >= a0 && e + 0 < vend) dst[0] = q.x;
        if (e + 1 >= a0 && e + 1 < vend) dst[1] = q.y;
        if (e + 2 >= a0 && e + 2 < vend) dst[2] = q.z;
        if (e + 3 >= a0 && e + 3 < vend) dst[3] = q.w;
      }
    }
    rv += kBlockInts;
  }

  // append one row: lane holds CPL consecutive cells starting at row position
  // lane*CPL.  ALL 64 lanes write, also those past the row's end: their cells land
  // at ring positions >= wv+W, which are not valid data yet (never flushed before
  // the next row overwrites them) and cannot reach back to unflushed cells because
  // 255 + 64*CPL <= R.  No per-lane predicate, no branch.
  template <int CPL>
  __device__ __forceinline__ void append_row(int lane, uint32_t W, const int (&mv)[CPL],
                                             const int (&av)[CPL], const int (&bv)[CPL]) {
    static_assert(kBlockInts - 1 + kWave * CPL <= R, "ring too small for unpredicated appends");
    const uint32_t at = wv + lane * CPL;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t i = (at + c) & (R - 1);
      ring[i] = mv[c];
      ring[R + i] = av[c];
      ring[2 * R + i] = bv[c];
    }
    wv += W;
    // reads below see the writes above: one wave, LDS ops execute in order; the
    // fence only stops the compiler from reordering them
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    while (wv - rv >= (uint32_t)kBlockInts) flush_block(lane);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }

  __device__ __forceinline__ void finish(int lane) {
    while (rv < wv) flush_block(lane);
  }
};

template <int CPL, int SUBST, bool GENERAL, int R>
__global__ void __launch_bounds__(kWave *kWavesPerBlock)
fill_stream_kernel(const SaFillParams p, const uint32_t table_ints) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int32_t *table = p.table;
  if constexpr (SUBST == SA_SUBST_LDS) {
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) lds[k] = p.table[k];
    __syncthreads();
    table = lds;
  }

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const uint32_t pair = blockIdx.x * kWavesPerBlock + wave;
  if (pair >= p.n_pairs) return;   // wave-uniform, after the only barrier

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair];
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const uint32_t W = la + 1;

  const SweepConsts k(p, table);
  const Border bd{p.floor, p.gap_open, p.ext, (p.flags & SA_F_IS_SW) != 0,
                  (p.flags & SA_F_NO_START_GAP) != 0};

  StreamOut<R> out;
  out.ring = lds + table_ints + wave * (3 * R);
  // the three arenas are congruent mod 1 KiB (checked on the host)
  out.a0 = (uint32_t)(((uintptr_t)(p.M + mo) >> 2) & (kBlockInts - 1));
  out.g0[0] = p.M + mo - out.a0;
  out.g0[1] = p.A + mo - out.a0;
  out.g0[2] = p.B + mo - out.a0;
  out.vend = out.a0 + W * (lb + 1);
  out.wv = out.a0;
  out.rv = 0;

  const uint32_t col0 = (uint32_t)(lane * CPL) - 1u;                 // matrix column lane*CPL + c
  const int ncol = max(0, min(CPL, (int)W - lane * CPL));

  RowSweep<CPL, SUBST, GENERAL, true> sw;
  sw.start_strip(p, k, bd, sa_, la, 0, col0);
  __builtin_amdgcn_s_waitcnt(kWaitVm0);   // seq_a codes landed (see RowFeed::load)

  {  // row 0 (reference alignment.c:46-69): (0,0) = 0; M = A = floor, B = edge
    int mv[CPL], av[CPL], bv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t ci = lane * CPL + c;
      mv[c] = av[c] = (ci == 0) ? 0 : k.floor_;
      bv[c] = (ci == 0) ? 0 : bd.edge_gap(ci);
    }
    out.template append_row<CPL>(lane, W, mv, av, bv);
  }

  int chunk_code = 0;
  for (uint32_t j = 1; j <= lb; ++j) {
    const int q = (j - 1) & (kWave - 1);
    if (q == 0) {   // every 64 rows: lane t fetches seq_b's code for row j+t
      const uint32_t r = j + lane;
      if (r <= lb) chunk_code = p.code[sb_[r - 1]];
      __builtin_amdgcn_s_waitcnt(kWaitVm0);   // see RowFeed::load
    }
    int mv[CPL], av[CPL], bv[CPL];
    sw.row(k, j, lb, la, W, lane, col0, ncol, read_lane(chunk_code, q), 0, 0, mv, av, bv, bd.edge_gap(j));
    out.template append_row<CPL>(lane, W, mv, av, bv);
  }
  out.finish(lane);

  const unsigned long long err = sw.reduce_err();
  if (lane == 0) p.status[pair] = err;
}

template <int CPL, int R>
static hipError_t launch_cpl(const SaFillParams &p, hipStream_t stream) {
  const bool general =
      p.flags & (SA_F_NO_END_GAP | SA_F_NO_GAPS_A | SA_F_NO_GAPS_B | SA_F_HAS_SENTINEL);
  const dim3 grid((p.n_pairs + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  size_t rings = (size_t)kWavesPerBlock * 3 * R * sizeof(int32_t);
  if (const char *env = getenv("SEQALIGN_LDS_PAD")) rings += (size_t)atoi(env);   // occupancy experiments
  if (p.K <= 1) {
    if (general) hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_SIMPLE, true, R>), grid, block, rings, stream, p, 0u);
    else hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_SIMPLE, false, R>), grid, block, rings, stream, p, 0u);
  } else if (p.K <= SA_LDS_TABLE_MAX_K) {
    const uint32_t tints = (p.K * p.K + 3u) & ~3u;   // keep the rings 16 B aligned
    const size_t lds = rings + tints * sizeof(int32_t);
    if (general) hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_LDS, true, R>), grid, block, lds, stream, p, tints);
    else hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_LDS, false, R>), grid, block, lds, stream, p, tints);
  } else {
    hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_GLOBAL, true, R>), grid, block, rings, stream, p, 0u);
  }
  return hipGetLastError();
}

}  // namespace sa

bool sa_stream_kernel_applicable(const SaFillParams &p, uint32_t max_len_a) {
  if (max_len_a + 1 > 8 * sa::kWave) return false;                 // a row must fit one wave
  const uintptr_t m = (uintptr_t)p.M, a = (uintptr_t)p.A, b = (uintptr_t)p.B;
  return ((m ^ a) & 1023) == 0 && ((m ^ b) & 1023) == 0;          // arenas congruent mod 1 KiB
}

hipError_t sa_launch_fill_stream(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  // columns per lane: the border column is a column too
  const uint32_t need = sa::columns_per_lane(max_len_a + 1);
  if (need <= 1) return sa::launch_cpl<1, 512>(p, stream);
  if (need <= 2) return sa::launch_cpl<2, 512>(p, stream);
  if (need <= 3) return sa::launch_cpl<3, 512>(p, stream);
  if (need <= 4) return sa::launch_cpl<4, 512>(p, stream);
  if (need <= 5) return sa::launch_cpl<5, 1024>(p, stream);
  if (need <= 6) return sa::launch_cpl<6, 1024>(p, stream);
  return sa::launch_cpl<8, 1024>(p, stream);
}
