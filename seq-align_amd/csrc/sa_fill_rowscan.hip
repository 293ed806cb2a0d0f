// sa_fill_rowscan.hip -- row-sweep fill, direct row stores.
//
// Replaces alignment_fill_matrices (reference src/alignment.c:28-168); same
// results as sa_fill_wavefront.hip, different schedule, chosen for the store
// side: the fill is bound by HBM writes (12 B per cell, three int32 matrices
// with the reference's dense pitch), and here every store instruction writes
// one whole matrix row segment -- lane l holds columns l*CPL+1 .. +CPL of the
// row, so the 64 lanes' dwordx3/x4 stores are back to back in memory (600 B
// per instruction at 150 columns) instead of 64 different rows.  One wave per
// pair, one matrix row per step, no skew, no barrier, no LDS.  The arithmetic
// (incl. the (max,+) prefix scan for gap_b) lives in sa_rowsweep.hpp.
//
// This kernel handles every shape (len_a > 512 in column strips).  For
// len_a <= 512 sa_fill_stream.hip does the same sweep but writes through an LDS
// ring in aligned 1 KiB blocks, which is what the profile asked for: here the
// rows start at arbitrary 4-byte offsets (pitch len_a+1 ints cannot be padded),
// the TA handles such stores lane by lane (TCP_TOTAL_WRITE = 64 per store
// instruction, profiles/r01_rowscan_c3.json) and the waves sit in store issue.
#include "sa_rowsweep.hpp"

namespace sa {

template <int CPL, int SUBST, bool GENERAL>
__global__ void __launch_bounds__(kWave *kWavesPerBlock)
fill_rowscan_kernel(const SaFillParams p) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds_table[];
  const int32_t *table = p.table;
  if constexpr (SUBST == SA_SUBST_LDS) {
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) lds_table[k] = p.table[k];
    __syncthreads();
    table = lds_table;
  }

  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t pair = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (pair >= p.n_pairs) return;   // wave-uniform

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair];
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  int32_t *__restrict__ Mg = p.M + mo;
  int32_t *__restrict__ Ag = p.A + mo;
  int32_t *__restrict__ Bg = p.B + mo;
  const uint32_t W = la + 1;

  const SweepConsts k(p, table);
  const Border bd{p.floor, p.gap_open, p.ext, (p.flags & SA_F_IS_SW) != 0,
                  (p.flags & SA_F_NO_START_GAP) != 0};

  // ---- borders (reference alignment.c:46-81)
  for (uint32_t i = lane; i <= la; i += kWave) {
    const int fl = (i == 0) ? 0 : k.floor_;
    Mg[i] = fl;
    Ag[i] = fl;
    Bg[i] = (i == 0) ? 0 : bd.edge_gap(i);
  }
  for (uint32_t j = 1 + lane; j <= lb; j += kWave) {
    const size_t c = (size_t)j * W;
    Mg[c] = k.floor_;
    Ag[c] = bd.edge_gap(j);
    Bg[c] = k.floor_;
  }

  RowSweep<CPL, SUBST, GENERAL> sw;
  constexpr uint32_t kStrip = kWave * CPL;

  for (uint32_t i0 = 0; i0 < la; i0 += kStrip) {
    const uint32_t cols = min(kStrip, la - i0);
    const uint32_t col0 = i0 + lane * CPL;
    const int ncol = max(0, min(CPL, (int)cols - lane * CPL));
    sw.start_strip(p, k, bd, sa_, la, i0, col0, lane);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);   // seq_a codes landed (see RowFeed::load)

    RowFeed feed;
    uint32_t off = W + col0 + 1;            // (row 1, my first column)
    for (uint32_t j = 1; j <= lb; ++j, off += W) {
      const int q = (j - 1) & (kWave - 1);
      if (q == 0) feed.load(p, k, bd, sb_, lb, W, i0, Mg, Ag, Bg, j + lane);
      int mv[CPL], av[CPL], bv[CPL];
      sw.row(k, j, lb, la, W, lane, col0, ncol, read_lane(feed.code, q), read_lane(feed.Z, q),
             read_lane(feed.B, q), mv, av, bv);
      if (ncol == CPL) {
        store_run<CPL, true>(Mg + off, mv);
        store_run<CPL, true>(Ag + off, av);
        store_run<CPL, true>(Bg + off, bv);
      } else if (ncol > 0) {
        store_partial<CPL>(Mg + off, mv, ncol);
        store_partial<CPL>(Ag + off, av, ncol);
        store_partial<CPL>(Bg + off, bv, ncol);
      }
    }
    // the next strip re-reads this strip's last column (same wave, same CU)
    if (i0 + kStrip < la) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }

  const unsigned long long err = sw.reduce_err();
  if (lane == 0) p.status[pair] = err;
}

template <int CPL>
static hipError_t launch_cpl(const SaFillParams &p, hipStream_t stream) {
  const bool general = needs_general(p);
  const dim3 grid((p.n_pairs + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  if (p.K <= 1) {
    if (general) hipLaunchKernelGGL((fill_rowscan_kernel<CPL, SA_SUBST_SIMPLE, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((fill_rowscan_kernel<CPL, SA_SUBST_SIMPLE, false>), grid, block, 0, stream, p);
  } else if (p.K <= SA_LDS_TABLE_MAX_K) {
    const size_t lds = (size_t)p.K * p.K * sizeof(int32_t);
    if (general) hipLaunchKernelGGL((fill_rowscan_kernel<CPL, SA_SUBST_LDS, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((fill_rowscan_kernel<CPL, SA_SUBST_LDS, false>), grid, block, lds, stream, p);
  } else {
    hipLaunchKernelGGL((fill_rowscan_kernel<CPL, SA_SUBST_GLOBAL, true>), grid, block, 0, stream, p);
  }
  return hipGetLastError();
}

}  // namespace sa

hipError_t sa_launch_fill_rowscan(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  const uint32_t need = sa::columns_per_lane(max_len_a, p.tune_cpl);
  if (need <= 1) return sa::launch_cpl<1>(p, stream);
  if (need <= 2) return sa::launch_cpl<2>(p, stream);
  if (need <= 3) return sa::launch_cpl<3>(p, stream);
  if (need <= 4) return sa::launch_cpl<4>(p, stream);
  if (need <= 5) return sa::launch_cpl<5>(p, stream);
  if (need <= 6) return sa::launch_cpl<6>(p, stream);
  return sa::launch_cpl<8>(p, stream);
}
