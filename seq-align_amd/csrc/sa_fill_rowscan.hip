// sa_fill_rowscan.hip -- row-sweep fill with a max-plus prefix scan for gap_b.
//
// Replaces alignment_fill_matrices (reference src/alignment.c:28-168); same
// results as sa_fill_wavefront.hip, different schedule, chosen for the store
// side: the fill is bound by HBM writes (12 B per cell, three int32 matrices
// with the reference's dense pitch), and here every store instruction writes
// one whole matrix row segment -- lane l holds columns l*CPL+1 .. +CPL of the
// row, so the 64 lanes' dwordx3/x4 stores are back to back in memory (600 B
// per instruction at 150 columns) instead of 64 different rows.
//
// One wave per pair, one matrix row per step, no skew, no barrier, no LDS:
//   match (i,j) <- max3 of (i-1,j-1)   own registers / one DPP wave_shr:1
//   gap_a (i,j) <- (i,j-1)             own registers
//   gap_b (i,j) <- (i-1,j)             a dependency ALONG the row:
//        B(i) = max(B(i-1) + ext, cin(i)),  cin(i) = max(max(M,A)(i-1) + open1, floor)
//     i.e. B(i) = max_k (cin(k) + (i-k)*ext): a prefix scan in the (max,+)
//     semiring.  Each lane scans its CPL columns serially, the 64 lane totals
//     are scanned with 6 DPP steps (row_shr 1,2,4,8, row_bcast 15, 31), and the
//     carry is applied on the way out.
// Exactness: the wave scan adds up to 63*CPL*ext to cells that may hold the NW
// floor INT_MIN+|min_penalty| (reference alignment.c:41); those adds SATURATE
// (v_add_i32 clamp).  A saturated term is below the floor, every true gap_b is
// >= floor, so it can never be the maximum: results are bit-identical to the
// serial recurrence.  All other adds are the reference's own (value >= floor
// plus one penalty >= -|min_penalty|).
#include "sa_fill_common.hpp"

namespace sa {

__device__ __forceinline__ int add_sat(int a, int b) {
  return __builtin_elementwise_add_sat(a, b);   // v_add_i32 ... clamp
}

// DPP moves; lanes without a source keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

// Inclusive (max,+) scan over the 64 lanes: I_l = max_{m<=l} (g_m + (l-m)*d).
// kd[0..3] = 1,2,4,8 * d; kb15 = ((lane&15)+1)*d; kb31 = ((lane&31)+1)*d.
__device__ __forceinline__ int wave_scan_maxplus(int g, int d1, int d2, int d4, int d8,
                                                 int kb15, int kb31) {
  constexpr int NEG = INT32_MIN;
  int v = g;
  v = max(v, add_sat(dpp_mov<0x111, 0xf>(NEG, v), d1));   // row_shr:1
  v = max(v, add_sat(dpp_mov<0x112, 0xf>(NEG, v), d2));   // row_shr:2
  v = max(v, add_sat(dpp_mov<0x114, 0xf>(NEG, v), d4));   // row_shr:4
  v = max(v, add_sat(dpp_mov<0x118, 0xf>(NEG, v), d8));   // row_shr:8
  v = max(v, add_sat(dpp_mov<0x142, 0xa>(NEG, v), kb15)); // row_bcast:15 -> rows 1,3
  v = max(v, add_sat(dpp_mov<0x143, 0xc>(NEG, v), kb31)); // row_bcast:31 -> rows 2,3
  return v;
}

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

  const int floor_ = p.floor, open1 = p.open1, ext = p.ext;
  const int gen_eq = p.gen_eq, gen_ne = p.gen_ne;
  const int K = (int)p.K;
  const uint32_t flags = p.flags;
  const Border bd{floor_, p.gap_open, ext, (flags & SA_F_IS_SW) != 0,
                  (flags & SA_F_NO_START_GAP) != 0};
  const bool no_end = flags & SA_F_NO_END_GAP;
  const bool no_gaps_a = flags & SA_F_NO_GAPS_A;
  const bool no_gaps_b = flags & SA_F_NO_GAPS_B;

  // ---- borders (reference alignment.c:46-81)
  for (uint32_t i = lane; i <= la; i += kWave) {
    const int fl = (i == 0) ? 0 : floor_;
    Mg[i] = fl;
    Ag[i] = fl;
    Bg[i] = (i == 0) ? 0 : bd.edge_gap(i);
  }
  for (uint32_t j = 1 + lane; j <= lb; j += kWave) {
    const size_t c = (size_t)j * W;
    Mg[c] = floor_;
    Ag[c] = bd.edge_gap(j);
    Bg[c] = floor_;
  }

  unsigned long long err = ~0ull;
  constexpr uint32_t kStrip = kWave * CPL;

  for (uint32_t i0 = 0; i0 < la; i0 += kStrip) {
    const uint32_t cols = min(kStrip, la - i0);
    const uint32_t col0 = i0 + lane * CPL;
    const int ncol = max(0, min(CPL, (int)cols - lane * CPL));

    int fa[CPL], arow[CPL], X[CPL], Y[CPL], Ap[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t idx = col0 + c;
      const int code = idx < la ? (int)p.code[sa_[idx]] : 0;
      fa[c] = code & 0xff;
      arow[c] = (code >> 8) * K;
      const int b0 = bd.edge_gap(idx + 1);
      X[c] = max(floor_, b0);
      Y[c] = max(floor_, b0);
      Ap[c] = floor_;
    }
    // (i0, row 0): max3 of the cell left of the strip
    int boundX = (i0 == 0) ? 0 : max(floor_, bd.edge_gap(i0));
    __builtin_amdgcn_s_waitcnt(kWaitVm0);   // see sa_fill_wavefront.hip on vmcnt

    int chunk_code = 0, chunk_Z = 0, chunk_B = 0;
    uint32_t off = W + col0 + 1;            // (row 1, my first column)

    for (uint32_t j = 1; j <= lb; ++j, off += W) {
      const int q = (j - 1) & (kWave - 1);
      if (q == 0) {   // every 64 rows: lane k fetches seq_b / left boundary of row j+k
        const uint32_t r = j + lane;
        if (r <= lb) {
          chunk_code = p.code[sb_[r - 1]];
          if (i0 == 0) {
            chunk_Z = max(floor_, bd.edge_gap(r));
            chunk_B = floor_;
          } else {
            const size_t c = (size_t)r * W + i0;
            chunk_Z = max(Mg[c], Ag[c]);
            chunk_B = Bg[c];
          }
        }
        __builtin_amdgcn_s_waitcnt(kWaitVm0);
      }
      const int code_b = read_lane(chunk_code, q);   // wave-uniform
      const int feedZ = read_lane(chunk_Z, q), feedB = read_lane(chunk_B, q);

      // ---- match and gap_a of the whole row segment
      int xd = wave_shr1(X[CPL - 1], boundX);        // max3 of (i-1, j-1)
      boundX = max(feedZ, feedB);
      int mv[CPL], av[CPL], z[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int s = subst_score<SUBST>(fa[c], arow[c], code_b, table, gen_eq, gen_ne);
        const int a_norm = max3i(addw(Y[c], open1), addw(Ap[c], ext), floor_);
        int m, a;
        if constexpr (GENERAL) {
          m = (s == SA_S_BLOCKED) ? floor_ : max(addw(xd, s), floor_);
          if (s == SA_S_UNKNOWN && c < ncol) {
            m = floor_;
            err = min(err, (unsigned long long)j * W + col0 + c + 1);
          }
          const bool last_col = (col0 + c + 1 == la);
          a = (last_col && no_end) ? max(Y[c], Ap[c])
              : (!no_gaps_a || last_col) ? a_norm : floor_;
        } else {
          m = max(addw(xd, s), floor_);
          a = a_norm;
        }
        xd = X[c];
        mv[c] = m; av[c] = a; z[c] = max(m, a);
      }

      // ---- gap_b: (max,+) prefix scan along the row
      int bv[CPL];
      int r_open = open1, r_ext = ext, r_floor = floor_;
      bool b_forced = false;
      if constexpr (GENERAL) {
        const bool last_row = (j == lb);             // wave-uniform
        if (last_row && no_end) { r_open = 0; r_ext = 0; r_floor = INT32_MIN; }
        else if (no_gaps_b && !last_row) b_forced = true;
      }
      if (b_forced) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) bv[c] = floor_;
      } else {
        const int zin = wave_shr1(z[CPL - 1], feedZ);   // max(M,A) of (i-1, j)
        int L[CPL];
        {
          const int cin0 = max(addw(zin, r_open), r_floor);
          // only lane 0 has a real left neighbour B value before the wave scan
          const int carry0 = (lane == 0) ? add_sat(feedB, r_ext) : INT32_MIN;
          L[0] = max(cin0, carry0);
        }
#pragma unroll
        for (int c = 1; c < CPL; ++c) {
          const int cin = max(addw(z[c - 1], r_open), r_floor);
          L[c] = max(addw(L[c - 1], r_ext), cin);
        }
        const int d = CPL * r_ext;
        const int incl = wave_scan_maxplus(L[CPL - 1], d, 2 * d, 4 * d, 8 * d,
                                           ((lane & 15) + 1) * d, ((lane & 31) + 1) * d);
        const int e = wave_shr1(incl, INT32_MIN);       // gap_b of (col0, j), lanes >= 1
        bv[0] = max(L[0], add_sat(e, r_ext));
#pragma unroll
        for (int c = 1; c < CPL; ++c) bv[c] = max(L[c], addw(bv[c - 1], r_ext));
      }

#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        X[c] = max(z[c], bv[c]);
        Y[c] = max(mv[c], bv[c]);
        Ap[c] = av[c];
      }

      if (ncol == CPL) {
        store_run<CPL>(Mg + off, mv);
        store_run<CPL>(Ag + off, av);
        store_run<CPL>(Bg + off, bv);
      } else if (ncol > 0) {
        store_partial<CPL>(Mg + off, mv, ncol);
        store_partial<CPL>(Ag + off, av, ncol);
        store_partial<CPL>(Bg + off, bv, ncol);
      }
    }
    if (i0 + kStrip < la) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }

  if constexpr (GENERAL) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) err = min(err, __shfl_xor(err, o));
  }
  if (lane == 0) p.status[pair] = err;
}

template <int CPL>
static hipError_t launch_cpl(const SaFillParams &p, hipStream_t stream) {
  const bool general =
      p.flags & (SA_F_NO_END_GAP | SA_F_NO_GAPS_A | SA_F_NO_GAPS_B | SA_F_HAS_SENTINEL);
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
  const uint32_t need = (max_len_a + sa::kWave - 1) / sa::kWave;
  if (need <= 1) return sa::launch_cpl<1>(p, stream);
  if (need <= 2) return sa::launch_cpl<2>(p, stream);
  if (need <= 3) return sa::launch_cpl<3>(p, stream);
  if (need <= 4) return sa::launch_cpl<4>(p, stream);
  if (need <= 5) return sa::launch_cpl<5>(p, stream);
  if (need <= 6) return sa::launch_cpl<6>(p, stream);
  return sa::launch_cpl<8>(p, stream);
}
