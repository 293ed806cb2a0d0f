// sa_fill_wavefront.hip -- anti-diagonal wavefront fill, one wave per pair.
//
// Replaces alignment_fill_matrices (reference src/alignment.c:28-168).
//
// Mapping.  Lane l of the pair's wave owns CPL consecutive columns
// i = i0 + l*CPL + 1 .. +CPL of the current column strip (i0 = 0 unless
// len_a > 64*CPL).  At step t lane l computes row j = t - l + 1 of its columns,
// so the 64 lanes sit on one anti-diagonal of the lane-block grid and step t
// only needs step t-1 (left, from lane l-1) and the lane's own previous row
// (up / up-left), all in registers:
//
//   up      (i, j-1)   own registers   Y=max(M,B), Ap=A        -> gap_a
//   left    (i-1, j)   lane l-1, one DPP wave_shr:1 of (max(M,A), B)  -> gap_b
//   upleft  (i-1, j-1) what "left" was one step earlier              -> match
//
// Lane 0's left neighbour is the border column (analytic) or, for strips after
// the first, the previous strip's last column read back from the matrices; both
// and the seq_b codes are fetched 64 rows at a time (one row per lane) and fed
// to lane 0 with v_readlane, so no lane ever waits on a per-step load.
//
// Stores: each lane writes CPL consecutive int32 per matrix per step
// (global_store_dwordx3 / x4 ...): 12-16 B runs, one row per lane.  The rows of
// a 16-step window abut in L2, which merges them into full lines before they
// reach HBM.  The row-sweep kernel (sa_fill_rowscan.hip) writes whole rows per
// instruction instead; bench.py reports which of the two is used.
#include "sa_fill_common.hpp"

namespace sa {

template <int CPL, int SUBST, bool GENERAL>
__global__ void __launch_bounds__(kWave *kWavesPerBlock)
fill_wavefront_kernel(const SaFillParams p) {
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

  // ---- borders: row 0 (coalesced) and column 0 (reference alignment.c:46-81)
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
    const uint32_t n_lanes = (cols + CPL - 1) / CPL;
    const uint32_t col0 = i0 + lane * CPL;             // matrix column left of my first
    const int ncol = max(0, min(CPL, (int)cols - lane * CPL));

    // my columns of seq_a, and row 0 as the "previous row"
    int fa[CPL], arow[CPL], X[CPL], Y[CPL], Ap[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t idx = col0 + c;                    // 0-based index into seq_a
      const int code = idx < la ? (int)p.code[sa_[idx]] : 0;
      fa[c] = code & 0xff;
      arow[c] = (code >> 8) * K;
      const int b0 = bd.edge_gap(idx + 1);              // row 0: M = A = floor, B = edge
      X[c] = max(floor_, b0);
      Y[c] = max(floor_, b0);
      Ap[c] = floor_;
    }
    // up-left of my first column on row 1 = cell (col0, 0)
    int XL = (col0 == 0) ? 0 : max(floor_, bd.edge_gap(col0));

    // seq_a codes must have landed before the step loop (see the note on vmcnt
    // below): otherwise every step would wait vmcnt(0) at their first use.
    __builtin_amdgcn_s_waitcnt(kWaitVm0);

    const uint32_t T = lb + n_lanes - 1;
    int pubZ = 0, pubB = 0, bchain = 0;
    int chunk_code = 0, chunk_Z = 0, chunk_B = 0;

    for (uint32_t t = 0; t < T; ++t) {
      // every 64 steps: lane q fetches what lane 0 will need for row t+q+1
      if ((t & (kWave - 1)) == 0) {
        const uint32_t r = t + lane + 1;
        if (r <= lb) {
          chunk_code = p.code[sb_[r - 1]];
          if (i0 == 0) {
            chunk_Z = max(floor_, bd.edge_gap(r));      // max(M,A) of (0,r)
            chunk_B = floor_;
          } else {
            const size_t c = (size_t)r * W + i0;
            chunk_Z = max(Mg[c], Ag[c]);
            chunk_B = Bg[c];
          }
        }
        // Land these loads HERE, once per 64 steps.  gfx9 has one vmcnt for loads
        // and stores; left to itself the compiler waits vmcnt(0) at the first use
        // in EVERY step, which would also drain the step's stores.
        __builtin_amdgcn_s_waitcnt(kWaitVm0);
      }
      const int q = t & (kWave - 1);
      const int inZ = wave_shr1(pubZ, read_lane(chunk_Z, q));
      const int inB = wave_shr1(pubB, read_lane(chunk_B, q));
      bchain = wave_shr1(bchain, read_lane(chunk_code, q));

      const uint32_t jm1 = t - (uint32_t)lane;          // row - 1 (wraps if t < lane)
      if (jm1 < lb && ncol > 0) {
        const uint32_t j = jm1 + 1;
        int zl = inZ, bl = inB, xd = XL;
        XL = max(inZ, inB);
        int mv[CPL], av[CPL], bv[CPL];
        bool last_row = false;
        if constexpr (GENERAL) last_row = (j == lb);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int s = subst_score<SUBST>(fa[c], arow[c], bchain, table, gen_eq, gen_ne);
          int m, a, b;
          const int a_norm = max3i(addw(Y[c], open1), addw(Ap[c], ext), floor_);
          const int b_norm = max3i(addw(zl, open1), addw(bl, ext), floor_);
          if constexpr (GENERAL) {
            // reference alignment.c:101-155
            m = (s == SA_S_BLOCKED) ? floor_ : max(addw(xd, s), floor_);
            if (s == SA_S_UNKNOWN && c < ncol) {
              m = floor_;
              err = min(err, (unsigned long long)j * W + col0 + c + 1);
            }
            const bool last_col = (col0 + c + 1 == la);
            a = (last_col && no_end) ? max(Y[c], Ap[c])
                : (!no_gaps_a || last_col) ? a_norm : floor_;
            b = (last_row && no_end) ? max(zl, bl)
                : (!no_gaps_b || last_row) ? b_norm : floor_;
          } else {
            m = max(addw(xd, s), floor_);
            a = a_norm;
            b = b_norm;
          }
          xd = X[c];
          const int z = max(m, a);
          X[c] = max(z, b);
          Y[c] = max(m, b);
          Ap[c] = a;
          zl = z;
          bl = b;
          mv[c] = m; av[c] = a; bv[c] = b;
        }
        pubZ = zl;
        pubB = bl;
        const size_t off = (size_t)j * W + col0 + 1;
        if (ncol == CPL) {
          store_run<CPL>(Mg + off, mv);
          store_run<CPL>(Ag + off, av);
          store_run<CPL>(Bg + off, bv);
        } else {
          store_partial<CPL>(Mg + off, mv, ncol);
          store_partial<CPL>(Ag + off, av, ncol);
          store_partial<CPL>(Bg + off, bv, ncol);
        }
      }
    }
    // the next strip re-reads this strip's last column (same wave, same CU)
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
    if (general) hipLaunchKernelGGL((fill_wavefront_kernel<CPL, SA_SUBST_SIMPLE, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((fill_wavefront_kernel<CPL, SA_SUBST_SIMPLE, false>), grid, block, 0, stream, p);
  } else if (p.K <= SA_LDS_TABLE_MAX_K) {
    const size_t lds = (size_t)p.K * p.K * sizeof(int32_t);
    if (general) hipLaunchKernelGGL((fill_wavefront_kernel<CPL, SA_SUBST_LDS, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((fill_wavefront_kernel<CPL, SA_SUBST_LDS, false>), grid, block, lds, stream, p);
  } else {
    hipLaunchKernelGGL((fill_wavefront_kernel<CPL, SA_SUBST_GLOBAL, true>), grid, block, 0, stream, p);
  }
  return hipGetLastError();
}

}  // namespace sa

hipError_t sa_launch_fill_wavefront(const SaFillParams &p, uint32_t max_len_a,
                                    hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  const uint32_t need = sa::columns_per_lane(max_len_a, p.tune_cpl);
  if (need <= 1) return sa::launch_cpl<1>(p, stream);
  if (need <= 2) return sa::launch_cpl<2>(p, stream);
  if (need <= 3) return sa::launch_cpl<3>(p, stream);
  if (need <= 4) return sa::launch_cpl<4>(p, stream);
  if (need <= 5) return sa::launch_cpl<5>(p, stream);
  if (need <= 6) return sa::launch_cpl<6>(p, stream);
  return sa::launch_cpl<8>(p, stream);   // longer rows: strips of 512 columns
}

// ---- DPP self-test -----------------------------------------------------------
namespace sa {
__global__ void dpp_probe_kernel(int32_t *out, int32_t fill) {
  const int lane = threadIdx.x;
  out[lane] = wave_shr1(lane * 3 + 1, fill);
}
}  // namespace sa

hipError_t sa_launch_dpp_probe(int32_t *out64, int32_t fill, hipStream_t stream) {
  hipLaunchKernelGGL(sa::dpp_probe_kernel, dim3(1), dim3(64), 0, stream, out64, fill);
  return hipGetLastError();
}
