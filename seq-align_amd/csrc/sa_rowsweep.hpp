// sa_rowsweep.hpp -- one matrix row per step: the arithmetic shared by the
// row-sweep kernels (sa_fill_rowscan.hip: direct row stores; sa_fill_stream.hip:
// rows appended to an LDS ring and flushed as aligned 1 KiB blocks).
//
// Recurrence: reference src/alignment.c:89-167.  Lane l holds
// columns col0+1 .. col0+CPL of the current row (col0 = i0 + l*CPL):
//   match (i,j) <- max3 of (i-1,j-1)   registers + one DPP wave_shr:1
//   gap_a (i,j) <- (i,j-1)             registers
//   gap_b (i,j) <- (i-1,j)             a dependency ALONG the row:
//        B(i) = max(B(i-1) + ext, cin(i)),  cin(i) = max(max(M,A)(i-1) + open1, floor)
//     i.e. B(i) = max_k (cin(k) + (i-k)*ext): a prefix scan in the (max,+)
//     semiring.  Each lane scans its CPL columns serially, the 64 lane totals are
//     scanned across the wave (wave_scan_maxplus: de-trended, 6 DPP max steps),
//     and the carry is applied on the way out.
// Exactness: cells may hold the NW floor INT_MIN+|min_penalty| (reference
// alignment.c:41), so no step may add a multiple of ext to an arbitrary cell.
// The wave scan never does (see wave_scan_maxplus); the carry into a lane's
// first column is one saturating add (v_add_i32 clamp) whose only saturating
// input is the "no carry" identity INT_MIN of lane 0.  Every other add is one the
// reference performs itself (value >= floor plus one penalty >= -|min_penalty|,
// SURVEY A.3-3).  Results are bit-identical to the serial recurrence.
#pragma once

#include "sa_fill_common.hpp"

namespace sa {

__device__ __forceinline__ int add_sat(int a, int b) {
  return __builtin_elementwise_add_sat(a, b);   // v_add_i32 ... clamp
}

// DPP move; lanes without a source keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

// Inclusive (max,+) scan over the 64 lanes: I_l = max_{m<=l} (g_m + (l-m)*d),
// d <= 0.  De-trended: with u_m = g_m - m*d the decay disappears,
//     I_l = l*d + max_{m<=l} u_m,
// so the wave part is a PLAIN max scan -- 6 v_max_i32 with a DPP source
// (row_shr 1,2,4,8, row_bcast 15/31; INT_MIN is max's identity, so lanes without
// a source are untouched) -- instead of 6 x (shift, saturating add, max).
// Exact without saturation: u_m = g_m + m*|d| cannot underflow, and
// I_l >= g_l >= floor because the maximum includes m = l.
// lane_d = lane * d.
__device__ __forceinline__ int wave_scan_maxplus(int g, int lane_d) {
  constexpr int NEG = INT32_MIN;
  int u = (int)((unsigned)g - (unsigned)lane_d);
  u = max(u, dpp_mov<0x111, 0xf>(NEG, u));   // row_shr:1
  u = max(u, dpp_mov<0x112, 0xf>(NEG, u));   // row_shr:2
  u = max(u, dpp_mov<0x114, 0xf>(NEG, u));   // row_shr:4
  u = max(u, dpp_mov<0x118, 0xf>(NEG, u));   // row_shr:8
  u = max(u, dpp_mov<0x142, 0xa>(NEG, u));   // row_bcast:15 -> rows 1,3
  u = max(u, dpp_mov<0x143, 0xc>(NEG, u));   // row_bcast:31 -> rows 2,3
  return addw(u, lane_d);
}

// wave-uniform scoring constants
struct SweepConsts {
  int floor_, open1, ext, gen_eq, gen_ne, K;
  bool no_end, no_gaps_a, no_gaps_b;
  const int32_t *table;
  __device__ __forceinline__ SweepConsts(const SaFillParams &p, const int32_t *tbl)
      : floor_(p.floor), open1(p.open1), ext(p.ext), gen_eq(p.gen_eq), gen_ne(p.gen_ne),
        K((int)p.K), no_end(p.flags & SA_F_NO_END_GAP), no_gaps_a(p.flags & SA_F_NO_GAPS_A),
        no_gaps_b(p.flags & SA_F_NO_GAPS_B), table(tbl) {}
};

// COL0: lane 0's first column is the BORDER column 0 (stream kernel: a row is
// then exactly len_a+1 consecutive cells owned by consecutive lanes, nothing is
// fed in from the left); col0 is then lane*CPL - 1 and wraps for lane 0.
template <int CPL, int SUBST, bool GENERAL, bool COL0 = false>
struct RowSweep {
  int fa[CPL], arow[CPL];          // my columns of seq_a: folded char, class*K
  int X[CPL], Y[CPL], Ap[CPL];     // previous row: max3(M,A,B), max(M,B), A
  int boundX;                      // max3 of (i0, j-1): lane 0's up-left
  unsigned long long err = ~0ull;  // first cell without a score (GENERAL)

  // columns col0+1.., previous row = row 0 (reference alignment.c:61-69)
  __device__ __forceinline__ void start_strip(const SaFillParams &p, const SweepConsts &k, const Border &bd,
                                              const uint8_t *__restrict__ seq_a, uint32_t la, uint32_t i0,
                                              uint32_t col0) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t idx = col0 + c;
      const int code = idx < la ? (int)p.code[seq_a[idx]] : 0;
      fa[c] = code & 0xff;
      arow[c] = (code >> 8) * k.K;
      const int b0 = bd.edge_gap(idx + 1);
      X[c] = max(k.floor_, b0);
      Y[c] = max(k.floor_, b0);
      Ap[c] = k.floor_;
      if constexpr (COL0) {
        if (c == 0 && idx == 0xFFFFFFFFu) X[c] = Y[c] = Ap[c] = 0;   // cell (0,0)
      }
    }
    boundX = (i0 == 0) ? 0 : max(k.floor_, bd.edge_gap(i0));
  }

  // Row j.  code_b: seq_b[j-1]'s code (uniform); feedZ/feedB: max(M,A) and B of
  // the cell left of the strip on row j (uniform).  Produces the row's M/A/B.
  __device__ __forceinline__ void row(const SweepConsts &k, uint32_t j, uint32_t lb, uint32_t la, uint32_t W,
                                      int lane, uint32_t col0, int ncol, int code_b, int feedZ, int feedB,
                                      int (&mv)[CPL], int (&av)[CPL], int (&bv)[CPL], int edge_a = 0) {
    // edge_a (COL0 only): gap_a of the border cell (0, j)
    int xd;                                        // max3 of (i-1, j-1)
    if constexpr (COL0) {
      xd = wave_shr1(X[CPL - 1], X[CPL - 1]);      // lane 0's value is overridden below
    } else {
      xd = wave_shr1(X[CPL - 1], boundX);
      boundX = max(feedZ, feedB);
    }
    int z[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int s = subst_score<SUBST>(fa[c], arow[c], code_b, k.table, k.gen_eq, k.gen_ne);
      const int a_norm = max3i(addw(Y[c], k.open1), addw(Ap[c], k.ext), k.floor_);
      int m, a;
      if constexpr (GENERAL) {
        // reference alignment.c:101-137
        m = (s == SA_S_BLOCKED) ? k.floor_ : max(addw(xd, s), k.floor_);
        if (s == SA_S_UNKNOWN && c < ncol && !(COL0 && c == 0 && lane == 0)) {
          m = k.floor_;
          err = min(err, (unsigned long long)j * W + col0 + c + 1);
        }
        const bool last_col = (col0 + c + 1 == la);
        a = (last_col && k.no_end) ? max(Y[c], Ap[c])
            : (!k.no_gaps_a || last_col) ? a_norm : k.floor_;
      } else {
        m = max(addw(xd, s), k.floor_);
        a = a_norm;
      }
      if constexpr (COL0) {
        if (c == 0) {   // border column (reference alignment.c:72-80)
          m = (lane == 0) ? k.floor_ : m;
          a = (lane == 0) ? edge_a : a;
        }
      }
      xd = X[c];
      mv[c] = m; av[c] = a; z[c] = max(m, a);
    }

    // gap_b: (max,+) prefix scan along the row (reference alignment.c:139-155)
    int r_open = k.open1, r_ext = k.ext, r_floor = k.floor_;
    bool b_forced = false;
    if constexpr (GENERAL) {
      const bool last_row = (j == lb);             // wave-uniform
      if (last_row && k.no_end) { r_open = 0; r_ext = 0; r_floor = INT32_MIN; }
      else if (k.no_gaps_b && !last_row) b_forced = true;
    }
    if (b_forced) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) bv[c] = k.floor_;
    } else {
      const int zin = COL0 ? wave_shr1(z[CPL - 1], z[CPL - 1])   // lane 0: overridden below
                           : wave_shr1(z[CPL - 1], feedZ);       // max(M,A) of (i-1, j)
      int L[CPL];
      {
        const int cin0 = max(addw(zin, r_open), r_floor);
        // only lane 0 has a real left-neighbour gap_b before the wave scan
        if constexpr (COL0) {
          L[0] = (lane == 0) ? k.floor_ : cin0;         // gap_b of (0, j) is the floor
        } else {
          const int carry0 = (lane == 0) ? add_sat(feedB, r_ext) : INT32_MIN;
          L[0] = max(cin0, carry0);
        }
      }
#pragma unroll
      for (int c = 1; c < CPL; ++c) {
        const int cin = max(addw(z[c - 1], r_open), r_floor);
        L[c] = max(addw(L[c - 1], r_ext), cin);
      }
      const int incl = wave_scan_maxplus(L[CPL - 1], lane * (CPL * r_ext));
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
  }

  __device__ __forceinline__ unsigned long long reduce_err() {
    unsigned long long e = err;
    if constexpr (GENERAL) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) e = min(e, __shfl_xor(e, o));
    }
    return e;
  }
};

// Every 64 rows lane q fetches what row j0+q needs: seq_b's code and, for strips
// after the first, the previous strip's last column.  Consumed with v_readlane.
struct RowFeed {
  int code = 0, Z = 0, B = 0;
  __device__ __forceinline__ void load(const SaFillParams &p, const SweepConsts &k, const Border &bd,
                                       const uint8_t *__restrict__ seq_b, uint32_t lb, uint32_t W, uint32_t i0,
                                       const int32_t *Mg, const int32_t *Ag, const int32_t *Bg, uint32_t r) {
    if (r <= lb) {
      code = p.code[seq_b[r - 1]];
      if (i0 == 0) {   // border column (reference alignment.c:72-80)
        Z = max(k.floor_, bd.edge_gap(r));
        B = k.floor_;
      } else {
        const size_t c = (size_t)r * W + i0;
        Z = max(Mg[c], Ag[c]);
        B = Bg[c];
      }
    }
    // Land the loads HERE, once per 64 rows: gfx9 has one vmcnt for loads and
    // stores, and left alone the compiler waits vmcnt(0) at the first use in
    // EVERY row, which would also drain that row's stores.
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
  }
};

}  // namespace sa
