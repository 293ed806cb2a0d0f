// sa_rowsweep.hpp -- one matrix row per step: the arithmetic shared by the
// row-sweep kernels (sa_fill_rowscan.hip: direct row stores; sa_fill_stream.hip:
// rows appended to an LDS ring and flushed as aligned 1 KiB blocks).
//
// Recurrence: reference src/alignment.c:89-167.  Lane l holds CPL
// consecutive columns of the current row, row position g = l*CPL + c:
//   match (i,j) <- max3 of (i-1,j-1)   registers + one DPP wave_shr:1
//   gap_a (i,j) <- (i,j-1)             registers
//   gap_b (i,j) <- (i-1,j)             a dependency ALONG the row:
//        B(g) = max(B(g-1) + ext, cin(g)),  cin(g) = max(max(M,A)(g-1) + open1, floor)
//     i.e. B(g) = max_{k<=g} (cin(k) + (g-k)*ext)            ((max,+) prefix scan)
//               = g*ext + max_{k<=g} (cin(k) - k*ext)        (de-trended)
//     so with per-column constants c1 = open1 - g*ext, c2 = floor - g*ext,
//     c3 = g*ext:
//        w(g) = max(max(M,A)(g-1) + c1, c2)     (2 ops)
//        P(g) = max(P(g-1), w(g))               (plain prefix max: serial over the
//               lane's CPL columns, 6 v_max_i32 with a DPP source across the wave
//               -- row_shr 1,2,4,8, row_bcast 15/31 -- one max for the carry)
//        B(g) = P(g) + c3                       (1 op)
// Exactness (cells may hold the NW floor INT_MIN+|min_penalty|, reference
// alignment.c:41): nothing here adds a multiple of ext to an arbitrary cell.
// w(g) = cin(g) + g*|ext| moves cin UP from >= floor, and B(g) = P(g) + g*ext
// >= w(g) + g*ext = cin(g) >= floor because the maximum includes k = g; so no add
// can wrap, no saturation is needed, and the result is the serial recurrence's
// bit for bit.  Every other add is one the reference performs itself (value >=
// floor plus one penalty >= -|min_penalty|).
// ext > 0 (legal upstream, a gap that pays): the same argument needs the trend
// to move cin UP again, so it is taken from the right end instead -- with
// t(g) = (G - g)*ext >= 0 for a G at or beyond the last column,
//   B(g) = max_{k<=g} (cin(k) + t(k)) - t(g),
// c1 = open1 + t(g), c2 = floor + t(g), c3 = -t(g); for ext <= 0, t(g) = -g*ext
// is the same thing with G = 0.  One constant (`trend0` = t(0)) is all that
// differs; the row loop is unchanged.
#pragma once

#include "sa_fill_common.hpp"

namespace sa {

// DPP move; lanes without a source keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

// Inclusive max scan over the 64 lanes.  INT_MIN is max's identity, so the
// compiler folds each step into one v_max_i32_dpp (lanes without a source are
// left untouched).
__device__ __forceinline__ int wave_scan_max(int u) {
  constexpr int NEG = INT32_MIN;
  u = max(u, dpp_mov<0x111, 0xf>(NEG, u));   // row_shr:1
  u = max(u, dpp_mov<0x112, 0xf>(NEG, u));   // row_shr:2
  u = max(u, dpp_mov<0x114, 0xf>(NEG, u));   // row_shr:4
  u = max(u, dpp_mov<0x118, 0xf>(NEG, u));   // row_shr:8
  u = max(u, dpp_mov<0x142, 0xa>(NEG, u));   // row_bcast:15 -> rows 1,3
  u = max(u, dpp_mov<0x143, 0xc>(NEG, u));   // row_bcast:31 -> rows 2,3
  return u;
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

// GENERAL: any of no_end_gap / no_gaps_in_a / no_gaps_in_b, a sentinel in the
// substitution scores, or gap_open > 0 (the fast path uses open1 <= ext).
// COL0: lane 0's first column is the BORDER column 0 (stream kernel: a row is
// then exactly len_a+1 consecutive cells owned by consecutive lanes, nothing is
// fed in from the left); col0 is then lane*CPL - 1 and wraps for lane 0.
template <int CPL, int SUBST, bool GENERAL, bool COL0 = false>
struct RowSweep {
  int fa[CPL], arow[CPL];          // my columns of seq_a: folded char, class*K
  int X[CPL], Ap[CPL];             // previous row: max3(M,A,B), A
  int Y[GENERAL ? CPL : 1];        // previous row: max(M,B) (GENERAL only, see row())
  int c1[CPL], c2[CPL], c3[CPL];   // gap_b scan constants (header comment)
  int boundX;                      // max3 of (i0, j-1): lane 0's up-left
  int trend0;                      // t(0) of the header comment: 0 for ext <= 0, G*ext for ext > 0
  unsigned long long err = ~0ull;  // first cell without a score (GENERAL)

  // columns col0+1.., previous row = row 0 (reference alignment.c:61-69)
  __device__ __forceinline__ void start_strip(const SaFillParams &p, const SweepConsts &k, const Border &bd,
                                              const uint8_t *__restrict__ seq_a, uint32_t la, uint32_t i0,
                                              uint32_t col0, int lane) {
    trend0 = k.ext > 0 ? (kWave * CPL) * k.ext : 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t idx = col0 + c;
      const int code = idx < la ? (int)p.code[seq_a[idx]] : 0;
      fa[c] = code & 0xff;
      arow[c] = (code >> 8) * k.K;
      const int b0 = bd.edge_gap(idx + 1);
      X[c] = max(k.floor_, b0);
      if constexpr (GENERAL) Y[c] = max(k.floor_, b0);
      Ap[c] = k.floor_;
      if constexpr (COL0) {
        if (c == 0 && idx == 0xFFFFFFFFu) {   // cell (0,0)
          X[c] = Ap[c] = 0;
          if constexpr (GENERAL) Y[c] = 0;
        }
      }
      const int g_ext = (lane * CPL + c) * k.ext - trend0;   // -t(g)
      c1[c] = k.open1 - g_ext;
      c2[c] = k.floor_ - g_ext;
      c3[c] = g_ext;
    }
    boundX = (i0 == 0) ? 0 : max(k.floor_, bd.edge_gap(i0));
  }

  // Row j.  code_b: seq_b[j-1]'s code (uniform); feedZ/feedB: max(M,A) and B of
  // the cell left of the strip on row j (uniform, unused with COL0); edge_a (COL0
  // only): gap_a of the border cell (0, j).  Produces the row's M/A/B.
  __device__ __forceinline__ void row(const SweepConsts &k, uint32_t j, uint32_t lb, uint32_t la, uint32_t W,
                                      int lane, uint32_t col0, int ncol, int code_b, int feedZ, int feedB,
                                      int (&mv)[CPL], int (&av)[CPL], int (&bv)[CPL], int edge_a = 0) {
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
      int m, a;
      if constexpr (GENERAL) {
        // reference alignment.c:101-137
        const int a_norm = max3i(addw(Y[c], k.open1), addw(Ap[c], k.ext), k.floor_);
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
        // max(M,B)+open1 vs A+ext: with open1 <= ext (gap_open <= 0, guaranteed by
        // the launcher for this path) A+open1 <= A+ext, so max3(M,A,B) may stand in
        // for max(M,B) and the separate max(M,B) register is not needed
        a = max3i(addw(X[c], k.open1), addw(Ap[c], k.ext), k.floor_);
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

    // gap_b (reference alignment.c:139-155)
    bool free_row = false, forced = false;
    if constexpr (GENERAL) {
      const bool last_row = (j == lb);             // wave-uniform
      free_row = last_row && k.no_end;             // max3 of the left cell, no penalty, no clamp
      forced = k.no_gaps_b && !last_row;
    }
    if (forced) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) bv[c] = k.floor_;
    } else {
      const int zin = COL0 ? wave_shr1(z[CPL - 1], z[CPL - 1])   // lane 0: overridden below
                           : wave_shr1(z[CPL - 1], feedZ);       // max(M,A) of (g-1, j)
      int P[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int zl = (c == 0) ? zin : z[c - 1];
        int w = free_row ? zl : max(addw(zl, c1[c]), c2[c]);
        if (c == 0) {
          if constexpr (COL0) {
            w = (lane == 0) ? (free_row ? k.floor_ : c2[0]) : w;   // gap_b of (0, j) is the floor
          } else {
            // lane 0 continues the previous strip: B(left) + ext, de-trended at g = 0
            const int carry = free_row ? feedB : addw(addw(feedB, k.ext), trend0);
            w = (lane == 0) ? max(w, carry) : w;
          }
        }
        P[c] = (c == 0) ? w : max(P[c - 1], w);
      }
      const int incl = wave_scan_max(P[CPL - 1]);
      const int e = wave_shr1(incl, INT32_MIN);     // prefix max of the lanes to my left
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int pm = max(P[c], e);
        bv[c] = free_row ? pm : addw(pm, c3[c]);
      }
    }

#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      X[c] = max(z[c], bv[c]);
      if constexpr (GENERAL) Y[c] = max(mv[c], bv[c]);
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

// fast path vs GENERAL, from the flattened scoring
inline bool needs_general(const SaFillParams &p) {
  return (p.flags & (SA_F_NO_END_GAP | SA_F_NO_GAPS_A | SA_F_NO_GAPS_B | SA_F_HAS_SENTINEL)) ||
         p.open1 > p.ext;
}

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
