// sa_fill_nw_dirs_x1.hpp -- one wave's work of fill_nw_dirs_kernel (sa_fill_dirs.hip): the directions-only Needleman-Wunsch
// fill of ONE pair.  A function of its own so that the mixed launch of sa_fill_dirs_x2.hip (pairs of the chunk's modal shape
// two per wave, the others one per wave, in ONE grid) can call it beside the packed body.
#pragma once

#include "sa_rowsweep.hpp"

namespace sa {

// LOCAL: the byte in its local form (sa_kernels.h: SA_LD_*) -- this cell's own five comparisons, for the tile walkers.
template <int CPL, int SUBST, int R, bool LOCAL = false>
__device__ __forceinline__ void nw_dirs_x1_wave(const SaFillParams &p, uint8_t *__restrict__ dirs_arena, const uint32_t pair,
                                                const int lane, uint8_t *ring_d, const int32_t *table) {
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const int open1 = p.open1, ext = p.ext, K = (int)p.K, gen_eq = p.gen_eq, gen_ne = p.gen_ne, floor_ = p.floor;
  const Border bd{p.floor, p.gap_open, p.ext, false, false};

  // BLK (round 6, sa_kernels.h): the direction bytes in blocks of 8 rows x 16 columns; the LDS buffer is one block row of the pair,
  // laid out as it lies in memory, and leaves 16 bytes per lane after its eighth row (sa_fill_dirs_x2.hip: nw_dirs_x2_wave).  The
  // host starts every pair on a multiple of 256 bytes then.
  constexpr bool BLK = SA_DIRS_BLOCKED != 0 && kWave * CPL <= 512;
  static_assert(!BLK || kWave * CPL * 8 <= R, "the LDS buffer holds a block row of 64 x CPL columns");
  const uint32_t nbx = (W + 15u) >> 4;
  uint32_t cur_row = 0;
  uint32_t cb[BLK ? CPL : 1];
  if constexpr (BLK) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) { const uint32_t g = (uint32_t)(lane * CPL + c); cb[c] = (g >> 4) * 128u + (g & 15u); }
  }
  const uint32_t a0 = (uint32_t)((uintptr_t)(dirs_arena + mo) & 255u);
  uint8_t *const gd = dirs_arena + mo - a0;
  const uint32_t vend = a0 + W * (lb + 1);
  uint32_t wv = a0, rv = 0;
  auto flush_block = [&]() __attribute__((always_inline)) {
    const uint32_t d4 = *reinterpret_cast<const uint32_t *>(ring_d + (rv & (R - 1)) + 4 * lane);
    if (rv >= a0 && rv + 256 <= vend) {
      __builtin_nontemporal_store(d4, reinterpret_cast<uint32_t *>(gd + rv + 4 * lane));
    } else {
      const uint32_t e = rv + 4 * lane;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (e + t >= a0 && e + t < vend) gd[e + t] = (uint8_t)(d4 >> (8 * t));
    }
    rv += 256;
  };
  auto append_row = [&](const uint32_t (&dv)[CPL]) __attribute__((always_inline)) {
    static_assert(255 + kWave * CPL <= R, "ring too small for unpredicated appends");
    if constexpr (BLK) {
      typedef uint32_t blk_v4 __attribute__((ext_vector_type(4)));
      const uint32_t ro = (cur_row & 7u) << 4;
#pragma unroll
      for (int c = 0; c < CPL; ++c) ring_d[cb[c] + ro] = (uint8_t)dv[c];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if ((cur_row & 7u) == 7u || cur_row == lb) {
        const uint32_t bytes = nbx * 128u;
        uint8_t *dst = dirs_arena + mo + (uint64_t)(cur_row >> 3) * bytes;
        for (uint32_t o = (uint32_t)lane * 16u; o < bytes; o += kWave * 16u)
          __builtin_nontemporal_store(*reinterpret_cast<const blk_v4 *>(ring_d + o), reinterpret_cast<blk_v4 *>(dst + o));
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      return;
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) ring_d[(wv + lane * CPL + c) & (R - 1)] = (uint8_t)dv[c];
    wv += W;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    while (wv - rv >= 256u) flush_block();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };

  int fa[CPL], arow[CPL], X[CPL], Yp[CPL], Ap[CPL], c1[CPL], c2[CPL], c3[CPL];
  uint32_t T[CPL], TY[CPL];
  int mv[CPL], av[CPL], bv[CPL];   // the row just computed (after the loop: the last row, for the end cell)
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const uint32_t g = lane * CPL + c;
    const int code = (g >= 1 && g <= la) ? (int)p.code[sa_[g - 1]] : 0;
    fa[c] = code & 0xff;
    arow[c] = (code >> 8) * K;
    // row 0 (alignment.c:46-69)
    mv[c] = av[c] = g ? floor_ : 0;
    bv[c] = g ? bd.edge_gap(g) : 0;
    X[c] = max3i(mv[c], av[c], bv[c]); Yp[c] = max(mv[c], bv[c]); Ap[c] = av[c];
    T[c] = (av[c] == X[c]) ? 1u : (bv[c] == X[c]) ? 2u : 0u;
    TY[c] = (bv[c] >= mv[c]) ? 2u : 0u;
    const int g_ext = (int)g * ext;                         // (ext <= 0 in this kernel's domain)
    c1[c] = open1 - g_ext; c2[c] = floor_ - g_ext; c3[c] = g_ext;
  }
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  {
    uint32_t dv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) dv[c] = 0;   // row 0 is never stood on with a move to make
    append_row(dv);
  }

  int chunk_code = 0;
  for (uint32_t j = 1; j <= lb; ++j) {
    const int q = (j - 1) & (kWave - 1);
    cur_row = j;
    if (q == 0) {
      const uint32_t r = j + lane;
      if (r <= lb) chunk_code = p.code[sb_[r - 1]];
      __builtin_amdgcn_s_waitcnt(kWaitVm0);
    }
    const int code_b = read_lane(chunk_code, q);
    const int x_ul = wave_shr1(X[CPL - 1], floor_);
    const uint32_t t_ul = (uint32_t)wave_shr1((int)T[CPL - 1], 0);
    const int edge_a = bd.edge_gap(j);   // gap_a of the border cell (0, j) (alignment.c:72-80)
    int z[CPL];
    uint32_t dv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int s = subst_score<SUBST>(fa[c], arow[c], code_b, table, gen_eq, gen_ne);
      const int xd = c ? X[c - (c ? 1 : 0)] : x_ul;
      const uint32_t td = c ? T[c - (c ? 1 : 0)] : t_ul;
      int m = max(addw(xd, s), floor_);                                                   // alignment.c:101-116
      const int ae = addw(Ap[c], ext);
      int a = max3i(addw(Yp[c], open1), ae, floor_);                                      // alignment.c:128-135
      if (c == 0) { m = lane == 0 ? floor_ : m; a = lane == 0 ? edge_a : a; }             // the border column
      const uint32_t dA = (ae == a) ? 1u : TY[c];
      mv[c] = m; av[c] = a; z[c] = max(m, a);
      if constexpr (LOCAL) dv[c] = (ae == a) ? SA_LD_CA : 0u;
      else dv[c] = td | (dA << 2);
    }
    {
      const int zin = wave_shr1(z[CPL - 1], z[CPL - 1]);
      int P[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int zl = (c == 0) ? zin : z[c - 1];
        int w = max(addw(zl, c1[c]), c2[c]);
        if (c == 0) w = (lane == 0) ? c2[0] : w;     // gap_b of (0, j) is the floor
        P[c] = (c == 0) ? w : max(P[c - 1], w);
      }
      const int incl = wave_scan_max(P[CPL - 1]);
      const int e = wave_shr1(incl, INT32_MIN);
#pragma unroll
      for (int c = 0; c < CPL; ++c) bv[c] = addw(max(P[c], e), c3[c]);
    }
    {
      const int al = wave_shr1(av[CPL - 1], 0), bl = wave_shr1(bv[CPL - 1], 0);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int aL = c ? av[c - (c ? 1 : 0)] : al, bL = c ? bv[c - (c ? 1 : 0)] : bl;
        const int b = bv[c];
        const uint32_t dB = (addw(aL, open1) == b) ? 1u : (addw(bL, ext) == b) ? 2u : 0u;
        const int xn = max(z[c], b);
        if constexpr (LOCAL) {
          dv[c] |= ((addw(aL, open1) == b) ? SA_LD_FA : 0u) | ((addw(bL, ext) == b) ? SA_LD_FB : 0u) |
                   ((av[c] == xn) ? SA_LD_GA : 0u) | ((b >= mv[c]) ? SA_LD_BM : 0u);
        } else {
          dv[c] |= dB << 4;
        }
        X[c] = xn; Yp[c] = max(mv[c], b); Ap[c] = av[c];
        T[c] = (av[c] == xn) ? 1u : (b == xn) ? 2u : 0u;
        TY[c] = (b >= mv[c]) ? 2u : 0u;
      }
    }
    if constexpr (LOCAL) dv[0] = lane == 0 ? 0u : dv[0];   // the border column: never stood on with a move to make, never arrived at with one either
    append_row(dv);
  }
  if constexpr (!BLK) { while (rv < wv) flush_block(); }

  // the end cell (la, lb): score and matrix the walk starts in (needleman_wunsch.c:53-66)
  const int owner = (int)(la / CPL), oc = (int)(la % CPL);
  int em = 0, ea = 0, eb = 0;
#pragma unroll
  for (int c = 0; c < CPL; ++c) { if (c == oc) { em = mv[c]; ea = av[c]; eb = bv[c]; } }
  if (lane == owner) {
    int score = em;
    uint32_t st = 0;                                  // MATCH
    if (eb >= score) { st = 2; score = eb; }          // GAP_B
    if (ea >= score) { st = 1; score = ea; }          // GAP_A
    p.best_score[pair] = score;
    p.best_index[pair] = st;
    p.status[pair] = ~0ull;
  }
}

}  // namespace sa
