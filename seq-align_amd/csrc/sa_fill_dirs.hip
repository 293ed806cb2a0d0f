// sa_fill_dirs.hip -- the fill of the SW multi-hit path: match_scores + one byte of DIRECTIONS per cell.
//
// seqalign_sw_batch(max_hits > 1) never hands its matrices to anybody: the reverse sweep (sa_sw_sweep.hip) and the
// hit tracebacks are their only readers, and all they take from gap_a_scores / gap_b_scores is the answer to one
// question per cell and state -- "which matrix does a walk standing here step back into?", the three equality tests
// of alignment_reverse_move (reference src/alignment.c:311-327: GAP_A first, then GAP_B, then MATCH).  Round 2's
// sweep re-derived that from 12 B per cell it had to load (6 row registers per column rotating every row, ~33 VALU
// instructions per cell for the decisions).  The fill has every operand of those tests in registers when it
// computes the cell, so this kernel answers the question THERE and stores
//     match_scores[cell]          int32, the reference's layout (dense pitch len_a + 1) -- candidates, keys
//     dirs[cell]                  uint8: bits 0-1 / 2-3 / 4-5 = where a walk in state MATCH / GAP_A / GAP_B goes from
//                                 this cell: 0 MATCH, 1 GAP_A, 2 GAP_B of the predecessor cell, 3 = this state's
//                                 score is 0: the walk ends here (smith_waterman.c:187-199: a hit)
// 5 B per cell instead of 12, and gap_a / gap_b never leave the registers.  Same arithmetic as the other row sweeps
// (sa_rowsweep.hpp: de-trended prefix max for gap_b), same LDS-ring stream writer idea as sa_fill_stream.hpp (every
// global store an aligned block: 1 KiB of scores, 256 B of directions), same candidate report (count, box, columns
// per row) as the stream kernel's SA_STREAM_CAND mode.
//
// Domain: Smith-Waterman, "plain" scorings (no free / forbidden gaps, no sentinel scores, gap_open <= 0 -- what every
// BASELINE SW config is), rows up to 1 024 columns (the sweep's rows-in-registers forms; 513 and up: round 5).  Everything else takes the
// three-matrix path.  The decisions are the SAME expressions the sweep evaluated (sa_sw_sweep.hip, `plain`), on the
// same int32 values; tests/test_gpu_parity.py runs every hit-list test through both paths.
#include <algorithm>

#include "sa_rowsweep.hpp"
#include "sa_fill_nw_dirs_x1.hpp"

namespace sa {

template <int CPL, int SUBST, int R>
__global__ void __launch_bounds__(kWave * 4)
fill_dirs_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  // LDS per wave: R ints (scores ring) + R bytes (directions ring); the substitution table behind the rings
  constexpr uint32_t kWaveLds = R * 4u + R;
  const uint32_t waves = blockDim.x >> 6;
  const int32_t *table = p.table;
  if constexpr (SUBST == SA_SUBST_LDS) {
    int32_t *tbl = lds + (waves * kWaveLds) / 4;
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) tbl[k] = p.table[k];
    __syncthreads();
    table = tbl;
  }
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t pair = blockIdx.x * waves + wave;
  if (pair >= p.n_pairs) return;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const int open1 = p.open1, ext = p.ext, K = (int)p.K, gen_eq = p.gen_eq, gen_ne = p.gen_ne;

  char *ring = reinterpret_cast<char *>(lds) + wave * kWaveLds;
  int32_t *ring_m = reinterpret_cast<int32_t *>(ring);
  uint8_t *ring_d = reinterpret_cast<uint8_t *>(ring + R * 4u);
  // stream positions (wave-uniform): virtual index v of a cell = a0 + its index in the pair; v % 256 == 0 is a
  // 1 KiB boundary of the scores and a 256 B boundary of the directions (arena bases are 1 KiB aligned: host check)
  const uint32_t a0 = (uint32_t)(((uintptr_t)(p.M + mo) >> 2) & 255u);
  int32_t *const gm = p.M + mo - a0;
  uint8_t *const gd = dirs_arena + mo - a0;
  const uint32_t vend = a0 + W * (lb + 1);
  uint32_t wv = a0, rv = 0;

  auto flush_block = [&]() __attribute__((always_inline)) {
    typedef int v4i_a __attribute__((ext_vector_type(4)));
    const uint32_t ro = rv & (R - 1);
    const v4i_a q = *reinterpret_cast<const v4i_a *>(ring_m + ro + 4 * lane);
    const uint32_t d4 = *reinterpret_cast<const uint32_t *>(ring_d + ro + 4 * lane);
    if (rv >= a0 && rv + 256 <= vend) {
      __builtin_nontemporal_store(q, reinterpret_cast<v4i_a *>(gm + rv + 4 * lane));
      __builtin_nontemporal_store(d4, reinterpret_cast<uint32_t *>(gd + rv + 4 * lane));
    } else {   // first / last block of the pair: per-cell predicates
      const uint32_t e = rv + 4 * lane;
      const int qs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (e + t >= a0 && e + t < vend) { gm[e + t] = qs[t]; gd[e + t] = (uint8_t)(d4 >> (8 * t)); }
    }
    rv += 256;
  };
  auto append_row = [&](const int (&mv)[CPL], const uint32_t (&dv)[CPL]) __attribute__((always_inline)) {
    static_assert(255 + kWave * CPL <= R, "ring too small for unpredicated appends");
    // all 64 lanes write, also those past the row's end (cells that are overwritten before they can be flushed)
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t at = (wv + lane * CPL + c) & (R - 1);
      ring_m[at] = mv[c];
      ring_d[at] = (uint8_t)dv[c];
    }
    wv += W;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // one wave: LDS ops execute in order; pins the compiler
    while (wv - rv >= 256u) flush_block();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };

  // my columns: matrix column g = lane * CPL + c (column 0 = the border column, lane 0).  Carried from the previous
  // row, per column: X = max3(M, A, B), Yp = max(M, B), Ap = A, and two tags that answer the traceback's questions
  // about that cell once, when it is produced, instead of at each of its three readers:
  //   T  = which of the three IS the max3, in the traceback's order (GAP_A = 1 first, then GAP_B = 2, else MATCH = 0):
  //        a walk in MATCH at (x, y) with M(x, y) > 0 came from exactly that -- M(x, y) = max3(x-1, y-1) + s, so
  //        "A + s == M" (alignment.c:311-327) is "A == max3";
  //   TY = 2 if B >= M: a walk in GAP_A with A(x, y) > 0 that does not continue the gap (A' + ext != A) opened it from
  //        max(M', B') + open1, from B' iff B' >= M'.
  int fa[CPL], arow[CPL], X[CPL], Yp[CPL], Ap[CPL], c1[CPL], c2[CPL], c3[CPL];
  uint32_t T[CPL], TY[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const uint32_t g = lane * CPL + c;
    const int code = (g >= 1 && g <= la) ? (int)p.code[sa_[g - 1]] : 0;
    fa[c] = code & 0xff;
    arow[c] = (code >> 8) * K;
    X[c] = Yp[c] = Ap[c] = 0;                               // row 0 (smith_waterman: borders are 0, alignment.c:51-57)
    T[c] = 1u; TY[c] = 2u;                                  // (A == max3 and B >= M hold on a row of zeros; never followed: scores are 0)
    const int g_ext = (int)g * ext;                         // (ext <= 0 in this kernel's domain)
    c1[c] = open1 - g_ext; c2[c] = 0 - g_ext; c3[c] = g_ext;   // floor = 0
  }
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  const int ncol = max(0, min(CPL, (int)W - lane * CPL));
  // The border column (lane 0, c = 0) needs no special case: its up-left is "minus infinity" (so M = max(.., 0) = 0),
  // A = max3(0 + open1, 0 + ext, 0) = 0 because open1, ext <= 0, and the scan gives B(0) = floor = 0.
  constexpr int kMinusInf = -(1 << 29);

  // candidates (as the stream kernel's SA_STREAM_CAND): any cell >= min_score, the box, the columns per row
  uint32_t cand_n = 0, box_rmin = 0xffffffffu, box_rmax = 0, box_cmin = 0xffffffffu, box_cmax = 0;
  const int cand_thr = max(p.cand_min[pair], 1);
  uint32_t *cand_rows = p.cand_rows + 2ull * p.cand_rows_off[pair];
  if (lane == 0) *reinterpret_cast<uint2 *>(cand_rows) = make_uint2(0xffffffffu, 0u);   // row 0: borders only
  uint32_t rr_lo = 0xffffffffu, rr_hi = 0;

  {  // row 0: scores 0, every state ends
    int mv[CPL];
    uint32_t dv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { mv[c] = 0; dv[c] = 0x3fu; }
    append_row(mv, dv);
  }

  int chunk_code = 0;
  for (uint32_t j = 1; j <= lb; ++j) {
    const int q = (j - 1) & (kWave - 1);
    if (q == 0) {
      const uint32_t r = j + lane;
      if (r <= lb) chunk_code = p.code[sb_[r - 1]];
      __builtin_amdgcn_s_waitcnt(kWaitVm0);
    }
    const int code_b = read_lane(chunk_code, q);
    // up-left of my first column: the left lane's last column on the previous row
    const int x_ul = wave_shr1(X[CPL - 1], kMinusInf);
    const uint32_t t_ul = (uint32_t)wave_shr1((int)T[CPL - 1], 0);
    int mv[CPL], av[CPL], bv[CPL], z[CPL];
    uint32_t dv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int s = subst_score<SUBST>(fa[c], arow[c], code_b, table, gen_eq, gen_ne);
      const int xd = c ? X[c - (c ? 1 : 0)] : x_ul;
      const uint32_t td = c ? T[c - (c ? 1 : 0)] : t_ul;
      const int m = max(addw(xd, s), 0);                                                  // alignment.c:101-116
      const int ae = addw(Ap[c], ext);
      const int a = max3i(addw(Yp[c], open1), ae, 0);                                     // alignment.c:128-135
      // where a walk goes from here (alignment.c:311-327: GAP_A tested first, then GAP_B, else MATCH); 3: score 0
      const uint32_t dM = m > 0 ? td : 3u;
      const uint32_t dA = a > 0 ? (ae == a ? 1u : TY[c]) : 3u;
      mv[c] = m; av[c] = a; z[c] = max(m, a);
      dv[c] = dM | (dA << 2);
    }
    // gap_b: de-trended prefix max (sa_rowsweep.hpp)
    {
      const int zin = wave_shr1(z[CPL - 1], z[CPL - 1]);
      int P[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int zl = (c == 0) ? zin : z[c - 1];
        int w = max(addw(zl, c1[c]), c2[c]);
        if (c == 0) w = (lane == 0) ? c2[0] : w;     // gap_b of (0, j) is the floor (0)
        P[c] = (c == 0) ? w : max(P[c - 1], w);
      }
      const int incl = wave_scan_max(P[CPL - 1]);
      const int e = wave_shr1(incl, INT32_MIN);
#pragma unroll
      for (int c = 0; c < CPL; ++c) bv[c] = addw(max(P[c], e), c3[c]);
    }
    {
      const int al = wave_shr1(av[CPL - 1], 0), bl = wave_shr1(bv[CPL - 1], 0);   // the left lane's last column, this row
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int aL = c ? av[c - (c ? 1 : 0)] : al, bL = c ? bv[c - (c ? 1 : 0)] : bl;
        const int b = bv[c];
        const uint32_t dB = b > 0 ? ((addw(aL, open1) == b) ? 1u : (addw(bL, ext) == b) ? 2u : 0u) : 3u;
        dv[c] |= dB << 4;
        // this cell for the row below
        const int xn = max(z[c], b);
        X[c] = xn; Yp[c] = max(mv[c], b); Ap[c] = av[c];
        T[c] = (av[c] == xn) ? 1u : (b == xn) ? 2u : 0u;
        TY[c] = (b >= mv[c]) ? 2u : 0u;
      }
    }
    append_row(mv, dv);


    // candidates of this row (lane granularity is enough: the sweep needs bounds)
    unsigned long long any = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) any |= __ballot(c < ncol && mv[c] >= cand_thr);
    uint32_t row_lo = 0xffffffffu, row_hi = 0;
    if (any) {
      row_lo = (uint32_t)__builtin_ctzll(any) * CPL;
      row_hi = (uint32_t)(63 - __builtin_clzll(any)) * CPL + (CPL - 1);
      cand_n = 1;
      box_cmin = min(box_cmin, row_lo);
      box_cmax = max(box_cmax, row_hi);
      box_rmin = min(box_rmin, j);
      box_rmax = j;
    }
    if (lane == q) { rr_lo = row_lo; rr_hi = row_hi; }
    if (q == kWave - 1 || j == lb) {
      if (lane <= q) *reinterpret_cast<uint2 *>(cand_rows + 2ull * (j - q + lane)) = make_uint2(rr_lo, min(rr_hi, W - 1));
    }
  }
  while (rv < wv) flush_block();

  if (lane == 0) {
    p.cand_count[pair] = cand_n;
    uint32_t *box = p.cand_box + 4ull * pair;
    box[0] = box_rmin; box[1] = box_rmax; box[2] = box_cmin; box[3] = min(box_cmax, W - 1);
    p.status[pair] = ~0ull;   // plain scorings have a score for every pair of characters
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Needleman-Wunsch for seqalign_nw_batch: the SAME idea one step further.  The host-level call returns strings and
// scores, never matrices, and its traceback (needleman_wunsch.c:53-145) needs from the three matrices exactly what
// the byte above holds -- so for plain scorings this kernel writes ONLY the direction byte per cell (1 B instead of
// 12) plus, per pair, the end cell's score and state (needleman_wunsch.c:53-66: GAP_A >= GAP_B >= MATCH on ties).
// The walk then costs one byte load per step instead of three ints in three arenas.
// Borders (alignment.c:46-81): (0,0) = 0; row 0: M = A = floor, B = gap_open + i*ext; column 0: M = B = floor,
// A = gap_open + j*ext.  No state ever "ends" (code 3 is not used): the walk stops at x == 0 or y == 0.
// Cells whose score is the floor by clamping carry a meaningless direction, like the reference's traceback (which
// exits on them, alignment.c:328-349) they are never stood on inside the parity domain.
template <int CPL, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_nw_dirs_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t waves = blockDim.x >> 6;
  const int32_t *table = p.table;
  if constexpr (SUBST == SA_SUBST_LDS) {
    int32_t *tbl = lds + (waves * R) / 4;
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) tbl[k] = p.table[k];
    __syncthreads();
    table = tbl;
  }
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = blockIdx.x * waves + wave;
  if (slot >= p.n_pairs) return;
  const uint32_t pair = p.pair_list ? p.pair_list[slot] : slot;

  nw_dirs_x1_wave<CPL, SUBST, R, LOCAL>(p, dirs_arena, pair, lane, reinterpret_cast<uint8_t *>(lds) + wave * R, table);
}

template <int CPL, int R0>
static hipError_t launch_nw_dirs_cpl(const SaFillParams &p, uint8_t *dirs, hipStream_t stream) {
  // (blocked direction bytes, sa_kernels.h: one block row of 64 x CPL columns per pair in LDS instead of the ring)
  constexpr int R = (SA_DIRS_BLOCKED != 0 && kWave * CPL <= 512 && kWave * CPL * 8 > R0) ? kWave * CPL * 8 : R0;
  const int wpb = 4;
  const dim3 grid((p.n_pairs + wpb - 1) / wpb), block(kWave * wpb);
  const size_t rings = (size_t)wpb * R;
  const size_t lds = rings + (((size_t)p.K * p.K + 3u) & ~(size_t)3u) * sizeof(int32_t);
  // (dirs_local: the byte's local form for the tile walkers, sa_kernels.h)
  if (p.K <= 1 && p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_kernel<CPL, SA_SUBST_SIMPLE, R, true>), grid, block, rings, stream, p, dirs);
  else if (p.K <= 1) hipLaunchKernelGGL((fill_nw_dirs_kernel<CPL, SA_SUBST_SIMPLE, R, false>), grid, block, rings, stream, p, dirs);
  else if (p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_kernel<CPL, SA_SUBST_LDS, R, true>), grid, block, lds, stream, p, dirs);
  else hipLaunchKernelGGL((fill_nw_dirs_kernel<CPL, SA_SUBST_LDS, R, false>), grid, block, lds, stream, p, dirs);
  return hipGetLastError();
}

template <int CPL, int R>
static hipError_t launch_dirs_cpl(const SaFillParams &p, uint8_t *dirs, hipStream_t stream) {
  const int wpb = 4;
  const dim3 grid((p.n_pairs + wpb - 1) / wpb), block(kWave * wpb);
  const size_t rings = (size_t)wpb * (R * 4u + R);
  if (p.K <= 1) {
    hipLaunchKernelGGL((fill_dirs_kernel<CPL, SA_SUBST_SIMPLE, R>), grid, block, rings, stream, p, dirs);
  } else {
    const size_t lds = rings + (((size_t)p.K * p.K + 3u) & ~(size_t)3u) * sizeof(int32_t);
    hipLaunchKernelGGL((fill_dirs_kernel<CPL, SA_SUBST_LDS, R>), grid, block, lds, stream, p, dirs);
  }
  return hipGetLastError();
}

}  // namespace sa

bool sa_dirs_fill_applicable(const SaFillParams &p, uint32_t max_len_a, const uint8_t *dirs) {
  // plain scorings only: no free / forbidden gaps, no sentinel scores (the sweep's and the walkers' `plain`), gap_open <= 0
  if (!sa_domain_sw_dirs(sa_traits_of(p), max_len_a)) return false;   // (ext <= 0: what makes the border column come out by itself)
  if (!p.cand_count || !p.cand_box || !p.cand_rows || !p.cand_rows_off || !p.cand_min) return false;
  return ((uintptr_t)p.M & 1023) == 0 && ((uintptr_t)dirs & 255) == 0;   // block boundaries of both streams coincide
}

hipError_t sa_launch_fill_dirs(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  sa_record_launch(SEQALIGN_K_FILL_SW_DIRS, p.n_pairs);
  const uint32_t need = sa::columns_per_lane(max_len_a + 1, p.tune_cpl);
  if (need <= 1) return sa::launch_dirs_cpl<1, 512>(p, dirs, stream);
  if (need <= 2) return sa::launch_dirs_cpl<2, 512>(p, dirs, stream);
  if (need <= 3) return sa::launch_dirs_cpl<3, 512>(p, dirs, stream);
  if (need <= 4) return sa::launch_dirs_cpl<4, 512>(p, dirs, stream);
  if (need <= 5) return sa::launch_dirs_cpl<5, 1024>(p, dirs, stream);
  if (need <= 6) return sa::launch_dirs_cpl<6, 1024>(p, dirs, stream);
  if (need <= 8) return sa::launch_dirs_cpl<8, 1024>(p, dirs, stream);
  if (need <= 12) return sa::launch_dirs_cpl<12, 1024>(p, dirs, stream);   // (rows of 513 .. 1 024 columns: round 5)
  return sa::launch_dirs_cpl<16, 2048>(p, dirs, stream);
}

// ---- Needleman-Wunsch: directions only (seqalign_nw_batch)
bool sa_nw_dirs_fill_applicable(const SaFillParams &p, uint32_t max_len_a, const uint8_t *dirs) {
  // plain scorings only: no flag at all, no sentinel scores, gap_open <= 0, gap_extend <= 0
  if (!sa_domain_nw_dirs(sa_traits_of(p), max_len_a)) return false;
  return dirs && p.best_score && p.best_index && ((uintptr_t)dirs & 255) == 0;
}

hipError_t sa_launch_fill_nw_dirs(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  sa_record_launch(SEQALIGN_K_FILL_NW_DIRS, p.n_pairs);
  const uint32_t need = sa::columns_per_lane(max_len_a + 1, sa_dirs_blocked_shape(max_len_a) ? std::min<uint32_t>(p.tune_cpl, 8u) : p.tune_cpl);
  if (need <= 1) return sa::launch_nw_dirs_cpl<1, 512>(p, dirs, stream);
  if (need <= 2) return sa::launch_nw_dirs_cpl<2, 512>(p, dirs, stream);
  if (need <= 3) return sa::launch_nw_dirs_cpl<3, 512>(p, dirs, stream);
  if (need <= 4) return sa::launch_nw_dirs_cpl<4, 512>(p, dirs, stream);
  if (need <= 5) return sa::launch_nw_dirs_cpl<5, 1024>(p, dirs, stream);
  if (need <= 6) return sa::launch_nw_dirs_cpl<6, 1024>(p, dirs, stream);
  if (need <= 8) return sa::launch_nw_dirs_cpl<8, 1024>(p, dirs, stream);
  if (need <= 12) return sa::launch_nw_dirs_cpl<12, 1024>(p, dirs, stream);   // (rows of 513 .. 1 024 columns: round 5)
  return sa::launch_nw_dirs_cpl<16, 2048>(p, dirs, stream);
}
