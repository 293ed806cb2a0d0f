// sa_fill_wgstream.hip -- long rows (513 .. 4 096 columns), many pairs: ONE WORKGROUP
// per pair, the row split over four waves, output through a shared LDS ring in aligned
// 1 KiB blocks.
//
// Replaces alignment_fill_matrices (reference src/alignment.c:28-168) where a row does
// not fit one wave.  Until now such pairs went through sa_fill_rowscan.hip / strips,
// whose row segments start at arbitrary 4-byte offsets (one TA cycle per lane and store:
// 2 000 x 2 000^2 ran at 0.33 of the HBM roofline); this is sa_fill_stream.hip's idea --
// a pair's matrix is one contiguous stream, flush it as a memset would -- with the row
// produced by 4 x 64 (up to 2 048 columns) or 8 x 64 lanes instead of 64.
//
// Per row (wave w owns columns [w*64*CPL, (w+1)*64*CPL), column 0 = the border column):
//   1. match / gap_a from registers (one DPP shift; lane 0 of wave w gets the previous
//      row's max3 of its left neighbour column, see 4.);
//   2. gap_b: the (max,+) prefix scan of sa_rowsweep.hpp, de-trended so that it is a plain
//      prefix max.  Each wave scans its own columns WITHOUT the term that comes in from
//      its left neighbour and publishes {z_last, local total} in LDS;
//   3. ONE barrier; every wave then combines the published values: carry into wave w =
//      max(totals of the waves to its left, the incoming terms t_u = z_last[u-1] + c1 of
//      waves 1..w).  max is associative, the adds are the ones the single-wave kernel does:
//      bit-identical results;
//   4. the same published values give the left neighbour's max3 of THIS row (needed by lane
//      0 in step 1 of the next row), so no second barrier; the slots are double-buffered
//      by row parity;
//   5. the row goes into the shared ring; blocks completed by the PREVIOUS row are flushed
//      (block k by wave k mod 4) -- every wave is past the barrier of this row, so all of
//      the previous row is in LDS.
//
// GENERAL (no_end_gap / no_gaps_in_* / sentinels / gap_open > 0) is a template flag with the
// case analysis of RowSweep<.., GENERAL>; a free last row combines without the penalties, a
// forced row (no_gaps_in_b) is the floor.
#include "sa_rowsweep.hpp"

namespace sa {

constexpr int kWgMaxWaves = 8;   // 4 waves up to 2 048 columns, 8 up to 4 096 (8 columns per lane)

// TIGHT (8 waves): the ring only holds the backlog plus ONE row (not the extra row of all lanes), so three
// workgroups fit a CU instead of one: idle lanes do not write, and a second barrier separates the flush of
// the previous rows from the append of the next.
// MODE: 0 = the matrices only; 1 (SW) = also the pair's best cell; 2 (SW) = also where the cells >= cand_min[pair] are
// (SaFillParams::cand_*: any at all, bounding box, lowest / highest column per row -- for the multi-hit path's sweep)
template <int CPL, int SUBST, int kWgWaves, bool GENERAL, int MODE>
__global__ void __launch_bounds__(kWave *kWgWaves)
fill_wgstream_kernel(const SaFillParams p, const uint32_t R /* ring ints per matrix */) {
  constexpr bool BEST = MODE == 1, CAND = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  // LDS: [3 rings of R ints][slots: 2 parities x 4 waves x {z_last, total}][substitution table]
  int32_t *ring = lds;
  int32_t *slots = lds + 3 * R;
  const int32_t *table = p.table;
  if constexpr (SUBST == SA_SUBST_LDS) {
    int32_t *tbl = slots + 2 * kWgWaves * 2;
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) tbl[k] = p.table[k];
    __syncthreads();
    table = tbl;
  }
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t pair = blockIdx.x;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair];
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const uint32_t W = la + 1;
  const int floor_ = p.floor, open1 = p.open1, ext = p.ext;
  const int trend0 = ext > 0 ? (int)(kWgWaves * kWave * CPL) * ext : 0;   // gap_extend > 0: trend from the right end (sa_rowsweep.hpp)
  const Border bd{p.floor, p.gap_open, p.ext, (p.flags & SA_F_IS_SW) != 0, (p.flags & SA_F_NO_START_GAP) != 0};
  // GENERAL (reference alignment.c:101-155): no_end_gap / no_gaps_in_a / no_gaps_in_b, sentinels in the
  // substitution scores, gap_open > 0 -- same case analysis as RowSweep<.., GENERAL> in sa_rowsweep.hpp
  const bool no_end = (p.flags & SA_F_NO_END_GAP) != 0, no_gaps_a = (p.flags & SA_F_NO_GAPS_A) != 0,
             no_gaps_b = (p.flags & SA_F_NO_GAPS_B) != 0;
  __shared__ unsigned long long s_err;          // first cell without a score (GENERAL)
  __shared__ unsigned long long s_best;         // BEST: max over the workgroup of score << 32 | ~(column << 21 | row)
  __shared__ uint32_t s_row_lo[2], s_row_hi[2]; // CAND: this row's candidate columns over the waves, by row parity
  if (threadIdx.x == 0) { s_err = ~0ull; s_best = 0ull; s_row_lo[0] = s_row_lo[1] = 0xffffffffu; s_row_hi[0] = s_row_hi[1] = 0u; }
  __syncthreads();   // the other waves' first atomics on these words (row 1, before the row's own barrier) come after the init
  unsigned long long err = ~0ull;
  int cand_thr = INT32_MAX;
  uint32_t *cand_rows = nullptr;
  uint32_t box_rmin = 0xffffffffu, box_rmax = 0, box_cmin = 0xffffffffu, box_cmax = 0;   // (thread 0)
  if constexpr (CAND) {
    cand_thr = max(p.cand_min[pair], 1);
    cand_rows = p.cand_rows + 2ull * p.cand_rows_off[pair];
    if (threadIdx.x == 0) *reinterpret_cast<uint2 *>(cand_rows) = make_uint2(0xffffffffu, 0u);   // row 0: borders only
  }

  // stream positions (all wave-uniform, identical in the four waves)
  const uint32_t a0 = (uint32_t)(((uintptr_t)(p.M + mo) >> 2) & 255u);   // arenas are congruent mod 4 KiB
  int32_t *const g0[3] = {p.M + mo - a0, p.A + mo - a0, p.B + mo - a0};
  const uint32_t vend = a0 + W * (lb + 1);
  uint32_t rv = 0;                              // flushed up to here (multiple of 256)

  // my columns: g = L*CPL + c, L = wave*64 + lane
  const uint32_t g_first = (wave * kWave + (uint32_t)lane) * CPL;
  int fa[CPL], arow[CPL], X[CPL], Ap[CPL], c1[CPL], c2[CPL], c3[CPL];
  int Y[GENERAL ? CPL : 1];                     // previous row: max(M, B) (GENERAL: gap_a opens from it)
  uint32_t wr[CPL];                             // ring index of my cells in the row being written
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const uint32_t g = g_first + c;             // matrix column; sequence index g-1
    const int code = (g >= 1 && g <= la) ? (int)p.code[sa_[g - 1]] : 0;
    fa[c] = code & 0xff;
    arow[c] = (code >> 8) * (int)p.K;
    const int b0 = bd.edge_gap(g);              // row 0: M = A = floor, B = edge
    X[c] = (g == 0) ? 0 : max(floor_, b0);
    if constexpr (GENERAL) Y[c] = X[c];
    Ap[c] = (g == 0) ? 0 : floor_;
    const int g_ext = (int)g * ext - trend0;   // -t(g), sa_rowsweep.hpp
    c1[c] = open1 - g_ext; c2[c] = floor_ - g_ext; c3[c] = g_ext;
    wr[c] = (a0 + g) % R;
  }
  int boundX = 0;                               // lane 0, wave >= 1: max3 of (my first column - 1, previous row)
  if (wave > 0) { const uint32_t g = wave * kWave * CPL - 1; boundX = max(floor_, bd.edge_gap(g)); }
  __builtin_amdgcn_s_waitcnt(kWaitVm0);

  constexpr bool TIGHT = kWgWaves > 4;
  auto append = [&](const int (&mv)[CPL], const int (&av)[CPL], const int (&bv)[CPL]) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      if (!TIGHT || g_first + c < W) { ring[wr[c]] = mv[c]; ring[R + wr[c]] = av[c]; ring[2 * R + wr[c]] = bv[c]; }
      uint32_t n = wr[c] + W;                   // W <= R
      wr[c] = n >= R ? n - R : n;
    }
  };
  // flush the complete blocks below `upto` (virtual position); block k belongs to wave k mod 4
  auto flush_upto = [&](uint32_t upto, bool tail) {
    for (; rv < upto && (tail || rv + 256 <= upto); rv += 256) {
      if (((rv >> 8) & (kWgWaves - 1)) != wave) continue;
      const uint32_t ro = rv % R;               // R is a multiple of 256: a block never wraps
      typedef int v4i_a __attribute__((ext_vector_type(4)));
      const bool inside = rv >= a0 && rv + 256 <= vend;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const v4i_a q = *reinterpret_cast<const v4i_a *>(ring + m * R + ro + 4 * lane);
        int32_t *dst = g0[m] + rv + 4 * lane;
        if (inside) {
          __builtin_nontemporal_store(q, reinterpret_cast<v4i_a *>(dst));
        } else {
          const uint32_t e = rv + 4 * lane;
          if (e + 0 >= a0 && e + 0 < vend) dst[0] = q.x;
          if (e + 1 >= a0 && e + 1 < vend) dst[1] = q.y;
          if (e + 2 >= a0 && e + 2 < vend) dst[2] = q.z;
          if (e + 3 >= a0 && e + 3 < vend) dst[3] = q.w;
        }
      }
    }
  };

  {  // row 0 (reference alignment.c:46-69)
    int mv[CPL], av[CPL], bv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t g = g_first + c;
      mv[c] = av[c] = (g == 0) ? 0 : floor_;
      bv[c] = (g == 0) ? 0 : bd.edge_gap(g);
    }
    append(mv, av, bv);
  }

  // BEST (SW, p.best_score set): the pair's best match_scores cell in the reference's hit order, as in
  // sa_fill_stream.hip -- per column the highest score and the first row that reached it
  int best_s[BEST ? CPL : 1], best_r[BEST ? CPL : 1];
  if constexpr (BEST) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) { best_s[c] = 0; best_r[c] = 0; }
  }

  int chunk_code = 0;
  for (uint32_t j = 1; j <= lb; ++j) {
    const int q = (j - 1) & (kWave - 1);
    if (q == 0) {   // every 64 rows: lane t fetches seq_b's code for row j+t
      const uint32_t r = j + lane;
      if (r <= lb) chunk_code = p.code[sb_[r - 1]];
      __builtin_amdgcn_s_waitcnt(kWaitVm0);
    }
    const int code_b = read_lane(chunk_code, q);
    const int edge_a = bd.edge_gap(j);

    // ---- 1. match / gap_a
    int xd = wave_shr1(X[CPL - 1], boundX);
    int mv[CPL], av[CPL], bv[CPL], z[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int s = subst_score<SUBST>(fa[c], arow[c], code_b, table, p.gen_eq, p.gen_ne);
      int m, a;
      if constexpr (GENERAL) {
        const uint32_t g = g_first + c;
        const int a_norm = max3i(addw(Y[c], open1), addw(Ap[c], ext), floor_);
        m = (s == SA_S_BLOCKED) ? floor_ : max(addw(xd, s), floor_);
        if (s == SA_S_UNKNOWN && g >= 1 && g <= la) {
          m = floor_;
          err = min(err, (unsigned long long)j * W + g);
        }
        const bool last_col = (g == la);
        a = (last_col && no_end) ? max(Y[c], Ap[c]) : (!no_gaps_a || last_col) ? a_norm : floor_;
      } else {
        m = max(addw(xd, s), floor_);
        a = max3i(addw(X[c], open1), addw(Ap[c], ext), floor_);
      }
      if (c == 0) {   // border column (reference alignment.c:72-80): wave 0, lane 0
        const bool border = (wave == 0) && (lane == 0);
        m = border ? floor_ : m;
        a = border ? edge_a : a;
      }
      xd = X[c];
      mv[c] = m; av[c] = a; z[c] = max(m, a);
    }

    if constexpr (BEST) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const bool up = mv[c] > best_s[c];
        best_s[c] = up ? mv[c] : best_s[c];
        best_r[c] = up ? (int)j : best_r[c];
      }
    }

    if constexpr (CAND) {   // my wave's candidate columns of this row (at lane granularity), merged in LDS before the barrier
      bool mine = false;
#pragma unroll
      for (int c = 0; c < CPL; ++c) mine |= (g_first + c <= la) && mv[c] >= cand_thr;
      const unsigned long long any = __ballot(mine);
      if (any && lane == 0) {
        const uint32_t base = wave * kWave * CPL;
        atomicMin(&s_row_lo[j & 1u], base + (uint32_t)__builtin_ctzll(any) * CPL);
        atomicMax(&s_row_hi[j & 1u], min(base + (uint32_t)(63 - __builtin_clzll(any)) * CPL + (CPL - 1), la));
      }
    }

    // ---- 2. gap_b: local part of the prefix max
    bool free_row = false, forced = false;              // wave-uniform (reference alignment.c:139-155)
    if constexpr (GENERAL) {
      const bool last_row = (j == lb);
      free_row = last_row && no_end;                    // max3 of the left cell: no penalty, no clamp
      forced = no_gaps_b && !last_row;                  // gap_b is the floor
    }
    const int zin = wave_shr1(z[CPL - 1], INT32_MIN);   // lane 0: the neighbour wave's term comes in step 3
    int P[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int zl = (c == 0) ? zin : z[c - 1];
      int w = free_row ? zl : max(addw(zl, c1[c]), c2[c]);
      // lane 0: wave 0 -- gap_b of (0, j) is the floor; other waves -- only the part that does not come
      // from the neighbour (the floor term, or nothing at all on a free row)
      if (c == 0) w = (lane == 0) ? (free_row ? (wave == 0 ? floor_ : INT32_MIN) : c2[0]) : w;
      P[c] = (c == 0) ? w : max(P[c - 1], w);
    }
    const int incl = wave_scan_max(P[CPL - 1]);
    const int e = wave_shr1(incl, INT32_MIN);
    const uint32_t par = (j & 1u) * (2 * kWgWaves);
    if (lane == kWave - 1) { slots[par + 2 * wave] = z[CPL - 1]; slots[par + 2 * wave + 1] = incl; }

    // ---- 3. one barrier, then the carries
    __syncthreads();
    if constexpr (CAND) {
      if (threadIdx.x == 0) {   // (the slot of this parity is next written after the NEXT row's barrier)
        const uint32_t lo = s_row_lo[j & 1u], hi = s_row_hi[j & 1u];
        s_row_lo[j & 1u] = 0xffffffffu; s_row_hi[j & 1u] = 0u;
        *reinterpret_cast<uint2 *>(cand_rows + 2ull * j) = make_uint2(lo, hi);
        if (lo <= hi) { box_rmin = min(box_rmin, j); box_rmax = j; box_cmin = min(box_cmin, lo); box_cmax = max(box_cmax, hi); }
      }
    }
    int carry = INT32_MIN;                              // into my wave
    int left_total = INT32_MIN, left_carry = INT32_MIN, left_z = 0;
#pragma unroll
    for (int u = 0; u < kWgWaves; ++u) {
      if ((uint32_t)u < wave) {
        const int zu = slots[par + 2 * u], tu = slots[par + 2 * u + 1];
        // wave u+1's incoming term: z_last[u] + c1(first column of wave u+1)
        const int g_next = (u + 1) * kWave * CPL;
        const int t_next = free_row ? zu : addw(zu, open1 - (g_next * ext - trend0));
        if ((uint32_t)u + 1 == wave) { left_total = tu; left_carry = carry; left_z = zu; }
        carry = max(carry, max(tu, t_next));
      }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int pm = max(max(P[c], e), carry);
      bv[c] = forced ? floor_ : (free_row ? pm : addw(pm, c3[c]));
      X[c] = max(z[c], bv[c]);
      if constexpr (GENERAL) Y[c] = max(mv[c], bv[c]);
      Ap[c] = av[c];
    }
    // ---- 4. my left neighbour column's max3 of this row, for the next row's diagonal
    if (wave > 0) {
      const int g_left = (int)(wave * kWave * CPL) - 1;
      const int pm_left = max(left_total, left_carry);
      const int b_left = forced ? floor_ : (free_row ? pm_left : addw(pm_left, g_left * ext - trend0));
      boundX = max(left_z, b_left);
    }

    // ---- 5. flush what the previous rows completed, then append this row
    flush_upto(a0 + j * W, false);                      // rows 0 .. j-1 are in LDS (barrier above)
    if constexpr (TIGHT) __syncthreads();               // nobody overwrites a block that is still being flushed
    append(mv, av, bv);
  }
  __syncthreads();
  flush_upto(vend, true);
  if constexpr (GENERAL) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) err = min(err, __shfl_xor(err, o));
    if (lane == 0 && err != ~0ull) atomicMin(&s_err, err);
    __syncthreads();
  }
  if (threadIdx.x == 0) p.status[pair] = GENERAL ? s_err : ~0ull;
  if constexpr (CAND) {
    if (threadIdx.x == 0) {
      p.cand_count[pair] = box_rmin <= box_rmax ? 1u : 0u;
      uint32_t *box = p.cand_box + 4ull * pair;
      box[0] = box_rmin; box[1] = box_rmax; box[2] = box_cmin; box[3] = box_cmax;
    }
  }
  if constexpr (BEST) {
    int b = 0;
    uint32_t tie = 0;   // (column << 21) | row of the best cell; lowest column, then lowest row, wins a tie
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t g = g_first + c;
      if (g >= 1 && g <= la && best_s[c] > b) { b = best_s[c]; tie = (g << 21) | (uint32_t)best_r[c]; }
    }
    unsigned long long key = ((unsigned long long)(uint32_t)b << 32) | (uint32_t)~tie;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o);
      key = other > key ? other : key;
    }
    if (lane == 0 && (key >> 32) != 0) atomicMax(&s_best, key);
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long k = s_best;
      const uint32_t t = ~(uint32_t)k, col = t >> 21, row = t & ((1u << 21) - 1);
      const int score = (int)(k >> 32);
      p.best_score[pair] = score;
      p.best_index[pair] = score > 0 ? (uint64_t)row * W + col : 0;
    }
  }
}

}  // namespace sa

bool sa_wgstream_kernel_applicable(const SaFillParams &p, uint32_t max_len_a) {
  if (max_len_a + 1 <= 512 || max_len_a + 1 > 8 * sa::kWave * sa::kWgMaxWaves) return false;
  const uintptr_t m = (uintptr_t)p.M, a = (uintptr_t)p.A, b = (uintptr_t)p.B;
  return ((m ^ a) & 4095) == 0 && ((m ^ b) & 4095) == 0;
}

namespace sa {
template <int CPL, int NW>
static hipError_t launch_wg(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream) {
  // ring: the unflushed backlog (< 256 + W) plus one row written by all NW*64*CPL lanes, in whole blocks;
  // TIGHT (8 waves): the backlog after the flush (< 256) plus one row of W cells
  const uint32_t span = NW * kWave * CPL;
  const uint32_t R = NW > 4 ? (256 + (max_len_a + 1) + 255) / 256 * 256
                            : (256 + (max_len_a + 1) + span + 255) / 256 * 256;
  const dim3 grid(p.n_pairs), block(kWave * NW);
  size_t lds = ((size_t)3 * R + 2 * NW * 2) * sizeof(int32_t);
  const bool general = needs_general(p);
  // best-cell reporting: SW only, rows / columns that fit the packed tie-break (21 / 11+ bits)
  const bool best = p.best_score && p.best_index && (p.flags & SA_F_IS_SW);
  const bool cand = p.cand_count && p.cand_box && p.cand_rows && p.cand_rows_off && p.cand_min && (p.flags & SA_F_IS_SW);
#define SA_WG_LAUNCH(SUBST_, GEN_, LDS_)                                                                              \
  do {                                                                                                                \
    if (cand) hipLaunchKernelGGL((fill_wgstream_kernel<CPL, SUBST_, NW, GEN_, 2>), grid, block, LDS_, stream, p, R);    \
    else if (best) hipLaunchKernelGGL((fill_wgstream_kernel<CPL, SUBST_, NW, GEN_, 1>), grid, block, LDS_, stream, p, R); \
    else hipLaunchKernelGGL((fill_wgstream_kernel<CPL, SUBST_, NW, GEN_, 0>), grid, block, LDS_, stream, p, R);         \
  } while (0)
  if (p.K <= 1) {
    if (general) SA_WG_LAUNCH(SA_SUBST_SIMPLE, true, lds);
    else SA_WG_LAUNCH(SA_SUBST_SIMPLE, false, lds);
  } else if (p.K <= SA_LDS_TABLE_MAX_K) {
    lds += (size_t)p.K * p.K * sizeof(int32_t);
    if (general) SA_WG_LAUNCH(SA_SUBST_LDS, true, lds);
    else SA_WG_LAUNCH(SA_SUBST_LDS, false, lds);
  } else {
    SA_WG_LAUNCH(SA_SUBST_GLOBAL, true, lds);
  }
#undef SA_WG_LAUNCH
  return hipGetLastError();
}
}  // namespace sa

bool sa_wgstream_kernel_emits_candidates(const SaFillParams &p, uint32_t max_len_a) {
  return p.cand_count && p.cand_box && p.cand_rows && p.cand_rows_off && p.cand_min && (p.flags & SA_F_IS_SW) &&
         sa_wgstream_kernel_applicable(p, max_len_a);
}

bool sa_wgstream_kernel_reports_best(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b) {
  return p.best_score && p.best_index && (p.flags & SA_F_IS_SW) && sa_wgstream_kernel_applicable(p, max_len_a) &&
         max_len_b < (1u << 21) && max_len_a + 1 <= (1u << 11);   // (column << 21 | row) must fit 32 bits
}

hipError_t sa_launch_fill_wgstream(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  // columns per lane from the batch's longest row: idle lanes still cost a ring slot and an issue slot
  const uint32_t W = max_len_a + 1;
  if (W <= 3 * sa::kWave * 4) return sa::launch_wg<3, 4>(p, max_len_a, stream);
  if (W <= 4 * sa::kWave * 4) return sa::launch_wg<4, 4>(p, max_len_a, stream);
  if (W <= 5 * sa::kWave * 4) return sa::launch_wg<5, 4>(p, max_len_a, stream);
  if (W <= 6 * sa::kWave * 4) return sa::launch_wg<6, 4>(p, max_len_a, stream);
  if (W <= 8 * sa::kWave * 4) return sa::launch_wg<8, 4>(p, max_len_a, stream);
  if (W <= 5 * sa::kWave * 8) return sa::launch_wg<5, 8>(p, max_len_a, stream);
  if (W <= 6 * sa::kWave * 8) return sa::launch_wg<6, 8>(p, max_len_a, stream);
  return sa::launch_wg<8, 8>(p, max_len_a, stream);
}
