// sa_fill_stream.hpp -- row-sweep fill writing through an LDS ring (kernel template; instantiated by
// sa_fill_stream.hip [plain / best-cell] and sa_fill_stream_cand.hip [candidate emission])
//
// row-sweep fill writing through an LDS ring: every global
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
//
// Tuning record (C2, in-process A/B, round 1: profiles/r01_variants_stream.txt; today: tools/ab_option.py):
//   flush unit 1 KiB vs 2 KiB, 1/2/4/8 pairs per workgroup (4 and 8 best), 8-24
//   resident waves per CU, cache-policy bits (plain / nt / sc1 / sc0 sc1, +-2 %;
//   nt kept), removing the arithmetic or the LDS ring altogether: none moves the
//   kernel by more than a few percent.  A store-only kernel with the same write
//   pattern (tools/probes/scatter_probe.hip) runs 0.445 ms vs 0.466 ms for this
//   kernel on a fast box: the kernel is within 5 % of what its own write pattern
//   can do; what is left is the gap between that pattern and a linear memset.
#pragma once
#include "sa_rowsweep.hpp"

namespace sa {

constexpr int kKiBInts = 256;     // one store instruction: 64 lanes x dwordx4 = 1 KiB

// FB = flush unit in ints (a multiple of 256): FB/256 back-to-back 1 KiB stores
// per matrix, FB*4-byte aligned.
//
// Everything that is the same for the 64 lanes (the pair's extent, the stream
// positions, the flush decision, the store base addresses) lives in SGPRs: the
// pair index is made wave-uniform with v_readfirstlane, so the flush test is an
// s_cmp, a full block is three ds_read_b128 + three global_store_dwordx4 with an
// SGPR base (one VALU op for the LDS address), and only the ring write addresses
// are per-lane state.  The kernel is bound by instruction issue as much as by
// HBM (SQ_ACTIVE_INST_ANY ~80 % of the issue slots), so every
// instruction taken out of the row loop is time.
typedef int v4i_a __attribute__((ext_vector_type(4)));   // 16 B aligned

template <int R, int CPL, int FB>
struct StreamOut {
  static constexpr int kBlockInts = FB;
  static constexpr uint32_t kRingBytes = 4u * R;
  char *lds;            // workgroup LDS base
  uint32_t ring_b;      // SGPR: byte offset of this wave's M ring, a multiple of 4R; A at +4R, B at +8R
  uint32_t ring_v;      // the same in a VGPR (v_and_or_b32 takes one SGPR operand only)
  uint32_t wr[CPL];     // VGPR: byte offset (M ring) of my CPL cells in the row being appended
  uint32_t rd_lane;     // VGPR: ring_b + 16*lane
  uint32_t st_lane;     // VGPR: 16*lane
  int32_t *g0[3];       // SGPR: matrix base minus a0 ints: g0 + v is 1 KiB aligned when v % 256 == 0
  uint32_t a0, vend;    // SGPR: virtual range of the pair: [a0, vend)
  uint32_t wv, rv;      // SGPR: virtual write / flush positions (rv % 256 == 0)

  __device__ __forceinline__ void flush_block() {
    const bool inside = (rv >= a0) && (rv + FB <= vend);   // s_cmp
    const uint32_t rd = rd_lane + ((rv & (R - 1)) << 2);
    v4i_a q[3][FB / kKiBInts];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int kb = 0; kb < FB / kKiBInts; ++kb)   // ds_read_b128, immediate offsets
        q[m][kb] = *reinterpret_cast<const v4i_a *>(lds + rd + m * kRingBytes + kb * 1024);
    if (inside) {
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int kb = 0; kb < FB / kKiBInts; ++kb) {
#ifdef SA_EXP_STORE_MASK   // experiment build: which of the three streams are written at all
          if (!((SA_EXP_STORE_MASK >> m) & 1)) continue;
#endif
          // aligned global_store_dwordx4 ... nt (write-once stream, see SA_STORE_VEC), SGPR base + lane offset
          char *blk = reinterpret_cast<char *>(g0[m] + rv + kb * kKiBInts);
          __builtin_nontemporal_store(q[m][kb], reinterpret_cast<v4i_a *>(blk + st_lane));
        }
    } else {   // first / last block of the pair: dword predicates
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int kb = 0; kb < FB / kKiBInts; ++kb) {
          const uint32_t e = rv + kb * kKiBInts + (st_lane >> 2);
          int32_t *dst = g0[m] + e;
          if (e + 0 >= a0 && e + 0 < vend) dst[0] = q[m][kb].x;
          if (e + 1 >= a0 && e + 1 < vend) dst[1] = q[m][kb].y;
          if (e + 2 >= a0 && e + 2 < vend) dst[2] = q[m][kb].z;
          if (e + 3 >= a0 && e + 3 < vend) dst[3] = q[m][kb].w;
        }
    }
    rv += FB;
  }

#ifdef SA_EXP_DEPHASE
  // Experiment build (profiles/r06/r06_dephase.txt): the three streams DE-PHASED inside the kernel.  A wave
  // flushes block k of match_scores, block k - 1 of gap_a_scores and block k - 2 of gap_b_scores at a time, so that the three arenas
  // are never written at the same relative offset at the same moment -- in case what makes three plainly allocated arenas
  // disturb each other (0.65 of the HBM peak against 0.84 placed) is a bank or channel all three hit at equal offsets.
  uint32_t rvm[3] = {0, 0, 0};
  __device__ __forceinline__ void flush_one(int m) {
    const uint32_t r = rvm[m];
    const bool inside = (r >= a0) && (r + FB <= vend);
    const v4i_a q = *reinterpret_cast<const v4i_a *>(lds + rd_lane + ((r & (R - 1)) << 2) + m * kRingBytes);
    if (inside) {
      __builtin_nontemporal_store(q, reinterpret_cast<v4i_a *>(reinterpret_cast<char *>(g0[m] + r) + st_lane));
    } else {
      const uint32_t e = r + (st_lane >> 2);
      int32_t *dst = g0[m] + e;
      if (e + 0 >= a0 && e + 0 < vend) dst[0] = q.x;
      if (e + 1 >= a0 && e + 1 < vend) dst[1] = q.y;
      if (e + 2 >= a0 && e + 2 < vend) dst[2] = q.z;
      if (e + 3 >= a0 && e + 3 < vend) dst[3] = q.w;
    }
    rvm[m] = r + FB;
  }
#endif
  __device__ __forceinline__ void start(int lane) {
    rd_lane = ring_b + 16u * lane;
    st_lane = 16u * lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(ring_v) : "s"(ring_b));
#pragma unroll
    for (int c = 0; c < CPL; ++c) wr[c] = (((wv + lane * CPL + c) & (R - 1)) << 2) | ring_b;
  }

  // append one row: lane holds CPL consecutive cells starting at row position
  // lane*CPL.  ALL 64 lanes write, also those past the row's end: their cells land
  // at ring positions >= wv+W, which are not valid data yet (never flushed before
  // the next row overwrites them) and cannot reach back to unflushed cells because
  // 255 + 64*CPL <= R.  No per-lane predicate, no branch; a write address advances
  // by W ints per row and wraps inside the 4R-byte aligned ring (v_add + v_and_or).
  __device__ __forceinline__ void append_row(uint32_t W, const int (&mv)[CPL], const int (&av)[CPL],
                                             const int (&bv)[CPL]) {
#ifdef SA_EXP_DEPHASE
    static_assert(FB == kKiBInts && 3 * FB - 1 + kWave * CPL <= R, "de-phased streams: the ring holds two more blocks");
#else
    static_assert(FB - 1 + kWave * CPL <= R, "ring too small for unpredicated appends");
#endif
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      char *cell = lds + wr[c];
      *reinterpret_cast<int32_t *>(cell) = mv[c];
      *reinterpret_cast<int32_t *>(cell + kRingBytes) = av[c];
      *reinterpret_cast<int32_t *>(cell + 2 * kRingBytes) = bv[c];
      // wr = ((wr + 4W) & (4R-1)) | ring_b; spelled out, or the compiler splits the OR off
      // into a third op per cell
      const uint32_t t = wr[c] + 4u * W;
      asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(wr[c]) : "v"(t), "s"(kRingBytes - 1), "v"(ring_v));
    }
    wv += W;
    // reads below see the writes above: one wave, LDS ops execute in order; the
    // fence only stops the compiler from reordering them
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#ifdef SA_EXP_DEPHASE
    while (wv - rvm[0] >= (uint32_t)FB) flush_one(0);
    while (wv - rvm[1] >= 2u * FB) flush_one(1);
    while (wv - rvm[2] >= 3u * FB) flush_one(2);
#else
    while (wv - rv >= (uint32_t)FB) flush_block();
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }

  __device__ __forceinline__ void finish() {
#ifdef SA_EXP_DEPHASE
    for (int m = 0; m < 3; ++m) while (rvm[m] < wv) flush_one(m);
#else
    while (rv < wv) flush_block();
#endif
  }
};

// LDS: the rings first (ring bases must be 4R-byte aligned), the substitution
// table behind them
// BEST (SW, p.best_score != nullptr): the kernel also reports the pair's best
// match_scores cell in the reference's hit order (smith_waterman.c:81-85: score
// desc, column asc, then index = row asc) -- 3 VALU ops per cell on data that is in
// registers anyway, instead of sa_reduce.hip's second pass over the matrix.
constexpr int kBestRowBits = 21;   // packed tie-break: column (11 bits) | row (21 bits)

// What else the kernel reports about match_scores while the values are in registers (SW only):
//   SA_STREAM_BEST  the best cell per pair (above) -- seqalign_sw_batch with max_hits = 1
//   SA_STREAM_CAND  whether any cell has score >= cand_min[pair] (cand_count: 0 / 1) and the cells' bounding box -- the candidate scan of
//                   smith_waterman.c:152-156 for the multi-hit path, which then sweeps only the box's rows
//                   (sa_sw_sweep.hip), and within a row only the columns between its lowest and highest candidate
//                   (cand_rows: two uint32 per row, 64 rows per store).  Per row: one ballot per column slot; rows
//                   without a candidate cost the ballots and one scalar branch.
enum { SA_STREAM_PLAIN = 0, SA_STREAM_BEST = 1, SA_STREAM_CAND = 2 };

template <int CPL, int SUBST, bool GENERAL, int R, int FB, int MODE>
__global__ void __launch_bounds__(kWave * 8)
fill_stream_kernel(const SaFillParams p, const uint32_t table_ints) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t waves = blockDim.x >> 6;
  const int32_t *table = p.table;
  if constexpr (SUBST == SA_SUBST_LDS) {
    int32_t *tbl = lds + waves * (3 * R);
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) tbl[k] = p.table[k];
    __syncthreads();
    table = tbl;
  }

  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: so is all per-pair state below
  const uint32_t pair = blockIdx.x * waves + wave;
  if (pair >= p.n_pairs) return;   // wave-uniform, after the only barrier
#ifdef SA_EXP_TRACE   // experiment build (make exp): where and when did this wave run?
  const uint64_t trace_t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair];
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const uint32_t W = la + 1;

  const SweepConsts k(p, table);
  const Border bd{p.floor, p.gap_open, p.ext, (p.flags & SA_F_IS_SW) != 0,
                  (p.flags & SA_F_NO_START_GAP) != 0};

  StreamOut<R, CPL, FB> out;
  out.lds = reinterpret_cast<char *>(lds);
  out.ring_b = wave * (3u * 4u * R);
  // the three arenas are congruent mod 4 KiB (checked on the host)
  out.a0 = (uint32_t)(((uintptr_t)(p.M + mo) >> 2) & (FB - 1));
  out.g0[0] = p.M + mo - out.a0;
  out.g0[1] = p.A + mo - out.a0;
  out.g0[2] = p.B + mo - out.a0;
  out.vend = out.a0 + W * (lb + 1);
  out.wv = out.a0;
  out.rv = 0;
  out.start(lane);

  const uint32_t col0 = (uint32_t)(lane * CPL) - 1u;                 // matrix column lane*CPL + c
  const int ncol = max(0, min(CPL, (int)W - lane * CPL));

  RowSweep<CPL, SUBST, GENERAL, true> sw;
  sw.start_strip(p, k, bd, sa_, la, 0, col0, lane);
  __builtin_amdgcn_s_waitcnt(kWaitVm0);   // seq_a codes landed (see RowFeed::load)

  {  // row 0 (reference alignment.c:46-69): (0,0) = 0; M = A = floor, B = edge
    int mv[CPL], av[CPL], bv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t ci = lane * CPL + c;
      mv[c] = av[c] = (ci == 0) ? 0 : k.floor_;
      bv[c] = (ci == 0) ? 0 : bd.edge_gap(ci);
    }
    out.append_row(W, mv, av, bv);
  }

  constexpr bool BEST = (MODE == SA_STREAM_BEST);
  constexpr bool CAND = (MODE == SA_STREAM_CAND);
  int best_s[BEST ? CPL : 1], best_r[BEST ? CPL : 1];   // per column: highest score, first row that reached it
  // candidate emission: wave-uniform running count and bounding box of the pair's candidates
  uint32_t cand_n = 0, box_rmin = 0xffffffffu, box_rmax = 0, box_cmin = 0xffffffffu, box_cmax = 0;
  int cand_thr = INT32_MAX;
  uint32_t rr_lo = 0xffffffffu, rr_hi = 0;
  uint32_t *cand_rows = nullptr;   // [len_b + 1][2]: lowest / highest candidate column of every row (lo > hi: none)
  if constexpr (CAND) {
    cand_thr = max(p.cand_min[pair], 1);   // candidates need match_scores > 0 (smith_waterman.c:154)
    cand_rows = p.cand_rows + 2ull * p.cand_rows_off[pair];
    if (lane == 0) *reinterpret_cast<uint2 *>(cand_rows) = make_uint2(0xffffffffu, 0u);   // row 0: borders only
  }
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
      __builtin_amdgcn_s_waitcnt(kWaitVm0);   // see RowFeed::load
    }
    int mv[CPL], av[CPL], bv[CPL];
    sw.row(k, j, lb, la, W, lane, col0, ncol, read_lane(chunk_code, q), 0, 0, mv, av, bv, bd.edge_gap(j));
    out.append_row(W, mv, av, bv);
    if constexpr (BEST) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const bool up = mv[c] > best_s[c];   // strict: the first (lowest) row keeps a tie
        best_s[c] = up ? mv[c] : best_s[c];
        best_r[c] = up ? (int)j : best_r[c];
      }
    }
    if constexpr (CAND) {
      unsigned long long bal[CPL], any = 0;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        bal[c] = __ballot(c < ncol && mv[c] >= cand_thr);
        any |= bal[c];
      }
      uint32_t row_lo = 0xffffffffu, row_hi = 0;   // this row's candidates: lowest / highest column (wave-uniform)
      if (any) {   // wave-uniform.  At lane granularity (a lane's CPL columns): the sweep only needs bounds
        row_lo = (uint32_t)__builtin_ctzll(any) * CPL;
        row_hi = (uint32_t)(63 - __builtin_clzll(any)) * CPL + (CPL - 1);
        cand_n = 1;
        box_cmin = min(box_cmin, row_lo);
        box_cmax = max(box_cmax, row_hi);
        box_rmin = min(box_rmin, j);
        box_rmax = j;
      }
      // per-row ranges: lane q keeps row j's, 64 rows leave as one coalesced store
      if (lane == q) { rr_lo = row_lo; rr_hi = row_hi; }
      if (q == kWave - 1 || j == lb) {
        if (lane <= q) *reinterpret_cast<uint2 *>(cand_rows + 2ull * (j - q + lane)) = make_uint2(rr_lo, min(rr_hi, W - 1));
      }
    }
  }
  out.finish();

  if constexpr (CAND) {
    if (lane == 0) {
      p.cand_count[pair] = cand_n;
      uint32_t *box = p.cand_box + 4ull * pair;
      box[0] = box_rmin; box[1] = box_rmax; box[2] = box_cmin; box[3] = min(box_cmax, W - 1);
    }
  }

  if constexpr (BEST) {
    // lane: lowest column wins a tie (c ascending, strict >); lanes past the row's end hold garbage
    int b = 0;
    uint32_t tie = 0;   // (column << kBestRowBits) | row of the best cell
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      if (c < ncol && best_s[c] > b) { b = best_s[c]; tie = ((uint32_t)(lane * CPL + c) << kBestRowBits) | (uint32_t)best_r[c]; }
    }
    // wave: max score, then min (column, row)
    unsigned long long key = ((unsigned long long)(uint32_t)b << 32) | (uint32_t)~tie;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o);
      key = other > key ? other : key;
    }
    if (lane == 0) {
      const uint32_t t = ~(uint32_t)key, col = t >> kBestRowBits, row = t & ((1u << kBestRowBits) - 1);
      const int score = (int)(key >> 32);
      p.best_score[pair] = score;
      p.best_index[pair] = score > 0 ? (uint64_t)row * W + col : 0;
    }
  }

#ifdef SA_EXP_TRACE
  // status <- xcc(4) | HW_ID[15:0] (wave, simd, pipe, cu, sh, se) | t0 (22 bits) | t1 (22 bits), 10 ns ticks
  const uint64_t trace_t1 = __builtin_amdgcn_s_memrealtime();
  const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));
  if (lane == 0)
    p.status[pair] = ((uint64_t)(xcc & 15) << 60) | ((uint64_t)(hw & 0xffff) << 44) |
                     ((trace_t0 & 0x3fffff) << 22) | (trace_t1 & 0x3fffff);
  (void)sw;
#else
  const unsigned long long err = sw.reduce_err();
  if (lane == 0) p.status[pair] = err;
#endif
}


template <int CPL, int R, int FB, int MODE, int WPB = kWavesPerBlock>
static hipError_t launch_cpl_mode(const SaFillParams &p, hipStream_t stream) {
  const bool general = needs_general(p);
  int wpb = WPB;   // pairs per workgroup; option "wpb" in {1,2,4,8} (tuning experiments)
  if (p.tune_wpb == 1 || p.tune_wpb == 2 || p.tune_wpb == 4 || p.tune_wpb == 8) wpb = (int)p.tune_wpb;
  const dim3 grid((p.n_pairs + wpb - 1) / wpb), block(kWave * wpb);
  size_t rings = (size_t)wpb * 3 * R * sizeof(int32_t);
  rings += p.tune_lds_pad;   // option "lds_pad": occupancy experiments
  if (p.K <= 1) {
    if (general) hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_SIMPLE, true, R, FB, MODE>), grid, block, rings, stream, p, 0u);
    else hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_SIMPLE, false, R, FB, MODE>), grid, block, rings, stream, p, 0u);
  } else if (p.K <= SA_LDS_TABLE_MAX_K) {
    const uint32_t tints = (p.K * p.K + 3u) & ~3u;
    const size_t lds = rings + tints * sizeof(int32_t);
    if (general) hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_LDS, true, R, FB, MODE>), grid, block, lds, stream, p, tints);
    else hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_LDS, false, R, FB, MODE>), grid, block, lds, stream, p, tints);
  } else {
    hipLaunchKernelGGL((fill_stream_kernel<CPL, SA_SUBST_GLOBAL, true, R, FB, MODE>), grid, block, rings, stream, p, 0u);
  }
  return hipGetLastError();
}

// columns per lane: the border column is a column too
// (a 2 KiB flush unit -- <CPL, 1024, 512> -- was measured in round 1: no difference, removed)
template <int MODE>
static hipError_t launch_stream_mode(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream) {
  const uint32_t need = columns_per_lane(max_len_a + 1, p.tune_cpl);
#ifdef SA_EXP_DEPHASE   // (two more blocks in the rings: 1 024 ints up to 3 columns per lane, 2 048 beyond)
  if (need <= 1) return launch_cpl_mode<1, 1024, 256, MODE>(p, stream);
  if (need <= 2) return launch_cpl_mode<2, 1024, 256, MODE>(p, stream);
  if (need <= 3) return launch_cpl_mode<3, 1024, 256, MODE>(p, stream);
  if (need <= 4) return launch_cpl_mode<4, 2048, 256, MODE, 2>(p, stream);
  if (need <= 5) return launch_cpl_mode<5, 2048, 256, MODE, 2>(p, stream);
  if (need <= 6) return launch_cpl_mode<6, 2048, 256, MODE, 2>(p, stream);
  if (need <= 8) return launch_cpl_mode<8, 2048, 256, MODE, 2>(p, stream);
  return hipErrorInvalidValue;
#else
  if (need <= 1) return launch_cpl_mode<1, 512, 256, MODE>(p, stream);
  if (need <= 2) return launch_cpl_mode<2, 512, 256, MODE>(p, stream);
  if (need <= 3) return launch_cpl_mode<3, 512, 256, MODE>(p, stream);
  if (need <= 4) return launch_cpl_mode<4, 512, 256, MODE>(p, stream);
  if (need <= 5) return launch_cpl_mode<5, 1024, 256, MODE>(p, stream);
  if (need <= 6) return launch_cpl_mode<6, 1024, 256, MODE>(p, stream);
  if (need <= 8) return launch_cpl_mode<8, 1024, 256, MODE>(p, stream);
  // 513..1023 columns: 12 / 16 columns per lane, 24 KiB of rings per wave -> 2 pairs per workgroup
  if (need <= 12) return launch_cpl_mode<12, 2048, 256, MODE, 2>(p, stream);
  return launch_cpl_mode<16, 2048, 256, MODE, 2>(p, stream);
#endif
}

}  // namespace sa
