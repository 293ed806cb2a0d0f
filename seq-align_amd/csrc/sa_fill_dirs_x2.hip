// sa_fill_dirs_x2.hip -- the directions-only fills (sa_fill_dirs.hip) with TWO pairs per wave in packed int16.
//
// Why: with one byte per cell instead of twelve, the directions-only fills are no longer bound by HBM but by VALU
// issue -- ~135 vector instructions per 151-cell row (profiles/r03: C2 0.318 ms where the bytes alone would take
// 0.03 ms).  Every one of those instructions is lane-wise 32-bit arithmetic on scores that, for read-sized pairs,
// fit 16 bits.  CDNA's packed-math VALU (v_pk_add_i16 / v_pk_max_i16 / v_pk_sub_i16 with the `clamp` modifier =
// saturation / v_pk_ashrrev_i16) works on two int16 per lane per instruction, and the bitwise ops are packed by
// nature.  So one wave takes TWO pairs of the same shape: pair 2u in the low halves, pair 2u + 1 in the high halves
// of every register -- no instruction ever mixes the halves, the DPP lane shifts and the prefix-max scan move both
// at once, and the instruction count per row serves two pairs.
//
// Same recurrence and the same direction byte as sa_fill_dirs.hip (reference src/alignment.c:89-167 for the scores,
// :311-327 for the decisions; sa_rowsweep.hpp for the de-trended prefix max).  What differs:
//   * int16 with saturating adds.  The NW floor (alignment.c:41) becomes -32768 = the bottom of the type: `max(x,
//     floor)` is the identity and floor + penalty saturates back to the floor, which is what the reference's clamp
//     does.  Every comparison that decides a direction is then between the same numbers as in 32 bits as long as
//     no real score leaves the type: the launcher admits a batch only if
//     (len_a + len_b + 2) * max|penalty| + (len_a + 1) * |gap_extend| <= 30000  (sa_x2_scores_fit; the NW fills, whose values are
//     de-trended by (column + row) x gap_extend: (len_a + len_b + 2) * (max|penalty| + |gap_extend|) <= 30000).
//   * comparisons are sign bits: x < y  <=>  (x -sat y) >> 15 = 0xFFFF per half; selections are v_bfi_b32.
//   * both pairs' rows have the same length, so ONE set of stream positions serves both rings; the caller lays the
//     direction bytes out with every pair starting on a 256-byte boundary (SaFillParams::uniform_stride), so every
//     global store is a whole aligned 256 B block and the last one may run into the pair's own padding.
// Domain: what sa_fill_dirs.hip takes, AND the pairs that share a wave have the same len_a and len_b (a uniform launch, or a
// pair list that the host has paired up by shape), AND scores inside the bound above (match / mismatch scoring, or a
// substitution table of up to SA_LDS_TABLE_MAX_K classes held in LDS as int16).  Anything else takes the one-pair-per-wave
// kernels.
// Round 4: the wave functions take LANES = 64 (two pairs per wave, as above) or 32 -- FOUR pairs per wave, a couple of pairs in
// the halves of lanes 0-31 and another in lanes 32-63 (a 151-column row: 160 cell slots instead of 192, and the per-row work
// that does not depend on the width serves four rows; NW and the SW best-hit fill: nw_dirs_x2_wave / sw_best_x2_wave) -- and a
// launch whose last round of four-per-wave waves would be less than half full runs that round two per wave in the same grid
// (fill_nw_dirs_x4x2_kernel).  DESIGN.md 3.5f; profiles/r04/r04_quad_fills.txt.
// Round 6: the NW and best-hit wave functions take LOCAL -- the direction byte as the cell's own comparisons, one raw bit each
// (sa_kernels.h: SA_LD_*; put_ge below), for chunks whose walks are tile walks -- and the best-hit fill has the NW fills' mixed
// grid too (fill_sw_best_x4x2_kernel).  DESIGN.md 3.5g; profiles/r06/r06_local_dirs.txt.
#include <algorithm>

#include "sa_rowsweep.hpp"
#include "sa_fill_nw_dirs_x1.hpp"

// Round 6: the rows' LDS traffic as a software pipeline (1 = on; `make exp EXPFLAGS=-DSA_X2_LDS_PIPE=0` builds the round-5 form for a
// same-box A/B).  What the counters said about BASELINE configs[3] (4 000 pairs = 2 000 waves, two per SIMD;
// profiles/r05/r05b_sweep_pmc_C4_after.json, fill_dirs_x2_kernel<5,1,1024>): a wave spends 26 % of its cycles parked at s_waitcnt
// (SQ_WAIT_ANY / SQ_WAVE_CYCLES) -- three LDS round trips per row, each issued and then waited for on the spot: the row's
// profile words (10 ds_read_b32), the next row's profile build (ds_read_u16 -> ds_write_b32), the rings' flush (ds_read ->
// global_store) -- and with two waves on a SIMD nobody covers them (0.55 of VALU issue; 0.68 with eight waves at 16 000 pairs).
// Now every LDS read is issued a row ahead of its use: the next row's profile words while this row is computed (the row after
// that's table entries too), and a ring block is read at the end of one row and stored to HBM at the end of the next.
#ifndef SA_X2_LDS_PIPE
#define SA_X2_LDS_PIPE 1
#endif
#ifndef SA_X2_FLUSH_DEFER
#define SA_X2_FLUSH_DEFER SA_X2_LDS_PIPE
#endif
// Round 6: the NW fills keep every value V of cell (g, j) as V' = V - (g + j) gap_extend.  gap_a's recurrence max(Y + open1, A + ext) becomes
// max(Y' + gap_open, A'), gap_b's scan -- already de-trended along the row -- needs no re-trend, the diagonal step's "- 2 ext" goes into the
// substitution scores (table and constants, once per launch), every comparison is between values of ONE cell or of cells whose g + j
// differ by a constant that goes into the constant: two packed adds per cell less (C2's fill 163 -> 153 us; `make exp
// EXPFLAGS=-DSA_NW_DETREND=0` builds the older arithmetic).  The end cell's score gets its (len_a + len_b) ext back; the admission
// bound grows by the de-trend's range (sa_domain_nw_x2_scores_fit).
#ifndef SA_NW_DETREND
#define SA_NW_DETREND 1
#endif
// from how many pairs the best-hit fill goes four per wave (sa_launch_fill_sw_best_x2; `make exp EXPFLAGS=-DSA_BEST_X4_MIN=8192u` for an A/B)
#ifndef SA_BEST_X4_MIN
#define SA_BEST_X4_MIN 4097u
#endif
namespace sa {
// bytes of LDS per pair of the NW / best-hit fills: the ring of the row-major form, or one block row of LANES x CPL columns
constexpr int x2_ring(int lanes, int cpl, int ring) { return (SA_DIRS_BLOCKED != 0 && lanes * cpl <= 512 && lanes * cpl * 8 > ring) ? lanes * cpl * 8 : ring; }
constexpr bool kLdsPipe = SA_X2_LDS_PIPE != 0;        // the row profile (table scorings) a row ahead
constexpr bool kFlushDefer = SA_X2_FLUSH_DEFER != 0;  // a ring block read at the end of one row, stored at the end of the next
constexpr bool kNwDetrend = SA_NW_DETREND != 0;
constexpr bool kDirsBlocked = SA_DIRS_BLOCKED != 0;    // the NW / best-hit fills write the direction bytes in 8 x 16 blocks (sa_kernels.h)

typedef short pk16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_bits(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 pk_from(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ pk16 pk_splat(int v) { return pk16{(short)v, (short)v}; }
__device__ __forceinline__ pk16 pk_adds(pk16 a, pk16 b) { return __builtin_elementwise_add_sat(a, b); }   // v_pk_add_i16 clamp
__device__ __forceinline__ pk16 pk_subs(pk16 a, pk16 b) { return __builtin_elementwise_sub_sat(a, b); }   // v_pk_sub_i16 clamp
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }        // v_pk_max_i16
// 0xFFFF in every half where x < y
__device__ __forceinline__ uint32_t pk_lt(pk16 x, pk16 y) { return pk_bits(pk_subs(x, y) >> 15); }
// mask ? a : b, bit by bit, and a three-way OR: gfx950's v_bitop3_b32 issues in 2 cycles where v_bfi_b32 / v_or3_b32 /
// v_and_or_b32 take 4 (tools/probes/valu_rate_probe.hip), and the compiler picks the latter when left to itself
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xCA); }
__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE); }
// 0xFFFF in every half where x > 0, for x >= 0
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b);
__device__ __forceinline__ uint32_t pos_mask(pk16 x) { return pk_bits(pk_splat(0) - __builtin_bit_cast(pk16, pk_min_u16(pk_bits(x), 0x00010001u))); }
__device__ __forceinline__ pk16 pk_shr1(pk16 src, pk16 lane0) {
  return pk_from((uint32_t)wave_shr1((int)pk_bits(src), (int)pk_bits(lane0)));
}
// DPP move with bound_ctrl: lanes without a source read 0 -- no `old` register to set up
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ pk16 pk_shr1_zero(pk16 src) { return pk_from(dpp_mov0<0x138>(pk_bits(src))); }   // lane 0 gets 0
// packed ops the compiler does not pick by itself
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ pk16 pk_mad(uint32_t a, pk16 b, pk16 c) {   // a * b + c per half (wrapping)
  uint32_t r;
  asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(pk_bits(b)), "v"(pk_bits(c)));
  return pk_from(r);
}
// Exclusive max scan over the 64 lanes, both halves at once: lane l gets the max of lanes < l, lane 0 the bottom of
// the type.  VOP3P has no DPP form, so each step is a DPP move and a max; done on BIASED values (x ^ 0x8000: signed
// order = unsigned order) so that 0 -- what bound_ctrl gives a lane without a source -- is the identity, and with
// full row masks: a max may take the same element twice, so rows 2 and 3 may also see row 1 through row_bcast:15.
__device__ __forceinline__ pk16 pk_wave_scan_max_excl(pk16 v) {
  uint32_t u = pk_bits(v) ^ 0x80008000u;
  u = pk_max_u16(u, dpp_mov0<0x111>(u));   // row_shr:1
  u = pk_max_u16(u, dpp_mov0<0x112>(u));   // row_shr:2
  u = pk_max_u16(u, dpp_mov0<0x114>(u));   // row_shr:4
  u = pk_max_u16(u, dpp_mov0<0x118>(u));   // row_shr:8
  u = pk_max_u16(u, dpp_mov0<0x142>(u));   // row_bcast:15
  u = pk_max_u16(u, dpp_mov0<0x143>(u));   // row_bcast:31
  return pk_from(dpp_mov0<0x138>(u) ^ 0x80008000u);   // wave_shr:1
}

// The same over each HALF of the wave separately (lanes 0-31, lanes 32-63: four pairs per wave, see LANES below): one step
// fewer -- row_bcast:15 into rows 1 and 3 only (row_mask 0xA; the other rows keep the 0 = identity), no row_bcast:31 -- and
// lane 32, which the last shift hands lane 31's value, gets the identity as lane 0 does.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov0_rows(uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ pk16 pk_half_scan_max_excl(pk16 v, bool first_lane) {
  uint32_t u = pk_bits(v) ^ 0x80008000u;
  u = pk_max_u16(u, dpp_mov0<0x111>(u));   // row_shr:1
  u = pk_max_u16(u, dpp_mov0<0x112>(u));   // row_shr:2
  u = pk_max_u16(u, dpp_mov0<0x114>(u));   // row_shr:4
  u = pk_max_u16(u, dpp_mov0<0x118>(u));   // row_shr:8
  u = pk_max_u16(u, dpp_mov0_rows<0x142, 0xA>(u));   // row_bcast:15 -> rows 1, 3
  const uint32_t r = dpp_mov0<0x138>(u);   // wave_shr:1
  return pk_from((first_lane ? 0u : r) ^ 0x80008000u);
}
template <int LANES>
__device__ __forceinline__ pk16 pk_scan_max_excl(pk16 v, bool first_lane) {
  if constexpr (LANES == 64) return pk_wave_scan_max_excl(v);
  else return pk_half_scan_max_excl(v, first_lane);
}
// lane l <- lane l - 1, the first lane of a span of LANES lanes <- 0
template <int LANES>
__device__ __forceinline__ pk16 pk_shr1_zero_in(pk16 src, bool first_lane) {
  const uint32_t r = dpp_mov0<0x138>(pk_bits(src));
  if constexpr (LANES == 64) return pk_from(r);
  else return pk_from(first_lane ? 0u : r);
}

// lane l <- lane l - 1, the first lane of a span <- `first`
template <int LANES>
__device__ __forceinline__ pk16 pk_shr1_in(pk16 src, pk16 first, bool first_lane) {
  if constexpr (LANES == 64) return pk_shr1(src, first);
  else {
    // (the shift first, with every lane active -- a lane that is masked out is no source for its neighbour -- then the choice)
    const uint32_t r = dpp_mov0<0x138>(pk_bits(src));
    return pk_from(first_lane ? pk_bits(first) : r);
  }
}
// this row's characters of seq_b (code of the low pair | code of the high pair << 16) out of the chunk a span's lanes hold
template <int LANES>
__device__ __forceinline__ uint32_t row_codes(uint32_t chunk_code, int q, uint32_t span) {
  if constexpr (LANES == 64) {
    return (uint32_t)read_lane((int)chunk_code, q);
  } else {
    const uint32_t r0 = (uint32_t)read_lane((int)chunk_code, q), r1 = (uint32_t)read_lane((int)chunk_code, q + 32);
    return span ? r1 : r0;
  }
}

constexpr uint32_t kBoth = 0x00010001u;   // a 1 in each half

// The LOCAL form of the direction byte (sa_kernels.h: SA_LD_*): a decision is the sign of a saturating difference d = x -sat y
// (set where x < y), and its bit of the byte -- set where x >= y, which among the recurrence's own operands means x == y -- is that
// sign shifted into place: one v_lshrrev_b32 + one v_bitop3_b32 (acc | (~shifted & bit)), both halves at once, where the older form
// smears the sign over its half (v_pk_ashrrev_i16), selects a two-bit code with it (v_bfi) and merges the codes.
template <int K>
__device__ __forceinline__ uint32_t put_ge(uint32_t acc, pk16 d) {
  static_assert(K >= 0 && K <= 7, "a bit of the direction byte");
  return __builtin_amdgcn_bitop3_b32(acc, pk_bits(d) >> (15 - K), (1u << K) * kBoth, 0xF2);   // a | (~b & c)
}

// The substitution score of my columns against this row's character, both pairs at once.
//   SA_SUBST_SIMPLE (K <= 1): equal characters -> gen_eq, else gen_ne = gen_eq + min(fa ^ fb, 1) * (gen_ne - gen_eq).
//   SA_SUBST_LDS: the K x K table (int16 in LDS, behind the rings), row = class of the seq_a character; the two pairs'
//   scores are two 16-bit LDS reads into the two halves of one register.  Class 0 x class 0 holds the "equal" score of
//   characters outside the table: different characters get gen_ne (subst_score in sa_fill_common.hpp).
//   PROFILE (round 5; SA_SUBST_LDS, two pairs per wave): the row's scores come from a ROW PROFILE instead of the table itself.
//   The two table reads per cell went to addresses all over the K x K table -- 64 lanes x 2 B at random: SQ_LDS_BANK_CONFLICT was
//   39 % of the LDS cycles of fill_dirs_x2_kernel<5,1,1024> (profiles/r05/r05a_sweep_pmc_C4_before.json), each with its own
//   address add, then a pack and the class-0 fix-up: the BLOSUM62 fills ran at 0.55-0.59 of the VALU issue peak where the DNA
//   instantiations of the same code reach 0.66-0.71.  Now lane k < K builds, once per row,
//       prof[k] = table[k][class of pair 0's character] | table[k][class of pair 1's character] << 16
//   (the striped-profile idea, one row at a time: K words per wave, double-buffered -- the next row's profile is written while
//   this row is worked on), and a cell reads prof[class of its seq_a character]: pair 0's word for the low half, pair 1's for the
//   high half, one bit-select.  Addresses are constant per column (no adds), K <= 64 consecutive words never collide in a bank
//   (equal addresses broadcast), and the class-0 fix-up (characters outside the table) runs only on rows whose seq_b character
//   is outside it (wave-uniform test).  Reference: scoring_lookup, src/alignment_scoring.c:133-182; table :268-292.
constexpr uint32_t kProfBytes = 512;      // per wave: two buffers of 64 words
template <int SUBST, int CPL, bool PROFILE = false>
struct SubstX2 {
  static_assert(!PROFILE || SUBST == SA_SUBST_LDS, "a row profile is a view of the LDS table");
  uint32_t fa[CPL];                                   // folded characters of seq_a, one pair per half
  uint32_t ar0[SUBST == SA_SUBST_LDS ? CPL : 1];      // LDS byte address of the table row of pair 0's / pair 1's character
  uint32_t ar1[SUBST == SA_SUBST_LDS ? CPL : 1];      //   (PROFILE: of the profile word of its class, within a buffer)
  uint32_t a0[SUBST == SA_SUBST_LDS ? CPL : 1];       // 0xFFFF per half where the seq_a character is of class 0
  pk16 s_eq, s_delta, s_ne;
  uint32_t fb = 0, kb0 = 0, kb1 = 0, b0 = 0;          // this row (wave-uniform): characters, table columns * 2, class-0 mask
  uint32_t prof = 0, bld_src = 0, bld_dst = 0;        // PROFILE: the wave's two buffers; this lane's table row / profile word
  pk16 sc[PROFILE ? CPL : 1];                         // PROFILE: this row's scores of my columns
  // PROFILE, pipelined (kLdsPipe): what is on its way out of LDS for the rows to come -- raw, not looked at until merge_next()
  uint32_t scn0[PROFILE ? CPL : 1], scn1[PROFILE ? CPL : 1];   // the next row's profile words of my columns (pair 0's, pair 1's)
  uint32_t tv0 = 0, tv1 = 0, tvw = 0;                 // the row after next: my table row's two entries on their way; packed
  pk16 scq[PROFILE ? CPL : 1];                        // the next row's scores, merged
  __device__ __forceinline__ void init(const SaFillParams &p, uint32_t tbl_lds = 0, int lane = 0, int shift = 0) {
    s_eq = pk_splat(p.gen_eq + shift); s_ne = pk_splat(p.gen_ne + shift); s_delta = pk_splat(p.gen_ne - p.gen_eq);
    if constexpr (PROFILE) {
      const uint32_t k = min((uint32_t)lane, p.K - 1u);   // (lanes beyond the table repeat its last row: same word, same value)
      prof = tbl_lds + ((p.K * p.K * 2u + 15u) & ~15u) + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * kProfBytes;
      bld_src = tbl_lds + k * p.K * 2u; bld_dst = k * 4u;
    }
  }
  __device__ __forceinline__ void set_column(int c, uint32_t code0, uint32_t code1, uint32_t K, uint32_t tbl_lds) {
    fa[c] = (code0 & 0xffu) | (code1 & 0xffu) << 16;
    if constexpr (SUBST == SA_SUBST_LDS) {
      if constexpr (PROFILE) { ar0[c] = (code0 >> 8) * 4u; ar1[c] = (code1 >> 8) * 4u; }
      else { ar0[c] = tbl_lds + (code0 >> 8) * K * 2u; ar1[c] = tbl_lds + (code1 >> 8) * K * 2u; }
      a0[c] = ((code0 >> 8) ? 0u : 0xffffu) | ((code1 >> 8) ? 0u : 0xffff0000u);
    }
  }
  __device__ __forceinline__ void set_row(uint32_t codes) {   // codes = pair 0's code | pair 1's code << 16 (uniform)
    fb = codes & 0x00ff00ffu;
    if constexpr (SUBST == SA_SUBST_LDS) {
      kb0 = ((codes >> 8) & 0xffu) * 2u; kb1 = (codes >> 24) * 2u;
      b0 = (kb0 ? 0u : 0xffffu) | (kb1 ? 0u : 0xffff0000u);
    }
  }
  // PROFILE: the profile of a row whose characters are `codes` into buffer `buf` (0 / 1) -- every lane, no predicate
  __device__ __forceinline__ void build_profile(uint32_t codes, uint32_t buf) const {
    if constexpr (PROFILE) {
      extern __shared__ __attribute__((aligned(16))) int32_t lds_base[];
      char *l = reinterpret_cast<char *>(lds_base);
      const uint32_t k0 = ((codes >> 8) & 0xffu) * 2u, k1 = (codes >> 24) * 2u;
      const uint32_t v0 = *reinterpret_cast<const unsigned short *>(l + bld_src + k0);
      const uint32_t v1 = *reinterpret_cast<const unsigned short *>(l + bld_src + k1);
      *reinterpret_cast<uint32_t *>(l + prof + buf * (kProfBytes / 2) + bld_dst) = v0 | v1 << 16;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (one wave: program order; the compiler must keep it)
    }
  }
  // PROFILE: this row's scores of my columns out of buffer `buf` (after set_row)
  __device__ __forceinline__ void load_profile(uint32_t buf) {
    if constexpr (PROFILE) {
      extern __shared__ __attribute__((aligned(16))) int32_t lds_base[];
      const char *l = reinterpret_cast<const char *>(lds_base) + prof + buf * (kProfBytes / 2);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(l + ar0[c]), w1 = *reinterpret_cast<const uint32_t *>(l + ar1[c]);
        sc[c] = pk_from(bfi(0x0000ffffu, w0, w1));
      }
      if (b0) {   // (wave-uniform) a seq_b character outside the table: class 0 x class 0 means "equal" only for equal characters
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const uint32_t ne01 = pk_min_u16(fa[c] ^ fb, 0x00010001u);
          const uint32_t differ = pk_bits(pk_splat(0) - pk_from(ne01));
          sc[c] = pk_from(bfi(a0[c] & b0 & differ, pk_bits(s_ne), pk_bits(sc[c])));
        }
      }
    }
  }
  // ---- the pipelined form: issue now, look later
  __device__ __forceinline__ void issue_table_reads(uint32_t codes) {   // tv <- table[my row][class of the characters `codes`]
    if constexpr (PROFILE) {
      extern __shared__ __attribute__((aligned(16))) int32_t lds_base[];
      const char *l = reinterpret_cast<const char *>(lds_base);
      tv0 = *reinterpret_cast<const unsigned short *>(l + bld_src + ((codes >> 8) & 0xffu) * 2u);
      tv1 = *reinterpret_cast<const unsigned short *>(l + bld_src + (codes >> 24) * 2u);
    }
  }
  __device__ __forceinline__ void pack_table_reads() { tvw = tv0 | tv1 << 16; }
  __device__ __forceinline__ void write_profile(uint32_t buf) const {   // my word of buffer `buf` <- tvw
    if constexpr (PROFILE) {
      extern __shared__ __attribute__((aligned(16))) int32_t lds_base[];
      *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds_base) + prof + buf * (kProfBytes / 2) + bld_dst) = tvw;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
  }
  __device__ __forceinline__ void issue_profile_loads(uint32_t buf) {   // scn <- buffer `buf`'s words of my columns' classes
    if constexpr (PROFILE) {
      extern __shared__ __attribute__((aligned(16))) int32_t lds_base[];
      const char *l = reinterpret_cast<const char *>(lds_base) + prof + buf * (kProfBytes / 2);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        scn0[c] = *reinterpret_cast<const uint32_t *>(l + ar0[c]);
        scn1[c] = *reinterpret_cast<const uint32_t *>(l + ar1[c]);
      }
    }
  }
  // scq <- the scores of the row whose characters are `codes` out of the words issue_profile_loads asked for (load_profile's
  // second half: the halves' select, and the class-0 fix-up on rows whose seq_b character is outside the table)
  __device__ __forceinline__ void merge_next(uint32_t codes) {
    if constexpr (PROFILE) {
      const uint32_t fbn = codes & 0x00ff00ffu;
      const uint32_t b0n = (((codes >> 8) & 0xffu) ? 0u : 0xffffu) | ((codes >> 24) ? 0u : 0xffff0000u);
#pragma unroll
      for (int c = 0; c < CPL; ++c) scq[c] = pk_from(bfi(0x0000ffffu, scn0[c], scn1[c]));
      if (b0n) {   // (wave-uniform)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const uint32_t ne01 = pk_min_u16(fa[c] ^ fbn, 0x00010001u);
          const uint32_t differ = pk_bits(pk_splat(0) - pk_from(ne01));
          scq[c] = pk_from(bfi(a0[c] & b0n & differ, pk_bits(s_ne), pk_bits(scq[c])));
        }
      }
    }
  }
  // The row's scores, pipelined: call row_begin() where set_row / build_profile / load_profile stood, row_end() before the row's
  // direction bytes go to the rings.  chunk: the lanes' seq_b codes of this chunk of 64 rows (row j is lane q's), lb: rows.
  //   row_begin(j): this row's scores = what row_end(j - 1) merged (a chunk's first row: built, loaded and merged here, waited
  //                 for on the spot -- one row in 64); then row j + 1's profile word is written (from the table entries read during
  //                 row j - 1), its words of my columns asked for, and row j + 2's table entries asked for;
  //   row_end(j):   those words -- asked for a row's work ago -- become row j + 1's scores, the table entries one packed word.
  // Everything outside the chunk's-first-row block is UNCONDITIONAL: a load under a branch is a value merged at the branch's
  // end, i.e. a register copy there, i.e. a wait for the load right behind its issue (the first version of this had exactly
  // that: s_waitcnt lgkmcnt(0) + v_mov at the join).  What is asked for beyond the last row, or beyond the chunk's 64 rows, is
  // some valid row's words (the lanes' codes are classes of the table whatever row they belong to) and is never looked at: a
  // chunk's first row rebuilds its own.
  __device__ __forceinline__ void row_begin(uint32_t chunk, int q, uint32_t j, uint32_t lb) {
    if constexpr (PROFILE) {
      const uint32_t codes = (uint32_t)read_lane((int)chunk, q);
      set_row(codes);
      if (q == 0) {
        build_profile(codes, j & 1u);
        issue_profile_loads(j & 1u);
        merge_next(codes);
        issue_table_reads((uint32_t)read_lane((int)chunk, 1));
        pack_table_reads();
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) sc[c] = scq[c];
      write_profile((j + 1u) & 1u);
      issue_profile_loads((j + 1u) & 1u);
      issue_table_reads((uint32_t)read_lane((int)chunk, (q + 2) & 63));
    }
  }
  __device__ __forceinline__ void row_end(uint32_t chunk, int q, uint32_t j, uint32_t lb) {
    if constexpr (PROFILE) {
      merge_next((uint32_t)read_lane((int)chunk, (q + 1) & 63));
      pack_table_reads();
      // (here, not wherever the scheduler likes: behind the ring appends and the next block's read the wait for these two
      // entries also waits for those)
      asm volatile("" : "+v"(tvw) : : "memory");
    }
  }
  __device__ __forceinline__ pk16 score(int c) const {
    if constexpr (PROFILE) {
      return sc[c];
    } else {
    const uint32_t ne01 = pk_min_u16(fa[c] ^ fb, 0x00010001u);
    if constexpr (SUBST == SA_SUBST_SIMPLE) {
      return pk_mad(ne01, s_delta, s_eq);
    } else {
      extern __shared__ __attribute__((aligned(16))) int32_t lds_base[];
      const char *l = reinterpret_cast<const char *>(lds_base);
      const short v0 = *reinterpret_cast<const short *>(l + ar0[c] + kb0);
      const short v1 = *reinterpret_cast<const short *>(l + ar1[c] + kb1);
      const uint32_t differ = pk_bits(pk_splat(0) - pk_from(ne01));            // 0xFFFF where the characters differ
      return pk_from(bfi(a0[c] & b0 & differ, pk_bits(s_ne), pk_bits(pk16{v0, v1})));
    }
    }
  }
};

// (behind the table: room for the row profiles of up to eight waves)
static inline size_t table_lds_bytes(const SaFillParams &p) { return ((((size_t)p.K * p.K * 2u) + 15u) & ~(size_t)15u) + 8 * kProfBytes; }

// the table into LDS as int16 (every thread of the workgroup, before anyone leaves)
template <int SUBST>
__device__ __forceinline__ void load_table_x2(const SaFillParams &p, uint32_t tbl_lds, int shift = 0) {
  if constexpr (SUBST == SA_SUBST_LDS) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds_base[];
    short *t = reinterpret_cast<short *>(reinterpret_cast<char *>(lds_base) + tbl_lds);
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) t[k] = (short)(p.table[k] + shift);
    __syncthreads();
  }
}

// ---- Needleman-Wunsch, directions only (the packed form of fill_nw_dirs_kernel)
// LANES = 64: one wave's work is the directions-only NW fill of two pairs, pair_lo in the low halves of every register and
// pair_hi in the high halves (has_hi = there is one), CPL columns per lane.
// LANES = 32 (round 4): FOUR pairs per wave -- lanes 0-31 carry one such couple of pairs, lanes 32-63 another (pair_lo /
// pair_hi / has_* differ between the two spans of lanes), CPL columns per lane of a span.  A 151-column row (BASELINE
// configs 2, 3, 5) takes 3 x 64 = 192 cell slots of a wave one way and 5 x 32 = 160 the other, and the per-row work that
// does not depend on the row's width (scan, lane shifts, this row's character, the loop) serves four rows instead of two.
// All pairs of a wave have the same shape: one set of stream positions, one row counter; the spans never exchange anything
// (the lane shifts' hand-over at lane 32 is cut: first_lane).
template <int CPL, int SUBST, int R, int LANES, bool LOCAL = false>
__device__ __forceinline__ void nw_dirs_x2_wave(const SaFillParams &p, uint8_t *__restrict__ dirs_arena, const uint32_t pair_lo,
                                                const uint32_t pair_hi, const bool has_lo, const bool has_hi, const int lane,
                                                uint8_t *ring_wave, const uint32_t tbl_lds) {
  static_assert(LANES == 64 || LANES == 32, "a span of lanes is the wave or half of it");
  constexpr int kBytes = 256 / LANES;      // of a 256-byte block that a lane moves from a ring to HBM: 4 or 8
  const int sl = LANES == 64 ? lane : (lane & 31);
  const uint32_t span = LANES == 64 ? 0u : (uint32_t)lane >> 5;
  const bool first_lane = sl == 0;
  const uint32_t la = p.len_a[pair_lo], lb = p.len_b[pair_lo], W = la + 1;   // (the same for every pair: the launcher checked)
  const uint8_t *__restrict__ sa0 = p.arena + p.off_a[pair_lo], *__restrict__ sa1 = p.arena + p.off_a[pair_hi];
  const uint8_t *__restrict__ sb0 = p.arena + p.off_b[pair_lo], *__restrict__ sb1 = p.arena + p.off_b[pair_hi];
  uint8_t *const gd0 = dirs_arena + p.mat_off[pair_lo], *const gd1 = dirs_arena + p.mat_off[pair_hi];   // 256-byte aligned
  // (kNwDetrend: every value V of cell (g, j) is kept as V - (g + j) ext: `open1` becomes gap_open, `+ ext` and the scan's re-trend vanish)
  const pk16 open1 = pk_splat(kNwDetrend ? p.gap_open : p.open1), ext = pk_splat(p.ext), floor_ = pk_splat(-32768);
  SubstX2<SUBST, CPL, (SUBST == SA_SUBST_LDS && LANES == 64)> sub;
  sub.init(p, tbl_lds, lane, kNwDetrend ? -2 * p.ext : 0);
  const Border bd{p.floor, p.gap_open, p.ext, false, false};

  constexpr bool BLK = kDirsBlocked && LANES * CPL <= 512;   // (rows of up to 512 columns: sa_dirs_blocked_shape)
  uint8_t *ring0 = ring_wave + span * (2 * R), *ring1 = ring0 + R;   // my span's rings: low halves, high halves
  uint32_t wv = 0, rv = 0;   // stream positions = cell indices: written up to wv, flushed up to rv
  auto flush_block = [&]() __attribute__((always_inline)) {
    const uint32_t o = (rv & (R - 1)) + kBytes * sl;
    if constexpr (LANES == 64) {
      const uint32_t d0 = *reinterpret_cast<const uint32_t *>(ring0 + o);
      const uint32_t d1 = *reinterpret_cast<const uint32_t *>(ring1 + o);
      if (has_lo) __builtin_nontemporal_store(d0, reinterpret_cast<uint32_t *>(gd0 + rv + kBytes * sl));
      if (has_hi) __builtin_nontemporal_store(d1, reinterpret_cast<uint32_t *>(gd1 + rv + kBytes * sl));
    } else {
      typedef uint32_t v2u_a __attribute__((ext_vector_type(2)));
      const v2u_a d0 = *reinterpret_cast<const v2u_a *>(ring0 + o);
      const v2u_a d1 = *reinterpret_cast<const v2u_a *>(ring1 + o);
      if (has_lo) __builtin_nontemporal_store(d0, reinterpret_cast<v2u_a *>(gd0 + rv + kBytes * sl));
      if (has_hi) __builtin_nontemporal_store(d1, reinterpret_cast<v2u_a *>(gd1 + rv + kBytes * sl));
    }
    rv += 256;
  };
  // kLdsPipe: a block is READ out of the rings at the end of the row that completed it and STORED at the end of the next row --
  // its words have had a row's work to arrive, where flush_block waits for them on the spot (every row completes at least one
  // block from W = 257 columns on; a second one in the same row goes the old way).  In-order LDS: the ring bytes a read was
  // issued for may be overwritten by later appends, the read still sees the old ones.
  typedef uint32_t fl_word_t __attribute__((ext_vector_type(LANES == 64 ? 1 : 2)));
  fl_word_t fd0 = 0, fd1 = 0;
  uint32_t pend_rv = 0;
  bool pend = false;
  auto flush_issue = [&]() __attribute__((always_inline)) {
    const uint32_t o = (rv & (R - 1)) + kBytes * sl;
    fd0 = *reinterpret_cast<const fl_word_t *>(ring0 + o);
    fd1 = *reinterpret_cast<const fl_word_t *>(ring1 + o);
    pend_rv = rv; pend = true;
    rv += 256;
  };
  auto flush_commit = [&]() __attribute__((always_inline)) {
    if (pend) {
      if (has_lo) __builtin_nontemporal_store(fd0, reinterpret_cast<fl_word_t *>(gd0 + pend_rv + kBytes * sl));
      if (has_hi) __builtin_nontemporal_store(fd1, reinterpret_cast<fl_word_t *>(gd1 + pend_rv + kBytes * sl));
      pend = false;
    }
  };
  auto flush_rest = [&]() __attribute__((always_inline)) {
    if constexpr (BLK) return;   // (the pair's last row flushed its block row)
    flush_commit();
    while (rv < wv) flush_block();
  };
  // BLK (round 6, sa_kernels.h: the direction bytes in blocks of 8 rows x 16 columns): the LDS buffer is ONE BLOCK ROW of the
  // pair -- nbx blocks of 128 bytes, laid out as they lie in memory -- my column g's byte of row j at
  // (g / 16) * 128 + (j % 8) * 16 + g % 16; after the block row's eighth row (or the pair's last) its nbx x 128 bytes leave as
  // 16 bytes per lane, contiguous.  No stream positions, no ring arithmetic per cell, one flush per eight rows.
  uint32_t cur_row = 0;                    // (BLK) the row append_row is called for
  uint32_t cb[BLK ? CPL : 1];              // (BLK) my columns' place in a block row
  const uint32_t nbx = (W + 15u) >> 4;
  if constexpr (BLK) {
    static_assert(!BLK || LANES * CPL * 8 <= R, "the LDS buffer holds a block row of LANES x CPL columns");
#pragma unroll
    for (int c = 0; c < CPL; ++c) { const uint32_t g = (uint32_t)(sl * CPL + c); cb[c] = (g >> 4) * 128u + (g & 15u); }
  }
  auto append_row = [&](const uint32_t (&dv)[CPL]) __attribute__((always_inline)) {
    static_assert(255 + LANES * CPL <= R, "ring too small for unpredicated appends");
    if constexpr (BLK) {
      typedef uint32_t blk_v4 __attribute__((ext_vector_type(4)));
      const uint32_t ro = (cur_row & 7u) << 4;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        ring0[cb[c] + ro] = (uint8_t)dv[c];
        ring1[cb[c] + ro] = (uint8_t)(dv[c] >> 16);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if ((cur_row & 7u) == 7u || cur_row == lb) {   // (wave-uniform) the block row is complete, or it is the pair's last
        const uint32_t bytes = nbx * 128u;
        const uint64_t at = (uint64_t)(cur_row >> 3) * bytes;
        for (uint32_t o = (uint32_t)sl * 16u; o < bytes; o += LANES * 16u) {
          const blk_v4 q0 = *reinterpret_cast<const blk_v4 *>(ring0 + o), q1 = *reinterpret_cast<const blk_v4 *>(ring1 + o);
          if (has_lo) __builtin_nontemporal_store(q0, reinterpret_cast<blk_v4 *>(gd0 + at + o));
          if (has_hi) __builtin_nontemporal_store(q1, reinterpret_cast<blk_v4 *>(gd1 + at + o));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      return;
    }
    if constexpr (kFlushDefer) flush_commit();
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t o = (wv + sl * CPL + c) & (R - 1);
      ring0[o] = (uint8_t)dv[c];             // ds_write_b8
      ring1[o] = (uint8_t)(dv[c] >> 16);     // ds_write_b8_d16_hi
    }
    wv += W;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if constexpr (kFlushDefer) {
      while (wv - rv >= 512u) flush_block();
      if (wv - rv >= 256u) flush_issue();
    } else {
      while (wv - rv >= 256u) flush_block();
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };

  pk16 X[CPL], Yp[CPL], Ap[CPL];          // previous row: max3(M,A,B), max(M,B), A
  pk16 c1[CPL], c3[CPL];                  // gap_b scan constants (sa_rowsweep.hpp): open1 - g*ext, g*ext
  uint32_t T[CPL], TY4[CPL];              // previous row: which of M/A/B is the max3 (bits 0-1), B >= M ? 2 : 0 (at bits 2-3)
  pk16 mv[CPL], av[CPL], bv[CPL];         // the row just computed (after the loop: the last row, for the end cell)
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const uint32_t g = sl * CPL + c;
    const uint32_t code0 = (g >= 1 && g <= la) ? p.code[sa0[g - 1]] : 0u;
    const uint32_t code1 = (g >= 1 && g <= la) ? p.code[sa1[g - 1]] : 0u;
    sub.set_column(c, code0, code1, p.K, tbl_lds);
    // row 0 (alignment.c:46-69): the same in both halves
    const int m0 = g ? -32768 : 0, a0 = m0, b0 = g ? bd.edge_gap(g) - (kNwDetrend ? (int)g * p.ext : 0) : 0;
    const int x0 = max3i(m0, a0, b0);
    mv[c] = pk_splat(m0); av[c] = pk_splat(a0); bv[c] = pk_splat(b0);
    X[c] = pk_splat(x0); Yp[c] = pk_splat(max(m0, b0)); Ap[c] = pk_splat(a0);
    T[c] = ((a0 == x0) ? 1u : (b0 == x0) ? 2u : 0u) * kBoth;
    TY4[c] = ((b0 >= m0) ? 8u : 0u) * kBoth;
    const int g_ext = (int)g * p.ext;
    c1[c] = pk_splat(kNwDetrend ? p.gap_open : p.open1 - g_ext); c3[c] = pk_splat(kNwDetrend ? 0 : g_ext);
  }
  // The border column (the span's first lane, my column 0; alignment.c:72-80) comes out of the recurrence by itself, no selects
  // in the loop:
  //   gap_a(0, j) = gap_open + j * ext = gap_a(0, j - 1) + ext if the chain starts at "gap_a(0, 0) = gap_open" (max(M, B) of the
  //   cell above is the floor from row 1 on, and 0 + open1 = the same value on row 1);
  //   gap_b(0, j) = the floor: the cell "to the left" is the 0 the lane shift gives that lane, plus a constant that is the floor.
  if (first_lane) { Ap[0] = pk_splat(p.gap_open); c1[0] = floor_; }
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  {
    uint32_t dv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) dv[c] = 0;   // row 0 is never stood on with a move to make
    append_row(dv);
  }

  uint32_t chunk_code = 0;
  for (uint32_t j = 1; j <= lb; ++j) {
    const int q = (j - 1) & (LANES - 1);
    cur_row = j;
    if (q == 0) {
      const uint32_t r = j + sl;
      if (r <= lb) chunk_code = (uint32_t)p.code[sb0[r - 1]] | (uint32_t)p.code[sb1[r - 1]] << 16;
      __builtin_amdgcn_s_waitcnt(kWaitVm0);
    }
    // this row's characters of seq_b: uniform over the wave, or over each span
    if constexpr (LANES == 64) {
      constexpr bool kProfiled = SUBST == SA_SUBST_LDS;
      if constexpr (kProfiled && kLdsPipe) {
        sub.row_begin(chunk_code, q, j, lb);   // (the pipelined form: SubstX2::row_begin / row_end)
      } else {
      const uint32_t codes = (uint32_t)read_lane((int)chunk_code, q);
      sub.set_row(codes);
      // the row's profile (table scorings): built during the row before, or here at a chunk's first row; the next row's goes
      // into the other buffer now, to be there when it is needed
      if (q == 0) sub.build_profile(codes, j & 1u);
      sub.load_profile(j & 1u);
      if (q != LANES - 1 && j < lb) sub.build_profile((uint32_t)read_lane((int)chunk_code, q + 1), (j + 1u) & 1u);
      }
    } else {
      const uint32_t r0 = (uint32_t)read_lane((int)chunk_code, q), r1 = (uint32_t)read_lane((int)chunk_code, q + 32);
      sub.set_row(span ? r1 : r0);
    }
    const pk16 x_ul = pk_shr1_zero(X[CPL - 1]);      // (a span's first lane: the border column, overridden below)
    const uint32_t t_ul = dpp_mov0<0x138>(T[CPL - 1]);
    pk16 z[CPL];
    uint32_t dv[CPL], dvA[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      // substitution: equal characters -> gen_eq, else gen_ne = gen_eq + min(fa ^ fb, 1) * (gen_ne - gen_eq)
      const pk16 s = sub.score(c);
      const pk16 xd = c ? X[c - (c ? 1 : 0)] : x_ul;
      const uint32_t td = c ? T[c - (c ? 1 : 0)] : t_ul;
      pk16 m = pk_adds(xd, s);                                                             // alignment.c:101-116
      const pk16 ae = kNwDetrend ? Ap[c] : pk_adds(Ap[c], ext);
      pk16 a = pk_max(pk_adds(Yp[c], open1), ae);                                          // alignment.c:128-135
      if (c == 0) m = first_lane ? floor_ : m;                                             // the border column's match score
      mv[c] = m; av[c] = a; z[c] = pk_max(m, a);
      if constexpr (LOCAL) {
        dv[c] = put_ge<2>(0u, pk_subs(ae, a));                                             // CA: gap_a + ext IS the max
      } else {
      const uint32_t opened = pk_lt(ae, a);                                                // gap_a + ext is NOT the max
      const uint32_t dA = bfi(opened, TY4[c], 4u * kBoth);                                 // GAP_A (1) first, else B >= M ? 2 : 0
      dv[c] = td; dvA[c] = dA;
      }
    }
    pk16 Pm[CPL];                         // de-trended gap_b: prefix max up to and including my column
    pk16 e;                               //                   prefix max of the lanes to my left
    {
      const pk16 zin = pk_shr1_zero_in<LANES>(z[CPL - 1], first_lane);
      pk16 P[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const pk16 zl = (c == 0) ? zin : z[c - 1];
        pk16 w = pk_adds(zl, c1[c]);
        P[c] = (c == 0) ? w : pk_max(P[c - 1], w);
      }
      e = pk_scan_max_excl<LANES>(P[CPL - 1], first_lane);
#pragma unroll
      for (int c = 0; c < CPL; ++c) { Pm[c] = pk_max(P[c], e); bv[c] = kNwDetrend ? Pm[c] : pk_adds(Pm[c], c3[c]); }
    }
    {
      const pk16 al = pk_shr1_zero(av[CPL - 1]);     // (a span's first lane: only the border cell's byte reads it, overridden below)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const pk16 aL = c ? av[c - (c ? 1 : 0)] : al;
        const pk16 b = bv[c];
        // alignment.c:311-327 for GAP_B: the left cell's gap_a + open1 first, then its gap_b + ext, else match.
        // gap_b(left) + ext == b  <=>  the prefix max did not grow at my column (de-trended: Pm(left) == Pm(mine))
        const pk16 yn = pk_max(mv[c], b);
        if constexpr (LOCAL) {
          // this cell's own comparisons, a bit each (sa_kernels.h): FA, FB, BM (B >= M), GA (A >= max(M, B)); no tags carried
          uint32_t d = put_ge<3>(dv[c], pk_subs(pk_adds(aL, open1), b));
          d = put_ge<4>(d, pk_subs(c ? Pm[c - (c ? 1 : 0)] : e, Pm[c]));
          d = put_ge<1>(d, pk_subs(b, mv[c]));
          dv[c] = put_ge<0>(d, pk_subs(av[c], yn));
          X[c] = pk_max(z[c], b); Yp[c] = yn; Ap[c] = av[c];
        } else {
        const uint32_t not_a = pk_lt(pk_adds(aL, open1), b);
        const uint32_t not_b = pk_lt(c ? Pm[c - (c ? 1 : 0)] : e, Pm[c]);
        const uint32_t dB = bfi(not_a, bfi(not_b, 0u, 32u * kBoth), 16u * kBoth);
        dv[c] = or3(dv[c], dvA[c], dB);
        const uint32_t m_wins = pk_lt(b, mv[c]);      // B < M
        const uint32_t a_loses = pk_lt(av[c], yn);    // A < max(M, B)
        const uint32_t ty4 = bfi(m_wins, 0u, 8u * kBoth);
        X[c] = pk_max(z[c], b); Yp[c] = yn; Ap[c] = av[c];
        T[c] = bfi(a_loses, ty4 >> 2, kBoth);         // GAP_A first, then GAP_B, then MATCH
        TY4[c] = ty4;
        }
      }
    }
    // the border cell (0, j): never stood on with a move to make, but byte for byte what fill_nw_dirs_kernel stores
    // there -- GAP_A continues down the column (its first step only if gap_open is 0), else max(M, B) = B (local form: 0)
    if constexpr (LOCAL) dv[0] = first_lane ? 0u : dv[0];
    else dv[0] = first_lane ? ((j == 1 && p.gap_open != 0) ? 8u : 4u) * kBoth : dv[0];
    if constexpr (LANES == 64 && SUBST == SA_SUBST_LDS && kLdsPipe) sub.row_end(chunk_code, q, j, lb);
    append_row(dv);
  }
  flush_rest();

  // the end cell (la, lb): score and matrix the walk starts in (needleman_wunsch.c:53-66)
  const int owner = (int)(la / CPL), oc = (int)(la % CPL);
  pk16 em = pk_splat(0), ea = em, eb = em;
#pragma unroll
  for (int c = 0; c < CPL; ++c) { if (c == oc) { em = mv[c]; ea = av[c]; eb = bv[c]; } }
  if (sl == owner) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!(h ? has_hi : has_lo)) continue;
      const int m = h ? em.y : em.x, a = h ? ea.y : ea.x, b = h ? eb.y : eb.x;
      int score = m;
      uint32_t st = 0;                                // MATCH
      if (b >= score) { st = 2; score = b; }          // GAP_B
      if (a >= score) { st = 1; score = a; }          // GAP_A
      const uint32_t pr = h ? pair_hi : pair_lo;
      if constexpr (kNwDetrend) score += (int)(la + lb) * p.ext;
      p.best_score[pr] = score;
      p.best_index[pr] = st;
      p.status[pr] = ~0ull;
    }
  }
}

template <int CPL, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_nw_dirs_x2_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t tbl_lds = (blockDim.x >> 6) * (2 * R);   // the table sits behind the rings
  load_table_x2<SUBST>(p, tbl_lds, kNwDetrend ? -2 * p.ext : 0);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t unit = blockIdx.x * (blockDim.x >> 6) + wave;
  if (2 * unit >= p.n_pairs) return;
  const bool two = 2 * unit + 1 < p.n_pairs;       // an odd launch: the last wave's high halves shadow its low ones
  const uint32_t pair0 = p.pair_list ? p.pair_list[2 * unit] : 2 * unit;
  const uint32_t pair1 = two ? (p.pair_list ? p.pair_list[2 * unit + 1] : 2 * unit + 1) : pair0;

  nw_dirs_x2_wave<CPL, SUBST, R, 64, LOCAL>(p, dirs_arena, pair0, pair1, true, two, lane, reinterpret_cast<uint8_t *>(lds) + wave * (2 * R), tbl_lds);
}

// four pairs per wave: pairs 4 unit .. 4 unit + 3 of the launch (all of one shape, no pair list); the pairs a short last wave
// does not have shadow its first one
template <int CPL, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_nw_dirs_x4_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t tbl_lds = (blockDim.x >> 6) * (4 * R);   // the table sits behind the rings
  load_table_x2<SUBST>(p, tbl_lds, kNwDetrend ? -2 * p.ext : 0);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t unit = blockIdx.x * (blockDim.x >> 6) + wave;
  if (4 * unit >= p.n_pairs) return;
  const uint32_t lo = 4 * unit + 2 * ((uint32_t)lane >> 5), hi = lo + 1;
  const bool has_lo = lo < p.n_pairs, has_hi = hi < p.n_pairs;
  nw_dirs_x2_wave<CPL, SUBST, R, 32, LOCAL>(p, dirs_arena, has_lo ? lo : 4 * unit, has_hi ? hi : 4 * unit, has_lo, has_hi, lane,
                                     reinterpret_cast<uint8_t *>(lds) + wave * (4 * R), tbl_lds);
}

// Four AND two pairs per wave in one grid, for a launch whose last round of four-per-wave waves would be less than half full:
// the first q_blocks workgroups take pairs [0, pairs_q) four per wave (pairs_q: a whole number of rounds of 1 024 waves), the
// others the remaining pairs two per wave.  C2's 10 000 pairs are 2 500 four-per-wave waves: 452 SIMDs get three of them and
// the rest wait with two (0.56 of the issue peak); as 2 048 four-per-wave + 904 two-per-wave waves every SIMD has two of the
// former and at most one of the latter, ~12 % less work on the busiest SIMD (sa_launch_fill_nw_dirs_x2 decides).
template <int CPL4, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_nw_dirs_x4x2_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena, const uint32_t q_blocks, const uint32_t pairs_q) {
  constexpr int CPL2 = (CPL4 + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t tbl_lds = waves * (4 * R);   // the table sits behind the rings (the two-per-wave waves use half of theirs)
  load_table_x2<SUBST>(p, tbl_lds, kNwDetrend ? -2 * p.ext : 0);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t *ring = reinterpret_cast<uint8_t *>(lds) + wave * (4 * R);
  if (blockIdx.x < q_blocks) {   // (uniform per workgroup)
    const uint32_t unit = blockIdx.x * waves + wave;
    if (4 * unit >= pairs_q) return;
    const uint32_t lo = 4 * unit + 2 * ((uint32_t)lane >> 5);
    nw_dirs_x2_wave<CPL4, SUBST, R, 32, LOCAL>(p, dirs_arena, lo, lo + 1, true, true, lane, ring, tbl_lds);
  } else {
    const uint32_t pair0 = pairs_q + 2 * ((blockIdx.x - q_blocks) * waves + wave);
    if (pair0 >= p.n_pairs) return;
    const bool two = pair0 + 1 < p.n_pairs;
    nw_dirs_x2_wave<CPL2, SUBST, R, 64, LOCAL>(p, dirs_arena, pair0, two ? pair0 + 1 : pair0, true, two, lane, ring, tbl_lds);
  }
}

// The same for a chunk whose pairs are MOSTLY of one shape, in ONE grid: the first x2_blocks workgroups take the n_modal
// pairs of the modal shape two per wave (p.pair_list[0 .. n_modal)), the others the n_rest remaining pairs one per wave
// (p.pair_list[n_modal ..), fill_nw_dirs_kernel's body).  Two launches would run one after the other, and a launch of a
// few hundred one-pair waves takes as long as its longest pair's rows however few they are (profiles/r03/r03_mixed_check.txt; today: tools/x2_check.py check_mixed).
template <int CPL, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_nw_dirs_mixed_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena, const uint32_t x2_blocks, const uint32_t n_modal,
                          const uint32_t n_rest) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t waves = blockDim.x >> 6;
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (blockIdx.x < x2_blocks) {   // (uniform per workgroup)
    const uint32_t tbl_lds = waves * (2 * R);
    load_table_x2<SUBST>(p, tbl_lds, kNwDetrend ? -2 * p.ext : 0);
    const uint32_t unit = blockIdx.x * waves + wave;
    if (2 * unit >= n_modal) return;
    const bool two = 2 * unit + 1 < n_modal;
    const uint32_t pair0 = p.pair_list[2 * unit], pair1 = two ? p.pair_list[2 * unit + 1] : pair0;
    nw_dirs_x2_wave<CPL, SUBST, R, 64, LOCAL>(p, dirs_arena, pair0, pair1, true, two, lane, reinterpret_cast<uint8_t *>(lds) + wave * (2 * R), tbl_lds);
  } else {
    const int32_t *table = p.table;
    if constexpr (SUBST == SA_SUBST_LDS) {
      int32_t *tbl = lds + (waves * R) / 4;
      for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) tbl[k] = p.table[k];
      __syncthreads();
      table = tbl;
    }
    const uint32_t slot = (blockIdx.x - x2_blocks) * waves + wave;
    if (slot >= n_rest) return;
    nw_dirs_x1_wave<CPL, SUBST, R, LOCAL>(p, dirs_arena, p.pair_list[n_modal + slot], lane, reinterpret_cast<uint8_t *>(lds) + wave * R, table);
  }
}

// ---- Smith-Waterman, match_scores + directions (the packed form of fill_dirs_kernel)
// Same outputs as fill_dirs_kernel for every pair of the wave: match_scores (int32 in HBM; the rings hold them as the
// int16 they are computed in, the flush widens them), the direction byte, the candidates' count / box / columns per row.
// Floor 0: max(x, 0) is a real instruction here, and a state whose score is 0 gets the code 3 ("the walk ends").
// (Two pairs per wave only: with four -- LANES = 32 as in nw_dirs_x2_wave -- this kernel, which also moves 2 B of scores per
// cell through LDS and 4 B to HBM, measured 20-30 % SLOWER at every batch size: profiles/r04/r04_quad_fills.txt.)
template <int CPL, int SUBST, int R>
__device__ __forceinline__ void sw_dirs_x2_wave(const SaFillParams &p, uint8_t *__restrict__ dirs_arena, const uint32_t pair_lo,
                                                const uint32_t pair_hi, const bool has_lo, const bool has_hi, const int lane,
                                                uint8_t *ring_wave, const uint32_t tbl_lds) {
  constexpr int LANES = 64;
  constexpr int kCells = 256 / LANES;      // of a 256-cell block that a lane moves from the rings to HBM
  const int sl = LANES == 64 ? lane : (lane & 31);
  const uint32_t span = LANES == 64 ? 0u : (uint32_t)lane >> 5;
  const bool first_lane = sl == 0;
  const uint32_t la = p.len_a[pair_lo], lb = p.len_b[pair_lo], W = la + 1;
  const uint8_t *__restrict__ sa0 = p.arena + p.off_a[pair_lo], *__restrict__ sa1 = p.arena + p.off_a[pair_hi];
  const uint8_t *__restrict__ sb0 = p.arena + p.off_b[pair_lo], *__restrict__ sb1 = p.arena + p.off_b[pair_hi];
  const uint64_t mo0 = p.mat_off[pair_lo], mo1 = p.mat_off[pair_hi];   // multiples of 256 cells
  int32_t *const gm0 = p.M + mo0, *const gm1 = p.M + mo1;
  uint8_t *const gd0 = dirs_arena + mo0, *const gd1 = dirs_arena + mo1;
  const pk16 open1 = pk_splat(p.open1), ext = pk_splat(p.ext), zero = pk_splat(0);
  SubstX2<SUBST, CPL, (SUBST == SA_SUBST_LDS && LANES == 64)> sub;
  sub.init(p, tbl_lds, lane);

  // LDS per span: two rings of R int16 scores, two rings of R direction bytes
  uint8_t *ring = ring_wave + span * (6 * R);
  uint16_t *rm0 = reinterpret_cast<uint16_t *>(ring), *rm1 = rm0 + R;
  uint8_t *rd0 = ring + 4 * R, *rd1 = rd0 + R;
  uint32_t wv = 0, rv = 0;
  auto flush_block = [&]() __attribute__((always_inline)) {
    typedef int v4i_a __attribute__((ext_vector_type(4)));
    const uint32_t o = (rv & (R - 1)) + kCells * sl;
    {
      const uint2 q0 = *reinterpret_cast<const uint2 *>(rm0 + o), q1 = *reinterpret_cast<const uint2 *>(rm1 + o);
      const uint32_t d0 = *reinterpret_cast<const uint32_t *>(rd0 + o), d1 = *reinterpret_cast<const uint32_t *>(rd1 + o);
      if (has_lo) {
        const v4i_a m0 = {(int)(q0.x & 0xffffu), (int)(q0.x >> 16), (int)(q0.y & 0xffffu), (int)(q0.y >> 16)};   // scores are >= 0
        __builtin_nontemporal_store(m0, reinterpret_cast<v4i_a *>(gm0 + rv + kCells * sl));
        __builtin_nontemporal_store(d0, reinterpret_cast<uint32_t *>(gd0 + rv + kCells * sl));
      }
      if (has_hi) {
        const v4i_a m1 = {(int)(q1.x & 0xffffu), (int)(q1.x >> 16), (int)(q1.y & 0xffffu), (int)(q1.y >> 16)};
        __builtin_nontemporal_store(m1, reinterpret_cast<v4i_a *>(gm1 + rv + kCells * sl));
        __builtin_nontemporal_store(d1, reinterpret_cast<uint32_t *>(gd1 + rv + kCells * sl));
      }
    }
    rv += 256;
  };
  // (kLdsPipe: read a block at the end of the row that completed it, store it at the end of the next -- see nw_dirs_x2_wave)
  uint2 fq0 = {0, 0}, fq1 = {0, 0};
  uint32_t fb0 = 0, fb1 = 0, pend_rv = 0;
  bool pend = false;
  auto flush_issue = [&]() __attribute__((always_inline)) {
    const uint32_t o = (rv & (R - 1)) + kCells * sl;
    fq0 = *reinterpret_cast<const uint2 *>(rm0 + o); fq1 = *reinterpret_cast<const uint2 *>(rm1 + o);
    fb0 = *reinterpret_cast<const uint32_t *>(rd0 + o); fb1 = *reinterpret_cast<const uint32_t *>(rd1 + o);
    pend_rv = rv; pend = true;
    rv += 256;
  };
  auto flush_commit = [&]() __attribute__((always_inline)) {
    typedef int v4i_a __attribute__((ext_vector_type(4)));
    if (pend) {
      if (has_lo) {
        const v4i_a m0 = {(int)(fq0.x & 0xffffu), (int)(fq0.x >> 16), (int)(fq0.y & 0xffffu), (int)(fq0.y >> 16)};
        __builtin_nontemporal_store(m0, reinterpret_cast<v4i_a *>(gm0 + pend_rv + kCells * sl));
        __builtin_nontemporal_store(fb0, reinterpret_cast<uint32_t *>(gd0 + pend_rv + kCells * sl));
      }
      if (has_hi) {
        const v4i_a m1 = {(int)(fq1.x & 0xffffu), (int)(fq1.x >> 16), (int)(fq1.y & 0xffffu), (int)(fq1.y >> 16)};
        __builtin_nontemporal_store(m1, reinterpret_cast<v4i_a *>(gm1 + pend_rv + kCells * sl));
        __builtin_nontemporal_store(fb1, reinterpret_cast<uint32_t *>(gd1 + pend_rv + kCells * sl));
      }
      pend = false;
    }
  };
  auto append_row = [&](const pk16 (&mv)[CPL], const uint32_t (&dv)[CPL]) __attribute__((always_inline)) {
    static_assert(255 + LANES * CPL <= R, "ring too small for unpredicated appends");
    if constexpr (kFlushDefer) flush_commit();
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t o = (wv + sl * CPL + c) & (R - 1);
      const uint32_t mb = pk_bits(mv[c]);
      rm0[o] = (uint16_t)mb; rm1[o] = (uint16_t)(mb >> 16);         // ds_write_b16 / ds_write_b16_d16_hi
      rd0[o] = (uint8_t)dv[c]; rd1[o] = (uint8_t)(dv[c] >> 16);     // ds_write_b8 / ds_write_b8_d16_hi
    }
    wv += W;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if constexpr (kFlushDefer) {
      while (wv - rv >= 512u) flush_block();
      if (wv - rv >= 256u) flush_issue();
    } else {
      while (wv - rv >= 256u) flush_block();
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };

  uint32_t valid[CPL];                    // all ones where my column exists
  pk16 X[CPL], Yp[CPL], Ap[CPL];          // previous row: max3(M,A,B), max(M,B), A
  pk16 c1[CPL], c2[CPL], c3[CPL];         // gap_b scan constants (sa_rowsweep.hpp), floor 0
  uint32_t T[CPL], TY4[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const uint32_t g = sl * CPL + c;
    const uint32_t code0 = (g >= 1 && g <= la) ? p.code[sa0[g - 1]] : 0u;
    const uint32_t code1 = (g >= 1 && g <= la) ? p.code[sa1[g - 1]] : 0u;
    sub.set_column(c, code0, code1, p.K, tbl_lds);
    valid[c] = g <= la ? 0xffffffffu : 0u;
    X[c] = Yp[c] = Ap[c] = zero;                            // row 0: borders are 0 (alignment.c:51-57)
    T[c] = 1u * kBoth; TY4[c] = 8u * kBoth;                 // (A == max3 and B >= M hold on a row of zeros; never followed)
    const int g_ext = (int)g * p.ext;
    c1[c] = pk_splat(p.open1 - g_ext); c2[c] = pk_splat(-g_ext); c3[c] = pk_splat(g_ext);
  }
  __builtin_amdgcn_s_waitcnt(kWaitVm0);

  // candidates per pair: whether any cell is >= min_score and the first / last ROW with one -- all the sweep behind this fill
  // (sw_sweep_dirs*_kernel: whole rows in registers) wants to know; the per-row columns and the box's columns that the
  // three-matrix fills report serve the LDS / strip forms of the sweep, which never run behind a direction fill.  Kept per
  // lane (no ballots, no scalar code in the row loop: ~12 instead of ~45 instructions per row) and reduced at the end.
  const pk16 thr = pk16{(short)min(max(p.cand_min[pair_lo], 1), 32767), (short)min(max(p.cand_min[pair_hi], 1), 32767)};
  uint32_t first_row[2] = {0xffffffffu, 0xffffffffu}, last_row[2] = {0, 0};

  {  // row 0: scores 0, every state ends
    pk16 mv[CPL];
    uint32_t dv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { mv[c] = zero; dv[c] = 0x3fu * kBoth; }
    append_row(mv, dv);
  }

  uint32_t chunk_code = 0;
  for (uint32_t j = 1; j <= lb; ++j) {
    const int q = (j - 1) & (LANES - 1);
    if (q == 0) {
      const uint32_t r = j + sl;
      if (r <= lb) chunk_code = (uint32_t)p.code[sb0[r - 1]] | (uint32_t)p.code[sb1[r - 1]] << 16;
      __builtin_amdgcn_s_waitcnt(kWaitVm0);
    }
    if constexpr (LANES == 64 && SUBST == SA_SUBST_LDS && kLdsPipe) {
      sub.row_begin(chunk_code, q, j, lb);   // (the pipelined form: SubstX2::row_begin / row_end)
    } else {
      const uint32_t codes = row_codes<LANES>(chunk_code, q, span);
      sub.set_row(codes);
      if constexpr (LANES == 64) {   // the row's profile (table scorings): see nw_dirs_x2_wave
        if (q == 0) sub.build_profile(codes, j & 1u);
        sub.load_profile(j & 1u);
        if (q != LANES - 1 && j < lb) sub.build_profile(row_codes<LANES>(chunk_code, q + 1, span), (j + 1u) & 1u);
      }
    }
    // up-left of my first column: the left lane's last column on the previous row; a span's first lane (the border column): far
    // enough below zero that M = max(.., 0) = 0
    const pk16 x_ul = pk_shr1_in<LANES>(X[CPL - 1], pk_splat(-16384), first_lane);
    const uint32_t t_ul = dpp_mov0<0x138>(T[CPL - 1]);
    pk16 mv[CPL], av[CPL], bv[CPL], z[CPL];
    uint32_t dv[CPL], dvA[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const pk16 s = sub.score(c);
      const pk16 xd = c ? X[c - (c ? 1 : 0)] : x_ul;
      const uint32_t td = c ? T[c - (c ? 1 : 0)] : t_ul;
      const pk16 m = pk_max(pk_adds(xd, s), zero);                                         // alignment.c:101-116
      const pk16 ae = pk_adds(Ap[c], ext);
      const pk16 a = pk_max(pk_max(pk_adds(Yp[c], open1), ae), zero);                      // alignment.c:128-135
      // where a walk goes from here (alignment.c:311-327); a state whose score is 0: 3
      const uint32_t opened = pk_lt(ae, a);
      const uint32_t dA = bfi(opened, TY4[c], 4u * kBoth);
      mv[c] = m; av[c] = a; z[c] = pk_max(m, a);
      dv[c] = bfi(pos_mask(m), td, 3u * kBoth); dvA[c] = bfi(pos_mask(a), dA, 12u * kBoth);
    }
    pk16 Pm[CPL], e;
    {
      const pk16 zin = pk_shr1_zero_in<LANES>(z[CPL - 1], first_lane);
      pk16 P[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const pk16 zl = (c == 0) ? zin : z[c - 1];
        pk16 w = pk_max(pk_adds(zl, c1[c]), c2[c]);
        // (a span's first lane, column 0: the lane shift gives 0 for the cell to the left, and max(0 + open1, 0) = 0 = gap_b of the border)
        P[c] = (c == 0) ? w : pk_max(P[c - 1], w);
      }
      e = pk_scan_max_excl<LANES>(P[CPL - 1], first_lane);
#pragma unroll
      for (int c = 0; c < CPL; ++c) { Pm[c] = pk_max(P[c], e); bv[c] = pk_adds(Pm[c], c3[c]); }
    }
    {
      const pk16 al = pk_shr1_zero(av[CPL - 1]);   // (a span's first lane: gap_b of the border is 0, its byte does not depend on this)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const pk16 aL = c ? av[c - (c ? 1 : 0)] : al;
        const pk16 b = bv[c];
        const uint32_t not_a = pk_lt(pk_adds(aL, open1), b);
        const uint32_t not_b = pk_lt(c ? Pm[c - (c ? 1 : 0)] : e, Pm[c]);
        const uint32_t dB = bfi(not_a, bfi(not_b, 0u, 32u * kBoth), 16u * kBoth);
        dv[c] = or3(dv[c], dvA[c], bfi(pos_mask(b), dB, 48u * kBoth));
        const pk16 yn = pk_max(mv[c], b);
        const uint32_t m_wins = pk_lt(b, mv[c]);
        const uint32_t a_loses = pk_lt(av[c], yn);
        const uint32_t ty4 = bfi(m_wins, 0u, 8u * kBoth);
        X[c] = pk_max(z[c], b); Yp[c] = yn; Ap[c] = av[c];
        T[c] = bfi(a_loses, ty4 >> 2, kBoth);
        TY4[c] = ty4;
      }
    }
    if constexpr (LANES == 64 && SUBST == SA_SUBST_LDS && kLdsPipe) sub.row_end(chunk_code, q, j, lb);
    append_row(mv, dv);

    // candidates of this row in my columns, per pair
    pk16 best = pk_from(pk_bits(mv[0]) & valid[0]);
#pragma unroll
    for (int c = 1; c < CPL; ++c) best = pk_max(best, pk_from(pk_bits(mv[c]) & valid[c]));
    const uint32_t below = pk_lt(best, thr);                      // 0xFFFF in the halves without a candidate in my columns
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool has = ((h ? below >> 16 : below & 0xffffu)) == 0;
      first_row[h] = has ? min(first_row[h], j) : first_row[h];
      last_row[h] = has ? j : last_row[h];
    }
  }
  flush_commit();
  while (rv < wv) flush_block();

#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) {
      first_row[h] = min(first_row[h], (uint32_t)__shfl_xor((int)first_row[h], o));
      last_row[h] = max(last_row[h], (uint32_t)__shfl_xor((int)last_row[h], o));
    }
  }
  if (first_lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!(h ? has_hi : has_lo)) continue;
      const uint32_t pr = h ? pair_hi : pair_lo;
      p.cand_count[pr] = first_row[h] != 0xffffffffu;
      uint32_t *box = p.cand_box + 4ull * pr;
      box[0] = first_row[h]; box[1] = last_row[h]; box[2] = 0; box[3] = W - 1;   // (columns: the whole row; see above)
      p.status[pr] = ~0ull;   // plain scorings have a score for every pair of characters
    }
  }
}

// pairs 2 unit, 2 unit + 1 of the launch -- or, with a pair list (ragged chunks: sa_batch.hip pairs up the pairs of equal
// shape), the pairs pair_list[2 unit], pair_list[2 unit + 1] of the descriptor arrays; an entry repeated = a pair that
// found no partner and has the wave to itself
__device__ __forceinline__ bool x2_pairs_of(const SaFillParams &p, uint32_t unit, uint32_t *pair0, uint32_t *pair1) {
  *pair0 = p.pair_list ? p.pair_list[2 * unit] : 2 * unit;
  *pair1 = p.pair_list ? p.pair_list[2 * unit + 1] : (2 * unit + 1 < p.n_pairs ? 2 * unit + 1 : *pair0);
  return *pair1 != *pair0;
}
// pairs 4 unit .. 4 unit + 3 of the launch (all of one shape, no pair list): my span's couple; the pairs a short last wave does
// not have shadow its first one
__device__ __forceinline__ void x4_pairs_of(const SaFillParams &p, uint32_t unit, int lane, uint32_t *lo, uint32_t *hi, bool *has_lo,
                                            bool *has_hi) {
  const uint32_t l = 4 * unit + 2 * ((uint32_t)lane >> 5), h = l + 1;
  *has_lo = l < p.n_pairs; *has_hi = h < p.n_pairs;
  *lo = *has_lo ? l : 4 * unit; *hi = *has_hi ? h : 4 * unit;
}

template <int CPL, int SUBST, int R>
__global__ void __launch_bounds__(kWave * 4)
fill_dirs_x2_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t tbl_lds = (blockDim.x >> 6) * (6 * R);   // the table sits behind the rings
  load_table_x2<SUBST>(p, tbl_lds);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t unit = blockIdx.x * (blockDim.x >> 6) + wave;
  if (2 * unit >= p.n_pairs) return;
  uint32_t pair0, pair1;
  const bool two = x2_pairs_of(p, unit, &pair0, &pair1);
  sw_dirs_x2_wave<CPL, SUBST, R>(p, dirs_arena, pair0, pair1, true, two, lane, reinterpret_cast<uint8_t *>(lds) + wave * (6 * R), tbl_lds);
}

// ---- Smith-Waterman, best hit only (seqalign_sw_batch with max_hits = 1): directions + the best cell, no match_scores
// The best-hit path needs from the matrices what seqalign_nw_batch needs -- where a walk goes -- plus where it starts:
// the best match_scores cell in the reference's hit order (score descending, then column, then row ascending:
// smith_waterman.c:71-86).  So this fill writes the direction byte only (1 B per cell, as fill_nw_dirs_x2_kernel) and
// tracks, per column, the highest score and the first row that reached it; the pair's best cell and score go to
// best_index / best_score (index 0, score 0: no cell above 0).
template <int CPL, int SUBST, int R, int LANES, bool LOCAL = false>
__device__ __forceinline__ void sw_best_x2_wave(const SaFillParams &p, uint8_t *__restrict__ dirs_arena, const uint32_t pair_lo,
                                                const uint32_t pair_hi, const bool has_lo, const bool has_hi, const int lane,
                                                uint8_t *ring_wave, const uint32_t tbl_lds) {
  constexpr int kBytes = 256 / LANES;
  const int sl = LANES == 64 ? lane : (lane & 31);
  const uint32_t span = LANES == 64 ? 0u : (uint32_t)lane >> 5;
  const bool first_lane = sl == 0;
  const uint32_t la = p.len_a[pair_lo], lb = p.len_b[pair_lo], W = la + 1;
  const uint8_t *__restrict__ sa0 = p.arena + p.off_a[pair_lo], *__restrict__ sa1 = p.arena + p.off_a[pair_hi];
  const uint8_t *__restrict__ sb0 = p.arena + p.off_b[pair_lo], *__restrict__ sb1 = p.arena + p.off_b[pair_hi];
  uint8_t *const gd0 = dirs_arena + p.mat_off[pair_lo], *const gd1 = dirs_arena + p.mat_off[pair_hi];   // 256-byte aligned
  const pk16 open1 = pk_splat(p.open1), ext = pk_splat(p.ext), zero = pk_splat(0);
  SubstX2<SUBST, CPL, (SUBST == SA_SUBST_LDS && LANES == 64)> sub;
  sub.init(p, tbl_lds, lane);

  constexpr bool BLK = kDirsBlocked && LANES * CPL <= 512;   // (rows of up to 512 columns: sa_dirs_blocked_shape)
  uint8_t *ring0 = ring_wave + span * (2 * R), *ring1 = ring0 + R;
  uint32_t wv = 0, rv = 0;
  auto flush_block = [&]() __attribute__((always_inline)) {
    const uint32_t o = (rv & (R - 1)) + kBytes * sl;
    if constexpr (LANES == 64) {
      const uint32_t d0 = *reinterpret_cast<const uint32_t *>(ring0 + o);
      const uint32_t d1 = *reinterpret_cast<const uint32_t *>(ring1 + o);
      if (has_lo) __builtin_nontemporal_store(d0, reinterpret_cast<uint32_t *>(gd0 + rv + kBytes * sl));
      if (has_hi) __builtin_nontemporal_store(d1, reinterpret_cast<uint32_t *>(gd1 + rv + kBytes * sl));
    } else {
      typedef uint32_t v2u_a __attribute__((ext_vector_type(2)));
      const v2u_a d0 = *reinterpret_cast<const v2u_a *>(ring0 + o);
      const v2u_a d1 = *reinterpret_cast<const v2u_a *>(ring1 + o);
      if (has_lo) __builtin_nontemporal_store(d0, reinterpret_cast<v2u_a *>(gd0 + rv + kBytes * sl));
      if (has_hi) __builtin_nontemporal_store(d1, reinterpret_cast<v2u_a *>(gd1 + rv + kBytes * sl));
    }
    rv += 256;
  };
  // kLdsPipe: a block is READ out of the rings at the end of the row that completed it and STORED at the end of the next row --
  // its words have had a row's work to arrive, where flush_block waits for them on the spot (every row completes at least one
  // block from W = 257 columns on; a second one in the same row goes the old way).  In-order LDS: the ring bytes a read was
  // issued for may be overwritten by later appends, the read still sees the old ones.
  typedef uint32_t fl_word_t __attribute__((ext_vector_type(LANES == 64 ? 1 : 2)));
  fl_word_t fd0 = 0, fd1 = 0;
  uint32_t pend_rv = 0;
  bool pend = false;
  auto flush_issue = [&]() __attribute__((always_inline)) {
    const uint32_t o = (rv & (R - 1)) + kBytes * sl;
    fd0 = *reinterpret_cast<const fl_word_t *>(ring0 + o);
    fd1 = *reinterpret_cast<const fl_word_t *>(ring1 + o);
    pend_rv = rv; pend = true;
    rv += 256;
  };
  auto flush_commit = [&]() __attribute__((always_inline)) {
    if (pend) {
      if (has_lo) __builtin_nontemporal_store(fd0, reinterpret_cast<fl_word_t *>(gd0 + pend_rv + kBytes * sl));
      if (has_hi) __builtin_nontemporal_store(fd1, reinterpret_cast<fl_word_t *>(gd1 + pend_rv + kBytes * sl));
      pend = false;
    }
  };
  auto flush_rest = [&]() __attribute__((always_inline)) {
    if constexpr (BLK) return;   // (the pair's last row flushed its block row)
    flush_commit();
    while (rv < wv) flush_block();
  };
  // BLK (round 6, sa_kernels.h: the direction bytes in blocks of 8 rows x 16 columns): the LDS buffer is ONE BLOCK ROW of the
  // pair -- nbx blocks of 128 bytes, laid out as they lie in memory -- my column g's byte of row j at
  // (g / 16) * 128 + (j % 8) * 16 + g % 16; after the block row's eighth row (or the pair's last) its nbx x 128 bytes leave as
  // 16 bytes per lane, contiguous.  No stream positions, no ring arithmetic per cell, one flush per eight rows.
  uint32_t cur_row = 0;                    // (BLK) the row append_row is called for
  uint32_t cb[BLK ? CPL : 1];              // (BLK) my columns' place in a block row
  const uint32_t nbx = (W + 15u) >> 4;
  if constexpr (BLK) {
    static_assert(!BLK || LANES * CPL * 8 <= R, "the LDS buffer holds a block row of LANES x CPL columns");
#pragma unroll
    for (int c = 0; c < CPL; ++c) { const uint32_t g = (uint32_t)(sl * CPL + c); cb[c] = (g >> 4) * 128u + (g & 15u); }
  }
  auto append_row = [&](const uint32_t (&dv)[CPL]) __attribute__((always_inline)) {
    static_assert(255 + LANES * CPL <= R, "ring too small for unpredicated appends");
    if constexpr (BLK) {
      typedef uint32_t blk_v4 __attribute__((ext_vector_type(4)));
      const uint32_t ro = (cur_row & 7u) << 4;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        ring0[cb[c] + ro] = (uint8_t)dv[c];
        ring1[cb[c] + ro] = (uint8_t)(dv[c] >> 16);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if ((cur_row & 7u) == 7u || cur_row == lb) {   // (wave-uniform) the block row is complete, or it is the pair's last
        const uint32_t bytes = nbx * 128u;
        const uint64_t at = (uint64_t)(cur_row >> 3) * bytes;
        for (uint32_t o = (uint32_t)sl * 16u; o < bytes; o += LANES * 16u) {
          const blk_v4 q0 = *reinterpret_cast<const blk_v4 *>(ring0 + o), q1 = *reinterpret_cast<const blk_v4 *>(ring1 + o);
          if (has_lo) __builtin_nontemporal_store(q0, reinterpret_cast<blk_v4 *>(gd0 + at + o));
          if (has_hi) __builtin_nontemporal_store(q1, reinterpret_cast<blk_v4 *>(gd1 + at + o));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      return;
    }
    if constexpr (kFlushDefer) flush_commit();
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t o = (wv + sl * CPL + c) & (R - 1);
      ring0[o] = (uint8_t)dv[c];
      ring1[o] = (uint8_t)(dv[c] >> 16);
    }
    wv += W;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if constexpr (kFlushDefer) {
      while (wv - rv >= 512u) flush_block();
      if (wv - rv >= 256u) flush_issue();
    } else {
      while (wv - rv >= 256u) flush_block();
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };

  uint32_t valid[CPL];
  pk16 X[CPL], Yp[CPL], Ap[CPL], c1[CPL], c2[CPL], c3[CPL];
  uint32_t T[CPL], TY4[CPL];
  pk16 best_s[CPL];                       // per column: the highest score so far ...
  uint32_t best_r[CPL];                   // ... and the first row that reached it (one pair per half)
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const uint32_t g = sl * CPL + c;
    const uint32_t code0 = (g >= 1 && g <= la) ? p.code[sa0[g - 1]] : 0u;
    const uint32_t code1 = (g >= 1 && g <= la) ? p.code[sa1[g - 1]] : 0u;
    sub.set_column(c, code0, code1, p.K, tbl_lds);
    valid[c] = g <= la ? 0xffffffffu : 0u;
    X[c] = Yp[c] = Ap[c] = zero;
    T[c] = 1u * kBoth; TY4[c] = 8u * kBoth;
    best_s[c] = zero; best_r[c] = 0;
    const int g_ext = (int)g * p.ext;
    c1[c] = pk_splat(p.open1 - g_ext); c2[c] = pk_splat(-g_ext); c3[c] = pk_splat(g_ext);
  }
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  {
    uint32_t dv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) dv[c] = (LOCAL ? 7u * SA_LD_END0 : 0x3fu) * kBoth;   // row 0: scores 0, every state ends
    append_row(dv);
  }

  uint32_t chunk_code = 0;
  for (uint32_t j = 1; j <= lb; ++j) {
    const int q = (j - 1) & (LANES - 1);
    cur_row = j;
    if (q == 0) {
      const uint32_t r = j + sl;
      if (r <= lb) chunk_code = (uint32_t)p.code[sb0[r - 1]] | (uint32_t)p.code[sb1[r - 1]] << 16;
      __builtin_amdgcn_s_waitcnt(kWaitVm0);
    }
    if constexpr (LANES == 64 && SUBST == SA_SUBST_LDS && kLdsPipe) {
      sub.row_begin(chunk_code, q, j, lb);   // (the pipelined form: SubstX2::row_begin / row_end)
    } else {
      const uint32_t codes = row_codes<LANES>(chunk_code, q, span);
      sub.set_row(codes);
      if constexpr (LANES == 64) {   // the row's profile (table scorings): see nw_dirs_x2_wave
        if (q == 0) sub.build_profile(codes, j & 1u);
        sub.load_profile(j & 1u);
        if (q != LANES - 1 && j < lb) sub.build_profile(row_codes<LANES>(chunk_code, q + 1, span), (j + 1u) & 1u);
      }
    }
    const pk16 x_ul = pk_shr1_in<LANES>(X[CPL - 1], pk_splat(-16384), first_lane);
    const uint32_t t_ul = dpp_mov0<0x138>(T[CPL - 1]);
    const uint32_t row_pk = j * kBoth;    // (rows < 32 768: the launcher's score bound implies it)
    pk16 mv[CPL], av[CPL], bv[CPL], z[CPL];
    uint32_t dv[CPL], dvA[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const pk16 s = sub.score(c);
      const pk16 xd = c ? X[c - (c ? 1 : 0)] : x_ul;
      const uint32_t td = c ? T[c - (c ? 1 : 0)] : t_ul;
      const pk16 m = pk_max(pk_adds(xd, s), zero);
      const pk16 ae = pk_adds(Ap[c], ext);
      const pk16 a = pk_max(pk_max(pk_adds(Yp[c], open1), ae), zero);
      mv[c] = m; av[c] = a; z[c] = pk_max(m, a);
      if constexpr (LOCAL) {
        // CA (gap_a + ext IS the max), and the states whose score is 0 (0 >= m, 0 >= a: a walk standing there ends)
        dv[c] = put_ge<6>(put_ge<5>(put_ge<2>(0u, pk_subs(ae, a)), pk_subs(zero, m)), pk_subs(zero, a));
      } else {
      const uint32_t opened = pk_lt(ae, a);
      const uint32_t dA = bfi(opened, TY4[c], 4u * kBoth);
      dv[c] = bfi(pos_mask(m), td, 3u * kBoth); dvA[c] = bfi(pos_mask(a), dA, 12u * kBoth);
      }
      // the best cell of my column: a strictly higher score moves it (the first row keeps a tie)
      const pk16 mine = pk_from(pk_bits(m) & valid[c]);
      const uint32_t up = pk_lt(best_s[c], mine);
      best_s[c] = pk_max(best_s[c], mine);
      best_r[c] = bfi(up, row_pk, best_r[c]);
    }
    pk16 Pm[CPL], e;
    {
      const pk16 zin = pk_shr1_zero_in<LANES>(z[CPL - 1], first_lane);
      pk16 P[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const pk16 zl = (c == 0) ? zin : z[c - 1];
        pk16 w = pk_max(pk_adds(zl, c1[c]), c2[c]);
        P[c] = (c == 0) ? w : pk_max(P[c - 1], w);
      }
      e = pk_scan_max_excl<LANES>(P[CPL - 1], first_lane);
#pragma unroll
      for (int c = 0; c < CPL; ++c) { Pm[c] = pk_max(P[c], e); bv[c] = pk_adds(Pm[c], c3[c]); }
    }
    {
      const pk16 al = pk_shr1_zero(av[CPL - 1]);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const pk16 aL = c ? av[c - (c ? 1 : 0)] : al;
        const pk16 b = bv[c];
        const pk16 yn = pk_max(mv[c], b);
        if constexpr (LOCAL) {
          uint32_t d = put_ge<3>(dv[c], pk_subs(pk_adds(aL, open1), b));     // FA
          d = put_ge<4>(d, pk_subs(c ? Pm[c - (c ? 1 : 0)] : e, Pm[c]));      // FB
          d = put_ge<7>(d, pk_subs(zero, b));                                 // gap_b's score is 0
          d = put_ge<1>(d, pk_subs(b, mv[c]));                                // BM
          dv[c] = put_ge<0>(d, pk_subs(av[c], yn));                           // GA
          X[c] = pk_max(z[c], b); Yp[c] = yn; Ap[c] = av[c];
        } else {
        const uint32_t not_a = pk_lt(pk_adds(aL, open1), b);
        const uint32_t not_b = pk_lt(c ? Pm[c - (c ? 1 : 0)] : e, Pm[c]);
        const uint32_t dB = bfi(not_a, bfi(not_b, 0u, 32u * kBoth), 16u * kBoth);
        dv[c] = or3(dv[c], dvA[c], bfi(pos_mask(b), dB, 48u * kBoth));
        const uint32_t m_wins = pk_lt(b, mv[c]);
        const uint32_t a_loses = pk_lt(av[c], yn);
        const uint32_t ty4 = bfi(m_wins, 0u, 8u * kBoth);
        X[c] = pk_max(z[c], b); Yp[c] = yn; Ap[c] = av[c];
        T[c] = bfi(a_loses, ty4 >> 2, kBoth);
        TY4[c] = ty4;
        }
      }
    }
    if constexpr (LANES == 64 && SUBST == SA_SUBST_LDS && kLdsPipe) sub.row_end(chunk_code, q, j, lb);
    append_row(dv);
  }
  flush_rest();

  // the pair's best cell: highest score, then lowest column, then lowest row (per column: the first row above)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int b = 0;
    uint32_t tie = 0;   // (column << 16) | row
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int sc_ = h ? (int)best_s[c].y : (int)best_s[c].x;
      const uint32_t row = h ? best_r[c] >> 16 : best_r[c] & 0xffffu;
      if (sc_ > b) { b = sc_; tie = ((uint32_t)(sl * CPL + c) << 16) | row; }
    }
    unsigned long long key = ((unsigned long long)(uint32_t)b << 32) | (uint32_t)~tie;
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o);
      key = other > key ? other : key;
    }
    if (first_lane && (h ? has_hi : has_lo)) {
      const uint32_t t = ~(uint32_t)key, col = t >> 16, row = t & 0xffffu;
      const int score = (int)(key >> 32);
      const uint32_t pr = h ? pair_hi : pair_lo;
      p.best_score[pr] = score;
      p.best_index[pr] = score > 0 ? (uint64_t)row * W + col : 0;
      p.status[pr] = ~0ull;
    }
  }
}

template <int CPL, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_sw_best_x2_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t tbl_lds = (blockDim.x >> 6) * (2 * R);   // the table sits behind the rings
  load_table_x2<SUBST>(p, tbl_lds);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t unit = blockIdx.x * (blockDim.x >> 6) + wave;
  if (2 * unit >= p.n_pairs) return;
  uint32_t pair0, pair1;
  const bool two = x2_pairs_of(p, unit, &pair0, &pair1);
  sw_best_x2_wave<CPL, SUBST, R, 64, LOCAL>(p, dirs_arena, pair0, pair1, true, two, lane, reinterpret_cast<uint8_t *>(lds) + wave * (2 * R), tbl_lds);
}

template <int CPL, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_sw_best_x4_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t tbl_lds = (blockDim.x >> 6) * (4 * R);
  load_table_x2<SUBST>(p, tbl_lds);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t unit = blockIdx.x * (blockDim.x >> 6) + wave;
  if (4 * unit >= p.n_pairs) return;
  uint32_t lo, hi;
  bool has_lo, has_hi;
  x4_pairs_of(p, unit, lane, &lo, &hi, &has_lo, &has_hi);
  sw_best_x2_wave<CPL, SUBST, R, 32, LOCAL>(p, dirs_arena, lo, hi, has_lo, has_hi, lane, reinterpret_cast<uint8_t *>(lds) + wave * (4 * R), tbl_lds);
}

// Four AND two pairs per wave in one grid (fill_nw_dirs_x4x2_kernel's idea for the best-hit fill): BASELINE configs[2]'s 10 000
// pairs of 151-column rows are 5 000 two-per-wave waves -- some SIMDs five, some four -- or 2 500 four-per-wave ones -- three or
// two; as 2 048 four-per-wave waves (two on every SIMD) + 904 two-per-wave ones the busiest SIMD has the least to do.
template <int CPL4, int SUBST, int R, bool LOCAL>
__global__ void __launch_bounds__(kWave * 4)
fill_sw_best_x4x2_kernel(const SaFillParams p, uint8_t *__restrict__ dirs_arena, const uint32_t q_blocks, const uint32_t pairs_q) {
  constexpr int CPL2 = (CPL4 + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t tbl_lds = waves * (4 * R);
  load_table_x2<SUBST>(p, tbl_lds);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t *ring = reinterpret_cast<uint8_t *>(lds) + wave * (4 * R);
  if (blockIdx.x < q_blocks) {   // (uniform per workgroup)
    const uint32_t unit = blockIdx.x * waves + wave;
    if (4 * unit >= pairs_q) return;
    const uint32_t lo = 4 * unit + 2 * ((uint32_t)lane >> 5);
    sw_best_x2_wave<CPL4, SUBST, R, 32, LOCAL>(p, dirs_arena, lo, lo + 1, true, true, lane, ring, tbl_lds);
  } else {
    const uint32_t pair0 = pairs_q + 2 * ((blockIdx.x - q_blocks) * waves + wave);
    if (pair0 >= p.n_pairs) return;
    const bool two = pair0 + 1 < p.n_pairs;
    sw_best_x2_wave<CPL2, SUBST, R, 64, LOCAL>(p, dirs_arena, pair0, two ? pair0 + 1 : pair0, true, two, lane, ring, tbl_lds);
  }
}

template <int CPL, int R0>
static hipError_t launch_sw_best_x2_cpl(const SaFillParams &p, uint8_t *dirs, hipStream_t stream) {
  constexpr int R = x2_ring(64, CPL, R0);   // (blocked direction bytes: a block row per pair instead of the ring)
  const int wpb = 4;
  const uint32_t units = (p.n_pairs + 1) / 2;
  const dim3 grid((units + wpb - 1) / wpb), block(kWave * wpb);
  if (p.K <= 1 && p.dirs_local) hipLaunchKernelGGL((fill_sw_best_x2_kernel<CPL, SA_SUBST_SIMPLE, R, true>), grid, block, (size_t)wpb * 2 * R, stream, p, dirs);
  else if (p.K <= 1) hipLaunchKernelGGL((fill_sw_best_x2_kernel<CPL, SA_SUBST_SIMPLE, R, false>), grid, block, (size_t)wpb * 2 * R, stream, p, dirs);
  else if (p.dirs_local) hipLaunchKernelGGL((fill_sw_best_x2_kernel<CPL, SA_SUBST_LDS, R, true>), grid, block, (size_t)wpb * 2 * R + table_lds_bytes(p), stream, p, dirs);
  else hipLaunchKernelGGL((fill_sw_best_x2_kernel<CPL, SA_SUBST_LDS, R, false>), grid, block, (size_t)wpb * 2 * R + table_lds_bytes(p), stream, p, dirs);
  return hipGetLastError();
}

template <int CPL, int R>
static hipError_t launch_dirs_x2_cpl(const SaFillParams &p, uint8_t *dirs, hipStream_t stream) {
  const int wpb = 4;
  const uint32_t units = (p.n_pairs + 1) / 2;
  const dim3 grid((units + wpb - 1) / wpb), block(kWave * wpb);
  if (p.K <= 1) hipLaunchKernelGGL((fill_dirs_x2_kernel<CPL, SA_SUBST_SIMPLE, R>), grid, block, (size_t)wpb * 6 * R, stream, p, dirs);
  else hipLaunchKernelGGL((fill_dirs_x2_kernel<CPL, SA_SUBST_LDS, R>), grid, block, (size_t)wpb * 6 * R + table_lds_bytes(p), stream, p, dirs);
  return hipGetLastError();
}

template <int CPL, int R0>
static hipError_t launch_nw_dirs_x2_cpl(const SaFillParams &p, uint8_t *dirs, hipStream_t stream) {
  constexpr int R = x2_ring(64, CPL, R0);   // (blocked direction bytes: a block row per pair instead of the ring)
  const int wpb = 4;
  const uint32_t units = (p.n_pairs + 1) / 2;
  const dim3 grid((units + wpb - 1) / wpb), block(kWave * wpb);
  if (p.K <= 1 && p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_x2_kernel<CPL, SA_SUBST_SIMPLE, R, true>), grid, block, (size_t)wpb * 2 * R, stream, p, dirs);
  else if (p.K <= 1) hipLaunchKernelGGL((fill_nw_dirs_x2_kernel<CPL, SA_SUBST_SIMPLE, R, false>), grid, block, (size_t)wpb * 2 * R, stream, p, dirs);
  else if (p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_x2_kernel<CPL, SA_SUBST_LDS, R, true>), grid, block, (size_t)wpb * 2 * R + table_lds_bytes(p), stream, p, dirs);
  else hipLaunchKernelGGL((fill_nw_dirs_x2_kernel<CPL, SA_SUBST_LDS, R, false>), grid, block, (size_t)wpb * 2 * R + table_lds_bytes(p), stream, p, dirs);
  return hipGetLastError();
}

template <int CPL>
static hipError_t launch_sw_best_x4_cpl(const SaFillParams &p, uint8_t *dirs, hipStream_t stream) {
  constexpr int R = x2_ring(32, CPL, 512);
  const int wpb = 4;
  const uint32_t units = (p.n_pairs + 3) / 4;
  const dim3 grid((units + wpb - 1) / wpb), block(kWave * wpb);
  if (p.K <= 1 && p.dirs_local) hipLaunchKernelGGL((fill_sw_best_x4_kernel<CPL, SA_SUBST_SIMPLE, R, true>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs);
  else if (p.K <= 1) hipLaunchKernelGGL((fill_sw_best_x4_kernel<CPL, SA_SUBST_SIMPLE, R, false>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs);
  else if (p.dirs_local) hipLaunchKernelGGL((fill_sw_best_x4_kernel<CPL, SA_SUBST_LDS, R, true>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs);
  else hipLaunchKernelGGL((fill_sw_best_x4_kernel<CPL, SA_SUBST_LDS, R, false>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs);
  return hipGetLastError();
}

template <int CPL>
static hipError_t launch_nw_dirs_x4_cpl(const SaFillParams &p, uint8_t *dirs, hipStream_t stream) {
  constexpr int R = x2_ring(32, CPL, 512);   // (255 + 32 * 8 columns fit)
  const int wpb = 4;
  const uint32_t units = (p.n_pairs + 3) / 4;
  const dim3 grid((units + wpb - 1) / wpb), block(kWave * wpb);
  if (p.K <= 1 && p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_x4_kernel<CPL, SA_SUBST_SIMPLE, R, true>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs);
  else if (p.K <= 1) hipLaunchKernelGGL((fill_nw_dirs_x4_kernel<CPL, SA_SUBST_SIMPLE, R, false>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs);
  else if (p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_x4_kernel<CPL, SA_SUBST_LDS, R, true>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs);
  else hipLaunchKernelGGL((fill_nw_dirs_x4_kernel<CPL, SA_SUBST_LDS, R, false>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs);
  return hipGetLastError();
}

template <int CPL4>
static hipError_t launch_nw_dirs_x4x2_cpl(const SaFillParams &p, uint8_t *dirs, uint32_t pairs_q, hipStream_t stream) {
  // (the four-per-wave waves: four pairs of 32 x CPL4 columns; the two-per-wave waves: two pairs of 64 x CPL2 in the same region)
  constexpr int R = x2_ring(64, (CPL4 + 1) / 2, x2_ring(32, CPL4, 512));
  const int wpb = 4;
  const uint32_t q_blocks = (pairs_q / 4 + wpb - 1) / wpb, x2_blocks = ((p.n_pairs - pairs_q + 1) / 2 + wpb - 1) / wpb;
  const dim3 grid(q_blocks + x2_blocks), block(kWave * wpb);
  if (p.K <= 1 && p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_x4x2_kernel<CPL4, SA_SUBST_SIMPLE, R, true>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs, q_blocks, pairs_q);
  else if (p.K <= 1) hipLaunchKernelGGL((fill_nw_dirs_x4x2_kernel<CPL4, SA_SUBST_SIMPLE, R, false>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs, q_blocks, pairs_q);
  else if (p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_x4x2_kernel<CPL4, SA_SUBST_LDS, R, true>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs, q_blocks, pairs_q);
  else hipLaunchKernelGGL((fill_nw_dirs_x4x2_kernel<CPL4, SA_SUBST_LDS, R, false>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs, q_blocks, pairs_q);
  return hipGetLastError();
}

template <int CPL4>
static hipError_t launch_sw_best_x4x2_cpl(const SaFillParams &p, uint8_t *dirs, uint32_t pairs_q, hipStream_t stream) {
  constexpr int R = x2_ring(64, (CPL4 + 1) / 2, x2_ring(32, CPL4, 512));
  const int wpb = 4;
  const uint32_t q_blocks = (pairs_q / 4 + wpb - 1) / wpb, x2_blocks = ((p.n_pairs - pairs_q + 1) / 2 + wpb - 1) / wpb;
  const dim3 grid(q_blocks + x2_blocks), block(kWave * wpb);
  if (p.K <= 1 && p.dirs_local) hipLaunchKernelGGL((fill_sw_best_x4x2_kernel<CPL4, SA_SUBST_SIMPLE, R, true>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs, q_blocks, pairs_q);
  else if (p.K <= 1) hipLaunchKernelGGL((fill_sw_best_x4x2_kernel<CPL4, SA_SUBST_SIMPLE, R, false>), grid, block, (size_t)wpb * 4 * R, stream, p, dirs, q_blocks, pairs_q);
  else if (p.dirs_local) hipLaunchKernelGGL((fill_sw_best_x4x2_kernel<CPL4, SA_SUBST_LDS, R, true>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs, q_blocks, pairs_q);
  else hipLaunchKernelGGL((fill_sw_best_x4x2_kernel<CPL4, SA_SUBST_LDS, R, false>), grid, block, (size_t)wpb * 4 * R + table_lds_bytes(p), stream, p, dirs, q_blocks, pairs_q);
  return hipGetLastError();
}

template <int CPL, int R0>
static hipError_t launch_nw_dirs_mixed_cpl(const SaFillParams &p, uint8_t *dirs, uint32_t n_modal, uint32_t n_rest, hipStream_t stream) {
  constexpr int R = x2_ring(64, CPL, R0);
  const int wpb = 4;
  const uint32_t x2_blocks = ((n_modal + 1) / 2 + wpb - 1) / wpb, x1_blocks = (n_rest + wpb - 1) / wpb;
  const dim3 grid(x2_blocks + x1_blocks), block(kWave * wpb);
  if (p.K <= 1) {
    if (p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_mixed_kernel<CPL, SA_SUBST_SIMPLE, R, true>), grid, block, (size_t)wpb * 2 * R, stream, p, dirs, x2_blocks, n_modal, n_rest);
    else hipLaunchKernelGGL((fill_nw_dirs_mixed_kernel<CPL, SA_SUBST_SIMPLE, R, false>), grid, block, (size_t)wpb * 2 * R, stream, p, dirs, x2_blocks, n_modal, n_rest);
  } else {
    const size_t lds = std::max((size_t)wpb * 2 * R + table_lds_bytes(p), (size_t)wpb * R + (((size_t)p.K * p.K + 3u) & ~(size_t)3u) * sizeof(int32_t));
    if (p.dirs_local) hipLaunchKernelGGL((fill_nw_dirs_mixed_kernel<CPL, SA_SUBST_LDS, R, true>), grid, block, lds, stream, p, dirs, x2_blocks, n_modal, n_rest);
    else hipLaunchKernelGGL((fill_nw_dirs_mixed_kernel<CPL, SA_SUBST_LDS, R, false>), grid, block, lds, stream, p, dirs, x2_blocks, n_modal, n_rest);
  }
  return hipGetLastError();
}

}  // namespace sa

// every score the recurrence can produce for pairs up to max_len_a x max_len_b, de-trended or not, stays inside int16
bool sa_x2_scores_fit(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b) {
  return sa_domain_x2_scores_fit(sa_traits_of(p), max_len_a, max_len_b);
}

bool sa_nw_dirs_x2_applicable(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b, const uint8_t *dirs) {
  if (!sa_nw_dirs_fill_applicable(p, max_len_a, dirs)) return false;
  if (p.K > SA_LDS_TABLE_MAX_K || p.uniform_stride == 0 || (p.uniform_stride & 255u)) return false;
  return sa_domain_nw_x2_scores_fit(sa_traits_of(p), max_len_a, max_len_b);   // (the NW fills' values are de-trended: their own bound)
}

// Four pairs per wave (32 lanes a couple) instead of two: every pair of the launch one shape (no pair list), rows up to 192
// columns (six columns per lane of a span: beyond that the registers cost more than the lanes save), and pairs enough that
// half as many waves still fill the chip (min_pairs: NW 4 097 -- round 4 measured 3-7 % faster from 4 096 to 40 000 pairs of 150 x 150; round 6:
// exactly one round is 9 % slower than two rounds of two-per-wave waves, more than one is ahead;
// SW best hit 4 097 -- round 4 measured 150 x 1000 level at 10 000 pairs, where 2 500 waves leave some SIMDs with three and some
// with two, and 8-9 % ahead from 16 000 on (profiles/r04/r04_quad_fills.txt); round 6 by whole rounds: ahead from more than one round (4 096 pairs) on, and
// the short rest two per wave in the same grid, sa_launch_fill_sw_best_x2) -- or whenever the shape allows (option quad = 2: tests), or never
// (quad = 1).  Returns the columns per lane of a span, 0 = two pairs per wave.
static int sa_x4_columns(const SaFillParams &p, uint32_t max_len_a, uint32_t min_pairs) {
  if (p.pair_list || p.tune_quad == 1 || p.tune_cpl) return 0;
  const uint32_t need = (max_len_a + 1 + 31u) / 32u;
  if (need > 6) return 0;
  if (p.tune_quad != 2 && p.n_pairs < min_pairs) return 0;
  return (int)need;
}

hipError_t sa_launch_fill_nw_dirs_x2(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  // (from MORE than one whole round of four-per-wave waves: exactly 4 096 pairs are one wave per SIMD four per wave, two per SIMD two per
  //  wave -- 96.7 against 87.8 us, 150 x 150; 5 000: 108.6 mixed / 119.1 two per wave, 8 192: 141.6 / 155.0, 16 384: 248.7 / 287.7:
  //  profiles/r06/r06_local_dirs.txt)
  if (const int c4 = sa_x4_columns(p, max_len_a, 4097u)) {
    // the last round of four-per-wave waves less than half full (and the choice left to the library): those pairs two per wave
    const uint32_t pairs_q = p.n_pairs / 4096u * 4096u, rest = p.n_pairs - pairs_q;
    if (p.tune_quad == 0 && pairs_q && rest && rest <= 2048u) {
      sa_record_launch(SEQALIGN_K_FILL_NW_DIRS_X4, pairs_q);
      sa_record_launch(SEQALIGN_K_FILL_NW_DIRS_X2, rest);
      switch (c4) {
        case 1: return sa::launch_nw_dirs_x4x2_cpl<1>(p, dirs, pairs_q, stream);
        case 2: return sa::launch_nw_dirs_x4x2_cpl<2>(p, dirs, pairs_q, stream);
        case 3: return sa::launch_nw_dirs_x4x2_cpl<3>(p, dirs, pairs_q, stream);
        case 4: return sa::launch_nw_dirs_x4x2_cpl<4>(p, dirs, pairs_q, stream);
        case 5: return sa::launch_nw_dirs_x4x2_cpl<5>(p, dirs, pairs_q, stream);
        default: return sa::launch_nw_dirs_x4x2_cpl<6>(p, dirs, pairs_q, stream);
      }
    }
    sa_record_launch(SEQALIGN_K_FILL_NW_DIRS_X4, p.n_pairs);
    switch (c4) {
      case 1: return sa::launch_nw_dirs_x4_cpl<1>(p, dirs, stream);
      case 2: return sa::launch_nw_dirs_x4_cpl<2>(p, dirs, stream);
      case 3: return sa::launch_nw_dirs_x4_cpl<3>(p, dirs, stream);
      case 4: return sa::launch_nw_dirs_x4_cpl<4>(p, dirs, stream);
      case 5: return sa::launch_nw_dirs_x4_cpl<5>(p, dirs, stream);
      default: return sa::launch_nw_dirs_x4_cpl<6>(p, dirs, stream);
    }
  }
  sa_record_launch(SEQALIGN_K_FILL_NW_DIRS_X2, p.n_pairs);
  // (blocked direction bytes are decided by the row's width -- host -- and by LANES x CPL <= 512 -- kernel: a forced wider CPL must not split them)
  const uint32_t need = sa::columns_per_lane(max_len_a + 1, sa_dirs_blocked_shape(max_len_a) ? std::min<uint32_t>(p.tune_cpl, 8u) : p.tune_cpl);
  if (need <= 1) return sa::launch_nw_dirs_x2_cpl<1, 512>(p, dirs, stream);
  if (need <= 2) return sa::launch_nw_dirs_x2_cpl<2, 512>(p, dirs, stream);
  if (need <= 3) return sa::launch_nw_dirs_x2_cpl<3, 512>(p, dirs, stream);
  if (need <= 4) return sa::launch_nw_dirs_x2_cpl<4, 512>(p, dirs, stream);
  if (need <= 5) return sa::launch_nw_dirs_x2_cpl<5, 1024>(p, dirs, stream);
  if (need <= 6) return sa::launch_nw_dirs_x2_cpl<6, 1024>(p, dirs, stream);
  if (need <= 8) return sa::launch_nw_dirs_x2_cpl<8, 1024>(p, dirs, stream);
  if (need <= 12) return sa::launch_nw_dirs_x2_cpl<12, 1024>(p, dirs, stream);   // (rows of 513 .. 1 024 columns: round 5)
  return sa::launch_nw_dirs_x2_cpl<16, 2048>(p, dirs, stream);
}

// ---- Smith-Waterman multi-hit: match_scores + directions, two pairs per wave
bool sa_dirs_x2_applicable(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b, const uint8_t *dirs) {
  if (!sa_dirs_fill_applicable(p, max_len_a, dirs)) return false;
  if (p.K > SA_LDS_TABLE_MAX_K || p.uniform_stride == 0 || (p.uniform_stride & 255u)) return false;
  return sa_x2_scores_fit(p, max_len_a, max_len_b);
}

hipError_t sa_launch_fill_dirs_x2(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  sa_record_launch(SEQALIGN_K_FILL_SW_DIRS_X2, p.n_pairs);
  const uint32_t need = sa::columns_per_lane(max_len_a + 1, p.tune_cpl);
  if (need <= 1) return sa::launch_dirs_x2_cpl<1, 512>(p, dirs, stream);
  if (need <= 2) return sa::launch_dirs_x2_cpl<2, 512>(p, dirs, stream);
  if (need <= 3) return sa::launch_dirs_x2_cpl<3, 512>(p, dirs, stream);
  if (need <= 4) return sa::launch_dirs_x2_cpl<4, 512>(p, dirs, stream);
  if (need <= 5) return sa::launch_dirs_x2_cpl<5, 1024>(p, dirs, stream);
  if (need <= 6) return sa::launch_dirs_x2_cpl<6, 1024>(p, dirs, stream);
  if (need <= 8) return sa::launch_dirs_x2_cpl<8, 1024>(p, dirs, stream);
  if (need <= 12) return sa::launch_dirs_x2_cpl<12, 1024>(p, dirs, stream);   // (rows of 513 .. 1 024 columns: round 5)
  return sa::launch_dirs_x2_cpl<16, 2048>(p, dirs, stream);
}

// ---- Smith-Waterman best hit: directions + the best cell, two pairs per wave
bool sa_sw_best_x2_applicable(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b, const uint8_t *dirs) {
  // the domain of the direction fills (sa_dirs_fill_applicable) without the candidates' outputs
  if (!sa_domain_sw_best_x2(sa_traits_of(p), max_len_a, max_len_b)) return false;
  if (!dirs || ((uintptr_t)dirs & 255) || !p.best_score || !p.best_index) return false;
  return p.uniform_stride != 0 && (p.uniform_stride & 255u) == 0;
}

hipError_t sa_launch_fill_sw_best_x2(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  // Four pairs per wave from MORE than one whole round of four-per-wave waves on (4 097 pairs; round 6 -- 16 384 before): the cost is a
  // step function of whole rounds of resident waves.  By batch size, 150 x 1 000, two / four per wave, us (profiles/r06/r06_local_dirs.txt):
  // 4 096 pairs 746 / 797; 7 000: 1 343 / 1 185; 8 192: 1 346 / 1 206; 11 000: 1 942 / 1 685; 12 288: 1 978 / 1 695; 16 384: 2 613 / 2 190.  A rest of
  // less than half a round of four-per-wave waves goes two per wave in the same grid (fill_sw_best_x4x2_kernel: 5 000 pairs 1 032 -> 893,
  // 6 144: 1 048 -> 914, 10 000: 1 643 / 1 680 -> 1 393, 14 000: 2 282 / 2 168 -> 1 886) when the choice is left to the library.
  if (const int c4 = sa_x4_columns(p, max_len_a, SA_BEST_X4_MIN)) {
    const uint32_t pairs_q = p.n_pairs / 4096u * 4096u, rest = p.n_pairs - pairs_q;
    if (p.tune_quad == 0 && pairs_q && rest && rest <= 2048u) {
      sa_record_launch(SEQALIGN_K_FILL_SW_BEST_X4, pairs_q);
      sa_record_launch(SEQALIGN_K_FILL_SW_BEST_X2, rest);
      switch (c4) {
        case 1: return sa::launch_sw_best_x4x2_cpl<1>(p, dirs, pairs_q, stream);
        case 2: return sa::launch_sw_best_x4x2_cpl<2>(p, dirs, pairs_q, stream);
        case 3: return sa::launch_sw_best_x4x2_cpl<3>(p, dirs, pairs_q, stream);
        case 4: return sa::launch_sw_best_x4x2_cpl<4>(p, dirs, pairs_q, stream);
        case 5: return sa::launch_sw_best_x4x2_cpl<5>(p, dirs, pairs_q, stream);
        default: return sa::launch_sw_best_x4x2_cpl<6>(p, dirs, pairs_q, stream);
      }
    }
    sa_record_launch(SEQALIGN_K_FILL_SW_BEST_X4, p.n_pairs);
    switch (c4) {
      case 1: return sa::launch_sw_best_x4_cpl<1>(p, dirs, stream);
      case 2: return sa::launch_sw_best_x4_cpl<2>(p, dirs, stream);
      case 3: return sa::launch_sw_best_x4_cpl<3>(p, dirs, stream);
      case 4: return sa::launch_sw_best_x4_cpl<4>(p, dirs, stream);
      case 5: return sa::launch_sw_best_x4_cpl<5>(p, dirs, stream);
      default: return sa::launch_sw_best_x4_cpl<6>(p, dirs, stream);
    }
  }
  sa_record_launch(SEQALIGN_K_FILL_SW_BEST_X2, p.n_pairs);
  // (blocked direction bytes are decided by the row's width -- host -- and by LANES x CPL <= 512 -- kernel: a forced wider CPL must not split them)
  const uint32_t need = sa::columns_per_lane(max_len_a + 1, sa_dirs_blocked_shape(max_len_a) ? std::min<uint32_t>(p.tune_cpl, 8u) : p.tune_cpl);
  if (need <= 1) return sa::launch_sw_best_x2_cpl<1, 512>(p, dirs, stream);
  if (need <= 2) return sa::launch_sw_best_x2_cpl<2, 512>(p, dirs, stream);
  if (need <= 3) return sa::launch_sw_best_x2_cpl<3, 512>(p, dirs, stream);
  if (need <= 4) return sa::launch_sw_best_x2_cpl<4, 512>(p, dirs, stream);
  if (need <= 5) return sa::launch_sw_best_x2_cpl<5, 1024>(p, dirs, stream);
  if (need <= 6) return sa::launch_sw_best_x2_cpl<6, 1024>(p, dirs, stream);
  if (need <= 8) return sa::launch_sw_best_x2_cpl<8, 1024>(p, dirs, stream);
  if (need <= 12) return sa::launch_sw_best_x2_cpl<12, 1024>(p, dirs, stream);   // (rows of 513 .. 1 024 columns: round 5)
  return sa::launch_sw_best_x2_cpl<16, 2048>(p, dirs, stream);
}

// ---- NW, a chunk whose pairs are mostly of one shape: both kinds of waves in one grid (p.pair_list: modal pairs, then the rest)
hipError_t sa_launch_fill_nw_dirs_mixed(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, uint32_t n_modal, uint32_t n_rest,
                                        hipStream_t stream) {
  if (n_modal + n_rest == 0) return hipSuccess;
  if (!p.pair_list) return hipErrorInvalidValue;
  // one grid, two kinds of waves: the modal shape's pairs two per wave, the rest one per wave
  if (n_modal) sa_record_launch(SEQALIGN_K_FILL_NW_DIRS_X2, n_modal);
  if (n_rest) sa_record_launch(SEQALIGN_K_FILL_NW_DIRS, n_rest);
  // (blocked direction bytes are decided by the row's width -- host -- and by LANES x CPL <= 512 -- kernel: a forced wider CPL must not split them)
  const uint32_t need = sa::columns_per_lane(max_len_a + 1, sa_dirs_blocked_shape(max_len_a) ? std::min<uint32_t>(p.tune_cpl, 8u) : p.tune_cpl);   // (of the widest pair: the packed waves take it too)
  if (need <= 1) return sa::launch_nw_dirs_mixed_cpl<1, 512>(p, dirs, n_modal, n_rest, stream);
  if (need <= 2) return sa::launch_nw_dirs_mixed_cpl<2, 512>(p, dirs, n_modal, n_rest, stream);
  if (need <= 3) return sa::launch_nw_dirs_mixed_cpl<3, 512>(p, dirs, n_modal, n_rest, stream);
  if (need <= 4) return sa::launch_nw_dirs_mixed_cpl<4, 512>(p, dirs, n_modal, n_rest, stream);
  if (need <= 5) return sa::launch_nw_dirs_mixed_cpl<5, 1024>(p, dirs, n_modal, n_rest, stream);
  if (need <= 6) return sa::launch_nw_dirs_mixed_cpl<6, 1024>(p, dirs, n_modal, n_rest, stream);
  if (need <= 8) return sa::launch_nw_dirs_mixed_cpl<8, 1024>(p, dirs, n_modal, n_rest, stream);
  if (need <= 12) return sa::launch_nw_dirs_mixed_cpl<12, 1024>(p, dirs, n_modal, n_rest, stream);
  return sa::launch_nw_dirs_mixed_cpl<16, 2048>(p, dirs, n_modal, n_rest, stream);
}
