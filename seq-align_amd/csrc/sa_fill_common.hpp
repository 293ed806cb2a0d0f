// sa_fill_common.hpp -- device helpers shared by the gfx950 fill kernels.
//
// CDNA4 notes (guides: cdna_hip_programming.md, MI355X_MICROARCH.md):
//  * a wavefront is 64 lanes; one pair is owned by ONE wave, so the intra-pair
//    dependency never crosses a wave and needs no barrier and no LDS;
//  * the one-lane hand-off uses the DPP "wave_shr:1" full-wave shift
//    (v_mov_b32_dpp, a VALU op) instead of __shfl_up, which lowers to
//    ds_bpermute_b32 through the LDS crossbar;
//  * all arithmetic is int32 VALU (v_max3_i32, v_add) -- nothing GEMM-shaped, so
//    MFMA is not used; the kernels are bound by HBM writes (12 B per cell).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sa_kernels.h"

// (SA_S_* sentinels and SA_F_* flags: host/sa_internal.h, through sa_kernels.h)

namespace sa {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;   // 4 pairs per 256-thread workgroup
// s_waitcnt vmcnt(0) only (gfx9 encoding: vmcnt [3:0]+[15:14], expcnt [6:4]=7,
// lgkmcnt [11:8]=15 left at "don't wait")
constexpr int kWaitVm0 = 0x0F70;

// wrap-around add: identical to the reference's int add wherever that one is
// defined (no overflow inside the parity domain), and no UB here otherwise
__device__ __forceinline__ int addw(int a, int b) {
  return (int)((unsigned)a + (unsigned)b);
}
__device__ __forceinline__ int max3i(int a, int b, int c) {
  return max(max(a, b), c);   // -> v_max3_i32
}

// Full-wave shift right by one lane: lane l receives src of lane l-1, lane 0
// receives lane0_value.  DPP ctrl 0x138 = wave_shr:1 (GFX9 family).
__device__ __forceinline__ int wave_shr1(int src, int lane0_value) {
  return __builtin_amdgcn_update_dpp(lane0_value, src, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int read_lane(int v, int lane_uniform) {
  return __builtin_amdgcn_readlane(v, lane_uniform);
}

// substitution score of (code_a, code_b); code = folded char | class << 8.
// arow = class(a) * K, precomputed per column.
template <int SUBST>
__device__ __forceinline__ int subst_score(int fa, int arow, int code_b,
                                           const int32_t *table, int gen_eq,
                                           int gen_ne) {
  const int fb = code_b & 0xff;
  if constexpr (SUBST == SA_SUBST_SIMPLE) {
    return fa == fb ? gen_eq : gen_ne;
  } else {
    const int kb = code_b >> 8;
    int s = table[arow + kb];
    // class 0 x class 0: the table holds the "equal" score
    if ((arow | kb) == 0 && fa != fb) s = gen_ne;
    return s;
  }
}

// unaligned-capable multi-dword stores (rows are 4-byte aligned only: the
// reference layout has pitch len_a+1 ints, which cannot be padded)
typedef int v2i_u __attribute__((ext_vector_type(2), aligned(4)));
typedef int v3i_u __attribute__((ext_vector_type(3), aligned(4)));
typedef int v4i_u __attribute__((ext_vector_type(4), aligned(4)));

// NT = non-temporal hint (global_store ... nt).  Measured on C2 (10k 150x150,
// same box, profiles/r01_variants.txt): row-sweep kernels gain 10-20 % with it
// (the matrices are write-once streams, nothing re-reads them from L2), the
// wavefront kernel loses 4x (its 12-B per-row pieces NEED to merge in L2).
// (a macro, not a template: the pointer must keep its aligned(4) typedef)
#define SA_STORE_VEC(NT, ptr, val)                                   \
  do {                                                               \
    if constexpr (NT) __builtin_nontemporal_store((val), (ptr));     \
    else *(ptr) = (val);                                             \
  } while (0)

template <int N, bool NT = false>
__device__ __forceinline__ void store_run(int32_t *dst, const int (&v)[N]) {
  if constexpr (N == 1) {
    SA_STORE_VEC(NT, dst, (int32_t)v[0]);
  } else if constexpr (N == 2) {
    SA_STORE_VEC(NT, reinterpret_cast<v2i_u *>(dst), (v2i_u{v[0], v[1]}));
  } else if constexpr (N == 3) {
    SA_STORE_VEC(NT, reinterpret_cast<v3i_u *>(dst), (v3i_u{v[0], v[1], v[2]}));
  } else if constexpr (N == 4) {
    SA_STORE_VEC(NT, reinterpret_cast<v4i_u *>(dst), (v4i_u{v[0], v[1], v[2], v[3]}));
  } else {
    SA_STORE_VEC(NT, reinterpret_cast<v4i_u *>(dst), (v4i_u{v[0], v[1], v[2], v[3]}));
    int rest[N - 4];
#pragma unroll
    for (int k = 0; k < N - 4; ++k) rest[k] = v[4 + k];
    store_run<N - 4, NT>(dst + 4, rest);
  }
}

template <int N>
__device__ __forceinline__ void store_partial(int32_t *dst, const int (&v)[N], int n) {
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k < n) dst[k] = v[k];
}

// columns per lane for a batch whose longest seq_a is max_len_a; `at_least`
// (the context's option "cpl", tuning experiments) may raise it
inline uint32_t columns_per_lane(uint32_t max_len_a, uint32_t at_least = 0) {
  uint32_t need = (max_len_a + kWave - 1) / kWave;
  if (at_least > need && at_least <= 16) need = at_least;
  return need;
}

// border values (reference alignment.c:46-81)
struct Border {
  int floor, gap_open, ext;
  bool is_sw, no_start;
  // gap_b_scores[i] on row 0 / gap_a_scores[j*W] on column 0, index k >= 1
  __device__ __forceinline__ int edge_gap(unsigned k) const {
    return (is_sw || no_start) ? 0 : addw(gap_open, (int)k * ext);
  }
  __device__ __forceinline__ int edge_floor() const { return is_sw ? 0 : floor; }
};

}  // namespace sa
