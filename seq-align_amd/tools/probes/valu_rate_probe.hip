// valu_rate_probe.hip -- issue rate of the VALU instructions the packed direction fills are made of (sa_fill_dirs_x2.hip):
// cycles per wave64 instruction per SIMD at full occupancy, next to plain 32-bit adds.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define BODY(name, text)                                                                                           \
  __global__ void __launch_bounds__(256) k_##name(uint32_t *out, int iters) {                                      \
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = 0x00010001u, d = threadIdx.x ^ 0x55u;                 \
    for (int i = 0; i < iters; ++i) {                                                                              \
      asm volatile(REP8(REP8(text)) : "+v"(a), "+v"(b), "+v"(d) : "v"(c) : "vcc", "s20", "s21", "s22", "s23");                                         \
    }                                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + d;                                                        \
  }

// 4 instructions per text (independent registers a, b, d so nothing waits on a result)
BODY(add_u32, "v_add_u32 %0, %0, %3\n v_add_u32 %1, %1, %3\n v_add_u32 %2, %2, %3\n v_add_u32 %0, %0, %3\n")
BODY(max_i32, "v_max_i32 %0, %0, %3\n v_max_i32 %1, %1, %3\n v_max_i32 %2, %2, %3\n v_max_i32 %0, %0, %3\n")
BODY(max3_i32, "v_max3_i32 %0, %0, %3, %1\n v_max3_i32 %1, %1, %3, %2\n v_max3_i32 %2, %2, %3, %0\n v_max3_i32 %0, %0, %3, %1\n")
BODY(pk_add_i16, "v_pk_add_i16 %0, %0, %3 clamp\n v_pk_add_i16 %1, %1, %3 clamp\n v_pk_add_i16 %2, %2, %3 clamp\n v_pk_add_i16 %0, %0, %3 clamp\n")
BODY(pk_sub_i16, "v_pk_sub_i16 %0, %0, %3 clamp\n v_pk_sub_i16 %1, %1, %3 clamp\n v_pk_sub_i16 %2, %2, %3 clamp\n v_pk_sub_i16 %0, %0, %3 clamp\n")
BODY(pk_max_i16, "v_pk_max_i16 %0, %0, %3\n v_pk_max_i16 %1, %1, %3\n v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %0, %0, %3\n")
BODY(pk_max_u16, "v_pk_max_u16 %0, %0, %3\n v_pk_max_u16 %1, %1, %3\n v_pk_max_u16 %2, %2, %3\n v_pk_max_u16 %0, %0, %3\n")
BODY(pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %1, 15, %1 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %2, 15, %2 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]\n")
BODY(pk_mad_i16, "v_pk_mad_i16 %0, %0, %3, %1\n v_pk_mad_i16 %1, %1, %3, %2\n v_pk_mad_i16 %2, %2, %3, %0\n v_pk_mad_i16 %0, %0, %3, %1\n")
BODY(bfi_b32, "v_bfi_b32 %0, %0, %3, %1\n v_bfi_b32 %1, %1, %3, %2\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %0, %0, %3, %1\n")
BODY(bitop3, "v_bitop3_b32 %0, %0, %3, %1 bitop3:0x6c\n v_bitop3_b32 %1, %1, %3, %2 bitop3:0x6c\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x6c\n v_bitop3_b32 %0, %0, %3, %1 bitop3:0x6c\n")
BODY(and_or, "v_and_or_b32 %0, %0, %3, %1\n v_and_or_b32 %1, %1, %3, %2\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %0, %0, %3, %1\n")
BODY(xor_b32, "v_xor_b32 %0, %0, %3\n v_xor_b32 %1, %1, %3\n v_xor_b32 %2, %2, %3\n v_xor_b32 %0, %0, %3\n")
BODY(cndmask, "v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %0, %0, %3, vcc\n")
BODY(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
BODY(max_dpp, "v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
BODY(mov_wave_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
BODY(pk_lshr_b16, "v_pk_lshrrev_b16 %0, 10, %0 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %1, 10, %1 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %2, 10, %2 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %0, 10, %0 op_sel_hi:[0,1]\n")
BODY(pk_min_u16, "v_pk_min_u16 %0, %0, %3\n v_pk_min_u16 %1, %1, %3\n v_pk_min_u16 %2, %2, %3\n v_pk_min_u16 %0, %0, %3\n")

BODY(and_b32, "v_and_b32 %0, %0, %3\n v_and_b32 %1, %1, %3\n v_and_b32 %2, %2, %3\n v_and_b32 %0, %0, %3\n ")
BODY(or_b32, "v_or_b32 %0, %0, %3\n v_or_b32 %1, %1, %3\n v_or_b32 %2, %2, %3\n v_or_b32 %0, %0, %3\n ")
BODY(sub_u32, "v_sub_u32 %0, %0, %3\n v_sub_u32 %1, %1, %3\n v_sub_u32 %2, %2, %3\n v_sub_u32 %0, %0, %3\n ")
BODY(lshlrev, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %0, 1, %0\n ")
BODY(lshrrev, "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %0, 1, %0\n ")
BODY(ashrrev, "v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %0, 1, %0\n ")
BODY(mov_b32, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %0\n v_mov_b32 %0, %1\n ")
BODY(or3, "v_or3_b32 %0, %0, %3, %1\n v_or3_b32 %1, %1, %3, %2\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %0, %0, %3, %1\n ")
BODY(add3, "v_add3_u32 %0, %0, %3, %1\n v_add3_u32 %1, %1, %3, %2\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %0, %0, %3, %1\n ")
BODY(lshl_or, "v_lshl_or_b32 %0, %0, 2, %1\n v_lshl_or_b32 %1, %1, 2, %2\n v_lshl_or_b32 %2, %2, 2, %0\n v_lshl_or_b32 %0, %0, 2, %1\n ")
BODY(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %1, %1, 2, %2\n v_lshl_add_u32 %2, %2, 2, %0\n v_lshl_add_u32 %0, %0, 2, %1\n ")
BODY(min_i32, "v_min_i32 %0, %0, %3\n v_min_i32 %1, %1, %3\n v_min_i32 %2, %2, %3\n v_min_i32 %0, %0, %3\n ")
BODY(max_u32, "v_max_u32 %0, %0, %3\n v_max_u32 %1, %1, %3\n v_max_u32 %2, %2, %3\n v_max_u32 %0, %0, %3\n ")
BODY(cndmask_s, "v_cndmask_b32_e64 %0, %0, %3, s[20:21]\n v_cndmask_b32_e64 %1, %1, %3, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %0, %0, %3, s[20:21]\n ")
BODY(cmp_lt, "v_cmp_lt_i32_e64 s[22:23], %0, %3\n v_cmp_lt_i32_e64 s[22:23], %1, %3\n v_cmp_lt_i32_e64 s[22:23], %2, %3\n v_cmp_lt_i32_e64 s[22:23], %0, %3\n ")
BODY(perm, "v_perm_b32 %0, %0, %3, %1\n v_perm_b32 %1, %1, %3, %2\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %0, %0, %3, %1\n ")
BODY(add_u16, "v_add_u16 %0, %0, %3\n v_add_u16 %1, %1, %3\n v_add_u16 %2, %2, %3\n v_add_u16 %0, %0, %3\n ")
BODY(max_i16, "v_max_i16 %0, %0, %3\n v_max_i16 %1, %1, %3\n v_max_i16 %2, %2, %3\n v_max_i16 %0, %0, %3\n ")
BODY(pk_add_u16, "v_pk_add_u16 %0, %0, %3\n v_pk_add_u16 %1, %1, %3\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %0, %0, %3\n ")
BODY(sub_co, "v_sub_co_u32 %0, vcc, %0, %3\n v_sub_co_u32 %1, vcc, %1, %3\n v_sub_co_u32 %2, vcc, %2, %3\n v_sub_co_u32 %0, vcc, %0, %3\n ")
BODY(bfe_i32, "v_bfe_i32 %0, %0, 0, 16\n v_bfe_i32 %1, %1, 0, 16\n v_bfe_i32 %2, %2, 0, 16\n v_bfe_i32 %0, %0, 0, 16\n ")
BODY(alignbit, "v_alignbit_b32 %0, %0, %1, 16\n v_alignbit_b32 %1, %1, %2, 16\n v_alignbit_b32 %2, %2, %0, 16\n v_alignbit_b32 %0, %0, %1, 16\n ")
BODY(xnor, "v_xnor_b32 %0, %0, %3\n v_xnor_b32 %1, %1, %3\n v_xnor_b32 %2, %2, %3\n v_xnor_b32 %0, %0, %3\n ")
BODY(med3, "v_med3_i32 %0, %0, %3, %1\n v_med3_i32 %1, %1, %3, %2\n v_med3_i32 %2, %2, %3, %0\n v_med3_i32 %0, %0, %3, %1\n ")
BODY(mad_u32_u24, "v_mad_u32_u24 %0, %0, %3, %1\n v_mad_u32_u24 %1, %1, %3, %2\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %0, %0, %3, %1\n ")
BODY(sad, "v_sad_u32 %0, %0, %3, %1\n v_sad_u32 %1, %1, %3, %2\n v_sad_u32 %2, %2, %3, %0\n v_sad_u32 %0, %0, %3, %1\n ")

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  uint32_t *out;
  const int blocks = cus * 8;          // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%d CUs, %.2f GHz (clockRate): cycles per wave64 instruction per SIMD, 8 waves per SIMD\n", cus, ghz);
  const int iters = 2000;
  struct { const char *name; void (*k)(uint32_t *, int); } ks[] = {
#define K(n) {#n, k_##n}
      K(add_u32), K(max_i32), K(max3_i32), K(pk_add_i16), K(pk_sub_i16), K(pk_max_i16), K(pk_max_u16), K(pk_ashr_i16), K(pk_lshr_b16),
      K(pk_min_u16), K(pk_mad_i16), K(bfi_b32), K(bitop3), K(and_or), K(xor_b32), K(cndmask), K(mov_dpp), K(max_dpp), K(mov_wave_shr), K(and_b32), K(or_b32), K(sub_u32), K(lshlrev), K(lshrrev), K(ashrrev), K(mov_b32), K(or3), K(add3), K(lshl_or), K(lshl_add), K(min_i32), K(max_u32), K(cndmask_s), K(cmp_lt), K(perm), K(add_u16), K(max_i16), K(pk_add_u16), K(sub_co), K(bfe_i32), K(alignbit), K(xnor), K(med3), K(mad_u32_u24), K(sad)};
  for (auto &k : ks) {
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)iters * 256.0 * 8;   // 64 texts x 4 instructions, 8 waves per SIMD
    printf("%-14s %8.3f ms  %6.2f cycles\n", k.name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
  }
  return 0;
}
