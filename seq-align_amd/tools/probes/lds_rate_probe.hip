// lds_rate_probe.hip -- how fast a CU takes the LDS writes of the direction-byte rings (sa_fill_dirs*.hip): byte writes at
// lane * CPL + c (two lanes share most dwords), 16-bit writes of two interleaved pairs, dword writes; cycles per wave64
// instruction per CU with 32 waves resident.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rate_probe lds_rate_probe.hip && ./lds_rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(256) k_lds(uint32_t *out, int iters) {
  __shared__ __attribute__((aligned(16))) uint8_t ring[4 * 4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *r = ring + wave * 4096;
  uint32_t v = threadIdx.x * 2654435761u;
  uint32_t pos = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {          // 3 byte writes at lane*3 + c (one pair per wave)
#pragma unroll
        for (int c = 0; c < 3; ++c) r[(pos + lane * 3 + c) & 511] = (uint8_t)(v >> (c * 8));
      } else if (MODE == 1) {   // the same into two rings (lo / hi byte of each half)
#pragma unroll
        for (int c = 0; c < 3; ++c) { r[(pos + lane * 3 + c) & 511] = (uint8_t)(v >> c); r[512 + ((pos + lane * 3 + c) & 511)] = (uint8_t)(v >> (16 + c)); }
      } else if (MODE == 2) {   // 3 16-bit writes at 2*(lane*3 + c): two pairs interleaved
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<uint16_t *>(r + 2 * ((pos + lane * 3 + c) & 511)) = (uint16_t)(v >> c);
      } else if (MODE == 3) {   // 1 dword write per lane (conflict-free)
        *reinterpret_cast<uint32_t *>(r + 4 * ((pos + lane) & 511)) = v;
      } else if (MODE == 4) {   // 3 byte writes at c*64 + lane (lanes on consecutive bytes: 4 lanes per dword)
#pragma unroll
        for (int c = 0; c < 3; ++c) r[(pos + c * 64 + lane) & 511] = (uint8_t)(v >> (c * 8));
      }
      pos += 151;
      v = v * 3 + 1;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = *reinterpret_cast<uint32_t *>(r + 4 * lane) + v;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate / 1e6;
  uint32_t *out;
  const int blocks = cus * 8;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 4000;
  struct { const char *name; void (*k)(uint32_t *, int); int per; } ks[] = {
      {"3 x b8 at lane*3+c", k_lds<0>, 3}, {"6 x b8, two rings", k_lds<1>, 6}, {"3 x b16 interleaved", k_lds<2>, 3},
      {"1 x b32", k_lds<3>, 1}, {"3 x b8 at c*64+lane", k_lds<4>, 3}};
  printf("%d CUs, %.2f GHz: LDS write instructions, 32 waves per CU\n", cus, ghz);
  for (auto &k : ks) {
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double groups_per_cu = (double)iters * 8 * 32;   // "rows" per CU
    const double cyc = ms * 1e-3 * ghz * 1e9 / groups_per_cu;
    printf("%-22s %8.3f ms  %7.1f cycles per row per CU-wave  (%5.1f per LDS instruction)\n", k.name, ms, cyc, cyc / k.per);
  }
  return 0;
}
