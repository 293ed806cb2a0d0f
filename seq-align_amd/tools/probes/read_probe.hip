// read_probe.hip -- what this GPU READS at all, by access pattern: the ceiling sw_reduce_kernel (sa_reduce.hip: one wave per
// pair streams the pair's match_scores, 4 B per cell) is measured against.  6 GB of int32 (C3's match_scores), every variant
// reduces to one max per wave so that nothing is optimised away.
//   hipcc --offload-arch=gfx950 -O3 -o read_probe read_probe.hip && ./read_probe
// Patterns:  linear   : wave w of W reads blocks w, w + W, w + 2 W ... of 1 KiB (the window of addresses in flight is narrow: a memcpy's)
//            regions  : wave w reads its own contiguous region of R KiB front to back (the reduction's: one region per pair)
// x loads in flight per wave (1 KiB each) x {default, nt} x waves per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT, bool REGIONS>
__global__ void __launch_bounds__(256) read_kernel(const int *__restrict__ src, uint64_t total_kib, uint64_t region_kib, int *out) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
  int best = 0;
  if (REGIONS) {
    // regions are dealt out round-robin: wave w takes regions w, w + n_waves, ...
    for (uint64_t r = w; r * region_kib < total_kib; r += n_waves) {
      const v4i *p = reinterpret_cast<const v4i *>(src + r * region_kib * 256) + lane;
      const uint64_t kib = std::min(region_kib, total_kib - r * region_kib);
      for (uint64_t k = 0; k < kib; k += UNROLL) {
        v4i q[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (k + u < kib) q[u] = NT ? __builtin_nontemporal_load(p + (k + u) * 64) : p[(k + u) * 64]; else q[u] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) best = max(best, max(max(q[u].x, q[u].y), max(q[u].z, q[u].w)));
      }
    }
  } else {
    for (uint64_t k = w * UNROLL; k < total_kib; k += n_waves * UNROLL) {
      v4i q[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const v4i *p = reinterpret_cast<const v4i *>(src + (k + u) * 256) + lane;
        q[u] = (k + u < total_kib) ? (NT ? __builtin_nontemporal_load(p) : *p) : v4i{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) best = max(best, max(max(q[u].x, q[u].y), max(q[u].z, q[u].w)));
    }
  }
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  if (lane == 0) out[w] = best;
}

template <int UNROLL, bool NT, bool REGIONS>
static float run(const int *src, uint64_t total_kib, uint64_t region_kib, int *out, unsigned blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 7; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((read_kernel<UNROLL, NT, REGIONS>), dim3(blocks), dim3(256), 0, 0, src, total_kib, region_kib, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const uint64_t bytes = 6046040000ull / 1024 * 1024, kib = bytes / 1024;
  int *src, *out;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&out, 4 << 20));
  CHECK(hipMemset(src, 1, bytes));
  CHECK(hipDeviceSynchronize());
  const uint64_t region = 590;   // KiB: a C3 pair's match_scores
#define ROW(name, U, NT, REG, blocks) { const float ms = run<U, NT, REG>(src, kib, region, out, blocks); \
    printf("%-44s %7.4f ms  %6.3f TB/s  %5.3f of 8 TB/s\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8); }
  ROW("linear, 4 KiB in flight per wave, 2048 wg", 4, false, false, 2048)
  ROW("linear, 4 KiB, nt", 4, true, false, 2048)
  ROW("linear, 8 KiB, nt", 8, true, false, 2048)
  ROW("linear, 2 KiB, nt, 4096 wg", 2, true, false, 4096)
  ROW("linear, 4 KiB, nt, 1024 wg", 4, true, false, 1024)
  ROW("regions (590 KiB each), 4 KiB, 2560 wg", 4, false, true, 2560)
  ROW("regions, 4 KiB, nt, 2560 wg", 4, true, true, 2560)
  ROW("regions, 8 KiB, nt, 2560 wg", 8, true, true, 2560)
  ROW("regions, 4 KiB, nt, 2048 wg (persistent)", 4, true, true, 2048)
  ROW("regions, 4 KiB, nt, 1024 wg (persistent)", 4, true, true, 1024)
  ROW("regions, 2 KiB, nt, 2560 wg", 2, true, true, 2560)
  return 0;
}
