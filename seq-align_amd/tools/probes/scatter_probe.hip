// scatter_probe.hip -- what does the MI355X memory system sustain for the stream
// kernel's WRITE PATTERN alone?  Each wave owns one contiguous "pair" region per
// arena (3 arenas) and writes it in 1 KiB blocks, round-robin over the arenas,
// exactly like sa_fill_stream.hip's flushes, with no other work.  Variants change
// which wave writes what, to find what the pattern costs against a linear memset.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

// mode 0: wave w writes region w (like the product: one pair per wave)
// mode 1: linear memset-like: block k of the arena is written by wave (k % n_waves) -- all waves sweep together
// mode 2: like 0 but blocks_per_burst consecutive KiB per arena before switching arena
__global__ void __launch_bounds__(256) probe(int32_t *M, int32_t *A, int32_t *B, uint32_t region_kib,
                                             uint32_t n_regions, int mode, int burst) {
  const int lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_regions) return;
  const v4i val = {(int)w, lane, 3, 4};
  if (mode == 1) {
    const uint32_t n_waves = n_regions;
    const uint64_t total = (uint64_t)region_kib * n_regions;
    for (uint64_t k = w; k < total; k += n_waves) {
      __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(M + k * 256 + lane * 4));
      __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(A + k * 256 + lane * 4));
      __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(B + k * 256 + lane * 4));
    }
    return;
  }
  const uint64_t base = (uint64_t)w * region_kib * 256;
  for (uint32_t k = 0; k < region_kib; k += burst) {
    for (int a = 0; a < 3; ++a) {
      int32_t *p = (a == 0 ? M : a == 1 ? A : B) + base;
      for (int b = 0; b < burst && k + b < region_kib; ++b)
        __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(p + (uint64_t)(k + b) * 256 + lane * 4));
    }
    if (mode == 3) __builtin_amdgcn_s_sleep(8);   // spread the bursts out a little
  }
}

int main(int argc, char **argv) {
  const uint32_t n_regions = argc > 1 ? atoi(argv[1]) : 10000, region_kib = argc > 2 ? atoi(argv[2]) : 89;
  const size_t ints = (size_t)n_regions * region_kib * 256;
  int32_t *buf;
  if (hipMalloc(&buf, 3 * ints * 4 + 4096) != hipSuccess) return 1;
  int32_t *M = buf, *A = buf + ints, *B = buf + 2 * ints;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct V { int mode, burst; const char *name; } vs[] = {
      {1, 1, "linear sweep (memset-like)"}, {0, 1, "one region per wave, 1 KiB x3 round-robin"},
      {2, 2, "one region per wave, 2 KiB bursts"}, {2, 4, "one region per wave, 4 KiB bursts"},
      {2, 16, "one region per wave, 16 KiB bursts"}, {3, 1, "1 KiB x3 + s_sleep"}};
  const double bytes = 3.0 * ints * 4;
  for (int rep = 0; rep < 2; ++rep)
    for (auto &v : vs) {
      float best = 1e9;
      for (int it = 0; it < 8; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3((n_regions + 3) / 4), dim3(256), 0, 0, M, A, B, region_kib, n_regions, v.mode, v.burst);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2 && ms < best) best = ms;
      }
      printf("%-48s %.4f ms  %.0f GB/s\n", v.name, best, bytes / best / 1e6);
    }
  return 0;
}
