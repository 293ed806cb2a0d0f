// stream_probe.hip -- how does the MI355X memory system treat K concurrent write
// streams per wave, as a function of the DISTANCE between the streams?
//
// sa_fill_stream.hip has every wave write three streams (M, A, B of its pair) in
// lock step: 1 KiB to M+x, 1 KiB to A+x, 1 KiB to B+x, then x += 1 KiB.  Its run
// time is bimodal per ALLOCATION (0.46 / 0.53 ms on C2) while a memset of the
// same arenas is not.  This probe replays only that store pattern: wave w owns
// region w (region_bytes apart) in each of the K streams, stream k starts at
// base + k*dist.  Sweeping dist inside ONE allocation separates "which physical
// pages" from "which address relation".
//
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
//   ./stream_probe [n_regions=10000] [region_kib=89]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int K>
__global__ void __launch_bounds__(256) probe(char *base, uint64_t dist, uint64_t region_bytes, uint32_t region_kib,
                                             uint32_t n_regions, int rotate, int burst) {
  extern __shared__ int lds_pad[];   // occupancy like the product: 24 KiB per workgroup
  const int lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= n_regions) return;
  const v4i val = {(int)w, lane, 3, 4};
  char *p = base + (uint64_t)w * region_bytes + lane * 16;
  if (rotate == 2) {   // memset-like: KiB block b of each stream is written by wave b % n_regions
    const uint64_t total = (uint64_t)region_kib * n_regions;
    for (uint64_t b = w; b < total; b += n_regions)
#pragma unroll
      for (int s = 0; s < K; ++s)
        __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(base + s * dist + b * 1024 + lane * 16));
    return;
  }
  if (burst > 1) {   // burst consecutive KiB per stream before switching streams
    for (uint32_t k = 0; k < region_kib; k += burst)
#pragma unroll
      for (int s = 0; s < K; ++s)
        for (uint32_t b = k; b < k + burst && b < region_kib; ++b)
          __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(p + s * dist + (uint64_t)b * 1024));
    return;
  }
  for (uint32_t k = 0; k < region_kib; ++k) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      // rotate: stream s of wave w starts (s*region_kib/K) blocks ahead (mod region): the K streams of a wave
      // are then never at the same offset x
      uint32_t kk = k;
      if (rotate) { kk = k + s * (region_kib / K); if (kk >= region_kib) kk -= region_kib; }
      __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(p + s * dist + (uint64_t)kk * 1024));
    }
  }
}

// fill_-like: short-lived workgroups, workgroup b writes `chunk_kib` KiB at offset b*chunk_kib KiB of each of K streams
template <int K>
__global__ void __launch_bounds__(256) chunk_fill(char *base, uint64_t dist, uint32_t chunk_kib, uint64_t total_kib) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const v4i val = {(int)blockIdx.x, lane, 3, 4};
  for (uint32_t k = wave; k < chunk_kib; k += 4) {
    const uint64_t b = (uint64_t)blockIdx.x * chunk_kib + k;
    if (b >= total_kib) return;
#pragma unroll
    for (int s = 0; s < K; ++s)
      __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(base + s * dist + b * 1024 + lane * 16));
  }
}
template <int K>
static float run_chunk(char *base, uint64_t dist, uint32_t chunk_kib, uint64_t total_kib) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(chunk_fill<K>, dim3((total_kib + chunk_kib - 1) / chunk_kib), dim3(256), 0, 0, base, dist, chunk_kib, total_kib);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 3) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  hipEventDestroy(e0); hipEventDestroy(e1);
  return t[t.size() / 2];
}
static int g_lds = 24576;
static float run(int K, char *base, uint64_t dist, uint64_t region_bytes, uint32_t region_kib, uint32_t n, int rotate, int burst = 1) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    dim3 g((n + 3) / 4), b(256);
    if (K == 1) hipLaunchKernelGGL(probe<1>, g, b, g_lds, 0, base, dist, region_bytes, region_kib, n, rotate, burst);
    if (K == 2) hipLaunchKernelGGL(probe<2>, g, b, g_lds, 0, base, dist, region_bytes, region_kib, n, rotate, burst);
    if (K == 3) hipLaunchKernelGGL(probe<3>, g, b, g_lds, 0, base, dist, region_bytes, region_kib, n, rotate, burst);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 3) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  hipEventDestroy(e0); hipEventDestroy(e1);
  return t[t.size() / 2];
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 10000, region_kib = argc > 2 ? atoi(argv[2]) : 89;
  const int n_alloc = argc > 3 ? atoi(argv[3]) : 3;
  const uint64_t region_bytes = (uint64_t)region_kib * 1024 + 128;   // like the product: 91264 B apart, not a KiB multiple
  const uint64_t S = ((uint64_t)n * region_bytes + 4095) / 4096 * 4096;
  const uint64_t slack = 64ull << 20;
  if (argc > 8) {   // domain map of one 64 GiB slab: two 256 MiB windows, short-lived 4 KiB chunks
    const uint64_t GiB = 1ull << 30, total = 64 * GiB, win_kib = (GiB / 4) / 1024;
    char *big;
    if (hipMalloc(&big, total) != hipSuccess) return 1;
    printf("slab at %p; K=1 256 MiB: %.0f us\n", (void *)big, 1000 * run_chunk<1>(big, 0, 4, win_kib));
    for (uint64_t b0 : {(uint64_t)0, 20 * GiB, 40 * GiB}) {
      printf("window1 at %2.0f GiB; window2 at j*0.5 GiB, j=0..127 (us; x = overlap)\n", (double)b0 / GiB);
      for (int j = 0; j < 128; ++j) {
        const uint64_t w2 = j * (GiB / 2);
        if (w2 + GiB / 4 > b0 && w2 < b0 + GiB / 4) { printf("  x"); continue; }
        // run_chunk writes base + s*dist: base = window1, dist = w2 - b0 (may be "negative": unsigned wrap is fine)
        printf(" %3.0f", 1000 * run_chunk<2>(big + b0, w2 - b0, 4, win_kib));
        if (j % 32 == 31) printf("\n");
      }
    }
    return 0;
  }
  if (argc > 7) {   // one big allocation: K=2 (short-lived 4 KiB chunks) as a function of where the two windows are
    const uint64_t GiB = 1ull << 30, total = 20 * GiB, total_kib = S / 1024;
    for (int a = 0; a < 2; ++a) {
      char *big;
      if (hipMalloc(&big, total) != hipSuccess) return 1;
      printf("big alloc %d at %p\n", a, (void *)big);
      for (uint64_t b0 : {(uint64_t)0, GiB / 2, GiB, 3 * GiB}) {
        printf("  window1 at %.1f GiB, window2 at +j*256 MiB, j=1..64 (us):\n   ", (double)b0 / GiB);
        for (int j = 1; j <= 64; ++j) printf(" %.0f", 1000 * run_chunk<2>(big + b0, j * (GiB / 4), 4, total_kib));
        printf("\n");
      }
      printf("  K=3 at strides 1,2,3,4,5,6 GiB: ");
      for (int g = 1; g <= 6; ++g) printf(" %.0f", 1000 * run_chunk<3>(big, g * GiB, 4, total_kib));
      printf("\n  K=3 at stride S: %.0f   K=1: %.0f\n", 1000 * run_chunk<3>(big, S, 4, total_kib), 1000 * run_chunk<1>(big, S, 4, total_kib));
    }
    return 0;
  }
  std::vector<char *> bufs;
  for (int a = 0; a < n_alloc; ++a) {
    char *buf;
    if (hipMalloc(&buf, 3 * S + 4 * slack) != hipSuccess) return 1;
    bufs.push_back(buf);
    const double gb = (double)n * region_kib * 1024 / 1e6;
    printf("alloc %d at %p  S=%llu\n", a, (void *)buf, (unsigned long long)S);
    printf("  K=1                      %.4f ms %6.0f GB/s\n", run(1, buf, 0, region_bytes, region_kib, n, 0), gb / run(1, buf, 0, region_bytes, region_kib, n, 0));
    if (argc > 6) {
      const uint64_t total_kib = S / 1024;
      printf("  chunk-fill  K=1 x3 launches: 4K %.4f 16K %.4f | K=3 one launch: 4K %.4f 16K %.4f 64K %.4f | region K=3 %.4f | linear K=3 %.4f\n",
             3 * run_chunk<1>(buf, S, 4, total_kib), 3 * run_chunk<1>(buf, S, 16, total_kib),
             run_chunk<3>(buf, S, 4, total_kib), run_chunk<3>(buf, S, 16, total_kib), run_chunk<3>(buf, S, 64, total_kib),
             run(3, buf, S, region_bytes, region_kib, n, 0), run(3, buf, S, region_bytes, region_kib, n, 2));
      continue;
    }
    if (argc > 5) {   // K=2 linear: time as a function of the distance between the two windows
      const uint64_t MB = 1 << 20;
      printf("  dist = j * 32 MB, j=1..56:\n   ");
      for (int j = 1; j <= 56; ++j) printf(" %.0f", 1000 * run(2, buf, j * 32 * MB, region_bytes, region_kib, n, 2));
      printf("\n  dist = 1024 MB + j * 2 MB, j=0..31:\n   ");
      for (int j = 0; j < 32; ++j) printf(" %.0f", 1000 * run(2, buf, 1024 * MB + j * 2 * MB, region_bytes, region_kib, n, 2));
      printf("\n  dist = 1024 MB + j * 64 KiB, j=0..31:\n   ");
      for (int j = 0; j < 32; ++j) printf(" %.0f", 1000 * run(2, buf, 1024 * MB + j * 65536, region_bytes, region_kib, n, 2));
      printf("\n  dist = 1024 MB + j * 4 KiB, j=0..15:\n   ");
      for (int j = 0; j < 16; ++j) printf(" %.0f", 1000 * run(2, buf, 1024 * MB + j * 4096, region_bytes, region_kib, n, 2));
      printf("\n");
      continue;
    }
    if (argc > 4) {
      printf("  K=2 S   %.4f | K=2 2S %.4f | K=3 %.4f | K=3 linear %.4f | K=1 linear %.4f\n",
             run(2, buf, S, region_bytes, region_kib, n, 0), run(2, buf, 2 * S, region_bytes, region_kib, n, 0),
             run(3, buf, S, region_bytes, region_kib, n, 0), run(3, buf, S, region_bytes, region_kib, n, 2),
             run(1, buf, S, region_bytes, region_kib, n, 2));
      continue;
    }
    for (int lds : {24576, 40000, 65536}) {
      g_lds = lds;
      for (int burst : {1, 2, 4, 8, 16, 89}) {
        const float ms = run(3, buf, S, region_bytes, region_kib, n, 0, burst);
        printf("  lds=%d K=3 burst=%2d KiB  misaligned(+128B) %.4f ms %6.0f GB/s", lds, burst, ms, 3 * gb / ms);
        const float ms2 = run(3, buf, S, (uint64_t)region_kib * 1024, region_kib, n, 0, burst);
        printf("   | KiB-aligned regions %.4f ms %6.0f GB/s\n", ms2, 3 * gb / ms2);
      }
    }
    g_lds = 24576;
  }
  return 0;
}
