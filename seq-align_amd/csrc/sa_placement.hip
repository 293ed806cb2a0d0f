// sa_placement.hip -- where the three output arenas live in HBM.
//
// Replaces nothing in the reference (its three matrices are three malloc()s,
// src/alignment.c:183-190); this is the MI355X side of "memory laid out for
// 288 GB of HBM3E".
//
// Measured (seq-align_amd/tools/placement_scan.py, tools/probes/stream_probe.hip,
// profiles/r01_placement_*.txt): the fill kernels write M, A and
// B concurrently, and write streams that sit within the same ~16 GiB granule of
// PHYSICAL address space slow each other down.  With the arenas allocated back
// to back (one allocation, or three consecutive ones -- what any allocator does
// by default) the C2 fill takes 0.52 ms; with one arena >= 24 GB away from the
// others 0.41 ms, the speed of a plain memset of the same bytes.  The relation
// repeats with a period of 128 GiB.  A single sequential stream does not care,
// which is why memset-style microbenchmarks never show it.
//
// Physical addresses are not visible from user space, but the driver hands out
// VRAM roughly in address order, so: allocate arena, spacer, arena, spacer,
// arena, then free the spacers (the arenas keep their placement, the spacers
// cost nothing afterwards; a spacer is a run of arena-sized allocations, see alloc3).  Because that is a heuristic, the result is CHECKED
// with a 3-stream write probe against a 1-stream baseline on the same memory and
// re-tried with a different spacing, keeping the best.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "sa_kernels.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

// the stream kernel's write pattern without the arithmetic: wave w owns region w
// of each arena and appends 1 KiB blocks to the three of them in lock step
__global__ void __launch_bounds__(256) probe_three_streams(char *a0, char *a1, char *a2, uint32_t region_kib,
                                                           uint32_t n_regions) {
  extern __shared__ int occupancy_pad[];   // 24 KiB per workgroup, like fill_stream_kernel<3,...>
  const int lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= n_regions) return;
  const v4i val = {(int)w, lane, 0, 0};
  const uint64_t base = (uint64_t)w * region_kib * 1024 + lane * 16;
  for (uint32_t k = 0; k < region_kib; ++k) {
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a0 + base + (uint64_t)k * 1024));
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a1 + base + (uint64_t)k * 1024));
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a2 + base + (uint64_t)k * 1024));
  }
}

// baseline: one sequential stream, short-lived workgroups, 4 KiB each
__global__ void __launch_bounds__(256) probe_one_stream(char *a, uint64_t total_kib) {
  const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= total_kib) return;
  const v4i val = {(int)blockIdx.x, 0, 0, 0};
  __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a + b * 1024 + (threadIdx.x & 63) * 16));
}

template <class F>
float median_ms(hipStream_t st, F launch) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
  std::vector<float> t;
  for (int it = 0; it < 5; ++it) {
    (void)hipEventRecord(e0, st);
    launch();
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) { t.clear(); break; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (it) t.push_back(ms);   // first = warm-up
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (t.empty()) return -1.f;
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

// bandwidth of the 3-stream pattern relative to one sequential stream over the
// same memory: ~0.75 when the arenas disturb each other, ~0.95 when they do not
float placement_quality(void *const a[3], size_t bytes, hipStream_t st) {
  const uint32_t region_kib = 88;
  const size_t use = std::min(bytes, (size_t)1 << 30) / 1024;            // KiB per arena
  const uint32_t n_regions = (uint32_t)(use / region_kib);
  if (n_regions < 2048) return -1.f;
  const uint64_t kib = (uint64_t)n_regions * region_kib;
  const float t3 = median_ms(st, [&] {
    hipLaunchKernelGGL(probe_three_streams, dim3((n_regions + 3) / 4), dim3(256), 24576, st, (char *)a[0], (char *)a[1],
                       (char *)a[2], region_kib, n_regions);
  });
  const float t1 = median_ms(st, [&] {
    hipLaunchKernelGGL(probe_one_stream, dim3((unsigned)((kib + 3) / 4)), dim3(256), 0, st, (char *)a[0], kib);
  });
  if (t3 <= 0 || t1 <= 0) return -1.f;
  return 3.f * t1 / t3;
}

void free3(void *a[3]) {
  for (int k = 0; k < 3; ++k) {
    if (a[k]) (void)hipFree(a[k]);
    a[k] = nullptr;
  }
}

// arena, spacer, arena, spacer, arena; the spacers are freed again.  A spacer is made of allocations of the
// ARENA's size: the driver's buddy allocator serves a request from the free lists of the block sizes it decomposes
// into, so one big spacer allocation would leave the fragments next to the previous arena for the next arena
// to land in; same-sized fillers use exactly those fragments up first.
hipError_t alloc3(size_t bytes, size_t spacer, void *out[3]) {
  out[0] = out[1] = out[2] = nullptr;
  std::vector<void *> fill;
  const size_t n_fill = spacer ? (spacer + bytes - 1) / bytes : 0;
  hipError_t e = hipSuccess;
  for (int k = 0; k < 3 && e == hipSuccess; ++k) {
    for (size_t f = 0; k && f < n_fill; ++f) {
      void *p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess) {   // no room: carry on with what we have
        (void)hipGetLastError();
        break;
      }
      fill.push_back(p);
    }
    e = hipMalloc(&out[k], bytes);
  }
  for (void *p : fill) (void)hipFree(p);
  if (e != hipSuccess) free3(out);
  return e;
}

}  // namespace

// SEQALIGN_ARENA_SPREAD_GIB: first spacer size (default 24; 0 = plain allocation)
// SEQALIGN_ARENA_TRIES: placements to try at most (default 4)
hipError_t sa_alloc_arenas_spread(size_t bytes, void *out[3], hipStream_t stream, float *quality) {
  size_t gib = 24;
  int tries = 4;
  if (const char *env = getenv("SEQALIGN_ARENA_SPREAD_GIB")) gib = (size_t)strtoull(env, nullptr, 10);
  if (const char *env = getenv("SEQALIGN_ARENA_TRIES")) tries = std::max(1, atoi(env));
  if (quality) *quality = -1.f;
  size_t spacer = gib << 30, free_b = 0, total_b = 0;
  // arenas as large as the granule span several of them anyway; small ones are not bandwidth-bound
  if (bytes >= spacer / 2 || bytes < ((size_t)256 << 20) || hipMemGetInfo(&free_b, &total_b) != hipSuccess)
    spacer = 0;
  if (!spacer) return alloc3(bytes, 0, out);

  void *best[3] = {nullptr, nullptr, nullptr};
  float best_q = -2.f;
  hipError_t e = hipSuccess;
  for (int attempt = 0; attempt < tries; ++attempt) {
    // the best placement so far stays allocated while the next one is made: different physical memory
    const size_t sp = spacer + (size_t)attempt * ((size_t)8 << 30);
    // Transient footprint of an attempt: the three arenas + two spacers (+ the best placement so far, which stays
    // allocated).  It must fit in HALF of what is free right now, so that another context / process on the same
    // GPU is never pushed out of memory by a placement search; otherwise allocate plainly.
    const size_t transient = 3 * bytes + 2 * sp + (best[0] ? 3 * bytes : 0);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || transient > free_b / 2) {
      if (best[0]) break;
      return alloc3(bytes, 0, out);   // not enough room to spread without crowding the device
    }
    void *cand[3];
    e = alloc3(bytes, sp, cand);
    if (e != hipSuccess) break;
    const float q = placement_quality(cand, bytes, stream);
    if (q > best_q) {
      free3(best);
      for (int k = 0; k < 3; ++k) best[k] = cand[k];
      best_q = q;
    } else {
      free3(cand);
    }
    if (best_q < 0 || best_q >= 0.97f) break;   // probe unavailable, or good enough (0.92-0.96 still costs 5-8 %)
  }
  if (!best[0]) return e != hipSuccess ? e : hipErrorOutOfMemory;
  (void)hipGetLastError();   // a failed later attempt must not surface as the next launch's error
  for (int k = 0; k < 3; ++k) out[k] = best[k];
  if (quality) *quality = best_q;
  return hipSuccess;
}
