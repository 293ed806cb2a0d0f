// alloc_kind_probe.hip -- does ANY allocation call hand out memory of another class than plain hipMalloc's, without a walk?
//
// sa_placement.hip reaches a second / third class of HBM by walking the allocation order (up to 160 GiB of 512 MiB chunks
// held for a moment) because the VRAM manager hands memory out in address order.  If some allocation KIND
// were served from elsewhere (top-down, another pool), three arenas of three kinds would need no walk at all.  This probe
// times the fill's store pattern (three streams, K3) on arenas of 870 MiB from every kind of device allocation the HIP
// runtime offers, alone and mixed:   ./alloc_kind_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) probe_streams(char *a0, char *a1, char *a2, uint32_t region_kib, uint32_t n_regions) {
  extern __shared__ int occupancy_pad[];
  const int lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= n_regions) return;
  const v4i val = {(int)w, lane, 0, 0};
  const uint64_t base = (uint64_t)w * region_kib * 1024 + lane * 16;
  for (uint32_t b = 0; b < region_kib; ++b) {
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a0 + base + (uint64_t)b * 1024));
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a1 + base + (uint64_t)b * 1024));
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a2 + base + (uint64_t)b * 1024));
  }
}
static const uint32_t kRegionKib = 88, kRegions = 10127;
static const size_t kBytes = (size_t)kRegionKib * 1024 * kRegions + (1 << 20);
static float K3(void *a0, void *a1, void *a2) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe_streams, dim3((kRegions + 3) / 4), dim3(256), 24576, 0, (char *)a0, (char *)a1, (char *)a2, kRegionKib, kRegions);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (it) t.push_back(ms);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

enum Kind { kMalloc, kFine, kUncached, kManaged, kAsync, kVmm, kKinds };
static const char *kName[kKinds] = {"hipMalloc", "ExtMalloc(finegrained)", "ExtMalloc(uncached)", "hipMallocManaged+prefetch", "hipMallocAsync", "hipMemCreate+map"};
static void *alloc(Kind k) {
  void *p = nullptr;
  hipError_t e = hipSuccess;
  switch (k) {
    case kMalloc: e = hipMalloc(&p, kBytes); break;
    case kFine: e = hipExtMallocWithFlags(&p, kBytes, hipDeviceMallocFinegrained); break;
    case kUncached: e = hipExtMallocWithFlags(&p, kBytes, hipDeviceMallocUncached); break;
    case kManaged:
      e = hipMallocManaged(&p, kBytes);
      if (e == hipSuccess) { hipMemAdvise(p, kBytes, hipMemAdviseSetPreferredLocation, 0); e = hipMemPrefetchAsync(p, kBytes, 0, 0); hipDeviceSynchronize(); }
      break;
    case kAsync: e = hipMallocAsync(&p, kBytes, 0); hipDeviceSynchronize(); break;
    case kVmm: {
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      size_t gran = 0;
      hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
      const size_t sz = (kBytes + gran - 1) / gran * gran;
      hipMemGenericAllocationHandle_t h;
      e = hipMemCreate(&h, sz, &prop, 0);
      if (e == hipSuccess) e = hipMemAddressReserve(&p, sz, 0, nullptr, 0);
      if (e == hipSuccess) e = hipMemMap(p, sz, 0, h, 0);
      hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      if (e == hipSuccess) e = hipMemSetAccess(p, sz, &acc, 1);
      break;
    }
    default: break;
  }
  if (e != hipSuccess) { printf("  (%s failed: %s)\n", kName[k], hipGetErrorString(e)); (void)hipGetLastError(); return nullptr; }
  hipMemset(p, 0, kBytes);
  hipDeviceSynchronize();
  return p;
}

int main() {
  // reference: three plain allocations (one class: ~0.52 ms; three classes: ~0.39 ms)
  void *m = alloc(kMalloc), *a = alloc(kMalloc), *b = alloc(kMalloc);
  printf("hipMalloc x 3: %.3f ms   (%p %p %p)\n", K3(m, a, b), m, a, b);
  for (int k = kFine; k < kKinds; ++k) {
    void *x = alloc((Kind)k);
    if (!x) continue;
    printf("hipMalloc, hipMalloc, %-28s %.3f ms   (%p)\n", kName[k], K3(m, a, x), x);
    void *y = alloc((Kind)k);
    if (y) printf("hipMalloc, %s x 2   %.3f ms   (%p)\n", kName[k], K3(m, x, y), y);
    for (int j = k + 1; j < kKinds; ++j) {
      void *z = alloc((Kind)j);
      if (!z) continue;
      printf("hipMalloc, %s, %s  %.3f ms\n", kName[k], kName[j], K3(m, x, z));
    }
  }
  return 0;
}
