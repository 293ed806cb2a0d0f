// vmm_reuse_probe.hip -- is memory from hipMemCreate / hipMemMap safe to write right after the mapping call returns,
// when chunks are created, released and re-created at a high rate (what sa_placement.hip's walk does)?
//
// Observed: a test poisoned a freshly placed arena with a torch fill_, ran the fill, and found ZEROS where the
// poison should have survived.  Suspects: (1) the driver clears VRAM asynchronously (on release or on allocation)
// and a late clear lands on top of our writes; (2) a stale translation after unmap + map at the same address.
// This probe writes a pattern right after mapping, waits, reads it back and classifies what it finds instead.
//
//   ./vmm_reuse_probe [rounds=6] [spacer_chunks=64]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("FAILED %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); \
      fflush(stdout);                                                          \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

__global__ void write_pattern(uint32_t *p, uint64_t n, uint32_t tag) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = tag ^ (uint32_t)i;
}
// counts[0]: as written, [1]: zero, [2]: an older tag's value, [3]: anything else
__global__ void classify(const uint32_t *p, uint64_t n, uint32_t tag, uint32_t old_tag, unsigned long long *counts) {
  unsigned long long c[4] = {0, 0, 0, 0};
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t v = p[i];
    if (v == (tag ^ (uint32_t)i)) c[0]++;
    else if (v == 0) c[1]++;
    else if (v == (old_tag ^ (uint32_t)i)) c[2]++;
    else c[3]++;
  }
  for (int k = 0; k < 4; ++k) if (c[k]) atomicAdd(&counts[k], c[k]);
}

typedef hipMemGenericAllocationHandle_t Handle;

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 6;
  const int spacer_chunks = argc > 2 ? atoi(argv[2]) : 64;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc;
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t chunk = (size_t)512 << 20;
  const int per = 2;
  const uint64_t n = per * chunk / 4;
  unsigned long long *counts;
  CK(hipMalloc(&counts, 32));
  uint32_t old_tag = 0;
  for (int r = 0; r < rounds; ++r) {
    // a burst of creates and releases, like the placement walk
    std::vector<Handle> sp(spacer_chunks);
    for (Handle &h : sp) CK(hipMemCreate(&h, chunk, &prop, 0));
    Handle hs[per];
    for (int k = 0; k < per; ++k) CK(hipMemCreate(&hs[k], chunk, &prop, 0));
    for (Handle &h : sp) CK(hipMemRelease(h));
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, per * chunk, 0, nullptr, 0));
    for (int k = 0; k < per; ++k) CK(hipMemMap((char *)va + k * chunk, chunk, 0, hs[k], 0));
    CK(hipMemSetAccess(va, per * chunk, &acc, 1));
    const uint32_t tag = 0x5A5A0000u + (uint32_t)r * 0x10001u;
    // first look: what does fresh memory hold?
    CK(hipMemset(counts, 0, 32));
    hipLaunchKernelGGL(classify, dim3(4096), dim3(256), 0, 0, (const uint32_t *)va, n, tag, old_tag, counts);
    unsigned long long c0[4];
    CK(hipMemcpy(c0, counts, 32, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(write_pattern, dim3(4096), dim3(256), 0, 0, (uint32_t *)va, n, tag);
    CK(hipDeviceSynchronize());
    unsigned long long c1[4], c2[4];
    CK(hipMemset(counts, 0, 32));
    hipLaunchKernelGGL(classify, dim3(4096), dim3(256), 0, 0, (const uint32_t *)va, n, tag, old_tag, counts);
    CK(hipMemcpy(c1, counts, 32, hipMemcpyDeviceToHost));
    usleep(300000);   // give any background clear time to land
    CK(hipMemset(counts, 0, 32));
    hipLaunchKernelGGL(classify, dim3(4096), dim3(256), 0, 0, (const uint32_t *)va, n, tag, old_tag, counts);
    CK(hipMemcpy(c2, counts, 32, hipMemcpyDeviceToHost));
    printf("round %d va %p  fresh: zero %llu old %llu other %llu | right after write: ok %llu zero %llu old %llu other %llu | 0.3 s later: ok %llu zero %llu old %llu other %llu\n",
           r, va, c0[1], c0[2], c0[3] + c0[0], c1[0], c1[1], c1[2], c1[3], c2[0], c2[1], c2[2], c2[3]);
    fflush(stdout);
    CK(hipMemUnmap(va, per * chunk));
    CK(hipMemAddressFree(va, per * chunk));
    for (int k = 0; k < per; ++k) CK(hipMemRelease(hs[k]));
    old_tag = tag;
  }
  // ---- the placement walk's pattern: two arenas stay mapped; candidates are mapped, written and unmapped one after
  // the other.  Afterwards every candidate's chunks are mapped once more at an address NEVER used before (inside one
  // big reservation) and must hold THEIR candidate's pattern.  Variants:
  //   0  same VA again for every candidate (reserve / free each time), handles kept alive until the end   [the walk as first written]
  //   1  same VA again, but a rejected candidate's handles are released before the next one is mapped
  //   2  every candidate at its own offset of one big reservation, handles kept alive until the end
  //   3  like 0, with a hipMalloc + hipFree after every unmap (does the classic path's TLB flush clean up?)
  //   4  no virtual address is ever given back (no hipMemAddressFree at all, check addresses included)
  const int only = argc > 3 ? atoi(argv[3]) : -1;
  for (int variant = 0; variant < 5; ++variant) {
    if (only >= 0 && variant != only) continue;
    const int n_cand = 8;
    const size_t asz = per * chunk;
    void *check_base = nullptr, *walk_base = nullptr;
    CK(hipMemAddressReserve(&check_base, n_cand * asz, 0, nullptr, 0));
    if (variant == 2) CK(hipMemAddressReserve(&walk_base, n_cand * asz, 0, nullptr, 0));
    std::vector<std::vector<Handle>> cand(n_cand, std::vector<Handle>(per));
    std::vector<Handle> sp;
    std::vector<int> wrong(n_cand, 0);
    for (int jn = 0; jn < n_cand; ++jn) {
      for (int k = 0; k < 6; ++k) { Handle h; CK(hipMemCreate(&h, chunk, &prop, 0)); sp.push_back(h); }
      for (int k = 0; k < per; ++k) CK(hipMemCreate(&cand[jn][k], chunk, &prop, 0));
      void *va = nullptr;
      if (variant == 2) va = (char *)walk_base + jn * asz;
      else CK(hipMemAddressReserve(&va, asz, 0, nullptr, 0));
      for (int k = 0; k < per; ++k) CK(hipMemMap((char *)va + k * chunk, chunk, 0, cand[jn][k], 0));
      CK(hipMemSetAccess(va, asz, &acc, 1));
      hipLaunchKernelGGL(write_pattern, dim3(4096), dim3(256), 0, 0, (uint32_t *)va, n, 0x22220000u + jn + 16 * variant);
      CK(hipDeviceSynchronize());
      CK(hipMemUnmap(va, asz));
      if (variant != 2 && variant != 4) CK(hipMemAddressFree(va, asz));
      if (variant == 3) { void *d = nullptr; CK(hipMalloc(&d, 2 << 20)); CK(hipFree(d)); }
      if (variant == 1) {   // check now (fresh address), then release before the next candidate is mapped
        void *cv = (char *)check_base + jn * asz;
        for (int k = 0; k < per; ++k) CK(hipMemMap((char *)cv + k * chunk, chunk, 0, cand[jn][k], 0));
        CK(hipMemSetAccess(cv, asz, &acc, 1));
        CK(hipMemset(counts, 0, 32));
        hipLaunchKernelGGL(classify, dim3(4096), dim3(256), 0, 0, (const uint32_t *)cv, n, 0x22220000u + jn + 16 * variant, 0, counts);
        unsigned long long c[4];
        CK(hipMemcpy(c, counts, 32, hipMemcpyDeviceToHost));
        wrong[jn] = c[0] != n;
        CK(hipMemUnmap(cv, asz));
        for (int k = 0; k < per; ++k) CK(hipMemRelease(cand[jn][k]));
      }
    }
    if (variant != 1) {
      for (int jn = 0; jn < n_cand; ++jn) {
        void *cv = (char *)check_base + jn * asz;
        for (int k = 0; k < per; ++k) CK(hipMemMap((char *)cv + k * chunk, chunk, 0, cand[jn][k], 0));
        CK(hipMemSetAccess(cv, asz, &acc, 1));
        CK(hipMemset(counts, 0, 32));
        hipLaunchKernelGGL(classify, dim3(4096), dim3(256), 0, 0, (const uint32_t *)cv, n, 0x22220000u + jn + 16 * variant, 0, counts);
        unsigned long long c[4];
        CK(hipMemcpy(c, counts, 32, hipMemcpyDeviceToHost));
        wrong[jn] = c[0] != n;
        uint32_t first[2];
        CK(hipMemcpy(first, cv, 8, hipMemcpyDeviceToHost));
        printf("  variant %d cand %d: ok %llu zero %llu other %llu; word0 %08x (want %08x) word1 %08x\n", variant, jn, c[0], c[1], c[2] + c[3],
               first[0], 0x22220000u + jn + 16 * variant, first[1]);
        CK(hipMemUnmap(cv, asz));
        for (int k = 0; k < per; ++k) CK(hipMemRelease(cand[jn][k]));
      }
    }
    for (Handle &h : sp) CK(hipMemRelease(h));
    if (variant != 4) CK(hipMemAddressFree(check_base, n_cand * asz));
    if (walk_base) CK(hipMemAddressFree(walk_base, n_cand * asz));
    printf("variant %d: candidates NOT holding their own pattern:", variant);
    for (int jn = 0; jn < n_cand; ++jn) printf(" %d", wrong[jn]);
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
