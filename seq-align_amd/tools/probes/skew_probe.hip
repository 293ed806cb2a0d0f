// skew_probe.hip -- WHICH address bits make concurrently written arenas disturb each other?
//
// vmm_probe.hip showed: arena-sized physical handles created 2 GiB apart (congruent mod 2 GiB) disturb each other
// heavily (K2 0.35 ms vs 0.265), handles whose low address bits differ only mildly or not at all.  If the heavy case
// is a matter of LOW address bits, the library can avoid it whatever the physical placement: allocate the arenas with
// slack and skew their base addresses against each other.  This probe measures K2 / K3 of the fill's store pattern
// (one 88 KiB region per wave, 1 KiB to each arena in lock step) as a function of that skew.
//
//   hipcc --offload-arch=gfx950 -O3 -o skew_probe skew_probe.hip && ./skew_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
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

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) probe_streams(char *a0, char *a1, char *a2, int k, uint32_t region_kib,
                                                     uint32_t n_regions) {
  extern __shared__ int occupancy_pad[];
  const int lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= n_regions) return;
  const v4i val = {(int)w, lane, 0, 0};
  const uint64_t base = (uint64_t)w * region_kib * 1024 + lane * 16;
  for (uint32_t b = 0; b < region_kib; ++b) {
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a0 + base + (uint64_t)b * 1024));
    if (k > 1) __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a1 + base + (uint64_t)b * 1024));
    if (k > 2) __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a2 + base + (uint64_t)b * 1024));
  }
}

template <class F>
static float median_ms(F launch, int iters = 6) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int it = 0; it < iters; ++it) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) t.push_back(ms);
  }
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

static const uint32_t kRegionKib = 88;
static const uint32_t kRegions = 10127;
static const size_t kArena = (size_t)kRegions * kRegionKib * 1024;   // 912 564 224 B written per arena

static float K(int k, void *a0, void *a1, void *a2) {
  return median_ms([&] {
    hipLaunchKernelGGL(probe_streams, dim3((kRegions + 3) / 4), dim3(256), 24576, 0, (char *)a0, (char *)a1, (char *)a2, k,
                       kRegionKib, kRegions);
  });
}

struct Phys {
  hipMemGenericAllocationHandle_t h;
  size_t bytes;
  char *va;
  size_t va_bytes;
};
static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_acc;

static Phys phys_create(size_t bytes) {
  Phys p;
  p.bytes = bytes;
  p.va = nullptr;
  p.va_bytes = 0;
  CK(hipMemCreate(&p.h, bytes, &g_prop, 0));
  return p;
}
static void phys_map(Phys &p, size_t va_offset = 0, size_t va_extra = 0) {
  void *va = nullptr;
  p.va_bytes = p.bytes + va_extra;
  CK(hipMemAddressReserve(&va, p.va_bytes, 0, nullptr, 0));
  p.va = (char *)va;
  CK(hipMemMap(p.va + va_offset, p.bytes, 0, p.h, 0));
  CK(hipMemSetAccess(p.va + va_offset, p.bytes, &g_acc, 1));
}
static void phys_unmap(Phys &p, size_t va_offset = 0) {
  CK(hipMemUnmap(p.va + va_offset, p.bytes));
  CK(hipMemAddressFree(p.va, p.va_bytes));
  p.va = nullptr;
}

int main() {
  CK(hipSetDevice(0));
  g_prop = {};
  g_prop.type = hipMemAllocationTypePinned;
  g_prop.location.type = hipMemLocationTypeDevice;
  g_prop.location.id = 0;
  g_acc.location = g_prop.location;
  g_acc.flags = hipMemAccessFlagsProtReadWrite;

  const size_t slack = (size_t)256 << 20;
  const size_t hbytes = ((kArena + slack) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  const size_t stride = (size_t)2 << 30;

  // G0..G3: handles 2 GiB apart in creation order (unmapped spacers in between)
  std::vector<Phys> G, spacers;
  for (int i = 0; i < 4; ++i) {
    if (i) spacers.push_back(phys_create(stride - hbytes));
    G.push_back(phys_create(hbytes));
    phys_map(G.back());
    printf("G%d va %p\n", i, (void *)G.back().va);
  }
  printf("K1 %.4f   K2(G0,G1) %.4f  K2(G0,G2) %.4f  K2(G1,G2) %.4f  K3(G0,G1,G2) %.4f\n", K(1, G[0].va, 0, 0),
         K(2, G[0].va, G[1].va, 0), K(2, G[0].va, G[2].va, 0), K(2, G[1].va, G[2].va, 0), K(3, G[0].va, G[1].va, G[2].va));

  // ---- E2: skew sweep.  K2(G0, G1 + d), K3(G0, G1 + d, G2 + 2d)
  printf("\nskew d        K2(G0,G1+d)  K3(G0,G1+d,G2+2d)\n");
  std::vector<size_t> ds = {0};
  for (size_t d = 1024; d <= ((size_t)64 << 20); d <<= 1) ds.push_back(d);
  for (size_t d : {(size_t)3 << 10, (size_t)5 << 10, (size_t)7 << 10, (size_t)3 << 12, (size_t)5 << 12, (size_t)9 << 12, (size_t)17 << 12,
                   (size_t)33 << 12, (size_t)65 << 12, (size_t)129 << 12, (size_t)257 << 12, (size_t)3 << 16, (size_t)3 << 20, (size_t)5 << 20,
                   (size_t)(88 * 1024), (size_t)(44 * 1024), (size_t)(29 * 1024 + 1024)})
    ds.push_back(d);
  for (size_t d : ds) {
    if (2 * d + kArena > hbytes) continue;
    printf("%10zu   %.4f       %.4f\n", d, K(2, G[0].va, G[1].va + d, 0), K(3, G[0].va, G[1].va + d, G[2].va + 2 * d));
    fflush(stdout);
  }

  // ---- E3: is it the VIRTUAL address (TLB sets) or the physical one?  Remap G1 at a shifted VA.
  printf("\nremap G1 at a VA shifted by s: K2(G0,G1)\n");
  for (size_t s : {(size_t)0, (size_t)4 << 10, (size_t)64 << 10, (size_t)2 << 20, (size_t)34 << 20, (size_t)1 << 30}) {
    phys_unmap(G[1]);
    phys_map(G[1], s, (size_t)2 << 30);
    printf("  s = %10zu  va %p   K2 %.4f   K3(G0,G1,G2) %.4f\n", s, (void *)(G[1].va + s), K(2, G[0].va, G[1].va + s, 0),
           K(3, G[0].va, G[1].va + s, G[2].va));
    phys_unmap(G[1], s);
    phys_map(G[1]);
  }

  // ---- E4: one hipMalloc holding three arenas back to back (what a pooling allocator gives), skewed
  {
    char *base = nullptr;
    const size_t one = (kArena + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    CK(hipMalloc((void **)&base, 3 * one + ((size_t)512 << 20)));
    printf("\none hipMalloc, arenas at 0, S+d, 2S+2d (S = %zu):\n", one);
    for (size_t d : ds) {
      if (d > ((size_t)128 << 20)) continue;
      printf("%10zu   K2 %.4f   K3 %.4f\n", d, K(2, base, base + one + d, 0), K(3, base, base + one + d, base + 2 * one + 2 * d));
      fflush(stdout);
    }
    CK(hipFree(base));
  }

  // ---- E5: inside ONE 6 GiB handle: K2(base, base + 2^k): which strides collide?
  {
    for (Phys &p : G) phys_unmap(p);
    Phys big = phys_create((size_t)6 << 30);
    phys_map(big);
    printf("\none 6 GiB handle: K2(base, base + d), K3(base, base + d, base + 2d)\n");
    for (size_t d = (size_t)1 << 30; d <= ((size_t)5 << 29); d += (size_t)1 << 29) {
      printf("%11zu   K2 %.4f", d, K(2, big.va, big.va + d, 0));
      if (2 * d + kArena <= big.bytes) printf("   K3 %.4f", K(3, big.va, big.va + d, big.va + 2 * d));
      printf("\n");
    }
    for (size_t d : {((size_t)1 << 30) + 4096, ((size_t)1 << 30) + (64 << 10), ((size_t)1 << 30) + (1 << 20), ((size_t)1 << 30) + (32 << 20),
                     ((size_t)2 << 30) + 4096, ((size_t)2 << 30) + (64 << 10), ((size_t)2 << 30) + (1 << 20), ((size_t)2 << 30) + (32 << 20), kArena,
                     kArena + 4096})
      printf("%11zu   K2 %.4f   K3 %.4f\n", d, K(2, big.va, big.va + d, 0), K(3, big.va, big.va + d, big.va + 2 * d));
    phys_unmap(big);
    CK(hipMemRelease(big.h));
  }
  for (Phys &p : G) CK(hipMemRelease(p.h));
  for (Phys &p : spacers) CK(hipMemRelease(p.h));
  return 0;
}
