// stripe_probe.hip -- arenas built from physical chunks of two "classes" of HBM.
//
// vmm_probe / skew_probe (round 3): the fill's store pattern (thousands of concurrent sequential write streams)
// runs at ~5 TB/s when everything it writes lies in one class of physical memory -- large blocks (tens of GiB) of
// the allocation order, no dependence on low address bits or on the virtual address -- and faster when the
// streams are spread over two classes (K2 0.265 instead of 0.36 ms).  A linear stream does not care.  That looks
// like bank / rank parallelism: more classes in use at once = more open rows.
//
// So: is an arena STRIPED over both classes (chunk i from class i % 2, mapped back to back with hipMemMap) better
// than whole arenas placed in different classes (round 2's strategy: M, A here, B there)?
//
//   ./stripe_probe [chunk_mib=64] [max_depth_gib=160]
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
__global__ void __launch_bounds__(256) probe_linear(char *a, uint64_t total_kib) {
  const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= total_kib) return;
  const v4i val = {(int)blockIdx.x, 0, 0, 0};
  __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a + b * 1024 + (threadIdx.x & 63) * 16));
}

template <class F>
static float median_ms(F launch, int iters = 8) {
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
static const size_t kArena = (size_t)kRegions * kRegionKib * 1024;

static float K(int k, void *a0, void *a1, void *a2) {
  return median_ms([&] {
    hipLaunchKernelGGL(probe_streams, dim3((kRegions + 3) / 4), dim3(256), 24576, 0, (char *)a0, (char *)a1, (char *)a2, k,
                       kRegionKib, kRegions);
  });
}
static float L(void *a) {
  const uint64_t kib = (uint64_t)kRegions * kRegionKib;
  return median_ms([&] { hipLaunchKernelGGL(probe_linear, dim3((unsigned)((kib + 3) / 4)), dim3(256), 0, 0, (char *)a, kib); });
}

static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_acc;
typedef hipMemGenericAllocationHandle_t Handle;

static bool create(Handle *h, size_t bytes) {
  hipError_t e = hipMemCreate(h, bytes, &g_prop, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

// a VA range backed by the given chunk handles, back to back
struct Arena {
  char *va = nullptr;
  size_t bytes = 0, chunk = 0;
  void map(const std::vector<Handle> &hs, size_t chunk_bytes) {
    chunk = chunk_bytes;
    bytes = hs.size() * chunk;
    void *p = nullptr;
    CK(hipMemAddressReserve(&p, bytes, 0, nullptr, 0));
    va = (char *)p;
    for (size_t i = 0; i < hs.size(); ++i) CK(hipMemMap(va + i * chunk, chunk, 0, hs[i], 0));
    CK(hipMemSetAccess(va, bytes, &g_acc, 1));
  }
  void unmap() {
    if (!va) return;
    for (size_t off = 0; off < bytes; off += chunk) CK(hipMemUnmap(va + off, chunk));
    CK(hipMemAddressFree(va, bytes));
    va = nullptr;
  }
};

int main(int argc, char **argv) {
  const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 20;
  const size_t max_depth = (size_t)(argc > 2 ? atoi(argv[2]) : 160) << 30;
  CK(hipSetDevice(0));
  g_prop = {};
  g_prop.type = hipMemAllocationTypePinned;
  g_prop.location.type = hipMemLocationTypeDevice;
  g_prop.location.id = 0;
  g_acc.location = g_prop.location;
  g_acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t per_arena = (kArena + chunk - 1) / chunk;       // chunks per arena
  const size_t pool_n = (3 * per_arena + 1) / 2 + 1;           // chunks per class for three striped arenas
  printf("chunk %zu MiB, %zu chunks per arena, %zu per pool\n", chunk >> 20, per_arena, pool_n);

  // class X: the first allocations
  std::vector<Handle> X(2 * per_arena + pool_n), Y;
  for (Handle &h : X) if (!create(&h, chunk)) { printf("X create failed\n"); return 2; }
  Arena x0, x1;
  x0.map(std::vector<Handle>(X.begin(), X.begin() + per_arena), chunk);
  x1.map(std::vector<Handle>(X.begin() + per_arena, X.begin() + 2 * per_arena), chunk);
  const float lin = L(x0.va), k1 = K(1, x0.va, 0, 0), k2xx = K(2, x0.va, x1.va, 0);
  printf("X arenas: linear %.4f  K1 %.4f  K2(x0,x1) %.4f\n", lin, k1, k2xx);

  // scan for memory that does not disturb x0: unmapped 4 GiB spacers, arena-sized candidates
  std::vector<Handle> spacers;
  size_t depth = 0;
  bool found = false;
  std::vector<Handle> cand(per_arena);
  Arena c;
  while (depth < max_depth) {
    Handle sp;
    if (!create(&sp, (size_t)4 << 30)) { printf("spacer create failed at depth %zu GiB\n", depth >> 30); break; }
    spacers.push_back(sp);
    depth += (size_t)4 << 30;
    bool ok = true;
    for (Handle &h : cand) ok = ok && create(&h, chunk);
    if (!ok) { printf("candidate create failed\n"); break; }
    c.map(cand, chunk);
    const float k2 = K(2, x0.va, c.va, 0);
    printf("depth %3zu GiB: K2(x0,cand) %.4f  (2*K1 = %.4f)\n", depth >> 30, k2, 2 * k1);
    fflush(stdout);
    if (k2 < 0.80f * 2 * k1) { found = true; break; }
    c.unmap();
    for (Handle &h : cand) CK(hipMemRelease(h));
  }
  if (!found) { printf("no second class found\n"); return 1; }
  // class Y: the candidate's chunks + more right behind it
  Y = cand;
  while (Y.size() < per_arena + pool_n) {
    Handle h;
    if (!create(&h, chunk)) { printf("Y create failed\n"); return 2; }
    Y.push_back(h);
  }
  Arena y1;
  y1.map(std::vector<Handle>(Y.begin() + per_arena, Y.begin() + 2 * per_arena), chunk);
  printf("Y arenas: K1 %.4f  K2(y0,y1) %.4f  K2(x0,y0) %.4f  K2(x1,y1) %.4f\n", K(1, c.va, 0, 0), K(2, c.va, y1.va, 0),
         K(2, x0.va, c.va, 0), K(2, x1.va, y1.va, 0));
  printf("whole arenas:   K3(x0,x1,y0) %.4f   K3(x0,y0,y1) %.4f   [3*linear = %.4f]\n", K(3, x0.va, x1.va, c.va),
         K(3, x0.va, c.va, y1.va), 3 * lin);
  y1.unmap();
  x1.unmap();
  c.unmap();
  x0.unmap();

  // striped arenas: chunk i of arena a from class (i + a) % 2 -- at several stripe widths (in chunks)
  for (size_t width : {(size_t)1, (size_t)2, (size_t)4}) {
    size_t nx = 0, ny = 0;
    Arena s[3];
    bool ok = true;
    for (int a = 0; a < 3 && ok; ++a) {
      std::vector<Handle> hs;
      for (size_t i = 0; i < per_arena; ++i) {
        const bool useX = ((i / width + a) & 1) == 0;
        if (useX) { if (nx >= X.size()) { ok = false; break; } hs.push_back(X[nx++]); }
        else { if (ny >= Y.size()) { ok = false; break; } hs.push_back(Y[ny++]); }
      }
      if (ok) s[a].map(hs, chunk);
    }
    if (!ok) { printf("stripe width %zu: pools too small\n", width); continue; }
    printf("striped, %zu x %zu MiB per stripe: linear %.4f  K1 %.4f  K2 %.4f  K3 %.4f\n", width, chunk >> 20, L(s[0].va),
           K(1, s[0].va, 0, 0), K(2, s[0].va, s[1].va, 0), K(3, s[0].va, s[1].va, s[2].va));
    fflush(stdout);
    for (int a = 0; a < 3; ++a) s[a].unmap();
  }
  for (Handle &h : X) CK(hipMemRelease(h));
  for (Handle &h : Y) CK(hipMemRelease(h));
  for (Handle &h : spacers) CK(hipMemRelease(h));
  return 0;
}
