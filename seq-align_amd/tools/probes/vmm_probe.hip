// vmm_probe.hip -- can the three matrix arenas be PLACED with the virtual-memory API instead of
// guessed at with spacer hipMallocs (sa_placement.hip, round 2)?
//
// Creates arena-sized physical handles (hipMemCreate) one after the other with unmapped spacer
// handles in between, maps the arena-sized ones, and measures with the fill's own store pattern
// (one region per wave, 1 KiB to each of K arenas in lock step) which combinations disturb each
// other.  Also times the API calls, and compares a VMM-mapped arena with a hipMalloc'ed one.
//
//   hipcc --offload-arch=gfx950 -O3 -o vmm_probe vmm_probe.hip
//   ./vmm_probe [depth_gib=140] [step_gib=2] [arena_bytes=912642048]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      printf("FAILED %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_));             \
      fflush(stdout);                                                                      \
      exit(2);                                                                             \
    }                                                                                      \
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

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
static float median_ms(F launch, int iters = 7) {
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
static uint32_t g_regions = 0;

static float t_streams(int k, void *a0, void *a1, void *a2) {
  return median_ms([&] {
    hipLaunchKernelGGL(probe_streams, dim3((g_regions + 3) / 4), dim3(256), 24576, 0, (char *)a0, (char *)a1, (char *)a2, k,
                       kRegionKib, g_regions);
  });
}
static float t_linear(void *a) {
  const uint64_t kib = (uint64_t)g_regions * kRegionKib;
  return median_ms([&] { hipLaunchKernelGGL(probe_linear, dim3((unsigned)((kib + 3) / 4)), dim3(256), 0, 0, (char *)a, kib); });
}

struct Phys {
  hipMemGenericAllocationHandle_t h;
  size_t bytes;
  void *va;   // nullptr: not mapped
};

static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_acc;

static bool phys_create(Phys &p, size_t bytes) {
  p.bytes = bytes;
  p.va = nullptr;
  hipError_t e = hipMemCreate(&p.h, bytes, &g_prop, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}
static void phys_map(Phys &p) {
  CK(hipMemAddressReserve(&p.va, p.bytes, 0, nullptr, 0));
  CK(hipMemMap(p.va, p.bytes, 0, p.h, 0));
  CK(hipMemSetAccess(p.va, p.bytes, &g_acc, 1));
}
static void phys_release(Phys &p) {
  if (p.va) {
    CK(hipMemUnmap(p.va, p.bytes));
    CK(hipMemAddressFree(p.va, p.bytes));
    p.va = nullptr;
  }
  CK(hipMemRelease(p.h));
}

int main(int argc, char **argv) {
  const size_t depth_gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 140;
  const size_t step_gib = argc > 2 ? strtoull(argv[2], nullptr, 10) : 2;
  size_t arena = argc > 3 ? strtoull(argv[3], nullptr, 10) : 912642048ull;
  CK(hipSetDevice(0));
  g_prop = {};
  g_prop.type = hipMemAllocationTypePinned;
  g_prop.location.type = hipMemLocationTypeDevice;
  g_prop.location.id = 0;
  g_acc.location = g_prop.location;
  g_acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &g_prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &g_prop, hipMemAllocationGranularityRecommended));
  size_t free_b = 0, total_b = 0;
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("granularity min %zu recommended %zu; free %.1f GiB of %.1f\n", gmin, grec, free_b / 1073741824.0,
         total_b / 1073741824.0);
  arena = (arena + grec - 1) / grec * grec;
  g_regions = (uint32_t)(arena / 1024 / kRegionKib);
  printf("arena %zu bytes, %u regions of %u KiB\n", arena, g_regions, kRegionKib);

  // ---- hipMalloc reference: three consecutive allocations
  {
    void *m[3];
    for (int k = 0; k < 3; ++k) CK(hipMalloc(&m[k], arena));
    const float l1 = t_linear(m[0]), s1 = t_streams(1, m[0], m[0], m[0]), s2 = t_streams(2, m[0], m[1], m[1]),
                s3 = t_streams(3, m[0], m[1], m[2]);
    printf("hipMalloc x3 back to back: linear1 %.4f  K1 %.4f  K2 %.4f  K3 %.4f ms  q3 = %.3f\n", l1, s1, s2, s3, 3 * l1 / s3);
    for (int k = 0; k < 3; ++k) CK(hipFree(m[k]));
  }

  // ---- API cost
  {
    Phys p;
    double t0 = now_ms();
    if (!phys_create(p, (size_t)4 << 30)) { printf("hipMemCreate 4 GiB failed\n"); return 2; }
    double t1 = now_ms();
    phys_map(p);
    double t2 = now_ms();
    phys_release(p);
    double t3 = now_ms();
    printf("4 GiB handle: create %.2f ms, reserve+map+access %.2f ms, unmap+release %.2f ms\n", t1 - t0, t2 - t1, t3 - t2);
  }

  // ---- the scan: M, A first; then spacer (unmapped) + candidate, every step_gib
  std::vector<Phys> cand, spacers;
  Phys M, A;
  if (!phys_create(M, arena) || !phys_create(A, arena)) return 2;
  phys_map(M);
  phys_map(A);
  const float lin = t_linear(M.va);
  printf("VMM M: linear1 %.4f ms, K1 %.4f; (M,A) K2 %.4f\n", lin, t_streams(1, M.va, M.va, M.va), t_streams(2, M.va, A.va, A.va));
  printf("depth_GiB  create_ms  map_ms   K3(M,A,Bj)  q3     K2(M,Bj)  q2\n");
  const size_t step = step_gib << 30;
  double held = 2.0 * arena;
  for (size_t d = 0; d <= depth_gib; d += step_gib) {
    double t0 = now_ms();
    if (d) {
      Phys sp;
      if (!phys_create(sp, step - arena)) { printf("spacer create failed at %zu GiB\n", d); break; }
      spacers.push_back(sp);
      held += step - arena;
    }
    Phys c;
    if (!phys_create(c, arena)) { printf("candidate create failed at %zu GiB\n", d); break; }
    double t1 = now_ms();
    phys_map(c);
    double t2 = now_ms();
    held += arena;
    cand.push_back(c);
    const float k3 = t_streams(3, M.va, A.va, c.va), k2 = t_streams(2, M.va, c.va, c.va);
    printf("%6zu     %7.2f   %7.2f   %.4f    %.3f   %.4f   %.3f\n", d, t1 - t0, t2 - t1, k3, 3 * lin / k3, k2, 2 * lin / k2);
    fflush(stdout);
  }
  printf("held %.1f GiB\n", held / 1073741824.0);

  // ---- pair structure among the candidates: is "disturb" an equivalence (same class) relation?
  const int n = (int)cand.size();
  printf("K2(cand i, cand j) ms, i rows / j cols, every 4th candidate\n      ");
  for (int j = 0; j < n; j += 4) printf("%5zu ", (size_t)j * step_gib);
  printf("\n");
  for (int i = 0; i < n; i += 4) {
    printf("%5zu ", (size_t)i * step_gib);
    for (int j = 0; j < n; j += 4) {
      if (j <= i) { printf("      "); continue; }
      printf("%.3f ", t_streams(2, cand[i].va, cand[j].va, cand[j].va));
    }
    printf("\n");
    fflush(stdout);
  }
  // three arenas in three different places vs two together + one apart
  if (n >= 30) {
    const int i0 = 0, i1 = n / 3, i2 = 2 * n / 3;
    printf("K3(c%d,c%d,c%d) %.4f   K3(c0,c1,c%d) %.4f   K3(c0,c1,c2) %.4f\n", i0, i1, i2,
           t_streams(3, cand[i0].va, cand[i1].va, cand[i2].va), i1, t_streams(3, cand[0].va, cand[1].va, cand[i1].va),
           t_streams(3, cand[0].va, cand[1].va, cand[2].va));
  }

  for (Phys &p : spacers) phys_release(p);
  double t0 = now_ms();
  for (Phys &p : cand) phys_release(p);
  printf("released %d mapped candidates in %.2f ms\n", n, now_ms() - t0);
  phys_release(M);
  phys_release(A);
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("free afterwards %.1f GiB\n", free_b / 1073741824.0);
  return 0;
}
