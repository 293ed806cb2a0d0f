// grade_probe.hip -- does moving the SECOND arena as well find better placements than walking only the third?
//
// sa_placement.hip keeps M and A at the start of the chunk pool and walks candidates for B.  The headline kernel runs at
// 0.399-0.404 ms when the walk finds a candidate of quality >= 1.04 and at 0.412-0.418 ms when the best is ~1.0-1.03;
// about every second box has no such candidate within 160 GiB.  This probe times K3(M, A_i, B_j) over a grid of
// positions of BOTH (uniform 512 MiB chunks, allocation order), to see whether some (i, j) beats every (0, j).
//
//   ./grade_probe [pool_gib=160] [grid_gib=8]
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
static float K3(void *a0, void *a1, void *a2) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe_streams, dim3((kRegions + 3) / 4), dim3(256), 24576, 0, (char *)a0, (char *)a1, (char *)a2, kRegionKib, kRegions);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) t.push_back(ms);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char **argv) {
  const size_t pool_gib = argc > 1 ? atoi(argv[1]) : 160, grid_gib = argc > 2 ? atoi(argv[2]) : 8;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t chunk = (size_t)512 << 20, per = 2, n_chunks = pool_gib * 2;
  std::vector<hipMemGenericAllocationHandle_t> pool(n_chunks);
  for (auto &h : pool) CK(hipMemCreate(&h, chunk, &prop, 0));
  // map an arena (2 chunks) at every grid position, each at its own never-reused address
  const size_t step = grid_gib * 2;
  std::vector<char *> va;
  for (size_t pos = 0; pos + per <= n_chunks; pos += step) {
    void *p = nullptr;
    CK(hipMemAddressReserve(&p, per * chunk, 0, nullptr, 0));
    for (size_t k = 0; k < per; ++k) CK(hipMemMap((char *)p + k * chunk, chunk, 0, pool[pos + k], 0));
    CK(hipMemSetAccess(p, per * chunk, &acc, 1));
    va.push_back((char *)p);
  }
  const int n = (int)va.size();
  // M = position 0 (and, second table, M = the second arena of the pool); rows: A at i, columns: B at j
  void *m2 = nullptr;
  CK(hipMemAddressReserve(&m2, per * chunk, 0, nullptr, 0));
  for (size_t k = 0; k < per; ++k) CK(hipMemMap((char *)m2 + k * chunk, chunk, 0, pool[per + k], 0));
  CK(hipMemSetAccess(m2, per * chunk, &acc, 1));
  printf("K3(M = arena at 0, A = row, B = column) in us; positions in GiB\n      ");
  for (int j = 1; j < n; ++j) printf("%5zu ", j * grid_gib);
  printf("\n");
  float best = 1e9f, best_row0 = 1e9f; int bi = 0, bj = 0;
  for (int i = 0; i < n; ++i) {
    printf("%5s ", i == 0 ? "adj" : "");
    if (i) printf("\b\b\b\b\b\b%5zu ", i * grid_gib);
    for (int j = 1; j < n; ++j) {
      if (j == i) { printf("      "); continue; }
      const float t = K3(va[0], i == 0 ? (char *)m2 : va[i], va[j]) * 1000.f;
      printf("%5.0f ", t);
      if (t < best) { best = t; bi = i; bj = j; }
      if (i == 0 && t < best_row0) best_row0 = t;
    }
    printf("\n");
    fflush(stdout);
  }
  printf("best with A next to M (what the library walks): %.0f us; best over the grid: %.0f us at A = %zu GiB, B = %zu GiB\n",
         best_row0, best, bi * grid_gib, bj * grid_gib);
  return 0;
}
