// d2h_probe.hip -- how the runtime moves 12 MB from device to pinned host memory, by how the copy is asked for: which
// calls go to the copy engines (SDMA) and which are run as a blit kernel over the chip, and at what rate.
//   hipcc --offload-arch=gfx950 -O3 -o d2h_probe d2h_probe.hip && ./d2h_probe        (rocprofv3 --kernel-trace --memory-copy-trace to see which)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void touch(uint32_t *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t bytes = 12u << 20;
  void *dev;
  CHECK(hipMalloc(&dev, bytes));
  hipStream_t s_kernel, s_copy;
  CHECK(hipStreamCreateWithFlags(&s_kernel, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
  hipEvent_t ev;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  struct { const char *name; unsigned flags; } kinds[] = {
      {"hipHostMallocDefault", hipHostMallocDefault}, {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
      {"hipHostMallocCoherent", hipHostMallocCoherent}, {"hipHostMallocNumaUser", hipHostMallocNumaUser},
      {"hipHostMallocPortable|Mapped", hipHostMallocPortable | hipHostMallocMapped}};
  for (auto &k : kinds) {
    void *host = nullptr;
    if (hipHostMalloc(&host, bytes, k.flags) != hipSuccess) { printf("%-30s alloc failed\n", k.name); (void)hipGetLastError(); continue; }
    for (int mode = 0; mode < 4; ++mode) {
      // 0: copy stream that only ever copies, after an event of a kernel on another stream; 1: same stream as the kernel;
      // 2: hipMemcpyDtoHAsync on the copy stream; 3: hipMemcpy (synchronous)
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(touch, dim3((bytes / 4 + 255) / 256), dim3(256), 0, s_kernel, (uint32_t *)dev, bytes / 4);
        CHECK(hipEventRecord(ev, s_kernel));
        CHECK(hipStreamWaitEvent(s_copy, ev, 0));
        CHECK(hipStreamSynchronize(s_kernel));
        const double t0 = now_us();
        if (mode == 0) { CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s_copy)); CHECK(hipStreamSynchronize(s_copy)); }
        else if (mode == 1) { CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s_kernel)); CHECK(hipStreamSynchronize(s_kernel)); }
        else if (mode == 2) { CHECK(hipMemcpyDtoHAsync(host, (hipDeviceptr_t)dev, bytes, s_copy)); CHECK(hipStreamSynchronize(s_copy)); }
        else { CHECK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost)); }
        const double t = now_us() - t0;
        if (t < best) best = t;
      }
      printf("%-30s mode %d: %8.1f us  %5.1f GB/s\n", k.name, mode, best, bytes / best / 1e3);
    }
    CHECK(hipHostFree(host));
  }
  // H2D for comparison
  void *host;
  CHECK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    const double t0 = now_us();
    CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s_copy)); CHECK(hipStreamSynchronize(s_copy));
    const double t = now_us() - t0; if (t < best) best = t;
  }
  printf("%-30s H2D   : %8.1f us  %5.1f GB/s\n", "hipHostMallocDefault", best, bytes / best / 1e3);
  return 0;
}
