// sa_placement.hip -- where the three output arenas live in HBM.
//
// Replaces nothing in the reference (its three matrices are three malloc()s, src/alignment.c:183-190); this is
// the MI355X side of "memory laid out for 288 GB of HBM3E".
//
// What is being dodged (tools/probes/{stream,vmm,skew,stripe}_probe.hip; profiles/r01_*placement*, r03_*probe*):
// the fill kernels write M, A and B concurrently as thousands of sequential streams (one per wave and matrix).
// That pattern runs at ~5 TB/s while everything it writes lies in ONE class of physical memory, and at the speed
// of a linear memset (6.6-6.9 TB/s) when one of the three arenas lies in another class.  A class is a property
// of the physical placement only: no low address bit matters (any skew from 1 KiB to 64 MiB between the arenas,
// any distance inside one allocation, any virtual address: the same), classes come as blocks of tens of GiB,
// and a single linear stream never notices -- consistent with bank / rank parallelism (more classes in flight =
// more open DRAM rows), which user space can neither see nor ask for.
//
// Round 2 guessed at the placement with spacer hipMallocs, four tries; a fresh box where the guess failed four
// times ran the headline kernel 18 % slower.  Why it could fail (stripe_probe): the VRAM manager is a buddy
// allocator -- a request is served from the free lists of the block sizes it decomposes into, so after an arena
// of 870 MiB the next one lands in the power-of-two remainders next to it no matter how much was allocated "in
// between", and a big spacer is taken from another list altogether.  Allocation order says little about where
// memory is unless every request has the SAME power-of-two size.
//
// So the arenas are built with the virtual-memory API from uniform power-of-two chunks:
//   * hipMemCreate hands out physical chunks (512 MiB; one buddy block each) without mapping them: creating
//     and releasing a chunk costs ~10 us, so holding a hundred GiB of them for a fraction of a second is free;
//   * M and A take the first chunks; the rest of the pool is walked in allocation order, every `step` chunks a
//     window of chunks is mapped as a candidate B and the fill's own store pattern is timed on (M, A, candidate)
//     against one linear stream over the same bytes (quality = 3 t1 / t3: ~0.75 when all three disturb each
//     other, >= 1.0 when they do not);
//   * the first candidate at or above `quality_stop` ends the walk, otherwise the best one seen is kept; every
//     other chunk goes straight back.  (Candidates come in grades: ~0.80 same class; 0.98-1.03 another class -- the
//     headline kernel at 0.412-0.418 ms; 1.045-1.055 the best there is -- 0.399-0.402 ms, what a memset of the same
//     bytes takes; about every second box has such memory within the first 160 GiB.  The default stop is 1.045.)  The result, the number of candidates tried and their qualities are
//     reported (seqalign_arenas_info, bench.py prints them): a placement below target is SAID, never silent.
// The walk is bounded by `scan_bytes` and by 60 % of what is free; an arena set that cannot be placed that way
// (tiny arenas, no VMM support, no room) is three plain hipMallocs and reports quality -1 / its probe value.
//
// A VIRTUAL ADDRESS IS NEVER MAPPED TWICE.  On this stack (ROCm 7.2, gfx950) hipMemUnmap does not invalidate the
// GPU's translations: after unmap -> map of other chunks at the same address, kernels kept reading and writing the
// OLD physical memory -- already released, possibly somebody else's (tools/probes/vmm_reuse_probe.hip: 7 of 8
// re-mapped ranges held the wrong data; with every mapping at a never-used address, 0 of 8; a hipMalloc + hipFree
// between unmap and map also cured it, i.e. the classic path flushes and the VMM path does not).  So every
// mapping made here gets its own address reservation, and reservations are never given back to the driver
// (hipMemAddressFree would let it hand the range out again): address space is the one resource a 64-bit process
// has plenty of -- a C2-sized walk uses <= 42 GiB of it, a context allocates its arenas a few times in its life --
// and past 32 TiB of retired ranges the allocator simply stops placing and allocates plainly.
//
// THE CHUNK POOL (round 5).  The chunks a walk created and did not use used to go straight back to the driver, which
// clears released VRAM before it hands it out again: after a walk that had used its whole budget (160 GiB) the process's
// next LARGE allocation waited for that -- the first seqalign_nw_batch of C5's share (2.9 GB of direction bytes) 3.7 s,
// C3's first seqalign_sw_batch 4.6 s (profiles/r04/r04_bench_C{3,5}.json: first_call_ms).  Now up to `keep_bytes`
// (option arena_keep_gib, default 16) of those chunks stay with the process, in a per-device pool, and the contexts'
// large scratch buffers (DevBuf: direction bytes, staging; the unplaced arena sets) are mapped from it (sa_pool_alloc):
// memory that never went back needs no clearing.  Buffers freed return their chunks to the pool; the pool is emptied
// when the last context of the device is destroyed (sa_pool_unref) or on seqalign_pool_trim.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <chrono>
#include <mutex>
#include <vector>

#include "sa_kernels.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef hipMemGenericAllocationHandle_t Handle;

// the stream kernel's write pattern without the arithmetic: wave w owns region w of each arena and appends
// 1 KiB blocks to the three of them in lock step; regions are `stride_kib` apart (>= region_kib), so the probe
// samples the whole arena when it is larger than n_regions * region_kib
__global__ void __launch_bounds__(256) probe_three_streams(char *a0, char *a1, char *a2, uint32_t region_kib,
                                                           uint64_t stride_kib, uint32_t n_regions) {
  extern __shared__ int occupancy_pad[];   // 24 KiB per workgroup, like fill_stream_kernel<3,...>
  const int lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= n_regions) return;
  const v4i val = {(int)w, lane, 0, 0};
  const uint64_t base = (uint64_t)w * stride_kib * 1024 + lane * 16;
  for (uint32_t k = 0; k < region_kib; ++k) {
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a0 + base + (uint64_t)k * 1024));
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a1 + base + (uint64_t)k * 1024));
    __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a2 + base + (uint64_t)k * 1024));
  }
}

// baseline: one sequential stream over as many bytes as ONE arena gets in the probe above, short-lived
// workgroups, 4 KiB each
__global__ void __launch_bounds__(256) probe_one_stream(char *a, uint64_t total_kib) {
  const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= total_kib) return;
  const v4i val = {(int)blockIdx.x, 0, 0, 0};
  __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a + b * 1024 + (threadIdx.x & 63) * 16));
}

template <class F>
float median_ms(hipStream_t st, int iters, F launch) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
  std::vector<float> t;
  for (int it = 0; it < iters; ++it) {
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

constexpr uint32_t kRegionKib = 88;        // a 150 x 150 pair's matrix is 89 KiB
constexpr uint32_t kProbeRegions = 10240;  // as many waves as a C2 launch has pairs

struct ProbeShape {
  uint32_t n_regions = 0;
  uint64_t stride_kib = 0;
  bool ok() const { return n_regions >= 2048; }
};
ProbeShape probe_shape(size_t bytes) {
  ProbeShape s;
  const uint64_t kib = bytes / 1024;
  s.n_regions = (uint32_t)std::min<uint64_t>(kProbeRegions, kib / kRegionKib);
  if (s.n_regions) s.stride_kib = std::max<uint64_t>(kRegionKib, kib / s.n_regions);
  return s;
}
float time_three(void *const a[3], const ProbeShape &s, hipStream_t st, int iters) {
  return median_ms(st, iters, [&] {
    hipLaunchKernelGGL(probe_three_streams, dim3((s.n_regions + 3) / 4), dim3(256), 24576, st, (char *)a[0], (char *)a[1],
                       (char *)a[2], kRegionKib, s.stride_kib, s.n_regions);
  });
}
float time_one(void *a, const ProbeShape &s, hipStream_t st) {
  const uint64_t kib = (uint64_t)s.n_regions * kRegionKib;   // contiguous from the arena's start: fits, stride >= region
  return median_ms(st, 6, [&] { hipLaunchKernelGGL(probe_one_stream, dim3((unsigned)((kib + 3) / 4)), dim3(256), 0, st, (char *)a, kib); });
}

// ------------------------------------------------------------------ chunks ---
struct VmmEnv {
  hipMemAllocationProp prop;
  hipMemAccessDesc access;
  bool ok = false;
};
VmmEnv vmm_env(int device) {
  VmmEnv v;
  v.prop = {};
  v.prop.type = hipMemAllocationTypePinned;
  v.prop.location.type = hipMemLocationTypeDevice;
  v.prop.location.id = device;
  v.access.location = v.prop.location;
  v.access.flags = hipMemAccessFlagsProtReadWrite;
  int supported = 0;
  if (hipDeviceGetAttribute(&supported, hipDeviceAttributeVirtualMemoryManagementSupported, device) != hipSuccess) supported = 0;
  size_t gran = 0;
  v.ok = supported && hipMemGetAllocationGranularity(&gran, &v.prop, hipMemAllocationGranularityRecommended) == hipSuccess;
  (void)hipGetLastError();
  return v;
}

std::atomic<unsigned long long> g_retired_va{0};   // address space of mappings taken down (never handed back, see header)
constexpr unsigned long long kRetiredVaLimit = 32ull << 40;

// a run of chunks mapped back to back at one virtual address range that is used for this and nothing else, ever
struct Mapping {
  char *va = nullptr;
  size_t bytes = 0, chunk = 0, mapped = 0;
  hipError_t map(const VmmEnv &env, const Handle *hs, size_t n, size_t chunk_bytes) {
    chunk = chunk_bytes;
    bytes = n * chunk;
    void *p = nullptr;
    hipError_t e = hipMemAddressReserve(&p, bytes, 0, nullptr, 0);
    if (e != hipSuccess) return e;
    va = static_cast<char *>(p);
    for (size_t i = 0; i < n; ++i) {
      if ((e = hipMemMap(va + i * chunk, chunk, 0, hs[i], 0)) != hipSuccess) { unmap(); return e; }
      mapped = i + 1;
    }
    if ((e = hipMemSetAccess(va, bytes, &env.access, 1)) != hipSuccess) { unmap(); return e; }
    return hipSuccess;
  }
  void unmap() {
    if (!va) return;
    for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap(va + i * chunk, chunk);
    g_retired_va += bytes;   // NOT hipMemAddressFree: this range must never be mapped again (header)
    va = nullptr; mapped = 0;
  }
};


// ------------------------------------------------------------------ the chunk pool ---
struct ChunkPool {
  std::mutex mu;
  std::vector<Handle> free;      // 512 MiB chunks kept from walks (and from pool buffers freed)
  size_t cap_chunks = 0;         // what may stay (set by the walk that feeds it: option arena_keep_gib)
  int contexts = 0;              // live contexts of the device (sa_pool_ref / sa_pool_unref)
};
ChunkPool g_pool[64];
constexpr size_t kPoolChunk = (size_t)512 << 20;

struct PoolBuf { Mapping map; std::vector<Handle> handles; int device; };
std::mutex g_pool_bufs_mu;
std::map<void *, PoolBuf *> g_pool_bufs;

// hand `h` to the pool, or back to the driver when the pool is full
void pool_put(int device, Handle h) {
  ChunkPool &p = g_pool[device];
  {
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.free.size() < p.cap_chunks) { p.free.push_back(h); return; }
  }
  (void)hipMemRelease(h);
}

}  // namespace

void *sa_pool_alloc(int device, size_t bytes) {
  if (device < 0 || device >= 64 || !bytes || g_retired_va.load() >= kRetiredVaLimit) return nullptr;
  const size_t n = (bytes + kPoolChunk - 1) / kPoolChunk;
  ChunkPool &p = g_pool[device];
  PoolBuf *b = new (std::nothrow) PoolBuf();
  if (!b) return nullptr;
  {
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.free.size() < n) { delete b; return nullptr; }
    b->handles.assign(p.free.end() - n, p.free.end());
    p.free.resize(p.free.size() - n);
  }
  b->device = device;
  const VmmEnv env = vmm_env(device);
  if (!env.ok || b->map.map(env, b->handles.data(), n, kPoolChunk) != hipSuccess) {
    (void)hipGetLastError();
    for (Handle h : b->handles) pool_put(device, h);
    delete b;
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_pool_bufs_mu);
  g_pool_bufs[b->map.va] = b;
  return b->map.va;
}

bool sa_pool_free(void *ptr) {
  PoolBuf *b = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pool_bufs_mu);
    auto it = g_pool_bufs.find(ptr);
    if (it == g_pool_bufs.end()) return false;
    b = it->second;
    g_pool_bufs.erase(it);
  }
  // hipFree waits for the device; hipMemUnmap + handing the chunks to the next mapping is not documented to.  Callers have
  // synchronised their own stream by now (DevBuf::reserve runs between calls, reserve_arenas syncs), but a kernel of ANOTHER
  // context's stream could still be writing through a view of this buffer only if its owner released it mid-flight -- make the
  // precondition hold here instead of at every caller: freeing a >= 128 MiB buffer is rare and never on a timed path.
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (cur != b->device) (void)hipSetDevice(b->device);
  (void)hipDeviceSynchronize();
  if (cur >= 0 && cur != b->device) (void)hipSetDevice(cur);
  b->map.unmap();
  for (Handle h : b->handles) pool_put(b->device, h);
  delete b;
  return true;
}

size_t sa_pool_bytes(int device) {
  if (device < 0 || device >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_pool[device].mu);
  return g_pool[device].free.size() * kPoolChunk;
}

void sa_pool_trim(int device, size_t keep_bytes) {
  if (device < 0 || device >= 64) return;
  std::vector<Handle> out;
  {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    std::vector<Handle> &f = g_pool[device].free;
    const size_t keep = keep_bytes / kPoolChunk;
    g_pool[device].cap_chunks = std::min(g_pool[device].cap_chunks, keep);
    while (f.size() > keep) { out.push_back(f.back()); f.pop_back(); }
  }
  for (Handle h : out) (void)hipMemRelease(h);
}

void sa_pool_ref(int device) {
  if (device < 0 || device >= 64) return;
  std::lock_guard<std::mutex> lk(g_pool[device].mu);
  g_pool[device].contexts++;
}
void sa_pool_unref(int device) {
  if (device < 0 || device >= 64) return;
  bool last;
  {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    last = --g_pool[device].contexts <= 0;
  }
  if (last) sa_pool_trim(device, 0);   // nobody left to use it: the memory goes back
}

namespace {
}  // namespace

// what seqalign_arenas_alloc hands out, and what seqalign_arenas_free needs to take it back
struct SaArenaSet {
  void *base[3] = {nullptr, nullptr, nullptr};
  size_t bytes = 0;                 // usable bytes per arena
  bool vmm = false;
  int device = 0;
  Mapping map[3];
  std::vector<Handle> handles[3];
  SaArenaInfo info;
};

namespace {

std::mutex g_sets_mu;
std::map<void *, SaArenaSet *> g_sets;   // by base[0]

// One walk per device at a time, process-wide: a walk sizes itself from the memory that is free when it starts and holds
// tens of GiB of chunks for a moment -- two of them at once (two contexts, two threads of the legacy API) would each claim
// their share of the SAME free memory and could drive the device out of memory under a third party's hipMalloc.
constexpr int kMaxDevices = 64;
std::mutex g_walk_mu[kMaxDevices];

void register_set(SaArenaSet *s) {
  std::lock_guard<std::mutex> lk(g_sets_mu);
  g_sets[s->base[0]] = s;
}
SaArenaSet *take_set(void *base0) {
  std::lock_guard<std::mutex> lk(g_sets_mu);
  auto it = g_sets.find(base0);
  if (it == g_sets.end()) return nullptr;
  SaArenaSet *s = it->second;
  g_sets.erase(it);
  return s;
}

float quality_of(void *const a[3], size_t bytes, hipStream_t st, int iters = 5) {
  const ProbeShape s = probe_shape(bytes);
  if (!s.ok()) return -1.f;
  const float t3 = time_three(a, s, st, iters), t1 = time_one(a[0], s, st);
  if (t3 <= 0 || t1 <= 0) return -1.f;
  return 3.f * t1 / t3;
}

hipError_t plain_set(int device, size_t bytes, hipStream_t st, bool probe, SaArenaSet *s) {
  hipError_t e = hipSuccess;
  for (int k = 0; k < 3 && e == hipSuccess; ++k) {
    // (large ones from the process's chunk pool when it has them: no freshly released VRAM for the driver to clear)
    if (bytes >= ((size_t)128 << 20) && (s->base[k] = sa_pool_alloc(device, bytes))) continue;
    e = hipMalloc(&s->base[k], bytes);
  }
  if (e != hipSuccess) {
    for (int k = 0; k < 3; ++k) { if (s->base[k] && !sa_pool_free(s->base[k])) (void)hipFree(s->base[k]); s->base[k] = nullptr; }
    return e;
  }
  s->bytes = bytes;
  s->vmm = false;
  s->info.quality = probe ? quality_of(s->base, bytes, st) : -1.f;
  return hipSuccess;
}

}  // namespace

static hipError_t arenas_create_untimed(int device, size_t bytes, hipStream_t stream, const SaPlacementOpts &opt, SaArenaSet **out) {
  *out = nullptr;
  SaArenaSet *s = new (std::nothrow) SaArenaSet();
  if (!s) return hipErrorOutOfMemory;
  s->info = SaArenaInfo();
  s->device = device;
  s->info.quality = -1.f;
  s->info.depth_a_gib = -1.f;
  s->info.target = opt.quality_stop;
  const size_t chunk = (size_t)512 << 20;
  size_t free_b = 0, total_b = 0;
  const VmmEnv env = vmm_env(device);
  std::unique_lock<std::mutex> one_walk;
  if (opt.scan_bytes && device >= 0 && device < kMaxDevices) one_walk = std::unique_lock<std::mutex>(g_walk_mu[device]);
  // small arenas are not bandwidth-bound; without the VMM API, or without room to look around, allocate plainly
  const bool place = opt.scan_bytes && device >= 0 && device < 64 && bytes >= ((size_t)256 << 20) && env.ok && g_retired_va.load() < kRetiredVaLimit &&
                     hipMemGetInfo(&free_b, &total_b) == hipSuccess;
  const size_t per = (bytes + chunk - 1) / chunk;                       // chunks per arena
  // what the walk may hold at its peak: the arenas + scan_bytes, and never more than `free_fraction` of what is free NOW --
  // re-checked as the pool grows (grow_to), because other tenants of the device keep allocating while we look around
  const double frac = opt.free_fraction > 0 ? opt.free_fraction : 0.6;
  const size_t budget = place ? std::min<size_t>(opt.scan_bytes + 3 * per * chunk, (size_t)(free_b * frac)) : 0;
  const size_t pool_max = budget / chunk;
  const size_t keep_free = place ? (size_t)(free_b * (1.0 - frac)) : 0;   // what must stay free for everybody else
  // the way out whenever placing is not possible (any more): three plain allocations, probed if large enough to matter
  auto plain = [&]() -> hipError_t {
    (void)hipGetLastError();
    // (probed when a walk had been asked for and could not be made; a caller that asked for plain arenas -- scan_bytes = 0: its
    // kernels are not HBM-bound -- is not made to wait 3 ms for a figure it has no use for)
    const hipError_t pe = plain_set(device, bytes, stream, opt.scan_bytes && bytes >= ((size_t)256 << 20), s);
    if (pe != hipSuccess) { delete s; return pe; }
    register_set(s);
    *out = s;
    return hipSuccess;
  };
  if (!place || pool_max < 3 * per) return plain();
  {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    g_pool[device].cap_chunks = std::max(g_pool[device].cap_chunks, opt.keep_bytes / kPoolChunk);
  }

  // ---- the pool: uniform chunks in allocation order; [0, per) = M, [per, 2 per) = A
  std::vector<Handle> pool;
  pool.reserve(pool_max);
  auto grow_to = [&](size_t n) {   // false: the device has no more to give (another tenant): work with what we have
    while (pool.size() < n) {
      if (pool.size() % 16 == 15) {   // every 8 GiB: is the device still as empty as the budget assumed?
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) == hipSuccess && f < keep_free) return false;
      }
      Handle h;
      if (hipMemCreate(&h, chunk, &env.prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
      pool.push_back(h);
    }
    return true;
  };
  auto release_from = [&](size_t keep_first, const std::vector<size_t> &keep_runs) {
    // release every chunk except [0, keep_first) and the runs [r, r + per) listed
    for (size_t i = keep_first; i < pool.size(); ++i) {
      bool keep = false;
      for (size_t r : keep_runs) keep = keep || (i >= r && i < r + per);
      if (!keep) pool_put(device, pool[i]);   // the first `keep_bytes` stay with the process (the chunk pool), the rest goes back
    }
    s->info.kept_gib = (float)((double)sa_pool_bytes(device) / 1073741824.0);
  };
  if (!grow_to(3 * per)) {
    release_from(0, {});
    return plain();
  }
  Mapping m, a;
  if (m.map(env, pool.data(), per, chunk) != hipSuccess || a.map(env, pool.data() + per, per, chunk) != hipSuccess) {
    m.unmap(); a.unmap();
    release_from(0, {});
    return plain();
  }
  const ProbeShape shape = probe_shape(bytes);
  const float t1 = time_one(m.va, shape, stream);

  // ---- two walks over the pool, candidates `step` chunks apart (8 chunks = 4 GiB, or back to back if the arenas are
  // larger than that):
  //   walk 1: B = pool[pos, pos + per) with A right behind M -- finds memory of another class than M's;
  //   walk 2: A = pool[pos, pos + per) with that B -- finds a THIRD class.  tools/probes/grade_probe.hip times the whole
  //           (A, B) grid: one class 0.52 ms, two classes 0.403-0.416, three classes 0.387-0.395 -- the "best grade",
  //           what a memset of the same bytes takes (profiles/r03/r03_grade_probe.txt).
  //   walk 3 (below, round 4): M = pool[pos, pos + per) with that A and B, only when neither of them reached the target.
  // Every walk ends early at `quality_stop` (a three-class placement; a later walk is skipped when an earlier one got there).
  const size_t step = std::max<size_t>(per, 8);
  auto probe = [&](void *x0, void *x1, void *x2) {
    void *trio[3] = {x0, x1, x2};
    float t3 = time_three(trio, shape, stream, 4);
    float q = (t3 > 0 && t1 > 0) ? 3.f * t1 / t3 : -1.f;
    if (q >= 1.02f) {   // a contender: once more with more repetitions (what separates the grades is 3 %)
      t3 = time_three(trio, shape, stream, 9);
      if (t3 > 0) q = 3.f * t1 / t3;
    }
    if (q >= opt.quality_stop) {
      // about to end the walk: measured from scratch, the one-stream time included -- t1 is in every quality of this walk, and a
      // t1 that came out 1 % long makes a second-class candidate look like the best grade -- and TWICE: round 4 accepted when
      // the mean of the walk's figure and one re-measurement reached the target, and two processes of six in a row on one box
      // (round 5, profiles/r05/r05_placement_repeat.txt) ended their walk at "1.045" / "1.046" on candidates that reported 1.003
      // / 1.030 afterwards (the headline kernel at 0.834 / 0.838 instead of 0.85).  Both re-measurements must confirm; a
      // candidate that does not keeps the lowest figure seen and the walk goes on.
      const float a1 = quality_of(trio, bytes, stream, 7), a2 = quality_of(trio, bytes, stream, 7);
      if (a1 >= 0 && a2 >= 0 && std::min(a1, a2) < opt.quality_stop) q = std::min(q, std::min(a1, a2));
    }
    return q;
  };
  auto record = [&](float q, size_t pos) {
    if (s->info.tries >= SA_ARENA_MAX_TRIES) return;
    s->info.try_quality[s->info.tries] = q;
    s->info.try_depth_gib[s->info.tries] = (float)((double)(pos - 2 * per) * chunk / 1073741824.0);
    s->info.tries++;
  };
  Mapping best_map;
  size_t best_pos = 0;
  float best_q = -2.f;
  for (size_t pos = 2 * per; pos + per <= pool_max && s->info.tries < SA_ARENA_MAX_TRIES / 2; pos += step) {
    if (!grow_to(pos + per)) break;
    Mapping c;
    if (c.map(env, pool.data() + pos, per, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
    const float q = probe(m.va, a.va, c.va);
    record(q, pos);
    if (q > best_q) {
      best_map.unmap();
      best_map = c; best_pos = pos; best_q = q;
    } else {
      c.unmap();
    }
    if (q < 0 || q >= opt.quality_stop) break;   // probe unavailable, or as good as it gets
  }
  s->info.scanned_gib = (float)((double)pool.size() * chunk / 1073741824.0);
  if (!best_map.va) {   // not even one candidate: B right behind A
    if (grow_to(3 * per) && best_map.map(env, pool.data() + 2 * per, per, chunk) == hipSuccess) best_pos = 2 * per;
  }
  if (!best_map.va) {
    m.unmap(); a.unmap();
    release_from(0, {});
    return plain();
  }
  size_t a_pos = per;   // where A's chunks are in the pool
  s->info.second_walk_from = s->info.tries;
  if (best_q >= 0 && best_q < opt.quality_stop) {
    float best_qa = best_q;
    for (size_t pos = 2 * per; pos + per <= pool.size() && s->info.tries < SA_ARENA_MAX_TRIES; pos += step) {
      if (pos + per > best_pos && pos < best_pos + per) continue;   // B's own chunks
      Mapping c;
      if (c.map(env, pool.data() + pos, per, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
      const float q = probe(m.va, c.va, best_map.va);
      record(q, pos);
      if (q > best_qa + 0.01f) {   // (worth moving for: more than the probe's noise)
        a.unmap();
        a = c; a_pos = pos; best_qa = q;
      } else {
        c.unmap();
      }
      if (q < 0 || q >= opt.quality_stop) break;
    }
    best_q = best_qa;
  }
  // walk 3 (round 4): neither B nor A found the target -- then it may be M that sits badly (what a process is handed first
  // differs from process to process on one box: six processes in a row ended at 1.026 ... 1.055, r04_placement_budget.txt).
  // M = pool[pos, pos + per) against that A and B, in as many steps as the record has room for.
  size_t m_pos = 0;
  if (best_q >= 0 && best_q < opt.quality_stop && s->info.tries + 4 <= SA_ARENA_MAX_TRIES) {
    const size_t left = SA_ARENA_MAX_TRIES - s->info.tries;
    const size_t span = pool.size() > 3 * per ? pool.size() - 3 * per : 0;
    const size_t step3 = std::max(step, (span + left - 1) / left);
    float best_qm = best_q;
    for (size_t pos = 2 * per; pos + per <= pool.size() && s->info.tries < SA_ARENA_MAX_TRIES; pos += step3) {
      if ((pos + per > best_pos && pos < best_pos + per) || (pos + per > a_pos && pos < a_pos + per)) continue;   // B's, A's own chunks
      Mapping c;
      if (c.map(env, pool.data() + pos, per, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
      const float q = probe(c.va, a.va, best_map.va);
      record(q, pos);
      if (q > best_qm + 0.01f) {
        m.unmap();
        m = c; m_pos = pos; best_qm = q;
      } else {
        c.unmap();
      }
      if (q < 0 || q >= opt.quality_stop) break;
    }
    best_q = best_qm;
  }
  release_from(0, {m_pos, a_pos, best_pos});
  (void)hipGetLastError();   // a failed create / map of the walks must not surface as the next launch's error
  s->vmm = true;
  s->bytes = per * chunk;
  s->map[0] = m; s->map[1] = a; s->map[2] = best_map;
  s->handles[0].assign(pool.begin() + m_pos, pool.begin() + m_pos + per);
  s->handles[1].assign(pool.begin() + a_pos, pool.begin() + a_pos + per);
  s->handles[2].assign(pool.begin() + best_pos, pool.begin() + best_pos + per);
  s->info.depth_a_gib = a_pos == per ? -1.f : (float)((double)(a_pos - 2 * per) * chunk / 1073741824.0);
  for (int k = 0; k < 3; ++k) s->base[k] = s->map[k].va;
  s->info.vmm = 1;
  s->info.chunk_mib = (uint32_t)(chunk >> 20);
  s->info.depth_gib = (float)((double)(best_pos - 2 * per) * chunk / 1073741824.0);
  // the figure that is reported: measured once more, on the final three, with more repetitions
  s->info.quality = quality_of(s->base, bytes, stream, 7);
  if (s->info.quality < 0) s->info.quality = best_q;
  register_set(s);
  *out = s;
  return hipSuccess;
}

// What the placement costs its caller, stated with the result (the bench line quotes a fraction that depends on a
// walk -- its wall clock belongs beside it): chunk creation, mappings, every timed candidate, the releases.
hipError_t sa_arenas_create(int device, size_t bytes, hipStream_t stream, const SaPlacementOpts &opt, SaArenaSet **out) {
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = arenas_create_untimed(device, bytes, stream, opt, out);
  if (e == hipSuccess && *out)
    (*out)->info.seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
  return e;
}

void sa_arenas_destroy(SaArenaSet *s) {
  if (!s) return;
  if (s->vmm) {
    for (int k = 0; k < 3; ++k) {
      s->map[k].unmap();
      for (Handle h : s->handles[k]) pool_put(s->device, h);
    }
  } else {
    for (int k = 0; k < 3; ++k) if (s->base[k] && !sa_pool_free(s->base[k])) (void)hipFree(s->base[k]);
  }
  delete s;
}

SaArenaSet *sa_arenas_take(void *base0) { return take_set(base0); }
const SaArenaInfo *sa_arenas_info(const SaArenaSet *s) { return &s->info; }
void *const *sa_arenas_base(const SaArenaSet *s) { return s->base; }
size_t sa_arenas_bytes(const SaArenaSet *s) { return s->bytes; }
bool sa_arenas_copy_info(void *base0, SaArenaInfo *out) {   // (copied under the registry's lock: the set may be freed by another thread)
  std::lock_guard<std::mutex> lk(g_sets_mu);
  auto it = g_sets.find(base0);
  if (it == g_sets.end()) return false;
  *out = it->second->info;
  return true;
}
