// sa_ctx.hpp -- what the translation units of the C-ABI shim share: the context, its grow-only
// device / pinned buffers, the host worker pool and the chunk machinery of the host-level entry
// points.  Internal: nothing here is part of include/seqalign_hip.h.
//   sa_device.hip     context, scoring upload, the device-level entry points (the hot path), legacy
//                     single-pair call
//   sa_batch.hip      host-level chunking, seqalign_fill_batch, seqalign_nw_batch
//   sa_batch_sw.hip   seqalign_sw_batch (best hit / device enumeration / host enumeration)
//   sa_multi.hip      the same calls over several contexts (GPUs) from one process
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "sa_kernels.h"

extern "C" {
#include "sa_internal.h"
}

namespace sa_host {


// Persistent host worker pool: run fn(0..n-1) over the workers + the caller.
// Pairs / memcpy pieces are independent.  SEQALIGN_HOST_THREADS overrides the
// worker count (default min(CPUs this process may run on, 32): a rank pinned to its
// GPU's NUMA node gets a pool of that size, on those CPUs -- the workers inherit
// the creating thread's affinity mask).  One job at a time.
class HostPool {
 public:
  static HostPool &get() { static HostPool pool; return pool; }
  unsigned threads() const { return (unsigned)workers_.size() + 1; }
  void run(uint64_t n, const std::function<void(uint64_t)> &fn) {
    if (n == 0) return;
    if (workers_.empty() || n == 1) { for (uint64_t k = 0; k < n; ++k) fn(k); return; }
    std::lock_guard<std::mutex> one_job(job_mu_);
    fn_ = &fn; n_ = n; next_.store(0, std::memory_order_relaxed);
    const uint64_t g = generation_.load(std::memory_order_relaxed) + 1;
    open_.store(g);                                 // the job workers may check in to ...
    generation_.store(g);                           // ... and the word they watch (seq_cst: ordered against the sleepers' count below)
    if (sleepers_.load() > 0) { { std::lock_guard<std::mutex> lk(mu_); } cv_.notify_all(); }
    for (uint64_t k; (k = next_.fetch_add(1, std::memory_order_relaxed)) < n;) fn(k);
    // Every task has been taken.  Close the job -- a worker that wakes up from here on finds it closed and takes nothing -- and
    // wait only for the workers that DID check in: they are at most one task behind.  (Round 4 waited for every worker to
    // check in; a worker the scheduler had taken off its CPU -- other processes on the same cores, a cgroup quota -- stalled
    // the caller for a time slice after all the work was done.)
    open_.store(0);
    for (unsigned spins = 0; active_.load() != 0; ++spins) {
      if (spins < 20000) cpu_relax(); else sched_yield();
    }
    fn_ = nullptr;
  }
  static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    sched_yield();
#endif
  }

 private:
  // A host-level call dispatches several short jobs a few tens of microseconds apart (descriptors, packing per sub-batch,
  // unpacking per group), and a caller that aligns batch after batch comes back within a millisecond: waking 31 sleeping
  // threads through a condition variable costs 30-60 us per dispatch, several times what a job of C2's size takes.  So a
  // worker that runs out of work keeps looking for the next job for kSpinNs before it goes to sleep.
  long long kSpinNs = 400000;   // SEQALIGN_HOST_SPIN_US overrides (0: sleep at once)
  HostPool() {
    if (const char *env = getenv("SEQALIGN_HOST_SPIN_US")) kSpinNs = 1000ll * std::max(0, atoi(env));
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t mask;
    if (sched_getaffinity(0, sizeof(mask), &mask) == 0 && CPU_COUNT(&mask) > 0) hw = (unsigned)CPU_COUNT(&mask);
    unsigned want = hw ? std::min(hw, 32u) : 4u;
    // A container's CPU quota (cgroup cpu.max) is what the scheduler enforces, whatever the affinity mask shows: on a box
    // that shows 256 CPUs and grants 16, a pool of 32 threads that keep looking for work between jobs burns the quota twice
    // over and the whole process is throttled for it (tools/host_scale.py on the 1-GPU boxes: 8 x 32 threads 45 ms per call,
    // 8 x 8 threads 3.7 ms).  So: no more threads than the quota grants.
    if (const unsigned q = cgroup_cpu_quota()) want = std::min(want, std::max(1u, q));
    if (const char *env = getenv("SEQALIGN_HOST_THREADS")) want = (unsigned)std::max(1, atoi(env));
    for (unsigned t = 1; t < want; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    stop_.store(true);
    generation_.fetch_add(1);
    { std::lock_guard<std::mutex> lk(mu_); }
    cv_.notify_all();
    for (auto &th : workers_) th.join();
  }
  // CPUs the cgroup's quota grants (rounded down, 0 = no quota / unknown): cgroup v2 cpu.max, v1 cpu.cfs_quota_us
  static unsigned cgroup_cpu_quota() {
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (fscanf(g, "%lld", &quota) != 1) quota = -1;
      fclose(g);
      if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (unsigned)std::max<long long>(1, quota / period);
  }
  static long long now_ns() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1000000000ll + t.tv_nsec; }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const long long t0 = now_ns();
      for (unsigned spins = 1; generation_.load(std::memory_order_acquire) == seen; ++spins) {
        cpu_relax();
        if ((spins & 255u) == 0 && now_ns() - t0 > kSpinNs) {
          std::unique_lock<std::mutex> lk(mu_);
          sleepers_.fetch_add(1);
          cv_.wait(lk, [&] { return generation_.load() != seen; });
          sleepers_.fetch_sub(1);
          break;
        }
      }
      seen = generation_.load(std::memory_order_acquire);
      if (stop_.load()) return;
      // check in, THEN look whether the job this worker woke up for is still open (seq_cst on both sides: either this worker
      // sees the job closed and touches nothing of it, or the caller sees the check-in and waits for it)
      active_.fetch_add(1);
      if (open_.load() == seen) {
        const std::function<void(uint64_t)> *fn = fn_;
        const uint64_t n = n_;
        for (uint64_t k; (k = next_.fetch_add(1, std::memory_order_relaxed)) < n;) (*fn)(k);
      }
      active_.fetch_sub(1);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, job_mu_;
  std::condition_variable cv_;
  const std::function<void(uint64_t)> *fn_ = nullptr;
  uint64_t n_ = 0;
  std::atomic<uint64_t> generation_{0}, next_{0}, open_{0};   // open_: the generation whose tasks may still be taken (0: none)
  std::atomic<unsigned> active_{0}, sleepers_{0};              // active_: workers checked in to a job
  std::atomic<bool> stop_{false};
};

template <class F>
void parallel_for(uint64_t n, F fn) {
  HostPool::get().run(n, std::function<void(uint64_t)>(fn));
}


// option "timing": wall-clock of the host-level stages on stderr (development aid)
struct StageTimer {
  bool on;
  explicit StageTimer(bool enabled) : on(enabled) {}
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[seqalign timing] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// ------------------------------------------------------------------ errors ---
int fail_hip(hipError_t e, const char *what);     // records the message, maps to SEQALIGN_E_*
void set_last_error(const std::string &msg);
#define HIP_TRY(expr)                                            \
  do {                                                           \
    hipError_t _e = (expr);                                      \
    if (_e != hipSuccess) return sa_host::fail_hip(_e, #expr);   \
  } while (0)


// On every exit path of a function that has enqueued asynchronous copies into (or out of) memory
// it owns -- function-local vectors, events -- wait for the stream first, so that a DMA never
// lands in freed memory after an early error return.
struct StreamSyncOnExit {
  hipStream_t st;
  explicit StreamSyncOnExit(hipStream_t s) : st(s) {}
  ~StreamSyncOnExit();
  StreamSyncOnExit(const StreamSyncOnExit &) = delete;
  StreamSyncOnExit &operator=(const StreamSyncOnExit &) = delete;
};

// hipEvent_t's destroyed on scope exit
struct EventList {
  std::vector<hipEvent_t> ev;
  ~EventList() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
  hipError_t add(unsigned flags = hipEventDefault) {
    hipEvent_t e;
    const hipError_t rc = hipEventCreateWithFlags(&e, flags);
    if (rc == hipSuccess) ev.push_back(e);
    return rc;
  }
};

// a result that is a fraction of a millisecond away: poll its event first (a blocking wait adds the wake-up of a sleeping thread)
inline hipError_t wait_event_spinning(hipEvent_t e) {
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 1;; ++spins) {
    const hipError_t q = hipEventQuery(e);
    if (q != hipErrorNotReady) return q;
    HostPool::cpu_relax();
    if ((spins & 63u) == 0) {
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 3000000ll) return hipEventSynchronize(e);
    }
  }
}

// ... and a stream whose work is: hipStreamSynchronize puts the caller to sleep after a short spin, and being woken costs
// ~70 us on this stack -- measured in round 5 on seqalign_sw_batch's best-hit call (C4: 0.91-0.95 ms with hipStreamSynchronize,
// 0.84-0.86 polling, same box, profiles/r05/r05_experiments.txt).  The host-level calls wait for fractions of a millisecond:
// they poll, and fall back to the blocking wait after 3 ms.
inline hipError_t stream_wait_spinning(hipStream_t st, long long budget_ns = 3000000ll) {
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 1;; ++spins) {
    const hipError_t q = hipStreamQuery(st);
    if (q != hipErrorNotReady) return q;
    HostPool::cpu_relax();
    if ((spins & 63u) == 0) {
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > budget_ns) return hipStreamSynchronize(st);
    }
  }
}

// (the guard of every exit path, the error paths included: on the common path the stream is already idle and the first query says
// so; where it is not, poll briefly and then sleep -- a thread per GPU polling 3 ms each under a cgroup CPU quota is noticeable)
inline StreamSyncOnExit::~StreamSyncOnExit() { (void)stream_wait_spinning(st, 200000ll); }

struct DevBuf {   // grow-only device scratch
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return SEQALIGN_OK;
    release();
    size_t want = bytes + bytes / 8 + 256;
    // large buffers (direction bytes, staging): from the chunks the arena placement's walk left with the process, when there
    // are any -- VRAM that never went back to the driver needs no clearing (sa_placement.hip: the chunk pool)
    int dev = -1;
    if (want >= kPoolFrom && hipGetDevice(&dev) == hipSuccess && (p = sa_pool_alloc(dev, want))) {
      cap = (want + kPoolChunk - 1) / kPoolChunk * kPoolChunk;
      return SEQALIGN_OK;
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return fail_hip(e, "hipMalloc"); }
    cap = want;
    return SEQALIGN_OK;
  }
  void release() { if (p && !sa_pool_free(p)) (void)hipFree(p); p = nullptr; cap = 0; }
  static constexpr size_t kPoolFrom = (size_t)128 << 20, kPoolChunk = (size_t)512 << 20;
  template <class T> T *as() const { return static_cast<T *>(p); }
};

struct HostBuf {  // grow-only pinned staging
  void *p = nullptr;
  void *dev = nullptr;   // the same memory as the GPU addresses it: kernels read / write it in place over PCIe
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return SEQALIGN_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; dev = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; return fail_hip(e, "hipHostMalloc"); }
    e = hipHostGetDevicePointer(&dev, p, 0);
    if (e != hipSuccess) { (void)hipHostFree(p); p = nullptr; dev = nullptr; return fail_hip(e, "hipHostGetDevicePointer"); }
    cap = want;
    return SEQALIGN_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; dev = nullptr; cap = 0; }
  template <class T> T *as() const { return static_cast<T *>(p); }
  template <class T> T *dev_as() const { return static_cast<T *>(dev); }
};


}  // namespace sa_host


// Everything that steers a context's choices.  Filled ONCE, in seqalign_ctx_create, from the SEQALIGN_* environment
// (sa_options_from_env, sa_device.hip); changed afterwards only through seqalign_ctx_set_option(ctx, key, value)
// with key = the variable's name without the prefix, in lower case ("kernel", "traceback", "sweep_mode", ...).
// Nothing below seqalign_ctx_create reads the environment: two contexts in one process can differ, and a test
// steers a kernel choice on ITS context, not on the process.
struct SaOptions {
  int kernel = 0;                 // kernel            auto|wavefront|rowscan|stream|strips|wgstream: what KERNEL_AUTO means
  uint32_t cpl = 0;               // cpl               columns per lane at least (tuning experiments)
  uint32_t wpb = 0;               // wpb               pairs per workgroup of the stream kernel: 1|2|4|8
  uint32_t lds_pad = 0;           // lds_pad           extra LDS bytes per workgroup (occupancy experiments)
  bool traceback_host = false;    // traceback         device|host: where seqalign_nw_batch / sw_batch walk the matrices
  uint32_t trace_kernel = 0;      // trace_kernel      auto|lane|wave: the device walker
  int sweep_mode = 0;             // sweep_mode        auto|pair|strips (sa_batch_sw.hip)
  uint32_t sweep_strip = 0;       // sweep_strip       64|128|256 columns per strip
  uint32_t sweep_cpl = 0;         // sweep_cpl         1|2|4: the LDS form of the sweep
  bool sweep_ev = true;           // sweep_ev          0|1: the direction-byte sweep carries a walk as one word key << 2 | state (one min3 per cell)
  bool sweep_trace = false;       // sweep_trace       per-pair counters of the sweep on stderr
  bool nw_dirs = true;            // nw_dirs           0|1: seqalign_nw_batch fills ONLY a byte of directions per cell (sa_fill_dirs.hip) where
                                  //                   it applies (plain scorings, rows <= 1 024 columns), instead of the three matrices
  bool sweep_dirs = true;         // sweep_dirs        0|1: the multi-hit path fills match_scores + direction bytes (sa_fill_dirs.hip)
                                  //                   where it applies, instead of the three matrices
  int pack16 = 1;                 // pack16            0|1|2: the direction-byte fills take two pairs per wave in packed int16 (sa_fill_dirs_x2.hip)
                                  //                   where they apply (every pair of the chunk the same shape, match / mismatch scoring):
                                  //                   never | chunks of > 1 024 pairs | whatever the chunk's size (tests)
  uint32_t quad = 0;              // quad              0|1|2: the NW / SW best-hit packed fills take FOUR pairs per wave (32 lanes a couple; uniform chunks, rows
                                  //                   up to 192 columns): chunks of >= 4 096 / 16 384 pairs | never | whatever the chunk's size (tests)
  bool walk_overlap = false;      // walk_overlap      0|1: seqalign_nw_batch's direction-byte path walks a group of sub-batches on its own stream
                                  //                   while the next group fills (a VALU-bound fill next to a latency-bound walk)
  bool nw_moves = true;           // nw_moves          0|1: seqalign_nw_batch's direction-byte path sends home two bits per alignment column (which of
                                  //                   the two strings has a gap there) and the host expands them against the sequences it still
                                  //                   holds (host/sa_moves.c), instead of the two gapped strings
  uint32_t zero_copy = 4;         // zero_copy         0..3 | auto: that path's kernels read the packed sequences + descriptors from (1) and write
                                  //                   the moves to (2) pinned host memory in place, instead of staging copies either way;
                                  //                   auto (4): moves in place when the walks run one wave each (coalesced words)
  uint32_t reduce_depth = 0;      // reduce_depth      0|4|8: KiB per wave and step of sw_reduce_kernel (0 = 4; 8 measured slower)
  bool timing = false;            // timing            stage laps of the host-level calls on stderr
  size_t chunk_bytes = 0;         // chunk_bytes       device memory one host-level chunk may use (0: 40 % of free, <= 48 GB)
  uint32_t upload_slices = 0;     // upload_slices     seqalign_nw_batch (moves path): slices a sub-batch's sequences are packed and uploaded in (0 = 1)
  uint32_t subbatches = 0;        // subbatches        sub-batches a chunk of seqalign_nw_batch is pipelined in (0: by size, 1: off)
  uint32_t arena_scan_gib = 160;  // arena_scan_gib    how much HBM the arena placement may hold transiently while it looks
                                  //                   for memory that does not disturb the first two arenas (0: allocate plainly)
  float arena_quality = 1.045f;    // arena_quality     placement probe ratio that ends the walk early (else: the best candidate of the whole walk)
  uint32_t arena_free_pct = 60;   // arena_free_pct    share of the memory free at the start that an explicit placement walk may hold (10 .. 90)
  uint32_t dirs_local = 1;        // dirs_local        1|0: chunks whose walks are tile walks (NW moves path, SW best hit) get the LOCAL form of the direction byte (sa_kernels.h): cheaper to write, resolved by the walker
  uint32_t walk_tile = 0;         // walk_tile         0|32|64: the local tile walker's tile edge in bytes (0: 32 for global walks, 64 for best-hit walks)
  uint32_t walk_stage = 1;        // walk_stage        1|0: the local tile walker writes a wave's moves as one contiguous run out of LDS (whole lines over PCIe) instead of two pieces per walk
  uint32_t walk_group = 0;        // walk_group        0|1|4|8: walks per wave of the tile walker on moves (0: four in lockstep on blocked direction bytes, else one; 1 / 4 / 8: forced)
  uint32_t async_lanes = 0;       // async_lanes       1..8 (0 = 3): batches seqalign_*_batch_submit keeps in flight per context (sa_async.hip)
  uint32_t arena_keep_gib = 16;   // arena_keep_gib    how much of a walk's unused chunks stays with the process (the chunk pool: large scratch
                                  //                   buffers are mapped from it instead of freshly released, not yet cleared VRAM); 0: none
};

struct seqalign_dev_scoring {
  sa_flat_scoring_t flat;   // host copy (table pointer owned)
  uint16_t *d_code = nullptr;
  int32_t *d_table = nullptr;
  int32_t table_abs_max = 0;   // largest |entry| of the table (sentinels excluded): the packed fills' int16 bound
};

struct seqalign_ctx {
  int device = 0;
  SaOptions opt;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // side stream (created on first use): work that overlaps the main stream's kernels
  hipStream_t copy_streams[3] = {nullptr, nullptr, nullptr};   // seqalign_nw_batch's pipeline (sa_batch.hip): upload, download, walks; created on first use
  size_t chunk_budget = 0;   // bytes of device memory one host-level chunk may use
  size_t chunk_budget_default = 0;
  // device scratch for the host-level entry points
  sa_host::DevBuf arena, off_a, pair_list, status;   // (off_a holds all five descriptor arrays; pair_list: nw_chunk_pipelined's mixed chunks)
  sa_host::DevBuf M, A, B;           // views of arena_set (sa_host::reserve_arenas); never reserved / released on their own
  SaArenaSet *arena_set = nullptr;   // the three matrix arenas, placed (sa_placement.hip)
  bool arena_placed = false;         // the set in arena_set went through the placement walk (reserve_arenas(.., placed = true))
  uint32_t arena_walks = 0;          // how often reserve_arenas has placed them (the first walk is the full one)
  sa_host::HostBuf h_one;            // the legacy single-pair call: descriptor + sequences + three matrices + status of ONE pair,
  void *one_dev = nullptr;           // pinned, read and written in place by the GPU (sa_fill_one_pair); its device address
  sa_host::DevBuf dirs;              // seqalign_nw_batch: one byte of directions per cell (sa_fill_dirs.hip)
  sa_host::DevBuf best_score, best_index, cand_count, cand_off, cand_cap, cand_index, cand_score;
  sa_host::DevBuf t_str_off, t_out_a, t_out_b, t_meta;   // device traceback outputs
  sa_host::DevBuf e[14];                                 // device SW enumeration scratch (see sw_chunk_device_enumerate)
  sa_host::DevBuf strip_progress;                        // sa_fill_strips.hip: rows done per (pair, strip)
  sa_host::HostBuf h_desc, h_arena, h_M, h_A, h_B, h_misc, h_ta, h_tb, h_tmeta;
  // the last scorings uploaded through cached_scoring (host-level entry points, legacy single-pair path): [is_sw]
  seqalign_call_info_t call_info = {};   // what the last call launched (seqalign_ctx_last_call_info)
  int call_depth = 0;                    // entry points nest (seqalign_nw_batch -> seqalign_fill_batch_device): the outermost resets
  seqalign_dev_scoring *cached[2] = {nullptr, nullptr};
  uint64_t cached_fp[2] = {0, 0};
  // what the host-level call in progress delivers per alignment (seqalign_*_batch_cigar set it for their duration; a context
  // serves one host-level call at a time): 0 = the two gapped strings, 1 = CIGAR with M, 2 = CIGAR with = / X
  void *async = nullptr;        // the lanes of seqalign_*_batch_submit (sa_async.hip), created by the first submit
  int cigar_format = 0;
  bool cigar_fold = false;      // format 2: letters compared case-folded (the scoring is case-insensitive)
};


namespace sa_host {

// Every C-ABI entry point that launches opens one: the outermost scope of a call clears the context's record and points
// the calling thread's recorder (sa_record_launch, sa_kernels.h) at it.
struct CallScope {
  seqalign_ctx *ctx;
  seqalign_call_info_t *prev;
  explicit CallScope(seqalign_ctx *c);
  ~CallScope();
  CallScope(const CallScope &) = delete;
  CallScope &operator=(const CallScope &) = delete;
};

void async_shutdown(seqalign_ctx *ctx);   // sa_async.hip: drain the submitted jobs, join the lanes (seqalign_ctx_destroy)

struct CigarScope {   // RAII: the call's output format, put back on every way out
  seqalign_ctx *ctx;
  CigarScope(seqalign_ctx *c, int format, bool fold) : ctx(c) { c->cigar_format = format; c->cigar_fold = fold; }
  ~CigarScope() { ctx->cigar_format = 0; ctx->cigar_fold = false; }
  CigarScope(const CigarScope &) = delete;
  CigarScope &operator=(const CigarScope &) = delete;
};

// One alignment given as its two gapped strings (the paths whose walkers produce strings: three matrices, the host) into the
// caller's buffer at out_a + at (and out_b + at): the strings, or in CIGAR mode the CIGAR of them (out_a only).  Returns the bytes
// used with the NUL, or 0 when `room` bytes do not hold it.
uint64_t put_alignment(const seqalign_ctx *ctx, const char *sa, const char *sb, uint32_t len, char *out_a, char *out_b, uint64_t at, uint64_t room);

// grow the context's three matrix arenas together (spread placement, sa_placement.hip)
int reserve_arenas(seqalign_ctx *ctx, size_t bytes, bool placed = true);

// The uploaded form of `scoring` for the host-level entry points: the context keeps the last one it flattened and
// uploaded (per NW / SW), keyed by a fingerprint of everything scoring_lookup can see, so that a caller who aligns
// batch after batch with one scoring_t pays for the flatten + two hipMallocs + two synchronous copies once, not per
// call (C2: ~0.1 ms of a 1.3 ms call).  Owned by the context; valid until the next call with a different scoring.
int cached_scoring(seqalign_ctx *ctx, const scoring_t *scoring, int is_sw, seqalign_dev_scoring **out);

// the fill on device-resident data; best_score / best_index (optional, SW): filled by the fill itself
// when the stream kernel runs (*best_done = true), otherwise the caller runs the separate reduction
// cand (optional, SW): ask the fill for what the multi-hit path needs to know about the candidates (count, box,
// columns per row); *cand_done tells whether the fill kernel reported them (the stream kernel does), otherwise
// the caller runs sa_launch_sw_box over match_scores
int fill_device(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, const seqalign_dev_batch_t *batch,
                int kernel, void *stream, int32_t *best_score, uint64_t *best_index, bool *best_done,
                const SaCandBox *cand = nullptr, bool *cand_done = nullptr);

// ---- host-level chunking (sa_batch.hip)
struct Chunk {
  uint64_t first = 0, count = 0;   // pairs [first, first+count)
  uint64_t cells = 0, seq_bytes = 0;
  uint32_t max_a = 0, max_b = 0;
};

// extra_bytes (optional, [n_pairs]): device bytes pair p needs besides its cells (the SW multi-hit path's scratch arena: a pair
// with a low min_score needs more of it than the batch's average)
std::vector<Chunk> plan_chunks(const seqalign_batch_t *b, size_t budget, size_t bytes_per_cell = 12, const uint64_t *extra_bytes = nullptr);
int run_chunk(seqalign_ctx *ctx, const seqalign_batch_t *b, const Chunk &c, const seqalign_dev_scoring *sc,
              seqalign_dev_batch_t *dev_out, bool *best_done = nullptr, const SaCandBox *cand = nullptr,
              bool *cand_done = nullptr, uint64_t uniform_stride = 0);
int ensure_copy_streams(seqalign_ctx *ctx, int count);
// run_chunk's uniform_stride: "a ragged chunk for the packed fills -- pair the pairs up by shape" (sa_batch.hip)
constexpr uint64_t kBucketShapes = ~(uint64_t)0;
constexpr uint64_t kShapeTableMax = (uint64_t)1 << 20;   // (len_a + 1) x (len_b + 1) entries of the pairing table: 4 MiB per host thread at most
int fetch_status(seqalign_ctx *ctx, const Chunk &c, uint64_t *status_out);
int nw_dirs_fill(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, const seqalign_dev_batch_t *batch, uint8_t *dirs,
                 int32_t *end_score, uint64_t *end_state, void *stream, bool *used, uint64_t uniform_stride = 0,
                 const uint32_t *pair_list = nullptr, uint32_t list_count = 0, bool local = false);
int nw_dirs_fill_mixed(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, const seqalign_dev_batch_t *batch, uint8_t *dirs,
                       int32_t *end_score, uint64_t *end_state, void *stream, const uint32_t *list, uint32_t n_modal, uint32_t n_rest,
                       uint32_t modal_a, uint32_t modal_b, bool local = false);
// (n_pairs: of the chunk -- FEW pairs with rows over 768 columns stay with three matrices, whose fills put several waves on a pair)
bool nw_dirs_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t max_len_a, uint64_t n_pairs = ~0ull);
// whether a chunk whose pairs all are len_a x len_b may take the packed two-pairs-per-wave fill (its layout: every pair's
// cells start on a multiple of 256)
bool nw_dirs_x2_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t len_a, uint32_t len_b);
// Two pairs per wave halve the waves of a launch.  Up to 1 024 pairs every wave of the one-pair kernels has a SIMD to itself and the
// two forms take the same time; from the 1 025th pair on two of their waves share a SIMD and the packed fills are ahead
// (tools/pack_by_batch_size.py, profiles/r05/r05_pack_by_batch_size.txt: 1 056 pairs -- NW C2's shape 0.228 -> 0.214 ms, SW up to 4
// hits C3's 1.28 -> 1.12, C4's 0.92 -> 0.88; 1 024 pairs: 0.200 / 0.199, 1.10 / 1.09, 0.75 / 0.78).  (Rounds 3-4: 2 048, from a
// record taken before the packed kernels' later gains.)
constexpr uint64_t kPackedFillMinPairs = 1025;
// ... and for RAGGED chunks (pairs bucketed by shape: the pairs that find a partner two per wave, the others one per wave, in two
// launches or one mixed grid) the rounds-3/4 threshold stands: 1 100-2 047 pairs of 100..150 x 100..150 are 2-12 % SLOWER bucketed
// (tools/ragged_bench.py 1600: 0.227 ms against 0.207 one pair per wave; 2 500: 0.254 / 0.250; 125 000: 3.6 / 4.6)
constexpr uint64_t kBucketedFillMinPairs = 2048;
bool sw_best_x2_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t len_a, uint32_t len_b);
int sw_traceback_dirs(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *sc, const seqalign_dev_batch_t *b, const seqalign_trace_t *t,
                      const uint8_t *dirs, const int32_t *start_score, void *stream);
bool sw_dirs_x2_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t len_a, uint32_t len_b);
int check_batch(const seqalign_batch_t *b);
// chunked fill of a host batch with an uploaded scoring, matrices copied back (also the legacy single-pair path)
int fill_batch_uploaded(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const seqalign_dev_scoring *sc,
                        const uint64_t *mat_off, int32_t *M, int32_t *A, int32_t *B, uint64_t *status);
bool traceback_on_host(const seqalign_ctx *ctx);
void parallel_memcpy(void *dst, const void *src, size_t bytes);

}  // namespace sa_host
