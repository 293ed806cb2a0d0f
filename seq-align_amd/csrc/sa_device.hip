// sa_device.hip -- the C-ABI shim (include/seqalign_hip.h) over the HIP kernels.
//
// Host code above this file is C; this file is the only place that talks to the
// HIP runtime.  No torch / C++ types cross the boundary.  No CPU fallback: when
// the runtime or a device is missing every entry point reports it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
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

namespace {

// Persistent host worker pool: run fn(0..n-1) over the workers + the caller.
// Pairs / memcpy pieces are independent.  SEQALIGN_HOST_THREADS overrides the
// worker count (default min(hardware threads, 32)).  One job at a time.
class HostPool {
 public:
  static HostPool &get() { static HostPool pool; return pool; }
  void run(uint64_t n, const std::function<void(uint64_t)> &fn) {
    if (n == 0) return;
    if (workers_.empty() || n == 1) { for (uint64_t k = 0; k < n; ++k) fn(k); return; }
    std::lock_guard<std::mutex> one_job(job_mu_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; n_ = n; next_.store(0); pending_ = (unsigned)workers_.size(); ++generation_;
    }
    cv_.notify_all();
    for (uint64_t k; (k = next_.fetch_add(1)) < n;) fn(k);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  HostPool() {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned want = hw ? std::min(hw, 32u) : 4u;
    if (const char *env = getenv("SEQALIGN_HOST_THREADS")) want = (unsigned)std::max(1, atoi(env));
    for (unsigned t = 1; t < want; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++generation_; }
    cv_.notify_all();
    for (auto &th : workers_) th.join();
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return generation_ != seen; });
      seen = generation_;
      if (stop_) return;
      const std::function<void(uint64_t)> *fn = fn_;
      const uint64_t n = n_;
      lk.unlock();
      for (uint64_t k; (k = next_.fetch_add(1)) < n;) (*fn)(k);
      lk.lock();
      if (--pending_ == 0) done_cv_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, job_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(uint64_t)> *fn_ = nullptr;
  uint64_t n_ = 0, generation_ = 0;
  std::atomic<uint64_t> next_{0};
  unsigned pending_ = 0;
  bool stop_ = false;
};

template <class F>
static void parallel_for(uint64_t n, F fn) {
  HostPool::get().run(n, std::function<void(uint64_t)>(fn));
}

}  // namespace

// ------------------------------------------------------------------ errors ---
static thread_local std::string g_last_error;

static int fail_hip(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? SEQALIGN_E_NO_DEVICE
                                                               : SEQALIGN_E_HIP;
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return fail_hip(_e, #expr);   \
  } while (0)

extern "C" const char *seqalign_last_error(void) { return g_last_error.c_str(); }

extern "C" const char *seqalign_strerror(int code) {
  switch (code) {
    case SEQALIGN_OK: return "ok";
    case SEQALIGN_E_NO_DEVICE: return "no HIP device (gfx950) available";
    case SEQALIGN_E_HIP: return "HIP runtime error";
    case SEQALIGN_E_ARG: return "invalid argument";
    case SEQALIGN_E_NOMEM: return "out of memory";
    case SEQALIGN_E_UNKNOWN_PAIR: return "unknown character pair and match/mismatch not set";
    case SEQALIGN_E_DOMAIN: return "scoring outside the defined domain (penalty below -|min_penalty|)";
    case SEQALIGN_E_TRACEBACK: return "traceback failed";
    case SEQALIGN_E_TOO_LARGE: return "pair too large (>= 2^31 cells)";
  }
  return "unknown error";
}

// ----------------------------------------------------------------- context ---
namespace {

struct DevBuf {   // grow-only device scratch
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return SEQALIGN_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return fail_hip(e, "hipMalloc"); }
    cap = want;
    return SEQALIGN_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

struct HostBuf {  // grow-only pinned staging
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return SEQALIGN_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; return fail_hip(e, "hipHostMalloc"); }
    cap = want;
    return SEQALIGN_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace

struct seqalign_dev_scoring {
  sa_flat_scoring_t flat;   // host copy (table pointer owned)
  uint16_t *d_code = nullptr;
  int32_t *d_table = nullptr;
};

struct seqalign_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  size_t chunk_budget = 0;   // bytes of device memory one host-level chunk may use
  // device scratch for the host-level entry points
  DevBuf arena, off_a, len_a, off_b, len_b, mat_off, M, A, B, status;
  DevBuf best_score, best_index, cand_count, cand_off, cand_cap, cand_index, cand_score;
  DevBuf t_str_off, t_out_a, t_out_b, t_meta;   // device traceback outputs
  DevBuf e[12];                                 // device SW enumeration scratch (see sw_chunk_device_enumerate)
  DevBuf strip_progress;                        // sa_fill_strips.hip: rows done per (pair, strip)
  HostBuf h_desc, h_arena, h_M, h_A, h_B, h_misc, h_ta, h_tb, h_tmeta;
  // cached flattened scoring for the legacy single-pair path
  seqalign_dev_scoring *cached = nullptr;
  uint64_t cached_fp = 0;
  int cached_is_sw = -1;
};

// grow the context's three matrix arenas together (spread placement, sa_placement.hip)
static int reserve_arenas(seqalign_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->M.cap && bytes <= ctx->A.cap && bytes <= ctx->B.cap) return SEQALIGN_OK;
  ctx->M.release(); ctx->A.release(); ctx->B.release();
  const size_t want = bytes + bytes / 8 + 4096;
  void *a[3];
  hipError_t e = sa_alloc_arenas_spread(want, a, ctx->stream, nullptr);
  if (e != hipSuccess) return fail_hip(e, "hipMalloc (matrix arenas)");
  ctx->M.p = a[0]; ctx->A.p = a[1]; ctx->B.p = a[2];
  ctx->M.cap = ctx->A.cap = ctx->B.cap = want;
  return SEQALIGN_OK;
}

extern "C" int seqalign_arenas_alloc(seqalign_ctx_t *ctx, uint64_t bytes_each, void *arenas[3], float *quality) {
  if (!ctx || !arenas || !bytes_each) return SEQALIGN_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  float q = -1.f;
  hipError_t e = sa_alloc_arenas_spread((size_t)bytes_each, arenas, ctx->stream, &q);
  if (e != hipSuccess) return fail_hip(e, "hipMalloc (matrix arenas)");
  if (quality) *quality = q;
  return SEQALIGN_OK;
}

extern "C" int seqalign_arenas_free(seqalign_ctx_t *ctx, void *arenas[3]) {
  if (!ctx || !arenas) return SEQALIGN_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  for (int k = 0; k < 3; ++k) {
    if (arenas[k]) (void)hipFree(arenas[k]);
    arenas[k] = nullptr;
  }
  return SEQALIGN_OK;
}

extern "C" int seqalign_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

extern "C" int seqalign_ctx_create(int device, seqalign_ctx_t **out) {
  if (!out) return SEQALIGN_E_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    g_last_error = "hipGetDeviceCount: no device";
    return SEQALIGN_E_NO_DEVICE;
  }
  if (device < 0 || device >= n) return SEQALIGN_E_ARG;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
    return SEQALIGN_E_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  seqalign_ctx *ctx = new (std::nothrow) seqalign_ctx();
  if (!ctx) return SEQALIGN_E_NOMEM;
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete ctx; return fail_hip(e, "hipStreamCreate"); }
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  // one chunk of a host-level batch may use up to 40 % of what is free now
  // (288 GB HBM3E: ~100 GB per chunk on an empty MI355X), overridable
  ctx->chunk_budget = free_b ? (free_b / 10) * 4 : (size_t)8 << 30;
  // ...but not more than 48 GB: beyond that a chunk only adds allocation time (page
  // tables for tens of GB) and delays the first results; SEQALIGN_CHUNK_BYTES overrides
  ctx->chunk_budget = std::min<size_t>(ctx->chunk_budget, (size_t)48 << 30);
  if (const char *env = getenv("SEQALIGN_CHUNK_BYTES")) {
    size_t v = strtoull(env, nullptr, 10);
    if (v >= (1u << 20)) ctx->chunk_budget = v;
  }
  *out = ctx;
  return SEQALIGN_OK;
}

extern "C" void seqalign_ctx_destroy(seqalign_ctx_t *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->cached) seqalign_scoring_release(ctx, ctx->cached);
  for (DevBuf *b : {&ctx->arena, &ctx->off_a, &ctx->len_a, &ctx->off_b, &ctx->len_b, &ctx->mat_off,
                    &ctx->M, &ctx->A, &ctx->B, &ctx->status, &ctx->best_score, &ctx->best_index,
                    &ctx->cand_count, &ctx->cand_off, &ctx->cand_cap, &ctx->cand_index, &ctx->cand_score,
                    &ctx->t_str_off, &ctx->t_out_a, &ctx->t_out_b, &ctx->t_meta})
    b->release();
  for (DevBuf &b : ctx->e) b.release();
  ctx->strip_progress.release();
  for (HostBuf *b : {&ctx->h_desc, &ctx->h_arena, &ctx->h_M, &ctx->h_A, &ctx->h_B, &ctx->h_misc, &ctx->h_ta,
                     &ctx->h_tb, &ctx->h_tmeta})
    b->release();
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int seqalign_ctx_device(const seqalign_ctx_t *ctx) { return ctx ? ctx->device : -1; }

// ----------------------------------------------------------------- scoring ---
extern "C" int seqalign_scoring_upload(seqalign_ctx_t *ctx, const scoring_t *scoring, int is_sw,
                                       seqalign_dev_scoring_t **out) {
  if (!ctx || !scoring || !out) return SEQALIGN_E_ARG;
  *out = nullptr;
  seqalign_dev_scoring *h = new (std::nothrow) seqalign_dev_scoring();
  if (!h) return SEQALIGN_E_NOMEM;
  int rc = sa_flatten_scoring(scoring, is_sw, &h->flat);
  if (rc != SEQALIGN_OK) { delete h; return rc; }
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t tb = sizeof(int32_t) * h->flat.n_classes * h->flat.n_classes;
  hipError_t e = hipMalloc((void **)&h->d_code, sizeof(h->flat.code));
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_table, tb);
  if (e == hipSuccess) e = hipMemcpy(h->d_code, h->flat.code, sizeof(h->flat.code), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_table, h->flat.table, tb, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    seqalign_scoring_release(ctx, h);
    return fail_hip(e, "scoring upload");
  }
  *out = h;
  return SEQALIGN_OK;
}

extern "C" void seqalign_scoring_release(seqalign_ctx_t *ctx, seqalign_dev_scoring_t *h) {
  if (!h) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  if (h->d_code) (void)hipFree(h->d_code);
  if (h->d_table) (void)hipFree(h->d_table);
  sa_flat_scoring_free(&h->flat);
  if (ctx && ctx->cached == h) { ctx->cached = nullptr; ctx->cached_is_sw = -1; }
  delete h;
}

namespace {
// releases an uploaded scoring on every exit path of the host-level entry points
struct ScoringGuard {
  seqalign_ctx *ctx;
  seqalign_dev_scoring *h = nullptr;
  explicit ScoringGuard(seqalign_ctx *c) : ctx(c) {}
  ~ScoringGuard() { if (h) seqalign_scoring_release(ctx, h); }
  ScoringGuard(const ScoringGuard &) = delete;
  ScoringGuard &operator=(const ScoringGuard &) = delete;
};
}  // namespace

// --------------------------------------------------------------- hot path ---
static SaFillParams make_params(const seqalign_dev_scoring_t *s, const seqalign_dev_batch_t *b) {
  SaFillParams p;
  p.arena = b->arena; p.off_a = b->off_a; p.len_a = b->len_a; p.off_b = b->off_b; p.len_b = b->len_b;
  p.mat_off = b->mat_off; p.M = b->match_scores; p.A = b->gap_a_scores; p.B = b->gap_b_scores;
  p.status = b->status; p.code = s->d_code; p.table = s->d_table;
  p.n_pairs = (uint32_t)b->n_pairs; p.K = s->flat.n_classes;
  p.gap_open = s->flat.gap_open; p.open1 = s->flat.open1; p.ext = s->flat.ext; p.floor = s->flat.floor;
  p.gen_eq = s->flat.gen_eq; p.gen_ne = s->flat.gen_ne; p.flags = s->flat.flags;
  p.best_score = nullptr; p.best_index = nullptr;
  return p;
}

static int pick_kernel(int kernel) {
  if (kernel != SEQALIGN_KERNEL_AUTO) return kernel;
  if (const char *env = getenv("SEQALIGN_KERNEL")) {
    if (!strcmp(env, "wavefront")) return SEQALIGN_KERNEL_WAVEFRONT;
    if (!strcmp(env, "rowscan")) return SEQALIGN_KERNEL_ROWSCAN;
    if (!strcmp(env, "stream")) return SEQALIGN_KERNEL_STREAM;
    if (!strcmp(env, "strips")) return SEQALIGN_KERNEL_STRIPS;
  }
  return SEQALIGN_KERNEL_STREAM;   // measured fastest (profiles/); falls back when not applicable
}

// best_score / best_index (optional, SW): filled by the fill itself when the stream kernel runs
// (*best_done = true); otherwise the caller runs the separate reduction
static int fill_device(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, const seqalign_dev_batch_t *batch,
                       int kernel, void *stream, int32_t *best_score, uint64_t *best_index, bool *best_done) {
  if (best_done) *best_done = false;
  if (!ctx || !scoring || !batch) return SEQALIGN_E_ARG;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  if (batch->n_pairs > 0xFFFFFFFFull) return SEQALIGN_E_ARG;
  if ((uint64_t)(batch->max_len_a + 1ull) * (batch->max_len_b + 1ull) >= (1ull << 31)) return SEQALIGN_E_TOO_LARGE;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SaFillParams p = make_params(scoring, batch);
  hipError_t e;
  int which = pick_kernel(kernel);
  // a positive gap_extend (legal, absurd) breaks the row scan's saturating-add
  // identity; the wavefront kernel is exact for any sign
  if (p.ext > 0) which = SEQALIGN_KERNEL_WAVEFRONT;
  const bool stream_ok = sa_stream_kernel_applicable(p, batch->max_len_a);
  if (kernel == SEQALIGN_KERNEL_AUTO && which == SEQALIGN_KERNEL_STREAM) {
    // One wave per pair needs pairs to fill the chip.  Measured (seq-align_amd/tools/long_pairs.py,
    // profiles/r01_long_pairs.txt): 64 x 1000x1000 -- strips 0.49 ms, rowscan 0.86, stream 2.09;
    // 256 x 1000x1000 -- rowscan 0.99, strips 1.23, stream 2.09; 1000 x 1000x1000 -- stream 3.2,
    // rowscan 4.7; 16 x 5000x5000 -- strips 2.9, rowscan 21; 512 x 2000x2000 -- rowscan 5.2, strips 10.5.
    if (batch->max_len_a > 512 && batch->n_pairs < 256) which = SEQALIGN_KERNEL_STRIPS;
    else if (!stream_ok || (batch->max_len_a > 767 && batch->n_pairs < 768)) which = SEQALIGN_KERNEL_ROWSCAN;
  }
  if (which == SEQALIGN_KERNEL_STREAM && !stream_ok) which = SEQALIGN_KERNEL_ROWSCAN;
  if (which == SEQALIGN_KERNEL_STREAM && best_score && best_index) {
    p.best_score = best_score; p.best_index = best_index;
    if (sa_stream_kernel_reports_best(p, batch->max_len_a, batch->max_len_b)) { if (best_done) *best_done = true; }
    else p.best_score = nullptr, p.best_index = nullptr;
  }
  switch (which) {
    case SEQALIGN_KERNEL_WAVEFRONT: e = sa_launch_fill_wavefront(p, batch->max_len_a, st); break;
    case SEQALIGN_KERNEL_STREAM: e = sa_launch_fill_stream(p, batch->max_len_a, st); break;
    case SEQALIGN_KERNEL_ROWSCAN:
      e = sa_launch_fill_rowscan(p, batch->max_len_a, st);
      break;
    case SEQALIGN_KERNEL_STRIPS: {
      const uint64_t words = batch->n_pairs * (uint64_t)sa_fill_strips_per_pair(batch->max_len_a);
      int rc = ctx->strip_progress.reserve(words * 4 + 16);
      if (rc) return rc;
      e = sa_launch_fill_strips(p, batch->max_len_a, ctx->strip_progress.as<uint32_t>(), st);
      break;
    }
    default: return SEQALIGN_E_ARG;
  }
  if (e != hipSuccess) return fail_hip(e, "fill kernel launch");
  return SEQALIGN_OK;
}

extern "C" int seqalign_fill_batch_device(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring,
                                          const seqalign_dev_batch_t *batch, int kernel, void *stream) {
  return fill_device(ctx, scoring, batch, kernel, stream, nullptr, nullptr, nullptr);
}

extern "C" int seqalign_time_fill_ms(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring,
                                     const seqalign_dev_batch_t *batch, int kernel, void *stream,
                                     int repeats, float *ms_each) {
  if (!ctx || repeats <= 0 || !ms_each) return SEQALIGN_E_ARG;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  std::vector<hipEvent_t> ev(2 * (size_t)repeats);
  for (auto &x : ev) HIP_TRY(hipEventCreate(&x));
  int rc = SEQALIGN_OK;
  for (int r = 0; r < repeats && rc == SEQALIGN_OK; ++r) {
    HIP_TRY(hipEventRecord(ev[2 * r], st));
    rc = seqalign_fill_batch_device(ctx, scoring, batch, kernel, st);
    HIP_TRY(hipEventRecord(ev[2 * r + 1], st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  for (int r = 0; r < repeats && rc == SEQALIGN_OK; ++r) HIP_TRY(hipEventElapsedTime(&ms_each[r], ev[2 * r], ev[2 * r + 1]));
  for (auto &x : ev) (void)hipEventDestroy(x);
  return rc;
}

static int launch_traceback(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *sc, const seqalign_dev_batch_t *b,
                            const seqalign_trace_t *t, void *stream, bool sw) {
  if (!ctx || !sc || !b || !t) return SEQALIGN_E_ARG;
  if (sw && (!t->start_index || !t->out_pos)) return SEQALIGN_E_ARG;
  if (b->n_pairs == 0) return SEQALIGN_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SaTraceParams p;
  p.arena = b->arena; p.off_a = b->off_a; p.len_a = b->len_a; p.off_b = b->off_b; p.len_b = b->len_b;
  p.mat_off = b->mat_off; p.M = b->match_scores; p.A = b->gap_a_scores; p.B = b->gap_b_scores;
  p.code = sc->d_code; p.table = sc->d_table; p.str_off = t->str_off; p.out_a = t->out_a; p.out_b = t->out_b;
  p.out_head = t->out_head; p.out_len = t->out_len; p.out_score = t->out_score; p.trace_status = t->status;
  p.start_index = sw ? t->start_index : nullptr; p.out_pos = sw ? t->out_pos : nullptr;
  p.n_pairs = (uint32_t)b->n_pairs; p.K = sc->flat.n_classes; p.open1 = sc->flat.open1; p.ext = sc->flat.ext;
  p.gen_eq = sc->flat.gen_eq; p.gen_ne = sc->flat.gen_ne; p.flags = sc->flat.flags;
  hipError_t e = sa_launch_nw_traceback(p, st);
  if (e != hipSuccess) return fail_hip(e, "traceback launch");
  return SEQALIGN_OK;
}

extern "C" int seqalign_nw_traceback_device(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *sc,
                                            const seqalign_dev_batch_t *b, const seqalign_trace_t *t,
                                            void *stream) {
  return launch_traceback(ctx, sc, b, t, stream, false);
}

extern "C" int seqalign_sw_traceback_device(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *sc,
                                            const seqalign_dev_batch_t *b, const seqalign_trace_t *t,
                                            void *stream) {
  return launch_traceback(ctx, sc, b, t, stream, true);
}

extern "C" int seqalign_sw_reduce_device(seqalign_ctx_t *ctx, const seqalign_sw_reduce_t *r, void *stream) {
  if (!ctx || !r) return SEQALIGN_E_ARG;
  if (r->n_pairs == 0) return SEQALIGN_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SaReduceParams p;
  p.len_a = r->len_a; p.len_b = r->len_b; p.mat_off = r->mat_off; p.M = r->match_scores;
  p.min_score = r->min_score; p.best_score = r->best_score; p.best_index = r->best_index;
  p.cand_count = r->cand_count; p.cand_off = r->cand_off; p.cand_cap = r->cand_cap;
  p.cand_index = r->cand_index; p.cand_score = r->cand_score; p.cand_key = nullptr;
  p.n_pairs = (uint32_t)r->n_pairs;
  hipError_t e = sa_launch_sw_reduce(p, st);
  if (e != hipSuccess) return fail_hip(e, "sw reduce launch");
  return SEQALIGN_OK;
}

// ------------------------------------------------- host-level: chunked fill ---
namespace {

struct Chunk {
  uint64_t first = 0, count = 0;   // pairs [first, first+count)
  uint64_t cells = 0, seq_bytes = 0;
  uint32_t max_a = 0, max_b = 0;
};

// split the batch into chunks whose matrices (12 B/cell) fit the budget
static std::vector<Chunk> plan_chunks(const seqalign_batch_t *b, size_t budget) {
  std::vector<Chunk> out;
  Chunk c;
  const uint64_t max_cells = std::max<uint64_t>(budget / 12, 1);
  for (uint64_t p = 0; p < b->n_pairs; ++p) {
    const uint64_t cells = (uint64_t)(b->len_a[p] + 1ull) * (b->len_b[p] + 1ull);
    if (c.count && c.cells + cells > max_cells) { out.push_back(c); c = Chunk(); c.first = p; }
    c.count++; c.cells += cells; c.seq_bytes += (uint64_t)b->len_a[p] + b->len_b[p];
    c.max_a = std::max(c.max_a, b->len_a[p]); c.max_b = std::max(c.max_b, b->len_b[p]);
  }
  if (c.count) out.push_back(c);
  return out;
}

// Upload one chunk (sequences packed back to back, matrices packed in pair
// order) and run the fill.  On return the device buffers of ctx hold the
// results; the stream is NOT synchronised.
// best_done (optional): ask the fill for the SW best cell per pair (into ctx->best_score / best_index);
// *best_done tells whether the fill kernel delivered it.
static int run_chunk(seqalign_ctx *ctx, const seqalign_batch_t *b, const Chunk &c,
                     const seqalign_dev_scoring *sc, seqalign_dev_batch_t *dev_out, bool *best_done = nullptr) {
  const uint64_t n = c.count;
  int rc;
  // pinned descriptor block: off_a, off_b, mat_off (u64) then len_a, len_b (u32)
  const size_t desc_bytes = n * (3 * sizeof(uint64_t) + 2 * sizeof(uint32_t));
  if ((rc = ctx->h_desc.reserve(desc_bytes))) return rc;
  if ((rc = ctx->h_arena.reserve(c.seq_bytes + 16))) return rc;
  uint64_t *h_off_a = ctx->h_desc.as<uint64_t>(), *h_off_b = h_off_a + n, *h_mat = h_off_b + n;
  uint32_t *h_len_a = reinterpret_cast<uint32_t *>(h_mat + n), *h_len_b = h_len_a + n;
  uint8_t *h_seq = ctx->h_arena.as<uint8_t>();
  uint64_t pos = 0, cell = 0;
  for (uint64_t k = 0; k < n; ++k) {   // offsets: a sequential prefix
    const uint64_t p = c.first + k;
    h_off_a[k] = pos; pos += b->len_a[p];
    h_off_b[k] = pos; pos += b->len_b[p];
    h_len_a[k] = b->len_a[p]; h_len_b[k] = b->len_b[p];
    h_mat[k] = cell; cell += (uint64_t)(b->len_a[p] + 1ull) * (b->len_b[p] + 1ull);
  }
  constexpr uint64_t kPack = 2048;     // bytes: in parallel, 2048 pairs per task
  parallel_for((n + kPack - 1) / kPack, [&](uint64_t blk) {
    for (uint64_t k = blk * kPack, e = std::min(n, (blk + 1) * kPack); k < e; ++k) {
      const uint64_t p = c.first + k;
      memcpy(h_seq + h_off_a[k], b->arena + b->off_a[p], b->len_a[p]);
      memcpy(h_seq + h_off_b[k], b->arena + b->off_b[p], b->len_b[p]);
    }
  });
  if ((rc = ctx->arena.reserve(c.seq_bytes + 16))) return rc;
  // the five descriptor arrays travel as the one block they are on the host (off_a: the device copy)
  if ((rc = ctx->off_a.reserve(desc_bytes)) || (rc = ctx->status.reserve(n * 8))) return rc;
  if ((rc = reserve_arenas(ctx, c.cells * 4))) return rc;
  hipStream_t st = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->arena.p, h_seq, c.seq_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ctx->off_a.p, h_off_a, desc_bytes, hipMemcpyHostToDevice, st));
  uint64_t *dv_off_a = ctx->off_a.as<uint64_t>(), *dv_off_b = dv_off_a + n, *dv_mat = dv_off_b + n;
  uint32_t *dv_len_a = reinterpret_cast<uint32_t *>(dv_mat + n), *dv_len_b = dv_len_a + n;
  seqalign_dev_batch_t d;
  d.n_pairs = n; d.arena = ctx->arena.as<uint8_t>();
  d.off_a = dv_off_a; d.len_a = dv_len_a;
  d.off_b = dv_off_b; d.len_b = dv_len_b;
  d.mat_off = dv_mat;
  d.match_scores = ctx->M.as<int32_t>(); d.gap_a_scores = ctx->A.as<int32_t>(); d.gap_b_scores = ctx->B.as<int32_t>();
  d.status = ctx->status.as<uint64_t>(); d.max_len_a = c.max_a; d.max_len_b = c.max_b;
  if (best_done) {
    if ((rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8))) return rc;
    rc = fill_device(ctx, sc, &d, SEQALIGN_KERNEL_AUTO, st, ctx->best_score.as<int32_t>(),
                     ctx->best_index.as<uint64_t>(), best_done);
  } else {
    rc = seqalign_fill_batch_device(ctx, sc, &d, SEQALIGN_KERNEL_AUTO, st);
  }
  if (rc) return rc;
  if (dev_out) *dev_out = d;
  return SEQALIGN_OK;
}

// fetch the per-pair status words; returns UNKNOWN_PAIR if any pair flagged
static int fetch_status(seqalign_ctx *ctx, const Chunk &c, uint64_t *status_out) {
  int rc;
  if ((rc = ctx->h_misc.reserve(c.count * 8))) return rc;
  uint64_t *h = ctx->h_misc.as<uint64_t>();
  HIP_TRY(hipMemcpyAsync(h, ctx->status.p, c.count * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  rc = SEQALIGN_OK;
  for (uint64_t k = 0; k < c.count; ++k) {
    if (status_out) status_out[c.first + k] = h[k];
    if (h[k] != ~0ull) rc = SEQALIGN_E_UNKNOWN_PAIR;
  }
  return rc;
}

static int check_batch(const seqalign_batch_t *b) {
  if (!b || (b->n_pairs && (!b->arena || !b->off_a || !b->off_b || !b->len_a || !b->len_b))) return SEQALIGN_E_ARG;
  for (uint64_t p = 0; p < b->n_pairs; ++p)
    if ((uint64_t)(b->len_a[p] + 1ull) * (b->len_b[p] + 1ull) >= (1ull << 31)) return SEQALIGN_E_TOO_LARGE;
  return SEQALIGN_OK;
}

}  // namespace

static int fill_batch_uploaded(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const seqalign_dev_scoring *sc,
                               const uint64_t *mat_off, int32_t *M, int32_t *A, int32_t *B, uint64_t *status);

// Device -> pageable host memory.  A plain hipMemcpy to pageable memory is staged
// by the runtime at ~12 GB/s; large copies go through our own two pinned buffers
// instead: the DMA of slice i+1 overlaps a multi-threaded memcpy of slice i into
// the caller's buffer.  The stream must be idle w.r.t. `src` producers (it is
// enqueued behind them) and is synchronised on return.
static void parallel_memcpy(void *dst, const void *src, size_t bytes);
static int copy_out_pipelined(seqalign_ctx *ctx, void *dst, const void *src_dev, size_t bytes) {
  const size_t kSlice = (size_t)32 << 20;
  if (bytes < (size_t)4 << 20) {
    HIP_TRY(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SEQALIGN_OK;
  }
  int rc;
  if ((rc = ctx->h_M.reserve(kSlice)) || (rc = ctx->h_A.reserve(kSlice))) return rc;
  void *pin[2] = {ctx->h_M.p, ctx->h_A.p};
  hipEvent_t ev[2];
  HIP_TRY(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  const size_t n_slices = (bytes + kSlice - 1) / kSlice;
  hipError_t e = hipSuccess;
  for (size_t i = 0; i <= n_slices && e == hipSuccess; ++i) {
    if (i < n_slices) {
      const size_t off = i * kSlice, len = std::min(kSlice, bytes - off);
      e = hipMemcpyAsync(pin[i & 1], static_cast<const char *>(src_dev) + off, len, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipEventRecord(ev[i & 1], ctx->stream);
    }
    if (i > 0 && e == hipSuccess) {
      const size_t j = i - 1, off = j * kSlice, len = std::min(kSlice, bytes - off);
      e = hipEventSynchronize(ev[j & 1]);
      if (e == hipSuccess) parallel_memcpy(static_cast<char *>(dst) + off, pin[j & 1], len);
    }
  }
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  if (e != hipSuccess) return fail_hip(e, "pipelined D2H");
  return SEQALIGN_OK;
}

extern "C" int seqalign_fill_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                   int is_sw, const uint64_t *mat_off, int32_t *M, int32_t *A, int32_t *B,
                                   uint64_t *status) {
  if (!ctx || !scoring || !mat_off || !M || !A || !B) return SEQALIGN_E_ARG;
  int rc = check_batch(batch);
  if (rc) return rc;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  ScoringGuard guard(ctx);
  if ((rc = seqalign_scoring_upload(ctx, scoring, is_sw, &guard.h))) return rc;
  return fill_batch_uploaded(ctx, batch, guard.h, mat_off, M, A, B, status);
}

static int fill_batch_uploaded(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const seqalign_dev_scoring *sc,
                               const uint64_t *mat_off, int32_t *M, int32_t *A, int32_t *B, uint64_t *status) {
  int rc = SEQALIGN_OK;
  int worst = SEQALIGN_OK;
  for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget)) {
    if ((rc = run_chunk(ctx, batch, c, sc, nullptr))) break;
    // copy back: runs of pairs that are contiguous in the caller's arenas go in one piece
    uint64_t k = 0, dev_cell = 0;
    while (k < c.count) {
      uint64_t run_cells = 0, j = k;
      const uint64_t host0 = mat_off[c.first + k];
      while (j < c.count && mat_off[c.first + j] == host0 + run_cells) {
        run_cells += (uint64_t)(batch->len_a[c.first + j] + 1ull) * (batch->len_b[c.first + j] + 1ull);
        ++j;
      }
      const size_t bytes = run_cells * 4;
      if ((rc = copy_out_pipelined(ctx, M + host0, ctx->M.as<int32_t>() + dev_cell, bytes)) ||
          (rc = copy_out_pipelined(ctx, A + host0, ctx->A.as<int32_t>() + dev_cell, bytes)) ||
          (rc = copy_out_pipelined(ctx, B + host0, ctx->B.as<int32_t>() + dev_cell, bytes)))
        break;
      dev_cell += run_cells;
      k = j;
    }
    if (rc) break;
    int src = fetch_status(ctx, c, status);   // also synchronises the stream
    if (src == SEQALIGN_E_UNKNOWN_PAIR) worst = src;
    else if (src) { rc = src; break; }
  }
  return rc ? rc : worst;
}

// ------------------------------------------- legacy single-pair entry point ---
static std::once_flag g_default_once;
static seqalign_ctx *g_default_ctx = nullptr;
static std::mutex g_default_mu;

extern "C" seqalign_ctx_t *sa_default_ctx_or_die(void) {
  std::call_once(g_default_once, [] {
    int dev = 0;
    if (const char *env = getenv("SEQALIGN_DEVICE")) dev = atoi(env);
    int rc = seqalign_ctx_create(dev, &g_default_ctx);
    if (rc != SEQALIGN_OK) {
      fprintf(stderr, "seqalign: cannot open GPU %d: %s (%s)\n"
                      "seqalign: this library has no CPU path; an MI355X (gfx950) is required\n",
              dev, seqalign_strerror(rc), seqalign_last_error());
      exit(EXIT_FAILURE);
    }
  });
  return g_default_ctx;
}

// FNV-1a over everything scoring_lookup can see (header fields, wildcard and swap
// bitsets, and the scores whose bit is set): the legacy per-pair API re-uses the
// uploaded scoring while the caller's scoring_t is unchanged.
static uint64_t scoring_fingerprint(const scoring_t *sc, int is_sw) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  };
  const int head[] = {sc->gap_open, sc->gap_extend, sc->no_start_gap_penalty, sc->no_end_gap_penalty,
                      sc->no_gaps_in_a, sc->no_gaps_in_b, sc->no_mismatches, sc->use_match_mismatch,
                      sc->match, sc->mismatch, sc->case_sensitive, sc->min_penalty, sc->max_penalty, is_sw};
  mix(head, sizeof(head));
  mix(sc->wildcards, sizeof(sc->wildcards));
  mix(sc->swap_set, sizeof(sc->swap_set));
  for (int a = 0; a < 256; ++a) {
    if ((sc->wildcards[a >> 5] >> (a & 31)) & 1u) mix(&sc->wildscores[a], sizeof(int));
    for (int w = 0; w < 8; ++w) {
      uint32_t bits = sc->swap_set[a][w];
      while (bits) {
        const int b = w * 32 + __builtin_ctz(bits);
        bits &= bits - 1;
        mix(&sc->swap_scores[a][b], sizeof(int));
      }
    }
  }
  return h;
}

extern "C" int sa_fill_one_pair(seqalign_ctx_t *ctx, const scoring_t *sc, int is_sw, const char *a, size_t len_a,
                                const char *b, size_t len_b, int32_t *M, int32_t *A, int32_t *B, uint64_t *status) {
  if (len_a > 0xFFFFFFFEull || len_b > 0xFFFFFFFEull) return SEQALIGN_E_TOO_LARGE;
  std::lock_guard<std::mutex> lock(g_default_mu);   // the default context is shared
  HIP_TRY(hipSetDevice(ctx->device));
  const uint64_t fp = scoring_fingerprint(sc, is_sw);
  if (!ctx->cached || ctx->cached_fp != fp || ctx->cached_is_sw != is_sw) {
    if (ctx->cached) seqalign_scoring_release(ctx, ctx->cached);
    ctx->cached = nullptr;
    int rc = seqalign_scoring_upload(ctx, sc, is_sw, &ctx->cached);
    if (rc) return rc;
    ctx->cached_fp = fp;
    ctx->cached_is_sw = is_sw;
  }
  // one arena: a then b
  std::vector<char> arena(len_a + len_b + 1);
  if (len_a) memcpy(arena.data(), a, len_a);
  if (len_b) memcpy(arena.data() + len_a, b, len_b);
  const uint64_t off_a = 0, off_b = len_a, mat_off = 0;
  const uint32_t la = (uint32_t)len_a, lb = (uint32_t)len_b;
  seqalign_batch_t batch;
  batch.n_pairs = 1; batch.arena = arena.data(); batch.arena_bytes = arena.size();
  batch.off_a = &off_a; batch.len_a = &la; batch.off_b = &off_b; batch.len_b = &lb;
  int rc = check_batch(&batch);
  if (rc) return rc;
  return fill_batch_uploaded(ctx, &batch, ctx->cached, &mat_off, M, A, B, status);
}

// ----------------------------------------------- host-level: NW over a batch ---
namespace {

struct Cand { uint32_t idx; int32_t score; };

struct PairHits {
  std::vector<seqalign_sw_hit_t> hits;   // str_off relative to str_a / str_b below
  std::string str_a, str_b;
};


}  // namespace

static void parallel_memcpy(void *dst, const void *src, size_t bytes) {
  const size_t kPiece = (size_t)2 << 20;
  const uint64_t pieces = (bytes + kPiece - 1) / kPiece;
  parallel_for(pieces, [&](uint64_t i) {
    const size_t off = i * kPiece;
    memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, std::min(kPiece, bytes - off));
  });
}

namespace {

// Successive local alignments of one pair in reference order (score desc, column
// asc, index asc), fresh visited mask, at most max_hits (smith_waterman.c:165-277).
// A high-scoring pair can have tens of thousands of cells above min_score and the
// enumeration usually stops after a few hits, so the candidates are heaped (O(n))
// and popped on demand instead of sorted (upstream sorts ~80 % of ALL cells, :159-161).
static int enumerate_hits(const sa_view_t &v, std::vector<Cand> &cand, uint32_t max_hits, PairHits &out) {
  const size_t W = v.len_a + 1, cells = W * (v.len_b + 1);
  auto later = [W](const Cand &x, const Cand &y) {   // true if x comes AFTER y
    if (x.score != y.score) return x.score < y.score;
    const uint32_t cx = x.idx % W, cy = y.idx % W;
    if (cx != cy) return cx > cy;
    return x.idx > y.idx;
  };
  std::make_heap(cand.begin(), cand.end(), later);
  std::vector<uint32_t> seen((cells + 31) / 32, 0u);
  int rc = SEQALIGN_OK;
  while (!cand.empty() && out.hits.size() < max_hits) {
    std::pop_heap(cand.begin(), cand.end(), later);
    const Cand cd = cand.back();
    cand.pop_back();
    if ((seen[cd.idx >> 5] >> (cd.idx & 31)) & 1u) continue;
    size_t x = cd.idx % W, y = cd.idx / W, steps = 0;
    int matrix = MATCH;
    int32_t score = cd.score;
    bool clash = false;
    for (;; ++steps) {   // pass 1: walk to score 0, marking; abandon on a marked cell
      const size_t at = y * W + x;
      if ((seen[at >> 5] >> (at & 31)) & 1u) { clash = true; break; }
      seen[at >> 5] |= 1u << (at & 31);
      if (score == 0) break;
      if ((rc = sa_reverse_move_rc(&v, &matrix, &score, &x, &y))) return rc;
    }
    if (clash) continue;
    const size_t off = out.str_a.size();
    out.str_a.resize(off + steps + 1);
    out.str_b.resize(off + steps + 1);
    char *ra = &out.str_a[off], *rb = &out.str_b[off];
    x = cd.idx % W; y = cd.idx / W; matrix = MATCH; score = cd.score;
    for (size_t w = steps; score > 0;) {   // pass 2: replay, writing right to left
      --w;
      ra[w] = (matrix == GAP_A) ? '-' : v.a[x - 1];
      rb[w] = (matrix == GAP_B) ? '-' : v.b[y - 1];
      if ((rc = sa_reverse_move_rc(&v, &matrix, &score, &x, &y))) return rc;
    }
    ra[steps] = rb[steps] = '\0';
    seqalign_sw_hit_t h;
    h.pair = 0; h.score = cd.score;
    h.pos_a = (uint32_t)x; h.pos_b = (uint32_t)y;
    h.len_a = (uint32_t)(cd.idx % W - x); h.len_b = (uint32_t)(cd.idx / W - y);
    h.length = (uint32_t)steps; h.str_off = off;
    out.hits.push_back(h);
  }
  return SEQALIGN_OK;
}

}  // namespace

static bool traceback_on_host() {
  const char *env = getenv("SEQALIGN_TRACEBACK");
  return env && !strcmp(env, "host");
}

// device traceback of one already-filled chunk; strings land in the caller's buffers
static int nw_chunk_device_traceback(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c,
                                     const seqalign_dev_scoring *sc, const seqalign_dev_batch_t &d,
                                     const uint64_t *str_off, char *out_a, char *out_b, uint32_t *out_len,
                                     int32_t *out_score) {
  const uint64_t n = c.count;
  int rc;
  // per-pair slots of len_a+len_b chars in a compact device arena
  if ((rc = ctx->h_tmeta.reserve(n * 8 + n * 16))) return rc;
  uint64_t *h_off = ctx->h_tmeta.as<uint64_t>();
  uint64_t total = 0;
  for (uint64_t k = 0; k < n; ++k) {
    h_off[k] = total;
    total += (uint64_t)batch->len_a[c.first + k] + batch->len_b[c.first + k];
  }
  if ((rc = ctx->t_str_off.reserve(n * 8)) || (rc = ctx->t_out_a.reserve(total + 16)) ||
      (rc = ctx->t_out_b.reserve(total + 16)) || (rc = ctx->t_meta.reserve(n * 16)) ||
      (rc = ctx->h_ta.reserve(total + 16)) || (rc = ctx->h_tb.reserve(total + 16)))
    return rc;
  hipStream_t st = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->t_str_off.p, h_off, n * 8, hipMemcpyHostToDevice, st));
  uint32_t *d_meta = ctx->t_meta.as<uint32_t>();   // head | len | score | status, n each
  seqalign_trace_t t;
  memset(&t, 0, sizeof(t));
  t.str_off = ctx->t_str_off.as<uint64_t>(); t.out_a = ctx->t_out_a.as<char>(); t.out_b = ctx->t_out_b.as<char>();
  t.out_head = d_meta; t.out_len = d_meta + n; t.out_score = reinterpret_cast<int32_t *>(d_meta + 2 * n);
  t.status = d_meta + 3 * n;
  if ((rc = seqalign_nw_traceback_device(ctx, sc, &d, &t, st))) return rc;
  uint32_t *h_meta = reinterpret_cast<uint32_t *>(h_off + n);
  HIP_TRY(hipMemcpyAsync(ctx->h_ta.p, ctx->t_out_a.p, total, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ctx->h_tb.p, ctx->t_out_b.p, total, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h_meta, d_meta, n * 16, hipMemcpyDeviceToHost, st));
  if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // fill status; synchronises the stream
  const char *ha = ctx->h_ta.as<char>(), *hb = ctx->h_tb.as<char>();
  for (uint64_t k = 0; k < n; ++k)
    if (h_meta[3 * n + k]) return (int)h_meta[3 * n + k];
  constexpr uint64_t kPack = 2048;
  parallel_for((n + kPack - 1) / kPack, [&](uint64_t blk) {
    for (uint64_t k = blk * kPack, e = std::min(n, (blk + 1) * kPack); k < e; ++k) {
      const uint64_t p = c.first + k;
      const uint32_t head = h_meta[k], len = h_meta[n + k];
      memcpy(out_a + str_off[p], ha + h_off[k] + head, len);   // left-align (needleman_wunsch.c:135-145)
      memcpy(out_b + str_off[p], hb + h_off[k] + head, len);
      out_a[str_off[p] + len] = out_b[str_off[p] + len] = '\0';
      out_len[p] = len;
      out_score[p] = reinterpret_cast<const int32_t *>(h_meta)[2 * n + k];
    }
  });
  return SEQALIGN_OK;
}

extern "C" int seqalign_nw_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                 const uint64_t *str_off, char *out_a, char *out_b, uint32_t *out_len,
                                 int32_t *out_score) {
  if (!ctx || !scoring || !str_off || !out_a || !out_b || !out_len || !out_score) return SEQALIGN_E_ARG;
  int rc = check_batch(batch);
  if (rc) return rc;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  ScoringGuard guard(ctx);
  if ((rc = seqalign_scoring_upload(ctx, scoring, 0, &guard.h))) return rc;
  seqalign_dev_scoring *sc = guard.h;
  const bool on_host = traceback_on_host();
  // host mode: matrices come back through pinned staging, so chunks are also bounded by host memory
  const size_t budget = on_host ? std::min<size_t>(ctx->chunk_budget, (size_t)6 << 30) : ctx->chunk_budget;
  for (const Chunk &c : plan_chunks(batch, budget)) {
    seqalign_dev_batch_t d;
    if ((rc = run_chunk(ctx, batch, c, sc, &d))) return rc;
    if (!on_host) {
      if ((rc = nw_chunk_device_traceback(ctx, batch, c, sc, d, str_off, out_a, out_b, out_len, out_score))) return rc;
      continue;
    }
    const size_t bytes = c.cells * 4;
    if ((rc = ctx->h_M.reserve(bytes)) || (rc = ctx->h_A.reserve(bytes)) || (rc = ctx->h_B.reserve(bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_M.p, ctx->M.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_A.p, ctx->A.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_B.p, ctx->B.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // syncs; unknown pair is fatal for NW
    std::vector<uint64_t> cell0(c.count);
    { uint64_t cell = 0;
      for (uint64_t k = 0; k < c.count; ++k) {
        cell0[k] = cell;
        cell += (uint64_t)(batch->len_a[c.first + k] + 1ull) * (batch->len_b[c.first + k] + 1ull);
      } }
    std::atomic<int> first_error{SEQALIGN_OK};
    parallel_for(c.count, [&](uint64_t k) {
      const uint64_t p = c.first + k;
      sa_view_t v;
      v.sc = scoring; v.a = batch->arena + batch->off_a[p]; v.b = batch->arena + batch->off_b[p];
      v.len_a = batch->len_a[p]; v.len_b = batch->len_b[p];
      v.M = ctx->h_M.as<int32_t>() + cell0[k]; v.A = ctx->h_A.as<int32_t>() + cell0[k];
      v.B = ctx->h_B.as<int32_t>() + cell0[k];
      size_t n = 0;
      int prc = sa_nw_traceback(&v, out_a + str_off[p], out_b + str_off[p], &n, &out_score[p]);
      out_len[p] = (uint32_t)n;
      if (prc != SEQALIGN_OK) { int expected = SEQALIGN_OK; first_error.compare_exchange_strong(expected, prc); }
    });
    if ((rc = first_error.load())) return rc;
  }
  return SEQALIGN_OK;
}

// ----------------------------------------------- host-level: SW over a batch ---

// up to this many hits per pair the enumeration runs on the device
static const uint32_t kDeviceEnumMaxHits = 16;

// SW hits of one already-filled chunk, enumerated on the device (sa_sw_enum.hip):
// reduce (count) -> reduce (compact + keys) -> segmented sort -> enumerate ->
// gather strings -> D2H.  Appends to the caller's hit array / string buffers.
//
// want_hits > max_hits (the caller asked for more hits than the device slots hold): pairs that fill all
// max_hits slots with candidates still left are finished on the host -- their matrices and candidates are
// still in the context's scratch -- with the full limit; the others (nearly all, in practice) are done.
static int sw_chunk_device_enumerate(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c,
                                     const scoring_t *scoring, const seqalign_dev_scoring *sc,
                                     const seqalign_dev_batch_t &d, const int32_t *min_score, uint32_t max_hits,
                                     uint32_t want_hits, seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *found,
                                     char *out_a, char *out_b, uint64_t str_cap, uint64_t *used_str) {
  const uint64_t n = c.count;
  hipStream_t st = ctx->stream;
  int rc;
  DevBuf &d_min = ctx->e[0], &d_key_in = ctx->e[1], &d_key_out = ctx->e[2], &d_idx_out = ctx->e[3],
         &d_tmp = ctx->e[4], &d_mask = ctx->e[5], &d_offs = ctx->e[6], &d_hits = ctx->e[7], &d_meta = ctx->e[8],
         &d_gath_a = ctx->e[9], &d_gath_b = ctx->e[10];

  int32_t thr = min_score[c.first];
  for (uint64_t k = 1; k < n; ++k) thr = std::min(thr, min_score[c.first + k]);

  // pass 1: how many cells >= threshold per pair
  if ((rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8)) ||
      (rc = ctx->cand_count.reserve(n * 4)) || (rc = ctx->cand_off.reserve((n + 1) * 8)) ||
      (rc = ctx->cand_cap.reserve(n * 4)) || (rc = d_min.reserve(n * 4)))
    return rc;
  SaReduceParams r;
  memset(&r, 0, sizeof(r));
  r.len_a = d.len_a; r.len_b = d.len_b; r.mat_off = d.mat_off; r.M = d.match_scores; r.min_score = thr;
  r.best_score = ctx->best_score.as<int32_t>(); r.best_index = ctx->best_index.as<uint64_t>();
  r.cand_count = ctx->cand_count.as<uint32_t>(); r.n_pairs = (uint32_t)n;
  hipError_t e = sa_launch_sw_reduce(r, st);
  if (e != hipSuccess) return fail_hip(e, "sw reduce");
  std::vector<uint32_t> count(n);
  std::vector<int32_t> best(n);
  HIP_TRY(hipMemcpyAsync(count.data(), ctx->cand_count.p, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(best.data(), ctx->best_score.p, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(d_min.p, min_score + c.first, n * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  // sort key = (cap - score) << column_bits | column: only its used bits are sorted
  int32_t key_cap = thr;
  uint32_t max_la = 1;
  for (uint64_t k = 0; k < n; ++k) {
    key_cap = std::max(key_cap, best[k]);
    max_la = std::max(max_la, batch->len_a[c.first + k]);
  }
  uint32_t key_shift = 1, span_bits = 1;
  while ((max_la >> key_shift) != 0) ++key_shift;                                   // column <= len_a
  while (span_bits < 32 && (((uint64_t)key_cap - (uint64_t)(int64_t)thr) >> span_bits) != 0) ++span_bits;

  // host prefixes: candidate segments, visited-bitmap words, string slots
  std::vector<uint64_t> offs(4 * (n + 1));
  uint64_t *cand_off = offs.data(), *mask_off = cand_off + n + 1, *str_off = mask_off + n + 1,
           *dst_off = str_off + n + 1;
  uint64_t total = 0, mask_words = 0, str_total = 0, max_mask_words = 0;
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t p = c.first + k, la = batch->len_a[p], lb = batch->len_b[p];
    cand_off[k] = total; total += count[k];
    mask_off[k] = mask_words; mask_words += ((la + 1) * (lb + 1) + 31) / 32;
    max_mask_words = std::max(max_mask_words, ((la + 1) * (lb + 1) + 31) / 32);
    str_off[k] = str_total; str_total += (uint64_t)max_hits * (la + lb);
  }
  cand_off[n] = total; mask_off[n] = mask_words; str_off[n] = str_total;
  if (total >= (1ull << 31)) return SEQALIGN_E_TOO_LARGE;

  if ((rc = ctx->cand_index.reserve(total * 4 + 4)) || (rc = d_key_in.reserve(total * 8 + 8)) ||
      (rc = d_key_out.reserve(total * 8 + 8)) || (rc = d_idx_out.reserve(total * 4 + 4)) ||
      (rc = d_mask.reserve(mask_words * 4 + 4)) || (rc = d_offs.reserve(offs.size() * 8)) ||
      (rc = ctx->t_out_a.reserve(str_total + 16)) || (rc = ctx->t_out_b.reserve(str_total + 16)) ||
      (rc = d_hits.reserve(n * max_hits * sizeof(SaDevHit) + 16)) || (rc = d_meta.reserve(n * 12)))
    return rc;
  HIP_TRY(hipMemcpyAsync(d_offs.p, offs.data(), 3 * (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ctx->cand_cap.p, count.data(), n * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(d_mask.p, 0, mask_words * 4, st));
  const uint64_t *dv_cand_off = d_offs.as<uint64_t>(), *dv_mask_off = dv_cand_off + n + 1,
                 *dv_str_off = dv_mask_off + n + 1;

  // pass 2: compaction with sort keys, then the stable segmented sort
  r.cand_off = dv_cand_off; r.cand_cap = ctx->cand_cap.as<uint32_t>();
  r.cand_index = ctx->cand_index.as<uint32_t>(); r.cand_key = d_key_in.as<uint64_t>();
  r.key_cap = key_cap; r.key_shift = key_shift;
  if ((e = sa_launch_sw_reduce(r, st)) != hipSuccess) return fail_hip(e, "sw reduce (compaction)");
  if (total) {
    size_t tmp_bytes = 0;
    e = sa_sort_candidates(nullptr, &tmp_bytes, d_key_in.as<uint64_t>(), d_key_out.as<uint64_t>(),
                           ctx->cand_index.as<uint32_t>(), d_idx_out.as<uint32_t>(), total, (uint32_t)n, dv_cand_off,
                           (int)(key_shift + span_bits), st);
    if (e != hipSuccess) return fail_hip(e, "segmented sort (size query)");
    if ((rc = d_tmp.reserve(tmp_bytes + 16))) return rc;
    e = sa_sort_candidates(d_tmp.p, &tmp_bytes, d_key_in.as<uint64_t>(), d_key_out.as<uint64_t>(),
                           ctx->cand_index.as<uint32_t>(), d_idx_out.as<uint32_t>(), total, (uint32_t)n, dv_cand_off,
                           (int)(key_shift + span_bits), st);
    if (e != hipSuccess) return fail_hip(e, "segmented sort");
  }

  // enumeration: one lane per pair
  SaEnumParams q;
  memset(&q, 0, sizeof(q));
  q.arena = d.arena; q.off_a = d.off_a; q.len_a = d.len_a; q.off_b = d.off_b; q.len_b = d.len_b;
  q.mat_off = d.mat_off; q.M = d.match_scores; q.A = d.gap_a_scores; q.B = d.gap_b_scores;
  q.code = sc->d_code; q.table = sc->d_table; q.cand_off = dv_cand_off; q.cand_count = ctx->cand_count.as<uint32_t>();
  q.sorted_key = d_key_out.as<uint64_t>(); q.sorted_index = d_idx_out.as<uint32_t>(); q.min_score = d_min.as<int32_t>();
  q.mask = d_mask.as<uint32_t>(); q.mask_off = dv_mask_off; q.str_off = dv_str_off;
  q.out_a = ctx->t_out_a.as<char>(); q.out_b = ctx->t_out_b.as<char>(); q.hits = d_hits.as<SaDevHit>();
  uint32_t *d_m = d_meta.as<uint32_t>();
  q.hit_count = d_m; q.str_used = d_m + n; q.enum_status = d_m + 2 * n;
  q.n_pairs = (uint32_t)n; q.K = sc->flat.n_classes; q.max_hits = max_hits; q.open1 = sc->flat.open1;
  q.ext = sc->flat.ext; q.gen_eq = sc->flat.gen_eq; q.gen_ne = sc->flat.gen_ne; q.flags = sc->flat.flags;
  q.max_mask_words = (uint32_t)std::min<uint64_t>(max_mask_words, 0xffffffffu);
  q.key_cap = key_cap; q.key_shift = key_shift;
  if ((e = sa_launch_sw_enumerate(q, st)) != hipSuccess) return fail_hip(e, "sw enumerate");

  std::vector<uint32_t> meta(3 * n);
  std::vector<SaDevHit> dev_hits(n * max_hits);
  HIP_TRY(hipMemcpyAsync(meta.data(), d_meta.p, n * 12, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(dev_hits.data(), d_hits.p, n * max_hits * sizeof(SaDevHit), hipMemcpyDeviceToHost, st));
  if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // fill status; synchronises the stream
  uint64_t gathered = 0;
  for (uint64_t k = 0; k < n; ++k) {
    const uint32_t status = meta[2 * n + k] & 0x7fffffffu;
    if (status) return (int)status;
    dst_off[k] = gathered;
    gathered += meta[n + k];
  }
  // pairs that ran into the slot limit while the caller wants more: host enumeration with the full limit
  std::vector<uint64_t> capped;
  if (want_hits > max_hits)
    for (uint64_t k = 0; k < n; ++k)
      if ((meta[2 * n + k] & 0x80000000u) && meta[k] >= max_hits) capped.push_back(k);
  std::vector<PairHits> redo(capped.size());
  std::vector<int64_t> redo_of(capped.empty() ? 0 : n, -1);
  if (!capped.empty()) {
    std::vector<uint64_t> cell0(n + 1, 0);
    for (uint64_t k = 0; k < n; ++k)
      cell0[k + 1] = cell0[k] + (uint64_t)(batch->len_a[c.first + k] + 1ull) * (batch->len_b[c.first + k] + 1ull);
    std::vector<uint64_t> m_off(capped.size() + 1, 0), c_off(capped.size() + 1, 0);
    for (size_t j = 0; j < capped.size(); ++j) {
      const uint64_t k = capped[j];
      redo_of[k] = (int64_t)j;
      m_off[j + 1] = m_off[j] + (cell0[k + 1] - cell0[k]);
      c_off[j + 1] = c_off[j] + count[k];
    }
    std::vector<int32_t> hM(m_off.back() + 1), hA(m_off.back() + 1), hB(m_off.back() + 1);
    std::vector<uint32_t> h_idx(c_off.back() + 1);
    std::vector<uint64_t> h_key(c_off.back() + 1);
    for (size_t j = 0; j < capped.size(); ++j) {
      const uint64_t k = capped[j], cells = cell0[k + 1] - cell0[k];
      HIP_TRY(hipMemcpyAsync(hM.data() + m_off[j], ctx->M.as<int32_t>() + cell0[k], cells * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(hA.data() + m_off[j], ctx->A.as<int32_t>() + cell0[k], cells * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(hB.data() + m_off[j], ctx->B.as<int32_t>() + cell0[k], cells * 4, hipMemcpyDeviceToHost, st));
      if (count[k]) {
        HIP_TRY(hipMemcpyAsync(h_idx.data() + c_off[j], ctx->cand_index.as<uint32_t>() + cand_off[k], count[k] * 4ull,
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_key.data() + c_off[j], d_key_in.as<uint64_t>() + cand_off[k], count[k] * 8ull,
                               hipMemcpyDeviceToHost, st));
      }
    }
    HIP_TRY(hipStreamSynchronize(st));
    std::atomic<int> first_error{SEQALIGN_OK};
    parallel_for(capped.size(), [&](uint64_t j) {
      const uint64_t k = capped[j], p = c.first + k;
      sa_view_t v;
      v.sc = scoring; v.a = batch->arena + batch->off_a[p]; v.b = batch->arena + batch->off_b[p];
      v.len_a = batch->len_a[p]; v.len_b = batch->len_b[p];
      v.M = hM.data() + m_off[j]; v.A = hA.data() + m_off[j]; v.B = hB.data() + m_off[j];
      std::vector<Cand> cand;
      cand.reserve(count[k]);
      for (uint64_t q = 0; q < count[k]; ++q) {
        const Cand cd{h_idx[c_off[j] + q], key_cap - (int32_t)(h_key[c_off[j] + q] >> key_shift)};
        if (cd.score >= min_score[p]) cand.push_back(cd);
      }
      const int prc = enumerate_hits(v, cand, want_hits, redo[j]);
      if (prc != SEQALIGN_OK) { int expected = SEQALIGN_OK; first_error.compare_exchange_strong(expected, prc); }
    });
    if ((rc = first_error.load())) return rc;
  }

  // pack every pair's strings back to back and bring them over in one copy
  if ((rc = d_gath_a.reserve(gathered + 16)) || (rc = d_gath_b.reserve(gathered + 16)) ||
      (rc = ctx->h_ta.reserve(gathered + 16)) || (rc = ctx->h_tb.reserve(gathered + 16)))
    return rc;
  uint64_t *dv_dst_off = d_offs.as<uint64_t>() + 3 * (n + 1);
  HIP_TRY(hipMemcpyAsync(dv_dst_off, dst_off, n * 8, hipMemcpyHostToDevice, st));
  if ((e = sa_launch_gather_strings(q.out_a, q.out_b, dv_str_off, q.str_used, dv_dst_off, d_gath_a.as<char>(),
                                    d_gath_b.as<char>(), (uint32_t)n, st)) != hipSuccess)
    return fail_hip(e, "gather strings");
  HIP_TRY(hipMemcpyAsync(ctx->h_ta.p, d_gath_a.p, gathered, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ctx->h_tb.p, d_gath_b.p, gathered, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));

  const char *ha = ctx->h_ta.as<char>(), *hb = ctx->h_tb.as<char>();
  for (uint64_t k = 0; k < n; ++k) {
    if (!capped.empty() && redo_of[k] >= 0) {   // finished on the host
      const PairHits &ph = redo[(size_t)redo_of[k]];
      for (const seqalign_sw_hit_t &src : ph.hits) {
        if (*found >= hit_cap || *used_str + src.length + 1 > str_cap) return SEQALIGN_E_NOMEM;
        memcpy(out_a + *used_str, ph.str_a.data() + src.str_off, src.length + 1);
        memcpy(out_b + *used_str, ph.str_b.data() + src.str_off, src.length + 1);
        seqalign_sw_hit_t &h = hits[(*found)++];
        h = src; h.pair = c.first + k; h.str_off = *used_str;
        *used_str += src.length + 1;
      }
      continue;
    }
    for (uint32_t i = 0; i < meta[k]; ++i) {
      const SaDevHit &src = dev_hits[k * max_hits + i];
      if (*found >= hit_cap || *used_str + src.length + 1 > str_cap) return SEQALIGN_E_NOMEM;
      memcpy(out_a + *used_str, ha + dst_off[k] + src.str_off, src.length);
      memcpy(out_b + *used_str, hb + dst_off[k] + src.str_off, src.length);
      out_a[*used_str + src.length] = out_b[*used_str + src.length] = '\0';
      seqalign_sw_hit_t &h = hits[(*found)++];
      h.pair = c.first + k; h.score = src.score; h.pos_a = src.pos_a; h.pos_b = src.pos_b;
      h.len_a = src.len_a; h.len_b = src.len_b; h.length = src.length; h.str_off = *used_str;
      *used_str += src.length + 1;
    }
  }
  return SEQALIGN_OK;
}

extern "C" int seqalign_sw_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                 const int32_t *min_score, uint32_t max_hits, seqalign_sw_hit_t *hits,
                                 uint64_t hit_cap, uint64_t *n_hits, char *out_a, char *out_b, uint64_t str_cap) {
  if (!ctx || !scoring || !min_score || !hits || !n_hits || !out_a || !out_b) return SEQALIGN_E_ARG;
  *n_hits = 0;
  int rc = check_batch(batch);
  if (rc) return rc;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  ScoringGuard guard(ctx);
  if ((rc = seqalign_scoring_upload(ctx, scoring, 1, &guard.h))) return rc;
  seqalign_dev_scoring *sc = guard.h;
  uint64_t used_str = 0, found = 0;
  if (max_hits == 0) return SEQALIGN_OK;
  if (max_hits == 1 && !traceback_on_host()) {
    // best hit only: nothing but the strings crosses PCIe
    for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget)) {
      seqalign_dev_batch_t d;
      bool have_best = false;   // the stream kernel reports the best cell itself
      if ((rc = run_chunk(ctx, batch, c, sc, &d, &have_best))) return rc;
      const uint64_t n = c.count;
      if (!have_best) {
        if ((rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8))) return rc;
        seqalign_sw_reduce_t r;
        memset(&r, 0, sizeof(r));
        r.n_pairs = n; r.len_a = d.len_a; r.len_b = d.len_b; r.mat_off = d.mat_off; r.match_scores = d.match_scores;
        r.min_score = 1; r.best_score = ctx->best_score.as<int32_t>(); r.best_index = ctx->best_index.as<uint64_t>();
        if ((rc = seqalign_sw_reduce_device(ctx, &r, ctx->stream))) return rc;
      }
      if ((rc = ctx->h_tmeta.reserve(n * 8 + n * 32))) return rc;
      uint64_t *h_off = ctx->h_tmeta.as<uint64_t>();
      uint64_t total = 0;
      for (uint64_t k = 0; k < n; ++k) {
        h_off[k] = total;
        total += (uint64_t)batch->len_a[c.first + k] + batch->len_b[c.first + k];
      }
      if ((rc = ctx->t_str_off.reserve(n * 8)) || (rc = ctx->t_out_a.reserve(total + 16)) ||
          (rc = ctx->t_out_b.reserve(total + 16)) || (rc = ctx->t_meta.reserve(n * 32)) ||
          (rc = ctx->h_ta.reserve(total + 16)) || (rc = ctx->h_tb.reserve(total + 16)))
        return rc;
      hipStream_t st = ctx->stream;
      HIP_TRY(hipMemcpyAsync(ctx->t_str_off.p, h_off, n * 8, hipMemcpyHostToDevice, st));
      uint32_t *d_meta = ctx->t_meta.as<uint32_t>();   // head | len | score | status | pos[4]
      seqalign_trace_t t;
      memset(&t, 0, sizeof(t));
      t.str_off = ctx->t_str_off.as<uint64_t>(); t.out_a = ctx->t_out_a.as<char>(); t.out_b = ctx->t_out_b.as<char>();
      t.out_head = d_meta; t.out_len = d_meta + n; t.out_score = reinterpret_cast<int32_t *>(d_meta + 2 * n);
      t.status = d_meta + 3 * n; t.out_pos = d_meta + 4 * n; t.start_index = ctx->best_index.as<uint64_t>();
      if ((rc = seqalign_sw_traceback_device(ctx, sc, &d, &t, st))) return rc;
      uint32_t *h_meta = reinterpret_cast<uint32_t *>(h_off + n);
      HIP_TRY(hipMemcpyAsync(ctx->h_ta.p, ctx->t_out_a.p, total, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(ctx->h_tb.p, ctx->t_out_b.p, total, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(h_meta, d_meta, n * 32, hipMemcpyDeviceToHost, st));
      if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // syncs
      const char *ha = ctx->h_ta.as<char>(), *hb = ctx->h_tb.as<char>();
      for (uint64_t k = 0; k < n; ++k) {
        const uint64_t p = c.first + k;
        const uint32_t head = h_meta[k], len = h_meta[n + k], status = h_meta[3 * n + k];
        const int32_t score = reinterpret_cast<const int32_t *>(h_meta)[2 * n + k];
        if (status) return (int)status;
        if (score <= 0 || score < min_score[p]) continue;
        if (found >= hit_cap || used_str + len + 1 > str_cap) { *n_hits = found; return SEQALIGN_E_NOMEM; }
        memcpy(out_a + used_str, ha + h_off[k] + head, len);
        memcpy(out_b + used_str, hb + h_off[k] + head, len);
        out_a[used_str + len] = out_b[used_str + len] = '\0';
        seqalign_sw_hit_t &h = hits[found++];
        const uint32_t *pos = h_meta + 4 * n + 4 * k;
        h.pair = p; h.score = score; h.pos_a = pos[0]; h.pos_b = pos[1]; h.len_a = pos[2]; h.len_b = pos[3];
        h.length = len; h.str_off = used_str;
        used_str += len + 1;
      }
    }
    *n_hits = found;
    return SEQALIGN_OK;
  }
  if (!traceback_on_host()) {
    // up to kDeviceEnumMaxHits hits per pair on the device; a pair that needs more is finished on the host
    const uint32_t slots = std::min(max_hits, kDeviceEnumMaxHits);
    for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget / 2)) {
      seqalign_dev_batch_t d;
      if ((rc = run_chunk(ctx, batch, c, sc, &d))) break;
      if ((rc = sw_chunk_device_enumerate(ctx, batch, c, scoring, sc, d, min_score, slots, max_hits, hits, hit_cap,
                                          &found, out_a, out_b, str_cap, &used_str)))
        break;
    }
    *n_hits = found;
    return rc;
  }
  const size_t budget = std::min<size_t>(ctx->chunk_budget, (size_t)6 << 30);

  // the reduction kernel takes one threshold per launch: group by threshold
  // inside a chunk (the CLI default depends only on the lengths, so batches of
  // equal-length pairs need a single launch)
  for (const Chunk &c : plan_chunks(batch, budget)) {
    seqalign_dev_batch_t d;
    if ((rc = run_chunk(ctx, batch, c, sc, &d))) break;
    const uint64_t n = c.count;
    int32_t thr = min_score[c.first];
    for (uint64_t k = 1; k < n; ++k) thr = std::min(thr, min_score[c.first + k]);

    // pass 1: counts (capacity 0), pass 2: compaction
    if ((rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8)) ||
        (rc = ctx->cand_count.reserve(n * 4)) || (rc = ctx->cand_off.reserve(n * 8)) ||
        (rc = ctx->cand_cap.reserve(n * 4)))
      break;
    seqalign_sw_reduce_t r;
    memset(&r, 0, sizeof(r));
    r.n_pairs = n; r.len_a = d.len_a; r.len_b = d.len_b; r.mat_off = d.mat_off; r.match_scores = d.match_scores;
    r.min_score = thr; r.best_score = ctx->best_score.as<int32_t>(); r.best_index = ctx->best_index.as<uint64_t>();
    r.cand_count = ctx->cand_count.as<uint32_t>();
    if ((rc = seqalign_sw_reduce_device(ctx, &r, ctx->stream))) break;
    if ((rc = ctx->h_misc.reserve(n * (4 + 8 + 4)))) break;
    uint32_t *h_count = ctx->h_misc.as<uint32_t>();
    HIP_TRY(hipMemcpyAsync(h_count, ctx->cand_count.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> c_off(n);
    std::vector<uint32_t> c_cap(h_count, h_count + n);
    uint64_t total = 0;
    for (uint64_t k = 0; k < n; ++k) { c_off[k] = total; total += c_cap[k]; }
    if ((rc = ctx->cand_index.reserve(total * 4 + 4)) || (rc = ctx->cand_score.reserve(total * 4 + 4))) break;
    HIP_TRY(hipMemcpyAsync(ctx->cand_off.p, c_off.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->cand_cap.p, c_cap.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    r.cand_off = ctx->cand_off.as<uint64_t>(); r.cand_cap = ctx->cand_cap.as<uint32_t>();
    r.cand_index = ctx->cand_index.as<uint32_t>(); r.cand_score = ctx->cand_score.as<int32_t>();
    if ((rc = seqalign_sw_reduce_device(ctx, &r, ctx->stream))) break;

    const size_t bytes = c.cells * 4;
    if ((rc = ctx->h_M.reserve(bytes)) || (rc = ctx->h_A.reserve(bytes)) || (rc = ctx->h_B.reserve(bytes))) break;
    std::vector<uint32_t> h_cidx(total + 1);
    std::vector<int32_t> h_cscore(total + 1);
    hipError_t e = hipMemcpyAsync(ctx->h_M.p, ctx->M.p, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_A.p, ctx->A.p, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_B.p, ctx->B.p, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && total) e = hipMemcpyAsync(h_cidx.data(), ctx->cand_index.p, total * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && total) e = hipMemcpyAsync(h_cscore.data(), ctx->cand_score.p, total * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) { rc = fail_hip(e, "D2H SW results"); break; }
    if ((rc = fetch_status(ctx, c, nullptr))) break;

    // host: hit enumeration with a fresh visited mask per pair (reference
    // smith_waterman.c:165-277 semantics).  Pairs are independent -> host threads.
    std::vector<uint64_t> cell0(n);
    { uint64_t cell = 0;
      for (uint64_t k = 0; k < n; ++k) {
        cell0[k] = cell;
        cell += (uint64_t)(batch->len_a[c.first + k] + 1ull) * (batch->len_b[c.first + k] + 1ull);
      } }
    std::vector<PairHits> per_pair(n);
    std::atomic<int> first_error{SEQALIGN_OK};
    parallel_for(n, [&](uint64_t k) {
      const uint64_t p = c.first + k;
      sa_view_t v;
      v.sc = scoring; v.a = batch->arena + batch->off_a[p]; v.b = batch->arena + batch->off_b[p];
      v.len_a = batch->len_a[p]; v.len_b = batch->len_b[p];
      v.M = ctx->h_M.as<int32_t>() + cell0[k]; v.A = ctx->h_A.as<int32_t>() + cell0[k];
      v.B = ctx->h_B.as<int32_t>() + cell0[k];
      std::vector<Cand> cand;
      cand.reserve(c_cap[k]);
      for (uint32_t q = 0; q < c_cap[k]; ++q) {
        const Cand cd{h_cidx[c_off[k] + q], h_cscore[c_off[k] + q]};
        if (cd.score >= min_score[p]) cand.push_back(cd);
      }
      int prc = enumerate_hits(v, cand, max_hits, per_pair[k]);
      if (prc != SEQALIGN_OK) { int expected = SEQALIGN_OK; first_error.compare_exchange_strong(expected, prc); }
    });
    if ((rc = first_error.load())) break;
    for (uint64_t k = 0; k < n && rc == SEQALIGN_OK; ++k) {
      const PairHits &ph = per_pair[k];
      for (size_t i = 0; i < ph.hits.size(); ++i) {
        const seqalign_sw_hit_t &src = ph.hits[i];
        if (found >= hit_cap || used_str + src.length + 1 > str_cap) { rc = SEQALIGN_E_NOMEM; break; }
        memcpy(out_a + used_str, ph.str_a.data() + src.str_off, src.length + 1);
        memcpy(out_b + used_str, ph.str_b.data() + src.str_off, src.length + 1);
        seqalign_sw_hit_t &h = hits[found++];
        h = src;
        h.pair = c.first + k;
        h.str_off = used_str;
        used_str += src.length + 1;
      }
    }
    if (rc) break;
  }
  *n_hits = found;
  return rc;
}

// ------------------------------------------------- several contexts (GPUs) ---
namespace {

// pairs [first, first + count) of a batch as a batch of their own (views, nothing copied)
seqalign_batch_t sub_batch(const seqalign_batch_t *b, uint64_t first, uint64_t count) {
  seqalign_batch_t s = *b;
  s.n_pairs = count;
  s.off_a = b->off_a + first; s.len_a = b->len_a + first;
  s.off_b = b->off_b + first; s.len_b = b->len_b + first;
  return s;
}

// run fn(g, first, count) for the n_ctx contiguous ranges, one host thread each; first error wins
template <class F>
int for_each_shard(int n_ctx, uint64_t n_pairs, F fn) {
  std::vector<int> rc((size_t)n_ctx, SEQALIGN_OK);
  std::vector<std::string> msg((size_t)n_ctx);
  std::vector<std::thread> th;
  for (int g = 0; g < n_ctx; ++g) {
    const uint64_t first = n_pairs * (uint64_t)g / (uint64_t)n_ctx, last = n_pairs * (uint64_t)(g + 1) / (uint64_t)n_ctx;
    th.emplace_back([&, g, first, last] {
      rc[g] = last > first ? fn(g, first, last - first) : SEQALIGN_OK;
      if (rc[g]) msg[g] = seqalign_last_error();   // the message lives in the worker's thread
    });
  }
  for (auto &t : th) t.join();
  for (int g = 0; g < n_ctx; ++g)
    if (rc[g]) { g_last_error = msg[g]; return rc[g]; }
  return SEQALIGN_OK;
}

bool bad_ctx_list(seqalign_ctx_t *const *ctxs, int n_ctx) {
  if (!ctxs || n_ctx <= 0) return true;
  for (int g = 0; g < n_ctx; ++g)
    if (!ctxs[g]) return true;
  return false;
}

}  // namespace

extern "C" int seqalign_fill_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch,
                                         const scoring_t *scoring, int is_sw, const uint64_t *mat_off,
                                         int32_t *match_scores, int32_t *gap_a_scores, int32_t *gap_b_scores,
                                         uint64_t *status) {
  if (bad_ctx_list(ctxs, n_ctx) || !batch || !mat_off) return SEQALIGN_E_ARG;
  int rc = check_batch(batch);
  if (rc) return rc;
  return for_each_shard(n_ctx, batch->n_pairs, [&](int g, uint64_t first, uint64_t count) {
    const seqalign_batch_t s = sub_batch(batch, first, count);   // mat_off[] are absolute cell offsets: shared arenas
    return seqalign_fill_batch(ctxs[g], &s, scoring, is_sw, mat_off + first, match_scores, gap_a_scores, gap_b_scores,
                               status ? status + first : nullptr);
  });
}

extern "C" int seqalign_nw_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch,
                                       const scoring_t *scoring, const uint64_t *str_off, char *out_a, char *out_b,
                                       uint32_t *out_len, int32_t *out_score) {
  if (bad_ctx_list(ctxs, n_ctx) || !batch || !str_off || !out_len || !out_score) return SEQALIGN_E_ARG;
  int rc = check_batch(batch);
  if (rc) return rc;
  return for_each_shard(n_ctx, batch->n_pairs, [&](int g, uint64_t first, uint64_t count) {
    const seqalign_batch_t s = sub_batch(batch, first, count);   // str_off[] are absolute: shared string buffers
    return seqalign_nw_batch(ctxs[g], &s, scoring, str_off + first, out_a, out_b, out_len + first, out_score + first);
  });
}

extern "C" int seqalign_sw_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch,
                                       const scoring_t *scoring, const int32_t *min_score, uint32_t max_hits,
                                       seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *n_hits, char *out_a,
                                       char *out_b, uint64_t str_cap) {
  if (bad_ctx_list(ctxs, n_ctx) || !batch || !min_score || !hits || !n_hits || !out_a || !out_b) return SEQALIGN_E_ARG;
  *n_hits = 0;
  int rc = check_batch(batch);
  if (rc) return rc;
  const uint64_t n = batch->n_pairs;
  if (n == 0) return SEQALIGN_OK;
  // every range writes into its own slice of the caller's buffers; the slices are closed up afterwards
  std::vector<uint64_t> h0((size_t)n_ctx + 1), s0((size_t)n_ctx + 1), got((size_t)n_ctx, 0), used((size_t)n_ctx, 0);
  for (int g = 0; g <= n_ctx; ++g) {
    const uint64_t first = n * (uint64_t)g / (uint64_t)n_ctx;
    h0[g] = (uint64_t)((long double)hit_cap * first / n);
    s0[g] = (uint64_t)((long double)str_cap * first / n);
  }
  rc = for_each_shard(n_ctx, n, [&](int g, uint64_t first, uint64_t count) {
    const seqalign_batch_t s = sub_batch(batch, first, count);
    uint64_t found = 0;
    const int r = seqalign_sw_batch(ctxs[g], &s, scoring, min_score + first, max_hits, hits + h0[g], h0[g + 1] - h0[g],
                                    &found, out_a + s0[g], out_b + s0[g], s0[g + 1] - s0[g]);
    got[g] = found;
    for (uint64_t i = 0; i < found; ++i) {
      const seqalign_sw_hit_t &h = hits[h0[g] + i];
      used[g] = std::max(used[g], h.str_off + h.length + 1);
    }
    return r;
  });
  if (rc) return rc;
  uint64_t nh = 0, ns = 0;
  for (int g = 0; g < n_ctx; ++g) {
    const uint64_t first = n * (uint64_t)g / (uint64_t)n_ctx;
    if (s0[g] != ns) {
      memmove(out_a + ns, out_a + s0[g], used[g]);
      memmove(out_b + ns, out_b + s0[g], used[g]);
    }
    for (uint64_t i = 0; i < got[g]; ++i) {
      seqalign_sw_hit_t h = hits[h0[g] + i];
      h.pair += first;            // range-relative -> batch index
      h.str_off += ns;            // slice-relative -> buffer offset
      hits[nh++] = h;
    }
    ns += used[g];
  }
  *n_hits = nh;
  return SEQALIGN_OK;
}

// ------------------------------------------------------------------- probes ---
extern "C" int sa_dpp_probe(seqalign_ctx_t *ctx, int32_t fill, int32_t *out64) {
  if (!ctx || !out64) return SEQALIGN_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ctx->status.reserve(64 * 8);
  if (rc) return rc;
  hipError_t e = sa_launch_dpp_probe(ctx->status.as<int32_t>(), fill, ctx->stream);
  if (e != hipSuccess) return fail_hip(e, "dpp probe");
  HIP_TRY(hipMemcpyAsync(out64, ctx->status.p, 64 * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SEQALIGN_OK;
}
