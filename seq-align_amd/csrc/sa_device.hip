// sa_device.hip -- the C-ABI shim (include/seqalign_hip.h) over the HIP kernels.
//
// Host code above this file is C; this file is the only place that talks to the
// HIP runtime.  No torch / C++ types cross the boundary.  No CPU fallback: when
// the runtime or a device is missing every entry point reports it.
#include "sa_ctx.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>

#include <errno.h>
#include <sched.h>
#include <strings.h>
#include <time.h>
#include <sys/syscall.h>
#include <unistd.h>

using namespace sa_host;

// ------------------------------------------------------------------ errors ---
static thread_local std::string g_last_error;

int sa_host::fail_hip(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? SEQALIGN_E_NO_DEVICE
                                                               : SEQALIGN_E_HIP;
}
void sa_host::set_last_error(const std::string &msg) { g_last_error = msg; }

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

// ------------------------------------------------ what a call launched ---
static thread_local seqalign_call_info_t *tl_recorder = nullptr;

void sa_record_launch(int kind, uint64_t items) {
  if (!tl_recorder || kind < 0 || kind >= SEQALIGN_K_COUNT) return;
  tl_recorder->launches[kind] += 1;
  tl_recorder->items[kind] += items;
}

sa_host::CallScope::CallScope(seqalign_ctx *c) : ctx(c), prev(tl_recorder) {
  if (!ctx) return;
  if (ctx->call_depth++ == 0) memset(&ctx->call_info, 0, sizeof(ctx->call_info));
  tl_recorder = &ctx->call_info;
}
sa_host::CallScope::~CallScope() {
  if (!ctx) return;
  --ctx->call_depth;
  tl_recorder = prev;
}

extern "C" int seqalign_ctx_last_call_info(const seqalign_ctx_t *ctx, seqalign_call_info_t *out) {
  if (!ctx || !out) return SEQALIGN_E_ARG;
  *out = ctx->call_info;
  return SEQALIGN_OK;
}

extern "C" const char *seqalign_kernel_kind_name(int kind) {
  static const char *names[SEQALIGN_K_COUNT] = {
      "fill_wavefront", "fill_rowscan", "fill_stream", "fill_strips", "fill_wgstream", "fill_nw_dirs", "fill_nw_dirs_x2",
      "fill_sw_dirs", "fill_sw_dirs_x2", "fill_sw_best_x2", "sw_reduce", "sw_box", "sweep_regs", "sweep_lds", "sweep_strips",
      "sweep_dirs", "sweep_dirs_x2", "walk_lane", "walk_wave", "walk_dirs_lane", "walk_dirs_tile", "walk_moves_lane",
      "walk_moves_tile", "fill_nw_dirs_x4", "fill_sw_best_x4"};
  return kind >= 0 && kind < SEQALIGN_K_COUNT ? names[kind] : nullptr;
}

// ----------------------------------------------------------------- context ---
// explicit = seqalign_arenas_alloc (the caller asked for placed arenas and keeps them); otherwise the context's own scratch
// growing under a host-level call: the first time a context gets the full walk, later growths -- rare: the arenas grow by
// half each time -- look around in a quarter of the free memory and 16 GiB at most, so that a long-lived process does not
// repeat a 160 GiB walk whenever a batch is larger than the last
static SaPlacementOpts placement_opts(const seqalign_ctx *ctx, bool explicit_call) {
  SaPlacementOpts o;
  o.scan_bytes = (size_t)ctx->opt.arena_scan_gib << 30;
  o.quality_stop = ctx->opt.arena_quality;
  o.keep_bytes = (size_t)ctx->opt.arena_keep_gib << 30;
  o.free_fraction = 0.01f * (float)ctx->opt.arena_free_pct;
  if (!explicit_call) {
    o.free_fraction = 0.25f;
    if (ctx->arena_walks > 0) o.scan_bytes = std::min<size_t>(o.scan_bytes, (size_t)16 << 30);
  }
  return o;
}

// placed = false: the caller's kernels are bound by instruction issue, not by HBM (the multi-hit path's packed fill writes
// match_scores + direction bytes at under half the HBM peak: profiles/r04, C3 the same with and without the walk) -- plain
// allocations, no walk (0.1-0.3 s and up to a quarter of the free memory held for that long, then cleared by the driver).
// A set made that way is replaced by a placed one as soon as somebody asks for placed arenas.
int sa_host::reserve_arenas(seqalign_ctx *ctx, size_t bytes, bool placed) {
  if (ctx->arena_set && bytes <= ctx->M.cap && (ctx->arena_placed || !placed)) return SEQALIGN_OK;
  if (ctx->arena_set) {
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    for (hipStream_t t : ctx->copy_streams) if (t) (void)hipStreamSynchronize(t);
    sa_arenas_destroy(sa_arenas_take(ctx->M.p));
    ctx->arena_set = nullptr;
  }
  ctx->M = DevBuf(); ctx->A = DevBuf(); ctx->B = DevBuf();   // views of the set, never freed on their own
  const size_t want = bytes + (ctx->arena_walks ? bytes / 2 : bytes / 8) + 4096;
  SaArenaSet *set = nullptr;
  SaPlacementOpts po = placement_opts(ctx, false);
  if (!placed) po.scan_bytes = 0;
  hipError_t e = sa_arenas_create(ctx->device, want, ctx->stream, po, &set);
  if (e != hipSuccess) return fail_hip(e, "matrix arenas");
  if (placed) ctx->arena_walks++;
  ctx->arena_placed = placed;
  ctx->arena_set = set;
  void *const *a = sa_arenas_base(set);
  ctx->M.p = a[0]; ctx->A.p = a[1]; ctx->B.p = a[2];
  ctx->M.cap = ctx->A.cap = ctx->B.cap = sa_arenas_bytes(set);
  return SEQALIGN_OK;
}

extern "C" int seqalign_arenas_alloc(seqalign_ctx_t *ctx, uint64_t bytes_each, void *arenas[3], float *quality) {
  if (!ctx || !arenas || !bytes_each) return SEQALIGN_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  SaArenaSet *set = nullptr;
  hipError_t e = sa_arenas_create(ctx->device, (size_t)bytes_each, ctx->stream, placement_opts(ctx, true), &set);
  if (e != hipSuccess) return fail_hip(e, "matrix arenas");
  for (int k = 0; k < 3; ++k) arenas[k] = sa_arenas_base(set)[k];
  if (quality) *quality = sa_arenas_info(set)->quality;
  return SEQALIGN_OK;
}

extern "C" int seqalign_arenas_free(seqalign_ctx_t *ctx, void *arenas[3]) {
  if (!ctx || !arenas) return SEQALIGN_E_ARG;
  if (!arenas[0]) return SEQALIGN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  SaArenaSet *set = sa_arenas_take(arenas[0]);
  if (!set) { set_last_error("seqalign_arenas_free: not arenas of seqalign_arenas_alloc"); return SEQALIGN_E_ARG; }
  sa_arenas_destroy(set);
  arenas[0] = arenas[1] = arenas[2] = nullptr;
  return SEQALIGN_OK;
}

static_assert(sizeof(seqalign_arena_info_t) == sizeof(SaArenaInfo), "seqalign_arena_info_t mirrors SaArenaInfo");
extern "C" int seqalign_arenas_info(seqalign_ctx_t *ctx, void *const arenas[3], seqalign_arena_info_t *info) {
  if (!ctx || !arenas || !info) return SEQALIGN_E_ARG;
  SaArenaInfo i;
  if (!sa_arenas_copy_info(arenas[0], &i)) return SEQALIGN_E_ARG;
  memcpy(info, &i, sizeof(*info));
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

// ----------------------------------------------------------------- options ---
// key = the environment variable's name without SEQALIGN_, lower case.  Returns false for an unknown key or a
// value outside the key's range (nothing is changed then).
static bool parse_int(const char *val, long long *out) {   // the whole string is one integer, nothing else
  if (!val || !*val) return false;
  char *end = nullptr;
  errno = 0;
  const long long v = strtoll(val, &end, 10);
  if (errno || end == val || *end) return false;
  *out = v;
  return true;
}
static bool parse_bool(const char *val, bool *out) {       // 1 / 0, true / false, on / off, yes / no (any case)
  static const char *yes[] = {"1", "true", "on", "yes"}, *no[] = {"0", "false", "off", "no"};
  for (const char *y : yes) if (!strcasecmp(val, y)) { *out = true; return true; }
  for (const char *n : no) if (!strcasecmp(val, n)) { *out = false; return true; }
  return false;
}

static const char *const kKernelNames[] = {"auto", "wavefront", "rowscan", "stream", "strips", "wgstream"};
static const char *const kWalkerNames[] = {"auto", "lane", "wave"};
static const char *const kSweepModes[] = {"auto", "pair", "strips"};

static bool set_option(seqalign_ctx *ctx, const char *key, const char *val) {
  SaOptions &o = ctx->opt;
  auto is = [&](const char *k) { return !strcmp(key, k); };
  auto eq = [&](const char *v) { return !strcmp(val, v); };
  auto name_of = [&](const char *const *names, int n, int *out) { for (int k = 0; k < n; ++k) if (eq(names[k])) { *out = k; return true; } return false; };
  // a numeric key takes an integer and nothing else ("abc", "1x", "" are refused, not read as 0), inside its range
  auto number = [&](long long lo, long long hi, long long *out) { long long v; if (!parse_int(val, &v) || v < lo || v > hi) return false; *out = v; return true; };
  auto flag = [&](bool *out) { return parse_bool(val, out); };
  long long num = 0;
  int idx = 0;
  if (is("kernel")) { if (!name_of(kKernelNames, 6, &idx)) return false; o.kernel = idx; return true; }
  if (is("cpl")) { if (!number(0, 16, &num)) return false; o.cpl = (uint32_t)num; return true; }
  if (is("wpb")) { if (!number(0, 8, &num) || (num != 0 && num != 1 && num != 2 && num != 4 && num != 8)) return false; o.wpb = (uint32_t)num; return true; }
  if (is("lds_pad")) { if (!number(0, 160 * 1024, &num)) return false; o.lds_pad = (uint32_t)num; return true; }
  if (is("traceback")) { if (!eq("host") && !eq("device")) return false; o.traceback_host = eq("host"); return true; }
  if (is("trace_kernel")) { if (!name_of(kWalkerNames, 3, &idx)) return false; o.trace_kernel = (uint32_t)idx; return true; }
  if (is("sweep_mode")) { if (!name_of(kSweepModes, 3, &idx)) return false; o.sweep_mode = idx; return true; }
  if (is("sweep_strip")) { if (!number(0, 256, &num) || (num != 0 && num != 64 && num != 128 && num != 256)) return false; o.sweep_strip = (uint32_t)num; return true; }
  if (is("sweep_cpl")) { if (!number(0, 4, &num) || num == 3) return false; o.sweep_cpl = (uint32_t)num; return true; }
  if (is("sweep_ev")) return flag(&o.sweep_ev);
  if (is("sweep_trace")) return flag(&o.sweep_trace);
  if (is("sweep_dirs")) return flag(&o.sweep_dirs);
  if (is("nw_dirs")) return flag(&o.nw_dirs);
  if (is("pack16")) { if (!number(0, 2, &num)) return false; o.pack16 = (int)num; return true; }
  if (is("quad")) { if (!number(0, 2, &num)) return false; o.quad = (uint32_t)num; return true; }
  if (is("walk_overlap")) return flag(&o.walk_overlap);
  if (is("nw_moves")) return flag(&o.nw_moves);
  if (is("zero_copy")) { if (eq("auto")) { o.zero_copy = 4; return true; } if (!number(0, 3, &num)) return false; o.zero_copy = (uint32_t)num; return true; }
  if (is("reduce_depth")) { if (!number(0, 8, &num) || (num != 0 && num != 4 && num != 8)) return false; o.reduce_depth = (uint32_t)num; return true; }
  if (is("timing")) return flag(&o.timing);
  if (is("chunk_bytes")) {
    if (!number(0, (long long)1 << 60, &num) || (num != 0 && num < (1 << 20))) return false;
    o.chunk_bytes = (size_t)num;
    if (num) ctx->chunk_budget = (size_t)num; else ctx->chunk_budget = ctx->chunk_budget_default;
    return true;
  }
  if (is("subbatches")) { if (!number(0, 256, &num)) return false; o.subbatches = (uint32_t)num; return true; }
  if (is("upload_slices")) { if (!number(0, 16, &num)) return false; o.upload_slices = (uint32_t)num; return true; }
  if (is("arena_scan_gib")) { if (!number(0, 1024, &num)) return false; o.arena_scan_gib = (uint32_t)num; return true; }
  if (is("arena_keep_gib")) { if (!number(0, 1024, &num)) return false; o.arena_keep_gib = (uint32_t)num; return true; }
  if (is("async_lanes")) { if (!number(0, 8, &num)) return false; o.async_lanes = (uint32_t)num; return true; }
  if (is("dirs_local")) { if (!number(0, 1, &num)) return false; o.dirs_local = (uint32_t)num; return true; }
  if (is("walk_stage")) { if (!number(0, 1, &num)) return false; o.walk_stage = (uint32_t)num; return true; }
  if (is("walk_tile")) { if (!number(0, 64, &num) || !(num == 0 || num == 32 || num == 64)) return false; o.walk_tile = (uint32_t)num; return true; }
  if (is("walk_group")) { if (!number(0, 8, &num) || !(num == 0 || num == 1 || num == 4 || num == 8)) return false; o.walk_group = (uint32_t)num; return true; }
  if (is("arena_free_pct")) { if (!number(10, 90, &num)) return false; o.arena_free_pct = (uint32_t)num; return true; }
  if (is("arena_quality")) {
    char *end = nullptr;
    const double q = strtod(val, &end);
    if (end == val || *end || !(q > 0.5 && q < 1.5)) return false;
    o.arena_quality = (float)q;
    return true;
  }
  return false;
}

// the value in force, as set_option would take it
static bool get_option(const seqalign_ctx *ctx, const char *key, std::string *out) {
  const SaOptions &o = ctx->opt;
  auto is = [&](const char *k) { return !strcmp(key, k); };
  auto n = [&](long long v) { *out = std::to_string(v); return true; };
  if (is("kernel")) { *out = kKernelNames[o.kernel]; return true; }
  if (is("cpl")) return n(o.cpl);
  if (is("wpb")) return n(o.wpb);
  if (is("lds_pad")) return n(o.lds_pad);
  if (is("traceback")) { *out = o.traceback_host ? "host" : "device"; return true; }
  if (is("trace_kernel")) { *out = kWalkerNames[o.trace_kernel]; return true; }
  if (is("sweep_mode")) { *out = kSweepModes[o.sweep_mode]; return true; }
  if (is("sweep_strip")) return n(o.sweep_strip);
  if (is("sweep_cpl")) return n(o.sweep_cpl);
  if (is("sweep_ev")) return n(o.sweep_ev);
  if (is("sweep_trace")) return n(o.sweep_trace);
  if (is("sweep_dirs")) return n(o.sweep_dirs);
  if (is("nw_dirs")) return n(o.nw_dirs);
  if (is("pack16")) return n(o.pack16);
  if (is("quad")) return n(o.quad);
  if (is("walk_overlap")) return n(o.walk_overlap);
  if (is("nw_moves")) return n(o.nw_moves);
  if (is("zero_copy")) { if (o.zero_copy == 4) { *out = "auto"; return true; } return n(o.zero_copy); }
  if (is("reduce_depth")) return n(o.reduce_depth);
  if (is("timing")) return n(o.timing);
  if (is("chunk_bytes")) return n((long long)o.chunk_bytes);
  if (is("subbatches")) return n(o.subbatches);
  if (is("upload_slices")) return n(o.upload_slices);
  if (is("arena_scan_gib")) return n(o.arena_scan_gib);
  if (is("arena_keep_gib")) return n(o.arena_keep_gib);
  if (is("async_lanes")) return n(o.async_lanes);
  if (is("dirs_local")) return n(o.dirs_local);
  if (is("walk_stage")) return n(o.walk_stage);
  if (is("walk_tile")) return n(o.walk_tile);
  if (is("walk_group")) return n(o.walk_group);
  if (is("arena_free_pct")) return n(o.arena_free_pct);
  if (is("arena_quality")) { char buf[32]; snprintf(buf, sizeof(buf), "%.6g", (double)o.arena_quality); *out = buf; return true; }
  return false;
}

// the ONE place the library reads SEQALIGN_* tuning variables (SEQALIGN_DEVICE: sa_default_ctx_or_die;
// SEQALIGN_HOST_THREADS: the process-wide worker pool, sa_ctx.hpp)
static void options_from_env(seqalign_ctx *ctx) {
  static const char *keys[] = {"kernel", "cpl", "wpb", "lds_pad", "traceback", "trace_kernel", "sweep_mode", "sweep_strip",
                               "sweep_cpl", "sweep_ev", "sweep_trace", "sweep_dirs", "nw_dirs", "pack16", "quad", "walk_overlap", "nw_moves", "zero_copy", "reduce_depth", "timing", "chunk_bytes", "subbatches", "arena_scan_gib", "arena_quality", "arena_keep_gib", "upload_slices", "arena_free_pct", "async_lanes", "walk_group", "dirs_local", "walk_stage", "walk_tile"};
  for (const char *k : keys) {
    std::string name = "SEQALIGN_";
    for (const char *c = k; *c; ++c) name += (char)toupper((unsigned char)*c);
    if (const char *v = getenv(name.c_str())) {
      if (!set_option(ctx, k, v)) fprintf(stderr, "seqalign: ignoring %s=%s (not a value of option \"%s\")\n", name.c_str(), v, k);
    }
  }
}

extern "C" int seqalign_ctx_set_option(seqalign_ctx_t *ctx, const char *key, const char *value) {
  if (!ctx || !key || !value) return SEQALIGN_E_ARG;
  if (!set_option(ctx, key, value)) {
    set_last_error(std::string("seqalign_ctx_set_option: no option \"") + key + "\" with value \"" + value + "\"");
    return SEQALIGN_E_ARG;
  }
  return SEQALIGN_OK;
}

extern "C" int seqalign_ctx_get_option(const seqalign_ctx_t *ctx, const char *key, char *value, size_t cap) {
  if (!ctx || !key || !value) return SEQALIGN_E_ARG;
  std::string v;
  if (!get_option(ctx, key, &v) || v.size() + 1 > cap) return SEQALIGN_E_ARG;
  memcpy(value, v.c_str(), v.size() + 1);
  return SEQALIGN_OK;
}

extern "C" int seqalign_ctx_create(int device, seqalign_ctx_t **out) {
  if (!out) return SEQALIGN_E_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    set_last_error("hipGetDeviceCount: no device");
    return SEQALIGN_E_NO_DEVICE;
  }
  if (device < 0 || device >= n) return SEQALIGN_E_ARG;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_last_error(std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    return SEQALIGN_E_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  seqalign_ctx *ctx = new (std::nothrow) seqalign_ctx();
  if (!ctx) return SEQALIGN_E_NOMEM;
  ctx->device = device;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete ctx; return fail_hip(e, "hipStreamCreate"); }
  sa_pool_ref(device);   // the device's chunk pool lives as long as one context of it does
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  // one chunk of a host-level batch may use up to 40 % of what is free now
  // (288 GB HBM3E: ~100 GB per chunk on an empty MI355X), overridable
  ctx->chunk_budget = free_b ? (free_b / 10) * 4 : (size_t)8 << 30;
  // ...but not more than 48 GB: beyond that a chunk only adds allocation time (page
  // tables for tens of GB) and delays the first results; the option chunk_bytes overrides
  ctx->chunk_budget = std::min<size_t>(ctx->chunk_budget, (size_t)48 << 30);
  ctx->chunk_budget_default = ctx->chunk_budget;
  options_from_env(ctx);
  *out = ctx;
  return SEQALIGN_OK;
}

extern "C" void seqalign_ctx_destroy(seqalign_ctx_t *ctx) {
  if (!ctx) return;
  async_shutdown(ctx);   // submitted jobs are run to the end, their lanes' contexts destroyed
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < 2; ++k) if (ctx->cached[k]) seqalign_scoring_release(ctx, ctx->cached[k]);
  if (ctx->arena_set) sa_arenas_destroy(sa_arenas_take(ctx->M.p));
  for (DevBuf *b : {&ctx->arena, &ctx->off_a, &ctx->pair_list,
                    &ctx->status, &ctx->best_score, &ctx->best_index, &ctx->dirs,
                    &ctx->cand_count, &ctx->cand_off, &ctx->cand_cap, &ctx->cand_index, &ctx->cand_score,
                    &ctx->t_str_off, &ctx->t_out_a, &ctx->t_out_b, &ctx->t_meta})
    b->release();
  for (DevBuf &b : ctx->e) b.release();
  ctx->strip_progress.release();
  for (HostBuf *b : {&ctx->h_one, &ctx->h_desc, &ctx->h_arena, &ctx->h_M, &ctx->h_A, &ctx->h_B, &ctx->h_misc, &ctx->h_ta,
                     &ctx->h_tb, &ctx->h_tmeta})
    b->release();
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  for (hipStream_t t : ctx->copy_streams) if (t) (void)hipStreamDestroy(t);
  (void)hipStreamDestroy(ctx->stream);
  sa_pool_unref(ctx->device);
  delete ctx;
}

extern "C" int seqalign_pool_trim(seqalign_ctx_t *ctx, uint64_t keep_bytes, uint64_t *held_bytes) {
  if (!ctx) return SEQALIGN_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  if (keep_bytes != UINT64_MAX) sa_pool_trim(ctx->device, (size_t)keep_bytes);
  if (held_bytes) *held_bytes = sa_pool_bytes(ctx->device);
  return SEQALIGN_OK;
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
  for (size_t k = 0; k < (size_t)h->flat.n_classes * h->flat.n_classes; ++k) {
    const int32_t v = h->flat.table[k];
    if (v == SA_S_BLOCKED || v == SA_S_UNKNOWN) continue;   // (such scorings never reach the packed fills)
    h->table_abs_max = std::max(h->table_abs_max, v < 0 ? (v == INT32_MIN ? INT32_MAX : -v) : v);
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
  if (ctx) for (int k = 0; k < 2; ++k) if (ctx->cached[k] == h) ctx->cached[k] = nullptr;
  delete h;
}


// --------------------------------------------------------------- hot path ---
static SaFillParams make_params(const seqalign_ctx *ctx, const seqalign_dev_scoring_t *s, const seqalign_dev_batch_t *b) {
  SaFillParams p;
  p.tune_cpl = ctx->opt.cpl; p.tune_wpb = ctx->opt.wpb; p.tune_lds_pad = ctx->opt.lds_pad; p.tune_quad = ctx->opt.quad;
  p.arena = b->arena; p.off_a = b->off_a; p.len_a = b->len_a; p.off_b = b->off_b; p.len_b = b->len_b;
  p.mat_off = b->mat_off; p.M = b->match_scores; p.A = b->gap_a_scores; p.B = b->gap_b_scores;
  p.status = b->status; p.code = s->d_code; p.table = s->d_table;
  p.n_pairs = (uint32_t)b->n_pairs; p.K = s->flat.n_classes;
  p.gap_open = s->flat.gap_open; p.open1 = s->flat.open1; p.ext = s->flat.ext; p.floor = s->flat.floor;
  p.gen_eq = s->flat.gen_eq; p.gen_ne = s->flat.gen_ne; p.flags = s->flat.flags;
  p.best_score = nullptr; p.best_index = nullptr;
  p.cand_min = nullptr; p.cand_count = nullptr; p.cand_box = nullptr; p.cand_rows = nullptr; p.cand_rows_off = nullptr;
  p.uniform_stride = 0;
  p.pair_list = nullptr;
  p.table_abs_max = s->table_abs_max;
  return p;
}

static int pick_kernel(const seqalign_ctx *ctx, int kernel) {
  if (kernel != SEQALIGN_KERNEL_AUTO) return kernel;
  if (ctx->opt.kernel != SEQALIGN_KERNEL_AUTO) return ctx->opt.kernel;   // option "kernel": what AUTO means on this context
  return SEQALIGN_KERNEL_STREAM;   // measured fastest (profiles/); falls back when not applicable
}

// best_score / best_index (optional, SW): filled by the fill itself when the stream kernel runs
// (*best_done = true); otherwise the caller runs the separate reduction
int sa_host::fill_device(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, const seqalign_dev_batch_t *batch,
                         int kernel, void *stream, int32_t *best_score, uint64_t *best_index, bool *best_done,
                         const SaCandBox *cand, bool *cand_done) {
  if (best_done) *best_done = false;
  if (cand_done) *cand_done = false;
  if (!ctx || !scoring || !batch) return SEQALIGN_E_ARG;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  if (batch->n_pairs > 0xFFFFFFFFull) return SEQALIGN_E_ARG;
  if ((uint64_t)(batch->max_len_a + 1ull) * (batch->max_len_b + 1ull) >= (1ull << 31)) return SEQALIGN_E_TOO_LARGE;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SaFillParams p = make_params(ctx, scoring, batch);
  hipError_t e;
  (void)hipGetLastError();   // the launchers report hipGetLastError(): start from a clean slate
  int which = pick_kernel(ctx, kernel);
  const bool stream_ok = sa_stream_kernel_applicable(p, batch->max_len_a);
  if (kernel == SEQALIGN_KERNEL_AUTO && ctx->opt.kernel == SEQALIGN_KERNEL_AUTO) {
    // One wave per pair needs pairs to fill the chip.  Measured (seq-align_amd/tools/long_pairs.py,
    // profiles/r01_long_pairs.txt): 64 x 1000x1000 -- strips 0.49 ms, rowscan 0.86, stream 2.09;
    // 256 x 1000x1000 -- rowscan 0.99, strips 1.23, stream 2.09; 1000 x 1000x1000 -- stream 3.2,
    // rowscan 4.7; 16 x 5000x5000 -- strips 2.9, rowscan 21; 512 x 2000x2000 -- rowscan 5.2, strips 10.5.
    // 2 000 x 2 000^2 -- wgstream 15.1, rowscan 39.5, strips 43.1; 1 000 x 1 500^2 -- wgstream 5.8, rowscan 10.3;
    // 500 x 4 000^2 -- rowscan 20.4, wgstream 22.3, strips 45.0; 64 x 2 000^2 -- strips 1.5, wgstream 3.3.
    // 200 x 2 000^2 -- wgstream 3.3, strips 4.2; 4 000 x 1 000^2 -- wgstream 7.5, stream (16 columns per lane) 13.3;
    // 2 000 x 800^2 -- wgstream 2.6, stream 5.8; 20 000 x 600^2 -- stream (12 per lane) 13.1, wgstream 15.1.
    // 500 x 4 000^2 (8 waves, one-row ring) -- wgstream 18.3, rowscan 20.2; 100 x 4 000^2 -- strips 9.4, wgstream 12.4.
    // 513..768 columns (stream: 12 per lane): 4 096 x 600^2 -- wgstream 2.89, stream 3.04; 20 000 x 600^2 -- 13.1 vs 15.1.
    const bool wg_ok = sa_wgstream_kernel_applicable(p, batch->max_len_a) &&
                       (batch->max_len_a + 1 > 768 || batch->n_pairs < 8192);
    const uint32_t few = !wg_ok ? 256u : (batch->max_len_a + 1 <= 2048 ? 128u : 256u);
    if (batch->max_len_a > 512 && batch->n_pairs < few) which = SEQALIGN_KERNEL_STRIPS;
    else if (wg_ok) which = SEQALIGN_KERNEL_WGSTREAM;
    else if (!stream_ok || (batch->max_len_a > 767 && batch->n_pairs < 768)) which = SEQALIGN_KERNEL_ROWSCAN;
  }
  if (which == SEQALIGN_KERNEL_STREAM && !stream_ok) which = SEQALIGN_KERNEL_ROWSCAN;
  if (which == SEQALIGN_KERNEL_WGSTREAM && !sa_wgstream_kernel_applicable(p, batch->max_len_a))
    which = SEQALIGN_KERNEL_ROWSCAN;
  if ((which == SEQALIGN_KERNEL_STREAM || which == SEQALIGN_KERNEL_WGSTREAM) && best_score && best_index) {
    p.best_score = best_score; p.best_index = best_index;
    const bool reports = which == SEQALIGN_KERNEL_STREAM
                             ? sa_stream_kernel_reports_best(p, batch->max_len_a, batch->max_len_b)
                             : sa_wgstream_kernel_reports_best(p, batch->max_len_a, batch->max_len_b);
    if (reports) { if (best_done) *best_done = true; }
    else p.best_score = nullptr, p.best_index = nullptr;
  }
  if (cand && cand->dirs_used) *cand->dirs_used = false;
  if (cand && cand->best_only) {
    // the SW best-hit path's own fill: direction bytes + the best cell, nothing else (fill_sw_best_x2_kernel); the caller
    // asked sw_best_x2_applicable first and has no matrices to fall back to
    p.best_score = best_score; p.best_index = best_index;
    p.uniform_stride = cand->uniform_stride;
    p.dirs_local = cand->dirs_local ? 1u : 0u;
    if (cand->pair_list) { p.pair_list = cand->pair_list; p.n_pairs = cand->list_count; }
    if (!cand->dirs || !cand->dirs_used || !sa_sw_best_x2_applicable(p, batch->max_len_a, batch->max_len_b, cand->dirs)) {
      set_last_error("internal error: the best-hit direction fill was asked for a batch outside its domain");
      return SEQALIGN_E_ARG;
    }
    e = sa_launch_fill_sw_best_x2(p, batch->max_len_a, cand->dirs, st);
    if (e != hipSuccess) return fail_hip(e, "fill kernel launch");
    *cand->dirs_used = true;
    if (best_done) *best_done = true;
    return SEQALIGN_OK;
  }
  if (cand && cand->dirs && cand->dirs_used && ctx->opt.sweep_dirs && kernel == SEQALIGN_KERNEL_AUTO &&
      ctx->opt.kernel == SEQALIGN_KERNEL_AUTO) {
    // the SW multi-hit path's own fill: match_scores + a byte of directions per cell (sa_fill_dirs.hip)
    p.cand_min = cand->cand_min; p.cand_count = cand->cand_count; p.cand_box = cand->cand_box; p.cand_rows = cand->cand_rows;
    p.cand_rows_off = cand->hit_off;
    if (sa_dirs_fill_applicable(p, batch->max_len_a, cand->dirs)) {
      p.uniform_stride = ctx->opt.pack16 ? cand->uniform_stride : 0;
      const bool packed = sa_dirs_x2_applicable(p, batch->max_len_a, batch->max_len_b, cand->dirs);
      if (packed && cand->pair_list) { p.pair_list = cand->pair_list; p.n_pairs = cand->list_count; }
      e = packed ? sa_launch_fill_dirs_x2(p, batch->max_len_a, cand->dirs, st)
                 : sa_launch_fill_dirs(p, batch->max_len_a, cand->dirs, st);
      if (e != hipSuccess) return fail_hip(e, "fill kernel launch");
      *cand->dirs_used = true;
      if (cand_done) *cand_done = true;
      return SEQALIGN_OK;
    }
    p.cand_min = nullptr; p.cand_count = nullptr; p.cand_box = nullptr; p.cand_rows = nullptr; p.cand_rows_off = nullptr;
  }
  if ((which == SEQALIGN_KERNEL_STREAM || which == SEQALIGN_KERNEL_WGSTREAM) && cand) {   // where the candidates are, straight from the fill's registers
    p.cand_min = cand->cand_min; p.cand_count = cand->cand_count; p.cand_box = cand->cand_box; p.cand_rows = cand->cand_rows;
    p.cand_rows_off = cand->hit_off;
    const bool reports = which == SEQALIGN_KERNEL_STREAM ? sa_stream_kernel_emits_candidates(p, batch->max_len_a)
                                                          : sa_wgstream_kernel_emits_candidates(p, batch->max_len_a);
    if (reports) { if (cand_done) *cand_done = true; }
    else p.cand_count = nullptr;
  }
  if (which >= SEQALIGN_KERNEL_WAVEFRONT && which <= SEQALIGN_KERNEL_WGSTREAM)
    sa_record_launch(SEQALIGN_K_FILL_WAVEFRONT + (which - SEQALIGN_KERNEL_WAVEFRONT), batch->n_pairs);
  switch (which) {
    case SEQALIGN_KERNEL_WAVEFRONT: e = sa_launch_fill_wavefront(p, batch->max_len_a, st); break;
    case SEQALIGN_KERNEL_STREAM: e = sa_launch_fill_stream(p, batch->max_len_a, st); break;
    case SEQALIGN_KERNEL_ROWSCAN:
      e = sa_launch_fill_rowscan(p, batch->max_len_a, st);
      break;
    case SEQALIGN_KERNEL_WGSTREAM: e = sa_launch_fill_wgstream(p, batch->max_len_a, st); break;
    case SEQALIGN_KERNEL_STRIPS: {
      const uint64_t words = ((batch->n_pairs + 7) / 8 * 8) * (uint64_t)sa_fill_strips_per_pair(batch->max_len_a) + 1;
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

// seqalign_nw_batch's own fill (sa_fill_dirs.hip): directions only + the end cell's score / state per pair.  *used = false
// (and nothing launched) when the scoring or the batch is outside that kernel's domain, or the option nw_dirs is off.
int sa_host::nw_dirs_fill(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, const seqalign_dev_batch_t *batch,
                          uint8_t *dirs, int32_t *end_score, uint64_t *end_state, void *stream, bool *used,
                          uint64_t uniform_stride, const uint32_t *pair_list, uint32_t list_count, bool local) {
  *used = false;
  if (!ctx->opt.nw_dirs || ctx->opt.kernel != SEQALIGN_KERNEL_AUTO || batch->n_pairs == 0 || batch->n_pairs > 0xFFFFFFFFull) return SEQALIGN_OK;
  SaFillParams p = make_params(ctx, scoring, batch);
  p.best_score = end_score; p.best_index = end_state;
  if (!sa_nw_dirs_fill_applicable(p, batch->max_len_a, dirs)) return SEQALIGN_OK;
  (void)hipGetLastError();
  if (pair_list) {   // pairs pair_list[0 .. list_count) of the batch's arrays (batch->max_len_a / _b: of THOSE pairs)
    if (list_count == 0) { *used = true; return SEQALIGN_OK; }
    p.pair_list = pair_list; p.n_pairs = list_count;
  }
  p.uniform_stride = ctx->opt.pack16 ? uniform_stride : 0;
  p.dirs_local = local ? 1u : 0u;
  hipError_t e = sa_nw_dirs_x2_applicable(p, batch->max_len_a, batch->max_len_b, dirs)
                     ? sa_launch_fill_nw_dirs_x2(p, batch->max_len_a, dirs, stream ? (hipStream_t)stream : ctx->stream)
                     : sa_launch_fill_nw_dirs(p, batch->max_len_a, dirs, stream ? (hipStream_t)stream : ctx->stream);
  if (e != hipSuccess) return fail_hip(e, "fill kernel launch");
  *used = true;
  return SEQALIGN_OK;
}
// the same for a chunk whose pairs are mostly of one shape (modal_a x modal_b): pairs list[0 .. n_modal) two per wave, pairs
// list[n_modal .. n_modal + n_rest) one per wave, in one grid (sa_fill_dirs_x2.hip: fill_nw_dirs_mixed_kernel).  `batch`: the whole
// chunk's arrays (the list holds indices into them); every pair's bytes start on a multiple of 256.
int sa_host::nw_dirs_fill_mixed(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, const seqalign_dev_batch_t *batch,
                                uint8_t *dirs, int32_t *end_score, uint64_t *end_state, void *stream, const uint32_t *list,
                                uint32_t n_modal, uint32_t n_rest, uint32_t modal_a, uint32_t modal_b, bool local) {
  if (n_modal + n_rest == 0) return SEQALIGN_OK;
  SaFillParams p = make_params(ctx, scoring, batch);
  p.best_score = end_score; p.best_index = end_state;
  p.pair_list = list; p.uniform_stride = 256; p.dirs_local = local ? 1u : 0u;
  if (!list || !sa_nw_dirs_fill_applicable(p, batch->max_len_a, dirs) || !sa_nw_dirs_x2_applicable(p, modal_a, modal_b, dirs)) {
    set_last_error("seqalign_nw_batch: internal error: the mixed directions-only fill was asked for a chunk outside its domain");
    return SEQALIGN_E_ARG;
  }
  (void)hipGetLastError();
  hipError_t e = sa_launch_fill_nw_dirs_mixed(p, batch->max_len_a, dirs, n_modal, n_rest, stream ? (hipStream_t)stream : ctx->stream);
  if (e != hipSuccess) return fail_hip(e, "fill kernel launch");
  return SEQALIGN_OK;
}
// ---- whether a chunk may be laid out for one of the direction-byte fills: decided BEFORE any buffer is reserved, from the
// flattened scoring and the shape alone (sa_kernels.h: the kernels' domains) and the context's options
static SaScoringTraits traits_of(const seqalign_dev_scoring_t *s) {
  return SaScoringTraits{s->flat.flags, s->flat.n_classes, s->flat.gap_open, s->flat.open1, s->flat.ext, s->flat.gen_eq, s->flat.gen_ne,
                         s->table_abs_max};
}
bool sa_host::nw_dirs_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t max_len_a, uint64_t n_pairs) {
  // One wave walks a pair's rows one after the other; with 16 columns per lane (rows of 769 .. 1 024 columns) a row is long enough
  // that a few pairs are done sooner by fills that put several waves on each (tools/nw_wide_few.py, 1 000 x 1 000: 8 pairs 1.29 ms
  // on three matrices / 1.88 on direction bytes, 128: 1.82 / 1.94, 256: 1.95 / 1.98, 512: 2.25 / 2.03; 700 x 700: ahead at every size)
  if (max_len_a + 1 > 12 * 64 && n_pairs < 384) return false;
  return ctx->opt.nw_dirs && ctx->opt.kernel == SEQALIGN_KERNEL_AUTO && sa_domain_nw_dirs(traits_of(scoring), max_len_a);
}
bool sa_host::nw_dirs_x2_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t len_a, uint32_t len_b) {
  return ctx->opt.pack16 && nw_dirs_applicable(ctx, scoring, len_a) && sa_domain_nw_dirs_x2(traits_of(scoring), len_a, len_b);
}
// ... the SW best-hit path's fill (directions + the best cell)
bool sa_host::sw_best_x2_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t len_a, uint32_t len_b) {
  return ctx->opt.pack16 && ctx->opt.sweep_dirs && ctx->opt.kernel == SEQALIGN_KERNEL_AUTO && !ctx->opt.traceback_host &&
         sa_domain_sw_best_x2(traits_of(scoring), len_a, len_b);
}
// ... and the SW multi-hit path's (match_scores + directions)
bool sa_host::sw_dirs_x2_applicable(const seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring, uint32_t len_a, uint32_t len_b) {
  return ctx->opt.pack16 && ctx->opt.sweep_dirs && ctx->opt.kernel == SEQALIGN_KERNEL_AUTO && sa_domain_sw_dirs_x2(traits_of(scoring), len_a, len_b);
}

// SW walks from start_index on direction bytes (the best-hit path): seqalign_sw_traceback_device's launch with the bytes
// and the start scores in place of the matrices
int sa_host::sw_traceback_dirs(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *sc, const seqalign_dev_batch_t *b,
                               const seqalign_trace_t *t, const uint8_t *dirs, const int32_t *start_score, void *stream) {
  if (!ctx || !sc || !b || !t || !dirs || !start_score || !t->start_index || !t->out_pos) return SEQALIGN_E_ARG;
  if (b->n_pairs == 0) return SEQALIGN_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SaTraceParams p;
  memset(&p, 0, sizeof(p));
  p.arena = b->arena; p.off_a = b->off_a; p.len_a = b->len_a; p.off_b = b->off_b; p.len_b = b->len_b;
  p.mat_off = b->mat_off; p.dirs = dirs; p.start_score = start_score; p.fill_status = nullptr;
  p.dirs_blocked = sa_dirs_blocked_shape(b->max_len_a);   // (the best-hit fill's layout for this chunk: sa_kernels.h)
  p.code = sc->d_code; p.table = sc->d_table; p.str_off = t->str_off; p.out_a = t->out_a; p.out_b = t->out_b;
  p.out_head = t->out_head; p.out_len = t->out_len; p.out_score = t->out_score; p.trace_status = t->status;
  p.start_index = t->start_index; p.out_pos = t->out_pos;
  p.n_pairs = (uint32_t)b->n_pairs; p.K = sc->flat.n_classes; p.open1 = sc->flat.open1; p.ext = sc->flat.ext;
  p.gen_eq = sc->flat.gen_eq; p.gen_ne = sc->flat.gen_ne; p.flags = sc->flat.flags;
  p.tune_walker = ctx->opt.trace_kernel; p.tune_group = ctx->opt.walk_group;
  hipError_t e = sa_launch_nw_traceback(p, st);
  if (e != hipSuccess) return fail_hip(e, "traceback launch");
  return SEQALIGN_OK;
}


extern "C" int seqalign_fill_batch_device(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring,
                                          const seqalign_dev_batch_t *batch, int kernel, void *stream) {
  CallScope scope(ctx);
  return fill_device(ctx, scoring, batch, kernel, stream, nullptr, nullptr, nullptr);
}

extern "C" void *seqalign_ctx_stream(seqalign_ctx_t *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int seqalign_time_fill_ms(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *scoring,
                                     const seqalign_dev_batch_t *batch, int kernel, void *stream,
                                     int repeats, float *ms_each) {
  if (!ctx || repeats <= 0 || !ms_each) return SEQALIGN_E_ARG;
  CallScope scope(ctx);
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  EventList events;   // destroyed on every exit path
  for (int r = 0; r < 2 * repeats; ++r) HIP_TRY(events.add());
  const std::vector<hipEvent_t> &ev = events.ev;
  StreamSyncOnExit sync(st);
  int rc = SEQALIGN_OK;
  for (int r = 0; r < repeats && rc == SEQALIGN_OK; ++r) {
    HIP_TRY(hipEventRecord(ev[2 * r], st));
    rc = seqalign_fill_batch_device(ctx, scoring, batch, kernel, st);
    HIP_TRY(hipEventRecord(ev[2 * r + 1], st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  for (int r = 0; r < repeats && rc == SEQALIGN_OK; ++r) HIP_TRY(hipEventElapsedTime(&ms_each[r], ev[2 * r], ev[2 * r + 1]));
  return rc;
}

static int launch_traceback(seqalign_ctx_t *ctx, const seqalign_dev_scoring_t *sc, const seqalign_dev_batch_t *b,
                            const seqalign_trace_t *t, void *stream, bool sw) {
  if (!ctx || !sc || !b || !t) return SEQALIGN_E_ARG;
  CallScope scope(ctx);
  if (sw && (!t->start_index || !t->out_pos)) return SEQALIGN_E_ARG;
  if (b->n_pairs == 0) return SEQALIGN_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SaTraceParams p;
  memset(&p, 0, sizeof(p));
  p.arena = b->arena; p.off_a = b->off_a; p.len_a = b->len_a; p.off_b = b->off_b; p.len_b = b->len_b;
  p.mat_off = b->mat_off; p.M = b->match_scores; p.A = b->gap_a_scores; p.B = b->gap_b_scores;
  p.code = sc->d_code; p.table = sc->d_table; p.str_off = t->str_off; p.out_a = t->out_a; p.out_b = t->out_b;
  p.out_head = t->out_head; p.out_len = t->out_len; p.out_score = t->out_score; p.trace_status = t->status;
  p.start_index = sw ? t->start_index : nullptr; p.out_pos = sw ? t->out_pos : nullptr;
  p.n_pairs = (uint32_t)b->n_pairs; p.K = sc->flat.n_classes; p.open1 = sc->flat.open1; p.ext = sc->flat.ext;
  p.gen_eq = sc->flat.gen_eq; p.gen_ne = sc->flat.gen_ne; p.flags = sc->flat.flags;
  p.tune_walker = ctx->opt.trace_kernel; p.tune_group = ctx->opt.walk_group;
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
  CallScope scope(ctx);
  if (r->n_pairs == 0) return SEQALIGN_OK;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  SaReduceParams p;
  p.len_a = r->len_a; p.len_b = r->len_b; p.mat_off = r->mat_off; p.M = r->match_scores;
  p.min_score = r->min_score; p.best_score = r->best_score; p.best_index = r->best_index;
  p.cand_count = r->cand_count; p.cand_off = r->cand_off; p.cand_cap = r->cand_cap;
  p.cand_index = r->cand_index; p.cand_score = r->cand_score;
  p.n_pairs = (uint32_t)r->n_pairs;
  p.slices = 0;
  p.tune_depth = ctx->opt.reduce_depth;
  hipError_t e = sa_launch_sw_reduce(p, st);
  if (e != hipSuccess) return fail_hip(e, "sw reduce launch");
  return SEQALIGN_OK;
}

// ------------------------------------------- legacy single-pair entry point ---
// The reference's aligner_align (src/alignment.c:170-193) touches nothing but its own aligner_t, so one aligner per
// thread runs in parallel (SURVEY 8b "Threading").  aligner_t is a public 72-byte struct with no room for a handle,
// so the device side of that promise hangs off the calling THREAD instead: every thread that uses the legacy API gets
// its own context -- stream, scratch, cached scoring -- on first use, and keeps it until it exits.  No lock is shared
// between callers (rounds 1-2: one context behind one mutex; 8 threads ran at the speed of one).
namespace {
struct ThreadContext {
  seqalign_ctx *ctx = nullptr;
  bool main_thread = false;
  ~ThreadContext() {
    // a worker thread's context goes with the thread; the main thread's is left to process exit (the HIP runtime
    // may already be shutting down when the main thread's thread_locals are destroyed)
    if (ctx && !main_thread) seqalign_ctx_destroy(ctx);
  }
};
thread_local ThreadContext tl_legacy;
}  // namespace

extern "C" seqalign_ctx_t *sa_default_ctx_or_die(void) {
  if (!tl_legacy.ctx) {
    int dev = 0;
    if (const char *env = getenv("SEQALIGN_DEVICE")) dev = atoi(env);
    int rc = seqalign_ctx_create(dev, &tl_legacy.ctx);
    if (rc != SEQALIGN_OK) {
      fprintf(stderr, "seqalign: cannot open GPU %d: %s (%s)\n"
                      "seqalign: this library has no CPU path; an MI355X (gfx950) is required\n",
              dev, seqalign_strerror(rc), seqalign_last_error());
      exit(EXIT_FAILURE);
    }
    tl_legacy.main_thread = (getpid() == (pid_t)syscall(SYS_gettid));
  }
  return tl_legacy.ctx;
}

// FNV-1a over everything scoring_lookup can see (header fields, wildcard and swap
// bitsets, and the scores whose bit is set): the legacy per-pair API re-uses the
// uploaded scoring while the caller's scoring_t is unchanged.
static uint64_t scoring_fingerprint(const scoring_t *sc, int is_sw) {
  uint64_t h = 1469598103934665603ull;
  // (round 6: eight bytes per multiply -- byte-wise FNV over the 8 KiB of bit sets was 6 of the legacy call's 7 us before its
  // launch, on a 9 x 10 pair whose whole call takes 32, profiles/r06/r06_legacy_latency.txt; every input is a multiple of 4 bytes)
  auto mix = [&h](const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    for (; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
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

int sa_host::cached_scoring(seqalign_ctx *ctx, const scoring_t *sc, int is_sw, seqalign_dev_scoring **out) {
  const int k = is_sw ? 1 : 0;
  const uint64_t fp = scoring_fingerprint(sc, is_sw);
  if (!ctx->cached[k] || ctx->cached_fp[k] != fp) {
    if (ctx->cached[k]) {
      (void)hipStreamSynchronize(ctx->stream);   // nothing of an earlier call may still be reading the old tables
      seqalign_scoring_release(ctx, ctx->cached[k]);
    }
    ctx->cached[k] = nullptr;
    int rc = seqalign_scoring_upload(ctx, sc, is_sw, &ctx->cached[k]);
    if (rc) return rc;
    ctx->cached_fp[k] = fp;
  }
  *out = ctx->cached[k];
  return SEQALIGN_OK;
}

// ---- the legacy single-pair call's pinned block and the combiner of concurrent callers
namespace {
// OneBlock: [0, 4 KiB) the leader's descriptor arrays for up to kOneCombine requests | sequences | M | A | B
constexpr size_t kOneCombine = 64;
constexpr size_t kOneSeqAt = 4096, kOneSeqBytes = 60 * 1024;
constexpr size_t kOneMatAt = 64 * 1024, kOneMatrixBytes = 704 * 1024;   // 180 224 cells per matrix (150 x 150: 22 801)
constexpr size_t kOneBlockBytes = kOneMatAt + 3 * kOneMatrixBytes;

struct OneRequest {
  seqalign_ctx *ctx = nullptr;
  seqalign_dev_scoring *dsc = nullptr;
  uint64_t fp = 0;
  int is_sw = 0;
  uint32_t len_a = 0, len_b = 0;
  uint64_t block_dev = 0;   // device address of the caller's OneBlock
  uint64_t status = 0;
  int rc = SEQALIGN_OK;
  bool taken = false;                 // (under the combiner's mutex) part of a launch that is in flight
  std::atomic<bool> done{false};      // set by that launch's leader after its wait; the owner spins on it
};
// At most kOneRounds launches in flight: a caller that finds fewer leads one (with everything waiting for its scoring,
// after giving the callers of the last launches up to 20 us to join), the others wait to be taken by the next leader --
// alone, a thread's every call is its own launch as before; with more, the launches carry several pairs each and overlap
// three deep.  A launch takes ~50 us whatever it carries (one wave per pair, 120 rows one after the other).  Measured
// with examples/legacy_threads.c (pairs per second against one thread): 2 threads 1.9x, 4: 2.9x, 8: 4.8x, 16: 7.0x
// (one launch per call, rounds 3: 8 threads 3.3-4.0x; kOneRounds 1 / 2 / 3 / 4 / 8 without the gathering: 3.6 / 4.0 /
// 4.3 / 4.2 / 3.8x).
constexpr int kOneRounds = 3;
struct OneCombiner {
  std::mutex m;
  std::vector<OneRequest *> waiting;
  std::atomic<int> in_flight{0};
  std::atomic<int> n_waiting{0};
  std::atomic<int> recent_group{1};   // size of the last launches' groups (decaying maximum): how many callers to expect
};
OneCombiner g_one;

// One launch + one wait for a group of requests that share the device and the scoring (the leader's upload is used: same
// fingerprint = same tables).  Descriptors go into the LEADER's block; every address in them is absolute (arena base 0),
// and the three matrices of every request sit kOneMatrixBytes apart, so gap_a / gap_b are "match_scores + a constant".
int run_one_group(OneRequest *lead, const std::vector<OneRequest *> &grp) {
  seqalign_ctx *ctx = lead->ctx;
  char *h = ctx->h_one.as<char>(), *d = static_cast<char *>(ctx->one_dev);
  uint64_t *off_a = reinterpret_cast<uint64_t *>(h), *off_b = off_a + kOneCombine, *mat = off_b + kOneCombine;
  uint32_t *la = reinterpret_cast<uint32_t *>(mat + kOneCombine), *lb = la + kOneCombine;
  uint64_t *st = reinterpret_cast<uint64_t *>(lb + kOneCombine);
  seqalign_dev_batch_t db;
  db.n_pairs = grp.size(); db.max_len_a = 0; db.max_len_b = 0;
  for (size_t k = 0; k < grp.size(); ++k) {
    const OneRequest *q = grp[k];
    off_a[k] = q->block_dev + kOneSeqAt; off_b[k] = off_a[k] + q->len_a;
    mat[k] = (q->block_dev + kOneMatAt) / 4;
    la[k] = q->len_a; lb[k] = q->len_b; st[k] = 0;
    db.max_len_a = std::max(db.max_len_a, q->len_a); db.max_len_b = std::max(db.max_len_b, q->len_b);
  }
  db.arena = nullptr;
  db.off_a = reinterpret_cast<const uint64_t *>(d); db.off_b = db.off_a + kOneCombine; db.mat_off = db.off_b + kOneCombine;
  db.len_a = reinterpret_cast<const uint32_t *>(db.mat_off + kOneCombine); db.len_b = db.len_a + kOneCombine;
  db.status = reinterpret_cast<uint64_t *>(d + (reinterpret_cast<char *>(st) - h));
  db.match_scores = nullptr;
  db.gap_a_scores = reinterpret_cast<int32_t *>(kOneMatrixBytes);
  db.gap_b_scores = reinterpret_cast<int32_t *>(2 * kOneMatrixBytes);
  hipStream_t stream = ctx->stream;
  StageTimer tm(ctx->opt.timing);   // (option timing: where a single pair's microseconds go -- tools/legacy_latency.py)
  int rc = sa_host::fill_device(ctx, lead->dsc, &db, SEQALIGN_KERNEL_AUTO, stream, nullptr, nullptr, nullptr);
  tm.lap("one pair: launch enqueued");
  // kernel end + wait: the GPU's writes to the (coherent) pinned blocks are visible
  // (polled: a blocking wait adds the wake-up of a sleeping thread to every pair; a launch of this size takes ~50 us)
  hipError_t e = hipErrorNotReady;
  for (unsigned spins = 0; rc == SEQALIGN_OK && spins < 20000 && (e = hipStreamQuery(stream)) == hipErrorNotReady; ++spins) __builtin_ia32_pause();
  if (e == hipErrorNotReady || rc != SEQALIGN_OK) e = hipStreamSynchronize(stream);
  if (rc == SEQALIGN_OK && e != hipSuccess) rc = fail_hip(e, "hipStreamSynchronize");
  tm.lap("one pair: polled until done");
  for (size_t k = 0; k < grp.size(); ++k) { grp[k]->status = st[k]; grp[k]->rc = rc; }
  return rc;
}

int combine_and_run(OneRequest *req) {
  OneCombiner &g = g_one;
  { std::lock_guard<std::mutex> lk(g.m); g.waiting.push_back(req); g.n_waiting.fetch_add(1, std::memory_order_relaxed); }
  constexpr long long gather_ns = 20000;   // measured (examples/legacy_threads.c, 8 threads): 0 -> 4.3x, 10-40 us -> 4.6-4.8x one thread
  for (unsigned spins = 0; !req->done.load(std::memory_order_acquire); ++spins) {
    if (g.in_flight.load(std::memory_order_relaxed) < kOneRounds) {
      if (spins == 0) {
        // callers that were in the last launches are probably on their way: give them a moment to join this one
        const int expect = g.recent_group.load(std::memory_order_relaxed);
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (g.n_waiting.load(std::memory_order_relaxed) < expect && !req->done.load(std::memory_order_acquire)) {
          clock_gettime(CLOCK_MONOTONIC, &t1);
          if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > gather_ns) break;
          __builtin_ia32_pause();
        }
        if (req->done.load(std::memory_order_acquire)) break;
      }
      std::unique_lock<std::mutex> lk(g.m);
      if (!req->taken && g.in_flight.load(std::memory_order_relaxed) < kOneRounds) {
        // lead: take every waiting request of my device and scoring, mine first
        g.in_flight.fetch_add(1, std::memory_order_relaxed);
        std::vector<OneRequest *> grp{req}, rest;
        for (OneRequest *q : g.waiting) {
          if (q == req) continue;
          const bool mine = q->ctx->device == req->ctx->device && q->fp == req->fp && q->is_sw == req->is_sw;
          (mine && grp.size() < kOneCombine ? grp : rest).push_back(q);
        }
        for (OneRequest *q : grp) q->taken = true;
        g.waiting.swap(rest);
        g.n_waiting.fetch_sub((int)grp.size(), std::memory_order_relaxed);
        { const int prev = g.recent_group.load(std::memory_order_relaxed);
          g.recent_group.store(std::max<int>((int)grp.size(), prev - (prev > 1 ? 1 : 0)), std::memory_order_relaxed); }
        lk.unlock();
        run_one_group(req, grp);
        g.in_flight.fetch_sub(1, std::memory_order_release);
        for (OneRequest *q : grp)
          if (q != req) q->done.store(true, std::memory_order_release);   // (q may be gone the moment this is visible)
        return req->rc;
      }
    }
    // a launch that carries (or will carry) my request is on its way: ~50 us.  Spin, politely after a while.
    if (spins < 4000) __builtin_ia32_pause(); else sched_yield();
  }
  return req->rc;
}
}  // namespace

extern "C" int sa_fill_one_pair(seqalign_ctx_t *ctx, const scoring_t *sc, int is_sw, const char *a, size_t len_a,
                                const char *b, size_t len_b, int32_t *M, int32_t *A, int32_t *B, uint64_t *status) {
  if (len_a > 0xFFFFFFFEull || len_b > 0xFFFFFFFEull) return SEQALIGN_E_TOO_LARGE;
  CallScope scope(ctx);
  StageTimer tm(ctx->opt.timing);
  HIP_TRY(hipSetDevice(ctx->device));
  seqalign_dev_scoring *dsc = nullptr;
  { int rc = cached_scoring(ctx, sc, is_sw, &dsc); if (rc) return rc; }
  tm.lap("one pair: device + scoring fingerprint");
  const uint64_t cells = ((uint64_t)len_a + 1) * ((uint64_t)len_b + 1);
  if (cells >= (1ull << 31)) return SEQALIGN_E_TOO_LARGE;
  if (cells * 4 <= kOneMatrixBytes && len_a + len_b <= kOneSeqBytes) {
    // A small pair -- what this entry point is for.  Everything the pair needs lives in ONE block of pinned host memory
    // per calling thread that the GPU reads and writes in place over PCIe (fixed geometry, OneBlock below): a few hundred
    // bytes of sequence read, the matrices written as the fill's usual aligned 1 KiB blocks (a 150 x 150 pair's 270 KB
    // take ~7 us of a ~50 us kernel); no staging copies, so no copy engine latency either.  One pair then costs TWO
    // runtime calls, a launch and a wait -- and with several threads in here at once it is the runtime's own
    // serialisation of those calls that bounds the pairs per second (8 threads, one launch each: 3.3-4.0x one thread).
    // So concurrent callers COMBINE (combine_and_run): whoever finds no launch in flight becomes the leader, takes every
    // waiting request with its scoring, fills them all with one launch + one wait into their own blocks, and wakes them.
    int rc;
    if (ctx->h_one.cap < kOneBlockBytes) {
      if ((rc = ctx->h_one.reserve(kOneBlockBytes))) return rc;
      ctx->one_dev = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&ctx->one_dev, ctx->h_one.p, 0));
    }
    char *h = ctx->h_one.as<char>();
    if (len_a) memcpy(h + kOneSeqAt, a, len_a);
    if (len_b) memcpy(h + kOneSeqAt + len_a, b, len_b);
    OneRequest req;
    req.ctx = ctx; req.dsc = dsc; req.fp = ctx->cached_fp[is_sw ? 1 : 0]; req.is_sw = is_sw ? 1 : 0;
    req.len_a = (uint32_t)len_a; req.len_b = (uint32_t)len_b;
    req.block_dev = reinterpret_cast<uint64_t>(ctx->one_dev);
    tm.lap("one pair: sequences into the block");
    if ((rc = combine_and_run(&req))) return rc;
    tm.lap("one pair: combine + launch + wait");
    memcpy(M, h + kOneMatAt, cells * 4);
    memcpy(A, h + kOneMatAt + kOneMatrixBytes, cells * 4);
    memcpy(B, h + kOneMatAt + 2 * kOneMatrixBytes, cells * 4);
    tm.lap("one pair: matrices out of the block");
    if (status) *status = req.status;
    return req.status == ~0ull ? SEQALIGN_OK : SEQALIGN_E_UNKNOWN_PAIR;
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
  return fill_batch_uploaded(ctx, &batch, dsc, &mat_off, M, A, B, status);
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

