// sa_batch.hip -- host-level entry points over HOST batches: chunking, packing, the fill with the
// matrices copied back (seqalign_fill_batch) and global alignment (seqalign_nw_batch).
#include "sa_ctx.hpp"

using namespace sa_host;


// ------------------------------------------------- host-level: chunked fill ---

// split the batch into chunks whose matrices (12 B/cell, plus whatever else the caller keeps per cell) fit the budget
std::vector<Chunk> sa_host::plan_chunks(const seqalign_batch_t *b, size_t budget, size_t bytes_per_cell, const uint64_t *extra_bytes) {
  std::vector<Chunk> out;
  Chunk c;
  const uint64_t per_cell = std::max<size_t>(bytes_per_cell, 1);
  uint64_t used = 0;   // device bytes of the chunk being built: its cells + what each of its pairs needs besides (extra_bytes[p])
  for (uint64_t p = 0; p < b->n_pairs; ++p) {
    const uint64_t cells = (uint64_t)(b->len_a[p] + 1ull) * (b->len_b[p] + 1ull);
    const uint64_t need = cells * per_cell + (extra_bytes ? extra_bytes[p] : 0);
    if (c.count && used + need > budget) { out.push_back(c); c = Chunk(); c.first = p; used = 0; }
    used += need;
    c.count++; c.cells += cells; c.seq_bytes += (uint64_t)b->len_a[p] + b->len_b[p];
    c.max_a = std::max(c.max_a, b->len_a[p]); c.max_b = std::max(c.max_b, b->len_b[p]);
  }
  if (c.count) out.push_back(c);
  return out;
}

// The context's auxiliary streams: [0] uploads, [1] downloads, [2] walks beside fills.  What they are for is running
// BESIDE the kernels of ctx->stream, and that takes separate hardware queues: the runtime multiplexes all streams of one
// priority over four of them, assigned as streams are first used -- a process that has used another stream or two of its
// own before the first host-level call (bench.py: a torch stream for the fills) gets the walk stream on the fills' queue,
// and the pipeline runs fill, walk and download one after the other (rocprofv3 timeline, C5's share: 7.2 instead of 5.8
// ms).  Streams of another PRIORITY come from another pool of queues, so uploads are a low-priority stream and downloads
// and walks high-priority ones, whatever the application has created at the default priority.
int sa_host::ensure_copy_streams(seqalign_ctx *ctx, int count) {
  int least = 0, greatest = 0;
  bool have_range = false;
  for (int k = 0; k < count; ++k) {
    if (ctx->copy_streams[k]) continue;
    if (!have_range) { HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest)); have_range = true; }
    HIP_TRY(hipStreamCreateWithPriority(&ctx->copy_streams[k], hipStreamNonBlocking, k == 0 ? least : greatest));
  }
  return SEQALIGN_OK;
}

// Upload one chunk (sequences packed back to back, matrices packed in pair
// order) and run the fill.  On return the device buffers of ctx hold the
// results; the stream is NOT synchronised.
// best_done (optional): ask the fill for the SW best cell per pair (into ctx->best_score / best_index);
// *best_done tells whether the fill kernel delivered it.
int sa_host::run_chunk(seqalign_ctx *ctx, const seqalign_batch_t *b, const Chunk &c,
                       const seqalign_dev_scoring *sc, seqalign_dev_batch_t *dev_out, bool *best_done,
                       const SaCandBox *cand, bool *cand_done, uint64_t uniform_stride) {
  // uniform_stride != 0 (the caller checked that every pair of the chunk has the same shape): pair k's cells start at
  // k * uniform_stride instead of back to back (the packed fills' layout, sa_fill_dirs_x2.hip).
  // uniform_stride == kBucketShapes (the caller checked that the packed fill takes the chunk's largest shape): a RAGGED chunk
  // for the packed fills -- every pair's cells start on a multiple of 256, and each slice of the chunk comes with a list that
  // pairs up its pairs of equal shape ("bucket by shape"; a pair without a partner has a wave to itself)
  const bool bucket = uniform_stride == kBucketShapes;
  if (bucket) uniform_stride = 0;
  const uint64_t n = c.count;
  int rc;
  StageTimer tm(ctx->opt.timing);
  // pinned descriptor block: off_a, off_b, mat_off (u64) then len_a, len_b (u32)
  const size_t desc_bytes = n * (3 * sizeof(uint64_t) + 2 * sizeof(uint32_t)) + (bucket ? 2 * n * sizeof(uint32_t) : 0);
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
    h_mat[k] = uniform_stride ? k * uniform_stride : cell;
    // (the best-hit fill's direction bytes -- nobody's but the walkers' -- in blocks of 8 x 16 cells: sa_kernels.h)
    const uint64_t cells_k = (cand && cand->best_only && sa_dirs_blocked_shape(c.max_a)) ? sa_dirs_blocked_bytes(b->len_a[p], b->len_b[p])
                                                                                       : (uint64_t)(b->len_a[p] + 1ull) * (b->len_b[p] + 1ull);
    cell += bucket ? ((cells_k + 255u) & ~(uint64_t)255u) : cells_k;
  }
  const uint64_t mat_total = uniform_stride ? n * uniform_stride : cell;
  uint32_t *h_list = reinterpret_cast<uint32_t *>(h_len_b + n);   // (bucket) 2 n entries: the slices' pair lists
  if ((rc = ctx->arena.reserve(c.seq_bytes + 16))) return rc;
  // the five descriptor arrays travel as the one block they are on the host (off_a: the device copy)
  if ((rc = ctx->off_a.reserve(desc_bytes)) || (rc = ctx->status.reserve(n * 8))) return rc;
  const bool no_matrices = cand && cand->best_only;   // direction bytes + the best cell only (the caller's cand->dirs)
  // (a caller that laid the chunk out for the packed direction-byte fill has reserved unplaced arenas: sa_batch_sw.hip -- asking for
  // placed ones here would replace the set it already points into)
  const bool packed_dirs = cand && cand->dirs && (bucket || uniform_stride);
  if (!no_matrices && (rc = reserve_arenas(ctx, mat_total * 4, !packed_dirs))) return rc;
  hipStream_t st = ctx->stream;
  uint32_t list_len[2] = {0, 0};   // (bucket) entries of the first slice's list and of the second's
  if (bucket) {
    // the slices the loop below will make: [0, first) and [first, n)
    const bool split = false;   // (one launch: see n_sub below)
    const uint64_t first = split ? 2048 : n;
    static thread_local std::vector<uint32_t> tl_table;   // the one pair of each shape still waiting for a partner
    const uint64_t Wb = (uint64_t)c.max_b + 1, entries = ((uint64_t)c.max_a + 1) * Wb;
    if (tl_table.size() < entries) tl_table.resize(entries);
    for (int sl = 0; sl < 2; ++sl) {
      const uint64_t k0 = sl ? first : 0, k1 = sl ? n : first;
      uint32_t *out = h_list + 2 * k0;
      uint32_t at = 0;
      for (uint64_t k = k0; k < k1; ++k) tl_table[(uint64_t)h_len_a[k] * Wb + h_len_b[k]] = ~0u;
      for (uint64_t k = k0; k < k1; ++k) {
        uint32_t &slot = tl_table[(uint64_t)h_len_a[k] * Wb + h_len_b[k]];
        if (slot == ~0u) { slot = (uint32_t)k; continue; }
        out[at++] = slot; out[at++] = (uint32_t)k;
        slot = ~0u;
      }
      for (uint64_t k = k0; k < k1; ++k) {
        uint32_t &slot = tl_table[(uint64_t)h_len_a[k] * Wb + h_len_b[k]];
        if (slot == (uint32_t)k) { out[at++] = (uint32_t)k; out[at++] = (uint32_t)k; slot = ~0u; }   // alone in its wave
      }
      list_len[sl] = at;
    }
  }
  HIP_TRY(hipMemcpyAsync(ctx->off_a.p, h_off_a, desc_bytes, hipMemcpyHostToDevice, st));
  uint64_t *dv_off_a = ctx->off_a.as<uint64_t>(), *dv_off_b = dv_off_a + n, *dv_mat = dv_off_b + n;
  uint32_t *dv_len_a = reinterpret_cast<uint32_t *>(dv_mat + n), *dv_len_b = dv_len_a + n;
  const uint32_t *dv_list = dv_len_b + n;
  if (best_done && ((rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8)))) return rc;
  // the fill of pairs [k0, k1) of the chunk (the whole chunk: 0, n) with whatever the caller asked the fill to report
  auto range_desc = [&](uint64_t k0, uint64_t k1) {
    seqalign_dev_batch_t d;
    d.n_pairs = k1 - k0; d.arena = ctx->arena.as<uint8_t>();
    d.off_a = dv_off_a + k0; d.len_a = dv_len_a + k0;
    d.off_b = dv_off_b + k0; d.len_b = dv_len_b + k0;
    d.mat_off = dv_mat + k0;
    d.match_scores = ctx->M.as<int32_t>(); d.gap_a_scores = ctx->A.as<int32_t>(); d.gap_b_scores = ctx->B.as<int32_t>();
    if (no_matrices) d.match_scores = d.gap_a_scores = d.gap_b_scores = nullptr;
    d.status = ctx->status.as<uint64_t>() + k0; d.max_len_a = c.max_a; d.max_len_b = c.max_b;
    return d;
  };
  auto fill_range = [&](uint64_t k0, uint64_t k1, bool *bd, bool *cd, bool *du) -> int {
    if (bucket) {
      // the slice's pairs through the packed fill by its list (entries index the CHUNK's arrays: nothing moves with the slice)
      seqalign_dev_batch_t dc = range_desc(0, n);
      dc.n_pairs = k1 - k0;
      SaCandBox sub = *cand;
      bool used = false;
      sub.dirs_used = &used; sub.uniform_stride = 256;
      sub.pair_list = dv_list + 2 * k0; sub.list_count = list_len[k0 ? 1 : 0];
      const int r = no_matrices
          ? fill_device(ctx, sc, &dc, SEQALIGN_KERNEL_AUTO, st, ctx->best_score.as<int32_t>(), ctx->best_index.as<uint64_t>(), bd, &sub, nullptr)
          : fill_device(ctx, sc, &dc, SEQALIGN_KERNEL_AUTO, st, nullptr, nullptr, nullptr, &sub, cd);
      if (du) *du = used;
      if (!r && !used) { set_last_error("internal error: the packed fill refused a ragged chunk it had accepted"); return SEQALIGN_E_HIP; }
      return r;
    }
    const seqalign_dev_batch_t d = range_desc(k0, k1);
    if (best_done && no_matrices) {
      SaCandBox sub = *cand;
      bool used = false;
      sub.dirs_used = &used; sub.uniform_stride = uniform_stride;
      const int r = fill_device(ctx, sc, &d, SEQALIGN_KERNEL_AUTO, st, ctx->best_score.as<int32_t>() + k0,
                                ctx->best_index.as<uint64_t>() + k0, bd, &sub, nullptr);
      if (du) *du = used;
      return r;
    }
    if (best_done)
      return fill_device(ctx, sc, &d, SEQALIGN_KERNEL_AUTO, st, ctx->best_score.as<int32_t>() + k0, ctx->best_index.as<uint64_t>() + k0, bd);
    if (cand) {
      SaCandBox sub = *cand;   // per-pair arrays move with the range; hit_off holds absolute offsets into the scratch arena
      sub.uniform_stride = uniform_stride;
      sub.cand_count += k0; sub.cand_box += 4 * k0; sub.cand_min += k0; sub.hit_off += k0;
      bool used = false;
      sub.dirs_used = cand->dirs_used ? &used : nullptr;
      const int r = fill_device(ctx, sc, &d, SEQALIGN_KERNEL_AUTO, st, nullptr, nullptr, nullptr, &sub, cd);
      if (du) *du = used;
      return r;
    }
    return seqalign_fill_batch_device(ctx, sc, &d, SEQALIGN_KERNEL_AUTO, st);
  };
  // The sequences: packed by the thread pool 2048 pairs per task, uploaded in slices on the context's upload stream;
  // with many pairs the fill is launched slice by slice behind them (the first quarter of the pairs is being filled
  // while the rest is still being packed and shipped: C3's 11.5 MB of sequences cost 0.35 ms before the fill could
  // start), otherwise once, behind the last slice.
  constexpr uint64_t kPack = 2048;
  // (the packed two-pairs-per-wave fills take the chunk in ONE launch behind the whole upload: a first slice of 2 048 pairs to
  // start on while the rest is shipped made C3 4.31-4.34 ms where one launch makes 4.22-4.28 -- half as many waves per launch
  // leave the chip half empty for the length of a pair; option subbatches = 1: one launch for the other fills too)
  const uint64_t n_sub = (n >= 8192 && c.seq_bytes >= ((uint64_t)4 << 20) && ctx->opt.subbatches != 1 && !uniform_stride && !bucket) ? 4 : 1;
  const bool two_slices = (uniform_stride || bucket) && n_sub > 1;   // (see the loop below: one small slice to start on, then the rest)
  // (one launch: the upload still goes in slices, packing slice i + 1 beside the copy of slice i -- 2 MiB each, so that the copy
  // engine starts after ~20 us of packing and never waits for the host: C3's 11.5 MB 0.33 -> 0.28 ms)
  const uint64_t slice_bytes = n_sub > 1 ? (c.seq_bytes + n_sub - 1) / n_sub : ((uint64_t)2 << 20);
  { int rc_s = ensure_copy_streams(ctx, 1); if (rc_s) return rc_s; }
  hipStream_t su = ctx->copy_streams[0];
  StreamSyncOnExit sync_u(su);
  EventList ev;
  HIP_TRY(ev.add(hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ev.ev[0], st));          // the descriptors are on their way on st: uploads go after them
  HIP_TRY(hipStreamWaitEvent(su, ev.ev[0], 0));
  bool first = true, consistent = true, bd0 = false, cd0 = false, du0 = false;
  uint64_t filled_to = 0;
  for (uint64_t k0 = 0; k0 < n;) {
    uint64_t k1 = k0;
    if (two_slices) {
      // the packed fills run two pairs per wave: slices of a quarter of C3's 10 000 pairs are 1 000-2 000 waves on a chip
      // with 8 192 wave slots, and the slices run one after the other (C3: 1.06 + 1.44 ms where one launch takes 2.15).
      // One small slice to start on while the rest is packed and shipped, then everything else in one launch.
      k1 = k0 == 0 ? std::min(n, kPack) : n;
    } else {
      while (k1 < n && (k1 == k0 || h_off_a[k1] - h_off_a[k0] < slice_bytes)) k1 = std::min(n, k1 + kPack);
    }
    if (n - k1 < kPack) k1 = n;   // no crumb at the end
    // (tasks of 256 pairs: C4's 4 000 pairs in tasks of 2 048 kept two threads busy and fourteen idle)
    constexpr uint64_t kTask = 256;
    parallel_for((k1 - k0 + kTask - 1) / kTask, [&](uint64_t blk) {
      for (uint64_t k = k0 + blk * kTask, e = std::min(k1, k0 + (blk + 1) * kTask); k < e; ++k) {
        const uint64_t p = c.first + k;
        memcpy(h_seq + h_off_a[k], b->arena + b->off_a[p], b->len_a[p]);
        memcpy(h_seq + h_off_b[k], b->arena + b->off_b[p], b->len_b[p]);
      }
    });
    const uint64_t lo = h_off_a[k0], hi = k1 < n ? h_off_a[k1] : c.seq_bytes;
    if (hi > lo) HIP_TRY(hipMemcpyAsync(ctx->arena.as<uint8_t>() + lo, h_seq + lo, hi - lo, hipMemcpyHostToDevice, su));
    HIP_TRY(ev.add(hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev.ev.back(), su));
    HIP_TRY(hipStreamWaitEvent(st, ev.ev.back(), 0));
    if (n_sub > 1 && consistent) {
      bool bd = false, cd = false, du = false;
      if ((rc = fill_range(k0, k1, &bd, &cd, &du))) return rc;
      if (first) { bd0 = bd; cd0 = cd; du0 = du; first = false; }
      else if (bd != bd0 || cd != cd0 || du != du0) consistent = false;   // (cannot happen with >= 2 048 pairs per slice; see below)
      filled_to = k1;
    }
    k0 = k1;
  }
  tm.lap("run_chunk: pack + H2D");
  const seqalign_dev_batch_t d = range_desc(0, n);
  if (n_sub > 1 && consistent && filled_to == n) {
    if (best_done) *best_done = bd0;
    if (cand_done) *cand_done = cd0;
    if (cand && cand->dirs_used) *cand->dirs_used = du0;
    rc = SEQALIGN_OK;
  } else {
    // one launch over the whole chunk (also the way out if the slices' fills had chosen differently: a refill is
    // idempotent)
    bool bd = false, cd = false, du = false;
    rc = fill_range(0, n, &bd, &cd, &du);
    if (best_done) *best_done = bd;
    if (cand_done) *cand_done = cd;
    if (cand && cand->dirs_used) *cand->dirs_used = du;
  }
  if (rc) return rc;
  tm.lap("run_chunk: enqueue fill");
  if (dev_out) *dev_out = d;
  return SEQALIGN_OK;
}

// fetch the per-pair status words; returns UNKNOWN_PAIR if any pair flagged
int sa_host::fetch_status(seqalign_ctx *ctx, const Chunk &c, uint64_t *status_out) {
  int rc;
  if ((rc = ctx->h_misc.reserve(c.count * 8))) return rc;
  uint64_t *h = ctx->h_misc.as<uint64_t>();
  HIP_TRY(hipMemcpyAsync(h, ctx->status.p, c.count * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(stream_wait_spinning(ctx->stream));
  rc = SEQALIGN_OK;
  for (uint64_t k = 0; k < c.count; ++k) {
    if (status_out) status_out[c.first + k] = h[k];
    if (h[k] != ~0ull) rc = SEQALIGN_E_UNKNOWN_PAIR;
  }
  return rc;
}

int sa_host::check_batch(const seqalign_batch_t *b) {
  if (!b || (b->n_pairs && (!b->arena || !b->off_a || !b->off_b || !b->len_a || !b->len_b))) return SEQALIGN_E_ARG;
  for (uint64_t p = 0; p < b->n_pairs; ++p)
    if ((uint64_t)(b->len_a[p] + 1ull) * (b->len_b[p] + 1ull) >= (1ull << 31)) return SEQALIGN_E_TOO_LARGE;
  return SEQALIGN_OK;
}


// Device -> pageable host memory.  A plain hipMemcpy to pageable memory is staged
// by the runtime at ~12 GB/s; large copies go through our own two pinned buffers
// instead: the DMA of slice i+1 overlaps a multi-threaded memcpy of slice i into
// the caller's buffer.  The stream must be idle w.r.t. `src` producers (it is
// enqueued behind them) and is synchronised on return.
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
  EventList events;   // destroyed on every exit path
  HIP_TRY(events.add(hipEventDisableTiming));
  HIP_TRY(events.add(hipEventDisableTiming));
  const hipEvent_t *ev = events.ev.data();
  StreamSyncOnExit sync(ctx->stream);   // the pinned slices are read by host threads: never leave with a DMA in flight
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
  if (e != hipSuccess) return fail_hip(e, "pipelined D2H");
  return SEQALIGN_OK;
}

extern "C" int seqalign_fill_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                   int is_sw, const uint64_t *mat_off, int32_t *M, int32_t *A, int32_t *B,
                                   uint64_t *status) {
  if (!ctx || !scoring || !mat_off || !M || !A || !B) return SEQALIGN_E_ARG;
  CallScope scope(ctx);
  int rc = check_batch(batch);
  if (rc) return rc;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  seqalign_dev_scoring *sc = nullptr;
  if ((rc = cached_scoring(ctx, scoring, is_sw, &sc))) return rc;
  return fill_batch_uploaded(ctx, batch, sc, mat_off, M, A, B, status);
}

int sa_host::fill_batch_uploaded(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const seqalign_dev_scoring *sc,
                                 const uint64_t *mat_off, int32_t *M, int32_t *A, int32_t *B, uint64_t *status) {
  int rc = SEQALIGN_OK;
  int worst = SEQALIGN_OK;
  for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget)) {
    if ((rc = run_chunk(ctx, batch, c, sc, nullptr))) break;
    // Small chunks -- the legacy one-pair-per-call API above all -- in ONE round trip: the three matrices and the status
    // words into pinned staging behind each other, one synchronisation, then plain memcpys.  (One pair used to cost
    // four: three pageable copies, each waited for, and the status.)
    const size_t small = c.cells * 4;
    if (3 * small <= ((size_t)4 << 20)) {
      const size_t status_at = (3 * small + 7) & ~(size_t)7;   // (uint64 words: 8-byte aligned whatever the cell count)
      if ((rc = ctx->h_M.reserve(status_at + c.count * 8 + 64))) break;
      char *pin = ctx->h_M.as<char>();
      hipStream_t st = ctx->stream;
      StreamSyncOnExit sync(st);
      HIP_TRY(hipMemcpyAsync(pin, ctx->M.p, small, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(pin + small, ctx->A.p, small, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(pin + 2 * small, ctx->B.p, small, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(pin + status_at, ctx->status.p, c.count * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      uint64_t dev_cell = 0;
      const uint64_t *h_status = reinterpret_cast<const uint64_t *>(pin + status_at);
      for (uint64_t k = 0; k < c.count; ++k) {
        const uint64_t p = c.first + k, cells = (uint64_t)(batch->len_a[p] + 1ull) * (batch->len_b[p] + 1ull);
        memcpy(M + mat_off[p], pin + dev_cell * 4, cells * 4);
        memcpy(A + mat_off[p], pin + small + dev_cell * 4, cells * 4);
        memcpy(B + mat_off[p], pin + 2 * small + dev_cell * 4, cells * 4);
        dev_cell += cells;
        if (status) status[p] = h_status[k];
        if (h_status[k] != ~0ull) worst = SEQALIGN_E_UNKNOWN_PAIR;
      }
      continue;
    }
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

// ----------------------------------------------- host-level: NW over a batch ---

void sa_host::parallel_memcpy(void *dst, const void *src, size_t bytes) {
  const size_t kPiece = (size_t)2 << 20;
  const uint64_t pieces = (bytes + kPiece - 1) / kPiece;
  parallel_for(pieces, [&](uint64_t i) {
    const size_t off = i * kPiece;
    memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, std::min(kPiece, bytes - off));
  });
}


bool sa_host::traceback_on_host(const seqalign_ctx *ctx) { return ctx->opt.traceback_host; }

// ---- seqalign_nw_batch, device traceback: one chunk as a PIPELINE of sub-batches -------------------------------
// The stages of a chunk -- host packs the sequences, H2D, fill (HBM-bound), traceback, D2H of the strings, host
// unpacks -- run strictly one after the other cost their sum (C2: 1.3-1.4 ms for a 0.41 ms fill; C5's per-GPU share:
// 12.2 ms for 5.1).  What CAN overlap, measured (profiles/r03/r03_nw_pipeline.md):
//   * the copies and the host work with the kernels: yes -- they use PCIe and the CPU;
//   * the traceback of one sub-batch with the fill of the next (two streams, the walkers at high priority): NO.
//     The walkers are not latency-bound at these sizes, they are bound by scattered 64-byte sectors (3 per step: one
//     cell of each matrix; 10 k pairs: 0.29 ms = 2 TB/s of sector traffic, 125 k pairs: 3.6 ms = the same rate), so
//     next to a fill that saturates HBM both simply slow down (walks 0.29 -> 0.5 ms per sub-batch, fills +50 %),
//     and a walk over few pairs takes its ~0.3 ms however few they are -- 16 sub-batch walks cost more than one.
// So: sub-batches of consecutive pairs for upload + fill, tracebacks over GROUPS of sub-batches (>= 32 k pairs, or
// the whole chunk) in the SAME stream as the fills, downloads and unpacking per group:
//   stream U (upload)   :  H2D(0) H2D(1) H2D(2) ...
//   stream F (kernels)  :  fill(0) fill(1) .. walk(group 0) fill(..) .. walk(group 1) ...       fill(s) waits for H2D(s)
//   stream D (download) :                        strings + words of group 0 | group 1 ...       waits for walk(g)
//   host                :  packs every sub-batch, enqueues as it goes (the GPU starts on the first at once), then
//                          unpacks the groups in order as they land
// Same kernels, same buffers (a sub-batch is a pair range of the chunk's descriptor arrays and arenas), same results.
// Per group the device sends back ONE block of characters (out_a | out_b of its pairs) and one of per-pair words
// (head, len, score, status interleaved: SaTraceParams::out_meta4; the fill's status is folded in by the walker).
// The descriptor arrays of a chunk whose pairs all have ONE shape are arithmetic progressions: written on the device (a few
// microseconds) instead of sent over PCIe (44 B per pair: 5.5 MB for a 125 k-pair share, 0.1 ms in front of the first fill
// and 15 % on top of the sequences' bytes).  Layout of the block: off_a[n] | off_b[n] | mat_off[n] | slot[n + 1] | len_a[n] | len_b[n].
__global__ void __launch_bounds__(256) uniform_descriptors_kernel(uint64_t *desc, uint64_t n, uint32_t la, uint32_t lb, uint64_t stride) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n) return;
  uint64_t *off_a = desc, *off_b = off_a + n, *mat = off_b + n, *slot = mat + n;
  uint32_t *len_a = reinterpret_cast<uint32_t *>(slot + n + 1), *len_b = len_a + n;
  const uint64_t pos = k * ((uint64_t)la + lb);
  slot[k] = pos;
  if (k == n) return;
  off_a[k] = pos; off_b[k] = pos + la; mat[k] = k * stride; len_a[k] = la; len_b[k] = lb;
}

static uint32_t pick_subbatches(const seqalign_ctx *ctx, const Chunk &c) {
  if (ctx->opt.subbatches) return (uint32_t)std::min<uint64_t>(ctx->opt.subbatches, std::max<uint64_t>(c.count, 1));
  // by size, measured (profiles/r03/r03_nw_pipeline.md): C2 (10 k pairs) gains nothing from 2 or 4 sub-batches (1.28 ->
  // 1.34 ms: its walk must stay one launch, and two half fills are slower than one), C5's share (125 k pairs) 12.6 ->
  // 11.4 (4) -> 10.2 (8) -> 10.8 ms (16).  So: sub-batches of >= 12 288 pairs and >= 256 M cells (3 GB of matrices),
  // at most 8.
  const uint64_t by_pairs = c.count / 12288, by_cells = c.cells / (256ull << 20);
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(by_pairs, by_cells), 8));
}

static int nw_chunk_pipelined(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c, const seqalign_dev_scoring *sc,
                              uint32_t n_sub, const uint64_t *str_off, char *out_a, char *out_b, uint32_t *out_len,
                              int32_t *out_score) {
  const uint64_t n = c.count;
  int rc;
  StageTimer tm(ctx->opt.timing);
  // pinned descriptor block: off_a, off_b, mat_off, slot offset (u64 x n + 1) then len_a, len_b (u32)
  const size_t desc_bytes = (4 * n + 1) * sizeof(uint64_t) + 2 * n * sizeof(uint32_t);
  if ((rc = ctx->h_desc.reserve(desc_bytes)) || (rc = ctx->h_arena.reserve(c.seq_bytes + 16))) return rc;
  uint64_t *h_off_a = ctx->h_desc.as<uint64_t>(), *h_off_b = h_off_a + n, *h_mat = h_off_b + n, *h_slot = h_mat + n;
  uint32_t *h_len_a = reinterpret_cast<uint32_t *>(h_slot + n + 1), *h_len_b = h_len_a + n;
  uint8_t *h_seq = ctx->h_arena.as<uint8_t>();
  uint64_t pos = 0, cell = 0;
  bool same_shape = true;
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t p = c.first + k;
    h_slot[k] = pos;                       // a pair's string slot is len_a + len_b chars: the same prefix as the sequences'
    h_off_a[k] = pos; pos += batch->len_a[p];
    h_off_b[k] = pos; pos += batch->len_b[p];
    h_len_a[k] = batch->len_a[p]; h_len_b[k] = batch->len_b[p];
    h_mat[k] = cell; cell += (uint64_t)(batch->len_a[p] + 1ull) * (batch->len_b[p] + 1ull);
    same_shape = same_shape && h_len_a[k] == h_len_a[0] && h_len_b[k] == h_len_b[0];
  }
  h_slot[n] = pos;
  const uint64_t total = pos;   // == c.seq_bytes
  // Plain scorings, rows up to 1 024 columns: the fill writes ONE byte of directions per cell and nothing else
  // (sa_fill_dirs.hip) -- the three matrices are never needed, so they are not even allocated.
  const bool use_dirs = nw_dirs_applicable(ctx, sc, c.max_a, n);
  // ... in blocks of 8 x 16 cells where every row of the chunk has at most 512 columns (sa_kernels.h, round 6): a pair takes
  // ceil(rows / 8) x ceil(columns / 16) x 128 bytes and starts on a multiple of 256
  const bool blocked = use_dirs && sa_dirs_blocked_shape(c.max_a);
  auto dir_bytes = [&](uint32_t la, uint32_t lb) -> uint64_t {
    return ((blocked ? sa_dirs_blocked_bytes(la, lb) : (uint64_t)(la + 1ull) * (lb + 1ull)) + 255u) & ~(uint64_t)255u;
  };
  if (blocked) {
    uint64_t at = 0;
    for (uint64_t k = 0; k < n; ++k) { h_mat[k] = at; at += dir_bytes(h_len_a[k], h_len_b[k]); }
    cell = at;
  }
  // Every pair the same shape (reads of one length), match / mismatch scoring: two pairs per wave in packed int16
  // (sa_fill_dirs_x2.hip); each pair's bytes then start on a 256-byte boundary
  uint64_t stride = 0, mat_total = blocked ? cell : c.cells;
  // ... and a chunk whose pairs are MOSTLY of one shape (reads of one length, some trimmed): the pairs of that shape through
  // the packed kernel, the others through the one-pair kernel, each launch with the list of its pairs (SaFillParams::
  // pair_list); every pair's bytes start on a 256-byte boundary then
  uint32_t modal_a = 0, modal_b = 0;
  bool mixed = false;
  if (use_dirs && same_shape && (n >= kPackedFillMinPairs || ctx->opt.pack16 == 2) && nw_dirs_x2_applicable(ctx, sc, c.max_a, c.max_b)) {
    stride = dir_bytes(c.max_a, c.max_b);
    for (uint64_t k = 0; k < n; ++k) h_mat[k] = k * stride;
    mat_total = n * stride;
  } else if (use_dirs && !same_shape && ctx->opt.pack16 && (n >= kBucketedFillMinPairs || ctx->opt.pack16 == 2)) {
    // the majority shape, if there is one (Boyer-Moore vote, then an exact count: two passes of compares, no hashing --
    // a hash map over 125 000 pairs cost more host time than the packed kernel saves)
    uint64_t best_key = 0, votes = 0, best_count = 0;
    for (uint64_t k = 0; k < n; ++k) {
      const uint64_t key = (uint64_t)h_len_a[k] << 32 | h_len_b[k];
      if (votes == 0) { best_key = key; votes = 1; } else if (key == best_key) ++votes; else --votes;
    }
    for (uint64_t k = 0; k < n; ++k) best_count += ((uint64_t)h_len_a[k] << 32 | h_len_b[k]) == best_key;
    modal_a = (uint32_t)(best_key >> 32); modal_b = (uint32_t)best_key;
    if ((best_count >= kBucketedFillMinPairs || (ctx->opt.pack16 == 2 && best_count >= 2)) && best_count * 2 >= n &&
        nw_dirs_x2_applicable(ctx, sc, modal_a, modal_b)) {
      mixed = true;
      uint64_t at = 0;
      for (uint64_t k = 0; k < n; ++k) {
        h_mat[k] = at;
        at += dir_bytes(h_len_a[k], h_len_b[k]);
      }
      mat_total = at;
    }
  }
  if ((rc = ctx->arena.reserve(c.seq_bytes + 16)) || (rc = ctx->off_a.reserve(desc_bytes)) || (rc = ctx->status.reserve(n * 8)) ||
      (rc = ctx->t_out_a.reserve(2 * total + 16)) || (rc = ctx->t_meta.reserve(n * 16)) ||
      (rc = ctx->h_ta.reserve(2 * total + 16)) || (rc = ctx->h_tmeta.reserve(n * 16)))
    return rc;
  if (use_dirs) {
    if ((rc = ctx->dirs.reserve(mat_total + 4096)) || (rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8))) return rc;
  } else if ((rc = reserve_arenas(ctx, c.cells * 4))) {
    return rc;
  }
  if ((rc = ensure_copy_streams(ctx, 3))) return rc;
  // The walks: with direction bytes the fill is bound by VALU issue and a walk by the latency of its dependent byte loads,
  // so a group's walk runs on its own stream beside the next group's fills; the three-matrix fill saturates HBM and a walk
  // beside it only slows both down (profiles/r03/r03_nw_pipeline.md), so there the walk stays in the fills' stream.
  hipStream_t sf = ctx->stream, su = ctx->copy_streams[0], sd = ctx->copy_streams[1];

  // sub-batch s = pairs [cut[s], cut[s + 1]) of the chunk, cut at equal cells; group g = sub-batches [gcut[g], gcut[g + 1])
  std::vector<uint64_t> cut(n_sub + 1, n);
  cut[0] = 0;
  { uint64_t k = 0;
    for (uint32_t s = 1; s < n_sub; ++s) {
      const uint64_t want = mat_total / n_sub * s;
      while (k < n && h_mat[k] < want) ++k;
      cut[s] = std::max(k, cut[s - 1]);
    } }
  constexpr uint64_t kGroupPairs = 32768;
  std::vector<uint32_t> gcut{0};
  for (uint32_t s = 1; s < n_sub; ++s)
    if (cut[s] - cut[gcut.back()] >= kGroupPairs && n - cut[s] >= kGroupPairs / 2) gcut.push_back(s);
  gcut.push_back(n_sub);
  const uint32_t n_grp = (uint32_t)gcut.size() - 1;
  const bool walk_beside = use_dirs && ctx->opt.walk_overlap && n_grp > 1;   // (one group: nothing to run beside)
  hipStream_t sw = walk_beside ? ctx->copy_streams[2] : sf;
  // pinned / device buffers are reused by the next call: never leave work in flight
  StreamSyncOnExit sync_f(sf), sync_u(su), sync_d(sd), sync_w(sw);

  // mixed chunk: per sub-batch, the pairs of the modal shape and the others (indices into the chunk's arrays)
  std::vector<uint32_t> list_at;   // [2 s] / [2 s + 1]: where sub-batch s's modal / other pairs start in the list; [2 n_sub]: its end
  uint32_t *dv_list = nullptr;
  if (mixed) {
    if ((rc = ctx->h_misc.reserve(n * 4 + 16)) || (rc = ctx->pair_list.reserve(n * 4 + 16))) return rc;
    uint32_t *h_list = ctx->h_misc.as<uint32_t>();
    list_at.assign(2 * n_sub + 1, 0);
    uint32_t at = 0;
    for (uint32_t s2 = 0; s2 < n_sub; ++s2) {
      list_at[2 * s2] = at;
      for (uint64_t k = cut[s2]; k < cut[s2 + 1]; ++k) if (h_len_a[k] == modal_a && h_len_b[k] == modal_b) h_list[at++] = (uint32_t)k;
      list_at[2 * s2 + 1] = at;
      for (uint64_t k = cut[s2]; k < cut[s2 + 1]; ++k) if (!(h_len_a[k] == modal_a && h_len_b[k] == modal_b)) h_list[at++] = (uint32_t)k;
    }
    list_at[2 * n_sub] = at;
    dv_list = ctx->pair_list.as<uint32_t>();
    HIP_TRY(hipMemcpyAsync(dv_list, h_list, n * 4, hipMemcpyHostToDevice, su));
  }

  EventList ev;   // [0, n_sub): upload of s done; [n_sub, n_sub + n_grp): walk of g done; then: download of g done; then: fills of g done
  for (uint32_t k = 0; k < n_sub + 3 * n_grp; ++k) HIP_TRY(ev.add(hipEventDisableTiming));
  HIP_TRY(hipMemcpyAsync(ctx->off_a.p, h_off_a, desc_bytes, hipMemcpyHostToDevice, su));

  uint64_t *dv_off_a = ctx->off_a.as<uint64_t>(), *dv_off_b = dv_off_a + n, *dv_mat = dv_off_b + n, *dv_slot = dv_mat + n;
  uint32_t *dv_len_a = reinterpret_cast<uint32_t *>(dv_slot + n + 1), *dv_len_b = dv_len_a + n;
  char *d_chars = ctx->t_out_a.as<char>();
  uint32_t *d_meta = ctx->t_meta.as<uint32_t>();
  char *h_chars = ctx->h_ta.as<char>();
  uint32_t *h_meta = ctx->h_tmeta.as<uint32_t>();
  auto dev_range = [&](uint64_t k0, uint64_t k1) {
    seqalign_dev_batch_t d;
    d.n_pairs = k1 - k0; d.arena = ctx->arena.as<uint8_t>();
    d.off_a = dv_off_a + k0; d.len_a = dv_len_a + k0; d.off_b = dv_off_b + k0; d.len_b = dv_len_b + k0;
    d.mat_off = dv_mat + k0;
    d.match_scores = ctx->M.as<int32_t>(); d.gap_a_scores = ctx->A.as<int32_t>(); d.gap_b_scores = ctx->B.as<int32_t>();
    d.status = ctx->status.as<uint64_t>() + k0; d.max_len_a = c.max_a; d.max_len_b = c.max_b;
    return d;
  };

  constexpr uint64_t kPack = 1024;
  uint32_t g = 0;
  for (uint32_t s = 0; s < n_sub; ++s) {
    const uint64_t k0 = cut[s], k1 = cut[s + 1];
    if (k1 > k0) {
      const uint64_t c0 = h_slot[k0], c1 = h_slot[k1];
      parallel_for((k1 - k0 + kPack - 1) / kPack, [&](uint64_t blk) {   // host: this sub-batch's sequences
        for (uint64_t k = k0 + blk * kPack, e = std::min(k1, k0 + (blk + 1) * kPack); k < e; ++k) {
          const uint64_t p = c.first + k;
          memcpy(h_seq + h_off_a[k], batch->arena + batch->off_a[p], batch->len_a[p]);
          memcpy(h_seq + h_off_b[k], batch->arena + batch->off_b[p], batch->len_b[p]);
        }
      });
      if (c1 > c0) HIP_TRY(hipMemcpyAsync(ctx->arena.as<uint8_t>() + c0, h_seq + c0, c1 - c0, hipMemcpyHostToDevice, su));
    }
    HIP_TRY(hipEventRecord(ev.ev[s], su));
    HIP_TRY(hipStreamWaitEvent(sf, ev.ev[s], 0));
    if (k1 > k0) {
      const seqalign_dev_batch_t d = dev_range(k0, k1);
      if (use_dirs && mixed) {
        // one launch over the whole chunk's arrays with this sub-batch's list: the modal shape's pairs two per wave, the others
        // one per wave, side by side in one grid
        const seqalign_dev_batch_t dm = dev_range(0, n);
        const uint32_t m0 = list_at[2 * s], m1 = list_at[2 * s + 1], r1 = list_at[2 * s + 2];
        if ((rc = nw_dirs_fill_mixed(ctx, sc, &dm, ctx->dirs.as<uint8_t>(), ctx->best_score.as<int32_t>(), ctx->best_index.as<uint64_t>(),
                                     sf, dv_list + m0, m1 - m0, r1 - m1, modal_a, modal_b)))
          return rc;
      } else if (use_dirs) {
        bool used = false;
        if ((rc = nw_dirs_fill(ctx, sc, &d, ctx->dirs.as<uint8_t>(), ctx->best_score.as<int32_t>() + k0,
                               ctx->best_index.as<uint64_t>() + k0, sf, &used, stride)))
          return rc;
        if (!used) { set_last_error("seqalign_nw_batch: internal error: directions-only fill refused a batch it had accepted"); return SEQALIGN_E_HIP; }
      } else if ((rc = seqalign_fill_batch_device(ctx, sc, &d, SEQALIGN_KERNEL_AUTO, sf))) {
        return rc;
      }
    }
    if (s + 1 == gcut[g + 1]) {   // the group is filled: walk it, send it home
      const uint64_t g0 = cut[gcut[g]], g1 = cut[s + 1];
      if (g1 > g0) {
        const uint64_t c0 = h_slot[g0], c1 = h_slot[g1];
        const seqalign_dev_batch_t d = dev_range(g0, g1);
        SaTraceParams t;
        memset(&t, 0, sizeof(t));
        t.arena = d.arena; t.off_a = d.off_a; t.len_a = d.len_a; t.off_b = d.off_b; t.len_b = d.len_b; t.mat_off = d.mat_off;
        t.M = d.match_scores; t.A = d.gap_a_scores; t.B = d.gap_b_scores; t.code = sc->d_code; t.table = sc->d_table;
        t.str_off = dv_slot + g0;
        t.out_a = d_chars + 2 * c0 - c0;                  // + slot offset: a-strings at [2 c0, 2 c0 + (c1 - c0))
        t.out_b = d_chars + 2 * c0 + (c1 - c0) - c0;      //                b-strings right behind them
        t.out_meta4 = d_meta + 4 * g0; t.fill_status = d.status;
        if (use_dirs) { t.dirs = ctx->dirs.as<uint8_t>(); t.dirs_blocked = blocked; t.nw_score = ctx->best_score.as<int32_t>() + g0; t.nw_state = ctx->best_index.as<uint64_t>() + g0; }
        t.n_pairs = (uint32_t)(g1 - g0); t.K = sc->flat.n_classes; t.open1 = sc->flat.open1; t.ext = sc->flat.ext;
        t.gen_eq = sc->flat.gen_eq; t.gen_ne = sc->flat.gen_ne; t.flags = sc->flat.flags;
        t.tune_walker = ctx->opt.trace_kernel; t.tune_group = ctx->opt.walk_group;
        // the last group's walk has no fill to run beside: it stays in the fills' stream, right behind the last fill (on
        // its own stream it waited for the previous group's download -- streams share hardware queues; rocprofv3
        // timeline of C5's share: 0.6 ms)
        hipStream_t sg = g + 1 < n_grp ? sw : sf;
        if (sg != sf) {
          HIP_TRY(hipEventRecord(ev.ev[n_sub + 2 * n_grp + g], sf));
          HIP_TRY(hipStreamWaitEvent(sg, ev.ev[n_sub + 2 * n_grp + g], 0));
        }
        hipError_t e = sa_launch_nw_traceback(t, sg);
        if (e != hipSuccess) return fail_hip(e, "traceback launch");
        HIP_TRY(hipEventRecord(ev.ev[n_sub + g], sg));
        HIP_TRY(hipStreamWaitEvent(sd, ev.ev[n_sub + g], 0));
        if (c1 > c0) HIP_TRY(hipMemcpyAsync(h_chars + 2 * c0, d_chars + 2 * c0, 2 * (c1 - c0), hipMemcpyDeviceToHost, sd));
        HIP_TRY(hipMemcpyAsync(h_meta + 4 * g0, d_meta + 4 * g0, (g1 - g0) * 16, hipMemcpyDeviceToHost, sd));
      }
      HIP_TRY(hipEventRecord(ev.ev[n_sub + n_grp + g], sd));
      ++g;
    }
  }
  tm.lap("nw pipelined: all sub-batches packed + enqueued");

  // an error is the LOWEST failing pair's, whichever worker meets one first; pairs without an error have been written by
  // then (outputs are unspecified after a failed call)
  std::atomic<uint64_t> first_bad{~0ull};
  double wait_ms = 0, copy_ms = 0;   // (option timing: the lap below split into waiting for the GPU / copying out)
  auto now_ms = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
  for (g = 0; g < n_grp; ++g) {   // strings of group g from the pinned block into the caller's buffers
    const double t_w = now_ms();
    HIP_TRY(hipEventSynchronize(ev.ev[n_sub + n_grp + g]));
    const double t_c = now_ms();
    wait_ms += t_c - t_w;
    const uint64_t k0 = cut[gcut[g]], k1 = cut[gcut[g + 1]];
    if (k1 == k0) continue;
    const uint64_t c0 = h_slot[k0], c1 = h_slot[k1];
    const char *ha = h_chars + 2 * c0 - c0, *hb = h_chars + 2 * c0 + (c1 - c0) - c0;   // + slot offset
    parallel_for((k1 - k0 + kPack - 1) / kPack, [&](uint64_t blk) {
      for (uint64_t k = k0 + blk * kPack, e = std::min(k1, k0 + (blk + 1) * kPack); k < e; ++k) {
        const uint64_t p = c.first + k;
        const uint32_t head = h_meta[4 * k], len = h_meta[4 * k + 1], status = h_meta[4 * k + 3];
        if (status) {
          uint64_t seen = first_bad.load(std::memory_order_relaxed);
          const uint64_t mine = k << 8 | (uint64_t)(status & 255u);
          while (mine < seen && !first_bad.compare_exchange_weak(seen, mine, std::memory_order_relaxed)) {}
          continue;
        }
        memcpy(out_a + str_off[p], ha + h_slot[k] + head, len);   // left-align (needleman_wunsch.c:135-145)
        memcpy(out_b + str_off[p], hb + h_slot[k] + head, len);
        out_a[str_off[p] + len] = out_b[str_off[p] + len] = '\0';
        out_len[p] = len;
        out_score[p] = (int32_t)h_meta[4 * k + 2];
      }
    });
    copy_ms += now_ms() - t_c;
  }
  if (first_bad.load() != ~0ull) return (int)(first_bad.load() & 255u);
  if (ctx->opt.timing) fprintf(stderr, "[seqalign timing] nw pipelined: waiting for the groups %.3f ms, copying their strings out %.3f ms\n", wait_ms, copy_ms);
  tm.lap("nw pipelined: all groups unpacked");
  return SEQALIGN_OK;
}

// ---- seqalign_nw_batch on direction bytes: MOVES home, nothing staged --------------------------------------------------
// What the round-3 pipeline above spent outside its kernels on C2 (10 k pairs, 0.89 ms per call, of which fill 0.20 + walk
// 0.12): a serial pass over the pairs for the offsets, packing on 10 of 32 threads, a descriptor copy and a sequence copy
// (two copy-engine latencies + 3 MB at 40 GB/s before the fill could start), 6 MB of gapped strings home through the
// runtime's blit kernel, unpacking on 10 threads again -- and each of the pool's dispatches paid for waking 31 sleeping
// threads.  Here, for chunks the direction-byte fill takes (plain scorings, rows <= 1 024 columns):
//   * the fill reads the packed sequences and the descriptor arrays IN PLACE from pinned host memory (300 B per pair over
//     PCIe, hidden behind 0.2 ms of arithmetic), so the kernel is launched the moment the host has packed -- no copy, no
//     event, no second stream on the way in (option zero_copy & 1);
//   * the walk sends home two bits per alignment column instead of two characters (sa_traceback.hip: which of the two
//     strings has a gap there), written in place into pinned host memory (zero_copy & 2): C2 0.83 MB instead of 6.1 MB, no
//     blit kernel beside the next fill; the host threads expand them against the caller's sequences straight into the
//     caller's buffers (host/sa_moves.c: vpexpandb, 64 columns per instruction);
//   * every pass over the pairs is parallel, in blocks of kHostBlk pairs: sizes per block, a serial prefix over the
//     BLOCKS, then offsets + packing per block; the pool's workers stay awake between the dispatches of a call.
// Same kernels for the fill, same sub-batch / group structure as above for chunks large enough to pipeline.
namespace {
// pairs per host task: 512 (C2: 20 tasks for 32 threads would leave a third of them idle -- but a task below ~10 us is
// all dispatch); fewer when the caller forces more sub-batches than such blocks (sub-batches are cut at block boundaries)
uint64_t host_block_pairs(const seqalign_ctx *ctx, uint64_t n) {
  uint64_t blk = 512;
  if (ctx->opt.subbatches > 1) while (blk > 1 && n / blk < 2ull * ctx->opt.subbatches) blk /= 2;
  return blk;
}
struct BlkSum {
  uint64_t chars = 0, cells = 0, cells256 = 0, blk256 = 0;   // (blk256: the pairs' direction bytes in 8 x 16 blocks, each rounded up to 256)
  uint32_t max_a = 0, max_b = 0;
  bool same = true, too_large = false;
  bool packed = true;   // the block's sequences lie in the caller's arena as they will in ours: a, b, a, b, ... back to back
};
// sizes of pairs [first, first + n) per block of kHostBlk pairs; `same`: every pair of the block has the shape of pair `first`
void scan_blocks(const seqalign_batch_t *b, uint64_t first, uint64_t n, uint64_t kHostBlk, std::vector<BlkSum> &blk) {
  const uint64_t nb = (n + kHostBlk - 1) / kHostBlk;
  blk.assign(nb, BlkSum());
  if (!n) return;
  const uint32_t la0 = b->len_a[first], lb0 = b->len_b[first];
  parallel_for(nb, [&](uint64_t bi) {
    BlkSum s;
    for (uint64_t k = bi * kHostBlk, e = std::min(n, (bi + 1) * kHostBlk); k < e; ++k) {
      const uint32_t la = b->len_a[first + k], lb = b->len_b[first + k];
      const uint64_t cells = (uint64_t)(la + 1ull) * (lb + 1ull);
      s.chars += (uint64_t)la + lb; s.cells += cells; s.cells256 += (cells + 255u) & ~(uint64_t)255u;
      s.blk256 += (sa_dirs_blocked_bytes(la, lb) + 255u) & ~(uint64_t)255u;
      s.max_a = std::max(s.max_a, la); s.max_b = std::max(s.max_b, lb);
      s.same = s.same && la == la0 && lb == lb0;
      s.too_large = s.too_large || cells >= (1ull << 31);
      s.packed = s.packed && b->off_b[first + k] == b->off_a[first + k] + la &&
                 (k + 1 == e || b->off_a[first + k + 1] == b->off_b[first + k] + lb);
    }
    blk[bi] = s;
  });
}
}  // namespace

static int nw_chunk_moves(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c, const seqalign_dev_scoring *sc,
                          const std::vector<BlkSum> &blk, uint64_t kHostBlk, const uint64_t *str_off, char *out_a, char *out_b,
                          uint32_t *out_len, int32_t *out_score) {
  const uint64_t n = c.count, nb = blk.size();
  int rc;
  StageTimer tm(ctx->opt.timing);
  // in place over PCIe, measured (tools/nw_moves_bench.py, profiles/r04/r04_nw_moves_bench.txt): READING the sequences in
  // place costs more than it saves (C2 0.58 -> 0.65 ms, C5's share 3.8 -> 4.6: the fills' 64-byte reads are bound by the
  // read requests the GPU keeps in flight on the link, ~11 GB/s, where the copy engine moves 40-50); WRITING the moves in
  // place pays when the walker stores them coalesced (one wave per walk, whole words of 64 lanes: C2 0.58 -> 0.51 ms) and
  // loses when every lane stores its own 4 bytes (one lane per walk, the large groups: C5's share 3.6 -> 4.4).  So auto =
  // sequences through the copy engine, moves in place exactly when the walks run one wave each.
  const bool auto_zc = ctx->opt.zero_copy == 4u;
  // (round 6: with the direction byte's local form the tile walks are level with or ahead of the lane walkers at every batch size --
  //  30 000 pairs 0.99 / 1.04 ms, 65 536: 1.63 / 1.78, 125 000: 2.93 / 2.93-2.95 -- so they walk every chunk; dirs_local = 0: round 4's rule)
  const bool tile_walks = ctx->opt.trace_kernel ? ctx->opt.trace_kernel == 2 : (ctx->opt.dirs_local != 0 || n < SA_WALK_TILE_MAX);
  const bool zc_in = !auto_zc && (ctx->opt.zero_copy & 1u) != 0, zc_out = auto_zc ? tile_walks : (ctx->opt.zero_copy & 2u) != 0;
  // (tile walks: the fills write the direction byte's LOCAL form -- a cell's own comparisons, resolved by the walker: sa_kernels.h)
  const bool local = tile_walks && ctx->opt.dirs_local;
  bool same_shape = true;
  for (const BlkSum &s : blk) same_shape = same_shape && s.same;

  // ---- layout of the direction bytes: pairs back to back / every pair of ONE shape on a multiple of 256 cells (the packed
  // two-pairs-per-wave fill) / mostly one shape: every pair on a multiple of 256, the modal shape's pairs packed
  enum { kBackToBack, kUniform, kMixed } layout = kBackToBack;
  uint64_t stride = 0;
  // (round 6) rows of up to 512 columns: the direction bytes in blocks of 8 x 16 cells (sa_kernels.h) -- a pair takes
  // ceil(rows / 8) x ceil(columns / 16) x 128 bytes and starts on a multiple of 256 whatever the layout of the chunk
  const bool blocked = sa_dirs_blocked_shape(c.max_a);
  auto dir_bytes = [&](uint32_t la, uint32_t lb) -> uint64_t {
    return ((blocked ? sa_dirs_blocked_bytes(la, lb) : (uint64_t)(la + 1ull) * (lb + 1ull)) + 255u) & ~(uint64_t)255u;
  };
  const bool may_pack = ctx->opt.pack16 && (n >= kPackedFillMinPairs || ctx->opt.pack16 == 2);
  if (same_shape && may_pack && nw_dirs_x2_applicable(ctx, sc, c.max_a, c.max_b)) {
    layout = kUniform;
    stride = dir_bytes(c.max_a, c.max_b);
  } else if (!same_shape && may_pack && (n >= kBucketedFillMinPairs || ctx->opt.pack16 == 2) && (uint64_t)(c.max_a + 1ull) * (c.max_b + 1ull) <= kShapeTableMax &&
             nw_dirs_x2_applicable(ctx, sc, c.max_a, c.max_b)) {
    // ragged (reads trimmed to various lengths): pairs of EQUAL shape are found per sub-batch and go two per wave, the ones
    // left over one per wave, all in one grid per sub-batch (below: pair_up)
    layout = kMixed;
  }
  // where each block's pairs start: characters (sequences, string slots) and cells
  std::vector<uint64_t> chars_at(nb + 1, 0), cells_at(nb + 1, 0);
  for (uint64_t bi = 0; bi < nb; ++bi) {
    chars_at[bi + 1] = chars_at[bi] + blk[bi].chars;
    const uint64_t in_blk = std::min(n, (bi + 1) * kHostBlk) - bi * kHostBlk;
    cells_at[bi + 1] = cells_at[bi] + (layout == kUniform ? in_blk * stride : blocked ? blk[bi].blk256 : layout == kMixed ? blk[bi].cells256 : blk[bi].cells);
  }
  const uint64_t total = chars_at[nb], mat_total = cells_at[nb];

  // ---- buffers.  Pinned: descriptors, sequences, moves, per-pair words; device: the direction bytes and what the fill
  // tells the walk (end score / state, status) -- plus staging twins of the pinned blocks when zero_copy is off
  const size_t desc_bytes = (4 * n + 1) * sizeof(uint64_t) + 2 * n * sizeof(uint32_t);
  const uint64_t move_words = 2 * ((total >> 5) + n) + 2;
  if ((rc = ctx->h_desc.reserve(desc_bytes)) || (rc = ctx->h_arena.reserve(total + 64)) ||
      (rc = ctx->h_ta.reserve(move_words * 4)) || (rc = ctx->h_tmeta.reserve(n * 8)) ||
      (rc = ctx->dirs.reserve(mat_total + 4096)) || (rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8)) ||
      (rc = ctx->status.reserve(n * 8)))
    return rc;
  if (!zc_in && ((rc = ctx->arena.reserve(total + 64)) || (rc = ctx->off_a.reserve(desc_bytes)))) return rc;
  if (!zc_out && ((rc = ctx->t_out_a.reserve(move_words * 4)) || (rc = ctx->t_meta.reserve(n * 8)))) return rc;
  if ((rc = ensure_copy_streams(ctx, 3))) return rc;
  hipStream_t sf = ctx->stream, su = ctx->copy_streams[0], sd = ctx->copy_streams[1];

  uint64_t *h_off_a = ctx->h_desc.as<uint64_t>(), *h_off_b = h_off_a + n, *h_mat = h_off_b + n, *h_slot = h_mat + n;
  uint32_t *h_len_a = reinterpret_cast<uint32_t *>(h_slot + n + 1), *h_len_b = h_len_a + n;
  uint8_t *h_seq = ctx->h_arena.as<uint8_t>();
  uint32_t *h_moves = ctx->h_ta.as<uint32_t>(), *h_meta = ctx->h_tmeta.as<uint32_t>();
  // offsets: per block from its base (parallel)
  parallel_for(nb, [&](uint64_t bi) {
    uint64_t pos = chars_at[bi], cell = cells_at[bi];
    for (uint64_t k = bi * kHostBlk, e = std::min(n, (bi + 1) * kHostBlk); k < e; ++k) {
      const uint32_t la = batch->len_a[c.first + k], lb = batch->len_b[c.first + k];
      h_slot[k] = pos; h_off_a[k] = pos; pos += la; h_off_b[k] = pos; pos += lb;
      h_len_a[k] = la; h_len_b[k] = lb;
      h_mat[k] = cell;
      const uint64_t cells = (uint64_t)(la + 1ull) * (lb + 1ull);
      cell += layout == kUniform ? stride : blocked ? dir_bytes(la, lb) : layout == kMixed ? ((cells + 255u) & ~(uint64_t)255u) : cells;
    }
  });
  h_slot[n] = total;

  // sub-batch s = blocks [bcut[s], bcut[s + 1]), cut at equal cells; group g = sub-batches [gcut[g], gcut[g + 1])
  // (tried: a short last sub-batch as a group of its own, so that the walk + results + expansion nothing runs beside are
  // short -- C5's share 3.95 -> 4.05 ms, the extra launch and the quarter-size fill cost what the shorter tail saves)
  const uint32_t n_sub = (uint32_t)std::min<uint64_t>(pick_subbatches(ctx, c), nb);
  std::vector<uint64_t> bcut(n_sub + 1, nb);
  bcut[0] = 0;
  if (layout == kUniform && n_sub > 1 && n / n_sub >= 8192 && kHostBlk <= 4096 && 4096 % kHostBlk == 0) {
    // one shape, four pairs per wave: sub-batches of whole ROUNDS of waves (4 096 pairs = 1 024 waves, one per SIMD) -- C5's
    // share as 7 x 16 384 + 10 312 pairs is 31 rounds where 8 x 15 625 are 32
    const uint64_t per = (((n + n_sub - 1) / n_sub + 4095u) & ~(uint64_t)4095u) / kHostBlk;
    for (uint32_t s = 1; s < n_sub; ++s) bcut[s] = std::min<uint64_t>(nb, s * per);
  } else {
    uint64_t bi = 0;
    for (uint32_t s = 1; s < n_sub; ++s) {
      const uint64_t want = mat_total / n_sub * s;
      while (bi < nb && cells_at[bi] < want) ++bi;
      bcut[s] = std::max(bi, bcut[s - 1]);
    }
  }
  auto pair_at = [&](uint64_t bi) { return std::min(n, bi * kHostBlk); };
  constexpr uint64_t kGroupPairs = 32768;   // (measured, C5 share, walks in stream order: 2 groups 3.49 ms, 3 groups 3.31, 4 groups 3.44; round 6, tile walks: groups of 16 384 are level (C5's share 2.93-3.11 / 2.91-3.05, 65 536 pairs 1.85 / 1.70);
                                            //  with the lane walker's look-ahead 3 / 4 / 8 groups and a short last group: 3.12-3.27, no order)
  std::vector<uint32_t> gcut{0};
  for (uint32_t s = 1; s < n_sub; ++s)
    if (pair_at(bcut[s]) - pair_at(bcut[gcut.back()]) >= kGroupPairs && n - pair_at(bcut[s]) >= kGroupPairs / 2) gcut.push_back(s);
  gcut.push_back(n_sub);
  const uint32_t n_grp = (uint32_t)gcut.size() - 1;
  const bool walk_beside = ctx->opt.walk_overlap && n_grp > 1;
  hipStream_t sw = walk_beside ? ctx->copy_streams[2] : sf;
  StreamSyncOnExit sync_f(sf), sync_u(su), sync_d(sd), sync_w(sw);   // pinned / device buffers are reused by the next call

  // mixed chunk: per sub-batch, which pairs go two per wave (consecutive list entries 2u, 2u + 1 have the same shape) and which
  // alone.  SURVEY 8e's "bucket by shape": a direct-mapped table over (len_a, len_b) holds the one pair of each shape that is
  // still waiting for a partner -- one pass pairs the pairs up, a second collects who is left -- so a chunk of reads of every
  // length between 100 and 150 (2 601 shapes, ~6 pairs of each per sub-batch) runs ~5/6 of its pairs through the packed kernel
  // where round 3's majority vote found no majority at all.  The sub-batches are paired up in parallel, one table each.
  std::vector<uint32_t> list_at;
  const uint32_t *dv_list = nullptr;
  if (layout == kMixed) {
    const uint64_t entries = (uint64_t)(c.max_a + 1ull) * (c.max_b + 1ull);
    if ((rc = ctx->h_misc.reserve(n * 4 + 16)) || (!zc_in && (rc = ctx->pair_list.reserve(n * 4 + 16)))) return rc;
    uint32_t *h_list = ctx->h_misc.as<uint32_t>();
    list_at.assign(2 * n_sub + 1, 0);
    parallel_for(n_sub, [&](uint64_t s2) {
      static thread_local std::vector<uint32_t> tl_table;   // (one table per worker thread, kept: <= 4 MiB)
      std::vector<uint32_t> &table = tl_table;
      if (table.size() < entries) table.resize(entries);
      const uint64_t k0 = pair_at(bcut[s2]), k1 = pair_at(bcut[s2 + 1]);
      const uint32_t Wb = c.max_b + 1;
      for (uint64_t k = k0; k < k1; ++k) table[(uint64_t)h_len_a[k] * Wb + h_len_b[k]] = ~0u;   // (only the shapes that occur are reset)
      uint32_t at = (uint32_t)k0;
      for (uint64_t k = k0; k < k1; ++k) {
        uint32_t &slot = table[(uint64_t)h_len_a[k] * Wb + h_len_b[k]];
        if (slot == ~0u) { slot = (uint32_t)k; continue; }
        h_list[at++] = slot; h_list[at++] = (uint32_t)k;
        slot = ~0u;
      }
      const uint32_t paired_end = at;
      for (uint64_t k = k0; k < k1; ++k) {
        uint32_t &slot = table[(uint64_t)h_len_a[k] * Wb + h_len_b[k]];
        if (slot == (uint32_t)k) { h_list[at++] = (uint32_t)k; slot = ~0u; }
      }
      list_at[2 * s2] = (uint32_t)k0; list_at[2 * s2 + 1] = paired_end;
    });
    list_at[2 * n_sub] = (uint32_t)n;
    if (zc_in) dv_list = ctx->h_misc.dev_as<uint32_t>();
    else { dv_list = ctx->pair_list.as<uint32_t>(); HIP_TRY(hipMemcpyAsync(ctx->pair_list.p, h_list, n * 4, hipMemcpyHostToDevice, su)); }
  }

  EventList ev;   // [0, n_sub): upload of s done; then per group: walk done / results home / fills done; last: descriptors up
  for (uint32_t k = 0; k < n_sub + 3 * n_grp + 1; ++k) HIP_TRY(ev.add(hipEventDisableTiming));
  if (!zc_in && layout == kUniform) {
    // one shape: the device writes them itself, in the fills' stream (uniform_descriptors_kernel)
    hipLaunchKernelGGL(uniform_descriptors_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, sf, ctx->off_a.as<uint64_t>(),
                       n, c.max_a, c.max_b, stride);
    HIP_TRY(hipGetLastError());
    if (sw != sf) {
      HIP_TRY(hipEventRecord(ev.ev[n_sub + 3 * n_grp], sf));
      HIP_TRY(hipStreamWaitEvent(sw, ev.ev[n_sub + 3 * n_grp], 0));
    }
  } else if (!zc_in) {
    // the descriptor arrays (44 B per pair: C5's share 5.5 MB, more than a sub-batch's sequences) go up on the download
    // stream, which has nothing to do yet, beside the first sub-batch's sequences on the upload stream: two copy engines
    // (C5's share 4.2 -> 4.05 ms against the same copy ahead of the sequences on the upload stream)
    HIP_TRY(hipMemcpyAsync(ctx->off_a.p, h_off_a, desc_bytes, hipMemcpyHostToDevice, sd));
    HIP_TRY(hipEventRecord(ev.ev[n_sub + 3 * n_grp], sd));
    HIP_TRY(hipStreamWaitEvent(sf, ev.ev[n_sub + 3 * n_grp], 0));
    if (sw != sf) HIP_TRY(hipStreamWaitEvent(sw, ev.ev[n_sub + 3 * n_grp], 0));
  }

  const uint64_t *dv_off_a = zc_in ? ctx->h_desc.dev_as<uint64_t>() : ctx->off_a.as<uint64_t>();
  const uint64_t *dv_off_b = dv_off_a + n, *dv_mat = dv_off_b + n, *dv_slot = dv_mat + n;
  const uint32_t *dv_len_a = reinterpret_cast<const uint32_t *>(dv_slot + n + 1), *dv_len_b = dv_len_a + n;
  const uint8_t *dv_seq = zc_in ? ctx->h_arena.dev_as<uint8_t>() : ctx->arena.as<uint8_t>();
  uint32_t *dv_moves = zc_out ? ctx->h_ta.dev_as<uint32_t>() : ctx->t_out_a.as<uint32_t>();
  uint32_t *dv_meta = zc_out ? ctx->h_tmeta.dev_as<uint32_t>() : ctx->t_meta.as<uint32_t>();
  auto dev_range = [&](uint64_t k0, uint64_t k1) {
    seqalign_dev_batch_t d;
    d.n_pairs = k1 - k0; d.arena = dv_seq;
    d.off_a = dv_off_a + k0; d.len_a = dv_len_a + k0; d.off_b = dv_off_b + k0; d.len_b = dv_len_b + k0;
    d.mat_off = dv_mat + k0;
    d.match_scores = d.gap_a_scores = d.gap_b_scores = nullptr;
    d.status = ctx->status.as<uint64_t>() + k0; d.max_len_a = c.max_a; d.max_len_b = c.max_b;
    return d;
  };
  auto move_word = [&](uint64_t k) { return 2 * ((h_slot[k] >> 5) + k); };   // (sa_traceback.hip: move_slot)
  tm.lap("nw moves: sizes, offsets, buffers");

  uint32_t g = 0;
  for (uint32_t s = 0; s < n_sub; ++s) {
    const uint64_t b0 = bcut[s], b1 = bcut[s + 1], k0 = pair_at(b0), k1 = pair_at(b1);
    if (k1 > k0) {
      // host: this sub-batch's sequences into the pinned arena, and up -- in SLICES (round 5): a slice's copy to the device runs
      // beside the packing of the next one.  C2's single sub-batch packed for 42 us and then uploaded for 72 (3 MB at 42 GB/s)
      // before its fill could start; in four slices of quarter-block tasks (every pool thread has work in every slice) the last
      // byte is up ~20 us after the last one is packed.
      const uint64_t n_blocks = b1 - b0;
      const uint64_t n_slices = zc_in ? 1 : std::min<uint64_t>(ctx->opt.upload_slices ? ctx->opt.upload_slices : 1, std::max<uint64_t>(1, n_blocks / 2));
      constexpr uint64_t kParts = 4;   // tasks per block
      for (uint64_t sl = 0; sl < n_slices; ++sl) {
        const uint64_t sb0 = b0 + n_blocks * sl / n_slices, sb1 = b0 + n_blocks * (sl + 1) / n_slices;
        parallel_for((sb1 - sb0) * kParts, [&](uint64_t ti) {
          const uint64_t bi = sb0 + ti / kParts, part = ti % kParts;
          const uint64_t kb = bi * kHostBlk, ke = std::min(n, (bi + 1) * kHostBlk), span = ke - kb;
          const uint64_t ka = kb + span * part / kParts, kz = kb + span * (part + 1) / kParts;
          if (kz <= ka) return;
          if (blk[bi].packed) {   // the caller's arena already has them back to back (any batch built pair by pair): one copy
            const uint64_t bytes = (kz < n ? h_off_a[kz] : total) - h_off_a[ka];
            memcpy(h_seq + h_off_a[ka], batch->arena + batch->off_a[c.first + ka], bytes);
            return;
          }
          for (uint64_t k = ka; k < kz; ++k) {
            const uint64_t p = c.first + k;
            memcpy(h_seq + h_off_a[k], batch->arena + batch->off_a[p], h_len_a[k]);
            memcpy(h_seq + h_off_b[k], batch->arena + batch->off_b[p], h_len_b[k]);
          }
        });
        if (!zc_in) {
          const uint64_t c0 = h_slot[pair_at(sb0)], c1 = h_slot[pair_at(sb1)];
          // (tried for C2's single sub-batch: the sequences in two halves on the upload and the download stream -- the runtime
          // runs both host-to-device copies on one engine, 39 + 38 us one after the other instead of 72)
          if (c1 > c0) HIP_TRY(hipMemcpyAsync(ctx->arena.as<uint8_t>() + c0, h_seq + c0, c1 - c0, hipMemcpyHostToDevice, su));
        }
      }
      if (!zc_in) {
        HIP_TRY(hipEventRecord(ev.ev[s], su));
        HIP_TRY(hipStreamWaitEvent(sf, ev.ev[s], 0));
      }
      const seqalign_dev_batch_t d = dev_range(k0, k1);
      if (layout == kMixed) {
        const seqalign_dev_batch_t dm = dev_range(0, n);
        const uint32_t m0 = list_at[2 * s], m1 = list_at[2 * s + 1], r1 = list_at[2 * s + 2];
        if ((rc = nw_dirs_fill_mixed(ctx, sc, &dm, ctx->dirs.as<uint8_t>(), ctx->best_score.as<int32_t>(), ctx->best_index.as<uint64_t>(),
                                     sf, dv_list + m0, m1 - m0, r1 - m1, c.max_a, c.max_b, local)))
          return rc;
      } else {
        bool used = false;
        if ((rc = nw_dirs_fill(ctx, sc, &d, ctx->dirs.as<uint8_t>(), ctx->best_score.as<int32_t>() + k0,
                               ctx->best_index.as<uint64_t>() + k0, sf, &used, layout == kUniform ? stride : 0, nullptr, 0, local)))
          return rc;
        if (!used) { set_last_error("seqalign_nw_batch: internal error: directions-only fill refused a batch it had accepted"); return SEQALIGN_E_HIP; }
      }
    }
    if (s + 1 == gcut[g + 1]) {   // the group is filled: walk it
      const uint64_t g0 = pair_at(bcut[gcut[g]]), g1 = k1;
      if (g1 > g0) {
        const seqalign_dev_batch_t d = dev_range(g0, g1);
        SaTraceParams t;
        memset(&t, 0, sizeof(t));
        t.arena = d.arena; t.off_a = d.off_a; t.len_a = d.len_a; t.off_b = d.off_b; t.len_b = d.len_b; t.mat_off = d.mat_off;
        t.code = sc->d_code; t.table = sc->d_table;
        t.str_off = dv_slot + g0;
        t.stage_words = (c.max_a + c.max_b + 31u) >> 5;
        t.moves = dv_moves + 2 * g0;        // walk w of the launch is pair g0 + w: words 2 ((slot >> 5) + g0 + w)
        t.out_meta2 = dv_meta + 2 * g0;
        t.fill_status = d.status;
        t.dirs = ctx->dirs.as<uint8_t>(); t.dirs_blocked = blocked; t.dirs_local = local; t.tune_stage = ctx->opt.walk_stage; t.tune_tile = ctx->opt.walk_tile; t.nw_score = ctx->best_score.as<int32_t>() + g0; t.nw_state = ctx->best_index.as<uint64_t>() + g0;
        t.n_pairs = (uint32_t)(g1 - g0); t.K = sc->flat.n_classes; t.open1 = sc->flat.open1; t.ext = sc->flat.ext;
        t.gen_eq = sc->flat.gen_eq; t.gen_ne = sc->flat.gen_ne; t.flags = sc->flat.flags;
        t.tune_walker = ctx->opt.trace_kernel; t.tune_group = ctx->opt.walk_group;
        // the last group's walk has no fill to run beside: it stays in the fills' stream, right behind the last fill
        hipStream_t sg = g + 1 < n_grp ? sw : sf;
        if (sg != sf) {
          HIP_TRY(hipEventRecord(ev.ev[n_sub + 2 * n_grp + g], sf));
          HIP_TRY(hipStreamWaitEvent(sg, ev.ev[n_sub + 2 * n_grp + g], 0));
        }
        hipError_t e = sa_launch_nw_traceback(t, sg);
        if (e != hipSuccess) return fail_hip(e, "traceback launch");
        if (zc_out) {
          HIP_TRY(hipEventRecord(ev.ev[n_sub + n_grp + g], sg));
        } else {
          HIP_TRY(hipEventRecord(ev.ev[n_sub + g], sg));
          HIP_TRY(hipStreamWaitEvent(sd, ev.ev[n_sub + g], 0));
          const uint64_t w0 = move_word(g0), w1 = g1 < n ? move_word(g1) : move_words - 2;
          HIP_TRY(hipMemcpyAsync(h_moves + w0, ctx->t_out_a.as<uint32_t>() + w0, (w1 - w0) * 4, hipMemcpyDeviceToHost, sd));
          HIP_TRY(hipMemcpyAsync(h_meta + 2 * g0, ctx->t_meta.as<uint32_t>() + 2 * g0, (g1 - g0) * 8, hipMemcpyDeviceToHost, sd));
          HIP_TRY(hipEventRecord(ev.ev[n_sub + n_grp + g], sd));
        }
      } else {
        HIP_TRY(hipEventRecord(ev.ev[n_sub + n_grp + g], sf));
      }
      ++g;
    }
  }
  tm.lap("nw moves: packed + enqueued");

  // the groups' moves, as they land, expanded into the caller's strings
  std::atomic<uint64_t> first_bad{~0ull};
  for (g = 0; g < n_grp; ++g) {
    { const hipError_t e = wait_event_spinning(ev.ev[n_sub + n_grp + g]); if (e != hipSuccess) return fail_hip(e, "waiting for a group's walk"); }
    const uint64_t b0 = bcut[gcut[g]], b1 = bcut[gcut[g + 1]];
    constexpr uint64_t kOut = 128;   // pairs per task: C2's 10 000 pairs over all 32 threads
    const uint64_t k_lo = pair_at(b0), k_hi = pair_at(b1);
    parallel_for((k_hi - k_lo + kOut - 1) / kOut, [&](uint64_t blk_i) {
      for (uint64_t k = k_lo + blk_i * kOut, e = std::min(k_hi, k_lo + (blk_i + 1) * kOut); k < e; ++k) {
        const uint64_t p = c.first + k;
        if (k + 6 < e) {   // the GPU wrote the moves and the words: every first touch is a miss -- have the pair six ahead on its way
          const uint64_t kn = k + 6;
          const uint32_t nwn = (h_len_a[kn] + h_len_b[kn] + 31u) >> 5;
          const uint32_t *pn = h_moves + move_word(kn);
          __builtin_prefetch(h_meta + 2 * kn); __builtin_prefetch(pn + nwn - 1); __builtin_prefetch(pn + 2 * nwn - 1);
          __builtin_prefetch(batch->arena + batch->off_a[c.first + kn]);
        }
        const uint32_t n_moves = h_meta[2 * k + 1];
        int prc = SEQALIGN_OK;
        if (n_moves >= SA_MOVES_ERR) {
          prc = (int)(n_moves & 15u);
        } else {
          const uint32_t nw = (h_len_a[k] + h_len_b[k] + 31u) >> 5;
          const uint32_t *pa = h_moves + move_word(k);
          if (ctx->cigar_format)   // seqalign_nw_batch_cigar: run lengths straight from the two planes, no strings (str_off has n + 1 entries: the slots)
            prc = sa_cigar_nw_moves(batch->arena + batch->off_a[p], h_len_a[k], batch->arena + batch->off_b[p], h_len_b[k], pa, pa + nw,
                                    nw, n_moves, ctx->cigar_format, ctx->cigar_fold, out_a + str_off[p], str_off[p + 1] - str_off[p],
                                    &out_len[p], nullptr);
          else
            prc = sa_expand_nw_moves(batch->arena + batch->off_a[p], h_len_a[k], batch->arena + batch->off_b[p], h_len_b[k], pa, pa + nw,
                                     nw, n_moves, out_a + str_off[p], out_b + str_off[p], &out_len[p]);
          out_score[p] = (int32_t)h_meta[2 * k];
        }
        if (prc != SEQALIGN_OK) {   // the LOWEST failing pair's code is the call's (whatever thread meets it first)
          uint64_t seen = first_bad.load(std::memory_order_relaxed);
          const uint64_t mine = k << 8 | (uint64_t)prc;
          while (mine < seen && !first_bad.compare_exchange_weak(seen, mine, std::memory_order_relaxed)) {}
        }
      }
    });
  }
  tm.lap("nw moves: groups expanded");
  if (first_bad.load() != ~0ull) return (int)(first_bad.load() & 255u);
  return SEQALIGN_OK;
}

// The HOST legs of nw_chunk_moves alone -- sizes, offsets, packing the sequences; then the expansion of (synthetic: all
// MATCH) moves into the caller's strings -- with no device involved: what a rank's CPU share must sustain per call when
// eight ranks do this at once on two sockets (tools/host_scale.py).  Plain memory instead of pinned; same loops, same pool.
extern "C" int seqalign_host_legs_nw(const seqalign_batch_t *batch, const uint64_t *str_off, char *out_a, char *out_b,
                                     uint32_t *out_len, int iterations, double *pack_ms, double *expand_ms) {
  if (!batch || !str_off || !out_a || !out_b || !out_len || iterations <= 0 || !pack_ms || !expand_ms) return SEQALIGN_E_ARG;
  const uint64_t n = batch->n_pairs, kHostBlk = 512;
  if (!n) return SEQALIGN_OK;
  std::vector<BlkSum> blk;
  std::vector<uint64_t> off_a(n), off_b(n), mat(n), slot(n + 1), chars_at, cells_at;
  std::vector<uint32_t> la(n), lb(n);
  std::vector<uint8_t> seq;
  std::vector<uint32_t> moves;
  auto now_ms = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
  double t_pack = 0, t_expand = 0;
  for (int it = 0; it < iterations; ++it) {
    const double t0 = now_ms();
    scan_blocks(batch, 0, n, kHostBlk, blk);
    const uint64_t nb = blk.size();
    chars_at.assign(nb + 1, 0); cells_at.assign(nb + 1, 0);
    for (uint64_t bi = 0; bi < nb; ++bi) { chars_at[bi + 1] = chars_at[bi] + blk[bi].chars; cells_at[bi + 1] = cells_at[bi] + blk[bi].cells256; }
    const uint64_t total = chars_at[nb];
    if (seq.size() < total + 64) seq.resize(total + 64);
    if (moves.size() < 2 * ((total >> 5) + n) + 2) moves.assign(2 * ((total >> 5) + n) + 2, 0u);
    parallel_for(nb, [&](uint64_t bi) {
      uint64_t pos = chars_at[bi], cell = cells_at[bi];
      for (uint64_t k = bi * kHostBlk, e = std::min(n, (bi + 1) * kHostBlk); k < e; ++k) {
        la[k] = batch->len_a[k]; lb[k] = batch->len_b[k];
        slot[k] = pos; off_a[k] = pos; pos += la[k]; off_b[k] = pos; pos += lb[k];
        mat[k] = cell; cell += (((uint64_t)(la[k] + 1ull) * (lb[k] + 1ull)) + 255u) & ~(uint64_t)255u;
      }
    });
    parallel_for(nb, [&](uint64_t bi) {
      if (blk[bi].packed) { memcpy(seq.data() + off_a[bi * kHostBlk], batch->arena + batch->off_a[bi * kHostBlk], blk[bi].chars); return; }
      for (uint64_t k = bi * kHostBlk, e = std::min(n, (bi + 1) * kHostBlk); k < e; ++k) {
        memcpy(seq.data() + off_a[k], batch->arena + batch->off_a[k], la[k]);
        memcpy(seq.data() + off_b[k], batch->arena + batch->off_b[k], lb[k]);
      }
    });
    const double t1 = now_ms();
    constexpr uint64_t kOut = 128;
    std::atomic<int> bad{0};
    parallel_for((n + kOut - 1) / kOut, [&](uint64_t bi) {
      for (uint64_t k = bi * kOut, e = std::min(n, (bi + 1) * kOut); k < e; ++k) {
        const uint32_t nw = (la[k] + lb[k] + 31u) >> 5;
        const uint32_t *pa = moves.data() + 2 * ((slot[k] >> 5) + k);
        if (sa_expand_nw_moves(batch->arena + batch->off_a[k], la[k], batch->arena + batch->off_b[k], lb[k], pa, pa + nw, nw,
                               std::min(la[k], lb[k]), out_a + str_off[k], out_b + str_off[k], &out_len[k]))
          bad.store(1);
      }
    });
    const double t2 = now_ms();
    if (bad.load()) return SEQALIGN_E_TRACEBACK;
    t_pack += t1 - t0; t_expand += t2 - t1;
  }
  *pack_ms = t_pack / iterations; *expand_ms = t_expand / iterations;
  return SEQALIGN_OK;
}

// (CIGAR mode -- ctx->cigar_format, set by seqalign_nw_batch_cigar: only the direction-byte path, whose walks come home as bit
// planes, delivers it; for every other path kNeedsStrings tells the caller to run the call for strings and convert them)
static constexpr int kNeedsStrings = -1;
static int nw_batch_impl(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                         const uint64_t *str_off, char *out_a, char *out_b, uint32_t *out_len,
                         int32_t *out_score) {
  const seqalign_batch_t *b = batch;
  if (!b || (b->n_pairs && (!b->arena || !b->off_a || !b->off_b || !b->len_a || !b->len_b))) return SEQALIGN_E_ARG;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  int rc;
  // one parallel pass over the lengths: validity (check_batch) and the sizes chunk planning needs
  std::vector<BlkSum> blk;
  const uint64_t blk_pairs = host_block_pairs(ctx, batch->n_pairs);
  scan_blocks(batch, 0, batch->n_pairs, blk_pairs, blk);
  Chunk whole;
  whole.count = batch->n_pairs;
  uint64_t cells256 = 0;
  for (const BlkSum &s : blk) {
    if (s.too_large) return SEQALIGN_E_TOO_LARGE;
    whole.cells += s.cells; whole.seq_bytes += s.chars; cells256 += std::max(s.cells256, s.blk256);   // (the larger of the two layouts' bytes: which one a chunk takes follows from its widest row)
    whole.max_a = std::max(whole.max_a, s.max_a); whole.max_b = std::max(whole.max_b, s.max_b);
  }
  HIP_TRY(hipSetDevice(ctx->device));
  seqalign_dev_scoring *sc = nullptr;
  if ((rc = cached_scoring(ctx, scoring, 0, &sc))) return rc;
  const bool on_host = traceback_on_host(ctx);
  if (!on_host && ctx->opt.nw_moves && nw_dirs_applicable(ctx, sc, whole.max_a, batch->n_pairs)) {
    // direction bytes (1 B per cell, every pair rounded up to 256 at most) + ~64 B per pair of descriptors and results:
    // C5's 1 M pairs are 23 GB -- one chunk, where 12 B per cell cut them into six
    if (cells256 + 64 * batch->n_pairs + whole.seq_bytes <= ctx->chunk_budget)
      return nw_chunk_moves(ctx, batch, whole, sc, blk, blk_pairs, str_off, out_a, out_b, out_len, out_score);
    std::vector<uint64_t> extra(batch->n_pairs, 256 + 64);
    if (SA_DIRS_BLOCKED)   // (in blocks of 8 x 16 cells a pair's direction bytes are up to 8 columns' + 16 rows' worth more than its cells)
      for (uint64_t p = 0; p < batch->n_pairs; ++p) {
        const uint64_t cells = (uint64_t)(batch->len_a[p] + 1ull) * (batch->len_b[p] + 1ull), bb = sa_dirs_blocked_bytes(batch->len_a[p], batch->len_b[p]);
        if (bb > cells) extra[p] += bb - cells;
      }
    for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget, 1, extra.data())) {
      const uint64_t bp = host_block_pairs(ctx, c.count);
      scan_blocks(batch, c.first, c.count, bp, blk);
      if ((rc = nw_chunk_moves(ctx, batch, c, sc, blk, bp, str_off, out_a, out_b, out_len, out_score))) return rc;
    }
    return SEQALIGN_OK;
  }
  if (ctx->cigar_format) return kNeedsStrings;
  // host mode: matrices come back through pinned staging, so chunks are also bounded by host memory
  const size_t budget = on_host ? std::min<size_t>(ctx->chunk_budget, (size_t)6 << 30) : ctx->chunk_budget;
  for (const Chunk &c : plan_chunks(batch, budget)) {
    seqalign_dev_batch_t d;
    if (!on_host) {
      // (one sub-batch = the plain sequence upload, fill, walk, download on the same code path)
      if ((rc = nw_chunk_pipelined(ctx, batch, c, sc, pick_subbatches(ctx, c), str_off, out_a, out_b, out_len, out_score))) return rc;
      continue;
    }
    if ((rc = run_chunk(ctx, batch, c, sc, &d))) return rc;
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

extern "C" int seqalign_nw_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                 const uint64_t *str_off, char *out_a, char *out_b, uint32_t *out_len,
                                 int32_t *out_score) {
  if (!ctx || !scoring || !str_off || !out_a || !out_b || !out_len || !out_score) return SEQALIGN_E_ARG;
  CallScope scope(ctx);
  return nw_batch_impl(ctx, batch, scoring, str_off, out_a, out_b, out_len, out_score);
}

uint64_t sa_host::put_alignment(const seqalign_ctx *ctx, const char *sa, const char *sb, uint32_t len, char *out_a, char *out_b,
                                uint64_t at, uint64_t room) {
  if (!ctx->cigar_format) {
    if ((uint64_t)len + 1 > room) return 0;
    memcpy(out_a + at, sa, len); memcpy(out_b + at, sb, len);
    out_a[at + len] = out_b[at + len] = '\0';
    return (uint64_t)len + 1;
  }
  if (!room) return 0;
  const size_t n = seqalign_cigar(sa, sb, len, ctx->cigar_format == 2, ctx->cigar_fold, out_a + at, room);
  return n == (size_t)-1 ? 0 : (uint64_t)n + 1;
}

// Global alignments as CIGAR (include/seqalign_hip.h).  On the direction-byte path the walks come home as two bit planes and the
// host run-length encodes those (host/sa_moves.c: sa_cigar_nw_moves): no string is ever written.  The other paths (three
// matrices, traceback = host) produce strings: the call runs for strings into its own buffers and encodes them.
extern "C" int seqalign_nw_batch_cigar(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring, int format,
                                       const uint64_t *cigar_off, char *cigar, uint32_t *cigar_len, int32_t *out_score) {
  if (!ctx || !scoring || !cigar_off || !cigar || !cigar_len || !out_score || (format != SEQALIGN_CIGAR_M && format != SEQALIGN_CIGAR_EQX))
    return SEQALIGN_E_ARG;
  CallScope scope(ctx);
  if (!batch) return SEQALIGN_E_ARG;
  for (uint64_t p = 0; p < batch->n_pairs; ++p)
    if (cigar_off[p + 1] < cigar_off[p]) return SEQALIGN_E_ARG;
  int rc;
  {
    CigarScope mode(ctx, format, !scoring->case_sensitive);
    rc = nw_batch_impl(ctx, batch, scoring, cigar_off, cigar, cigar, cigar_len, out_score);
  }
  if (rc != kNeedsStrings) return rc;
  const uint64_t n = batch->n_pairs;
  std::vector<uint64_t> off(n + 1, 0);
  for (uint64_t p = 0; p < n; ++p) off[p + 1] = off[p] + batch->len_a[p] + batch->len_b[p] + 1;
  std::vector<char> sa(off[n] + 1), sb(off[n] + 1);
  std::vector<uint32_t> len(n);
  if ((rc = nw_batch_impl(ctx, batch, scoring, off.data(), sa.data(), sb.data(), len.data(), out_score))) return rc;
  std::atomic<uint64_t> first_bad{~0ull};
  parallel_for((n + 255) / 256, [&](uint64_t blk) {
    for (uint64_t p = blk * 256, e = std::min(n, (blk + 1) * 256); p < e; ++p) {
      const uint64_t cap = cigar_off[p + 1] - cigar_off[p];
      const size_t got = cap ? seqalign_cigar(sa.data() + off[p], sb.data() + off[p], len[p], format == SEQALIGN_CIGAR_EQX,
                                              !scoring->case_sensitive, cigar + cigar_off[p], cap) : (size_t)-1;
      if (got == (size_t)-1) {
        uint64_t seen = first_bad.load(std::memory_order_relaxed);
        while (p < seen && !first_bad.compare_exchange_weak(seen, p, std::memory_order_relaxed)) {}
      } else {
        cigar_len[p] = (uint32_t)got;
      }
    }
  });
  return first_bad.load() == ~0ull ? SEQALIGN_OK : SEQALIGN_E_NOMEM;
}
