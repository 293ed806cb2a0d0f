// sa_batch_sw.hip -- seqalign_sw_batch: local alignment over a HOST batch (best hit on the device,
// multi-hit enumeration on the device, or the literal host enumeration).
#include "sa_ctx.hpp"

using namespace sa_host;

// ----------------------------------------------- host-level: SW over a batch ---
namespace {

struct Cand { uint32_t idx; int32_t score; };

struct PairHits {
  std::vector<seqalign_sw_hit_t> hits;   // str_off relative to str_a / str_b below
  std::string str_a, str_b;
};


}  // namespace

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


// up to this many hits per pair the enumeration runs on the device
static const uint32_t kDeviceEnumMaxHits = 16;

static uint32_t bits_for(uint64_t v) {   // bits needed to hold values 0..v
  uint32_t b = 1;
  while (b < 64 && (v >> b) != 0) ++b;
  return b;
}

// How a chunk's candidate keys are laid out (SaFillParams): row and column fields sized by the chunk's longest
// sequences, the score field by the largest score the scoring can produce on them.
static SaKeyLayout key_layout(const seqalign_dev_scoring *sc, const Chunk &c, int32_t thr_min) {
  const sa_flat_scoring_t &f = sc->flat;
  int64_t best_step = std::max<int64_t>(1, std::max(f.gen_eq, f.gen_ne));
  for (uint64_t k = 0; k < (uint64_t)f.n_classes * f.n_classes; ++k)
    if (f.table[k] != SA_S_BLOCKED && f.table[k] != SA_S_UNKNOWN) best_step = std::max<int64_t>(best_step, f.table[k]);
  // every move adds at most best_step; gaps only add when a gap score is positive (legal, absurd)
  int64_t cap = (int64_t)std::min(c.max_a, c.max_b) * best_step;
  if (f.ext > 0 || f.open1 > 0)
    cap = ((int64_t)c.max_a + c.max_b) * std::max<int64_t>(best_step, std::max(f.ext, f.open1));
  cap = std::min<int64_t>(std::max<int64_t>(cap, thr_min), INT32_MAX);
  SaKeyLayout l;
  l.cap = (int32_t)cap;
  l.row_bits = bits_for(c.max_b);
  l.col_bits = bits_for(c.max_a);
  l.score_bits = bits_for((uint64_t)(cap - std::min<int64_t>(cap, std::max(thr_min, 1))));
  l.key64 = (l.row_bits + l.col_bits + l.score_bits > 32) ? 1u : 0u;
  return l;
}

// SW hits of one chunk, enumerated on the device:
//   fill (emits the candidate keys) -> per-pair key sort (sa_sort.hip) -> enumeration (sa_sw_enum_window.hip,
//   generic kernel for the pairs it flags) -> gather strings -> D2H.
// One host round trip in the middle (the candidates' bounding boxes size the enumeration's LDS window).
// Appends to the caller's hit array / string buffers.
//
// want_hits > max_hits (the caller asked for more hits than the device slots hold): pairs that fill all
// max_hits slots with candidates still left are finished on the host -- their matrices and sorted keys are
// still in the context's scratch -- with the full limit; the others (nearly all, in practice) are done.
static int sw_chunk_device_enumerate(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c,
                                     const scoring_t *scoring, const seqalign_dev_scoring *sc,
                                     const int32_t *min_score, uint32_t max_hits,
                                     uint32_t want_hits, seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *found,
                                     char *out_a, char *out_b, uint64_t str_cap, uint64_t *used_str) {
  const uint64_t n = c.count;
  hipStream_t st = ctx->stream;
  int rc;
  DevBuf &d_min = ctx->e[0], &d_keys = ctx->e[1], &d_tmp = ctx->e[2], &d_box = ctx->e[3], &d_dir = ctx->e[4],
         &d_mask = ctx->e[5], &d_offs = ctx->e[6], &d_hits = ctx->e[7], &d_meta = ctx->e[8],
         &d_gath_a = ctx->e[9], &d_gath_b = ctx->e[10];
  StreamSyncOnExit sync_on_exit(st);   // async copies below target function-local vectors
  StageTimer tm;

  int32_t thr = min_score[c.first];
  for (uint64_t k = 1; k < n; ++k) thr = std::min(thr, min_score[c.first + k]);
  const SaKeyLayout layout = key_layout(sc, c, thr);
  const size_t key_bytes = layout.key64 ? 8 : 4;

  // ---- fill + candidate keys
  if ((rc = d_min.reserve(n * 4)) || (rc = ctx->cand_count.reserve(n * 4)) || (rc = d_box.reserve(n * 16)) ||
      (rc = d_keys.reserve(c.cells * key_bytes + 16)) || (rc = d_tmp.reserve(c.cells * key_bytes + 16)))
    return rc;
  HIP_TRY(hipMemcpyAsync(d_min.p, min_score + c.first, n * 4, hipMemcpyHostToDevice, st));
  SaCandKeys cand;
  cand.keys = d_keys.p; cand.tmp = d_tmp.p; cand.cand_count = ctx->cand_count.as<uint32_t>();
  cand.cand_box = d_box.as<uint32_t>(); cand.cand_min = d_min.as<int32_t>(); cand.layout = layout;
  seqalign_dev_batch_t d;
  bool emitted = false;
  if ((rc = run_chunk(ctx, batch, c, sc, &d, nullptr, &cand, &emitted))) return rc;
  hipError_t e;
  if (!emitted) {   // a fill kernel that cannot emit keys itself: one pass over match_scores
    SaReduceParams r;
    memset(&r, 0, sizeof(r));
    r.len_a = d.len_a; r.len_b = d.len_b; r.mat_off = d.mat_off; r.M = d.match_scores; r.n_pairs = (uint32_t)n;
    if ((e = sa_launch_sw_emit(r, cand, st)) != hipSuccess) return fail_hip(e, "sw candidate emission");
  }
  // Two streams from here on.  Main: key sort, then the enumeration.  Side: everything that needs the fill's
  // matrices and boxes but not the sorted keys -- the boxes' way to the host, the class decision, the direction
  // bytes -- so that the direction kernels run next to the sort instead of after it.
  if (!ctx->stream2) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  hipStream_t side = ctx->stream2;
  StreamSyncOnExit sync_side_on_exit(side);
  EventList events;   // 0: fill done, 1: sort done, 2: direction bytes done, 3: side stream's generic kernel done
  for (int k = 0; k < 4; ++k) HIP_TRY(events.add(hipEventDisableTiming));
  HIP_TRY(hipEventRecord(events.ev[0], st));
  SaSortParams sp;
  memset(&sp, 0, sizeof(sp));
  sa_sort_plan(layout, &sp);
  sp.mat_off = d.mat_off; sp.cand_count = cand.cand_count; sp.keys = cand.keys; sp.tmp = cand.tmp; sp.n_pairs = (uint32_t)n;
  if ((e = sa_launch_sort_keys(sp, st)) != hipSuccess) return fail_hip(e, "candidate sort");
  HIP_TRY(hipEventRecord(events.ev[1], st));
  const void *sorted = (sp.n_passes & 1) ? cand.tmp : cand.keys;
  tm.lap("sw: enqueue fill + emit + sort");

  // ---- the one round trip: counts + bounding boxes (as soon as the fill is done)
  std::vector<uint32_t> count(n), box(4 * n);
  HIP_TRY(hipStreamWaitEvent(side, events.ev[0], 0));
  HIP_TRY(hipMemcpyAsync(count.data(), ctx->cand_count.p, n * 4, hipMemcpyDeviceToHost, side));
  HIP_TRY(hipMemcpyAsync(box.data(), d_box.p, n * 16, hipMemcpyDeviceToHost, side));

  // host prefixes meanwhile: visited-bitmap words (generic lane kernel), string slots
  std::vector<uint64_t> offs(3 * (n + 1)), cell0(n + 1, 0);
  uint64_t *mask_off = offs.data(), *str_off = mask_off + n + 1, *dst_off = str_off + n + 1;
  uint64_t mask_words = 0, str_total = 0, max_mask_words = 0;
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t p = c.first + k, la = batch->len_a[p], lb = batch->len_b[p];
    cell0[k + 1] = cell0[k] + (la + 1) * (lb + 1);
    mask_off[k] = mask_words; mask_words += ((la + 1) * (lb + 1) + 31) / 32;
    max_mask_words = std::max(max_mask_words, ((la + 1) * (lb + 1) + 31) / 32);
    str_off[k] = str_total; str_total += (uint64_t)max_hits * (la + lb);
  }
  mask_off[n] = mask_words; str_off[n] = str_total;
  const char *force = getenv("SEQALIGN_SW_ENUM");   // "wave" / "lane": the generic kernels for every pair (tests, experiments)
  const bool lane_kernel = max_mask_words * 4 > 65536 || (force && force[0] == 'l');   // generic kernel, bitmap in HBM
  if ((rc = d_offs.reserve(offs.size() * 8)) || (rc = ctx->t_out_a.reserve(str_total + 16)) ||
      (rc = ctx->t_out_b.reserve(str_total + 16)) || (rc = d_hits.reserve(n * max_hits * sizeof(SaDevHit) + 16)) ||
      (rc = d_meta.reserve(n * 12)) || (lane_kernel && (rc = d_mask.reserve(mask_words * 4 + 4))) ||
      (rc = d_dir.reserve(sa_dir_bytes(c.cells, n))))
    return rc;
  const bool trace = getenv("SEQALIGN_ENUM_TRACE") != nullptr;   // development aid: per-pair phase cycles on stderr
  HIP_TRY(hipMemsetAsync(d_meta.p, 0, n * 12, side));   // enum_status starts clean: the direction kernel may flag pairs
  HIP_TRY(hipMemcpyAsync(d_offs.p, offs.data(), 2 * (n + 1) * 8, hipMemcpyHostToDevice, side));
  if (lane_kernel) HIP_TRY(hipMemsetAsync(d_mask.p, 0, mask_words * 4, side));
  const uint64_t *dv_mask_off = d_offs.as<uint64_t>(), *dv_str_off = dv_mask_off + n + 1;
  HIP_TRY(hipStreamSynchronize(side));
  tm.lap("sw: wait (fill), boxes");

  // window the enumeration wants: the candidates' box plus room for the part of a hit that lies below
  // min_score (the kernel extends it further where LDS allows and flags a pair whose walk leaves it)
  const sa_flat_scoring_t &f = sc->flat;
  int64_t best_step = std::max<int64_t>(1, std::max(f.gen_eq, f.gen_ne));
  for (uint64_t k = 0; k < (uint64_t)f.n_classes * f.n_classes; ++k)
    if (f.table[k] != SA_S_BLOCKED && f.table[k] != SA_S_UNKNOWN) best_step = std::max<int64_t>(best_step, f.table[k]);
  // LDS class of every pair with candidates (sa_enum_classes): the first whose window holds the box + margin
  SaEnumClass cls[4];
  const int n_cls = sa_enum_classes(layout.key64, cls);
  if (const char *env = getenv("SEQALIGN_ENUM_THREADS")) {   // tests: one class with that many threads for every pair
    const int t = atoi(env);
    if (t == 256 || t == 512 || t == 1024) {
      for (int k = 0; k < n_cls; ++k) { cls[k] = cls[2]; cls[k].threads = (uint32_t)t; }
    }
  }
  // margin the class decision reserves above / left of the candidates' box, in cells: base + mult * (min_score / best
  // move) -- the part of a hit that lies below min_score; SEQALIGN_ENUM_MARGIN="base,mult" (tuning experiments)
  uint64_t margin_base = 16, margin_mult = 1;
  if (const char *env = getenv("SEQALIGN_ENUM_MARGIN")) {
    unsigned long long b = 0, m = 0;
    if (sscanf(env, "%llu,%llu", &b, &m) == 2) { margin_base = b; margin_mult = m; }
  }
  std::vector<std::vector<uint32_t>> members((size_t)n_cls);
  uint64_t class_need[4] = {0, 0, 0, 0};   // largest window a member asks for
  std::vector<uint64_t> need_of(trace ? n : 0, 0);
  uint64_t unplaced = 0;
  for (uint64_t k = 0; k < n; ++k) {
    if (!count[k]) continue;
    const uint64_t rmin = box[4 * k], rmax = box[4 * k + 1], cmin = box[4 * k + 2], cmax = box[4 * k + 3];
    const uint64_t margin = margin_base + margin_mult * (uint64_t)((std::max(min_score[c.first + k], 1) + best_step - 1) / best_step);
    const uint64_t r0 = rmin > margin ? rmin - margin : 0, c0 = cmin > margin ? cmin - margin : 0;
    const uint64_t need = (rmax - r0 + 2) * (cmax - c0 + 2);   // stored with a sentinel row and column
    const uint64_t bare = (rmax - rmin + 2) * (cmax - cmin + 2);
    if (trace) need_of[k] = need;
    int pick = -1;
    for (int q2 = 0; q2 < n_cls && pick < 0; ++q2)
      if (need <= cls[q2].window_bytes) pick = q2;
    if (pick < 0 && bare <= cls[n_cls - 1].window_bytes) pick = n_cls - 1;   // no room for the full margin
    if (pick < 0) { ++unplaced; continue; }                    // generic kernel (flagged below)
    members[(size_t)pick].push_back((uint32_t)k);
    class_need[pick] = std::max(class_need[pick], std::min<uint64_t>(need, cls[pick].window_bytes));
  }
  // LDS a class's members leave unused goes to its claim table (fewer hash collisions, fewer iterations)
  for (int q2 = 0; q2 < n_cls; ++q2) {
    const size_t per_cu = q2 == 0 ? 3 : q2 == 1 ? 2 : 1, total = std::min<size_t>(160u * 1024u / per_cu - 2048u, sa_enum_window_lds_limit());
    const size_t other = (size_t)12 * cls[q2].threads + (size_t)2 * cls[q2].threads * key_bytes + ((class_need[q2] + 15) & ~(uint64_t)15);
    while (cls[q2].claim_bits < 13 && ((size_t)4 << (cls[q2].claim_bits + 1)) + other <= total) ++cls[q2].claim_bits;
    // ... and what is left after that is the window (room for a larger margin than the members asked for)
    const size_t fixed_part = ((size_t)4 << cls[q2].claim_bits) + (size_t)12 * cls[q2].threads + (size_t)2 * cls[q2].threads * key_bytes;
    if (total > fixed_part) cls[q2].window_bytes = (uint32_t)std::max<size_t>(class_need[q2], (total - fixed_part) & ~(size_t)15);
  }
  std::vector<uint32_t> pair_list;
  pair_list.reserve(n);
  for (int q2 = 0; q2 < n_cls; ++q2) pair_list.insert(pair_list.end(), members[(size_t)q2].begin(), members[(size_t)q2].end());
  DevBuf &d_list = ctx->e[11];
  if ((rc = d_list.reserve(n * 4 + 128 * n * (trace ? 1 : 0) + 32))) return rc;
  if (!pair_list.empty())
    HIP_TRY(hipMemcpyAsync(d_list.p, pair_list.data(), pair_list.size() * 4, hipMemcpyHostToDevice, side));

  // ---- enumeration
  SaEnumParams q;
  memset(&q, 0, sizeof(q));
  q.arena = d.arena; q.off_a = d.off_a; q.len_a = d.len_a; q.off_b = d.off_b; q.len_b = d.len_b;
  q.mat_off = d.mat_off; q.M = d.match_scores; q.A = d.gap_a_scores; q.B = d.gap_b_scores;
  q.code = sc->d_code; q.table = sc->d_table; q.keys = sorted; q.cand_count = cand.cand_count;
  q.cand_box = cand.cand_box; q.min_score = d_min.as<int32_t>();
  q.mask = lane_kernel ? d_mask.as<uint32_t>() : nullptr; q.mask_off = dv_mask_off; q.str_off = dv_str_off;
  q.out_a = ctx->t_out_a.as<char>(); q.out_b = ctx->t_out_b.as<char>(); q.hits = d_hits.as<SaDevHit>();
  uint32_t *d_m = d_meta.as<uint32_t>();
  q.hit_count = d_m; q.str_used = d_m + n; q.enum_status = d_m + 2 * n;
  q.n_pairs = (uint32_t)n; q.K = sc->flat.n_classes; q.max_hits = max_hits; q.open1 = sc->flat.open1;
  q.ext = sc->flat.ext; q.gen_eq = sc->flat.gen_eq; q.gen_ne = sc->flat.gen_ne; q.flags = sc->flat.flags;
  q.max_mask_words = (uint32_t)std::min<uint64_t>(max_mask_words, 0xffffffffu);
  q.layout = layout;
  q.best_step = (uint32_t)std::min<int64_t>(best_step, INT32_MAX);
  q.max_len_a = c.max_a; q.max_len_b = c.max_b;
  q.dir = d_dir.as<uint8_t>();
  unsigned long long *d_trace = reinterpret_cast<unsigned long long *>(d_list.as<char>() + ((n * 4 + 15) & ~(uint64_t)15));
  q.trace = trace ? d_trace : nullptr;
  if (trace) HIP_TRY(hipMemsetAsync(d_trace, 0, n * 128, side));
  const bool generic_only = force && (force[0] == 'w' || force[0] == 'l');
  std::vector<uint32_t> flags;   // (function scope: the copy below is asynchronous)
  bool side_generic = false;
  if (!generic_only && unplaced) {   // pairs whose candidates' box fits no window: flagged for the generic kernel
    flags.assign(n, 0u);
    std::vector<char> placed(n, 0);
    for (uint32_t k : pair_list) placed[k] = 1;
    for (uint64_t k = 0; k < n; ++k) flags[k] = (count[k] && !placed[k]) ? SA_ENUM_GENERIC : 0u;
    HIP_TRY(hipMemcpyAsync(q.enum_status, flags.data(), n * 4, hipMemcpyHostToDevice, side));   // before the window kernels
  }
  if (!generic_only) {
    // direction bytes, class by class, on the side stream (next to the sort) ...
    std::vector<SaEnumParams> launches;
    uint32_t first = 0;
    for (int q2 = 0; q2 < n_cls; ++q2) {
      const uint32_t cnt = (uint32_t)members[(size_t)q2].size();
      if (!cnt) continue;
      SaEnumParams w = q;
      w.pair_list = d_list.as<uint32_t>() + first; w.n_list = cnt;
      w.threads = cls[q2].threads; w.claim_bits = cls[q2].claim_bits; w.window_bytes = cls[q2].window_bytes;
      if ((e = sa_launch_sw_direction(w, side)) != hipSuccess) return fail_hip(e, "sw direction bytes");
      launches.push_back(w);
      first += cnt;
    }
    HIP_TRY(hipEventRecord(events.ev[2], side));
    // ... then the enumeration on the main stream, behind the sort
    HIP_TRY(hipStreamWaitEvent(st, events.ev[2], 0));
    for (const SaEnumParams &w : launches)
      if ((e = sa_launch_sw_enumerate_window(w, st)) != hipSuccess) return fail_hip(e, "sw enumerate (window)");
    q.only_flagged = 1;
    if (unplaced) {
      // the few pairs no window takes are slow (one wave each): start them on the side stream, under the window
      // kernels; they read the sorted keys
      HIP_TRY(hipStreamWaitEvent(side, events.ev[1], 0));
      if ((e = sa_launch_sw_enumerate(q, side)) != hipSuccess) return fail_hip(e, "sw enumerate (generic, side stream)");
      HIP_TRY(hipEventRecord(events.ev[3], side));
      side_generic = true;
    }
  } else {
    HIP_TRY(hipEventRecord(events.ev[2], side));
    HIP_TRY(hipStreamWaitEvent(st, events.ev[2], 0));   // offsets / cleared status come from the side stream
  }
  tm.lap("sw: classes + enqueue enumeration");
  // per pair: hit count | string bytes used | status -- through pinned memory; the hit records themselves come
  // back later, packed (16 slots of 28 B per pair would be 4.5 MB for 10 000 pairs that have one hit each)
  if ((rc = ctx->h_tmeta.reserve(n * 12 + n * 16 + 64))) return rc;
  uint32_t *meta = ctx->h_tmeta.as<uint32_t>();
  if (generic_only && (e = sa_launch_sw_enumerate(q, st)) != hipSuccess) return fail_hip(e, "sw enumerate");
  if (side_generic) HIP_TRY(hipStreamWaitEvent(st, events.ev[3], 0));   // the side stream's pairs are done
  HIP_TRY(hipMemcpyAsync(meta, d_meta.p, n * 12, hipMemcpyDeviceToHost, st));
  if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // fill status; synchronises the stream
  tm.lap("sw: wait (enumeration), meta + hits");
  if (!generic_only) {
    // second phase, only when a pair was flagged: escaped walks run again in the largest window, what the window
    // kernels cannot take goes to the generic kernel; then the results are fetched again
    std::vector<uint32_t> again;
    bool any_generic = false;
    for (uint64_t k = 0; k < n; ++k) {
      if (meta[2 * n + k] & SA_ENUM_FALLBACK) again.push_back((uint32_t)k);
      any_generic |= (meta[2 * n + k] & SA_ENUM_GENERIC) != 0;
    }
    if (!again.empty() || any_generic) {
      if (!again.empty()) {
        HIP_TRY(hipMemcpyAsync(d_list.p, again.data(), again.size() * 4, hipMemcpyHostToDevice, st));
        SaEnumParams w = q;
        w.only_flagged = 0; w.retry = 1;
        w.pair_list = d_list.as<uint32_t>(); w.n_list = (uint32_t)again.size();
        SaEnumClass big[4];   // the configuration with the largest window LDS can hold at all
        sa_enum_classes(layout.key64, big);
        w.threads = big[3].threads; w.claim_bits = big[3].claim_bits; w.window_bytes = big[3].window_bytes;
        if ((e = sa_launch_sw_direction(w, st)) != hipSuccess || (e = sa_launch_sw_enumerate_window(w, st)) != hipSuccess)
          return fail_hip(e, "sw enumerate (window, retry)");
      }
      if ((e = sa_launch_sw_enumerate(q, st)) != hipSuccess) return fail_hip(e, "sw enumerate");   // only_flagged
      HIP_TRY(hipMemcpyAsync(meta, d_meta.p, n * 12, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
  }
  if (trace) {
    std::sort(need_of.begin(), need_of.end());
    fprintf(stderr, "[seqalign enum trace] window need (bytes): p10 %llu  p50 %llu  p90 %llu  p99 %llu  max %llu\n",
            (unsigned long long)need_of[n / 10], (unsigned long long)need_of[n / 2], (unsigned long long)need_of[n * 9 / 10],
            (unsigned long long)need_of[n * 99 / 100], (unsigned long long)need_of[n - 1]);
    std::vector<unsigned long long> t(16 * n);
    HIP_TRY(hipMemcpy(t.data(), d_trace, n * 128, hipMemcpyDeviceToHost));
    double load = 0, rounds_c = 0, iters = 0, rounds = 0, fallback = 0, cands = 0, ph[5] = {0, 0, 0, 0, 0};
    for (uint64_t k = 0; k < n; ++k) {
      load += (double)t[16 * k]; rounds_c += (double)t[16 * k + 1]; iters += (double)t[16 * k + 2]; rounds += (double)t[16 * k + 3];
      for (int q2 = 0; q2 < 5; ++q2) ph[q2] += (double)t[16 * k + 4 + q2];
      cands += count[k];
      fallback += t[16 * k + 3] == 0 && count[k] != 0;
    }
    fprintf(stderr, "[seqalign enum trace] cycles per pair by phase: hand-out+clear %.0f, claim inline %.0f, claim queue %.0f, "
                    "commit inline %.0f, commit queue+end %.0f\n", ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n);
    fprintf(stderr, "[seqalign enum trace] pairs %llu  candidates/pair %.0f  classes (pairs): %zu %zu %zu %zu  per pair: load %.0f cycles, "
                    "walks %.0f cycles, %.1f iterations  (pairs not finished by the window kernel: %.0f)\n",
            (unsigned long long)n, cands / n, members[0].size(), members[1].size(), members[2].size(), members[3].size(),
            load / n, rounds_c / n, iters / n, fallback);
    (void)rounds;
  }
  uint64_t gathered = 0, n_dev_hits = 0;
  uint64_t *hit_dst = reinterpret_cast<uint64_t *>(meta + 3 * n + (n & 1));   // pinned, behind meta (8-byte aligned)
  for (uint64_t k = 0; k < n; ++k) {
    const uint32_t status = meta[2 * n + k] & ~SA_ENUM_STOPPED_AT_MAX;
    if (status) return (status & (SA_ENUM_FALLBACK | SA_ENUM_GENERIC)) ? SEQALIGN_E_HIP : (int)status;
    dst_off[k] = gathered;
    gathered += meta[n + k];
    hit_dst[k] = n_dev_hits;
    n_dev_hits += meta[k];
  }
  // pairs that ran into the slot limit while the caller wants more: host enumeration with the full limit
  std::vector<uint64_t> capped;
  if (want_hits > max_hits)
    for (uint64_t k = 0; k < n; ++k)
      if ((meta[2 * n + k] & SA_ENUM_STOPPED_AT_MAX) && meta[k] >= max_hits) capped.push_back(k);
  std::vector<PairHits> redo(capped.size());
  std::vector<int64_t> redo_of(capped.empty() ? 0 : n, -1);
  if (!capped.empty()) {
    std::vector<uint64_t> m_off(capped.size() + 1, 0), c_off(capped.size() + 1, 0);
    for (size_t j = 0; j < capped.size(); ++j) {
      const uint64_t k = capped[j];
      redo_of[k] = (int64_t)j;
      m_off[j + 1] = m_off[j] + (cell0[k + 1] - cell0[k]);
      c_off[j + 1] = c_off[j] + count[k];
    }
    std::vector<int32_t> hM(m_off.back() + 1), hA(m_off.back() + 1), hB(m_off.back() + 1);
    std::vector<unsigned char> h_keys((c_off.back() + 1) * key_bytes);
    for (size_t j = 0; j < capped.size(); ++j) {
      const uint64_t k = capped[j], cells = cell0[k + 1] - cell0[k];
      HIP_TRY(hipMemcpyAsync(hM.data() + m_off[j], ctx->M.as<int32_t>() + cell0[k], cells * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(hA.data() + m_off[j], ctx->A.as<int32_t>() + cell0[k], cells * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(hB.data() + m_off[j], ctx->B.as<int32_t>() + cell0[k], cells * 4, hipMemcpyDeviceToHost, st));
      if (count[k])
        HIP_TRY(hipMemcpyAsync(h_keys.data() + c_off[j] * key_bytes, static_cast<const char *>(sorted) + cell0[k] * key_bytes,
                               count[k] * key_bytes, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    std::atomic<int> first_error{SEQALIGN_OK};
    parallel_for(capped.size(), [&](uint64_t j) {
      const uint64_t k = capped[j], p = c.first + k;
      sa_view_t v;
      v.sc = scoring; v.a = batch->arena + batch->off_a[p]; v.b = batch->arena + batch->off_b[p];
      v.len_a = batch->len_a[p]; v.len_b = batch->len_b[p];
      v.M = hM.data() + m_off[j]; v.A = hA.data() + m_off[j]; v.B = hB.data() + m_off[j];
      const uint64_t W = v.len_a + 1;
      std::vector<Cand> cand_list;
      cand_list.reserve(count[k]);
      for (uint64_t q2 = 0; q2 < count[k]; ++q2) {
        uint64_t key;
        if (layout.key64) memcpy(&key, h_keys.data() + (c_off[j] + q2) * 8, 8);
        else { uint32_t k32; memcpy(&k32, h_keys.data() + (c_off[j] + q2) * 4, 4); key = k32; }
        const uint64_t row = key & ((1ull << layout.row_bits) - 1), col = (key >> layout.row_bits) & ((1ull << layout.col_bits) - 1);
        const Cand cd{(uint32_t)(row * W + col), layout.cap - (int32_t)(uint32_t)(key >> (layout.row_bits + layout.col_bits))};
        if (cd.score >= min_score[p]) cand_list.push_back(cd);
      }
      const int prc = enumerate_hits(v, cand_list, want_hits, redo[j]);
      if (prc != SEQALIGN_OK) { int expected = SEQALIGN_OK; first_error.compare_exchange_strong(expected, prc); }
    });
    if ((rc = first_error.load())) return rc;
  }

  tm.lap("sw: second phase (if any)");
  // pack every pair's strings and hit records back to back and bring them over in one copy each
  DevBuf &d_hits_packed = ctx->t_meta, &d_hit_dst = ctx->t_str_off;
  if ((rc = d_gath_a.reserve(gathered + 16)) || (rc = d_gath_b.reserve(gathered + 16)) ||
      (rc = ctx->h_ta.reserve(gathered + 16)) || (rc = ctx->h_tb.reserve(gathered + 16)) ||
      (rc = d_hits_packed.reserve(n_dev_hits * sizeof(SaDevHit) + 16)) || (rc = d_hit_dst.reserve(n * 8 + 16)) ||
      (rc = ctx->h_misc.reserve(n_dev_hits * sizeof(SaDevHit) + 16)))
    return rc;
  uint64_t *dv_dst_off = d_offs.as<uint64_t>() + 2 * (n + 1);
  HIP_TRY(hipMemcpyAsync(dv_dst_off, dst_off, n * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_hit_dst.p, hit_dst, n * 8, hipMemcpyHostToDevice, st));
  if ((e = sa_launch_gather_strings(q.out_a, q.out_b, dv_str_off, q.str_used, dv_dst_off, d_gath_a.as<char>(),
                                    d_gath_b.as<char>(), q.hits, q.hit_count, d_hit_dst.as<uint64_t>(),
                                    d_hits_packed.as<SaDevHit>(), max_hits, (uint32_t)n, st)) != hipSuccess)
    return fail_hip(e, "gather strings");
  HIP_TRY(hipMemcpyAsync(ctx->h_ta.p, d_gath_a.p, gathered, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ctx->h_tb.p, d_gath_b.p, gathered, hipMemcpyDeviceToHost, st));
  if (n_dev_hits)
    HIP_TRY(hipMemcpyAsync(ctx->h_misc.p, d_hits_packed.p, n_dev_hits * sizeof(SaDevHit), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const SaDevHit *dev_hits = ctx->h_misc.as<SaDevHit>();

  tm.lap("sw: gather + strings D2H");
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
      const SaDevHit &src = dev_hits[hit_dst[k] + i];
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
  tm.lap("sw: unpack hits");
  return SEQALIGN_OK;
}

// Test hook (not part of include/seqalign_hip.h): the candidate keys of a batch that fits one chunk, after the
// fill's emission (or, emit_pass != 0, the separate pass over match_scores) and the per-pair sort.
// keys_out: one uint64 per matrix cell (pair p's sorted keys at its cell offset), count_out[n], box_out[4n],
// layout_out = {cap, row_bits, col_bits, score_bits, key64}.
extern "C" int sa_sw_candidates_debug(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                      const int32_t *min_score, int emit_pass, uint64_t *keys_out,
                                      uint32_t *count_out, uint32_t *box_out, int32_t *layout_out) {
  if (!ctx || !batch || !scoring || !min_score || !keys_out || !count_out || !box_out || !layout_out) return SEQALIGN_E_ARG;
  int rc = check_batch(batch);
  if (rc) return rc;
  const std::vector<Chunk> chunks = plan_chunks(batch, ctx->chunk_budget, 12 + 16);
  if (chunks.size() != 1) return SEQALIGN_E_ARG;
  const Chunk &c = chunks[0];
  HIP_TRY(hipSetDevice(ctx->device));
  ScoringGuard guard(ctx);
  if ((rc = seqalign_scoring_upload(ctx, scoring, 1, &guard.h))) return rc;
  const uint64_t n = c.count;
  int32_t thr = min_score[0];
  for (uint64_t k = 1; k < n; ++k) thr = std::min(thr, min_score[k]);
  const SaKeyLayout layout = key_layout(guard.h, c, thr);
  const size_t kb = layout.key64 ? 8 : 4;
  hipStream_t st = ctx->stream;
  StreamSyncOnExit sync_on_exit(st);
  DevBuf &d_min = ctx->e[0], &d_keys = ctx->e[1], &d_tmp = ctx->e[2], &d_box = ctx->e[3];
  if ((rc = d_min.reserve(n * 4)) || (rc = ctx->cand_count.reserve(n * 4)) || (rc = d_box.reserve(n * 16)) ||
      (rc = d_keys.reserve(c.cells * kb + 16)) || (rc = d_tmp.reserve(c.cells * kb + 16)))
    return rc;
  HIP_TRY(hipMemcpyAsync(d_min.p, min_score, n * 4, hipMemcpyHostToDevice, st));
  SaCandKeys cand;
  cand.keys = d_keys.p; cand.tmp = d_tmp.p; cand.cand_count = ctx->cand_count.as<uint32_t>();
  cand.cand_box = d_box.as<uint32_t>(); cand.cand_min = d_min.as<int32_t>(); cand.layout = layout;
  seqalign_dev_batch_t d;
  bool emitted = false;
  if ((rc = run_chunk(ctx, batch, c, guard.h, &d, nullptr, emit_pass ? nullptr : &cand, &emitted))) return rc;
  hipError_t e;
  if (!emitted) {
    SaReduceParams r;
    memset(&r, 0, sizeof(r));
    r.len_a = d.len_a; r.len_b = d.len_b; r.mat_off = d.mat_off; r.M = d.match_scores; r.n_pairs = (uint32_t)n;
    if ((e = sa_launch_sw_emit(r, cand, st)) != hipSuccess) return fail_hip(e, "sw candidate emission");
  }
  SaSortParams sp;
  memset(&sp, 0, sizeof(sp));
  sa_sort_plan(layout, &sp);
  sp.mat_off = d.mat_off; sp.cand_count = cand.cand_count; sp.keys = cand.keys; sp.tmp = cand.tmp; sp.n_pairs = (uint32_t)n;
  if ((e = sa_launch_sort_keys(sp, st)) != hipSuccess) return fail_hip(e, "candidate sort");
  const void *sorted = (sp.n_passes & 1) ? cand.tmp : cand.keys;
  std::vector<unsigned char> raw(c.cells * kb);
  HIP_TRY(hipMemcpyAsync(raw.data(), sorted, c.cells * kb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(count_out, cand.cand_count, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(box_out, cand.cand_box, n * 16, hipMemcpyDeviceToHost, st));
  if ((rc = fetch_status(ctx, c, nullptr))) return rc;
  for (uint64_t i = 0; i < c.cells; ++i) {
    if (layout.key64) memcpy(&keys_out[i], raw.data() + i * 8, 8);
    else { uint32_t k32; memcpy(&k32, raw.data() + i * 4, 4); keys_out[i] = k32; }
  }
  layout_out[0] = layout.cap; layout_out[1] = (int32_t)layout.row_bits; layout_out[2] = (int32_t)layout.col_bits;
  layout_out[3] = (int32_t)layout.score_bits; layout_out[4] = (int32_t)layout.key64;
  return SEQALIGN_OK;
}

// best hit of every pair of one chunk: fill (+ best cell) -> device traceback -> strings back
static int sw_chunk_best_hit(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c,
                             const seqalign_dev_scoring *sc, const int32_t *min_score, seqalign_sw_hit_t *hits,
                             uint64_t hit_cap, uint64_t *n_hits, uint64_t &found, char *out_a, char *out_b,
                             uint64_t str_cap, uint64_t &used_str) {
  int rc;
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
  return SEQALIGN_OK;
}

// every pair through the host: matrices and compacted candidates copied back, hits enumerated by host threads
static int sw_batch_host_enumeration(seqalign_ctx *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                     const seqalign_dev_scoring *sc, const int32_t *min_score, uint32_t max_hits,
                                     seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *n_hits, char *out_a,
                                     char *out_b, uint64_t str_cap) {
  int rc = SEQALIGN_OK;
  uint64_t used_str = 0, found = 0;
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
  if (max_hits == 1 && !traceback_on_host()) {   // best hit only: nothing but the strings crosses PCIe
    for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget))
      if ((rc = sw_chunk_best_hit(ctx, batch, c, sc, min_score, hits, hit_cap, n_hits, found, out_a, out_b, str_cap,
                                  used_str)))
        return rc;
    *n_hits = found;
    return SEQALIGN_OK;
  }
  if (!traceback_on_host()) {
    // up to kDeviceEnumMaxHits hits per pair on the device; a pair that needs more is finished on the host
    const uint32_t slots = std::min(max_hits, kDeviceEnumMaxHits);
    // per cell: the three matrices + two key buffers (8-byte keys assumed: the layout is per chunk)
    for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget, 12 + 2 * 8)) {
      if ((rc = sw_chunk_device_enumerate(ctx, batch, c, scoring, sc, min_score, slots, max_hits, hits, hit_cap,
                                          &found, out_a, out_b, str_cap, &used_str)))
        break;
    }
    *n_hits = found;
    return rc;
  }
  return sw_batch_host_enumeration(ctx, batch, scoring, sc, min_score, max_hits, hits, hit_cap, n_hits, out_a, out_b,
                                   str_cap);
}

