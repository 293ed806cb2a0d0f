// sa_batch_sw.hip -- seqalign_sw_batch: local alignment over a HOST batch (best hit on the device,
// multi-hit enumeration on the device -- the reverse sweep, sa_sw_sweep.hip -- or the literal host enumeration).
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


static uint32_t bits_for(uint64_t v) {   // bits needed to hold values 0..v
  uint32_t b = 1;
  while (b < 64 && (v >> b) != 0) ++b;
  return b;
}

// the most one move of an alignment can add to its score (a substitution; a gap only when a gap score is positive --
// legal, absurd)
static int64_t best_move(const seqalign_dev_scoring *sc) {
  const sa_flat_scoring_t &f = sc->flat;
  int64_t best = std::max<int64_t>(1, std::max(f.gen_eq, f.gen_ne));
  for (uint64_t k = 0; k < (uint64_t)f.n_classes * f.n_classes; ++k)
    if (f.table[k] != SA_S_BLOCKED && f.table[k] != SA_S_UNKNOWN) best = std::max<int64_t>(best, f.table[k]);
  return std::max<int64_t>(best, std::max(f.ext, f.open1));
}

// uint64 elements of the multi-hit path's scratch arena a pair needs: the candidate columns of its rows, then its
// hits' keys.  A hit is a walk from a cell with score >= min_score down to score 0, every move takes off at most
// best_move(), and the cells it stands on are won by it alone: ceil(min_score / best move) + 1 cells per hit.
static uint64_t hit_arena_elements(uint32_t len_a, uint32_t len_b, int32_t min_score, int64_t best) {
  const uint64_t cells = ((uint64_t)len_a + 1) * ((uint64_t)len_b + 1);
  const uint64_t per_hit = (uint64_t)((std::max<int64_t>(min_score, 1) + best - 1) / best) + 1;
  return (uint64_t)len_b + 1 + cells / per_hit + 1;
}

// How the sweep packs a cell into a key (SaKeyLayout): row and column fields sized by the longest sequences, the
// score field by the largest score the scoring can produce on them.
static SaKeyLayout key_layout(const seqalign_dev_scoring *sc, uint32_t max_a, uint32_t max_b) {
  const sa_flat_scoring_t &f = sc->flat;
  const int64_t best_step = best_move(sc);
  int64_t cap = (int64_t)std::min(max_a, max_b) * best_step;
  if (f.ext > 0 || f.open1 > 0) cap = ((int64_t)max_a + max_b) * best_step;
  cap = std::min<int64_t>(std::max<int64_t>(cap, 1), INT32_MAX);
  SaKeyLayout l;
  l.cap = (int32_t)cap;
  l.row_bits = bits_for(max_b);
  l.col_bits = bits_for(max_a);
  l.score_bits = bits_for((uint64_t)cap);
  return l;
}
// a key is at most 64 bits wide, all ones excluded
static bool key_layout_fits(const SaKeyLayout &l) { return l.row_bits + l.col_bits + l.score_bits <= 63; }

// SW hits of one chunk, enumerated on the device:
//   fill (reports the candidates' count, box and columns per row) -> reverse sweep (sa_sw_sweep.hip: every hit's
//   key, in order)
//   -> one traceback per wanted hit -> strings packed -> D2H.
// Host round trips: the hit counts (they size the traceback), the hits' lengths (they size the packing), the strings.
// Appends to the caller's hit array / string buffers.
static int sw_chunk_device_enumerate(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c,
                                     const seqalign_dev_scoring *sc, const int32_t *min_score, uint32_t max_hits,
                                     seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *found,
                                     char *out_a, char *out_b, uint64_t str_cap, uint64_t *used_str) {
  const uint64_t n = c.count;
  hipStream_t st = ctx->stream;
  int rc;
  DevBuf &d_min = ctx->e[0], &d_keys = ctx->e[1], &d_box = ctx->e[3], &d_rows = ctx->e[4],
         &d_rowoff = ctx->e[5], &d_walk = ctx->e[6], &d_hits = ctx->e[7], &d_meta = ctx->e[8],
         &d_gath_a = ctx->e[9], &d_dst = ctx->e[11];
  // sources of asynchronous H2D copies: declared BEFORE the guard below, so that they are destroyed after its
  // hipStreamSynchronize on every exit path
  std::vector<uint64_t> hit_off(n + 1, 0), row_off, walk_str, dst_off;
  std::vector<uint32_t> walk_pair, walk_rank;
  StreamSyncOnExit sync_on_exit(st);
  StageTimer tm(ctx->opt.timing);
  const SaKeyLayout layout = key_layout(sc, c.max_a, c.max_b);
  if (!key_layout_fits(layout)) return SEQALIGN_E_TOO_LARGE;   // (seqalign_sw_batch checks the whole batch first)

  // ---- fill + candidates' count and box
  // every pair's part of the scratch arena (its rows' candidate columns, then its hits' keys)
  {
    const int64_t best = best_move(sc);
    uint32_t pa = ~0u, pb = ~0u; int32_t pm = 0; uint64_t pe = 0;   // (reads of one length: the division once per run of equal pairs)
    for (uint64_t k = 0; k < n; ++k) {
      const uint32_t xa = batch->len_a[c.first + k], xb = batch->len_b[c.first + k];
      const int32_t xm = min_score[c.first + k];
      if (xa != pa || xb != pb || xm != pm) { pa = xa; pb = xb; pm = xm; pe = hit_arena_elements(xa, xb, xm, best); }
      hit_off[k + 1] = hit_off[k] + pe;
    }
  }
  DevBuf &d_hitoff = ctx->e[13];
  if ((rc = d_min.reserve(n * 4)) || (rc = ctx->cand_count.reserve(n * 4)) || (rc = d_box.reserve(n * 16)) ||
      (rc = d_keys.reserve(hit_off[n] * 8 + 16)) || (rc = d_meta.reserve(n * 16 + 16)) || (rc = d_hitoff.reserve((n + 1) * 8)))
    return rc;
  // (from pinned staging: a copy from the caller's pageable array is staged by the runtime and waited for)
  if ((rc = ctx->h_tb.reserve((n + 1) * 8 + n * 4 + 16))) return rc;
  { uint64_t *hs_off = ctx->h_tb.as<uint64_t>();
    int32_t *hs_min = reinterpret_cast<int32_t *>(hs_off + n + 1);
    memcpy(hs_off, hit_off.data(), (n + 1) * 8); memcpy(hs_min, min_score + c.first, n * 4);
    HIP_TRY(hipMemcpyAsync(d_hitoff.p, hs_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_min.p, hs_min, n * 4, hipMemcpyHostToDevice, st)); }
  SaCandBox cand{};
  cand.cand_count = ctx->cand_count.as<uint32_t>(); cand.cand_box = d_box.as<uint32_t>(); cand.cand_min = d_min.as<int32_t>();
  cand.cand_rows = d_keys.as<uint32_t>(); cand.hit_off = d_hitoff.as<uint64_t>();
  // plain scorings, rows up to 1 024 columns (513 and up: see wide_ok below): the fill writes match_scores + one byte of directions per cell instead of the
  // three matrices (sa_fill_dirs.hip); the directions go where gap_a_scores would have gone
  bool dirs_used = false;
  // every pair of the chunk the same shape (reads against windows of one length), match / mismatch scoring: the packed
  // two-pairs-per-wave fill (sa_fill_dirs_x2.hip); every pair's cells then start on a multiple of 256
  uint64_t stride = 0;
  {
    bool same_shape = true;
    for (uint64_t k = 1; k < n && same_shape; ++k)
      same_shape = batch->len_a[c.first + k] == batch->len_a[c.first] && batch->len_b[c.first + k] == batch->len_b[c.first];
    if (same_shape && (n >= kPackedFillMinPairs || ctx->opt.pack16 == 2) && sw_dirs_x2_applicable(ctx, sc, c.max_a, c.max_b))
      stride = (((uint64_t)(c.max_a + 1ull) * (c.max_b + 1ull)) + 255u) & ~(uint64_t)255u;
    else if (!same_shape && (n >= kBucketedFillMinPairs || ctx->opt.pack16 == 2) && ((uint64_t)c.max_a + 1) * ((uint64_t)c.max_b + 1) <= kShapeTableMax &&
             sw_dirs_x2_applicable(ctx, sc, c.max_a, c.max_b))
      stride = kBucketShapes;   // ragged: run_chunk pairs up the pairs of equal shape, every pair on a multiple of 256 cells
  }
  uint64_t cells256 = 0;
  if (stride == kBucketShapes)
    for (uint64_t k = 0; k < n; ++k) cells256 += (((uint64_t)batch->len_a[c.first + k] + 1) * ((uint64_t)batch->len_b[c.first + k] + 1) + 255u) & ~(uint64_t)255u;
  // (only when the sweep will run in its rows-in-registers form: not with the strip / LDS forms forced by an option)
  // and rows of 513 .. 1 024 columns only where the sweep's one-word form takes them (sa_launch_sw_sweep) and a wave per pair fills
  // the chip (few wide pairs: strips, below, as before)
  const bool wide_row = c.max_a + 1 > 512;
  const bool wide_ok = ctx->opt.sweep_ev && layout.row_bits + layout.col_bits + layout.score_bits <= 62 &&
                       (n >= (c.max_a + 1 > 768 ? 640u : 128u) || ctx->opt.sweep_mode == 1);
  // (tools/sw_wide_few.py, 700 x 1000 -- 12 columns per lane: 128 pairs 4.04 -> 3.69 ms, 64: 3.51 -> 3.67; 900 x 1000 -- 16 per lane:
  //  512 pairs 5.8 on strips / 6.3 here, 1 024 pairs 8.1 / 6.2)
  const bool allow_dirs = ctx->opt.sweep_mode != 2 && ctx->opt.sweep_cpl == 0 && (!wide_row || wide_ok);
  // (the packed direction-byte fill is bound by instruction issue: its arenas need no placement walk)
  if ((rc = reserve_arenas(ctx, (stride == kBucketShapes ? cells256 : stride ? n * stride : c.cells) * 4, !(allow_dirs && stride != 0)))) return rc;
  cand.dirs = allow_dirs ? ctx->A.as<uint8_t>() : nullptr; cand.dirs_used = &dirs_used;
  if (!allow_dirs) stride = 0;
  cand.uniform_stride = stride == kBucketShapes ? 256 : stride;
  // ---- ONE trip for everything, when the call is the common kind: direction bytes, one wave per pair in the sweep, a few
  // hits per pair wanted.  The hit walks are launched BEFORE anybody has seen the sweep's counts -- max_hits walks per pair,
  // walk w = hit w % max_hits of pair w / max_hits, those beyond a pair's hits return at once (sa_traceback.hip) -- and send
  // home two bits per column into fixed slots of pinned memory (host/sa_moves.c), so nothing has to be sized, gathered or
  // copied by a second and third launch + wait: counts, per-walk words and moves are all there after one synchronisation,
  // and the host threads expand the hits straight into the caller's buffers.  (Round 3: counts -> walker lists -> walks ->
  // lengths -> gather -> strings, three waits.)  Sweep and walks follow the chunk's last fill, in the fills' stream.
  // (Measured, profiles/r04/r04_sw_pipeline.txt: on a second stream, with equal slices so that a slice's sweep runs beside
  // the next slice's fill -- both issue vector instructions 50-70 % of the time, neither is bound by memory -- the two
  // kernels simply share the SIMDs: C3's sweep of 4 096 pairs 0.62 -> 1.97 ms beside a fill, the call 4.6 -> 5.4 ms; and a
  // sweep on the second, high-priority stream alone behind an event starts 80 us late and runs 0.93 instead of 0.80 ms on C4.)
  // A pair that needs the host between sweep and walks -- more than 64 hits (keys unsorted), an error to weigh against
  // max_hits -- sends the whole chunk down the three-trip path below, which starts from the same counts.
  const bool trace = ctx->opt.sweep_trace;   // development aid: per-pair counters on stderr
  DevBuf &d_trace = ctx->e[12];
  const bool may_one_trip = allow_dirs && ctx->opt.sweep_mode == 0 && !trace && ctx->opt.nw_moves && max_hits <= 8 &&
                            n * max_hits <= ((uint64_t)4 << 20) && c.max_a + 1 <= 1024;
  SaSweepParams q;
  memset(&q, 0, sizeof(q));
  auto sweep_params = [&](const seqalign_dev_batch_t &dd, uint64_t k0, uint64_t k1, bool dirs) {
    SaSweepParams r;
    memset(&r, 0, sizeof(r));
    r.arena = dd.arena; r.off_a = dd.off_a; r.len_a = dd.len_a; r.off_b = dd.off_b; r.len_b = dd.len_b;
    r.mat_off = dd.mat_off; r.M = dd.match_scores; r.A = dd.gap_a_scores; r.B = dd.gap_b_scores;
    r.code = sc->d_code; r.table = sc->d_table; r.cand_count = cand.cand_count + k0; r.cand_box = cand.cand_box + 4 * k0;
    r.min_score = d_min.as<int32_t>() + k0; r.hit_keys = d_keys.as<unsigned long long>(); r.hit_off = d_hitoff.as<uint64_t>() + k0;
    r.err_key = d_meta.as<unsigned long long>() + k0;
    r.hit_count = reinterpret_cast<uint32_t *>(d_meta.as<unsigned long long>() + n) + k0; r.status = r.hit_count + n;
    r.n_pairs = (uint32_t)(k1 - k0); r.K = sc->flat.n_classes; r.open1 = sc->flat.open1; r.ext = sc->flat.ext;
    r.gen_eq = sc->flat.gen_eq; r.gen_ne = sc->flat.gen_ne; r.flags = sc->flat.flags;
    r.max_len_a = c.max_a; r.max_len_b = c.max_b; r.layout = layout; r.tune_cpl = ctx->opt.sweep_cpl; r.tune_ev = ctx->opt.sweep_ev;
    r.dirs = dirs ? cand.dirs : nullptr;
    return r;
  };
  hipError_t e;
  bool piped = may_one_trip, piped_any = false;
  const uint64_t nwalk = n * max_hits;
  if (may_one_trip) {
    const uint64_t move_words = 2ull * max_hits * ((c.seq_bytes >> 5) + n) + 2;
    if ((rc = ctx->h_ta.reserve(move_words * 4)) || (rc = ctx->h_B.reserve(nwalk * 16 + 16)) || (rc = ctx->h_tmeta.reserve(n * 16 + 64)))   // (h_misc is fetch_status')
      return rc;
  }
  auto after_fill = [&](uint64_t k0, uint64_t k1, const seqalign_dev_batch_t &dd, bool dirs, bool cand_reported) -> int {
    if (!piped || !dirs || !cand_reported) { piped = false; return SEQALIGN_OK; }   // not the common kind: everything below, for the whole chunk
    hipError_t he;
    const SaSweepParams r = sweep_params(dd, k0, k1, true);
    if ((he = sa_launch_sw_sweep(r, st)) != hipSuccess) return fail_hip(he, "sw sweep");
    SaTraceParams t;
    memset(&t, 0, sizeof(t));
    t.arena = dd.arena; t.off_a = dd.off_a; t.len_a = dd.len_a; t.off_b = dd.off_b; t.len_b = dd.len_b; t.mat_off = dd.mat_off;
    t.M = dd.match_scores; t.code = sc->d_code; t.table = sc->d_table;
    t.str_off = dd.off_a;                       // (run_chunk packs a, b, a, b, ...: off_a IS the prefix of len_a + len_b)
    // walk w of the launch is hit w % max_hits of the slice's pair w / max_hits = the chunk's pair k0 + that: its slot is
    // 2 max_hits ((off_a >> 5) + chunk pair) + ..., so the bases move by the slice's first pair
    t.stage_words = (c.max_a + c.max_b + 31u) >> 5;
    t.moves = ctx->h_ta.dev_as<uint32_t>() + 2ull * max_hits * k0; t.out_meta4 = ctx->h_B.dev_as<uint32_t>() + 4ull * max_hits * k0;
    t.walks_per_pair = max_hits; t.hit_count = r.hit_count; t.sweep_status = r.status;
    t.hit_keys = r.hit_keys; t.hit_off = r.hit_off; t.layout = layout; t.fill_status = dd.status;
    t.n_pairs = (uint32_t)((k1 - k0) * max_hits); t.K = r.K; t.open1 = r.open1; t.ext = r.ext; t.gen_eq = r.gen_eq; t.gen_ne = r.gen_ne;
    t.flags = r.flags; t.tune_walker = ctx->opt.trace_kernel; t.tune_group = ctx->opt.walk_group; t.dirs = r.dirs;
    if ((he = sa_launch_nw_traceback(t, st)) != hipSuccess) return fail_hip(he, "sw hit traceback");
    piped_any = true;
    return SEQALIGN_OK;
  };
  seqalign_dev_batch_t d;
  bool reported = false;
  if ((rc = run_chunk(ctx, batch, c, sc, &d, nullptr, &cand, &reported, stride))) return rc;
  // (behind ALL the fills: per slice -- a small first slice's sweep and walks, then the rest's -- C3 took 5.2 instead of 4.7 ms)
  if (may_one_trip && (rc = after_fill(0, n, d, dirs_used, reported))) return rc;
  const bool strips_needed = !dirs_used && c.max_a + 1 > 512 && (c.max_a + 1 > SA_SWEEP_LDS_COLUMNS || n < 1024);
  bool strips = false;
  if (piped && piped_any) {
    // everything is enqueued on the one stream: one wait for all of it
    q = sweep_params(d, 0, n, true);
    tm.lap("sw: fills, sweeps, walks enqueued");
  } else {
  if (!reported) {   // a fill kernel that cannot report them itself: one pass over match_scores
    SaReduceParams r;
    memset(&r, 0, sizeof(r));
    r.len_a = d.len_a; r.len_b = d.len_b; r.mat_off = d.mat_off; r.M = d.match_scores; r.n_pairs = (uint32_t)n;
    if ((e = sa_launch_sw_box(r, cand, c.max_b, st)) != hipSuccess) return fail_hip(e, "sw candidate box");
  }

  // ---- the sweep: every hit of every pair
  q = sweep_params(d, 0, n, dirs_used);
  // How the pairs are laid over waves (sa_sw_sweep.hip): one wave per pair -- rows in registers (up to 512 columns on three matrices, up to 1 024 on direction bytes),
  // wider ones in segments with the winners of two rows in LDS -- or, for FEW wide pairs (a wave per pair would leave
  // the chip empty) and for rows too wide for LDS, one wave per 256-column strip.  The option sweep_mode = strips | pair
  // forces one (tests, experiments).
  const int mode_opt = ctx->opt.sweep_mode;   // 0 auto, 1 pair, 2 strips
  const uint32_t w_max = c.max_a + 1;
  strips = strips_needed;
  if (mode_opt == 2) strips = true;
  if (mode_opt == 1 && w_max <= SA_SWEEP_LDS_COLUMNS) strips = false;
  if (w_max <= SA_SWEEP_LDS_COLUMNS) q.lds_columns = (c.max_a + 2u) & ~1u;
  DevBuf &d_prog = ctx->e[2];
  if (strips) {
    // strip width by how many waves that makes: 128-column strips (and progress published every 16 rows instead of
    // 64) when the pairs are so few that 256-column strips would leave most of the chip idle (2 x 10 000^2: 42.7 ->
    // 32.8 ms; 64-column strips gain nothing more: a row's dependent passes cost ~2 us whatever its width)
    uint32_t cols = sa_sweep_strip_blocks((uint32_t)n, c.max_a, 256) < 1024 ? 128 : 256;
    if (const uint32_t v = ctx->opt.sweep_strip; v == 64 || v == 128 || v == 256) cols = v;
    q.strip_columns = cols; q.strip_interval = cols == 256 ? 64 : 16;
    const uint32_t spp = sa_sweep_strips_per_pair(c.max_a, cols);
    const uint64_t blocks = sa_sweep_strip_blocks((uint32_t)n, c.max_a, cols);
    row_off.resize(n);
    uint64_t rows_total = 0;
    for (uint64_t k = 0; k < n; ++k) { row_off[k] = rows_total; rows_total += (uint64_t)batch->len_b[c.first + k] + 1; }
    if ((rc = d_prog.reserve((2 * blocks + 1) * 4 + 16)) || (rc = d_rows.reserve(rows_total * spp * 16 + 16)) ||
        (rc = d_rowoff.reserve(n * 8)))
      return rc;
    HIP_TRY(hipMemsetAsync(d_prog.p, 0, (2 * blocks + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(q.err_key, 0xff, n * 8, st));     // the strips of a pair report into the same words
    HIP_TRY(hipMemsetAsync(q.hit_count, 0, n * 8, st));      // hit_count | status
    HIP_TRY(hipMemcpyAsync(d_rowoff.p, row_off.data(), n * 8, hipMemcpyHostToDevice, st));
    q.strip_progress = d_prog.as<uint32_t>(); q.strips_per_pair = spp;
    q.bnd = d_rows.as<unsigned long long>(); q.row_off = d_rowoff.as<uint64_t>();
  }
  if (trace) {
    if ((rc = d_trace.reserve(n * 64))) return rc;
    HIP_TRY(hipMemsetAsync(d_trace.p, 0, n * 64, st));
    q.trace = d_trace.as<unsigned long long>();
  }
  if ((e = sa_launch_sw_sweep(q, st)) != hipSuccess) return fail_hip(e, "sw sweep");
  tm.lap("sw: enqueue fill + sweep");
  }

  const bool one_trip = piped && piped_any;
  if (one_trip) {
    unsigned long long *h_err = ctx->h_tmeta.as<unsigned long long>();
    const uint32_t *h_cnt = reinterpret_cast<const uint32_t *>(h_err + n), *h_st = h_cnt + n;
    HIP_TRY(hipMemcpyAsync(h_err, d_meta.p, n * 16, hipMemcpyDeviceToHost, st));
    if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // the fill's status words; synchronises the stream
    tm.lap("sw one trip: wait (fill, sweep, walks)");
    bool clean = true;
    for (uint64_t k = 0; k < n && clean; ++k) clean = h_st[k] == 0;
    tm.lap("sw one trip: statuses checked");
    if (clean) {
      const uint32_t *h_w = ctx->h_B.as<uint32_t>(), *h_moves = ctx->h_ta.as<uint32_t>();
      // where every hit goes in the caller's buffers: hits and string bytes per block of pairs (parallel), a prefix over the
      // blocks, then every block places and expands its own hits
      constexpr uint64_t kBlk = 64;    // (pairs per task: C3's 10 000 pairs are 157 tasks for 16-32 threads; 256 left some threads a third more than others)
      const uint64_t nblk = (n + kBlk - 1) / kBlk;
      std::vector<uint64_t> blk_hits(nblk + 1, 0), blk_bytes(nblk + 1, 0);
      std::atomic<int> bad{SEQALIGN_OK};
      // where hit j of pair k lies: its planes, and what it takes in the caller's string buffer -- its columns + NUL, or in CIGAR
      // mode (seqalign_sw_batch_cigar) its CIGAR's length + NUL, counted from the planes (host/sa_moves.c)
      auto planes_of = [&](uint64_t k, uint32_t j, uint32_t *nwd_out) {
        const uint64_t pp = c.first + k;
        const uint32_t nwd = (batch->len_a[pp] + batch->len_b[pp] + 31u) >> 5;
        *nwd_out = nwd;
        return h_moves + 2ull * max_hits * ((ctx->h_desc.as<uint64_t>()[k] >> 5) + k) + 2ull * j * nwd;
      };
      auto out_bytes = [&](uint64_t k, uint32_t j, uint32_t len) -> uint64_t {
        if (!ctx->cigar_format) return (uint64_t)len + 1;
        const uint64_t pp = c.first + k, w = k * max_hits + j;
        uint32_t nwd, pos[4], clen = 0;
        const uint32_t *pa = planes_of(k, j, &nwd);
        if (sa_cigar_sw_moves(batch->arena + batch->off_a[pp], batch->arena + batch->off_b[pp], h_w[4 * w + 2], h_w[4 * w + 3], pa, pa + nwd,
                              nwd, len, ctx->cigar_format, ctx->cigar_fold, nullptr, 0, pos, &clen))
          return 1;   // (the placement pass meets the same error and reports it)
        return (uint64_t)clen + 1;
      };
      parallel_for(nblk, [&](uint64_t bi) {
        uint64_t hs = 0, bytes = 0;
        for (uint64_t k = bi * kBlk, e2 = std::min(n, (bi + 1) * kBlk); k < e2; ++k) {
          const uint32_t take = std::min(h_cnt[k], max_hits);
          for (uint32_t j = 0; j < take; ++j) {
            const uint32_t len = h_w[4 * (k * max_hits + j) + 1];
            if (len >= SA_MOVES_ERR) { int expected = SEQALIGN_OK; bad.compare_exchange_strong(expected, (int)(len & 15u)); continue; }
            ++hs; bytes += out_bytes(k, j, len);
          }
        }
        blk_hits[bi + 1] = hs; blk_bytes[bi + 1] = bytes;
      });
      if (bad.load()) return bad.load();
      tm.lap("sw one trip: hits counted");
      for (uint64_t bi = 0; bi < nblk; ++bi) { blk_hits[bi + 1] += blk_hits[bi]; blk_bytes[bi + 1] += blk_bytes[bi]; }
      // what fits the caller's buffers: whole blocks while they fit, the first one that does not hit by hit (reported after
      // what fits is delivered, as the three-trip path does)
      const uint64_t hit_room = hit_cap > *found ? hit_cap - *found : 0, str_room = str_cap > *used_str ? str_cap - *used_str : 0;
      bool no_room = blk_hits[nblk] > hit_room || blk_bytes[nblk] > str_room;
      const uint64_t first_hit = *found, first_str = *used_str;
      std::atomic<uint64_t> delivered_hits{0}, delivered_bytes{0};
      parallel_for(nblk, [&](uint64_t bi) {
        uint64_t hi = blk_hits[bi], at = blk_bytes[bi], done_h = 0, done_b = 0;
        bool stop = false;
        for (uint64_t k = bi * kBlk, e2 = std::min(n, (bi + 1) * kBlk); k < e2 && !stop; ++k) {
          const uint64_t pp = c.first + k;
          const uint32_t take = std::min(h_cnt[k], max_hits), la = batch->len_a[pp], lb = batch->len_b[pp], nwd = (la + lb + 31u) >> 5;
          // (run_chunk's packed offsets: the prefix of len_a + len_b over the chunk's pairs, in the pinned descriptor block)
          const uint64_t slot = ctx->h_desc.as<uint64_t>()[k];
          if (k + 6 < e2) {   // the GPU wrote these lines: every first touch is a miss -- have the pair six ahead on its way
            const uint64_t kn = k + 6, sn = ctx->h_desc.as<uint64_t>()[kn];
            const uint32_t nwn = (batch->len_a[c.first + kn] + batch->len_b[c.first + kn] + 31u) >> 5;
            const uint32_t *pn = h_moves + 2ull * max_hits * ((sn >> 5) + kn);
            __builtin_prefetch(h_w + 4 * kn * max_hits); __builtin_prefetch(pn + nwn - 1); __builtin_prefetch(pn + 2 * nwn - 1);
            __builtin_prefetch(batch->arena + batch->off_a[c.first + kn]);
          }
          for (uint32_t j = 0; j < take; ++j, ++hi) {
            const uint64_t w = k * max_hits + j;
            const uint32_t len = h_w[4 * w + 1];
            const uint64_t need = out_bytes(k, j, len);
            if (hi >= hit_room || at + need > str_room) { stop = true; break; }   // (no_room is set: the call reports it)
            const uint32_t *pa = h_moves + 2ull * max_hits * ((slot >> 5) + k) + 2ull * j * nwd;
            seqalign_sw_hit_t &h = hits[first_hit + hi];
            uint32_t pos[4], clen = 0;
            const int prc = ctx->cigar_format
              ? sa_cigar_sw_moves(batch->arena + batch->off_a[pp], batch->arena + batch->off_b[pp], h_w[4 * w + 2], h_w[4 * w + 3], pa, pa + nwd,
                                  nwd, len, ctx->cigar_format, ctx->cigar_fold, out_a + first_str + at, need, pos, &clen)
              : sa_expand_sw_moves(batch->arena + batch->off_a[pp], batch->arena + batch->off_b[pp], h_w[4 * w + 2], h_w[4 * w + 3],
                                   pa, pa + nwd, nwd, len, out_a + first_str + at, out_b + first_str + at, pos);
            if (prc) { int expected = SEQALIGN_OK; bad.compare_exchange_strong(expected, prc); stop = true; break; }
            h.pair = pp; h.score = (int32_t)h_w[4 * w]; h.pos_a = pos[0]; h.pos_b = pos[1]; h.len_a = pos[2]; h.len_b = pos[3];
            h.length = len; h.str_off = first_str + at;
            at += need; ++done_h; done_b += need;
          }
        }
        delivered_hits.fetch_add(done_h); delivered_bytes.fetch_add(done_b);
      });
      const uint64_t n_out = no_room ? std::min<uint64_t>(delivered_hits.load(), hit_room) : blk_hits[nblk];
      *used_str += no_room ? delivered_bytes.load() : blk_bytes[nblk];
      if (bad.load()) return bad.load();
      *found += n_out;
      tm.lap("sw one trip: hits expanded");
      return no_room ? SEQALIGN_E_NOMEM : SEQALIGN_OK;
    }
  }

  // ---- round trip 1: hit counts and status
  if ((rc = ctx->h_tmeta.reserve(n * 16 + 64))) return rc;
  unsigned long long *h_err_key = ctx->h_tmeta.as<unsigned long long>();
  const uint32_t *h_count = reinterpret_cast<const uint32_t *>(h_err_key + n), *h_status = h_count + n;
  HIP_TRY(hipMemcpyAsync(h_err_key, d_meta.p, n * 16, hipMemcpyDeviceToHost, st));
  if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // fill status; synchronises the stream
  tm.lap("sw: wait (fill + sweep), counts");
  if (trace) {
    std::vector<unsigned long long> t(8 * n);
    HIP_TRY(hipMemcpyAsync(t.data(), d_trace.p, n * 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(stream_wait_spinning(st));
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hits_total = 0;
    for (uint64_t k = 0; k < n; ++k) {
      for (int j = 0; j < 8; ++j) sum[j] += (double)t[8 * k + j];
      hits_total += h_count[k];
    }
    fprintf(stderr, "[seqalign sweep trace] pairs %llu  per pair: %.0f cycles (s_memtime), %.1f rows, %.1f active row segments, "
                    "%.1f passes, %.0f cycles in active segments, %.2f hits; wide pairs, per row: prefetch issue %.0f, segments %.0f, fence + rotate %.0f cycles\n",
            (unsigned long long)n, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, hits_total / n, sum[5] / n, sum[6] / n, sum[7] / n);
  }

  uint64_t str_total = 0;
  bool overflow = false;
  std::vector<unsigned long long> big;
  for (uint64_t k = 0; k < n && !overflow; ++k) {
    const uint32_t cnt = h_count[k], take = std::min(cnt, max_hits);
    unsigned long long *dev_keys = d_keys.as<unsigned long long>() + hit_off[k] + batch->len_b[c.first + k] + 1;
    if (h_status[k] & SA_SWEEP_UNSORTED) {   // more than 64 hits in one pair: ordered here (rare; that pair's keys only)
      // (on the context's stream, waited for both ways: the kernels that wrote the keys and the walkers that read them
      // run on that stream, and it is a non-blocking one -- a null-stream hipMemcpy would not be ordered with it)
      big.resize(cnt);
      HIP_TRY(hipMemcpyAsync(big.data(), dev_keys, (size_t)cnt * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(stream_wait_spinning(st));
      std::sort(big.begin(), big.end());
      HIP_TRY(hipMemcpyAsync(dev_keys, big.data(), (size_t)cnt * 8, hipMemcpyHostToDevice, st));
      HIP_TRY(stream_wait_spinning(st));
    }
    if (h_status[k] & SA_SWEEP_OVERFLOW) {
      set_last_error("seqalign_sw_batch: internal error: pair " + std::to_string(c.first + k) + " has more hits than its share of the scratch arena");
      return SEQALIGN_E_HIP;
    }
    if (const uint32_t err = h_status[k] & ~SA_SWEEP_UNSORTED) {
      // a walk met an error.  The reference would have met it too unless it had stopped before: max_hits hits
      // found among the walks in front of this one
      bool moot = false;
      if (cnt >= max_hits) {
        unsigned long long last;
        HIP_TRY(hipMemcpyAsync(&last, dev_keys + (max_hits - 1), 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(stream_wait_spinning(st));
        moot = last < h_err_key[k];
      }
      if (!moot) return (int)err;
    }
    const uint64_t slot = (uint64_t)batch->len_a[c.first + k] + batch->len_b[c.first + k];
    for (uint32_t j = 0; j < take; ++j) {
      if (*found + walk_pair.size() >= hit_cap) { overflow = true; break; }   // reported after what fits is delivered
      walk_pair.push_back((uint32_t)k); walk_rank.push_back(j); walk_str.push_back(str_total);
      str_total += slot;
    }
  }
  const uint64_t nw = walk_pair.size();
  if (nw > 0xffffffffull) return SEQALIGN_E_TOO_LARGE;

  // ---- one traceback per hit (smith_waterman.c:217-255; the traceback kernels of sa_traceback.hip, one walk per hit)
  // per walk: head | len | score | status | pos[4]
  if ((rc = d_walk.reserve(nw * 16 + 16)) || (rc = d_hits.reserve(nw * 32 + 16)) ||
      (rc = ctx->t_out_a.reserve(str_total + 16)) || (rc = ctx->t_out_b.reserve(str_total + 16)) ||
      (rc = ctx->h_misc.reserve(nw * 32 + 16)) || (rc = d_dst.reserve(nw * 8 + 16)))
    return rc;
  uint64_t *dv_walk_str = d_walk.as<uint64_t>();
  uint32_t *dv_walk_pair = reinterpret_cast<uint32_t *>(dv_walk_str + nw), *dv_walk_rank = dv_walk_pair + nw;
  uint32_t *dv_meta = d_hits.as<uint32_t>();
  const uint32_t *h_meta = ctx->h_misc.as<uint32_t>();
  if (nw) {
    // the three lists as the one block they are on the device, from pinned memory: one copy instead of three staged ones
    if ((rc = ctx->h_desc.reserve(nw * 16 + 16))) return rc;
    { uint64_t *hw_str = ctx->h_desc.as<uint64_t>();
      uint32_t *hw_pair = reinterpret_cast<uint32_t *>(hw_str + nw), *hw_rank = hw_pair + nw;
      memcpy(hw_str, walk_str.data(), nw * 8); memcpy(hw_pair, walk_pair.data(), nw * 4); memcpy(hw_rank, walk_rank.data(), nw * 4);
      HIP_TRY(hipMemcpyAsync(dv_walk_str, hw_str, nw * 16, hipMemcpyHostToDevice, st)); }
    SaTraceParams t;
    memset(&t, 0, sizeof(t));
    t.arena = d.arena; t.off_a = d.off_a; t.len_a = d.len_a; t.off_b = d.off_b; t.len_b = d.len_b;
    t.mat_off = d.mat_off; t.M = d.match_scores; t.A = d.gap_a_scores; t.B = d.gap_b_scores;
    t.code = sc->d_code; t.table = sc->d_table; t.str_off = dv_walk_str;
    t.out_a = ctx->t_out_a.as<char>(); t.out_b = ctx->t_out_b.as<char>();
    t.out_head = dv_meta; t.out_len = dv_meta + nw; t.out_score = reinterpret_cast<int32_t *>(dv_meta + 2 * nw);
    t.trace_status = dv_meta + 3 * nw; t.out_pos = dv_meta + 4 * nw;
    t.walker_pair = dv_walk_pair; t.walker_rank = dv_walk_rank; t.hit_keys = q.hit_keys; t.hit_off = q.hit_off; t.layout = layout;
    t.n_pairs = (uint32_t)nw; t.K = q.K; t.open1 = q.open1; t.ext = q.ext; t.gen_eq = q.gen_eq; t.gen_ne = q.gen_ne;
    t.flags = q.flags; t.tune_walker = ctx->opt.trace_kernel; t.tune_group = ctx->opt.walk_group; t.dirs = q.dirs;
    if ((e = sa_launch_nw_traceback(t, st)) != hipSuccess) return fail_hip(e, "sw hit traceback");
    // ---- round trip 2: the hits (their lengths size the packing)
    HIP_TRY(hipMemcpyAsync(ctx->h_misc.p, dv_meta, nw * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(stream_wait_spinning(st));
  }
  tm.lap("sw: hit tracebacks");
  dst_off.resize(nw);
  uint64_t gathered = 0;
  for (uint64_t w = 0; w < nw; ++w) {
    if (h_meta[3 * nw + w]) return (int)h_meta[3 * nw + w];
    dst_off[w] = gathered;
    gathered += h_meta[nw + w];
  }
  const uint64_t gath_b_at = (gathered + 15) & ~(uint64_t)15;   // a-strings, then b-strings: one buffer, one copy home
  if (nw) {
    if ((rc = d_gath_a.reserve(gath_b_at + gathered + 16)) || (rc = ctx->h_ta.reserve(gath_b_at + gathered + 16)) ||
        (rc = ctx->h_desc.reserve(nw * 16 + 16)))
      return rc;
    memcpy(ctx->h_desc.p, dst_off.data(), nw * 8);   // (pinned; the walker lists it held are on the device by now: the stream was waited for)
    HIP_TRY(hipMemcpyAsync(d_dst.p, ctx->h_desc.p, nw * 8, hipMemcpyHostToDevice, st));
    if ((e = sa_launch_gather_hits(ctx->t_out_a.as<char>(), ctx->t_out_b.as<char>(), dv_walk_str, dv_meta, dv_meta + nw,
                                   d_dst.as<uint64_t>(), d_gath_a.as<char>(), d_gath_a.as<char>() + gath_b_at, (uint32_t)nw, st)) != hipSuccess)
      return fail_hip(e, "gather hits");
    HIP_TRY(hipMemcpyAsync(ctx->h_ta.p, d_gath_a.p, gath_b_at + gathered, hipMemcpyDeviceToHost, st));
    HIP_TRY(stream_wait_spinning(st));
  }
  tm.lap("sw: gather + strings D2H");
  const char *ha = ctx->h_ta.as<char>(), *hb = ha + gath_b_at;
  // where every hit goes in the caller's buffers (a prefix over the lengths), then the copies on the thread pool
  uint64_t n_out = 0;
  bool no_room = false;
  std::vector<uint64_t> out_off(nw);
  for (uint64_t w = 0; w < nw; ++w) {
    const uint32_t len = h_meta[nw + w];
    if (*found + n_out >= hit_cap) { no_room = true; break; }
    if (ctx->cigar_format) {   // CIGAR mode on a path whose walkers write strings: encode them, one after the other (not the common kind)
      const uint64_t took = put_alignment(ctx, ha + dst_off[w], hb + dst_off[w], len, out_a, out_b, *used_str, str_cap > *used_str ? str_cap - *used_str : 0);
      if (!took) { no_room = true; break; }
      out_off[w] = *used_str;
      *used_str += took;
      ++n_out;
      continue;
    }
    if (*used_str + len + 1 > str_cap) { no_room = true; break; }
    out_off[w] = *used_str;
    *used_str += len + 1;
    ++n_out;
  }
  const uint64_t first_hit = *found;
  constexpr uint64_t kPack = 1024;
  parallel_for((n_out + kPack - 1) / kPack, [&](uint64_t blk) {
    for (uint64_t w = blk * kPack, e2 = std::min(n_out, (blk + 1) * kPack); w < e2; ++w) {
      const uint32_t len = h_meta[nw + w], *pos = h_meta + 4 * nw + 4 * w;
      if (!ctx->cigar_format) {
        memcpy(out_a + out_off[w], ha + dst_off[w], len);
        memcpy(out_b + out_off[w], hb + dst_off[w], len);
        out_a[out_off[w] + len] = out_b[out_off[w] + len] = '\0';
      }
      seqalign_sw_hit_t &h = hits[first_hit + w];
      h.pair = c.first + walk_pair[w]; h.score = reinterpret_cast<const int32_t *>(h_meta)[2 * nw + w];
      h.pos_a = pos[0]; h.pos_b = pos[1]; h.len_a = pos[2]; h.len_b = pos[3]; h.length = len; h.str_off = out_off[w];
    }
  });
  *found += n_out;
  if (no_room) return SEQALIGN_E_NOMEM;
  tm.lap("sw: unpack hits");
  return overflow ? SEQALIGN_E_NOMEM : SEQALIGN_OK;
}

// best hit of every pair of one chunk: fill (+ best cell) -> device traceback -> strings back
static int sw_chunk_best_hit(seqalign_ctx *ctx, const seqalign_batch_t *batch, const Chunk &c,
                             const seqalign_dev_scoring *sc, const int32_t *min_score, seqalign_sw_hit_t *hits,
                             uint64_t hit_cap, uint64_t *n_hits, uint64_t &found, char *out_a, char *out_b,
                             uint64_t str_cap, uint64_t &used_str) {
  int rc;
  seqalign_dev_batch_t d;
  bool have_best = false;   // the stream kernel reports the best cell itself
  const uint64_t n = c.count;
  StageTimer tm(ctx->opt.timing);
  // Every pair of the chunk the same shape, match / mismatch scoring, scores inside int16: the fill writes only a byte of
  // directions per cell and finds the best cell itself, two pairs per wave (sa_fill_dirs_x2.hip: fill_sw_best_x2_kernel);
  // the walk follows the bytes.  No match_scores, no gap matrices: they are not even allocated.
  bool dirs_used = false;
  uint64_t stride = 0;
  // (the walks below as tile walks sending home moves: the fill writes the direction byte's LOCAL form, sa_kernels.h)
  // (whatever the chunk's size: 26 000-40 000 pairs of configs[2]'s / [3]'s shape are 0-9 % faster with tile walks on the local form than with
  //  lane walks on the older one: profiles/r06/r06_local_dirs.txt)
  const bool local = ctx->opt.dirs_local && ctx->opt.nw_moves && (ctx->opt.trace_kernel ? ctx->opt.trace_kernel == 2 : true);
  {
    bool same_shape = true;
    for (uint64_t k = 1; k < n && same_shape; ++k)
      same_shape = batch->len_a[c.first + k] == batch->len_a[c.first] && batch->len_b[c.first + k] == batch->len_b[c.first];
    // From how many pairs: what this fill competes with below that is not a one-pair direction fill (the NW / multi-hit paths'
    // kPackedFillMinPairs) but three matrices and their walker.  tools/sw_best_few.py, same process, packed against not:
    // BLOSUM62 300 x 300 -- ahead at every size (128 pairs 0.52 -> 0.46 ms, 2 047 pairs 0.92 -> 0.58); match / mismatch, 150 x 1 000 --
    // level at ~1 400 pairs (1 024: 0.68 -> 0.76, 2 047: 0.99 -> 0.83); 700 x 1 000 -- at ~700 (512: 1.63 -> 1.94, 1 024: 2.25 -> 2.00,
    // 2 047: 4.10 -> 2.11): a wave's row costs more with two pairs in it, which is what a launch too small to fill the chip pays.
    const uint64_t best_min_pairs = sc->flat.n_classes > 1 ? 128 : c.max_a + 1 > 512 ? 1024 : 1536;
    if (same_shape && (n >= best_min_pairs || ctx->opt.pack16 == 2) && sw_best_x2_applicable(ctx, sc, c.max_a, c.max_b))
      stride = ((sa_dirs_blocked_shape(c.max_a) ? sa_dirs_blocked_bytes(c.max_a, c.max_b) : (uint64_t)(c.max_a + 1ull) * (c.max_b + 1ull)) + 255u) & ~(uint64_t)255u;
    else if (!same_shape && (n >= kBucketedFillMinPairs || ctx->opt.pack16 == 2) && ((uint64_t)c.max_a + 1) * ((uint64_t)c.max_b + 1) <= kShapeTableMax &&
             sw_best_x2_applicable(ctx, sc, c.max_a, c.max_b))
      stride = kBucketShapes;   // ragged: run_chunk pairs up the pairs of equal shape
  }
  if (stride) {
    uint64_t dir_bytes = n * stride;
    if (stride == kBucketShapes) {
      dir_bytes = 0;
      for (uint64_t k = 0; k < n; ++k) {
        const uint32_t xa = batch->len_a[c.first + k], xb = batch->len_b[c.first + k];
        dir_bytes += ((sa_dirs_blocked_shape(c.max_a) ? sa_dirs_blocked_bytes(xa, xb) : ((uint64_t)xa + 1) * ((uint64_t)xb + 1)) + 255u) & ~(uint64_t)255u;
      }
    }
    if ((rc = ctx->dirs.reserve(dir_bytes + 4096))) return rc;
    SaCandBox bc;
    memset(&bc, 0, sizeof(bc));
    bc.dirs = ctx->dirs.as<uint8_t>(); bc.dirs_used = &dirs_used; bc.best_only = true; bc.dirs_local = local; bc.uniform_stride = stride == kBucketShapes ? 256 : stride;
    if ((rc = run_chunk(ctx, batch, c, sc, &d, &have_best, &bc, nullptr, stride))) return rc;
    if (!dirs_used || !have_best) { set_last_error("seqalign_sw_batch: internal error: the best-hit direction fill did not run"); return SEQALIGN_E_HIP; }
  } else if ((rc = run_chunk(ctx, batch, c, sc, &d, &have_best))) {
    return rc;
  }
  if (!have_best) {
    if ((rc = ctx->best_score.reserve(n * 4)) || (rc = ctx->best_index.reserve(n * 8))) return rc;
    SaReduceParams r;
    memset(&r, 0, sizeof(r));
    r.n_pairs = (uint32_t)n; r.len_a = d.len_a; r.len_b = d.len_b; r.mat_off = d.mat_off; r.M = d.match_scores;
    r.min_score = 1; r.best_score = ctx->best_score.as<int32_t>(); r.best_index = ctx->best_index.as<uint64_t>();
    // few long pairs: several waves per pair (a wave streams ~4 GB/s on its own)
    const uint64_t max_cells = ((uint64_t)c.max_a + 1) * ((uint64_t)c.max_b + 1);
    if (n < 2048) r.slices = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(max_cells >> 18, 1), 4096);
    hipError_t e = sa_launch_sw_reduce(r, ctx->stream);
    if (e != hipSuccess) return fail_hip(e, "sw reduce launch");
  }
  if ((rc = ctx->h_tmeta.reserve(n * 8 + n * 32))) return rc;
  uint64_t *h_off = ctx->h_tmeta.as<uint64_t>();
  uint64_t total = 0;
  for (uint64_t k = 0; k < n; ++k) {
    h_off[k] = total;
    total += (uint64_t)batch->len_a[c.first + k] + batch->len_b[c.first + k];
  }
  if ((rc = ctx->t_str_off.reserve(n * 8)) || (rc = ctx->t_out_a.reserve(total + 16)) ||
      (rc = ctx->t_out_b.reserve(total + 16)) || (rc = ctx->t_meta.reserve(n * 32)))
    return rc;
  hipStream_t st = ctx->stream;
  std::vector<uint64_t> dst_off(n);    // (source of an asynchronous copy: declared before the guard, destroyed after its sync)
  StreamSyncOnExit sync_on_exit(st);
  HIP_TRY(hipMemcpyAsync(ctx->t_str_off.p, h_off, n * 8, hipMemcpyHostToDevice, st));
  uint32_t *d_meta = ctx->t_meta.as<uint32_t>();   // head | len | score | status | pos[4]
  seqalign_trace_t t;
  memset(&t, 0, sizeof(t));
  t.str_off = ctx->t_str_off.as<uint64_t>(); t.out_a = ctx->t_out_a.as<char>(); t.out_b = ctx->t_out_b.as<char>();
  t.out_head = d_meta; t.out_len = d_meta + n; t.out_score = reinterpret_cast<int32_t *>(d_meta + 2 * n);
  t.status = d_meta + 3 * n; t.out_pos = d_meta + 4 * n; t.start_index = ctx->best_index.as<uint64_t>();
  if (dirs_used && ctx->opt.nw_moves) {
    // the walk sends home two bits per column into the pair's slot of pinned memory, the host expands (as seqalign_nw_batch):
    // one launch, one wait -- no lengths to fetch before a gather, no gather, no strings over PCIe
    const uint64_t move_words = 2ull * ((total >> 5) + n) + 2;
    if ((rc = ctx->h_ta.reserve(move_words * 4)) || (rc = ctx->h_B.reserve(n * 16 + 16))) return rc;   // (h_misc is fetch_status')
    SaTraceParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.arena = d.arena; tp.off_a = d.off_a; tp.len_a = d.len_a; tp.off_b = d.off_b; tp.len_b = d.len_b; tp.mat_off = d.mat_off;
    tp.code = sc->d_code; tp.table = sc->d_table; tp.str_off = d.off_a;
    tp.stage_words = (c.max_a + c.max_b + 31u) >> 5;
    tp.moves = ctx->h_ta.dev_as<uint32_t>(); tp.out_meta4 = ctx->h_B.dev_as<uint32_t>();
    tp.start_index = ctx->best_index.as<uint64_t>(); tp.start_score = ctx->best_score.as<int32_t>();
    tp.dirs = ctx->dirs.as<uint8_t>(); tp.dirs_blocked = sa_dirs_blocked_shape(c.max_a); tp.dirs_local = local; tp.tune_stage = ctx->opt.walk_stage; tp.tune_tile = ctx->opt.walk_tile; tp.fill_status = d.status;
    tp.n_pairs = (uint32_t)n; tp.K = sc->flat.n_classes; tp.open1 = sc->flat.open1; tp.ext = sc->flat.ext;
    tp.gen_eq = sc->flat.gen_eq; tp.gen_ne = sc->flat.gen_ne; tp.flags = sc->flat.flags; tp.tune_walker = ctx->opt.trace_kernel; tp.tune_group = ctx->opt.walk_group;
    hipError_t e2 = sa_launch_nw_traceback(tp, st);
    if (e2 != hipSuccess) return fail_hip(e2, "sw best-hit traceback");
    tm.lap("sw best hit: fill + walk enqueued");
    // (no fetch_status here: a direction fill is only admitted for scorings in which every character pair has a score, and the
    // walks carry a pair's fill status home in their own word anyway -- the copy and its scan were 20 us of nothing)
    HIP_TRY(stream_wait_spinning(st));
    tm.lap("sw best hit: wait (upload, fill, walk)");
    const uint32_t *h_w = ctx->h_B.as<uint32_t>(), *h_moves = ctx->h_ta.as<uint32_t>();
    // Where every hit goes in the caller's buffers: hits and string bytes per block of pairs (parallel -- round 4 counted in one
    // serial pass over words the GPU had just written, every line of them a miss: 0.1 ms of C4's 0.89 ms call), a prefix over
    // the blocks, then every block places and expands its own hits.
    constexpr uint64_t kBlk = 64;
    const uint64_t nblk = (n + kBlk - 1) / kBlk;
    std::vector<uint64_t> blk_hits(nblk + 1, 0), blk_bytes(nblk + 1, 0);
    std::atomic<uint64_t> first_err{~0ull};   // pair << 8 | code of the LOWEST failing pair
    auto is_hit = [&](uint64_t k) { const int32_t sc_ = (int32_t)h_w[4 * k]; return sc_ > 0 && sc_ >= min_score[c.first + k]; };
    // what pair k's hit takes in the caller's string buffer: its columns + NUL, or (seqalign_sw_batch_cigar) its CIGAR + NUL, counted from the planes
    auto out_bytes = [&](uint64_t k, uint32_t len) -> uint64_t {
      if (!ctx->cigar_format) return (uint64_t)len + 1;
      const uint64_t p = c.first + k;
      const uint32_t nwd = (batch->len_a[p] + batch->len_b[p] + 31u) >> 5;
      const uint32_t *pa = h_moves + 2ull * ((h_off[k] >> 5) + k);
      uint32_t pos[4], clen = 0;
      if (sa_cigar_sw_moves(batch->arena + batch->off_a[p], batch->arena + batch->off_b[p], h_w[4 * k + 2], h_w[4 * k + 3], pa, pa + nwd, nwd, len,
                            ctx->cigar_format, ctx->cigar_fold, nullptr, 0, pos, &clen))
        return 1;   // (the placement pass meets the same error and reports it)
      return (uint64_t)clen + 1;
    };
    parallel_for(nblk, [&](uint64_t bi) {
      uint64_t hs = 0, bytes = 0;
      for (uint64_t k = bi * kBlk, e3 = std::min(n, (bi + 1) * kBlk); k < e3; ++k) {
        const uint32_t len = h_w[4 * k + 1];
        if (len >= SA_MOVES_ERR) {
          uint64_t seen = first_err.load(std::memory_order_relaxed);
          const uint64_t mine = k << 8 | (len & 15u);
          while (mine < seen && !first_err.compare_exchange_weak(seen, mine, std::memory_order_relaxed)) {}
          continue;
        }
        if (is_hit(k)) { ++hs; bytes += out_bytes(k, len); }
      }
      blk_hits[bi + 1] = hs; blk_bytes[bi + 1] = bytes;
    });
    if (first_err.load() != ~0ull) return (int)(first_err.load() & 255u);
    for (uint64_t bi = 0; bi < nblk; ++bi) { blk_hits[bi + 1] += blk_hits[bi]; blk_bytes[bi + 1] += blk_bytes[bi]; }
    // what fits: whole blocks while they do, the first that does not pair by pair; the call reports SEQALIGN_E_NOMEM AFTER
    // delivering what fits (as the string path below and the one-trip multi-hit path do)
    const uint64_t hit_room = hit_cap > found ? hit_cap - found : 0, str_room = str_cap > used_str ? str_cap - used_str : 0;
    const bool out_of_room = blk_hits[nblk] > hit_room || blk_bytes[nblk] > str_room;
    const uint64_t first_hit = found, first_str = used_str;
    std::atomic<uint64_t> delivered_hits{0}, delivered_bytes{0};
    std::atomic<int> bad{SEQALIGN_OK};
    parallel_for(nblk, [&](uint64_t bi) {
      uint64_t hi = blk_hits[bi], at = blk_bytes[bi], done_h = 0, done_b = 0;
      if (hi >= hit_room || at >= str_room) return;   // (everything from here on is beyond the caller's room)
      for (uint64_t k = bi * kBlk, e3 = std::min(n, (bi + 1) * kBlk); k < e3; ++k) {
        if (k + 6 < e3) {   // the GPU wrote these lines: every first touch is a miss -- have the pair six ahead on its way
          const uint64_t kn = k + 6;
          const uint32_t nwn = (batch->len_a[c.first + kn] + batch->len_b[c.first + kn] + 31u) >> 5;
          const uint32_t *pn = h_moves + 2ull * ((h_off[kn] >> 5) + kn);
          __builtin_prefetch(h_w + 4 * kn); __builtin_prefetch(pn + nwn - 1); __builtin_prefetch(pn + 2 * nwn - 1);
          __builtin_prefetch(batch->arena + batch->off_a[c.first + kn]);
        }
        if (!is_hit(k)) continue;
        const uint64_t p = c.first + k;
        const uint32_t len = h_w[4 * k + 1];
        const uint64_t need = out_bytes(k, len);
        if (hi >= hit_room || at + need > str_room) break;   // (out_of_room is set: the call reports it)
        const uint32_t la = batch->len_a[p], lb = batch->len_b[p], nwd = (la + lb + 31u) >> 5;
        const uint32_t *pa = h_moves + 2ull * ((h_off[k] >> 5) + k);
        uint32_t pos[4], clen = 0;
        const int prc = ctx->cigar_format
          ? sa_cigar_sw_moves(batch->arena + batch->off_a[p], batch->arena + batch->off_b[p], h_w[4 * k + 2], h_w[4 * k + 3], pa, pa + nwd, nwd, len,
                              ctx->cigar_format, ctx->cigar_fold, out_a + first_str + at, need, pos, &clen)
          : sa_expand_sw_moves(batch->arena + batch->off_a[p], batch->arena + batch->off_b[p], h_w[4 * k + 2], h_w[4 * k + 3],
                               pa, pa + nwd, nwd, len, out_a + first_str + at, out_b + first_str + at, pos);
        if (prc) { int expected = SEQALIGN_OK; bad.compare_exchange_strong(expected, prc); break; }
        seqalign_sw_hit_t &h = hits[first_hit + hi];
        h.pair = p; h.score = (int32_t)h_w[4 * k]; h.pos_a = pos[0]; h.pos_b = pos[1]; h.len_a = pos[2]; h.len_b = pos[3];
        h.length = len; h.str_off = first_str + at;
        ++hi; at += need; ++done_h; done_b += need;
      }
      delivered_hits.fetch_add(done_h, std::memory_order_relaxed); delivered_bytes.fetch_add(done_b, std::memory_order_relaxed);
    });
    const uint64_t n_out = delivered_hits.load();
    used_str += delivered_bytes.load();
    tm.lap("sw best hit: hits collected + expanded");
    if (bad.load()) return bad.load();
    found += n_out;
    if (out_of_room) { *n_hits = found; return SEQALIGN_E_NOMEM; }
    return SEQALIGN_OK;
  }
  if (dirs_used) rc = sw_traceback_dirs(ctx, sc, &d, &t, ctx->dirs.as<uint8_t>(), ctx->best_score.as<int32_t>(), st);
  else rc = seqalign_sw_traceback_device(ctx, sc, &d, &t, st);
  if (rc) return rc;
  uint32_t *h_meta = reinterpret_cast<uint32_t *>(h_off + n);
  HIP_TRY(hipMemcpyAsync(h_meta, d_meta, n * 32, hipMemcpyDeviceToHost, st));
  if ((rc = fetch_status(ctx, c, nullptr))) return rc;   // syncs
  // a hit is much shorter than its slot of len_a + len_b characters: pack the strings on the device, bring back
  // what was written (C3: 2 x 1.7 MB instead of 2 x 11.5 MB over PCIe)
  DevBuf &d_dst = ctx->e[11], &d_gath_a = ctx->e[9];
  uint64_t gathered = 0;
  for (uint64_t k = 0; k < n; ++k) {
    if (h_meta[3 * n + k]) return (int)h_meta[3 * n + k];
    dst_off[k] = gathered;
    gathered += h_meta[n + k];
  }
  const uint64_t gath_b_at = (gathered + 15) & ~(uint64_t)15;   // a-strings, then b-strings: one buffer, one copy home
  if ((rc = d_dst.reserve(n * 8 + 16)) || (rc = d_gath_a.reserve(gath_b_at + gathered + 16)) ||
      (rc = ctx->h_ta.reserve(gath_b_at + gathered + 16)) || (rc = ctx->h_desc.reserve(n * 8 + 16)))
    return rc;
  memcpy(ctx->h_desc.p, dst_off.data(), n * 8);   // (pinned; run_chunk's descriptors left it with the stream's last wait)
  HIP_TRY(hipMemcpyAsync(d_dst.p, ctx->h_desc.p, n * 8, hipMemcpyHostToDevice, st));
  hipError_t e = sa_launch_gather_hits(ctx->t_out_a.as<char>(), ctx->t_out_b.as<char>(), ctx->t_str_off.as<uint64_t>(), d_meta,
                                       d_meta + n, d_dst.as<uint64_t>(), d_gath_a.as<char>(), d_gath_a.as<char>() + gath_b_at, (uint32_t)n, st);
  if (e != hipSuccess) return fail_hip(e, "gather hits");
  HIP_TRY(hipMemcpyAsync(ctx->h_ta.p, d_gath_a.p, gath_b_at + gathered, hipMemcpyDeviceToHost, st));
  HIP_TRY(stream_wait_spinning(st));
  const char *ha = ctx->h_ta.as<char>(), *hb = ha + gath_b_at;
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t p = c.first + k;
    const uint32_t len = h_meta[n + k];
    const int32_t score = reinterpret_cast<const int32_t *>(h_meta)[2 * n + k];
    if (score <= 0 || score < min_score[p]) continue;
    const uint64_t took = found >= hit_cap ? 0 : put_alignment(ctx, ha + dst_off[k], hb + dst_off[k], len, out_a, out_b, used_str, str_cap > used_str ? str_cap - used_str : 0);
    if (!took) { *n_hits = found; return SEQALIGN_E_NOMEM; }
    seqalign_sw_hit_t &h = hits[found++];
    const uint32_t *pos = h_meta + 4 * n + 4 * k;
    h.pair = p; h.score = score; h.pos_a = pos[0]; h.pos_b = pos[1]; h.len_a = pos[2]; h.len_b = pos[3];
    h.length = len; h.str_off = used_str;
    used_str += took;
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
    // sources / targets of asynchronous copies: declared before the guard, so that they outlive its
    // hipStreamSynchronize on every way out of this iteration
    std::vector<uint64_t> c_off;
    std::vector<uint32_t> c_cap, h_cidx;
    std::vector<int32_t> h_cscore;
    StreamSyncOnExit chunk_sync(ctx->stream);
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
    HIP_TRY(stream_wait_spinning(ctx->stream));
    c_off.resize(n);
    c_cap.assign(h_count, h_count + n);
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
    h_cidx.resize(total + 1);
    h_cscore.resize(total + 1);
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
        const uint64_t took = found >= hit_cap ? 0 : put_alignment(ctx, ph.str_a.data() + src.str_off, ph.str_b.data() + src.str_off, src.length, out_a, out_b,
                                                                   used_str, str_cap > used_str ? str_cap - used_str : 0);
        if (!took) { rc = SEQALIGN_E_NOMEM; break; }
        seqalign_sw_hit_t &h = hits[found++];
        h = src;
        h.pair = c.first + k;
        h.str_off = used_str;
        used_str += took;
      }
    }
    if (rc) break;
  }
  *n_hits = found;
  return rc;
}

static int sw_batch_impl(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                         const int32_t *min_score, uint32_t max_hits, seqalign_sw_hit_t *hits,
                         uint64_t hit_cap, uint64_t *n_hits, char *out_a, char *out_b, uint64_t str_cap) {
  *n_hits = 0;
  int rc = check_batch(batch);
  if (rc) return rc;
  if (batch->n_pairs == 0) return SEQALIGN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  seqalign_dev_scoring *sc = nullptr;
  if ((rc = cached_scoring(ctx, scoring, 1, &sc))) return rc;
  uint64_t used_str = 0, found = 0;
  if (max_hits == 0) return SEQALIGN_OK;
  if (max_hits == 1 && !traceback_on_host(ctx)) {   // best hit only: nothing but the strings crosses PCIe
    for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget))
      if ((rc = sw_chunk_best_hit(ctx, batch, c, sc, min_score, hits, hit_cap, n_hits, found, out_a, out_b, str_cap,
                                  used_str)))
        return rc;
    *n_hits = found;
    return SEQALIGN_OK;
  }
  uint32_t max_a = 0, max_b = 0;
  for (uint64_t p = 0; p < batch->n_pairs; ++p) { max_a = std::max(max_a, batch->len_a[p]); max_b = std::max(max_b, batch->len_b[p]); }
  // (a key that does not fit the sweep's records -- scores beyond 2^28 on sequences beyond 2^16 -- goes to the host)
  if (!traceback_on_host(ctx) && key_layout_fits(key_layout(sc, max_a, max_b))) {
    // a chunk holds its pairs' three matrices + each pair's OWN part of the scratch arena (rows' candidate columns, hits'
    // keys: a pair with a low min_score needs more of it than the batch's average, so it is counted per pair)
    std::vector<uint64_t> extra(batch->n_pairs);
    { const int64_t best = best_move(sc);
      for (uint64_t p = 0; p < batch->n_pairs; ++p)
        extra[p] = 8 * hit_arena_elements(batch->len_a[p], batch->len_b[p], min_score[p], best) + 64;
    }
    for (const Chunk &c : plan_chunks(batch, ctx->chunk_budget, 12, extra.data())) {
      if ((rc = sw_chunk_device_enumerate(ctx, batch, c, sc, min_score, max_hits, hits, hit_cap, &found, out_a, out_b,
                                          str_cap, &used_str)))
        break;
    }
    *n_hits = found;
    return rc;
  }
  return sw_batch_host_enumeration(ctx, batch, scoring, sc, min_score, max_hits, hits, hit_cap, n_hits, out_a, out_b,
                                   str_cap);
}

extern "C" int seqalign_sw_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                 const int32_t *min_score, uint32_t max_hits, seqalign_sw_hit_t *hits,
                                 uint64_t hit_cap, uint64_t *n_hits, char *out_a, char *out_b, uint64_t str_cap) {
  if (!ctx || !scoring || !min_score || !hits || !n_hits || !out_a || !out_b) return SEQALIGN_E_ARG;
  CallScope scope(ctx);
  return sw_batch_impl(ctx, batch, scoring, min_score, max_hits, hits, hit_cap, n_hits, out_a, out_b, str_cap);
}

// Local hits as CIGAR (include/seqalign_hip.h): the same call with one text per hit instead of two.  On the direction-byte paths
// (the one-trip multi-hit call, the packed best-hit call) the walks come home as bit planes and the CIGAR is run-length encoded
// from those, its exact length counted first so that the hits still lie back to back in `cigar`; the paths whose walkers write
// strings (three matrices, the host enumeration) encode those strings (put_alignment).
extern "C" int seqalign_sw_batch_cigar(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                       const int32_t *min_score, uint32_t max_hits, int format, seqalign_sw_hit_t *hits,
                                       uint64_t hit_cap, uint64_t *n_hits, char *cigar, uint64_t cigar_cap) {
  if (!ctx || !scoring || !min_score || !hits || !n_hits || !cigar || (format != SEQALIGN_CIGAR_M && format != SEQALIGN_CIGAR_EQX))
    return SEQALIGN_E_ARG;
  CallScope scope(ctx);
  CigarScope mode(ctx, format, !scoring->case_sensitive);
  return sw_batch_impl(ctx, batch, scoring, min_score, max_hits, hits, hit_cap, n_hits, cigar, cigar, cigar_cap);
}
