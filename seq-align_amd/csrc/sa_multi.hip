// sa_multi.hip -- the host-level calls over several contexts (GPUs) from one process.
#include "sa_ctx.hpp"

using namespace sa_host;

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

// n_ctx + 1 pair indices cutting the batch into contiguous ranges of (nearly) equal DP cells
// (balance ragged batches by W*H; contiguous, so results stay in pair order): edge k is the
// pair boundary nearest to k/n_ctx of the cell total, ties to the later boundary -- the rule of
// workloads.shard_edges_cells on the Python side.
std::vector<uint64_t> shard_edges(const seqalign_batch_t *b, int n_ctx) {
  const uint64_t n = b->n_pairs;
  std::vector<uint64_t> cum(n + 1, 0);
  for (uint64_t p = 0; p < n; ++p) cum[p + 1] = cum[p] + (uint64_t)(b->len_a[p] + 1ull) * (b->len_b[p] + 1ull);
  const uint64_t total = cum[n];
  std::vector<uint64_t> edges((size_t)n_ctx + 1, 0);
  for (int k = 1; k < n_ctx; ++k) {
    const uint64_t t = (uint64_t)(((unsigned __int128)total * (unsigned)k) / (unsigned)n_ctx);
    uint64_t i = (uint64_t)(std::lower_bound(cum.begin(), cum.end(), t) - cum.begin());
    if (i > n) i = n;
    if (i > 0 && t - cum[i - 1] < cum[i] - t) --i;
    edges[k] = std::max(i, edges[k - 1]);
  }
  edges[n_ctx] = n;
  return edges;
}

// run fn(g, first, count) for the n_ctx contiguous ranges, one host thread each; first error wins
template <class F>
int for_each_shard(const std::vector<uint64_t> &edges, F fn) {
  const int n_ctx = (int)edges.size() - 1;
  std::vector<int> rc((size_t)n_ctx, SEQALIGN_OK);
  std::vector<std::string> msg((size_t)n_ctx);
  std::vector<std::thread> th;
  for (int g = 0; g < n_ctx; ++g) {
    const uint64_t first = edges[g], last = edges[g + 1];
    th.emplace_back([&, g, first, last] {
      rc[g] = last > first ? fn(g, first, last - first) : SEQALIGN_OK;
      if (rc[g]) msg[g] = seqalign_last_error();   // the message lives in the worker's thread
    });
  }
  for (auto &t : th) t.join();
  for (int g = 0; g < n_ctx; ++g)
    if (rc[g]) { set_last_error(msg[g]); return rc[g]; }
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
  return for_each_shard(shard_edges(batch, n_ctx), [&](int g, uint64_t first, uint64_t count) {
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
  return for_each_shard(shard_edges(batch, n_ctx), [&](int g, uint64_t first, uint64_t count) {
    const seqalign_batch_t s = sub_batch(batch, first, count);   // str_off[] are absolute: shared string buffers
    return seqalign_nw_batch(ctxs[g], &s, scoring, str_off + first, out_a, out_b, out_len + first, out_score + first);
  });
}

extern "C" int seqalign_nw_batch_cigar_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch,
                                             const scoring_t *scoring, int format, const uint64_t *cigar_off, char *cigar,
                                             uint32_t *cigar_len, int32_t *out_score) {
  if (bad_ctx_list(ctxs, n_ctx) || !batch || !cigar_off || !cigar || !cigar_len || !out_score) return SEQALIGN_E_ARG;
  int rc = check_batch(batch);
  if (rc) return rc;
  return for_each_shard(shard_edges(batch, n_ctx), [&](int g, uint64_t first, uint64_t count) {
    const seqalign_batch_t s = sub_batch(batch, first, count);   // cigar_off[] are absolute: one shared buffer
    return seqalign_nw_batch_cigar(ctxs[g], &s, scoring, format, cigar_off + first, cigar, cigar_len + first, out_score + first);
  });
}

// format 0: the two strings (seqalign_sw_batch_multi); 1 / 2: CIGAR (seqalign_sw_batch_cigar_multi; out_b unused)
static int sw_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch,
                          const scoring_t *scoring, const int32_t *min_score, uint32_t max_hits, int format,
                          seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *n_hits, char *out_a,
                          char *out_b, uint64_t str_cap) {
  if (bad_ctx_list(ctxs, n_ctx) || !batch || !min_score || !hits || !n_hits || !out_a || (!format && !out_b)) return SEQALIGN_E_ARG;
  *n_hits = 0;
  int rc = check_batch(batch);
  if (rc) return rc;
  const uint64_t n = batch->n_pairs;
  if (n == 0) return SEQALIGN_OK;
  // every range writes into its own slice of the caller's buffers; the slices are closed up afterwards
  std::vector<uint64_t> h0((size_t)n_ctx + 1), s0((size_t)n_ctx + 1), got((size_t)n_ctx, 0), used((size_t)n_ctx, 0);
  const std::vector<uint64_t> edges = shard_edges(batch, n_ctx);
  for (int g = 0; g <= n_ctx; ++g) {
    const uint64_t first = edges[g];
    h0[g] = (uint64_t)((long double)hit_cap * first / n);
    s0[g] = (uint64_t)((long double)str_cap * first / n);
  }
  rc = for_each_shard(edges, [&](int g, uint64_t first, uint64_t count) {
    const seqalign_batch_t s = sub_batch(batch, first, count);
    uint64_t found = 0;
    const int r = format ? seqalign_sw_batch_cigar(ctxs[g], &s, scoring, min_score + first, max_hits, format, hits + h0[g], h0[g + 1] - h0[g],
                                                   &found, out_a + s0[g], s0[g + 1] - s0[g])
                         : seqalign_sw_batch(ctxs[g], &s, scoring, min_score + first, max_hits, hits + h0[g], h0[g + 1] - h0[g],
                                             &found, out_a + s0[g], out_b + s0[g], s0[g + 1] - s0[g]);
    got[g] = found;
    for (uint64_t i = 0; i < found; ++i) {
      const seqalign_sw_hit_t &h = hits[h0[g] + i];
      // (a hit's text: its columns, or in CIGAR mode the CIGAR -- the hits lie back to back, so only the last one's end matters)
      used[g] = std::max(used[g], h.str_off + (format ? strlen(out_a + s0[g] + h.str_off) : (size_t)h.length) + 1);
    }
    return r;
  });
  if (rc) return rc;
  uint64_t nh = 0, ns = 0;
  for (int g = 0; g < n_ctx; ++g) {
    const uint64_t first = edges[g];
    if (s0[g] != ns) {
      memmove(out_a + ns, out_a + s0[g], used[g]);
      if (!format) memmove(out_b + ns, out_b + s0[g], used[g]);
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

extern "C" int seqalign_sw_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch,
                                       const scoring_t *scoring, const int32_t *min_score, uint32_t max_hits,
                                       seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *n_hits, char *out_a,
                                       char *out_b, uint64_t str_cap) {
  return sw_batch_multi(ctxs, n_ctx, batch, scoring, min_score, max_hits, 0, hits, hit_cap, n_hits, out_a, out_b, str_cap);
}

extern "C" int seqalign_sw_batch_cigar_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch,
                                             const scoring_t *scoring, const int32_t *min_score, uint32_t max_hits, int format,
                                             seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *n_hits, char *cigar,
                                             uint64_t cigar_cap) {
  if (format != SEQALIGN_CIGAR_M && format != SEQALIGN_CIGAR_EQX) return SEQALIGN_E_ARG;
  return sw_batch_multi(ctxs, n_ctx, batch, scoring, min_score, max_hits, format, hits, hit_cap, n_hits, cigar, nullptr, cigar_cap);
}
