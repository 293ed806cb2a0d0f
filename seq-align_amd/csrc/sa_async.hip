// sa_async.hip -- asynchronous host-level calls: seqalign_{nw,sw}_batch_submit + seqalign_job_wait.
//
// A host-level call on one batch is a serial chain -- pack, upload, fill, walk, results home, expansion -- and on BASELINE
// configs[1] the kernels are only 55 % of it (0.25 of 0.45 ms); two rounds of overlap INSIDE a call found nothing left to overlap
// (profiles/r05/r05_experiments.txt).  ACROSS calls the chain overlaps trivially: batch k + 1's packing and upload need nothing of
// batch k's walk and expansion.  The reference has one pair in flight at a time (src/alignment_cmdline.c:611-622: read a pair,
// align it, print it); a caller streaming batches wants several.
//
// Design: a context that is submitted to grows `async_lanes` LANES (option, default 3) -- each a host thread with a context of
// its own on the same device: own streams, own pinned staging, own device scratch.  A job is a synchronous host-level call bound to
// its arguments; lanes take jobs in submission order and run them to completion, so up to `async_lanes` batches are in flight and
// the GPU sees their kernels and copies on independent streams.  Nothing of the synchronous path changes -- a job IS that path -- so
// results are identical by construction (tests/test_gpu_async.py checks it with interleaved submits from two threads).
//   * the job takes a snapshot of the submitting context's OPTIONS at submit time: a test that steers its context steers its jobs;
//   * seqalign_job_wait returns the call's code, hands the job's error text to the waiting thread (seqalign_last_error) and its
//     launch record to the context (seqalign_ctx_last_call_info), and frees the ticket;
//   * the caller's buffers (batch arrays, scoring, outputs) are borrowed until the wait returns;
//   * seqalign_ctx_destroy drains the queue and joins the lanes.
// The host worker pool (HostPool) is one per process and runs one parallel loop at a time: lanes take turns at it loop by loop.
#include <deque>

#include "sa_ctx.hpp"

using namespace sa_host;

struct seqalign_job {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  int rc = SEQALIGN_OK;
  std::string err;
  seqalign_call_info_t info = {};
  seqalign_ctx *parent = nullptr;
  SaOptions opt;
  size_t chunk_budget = 0;
  std::function<int(seqalign_ctx *)> run;
};

namespace {

struct Lanes {
  seqalign_ctx *parent = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<seqalign_job *> queue;
  std::vector<std::thread> threads;
  std::vector<seqalign_ctx *> ctxs;
  bool stop = false;

  void lane_main(unsigned lane) {
    (void)hipSetDevice(parent->device);
    for (;;) {
      seqalign_job *job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;   // (stop, and nothing left: the queue is drained before the lanes go)
        job = queue.front();
        queue.pop_front();
      }
      int rc = SEQALIGN_OK;
      if (!ctxs[lane]) {
        seqalign_ctx_t *c = nullptr;
        rc = seqalign_ctx_create(parent->device, &c);
        ctxs[lane] = c;
      }
      if (rc == SEQALIGN_OK) {
        seqalign_ctx *c = ctxs[lane];
        c->opt = job->opt;
        c->chunk_budget = job->chunk_budget;
        rc = job->run(c);
        job->info = c->call_info;
      }
      {
        std::lock_guard<std::mutex> lk(job->mu);
        job->rc = rc;
        if (rc) job->err = seqalign_last_error();
        job->done = true;
      }
      job->cv.notify_all();
    }
  }
};

Lanes *lanes_of(seqalign_ctx *ctx) {
  static std::mutex create_mu;
  std::lock_guard<std::mutex> lk(create_mu);
  if (!ctx->async) {
    Lanes *l = new Lanes();
    l->parent = ctx;
    const unsigned n = std::max(1u, std::min(8u, ctx->opt.async_lanes ? ctx->opt.async_lanes : 3u));
    l->ctxs.assign(n, nullptr);
    for (unsigned k = 0; k < n; ++k) l->threads.emplace_back([l, k] { l->lane_main(k); });
    ctx->async = l;
  }
  return static_cast<Lanes *>(ctx->async);
}

int submit(seqalign_ctx *ctx, std::function<int(seqalign_ctx *)> run, seqalign_job_t **out) {
  seqalign_job *job = new (std::nothrow) seqalign_job();
  if (!job) return SEQALIGN_E_NOMEM;
  job->parent = ctx;
  job->opt = ctx->opt;
  job->chunk_budget = ctx->chunk_budget;
  job->run = std::move(run);
  Lanes *l = lanes_of(ctx);
  {
    std::lock_guard<std::mutex> lk(l->mu);
    l->queue.push_back(job);
  }
  l->cv.notify_one();
  *out = job;
  return SEQALIGN_OK;
}

}  // namespace

// (sa_device.hip: seqalign_ctx_destroy)
void sa_host::async_shutdown(seqalign_ctx *ctx) {
  if (!ctx->async) return;
  Lanes *l = static_cast<Lanes *>(ctx->async);
  {
    std::lock_guard<std::mutex> lk(l->mu);
    l->stop = true;
  }
  l->cv.notify_all();
  for (std::thread &t : l->threads) t.join();   // every lane leaves only when the queue is empty: submitted jobs are run, not dropped
  for (seqalign_ctx *c : l->ctxs) if (c) seqalign_ctx_destroy(c);
  delete l;
  ctx->async = nullptr;
}

extern "C" int seqalign_nw_batch_submit(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                        const uint64_t *str_off, char *out_a, char *out_b, uint32_t *out_len,
                                        int32_t *out_score, seqalign_job_t **job) {
  if (!ctx || !batch || !scoring || !str_off || !out_a || !out_b || !out_len || !out_score || !job) return SEQALIGN_E_ARG;
  const seqalign_batch_t b = *batch;   // (the descriptor by value; the arrays it points to are the caller's until the wait)
  return submit(ctx, [=](seqalign_ctx *c) { return seqalign_nw_batch(c, &b, scoring, str_off, out_a, out_b, out_len, out_score); }, job);
}

extern "C" int seqalign_sw_batch_submit(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                        const int32_t *min_score, uint32_t max_hits, seqalign_sw_hit_t *hits, uint64_t hit_cap,
                                        uint64_t *n_hits, char *out_a, char *out_b, uint64_t str_cap, seqalign_job_t **job) {
  if (!ctx || !batch || !scoring || !min_score || !hits || !n_hits || !out_a || !out_b || !job) return SEQALIGN_E_ARG;
  const seqalign_batch_t b = *batch;
  return submit(ctx, [=](seqalign_ctx *c) { return seqalign_sw_batch(c, &b, scoring, min_score, max_hits, hits, hit_cap, n_hits, out_a, out_b, str_cap); }, job);
}

extern "C" int seqalign_nw_batch_cigar_submit(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring, int format,
                                              const uint64_t *cigar_off, char *cigar, uint32_t *cigar_len, int32_t *out_score,
                                              seqalign_job_t **job) {
  if (!ctx || !batch || !scoring || !cigar_off || !cigar || !cigar_len || !out_score || !job) return SEQALIGN_E_ARG;
  const seqalign_batch_t b = *batch;
  return submit(ctx, [=](seqalign_ctx *c) { return seqalign_nw_batch_cigar(c, &b, scoring, format, cigar_off, cigar, cigar_len, out_score); }, job);
}

extern "C" int seqalign_job_done(seqalign_job_t *job) {
  if (!job) return 1;
  std::lock_guard<std::mutex> lk(job->mu);
  return job->done ? 1 : 0;
}

extern "C" int seqalign_job_wait(seqalign_job_t *job) {
  if (!job) return SEQALIGN_E_ARG;
  {
    std::unique_lock<std::mutex> lk(job->mu);
    job->cv.wait(lk, [&] { return job->done; });
  }
  const int rc = job->rc;
  if (rc) set_last_error(job->err);
  if (job->parent) job->parent->call_info = job->info;   // seqalign_ctx_last_call_info: what the job waited for last launched
  delete job;
  return rc;
}
