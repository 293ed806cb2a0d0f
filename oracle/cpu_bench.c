/*
 * cpu_bench.c -- TEST / MEASUREMENT INFRASTRUCTURE (checker side, never linked into the product).
 *
 * pthread harness for bench.py's `cpu_baseline`: the reference CPU path over a batch of pairs on
 * N host threads, one aligner object per thread, pairs dealt round-robin by index -- the shape
 * SURVEY 8(d) asks for ("pthreads over pair index in C, all physical cores").  What is timed is
 * passed in as a function pointer, so the same harness drives
 *   kind "reference": aligner_align of the compiled reference (oracle/_ref/libseqalign_ref.so,
 *                     reference src/alignment.c:170-193; the thread's aligner_t is re-used across
 *                     pairs exactly as the reference tools re-use theirs, and freed with
 *                     aligner_destroy);
 *   kind "port"     : orc_fill of the restatement (oracle/seqalign_oracle.c) into per-thread buffers.
 * and, for the fill + traceback leg that stands beside bench.py's `e2e` ("fill-only and
 * fill+traceback"):
 *   mode 2 "reference": needleman_wunsch_align2 of the compiled reference (src/needleman_wunsch.c:34-146:
 *                     aligner_align + end-cell pick + alignment_reverse_move chain into an alignment_t);
 *                     one nw_aligner_t and one alignment_t per thread, re-used across pairs as
 *                     src/tools/nw_cmdline.c does;
 *   mode 3 "port"   : orc_nw_align of the restatement (fill + traceback into strings);
 *   mode 4 "port"   : SW best hit of the restatement: orc_fill + orc_sw_hits(max_hits = 1) -- the compiled
 *                     reference has no SW front-end here (smith_waterman.c needs the absent sort_r).
 * Threads are pinned to distinct CPUs of the process's affinity mask.  Every thread loops over its
 * pairs until the deadline, so the sample is bounded by time, not by the batch.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*align_fn)(void *aligner, const char *a, const char *b, size_t la, size_t lb,
                         const void *scoring, char is_sw);
typedef void (*destroy_fn)(void *aligner);
typedef int (*fill_fn)(const void *scoring, const char *a, size_t la, const char *b, size_t lb, int is_sw,
                       int32_t *M, int32_t *A, int32_t *B);

/* needleman_wunsch.h:22-34, alignment.h:62-64 of the reference */
typedef void *(*nw_new_fn)(void);
typedef void (*nw_free_fn)(void *nw);
typedef void *(*aln_create_fn)(size_t capacity);
typedef void (*aln_free_fn)(void *aln);
typedef void (*nw_align2_fn)(const char *a, const char *b, size_t la, size_t lb, const void *scoring, void *nw,
                             void *result);
/* oracle/seqalign_oracle.h */
typedef int (*orc_nw_align_fn)(const void *scoring, const char *a, size_t la, const char *b, size_t lb, char *out_a,
                               char *out_b, size_t *out_len, int32_t *out_score);
typedef struct { int32_t score; uint64_t pos_a, pos_b, len_a, len_b, length, str_off; } hit_t;
typedef int (*orc_sw_hits_fn)(const void *scoring, const char *a, size_t la, const char *b, size_t lb, const int32_t *M,
                              const int32_t *A, const int32_t *B, int32_t min_score, size_t max_hits, hit_t *hits,
                              size_t *n_hits, char *str_a, char *str_b, size_t str_cap);

typedef struct {
  int tid, n_threads, cpu, is_sw, mode;   /* mode 0: aligner_align-shaped, 1: orc_fill-shaped, 2-4: see the header */
  int32_t min_score;
  void *fn, *destroy;
  void *const *aux;                       /* mode 2: {nw_new, nw_free, alignment_create, alignment_free}; mode 4: {orc_sw_hits} */
  size_t max_len;
  const void *scoring;
  const char *arena;
  const uint64_t *off_a, *off_b;
  const uint32_t *len_a, *len_b;
  size_t n_pairs, max_cells;
  double deadline;                        /* CLOCK_MONOTONIC seconds */
  uint64_t cells, pairs, sink;
} job_t;

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void *worker(void *arg) {
  job_t *j = arg;
  if (j->cpu >= 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(j->cpu, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
  }
  /* aligner_t is 72 bytes upstream (src/alignment.h:23-30), zero-initialised by aligner_init */
  uint64_t aligner[16];
  memset(aligner, 0, sizeof(aligner));
  int32_t *buf = NULL;
  if (j->mode == 1 || j->mode == 4) buf = malloc(3 * j->max_cells * sizeof(int32_t));
  void *nw = NULL, *aln = NULL;
  char *str = NULL;
  if (j->mode == 2) {
    nw = ((nw_new_fn)j->aux[0])();
    aln = ((aln_create_fn)j->aux[2])(256);
  }
  if (j->mode == 3 || j->mode == 4) str = malloc(2 * (2 * j->max_len + 2));
  for (;;) {
    for (size_t p = (size_t)j->tid; p < j->n_pairs; p += (size_t)j->n_threads) {
      const char *a = j->arena + j->off_a[p], *b = j->arena + j->off_b[p];
      if (j->mode == 0) {
        ((align_fn)j->fn)(aligner, a, b, j->len_a[p], j->len_b[p], j->scoring, (char)j->is_sw);
        /* match_scores pointer is the 6th word of aligner_t; read the last cell so the call is observable */
        const int32_t *M = (const int32_t *)aligner[5];
        j->sink += (uint32_t)M[((size_t)j->len_a[p] + 1) * ((size_t)j->len_b[p] + 1) - 1];
      } else if (j->mode == 1) {
        ((fill_fn)j->fn)(j->scoring, a, j->len_a[p], b, j->len_b[p], j->is_sw, buf, buf + j->max_cells,
                         buf + 2 * j->max_cells);
        j->sink += (uint32_t)buf[((size_t)j->len_a[p] + 1) * ((size_t)j->len_b[p] + 1) - 1];
      } else if (j->mode == 2) {
        ((nw_align2_fn)j->fn)(a, b, j->len_a[p], j->len_b[p], j->scoring, nw, aln);
        /* alignment_t: result_a, result_b, capacity, length, ... (src/alignment.h:33-40) */
        j->sink += ((const uint64_t *)aln)[3] + (uint8_t)((const char *const *)aln)[0][0];
      } else if (j->mode == 3) {
        size_t n = 0;
        int32_t score = 0;
        ((orc_nw_align_fn)j->fn)(j->scoring, a, j->len_a[p], b, j->len_b[p], str, str + 2 * j->max_len + 2, &n, &score);
        j->sink += n + (uint32_t)score;
      } else {
        hit_t hit;
        size_t n_hits = 0;
        ((fill_fn)j->fn)(j->scoring, a, j->len_a[p], b, j->len_b[p], 1, buf, buf + j->max_cells, buf + 2 * j->max_cells);
        ((orc_sw_hits_fn)j->aux[0])(j->scoring, a, j->len_a[p], b, j->len_b[p], buf, buf + j->max_cells,
                                    buf + 2 * j->max_cells, j->min_score, 1, &hit, &n_hits, str,
                                    str + 2 * j->max_len + 2, 2 * j->max_len + 2);
        j->sink += n_hits;
      }
      j->cells += (uint64_t)j->len_a[p] * j->len_b[p];
      j->pairs++;
      if ((j->pairs & 15) == 0 && now_s() >= j->deadline) goto done;
    }
    if (now_s() >= j->deadline) break;
  }
done:
  if (j->mode == 0 && j->destroy) ((destroy_fn)j->destroy)(aligner);
  if (j->mode == 2) {
    ((nw_free_fn)j->aux[1])(nw);
    ((aln_free_fn)j->aux[3])(aln);
  }
  free(buf);
  free(str);
  return NULL;
}

/* Runs for ~seconds on n_threads threads; returns elapsed wall seconds (< 0 on error) and the totals. */
double cpubench_run_aux(void *fn, void *destroy, void *const *aux, int mode, const void *scoring, const char *arena,
                        const uint64_t *off_a, const uint32_t *len_a, const uint64_t *off_b, const uint32_t *len_b,
                        size_t n_pairs, int is_sw, int32_t min_score, int n_threads, double seconds,
                        uint64_t *cells_out, uint64_t *pairs_out) {
  if (n_threads < 1 || !n_pairs) return -1.0;
  if ((mode == 2 || mode == 4) && !aux) return -1.0;
  cpu_set_t allowed;
  int cpus[4096], n_cpus = 0;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
    for (int c = 0; c < CPU_SETSIZE && n_cpus < 4096; c++)
      if (CPU_ISSET(c, &allowed)) cpus[n_cpus++] = c;
  size_t max_cells = 0, max_len = 0;
  for (size_t p = 0; p < n_pairs; p++) {
    size_t c = ((size_t)len_a[p] + 1) * ((size_t)len_b[p] + 1);
    if (c > max_cells) max_cells = c;
    if ((size_t)len_a[p] + len_b[p] > max_len) max_len = (size_t)len_a[p] + len_b[p];
  }
  job_t *jobs = calloc((size_t)n_threads, sizeof(job_t));
  pthread_t *th = calloc((size_t)n_threads, sizeof(pthread_t));
  if (!jobs || !th) return -1.0;
  const double t0 = now_s();
  for (int t = 0; t < n_threads; t++) {
    job_t *j = &jobs[t];
    j->tid = t; j->n_threads = n_threads; j->cpu = n_cpus ? cpus[t % n_cpus] : -1; j->is_sw = is_sw; j->mode = mode;
    j->fn = fn; j->destroy = destroy; j->scoring = scoring; j->arena = arena;
    j->off_a = off_a; j->off_b = off_b; j->len_a = len_a; j->len_b = len_b;
    j->n_pairs = n_pairs; j->max_cells = max_cells; j->deadline = t0 + seconds;
    j->aux = aux; j->max_len = max_len; j->min_score = min_score;
    if (pthread_create(&th[t], NULL, worker, j) != 0) { n_threads = t; break; }
  }
  uint64_t cells = 0, pairs = 0;
  for (int t = 0; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    cells += jobs[t].cells;
    pairs += jobs[t].pairs;
  }
  const double dt = now_s() - t0;
  if (cells_out) *cells_out = cells;
  if (pairs_out) *pairs_out = pairs;
  free(jobs);
  free(th);
  return dt;
}

double cpubench_run(void *fn, void *destroy, int mode, const void *scoring, const char *arena,
                    const uint64_t *off_a, const uint32_t *len_a, const uint64_t *off_b, const uint32_t *len_b,
                    size_t n_pairs, int is_sw, int n_threads, double seconds, uint64_t *cells_out,
                    uint64_t *pairs_out) {
  if (mode > 1) return -1.0;
  return cpubench_run_aux(fn, destroy, NULL, mode, scoring, arena, off_a, len_a, off_b, len_b, n_pairs, is_sw, 0,
                          n_threads, seconds, cells_out, pairs_out);
}
