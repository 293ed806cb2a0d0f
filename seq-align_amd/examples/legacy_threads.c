/*
 * legacy_threads.c -- one nw_aligner_t per thread through the reference-shaped API, as a seq-align user would write
 * it (SURVEY 8b "Threading": the reference's aligner_align mutates only its own aligner_t, src/alignment.c:170-202).
 * Prints pairs per second with 1 and with N threads and checks that every thread got the single-thread answers.
 *
 *   legacy_threads [threads=8] [rounds=6]
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "needleman_wunsch.h"

#define N_PAIRS 48
#define LEN 120

static char seq_a[N_PAIRS][LEN + 1], seq_b[N_PAIRS][LEN + 1];
static int want_score[N_PAIRS];
static char want_a[N_PAIRS][2 * LEN + 2], want_b[N_PAIRS][2 * LEN + 2];
static scoring_t scoring;

typedef struct { int rounds, ok, fill_expected; } job_t;

static void *worker(void *arg)
{
  job_t *job = (job_t *)arg;
  nw_aligner_t *nw = needleman_wunsch_new();
  alignment_t *res = alignment_create(2 * LEN + 2);
  int r, p;
  job->ok = 1;
  for(r = 0; r < job->rounds; r++) {
    for(p = 0; p < N_PAIRS; p++) {
      needleman_wunsch_align(seq_a[p], seq_b[p], &scoring, nw, res);
      if(job->fill_expected) {
        want_score[p] = res->score;
        strcpy(want_a[p], res->result_a);
        strcpy(want_b[p], res->result_b);
      } else if(res->score != want_score[p] || strcmp(res->result_a, want_a[p]) || strcmp(res->result_b, want_b[p])) {
        job->ok = 0;
      }
    }
  }
  alignment_free(res);
  needleman_wunsch_free(nw);
  return NULL;
}

static double run(int threads, int rounds, int fill_expected, int *ok)
{
  pthread_t tid[64];
  job_t job[64];
  struct timespec t0, t1;
  int t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for(t = 0; t < threads; t++) {
    job[t].rounds = rounds; job[t].fill_expected = fill_expected; job[t].ok = 1;
    pthread_create(&tid[t], NULL, worker, &job[t]);
  }
  *ok = 1;
  for(t = 0; t < threads; t++) { pthread_join(tid[t], NULL); *ok &= job[t].ok; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)threads * rounds * N_PAIRS / ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
}

int main(int argc, char **argv)
{
  int threads = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 6, ok1, okn, p, i, k;
  unsigned long long x = 88172645463325252ull;
  double one = 0, many = 0, r;
  if(threads < 1 || threads > 64) threads = 8;
  scoring_system_default(&scoring);
  for(p = 0; p < N_PAIRS; p++) {   /* b = a with substitutions and a deletion: alignments with gaps */
    for(i = 0; i < LEN; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; seq_a[p][i] = "ACGT"[x & 3]; }
    for(i = 0, k = 0; i < LEN; i++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      if(x % 41 == 0) continue;
      seq_b[p][k++] = (x % 17 == 0) ? "ACGT"[(x >> 8) & 3] : seq_a[p][i];
    }
    seq_b[p][k] = '\0';
  }
  run(1, 1, 1, &ok1);                 /* the answers; and the main thread's... no: a worker's context, first use allocates */
  run(threads, 1, 0, &okn);           /* every thread's context and scratch exist */
  for(i = 0; i < 3; i++) { r = run(1, rounds, 0, &ok1); if(r > one) one = r; if(!ok1) break; }
  for(i = 0; i < 3; i++) { r = run(threads, rounds, 0, &okn); if(r > many) many = r; if(!okn) break; }
  printf("{\"threads\": %d, \"pairs_per_s_1\": %.0f, \"pairs_per_s_n\": %.0f, \"speedup\": %.2f, \"identical\": %s}\n",
         threads, one, many, many / one, (ok1 && okn) ? "true" : "false");
  return (ok1 && okn) ? 0 : 1;
}
