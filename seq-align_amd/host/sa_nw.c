/*
 * sa_nw.c -- Needleman-Wunsch front-end (host, C).  Interface:
 * include/needleman_wunsch.h (mirror of reference src/needleman_wunsch.h).
 * Fill on the GPU via aligner_align(); traceback on the host from the
 * GPU-produced matrices (reference needleman_wunsch.c:34-146 semantics).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "needleman_wunsch.h"
#include "sa_internal.h"

nw_aligner_t* needleman_wunsch_new()
{
  nw_aligner_t *nw = calloc(1, sizeof(nw_aligner_t));
  return nw;
}

void needleman_wunsch_free(nw_aligner_t *nw)
{
  aligner_destroy(nw);
  free(nw);
}

void needleman_wunsch_align(const char *a, const char *b,
                            const scoring_t *scoring,
                            nw_aligner_t *nw, alignment_t *result)
{
  needleman_wunsch_align2(a, b, strlen(a), strlen(b), scoring, nw, result);
}

void needleman_wunsch_align2(const char *a, const char *b,
                             size_t len_a, size_t len_b,
                             const scoring_t *scoring,
                             nw_aligner_t *nw, alignment_t *result)
{
  sa_view_t v;
  int32_t score = 0;
  size_t n = 0;
  int rc;

  aligner_align(nw, a, b, len_a, len_b, scoring, 0);      /* GPU fill */
  alignment_ensure_capacity(result, len_a + len_b);

  v.sc = scoring; v.a = a; v.b = b; v.len_a = len_a; v.len_b = len_b;
  v.M = nw->match_scores; v.A = nw->gap_a_scores; v.B = nw->gap_b_scores;
  rc = sa_nw_traceback(&v, result->result_a, result->result_b, &n, &score);
  if(rc != SEQALIGN_OK) {
    alignment_print_matrices(nw);
    fprintf(stderr, "Program error: traceback fail (%s)\n", seqalign_strerror(rc));
    exit(EXIT_FAILURE);
  }
  result->score = score;
  result->length = n;
}
