/*
 * sa_sw.c -- Smith-Waterman front-end (host, C).  Interface:
 * include/smith_waterman.h (mirror of reference src/smith_waterman.h).
 *
 * Fill on the GPU via aligner_align(); the candidate order and the
 * visited-mask hit enumeration follow reference smith_waterman.c:137-277 with
 * two documented differences: the mask is cleared completely on every align
 *  and equal-score/equal-column candidates are ordered by cell
 * index (what a stable sort of the ascending scan yields).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smith_waterman.h"
#include "sa_internal.h"

struct sw_aligner_t
{
  aligner_t aligner;
  uint32_t *seen;            /* visited cells, 1 bit per cell */
  size_t seen_words;
  size_t *order;             /* candidate cells in hit order */
  size_t order_cap, n_cand, next;
};

typedef struct { const score_t *m; size_t width; } order_key_t;

/* qsort_r-style comparator, exported like the reference's
 * (smith_waterman.c:71-86): score descending, then column ascending. */
int sort_match_indices(const void *aa, const void *bb, void *arg)
{
  size_t p = *(const size_t*)aa, q = *(const size_t*)bb;
  const order_key_t *k = arg;
  if(k->m[p] != k->m[q]) return k->m[p] > k->m[q] ? -1 : 1;
  size_t cp = p % k->width, cq = q % k->width;
  if(cp != cq) return cp < cq ? -1 : 1;
  return 0;
}

/* total order used internally: the comparator above, ties by cell index */
static int cmp_total(const void *aa, const void *bb, void *arg)
{
  int c = sort_match_indices(aa, bb, arg);
  if(c) return c;
  size_t p = *(const size_t*)aa, q = *(const size_t*)bb;
  return p < q ? -1 : (p > q);
}

/* bottom-up merge sort with context (C99 has no qsort_r) */
static void sort_cells(size_t *v, size_t n, order_key_t *key)
{
  size_t *tmp, *src = v, *dst, width, lo;
  if(n < 2) return;
  tmp = malloc(n * sizeof(size_t));
  if(!tmp) { fprintf(stderr, "%s:%i: Out of memory\n", __FILE__, __LINE__); exit(EXIT_FAILURE); }
  dst = tmp;
  for(width = 1; width < n; width *= 2) {
    for(lo = 0; lo < n; lo += 2*width) {
      size_t mid = lo + width < n ? lo + width : n;
      size_t hi = lo + 2*width < n ? lo + 2*width : n;
      size_t i = lo, j = mid, k = lo;
      while(i < mid && j < hi)
        dst[k++] = cmp_total(&src[j], &src[i], key) < 0 ? src[j++] : src[i++];
      while(i < mid) dst[k++] = src[i++];
      while(j < hi) dst[k++] = src[j++];
    }
    { size_t *t = src; src = dst; dst = t; }
  }
  if(src != v) memcpy(v, src, n * sizeof(size_t));
  free(tmp);
}

sw_aligner_t* smith_waterman_new()
{
  sw_aligner_t *sw = calloc(1, sizeof(sw_aligner_t));
  return sw;
}

void smith_waterman_free(sw_aligner_t *sw)
{
  aligner_destroy(&sw->aligner);
  free(sw->seen);
  free(sw->order);
  free(sw);
}

aligner_t* smith_waterman_get_aligner(sw_aligner_t *sw)
{
  return &sw->aligner;
}

void smith_waterman_align(const char *a, const char *b,
                          const scoring_t *scoring, sw_aligner_t *sw)
{
  smith_waterman_align2(a, b, strlen(a), strlen(b), scoring, sw);
}

void smith_waterman_align2(const char *a, const char *b,
                           size_t len_a, size_t len_b,
                           const scoring_t *scoring, sw_aligner_t *sw)
{
  aligner_t *al = &sw->aligner;
  size_t cells, words, p;
  order_key_t key;

  aligner_align(al, a, b, len_a, len_b, scoring, 1);      /* GPU fill */

  cells = al->score_width * al->score_height;
  words = (cells + 31) / 32;
  if(words > sw->seen_words) {
    sw->seen = realloc(sw->seen, words * sizeof(uint32_t));
    sw->seen_words = words;
  }
  if(cells > sw->order_cap) {
    sw->order_cap = ROUNDUP2POW(cells);
    sw->order = realloc(sw->order, sw->order_cap * sizeof(size_t));
  }
  if(!sw->seen || !sw->order) {
    fprintf(stderr, "%s:%i: Out of memory\n", __FILE__, __LINE__);
    exit(EXIT_FAILURE);
  }
  memset(sw->seen, 0, words * sizeof(uint32_t));           /* the WHOLE mask */

  sw->n_cand = sw->next = 0;
  for(p = 0; p < cells; p++)
    if(al->match_scores[p] > 0) sw->order[sw->n_cand++] = p;

  key.m = al->match_scores; key.width = al->score_width;
  sort_cells(sw->order, sw->n_cand, &key);
}

/* Walk one candidate back to score 0 (reference smith_waterman.c:165-258). */
static int follow_hit(sw_aligner_t *sw, size_t end, alignment_t *result)
{
  const aligner_t *al = &sw->aligner;
  size_t W = al->score_width;
  size_t x = end % W, y = end / W, at = end, steps = 0, w;
  enum Matrix matrix = MATCH;
  score_t score = al->match_scores[end];

  for(;; steps++) {
    if(bitset32_get(sw->seen, at)) return 0;   /* overlaps an earlier hit */
    bitset32_set(sw->seen, at);
    if(score == 0) break;
    alignment_reverse_move(&matrix, &score, &x, &y, &at, al);
  }

  alignment_ensure_capacity(result, steps);
  result->length = steps;

  x = end % W; y = end / W; at = end; matrix = MATCH;
  score = al->match_scores[end];
  for(w = steps; score > 0; ) {
    w--;
    result->result_a[w] = (matrix == GAP_A) ? '-' : al->seq_a[x - 1];
    result->result_b[w] = (matrix == GAP_B) ? '-' : al->seq_b[y - 1];
    alignment_reverse_move(&matrix, &score, &x, &y, &at, al);
  }
  result->result_a[steps] = result->result_b[steps] = '\0';

  result->score = al->match_scores[end];
  result->pos_a = x;
  result->pos_b = y;
  result->len_a = end % W - x;
  result->len_b = end / W - y;
  return 1;
}

int smith_waterman_fetch(sw_aligner_t *sw, alignment_t *result)
{
  while(sw->next < sw->n_cand) {
    size_t cell = sw->order[sw->next++];
    if(!bitset32_get(sw->seen, cell) && follow_hit(sw, cell, result)) return 1;
  }
  return 0;
}
