/*
 * sa_traceback.c -- host-side consumers of the GPU-produced matrices.
 *
 * north_star keeps traceback on the host: no direction bits are stored, the
 * predecessor of a cell is re-derived from the three score matrices with
 * equality tests (reference src/alignment.c:244-350), so these routines need
 * nothing but the matrices the fill kernel wrote.  Fresh code; behaviour pinned
 * by tests/golden (NW strings from the compiled reference, SW known answer).
 */
#include <stdlib.h>
#include <string.h>

#include "sa_internal.h"

/* One step back from (*x,*y) in *matrix.  Mirrors the decision order of
 * reference alignment.c:244-350 but reports instead of exiting. */
int sa_reverse_move_rc(const sa_view_t *v, int *matrix, int32_t *score,
                       size_t *x, size_t *y)
{
  const scoring_t *sc = v->sc;
  const size_t W = v->len_a + 1;
  int sub, same;
  /* gap costs for leaving (*x,*y): free in the last column/row under
   * no_end_gap_penalty, free on the borders under no_start_gap_penalty */
  int64_t open_a = (int64_t)sc->gap_open + sc->gap_extend, ext_a = sc->gap_extend;
  int64_t open_b = open_a, ext_b = ext_a;
  int64_t via_m, via_a, via_b, cur = *score;
  size_t at;

  if(sa_scoring_lookup_rc(sc, v->a[*x - 1], v->b[*y - 1], &sub, &same) != 0)
    return SEQALIGN_E_UNKNOWN_PAIR;

  if(sc->no_end_gap_penalty) {
    if(*x == v->len_a) open_a = ext_a = 0;
    if(*y == v->len_b) open_b = ext_b = 0;
  }
  if(sc->no_start_gap_penalty) {
    if(*x == 0) open_a = ext_a = 0;
    if(*y == 0) open_b = ext_b = 0;
  }

  if(*matrix == MATCH)      { via_m = via_a = via_b = sub; (*x)--; (*y)--; }
  else if(*matrix == GAP_A) { via_m = via_b = open_a; via_a = ext_a; (*y)--; }
  else if(*matrix == GAP_B) { via_m = via_a = open_b; via_b = ext_b; (*x)--; }
  else return SEQALIGN_E_TRACEBACK;

  at = *y * W + *x;
  if((!sc->no_gaps_in_a || *x == 0 || *x == v->len_a) && v->A[at] + via_a == cur) {
    *matrix = GAP_A; *score = v->A[at];
  } else if((!sc->no_gaps_in_b || *y == 0 || *y == v->len_b) && v->B[at] + via_b == cur) {
    *matrix = GAP_B; *score = v->B[at];
  } else if(v->M[at] + via_m == cur) {
    *matrix = MATCH; *score = v->M[at];
  } else {
    return SEQALIGN_E_TRACEBACK;
  }
  return SEQALIGN_OK;
}

/* Global traceback (reference needleman_wunsch.c:53-145).  out_a/out_b hold
 * len_a+len_b+1 bytes; columns are produced right to left, then moved down. */
int sa_nw_traceback(const sa_view_t *v, char *out_a, char *out_b,
                    size_t *out_len, int32_t *out_score)
{
  const size_t corner = (v->len_a + 1) * (v->len_b + 1) - 1;
  const size_t total = v->len_a + v->len_b;
  size_t x = v->len_a, y = v->len_b, head = total, n;
  int matrix = MATCH, rc;
  int32_t score = v->M[corner];

  /* ties at the corner resolve GAP_A > GAP_B > MATCH */
  if(v->B[corner] >= score) { matrix = GAP_B; score = v->B[corner]; }
  if(v->A[corner] >= score) { matrix = GAP_A; score = v->A[corner]; }
  *out_score = score;

  while(x > 0 && y > 0) {
    head--;
    out_a[head] = (matrix == GAP_A) ? '-' : v->a[x - 1];
    out_b[head] = (matrix == GAP_B) ? '-' : v->b[y - 1];
    rc = sa_reverse_move_rc(v, &matrix, &score, &x, &y);
    if(rc != SEQALIGN_OK) return rc;
  }
  for(; y > 0; y--) { head--; out_a[head] = '-'; out_b[head] = v->b[y - 1]; }
  for(; x > 0; x--) { head--; out_a[head] = v->a[x - 1]; out_b[head] = '-'; }

  n = total - head;
  memmove(out_a, out_a + head, n);
  memmove(out_b, out_b + head, n);
  out_a[n] = out_b[n] = '\0';
  *out_len = n;
  return SEQALIGN_OK;
}
