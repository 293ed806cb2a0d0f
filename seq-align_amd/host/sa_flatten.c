/*
 * sa_flatten.c -- flatten scoring_t into the device form.
 *
 * scoring_lookup (reference src/alignment_scoring.c:133-182) is called once per
 * DP cell upstream and costs ~25 % of the fill.  It is a pure function of
 * (scoring, a, b), so it is evaluated here ONCE per class pair:
 *
 *   folded char f = tolower(c) unless case_sensitive
 *   class(f)      = 0 ("generic") when f is no wildcard and appears in no
 *                   swap_set row/column; otherwise its own class 1..K-1
 *   table[ca][cb] = lookup(rep(ca), rep(cb))        (ca, cb not both 0)
 *   gen_eq/gen_ne = lookup of two generic chars that are equal / different
 *
 * which is exact because two chars of class 0 only ever reach the
 * match/mismatch fallback (or the no_mismatches / unknown-pair outcomes), and a
 * class >= 1 holds exactly one folded char.  no_mismatches and the "unknown pair
 * -> exit" case become the sentinels SA_S_BLOCKED / SA_S_UNKNOWN.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "sa_internal.h"

static int32_t cell_value(const scoring_t *sc, int a, int b, int *bad)
{
  int score, same;
  if(sa_scoring_lookup_rc(sc, a, b, &score, &same) != 0) return SA_S_UNKNOWN;
  if(sc->no_mismatches && !same) return SA_S_BLOCKED;   /* alignment.c:101-104 */
  if(score <= SA_S_UNKNOWN) *bad = 1;
  return score;
}

int sa_flatten_scoring(const scoring_t *sc, int is_sw, sa_flat_scoring_t *out)
{
  int known[256], rep[257], cls_of[256];
  int f, c, n_known = 0, g1 = -1, g2 = -1, bad = 0;
  uint32_t K;

  memset(out, 0, sizeof(*out));
  out->gap_open = sc->gap_open;
  out->open1 = sc->gap_open + sc->gap_extend;           /* alignment.c:38 */
  out->ext = sc->gap_extend;                            /* alignment.c:39 */
  out->floor = is_sw ? 0 : INT_MIN + abs(sc->min_penalty); /* alignment.c:41 */
  out->flags = (sc->no_start_gap_penalty ? SA_F_NO_START_GAP : 0) |
               (sc->no_end_gap_penalty ? SA_F_NO_END_GAP : 0) |
               (sc->no_gaps_in_a ? SA_F_NO_GAPS_A : 0) |
               (sc->no_gaps_in_b ? SA_F_NO_GAPS_B : 0) |
               (sc->no_mismatches ? SA_F_NO_MISMATCH : 0) |
               (is_sw ? SA_F_IS_SW : 0);

  /* which folded chars have table entries of their own */
  for(f = 0; f < 256; f++) {
    known[f] = 0;
    if(sa_fold_char(sc, f) != f) continue;     /* unreachable after folding */
    if(get_wildcard_bit(sc, f)) known[f] = 1;
  }
  for(f = 0; f < 256; f++) {
    int g;
    if(sa_fold_char(sc, f) != f) continue;
    for(g = 0; g < 256; g++) {
      if(sa_fold_char(sc, g) != g) continue;
      if(get_swap_bit(sc, f, g)) known[f] = known[g] = 1;
    }
  }
  rep[0] = -1;
  for(f = 0; f < 256; f++) {
    cls_of[f] = 0;
    if(sa_fold_char(sc, f) != f) continue;
    if(known[f]) { n_known++; cls_of[f] = n_known; rep[n_known] = f; }
    else if(g1 < 0) g1 = f;
    else if(g2 < 0) g2 = f;
  }
  if(n_known > 255) return SEQALIGN_E_ARG;      /* class id is 8 bits */
  K = (uint32_t)n_known + 1;
  rep[0] = g1;
  out->n_classes = K;

  for(c = 0; c < 256; c++) {
    f = sa_fold_char(sc, c);
    out->code[c] = (uint16_t)(f | (cls_of[f] << 8));
  }

  out->gen_eq = (g1 >= 0) ? cell_value(sc, g1, g1, &bad) : SA_S_UNKNOWN;
  out->gen_ne = (g2 >= 0) ? cell_value(sc, g1, g2, &bad) : out->gen_eq;
  if(g1 >= 0 && g2 < 0) out->gen_ne = SA_S_UNKNOWN; /* cannot occur in data */

  out->table = (int32_t*)malloc(sizeof(int32_t) * K * K);
  if(!out->table) return SEQALIGN_E_NOMEM;
  for(uint32_t ca = 0; ca < K; ca++) {
    for(uint32_t cb = 0; cb < K; cb++) {
      int32_t v;
      if(ca == 0 && cb == 0) v = out->gen_eq;   /* kernel resolves eq/ne */
      else if(rep[ca] < 0 || rep[cb] < 0) v = SA_S_UNKNOWN;
      else v = cell_value(sc, rep[ca], rep[cb], &bad);
      out->table[ca*K + cb] = v;
    }
  }

  {
    int sentinel = (out->gen_eq <= SA_S_UNKNOWN) || (out->gen_ne <= SA_S_UNKNOWN);
    for(uint32_t k = 0; k < K*K && !sentinel; k++)
      if(out->table[k] <= SA_S_UNKNOWN) sentinel = 1;
    if(sentinel) out->flags |= SA_F_HAS_SENTINEL;
  }

  /* Parity domain: in NW every value added to a cell that may
   * hold the floor must be >= -|min_penalty|, else the reference overflows. */
  if(!is_sw) {
    int lim = -abs(sc->min_penalty);
    /* gap penalties are always in use: even with no_gaps_in_a AND no_gaps_in_b the last column / row still
     * opens and extends gaps (alignment.c:128,146), while scoring_init leaves them out of min_penalty in
     * exactly that case (alignment_scoring.c:49-54) -- upstream then wraps around in the last row / column */
    if(out->open1 < lim || out->ext < lim) bad = 1;
    if(out->gen_eq > SA_S_UNKNOWN && out->gen_eq < lim) bad = 1;
    if(out->gen_ne > SA_S_UNKNOWN && out->gen_ne < lim) bad = 1;
    for(uint32_t k = 0; k < K*K; k++)
      if(out->table[k] > SA_S_UNKNOWN && out->table[k] < lim) bad = 1;
  }
  if(bad) { sa_flat_scoring_free(out); return SEQALIGN_E_DOMAIN; }
  return SEQALIGN_OK;
}

void sa_flat_scoring_free(sa_flat_scoring_t *f)
{
  free(f->table);
  f->table = NULL;
}
