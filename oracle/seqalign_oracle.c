/*
 * seqalign_oracle.c -- plain-C CPU restatement of the seq-align DP hot path.
 *
 * TEST INFRASTRUCTURE (see seqalign_oracle.h).  Not shipped, not linked into
 * libseqalign_hip.so.  Every function cites the reference lines it restates;
 * nothing here is copied from the reference -- it is re-derived from the
 * behavioural spec in SURVEY.md appendix A and checked bit-for-bit against the
 * compiled reference (oracle/_ref) by tests/test_oracle_vs_ref.py and against
 * the committed golden vectors by tests/test_oracle_golden.py.
 */
#define _POSIX_C_SOURCE 200809L
#include "seqalign_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ helpers */

static inline int fold_char(const orc_scoring_t *sc, char c) {
  /* alignment_scoring.c:136-140 -- tolower() in the C locale */
  if (!sc->case_sensitive && c >= 'A' && c <= 'Z') return c - 'A' + 'a';
  return c;
}

static inline int bit_get(const uint32_t *words, int idx) {
  return (int)((words[idx >> 5] >> (idx & 31)) & 1u);
}
static inline void bit_set(uint32_t *words, int idx) {
  words[idx >> 5] |= (uint32_t)1 << (idx & 31);
}

static inline int imin(int x, int y) { return x < y ? x : y; }
static inline int imax(int x, int y) { return x > y ? x : y; }
static inline int64_t lmax(int64_t x, int64_t y) { return x > y ? x : y; }

size_t orc_sizeof_scoring(void) { return sizeof(orc_scoring_t); }

/* ----------------------------------------------------------- scoring model */

/* alignment_scoring.c:21-55 */
void orc_scoring_init(orc_scoring_t *sc, int match, int mismatch, int gap_open,
                      int gap_extend, int no_start_gap_penalty,
                      int no_end_gap_penalty, int no_gaps_in_a, int no_gaps_in_b,
                      int no_mismatches, int case_sensitive) {
  sc->gap_open = gap_open;
  sc->gap_extend = gap_extend;
  sc->no_start_gap_penalty = no_start_gap_penalty != 0;
  sc->no_end_gap_penalty = no_end_gap_penalty != 0;
  sc->no_gaps_in_a = no_gaps_in_a != 0;
  sc->no_gaps_in_b = no_gaps_in_b != 0;
  sc->no_mismatches = no_mismatches != 0;
  sc->use_match_mismatch = true;
  sc->match = match;
  sc->mismatch = mismatch;
  sc->case_sensitive = case_sensitive != 0;
  memset(sc->wildcards, 0, sizeof sc->wildcards);
  memset(sc->swap_set, 0, sizeof sc->swap_set);
  /* wildscores / swap_scores are deliberately left untouched, as upstream */
  sc->min_penalty = imin(match, mismatch);
  sc->max_penalty = imax(match, mismatch);
  if (!no_gaps_in_a || !no_gaps_in_b) {
    int first_gap = gap_open + gap_extend;
    sc->min_penalty = imin(sc->min_penalty, imin(first_gap, gap_extend));
    sc->max_penalty = imax(sc->max_penalty, imax(first_gap, gap_extend));
  }
}

/* alignment_scoring.c:57-64 */
void orc_scoring_add_wildcard(orc_scoring_t *sc, char c, int score) {
  int k = fold_char(sc, c);
  bit_set(sc->wildcards, k);
  sc->wildscores[k] = score;
  sc->min_penalty = imin(sc->min_penalty, score);
  sc->max_penalty = imax(sc->max_penalty, score);
}

/* alignment_scoring.c:66-72 -- note: no case folding here */
void orc_scoring_add_mutation(orc_scoring_t *sc, char a, char b, int score) {
  sc->swap_scores[(int)a][(int)b] = score;
  bit_set(sc->swap_set[(int)a], (int)b);
  sc->min_penalty = imin(sc->min_penalty, score);
  sc->max_penalty = imax(sc->max_penalty, score);
}

/* alignment_scoring.c:74-95; scores is read as scores[j*len + i] for the pair
 * (str[i], str[j]) (ARR_LOOKUP(scores,len,i,j), alignment_macros.h:11-12). */
void orc_scoring_add_mutations(orc_scoring_t *sc, const char *str,
                               const int *scores, int use_match_mismatch) {
  size_t n = strlen(str);
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < n; j++)
      orc_scoring_add_mutation(sc, (char)fold_char(sc, str[i]),
                               (char)fold_char(sc, str[j]), scores[j * n + i]);
  sc->use_match_mismatch = use_match_mismatch != 0;
}

/* alignment_scoring.c:115-129: score = min of the wildscores of whichever of
 * a, b are wildcards; returns whether either is one. */
static int wildcard_score(const orc_scoring_t *sc, int a, int b, int *score) {
  int have = 0, best = 0;
  if (bit_get(sc->wildcards, a)) { best = sc->wildscores[a]; have = 1; }
  if (bit_get(sc->wildcards, b)) {
    best = have ? imin(best, sc->wildscores[b]) : sc->wildscores[b];
    have = 1;
  }
  *score = have ? best : 0;
  return have;
}

/* alignment_scoring.c:133-182 -- decision order of SURVEY A.2 */
int orc_scoring_lookup(const orc_scoring_t *sc, char ca, char cb, int *score,
                       int *is_match) {
  int a = fold_char(sc, ca), b = fold_char(sc, cb);
  *is_match = (a == b);

  if (sc->no_mismatches && !*is_match) {         /* :148-153 */
    *is_match = wildcard_score(sc, a, b, score);
    return ORC_OK;
  }
  if (bit_get(sc->swap_set[a], b)) {             /* :156-160 */
    *score = sc->swap_scores[a][b];
    return ORC_OK;
  }
  if (wildcard_score(sc, a, b, score)) {         /* :165-169 */
    *is_match = 1;
    return ORC_OK;
  }
  if (sc->use_match_mismatch) {                  /* :172-176 */
    *score = *is_match ? sc->match : sc->mismatch;
    return ORC_OK;
  }
  *score = 0;
  return ORC_ERR_UNKNOWN_PAIR;                   /* :178-181 */
}

/* ------------------------------------------------------------------- fill */

static inline int32_t floor_score(const orc_scoring_t *sc, int is_sw) {
  /* alignment.c:41 */
  return is_sw ? 0 : (int32_t)(INT_MIN + abs(sc->min_penalty));
}

/* alignment.c:28-168.  Arithmetic is carried in 64 bits and narrowed, so the
 * oracle itself has no UB; inside the parity domain (every penalty >=
 * min_penalty) no narrowing ever changes a value. */
int orc_fill(const orc_scoring_t *sc, const char *a, size_t len_a,
             const char *b, size_t len_b, int is_sw, int32_t *M, int32_t *A,
             int32_t *B) {
  const size_t W = len_a + 1;
  const int64_t open1 = (int64_t)sc->gap_open + sc->gap_extend; /* :38 */
  const int64_t ext = sc->gap_extend;                          /* :39 */
  const int64_t lo = floor_score(sc, is_sw);

  M[0] = A[0] = B[0] = 0;                                      /* :47-49 */
  for (size_t i = 1; i <= len_a; i++) {                        /* row 0 */
    if (is_sw) { M[i] = A[i] = B[i] = 0; continue; }           /* :53-54 */
    M[i] = A[i] = (int32_t)lo;                                 /* :63-66 */
    B[i] = sc->no_start_gap_penalty
               ? 0 : (int32_t)(sc->gap_open + (int)i * sc->gap_extend);
  }
  for (size_t j = 1; j <= len_b; j++) {                        /* column 0 */
    size_t c = j * W;
    if (is_sw) { M[c] = A[c] = B[c] = (int32_t)lo; continue; } /* :55-56 */
    M[c] = B[c] = (int32_t)lo;                                 /* :74-79 */
    A[c] = sc->no_start_gap_penalty
               ? 0 : (int32_t)(sc->gap_open + (int)j * sc->gap_extend);
  }

  for (size_t j = 1; j <= len_b; j++) {
    const int bottom = (j == len_b);
    for (size_t i = 1; i <= len_a; i++) {
      const int rightmost = (i == len_a);
      const size_t cur = j * W + i, up = cur - W, left = cur - 1, diag = up - 1;
      int s, same;
      int rc = orc_scoring_lookup(sc, a[i - 1], b[j - 1], &s, &same); /* :98 */
      if (rc != ORC_OK) return rc;

      /* M: :101-116 */
      if (sc->no_mismatches && !same) {
        M[cur] = (int32_t)lo;
      } else {
        int64_t best = lmax(lmax(M[diag], A[diag]), B[diag]) + s;
        M[cur] = (int32_t)lmax(best, lo);
      }

      /* A (gap in a, vertical, predecessor row j-1): :121-137 */
      if (rightmost && sc->no_end_gap_penalty) {
        A[cur] = (int32_t)lmax(lmax(M[up], A[up]), B[up]);
      } else if (!sc->no_gaps_in_a || rightmost) {
        int64_t best = lmax(lmax(M[up] + open1, B[up] + open1), A[up] + ext);
        A[cur] = (int32_t)lmax(best, lo);
      } else {
        A[cur] = (int32_t)lo;
      }

      /* B (gap in b, horizontal, predecessor column i-1): :139-155 */
      if (bottom && sc->no_end_gap_penalty) {
        B[cur] = (int32_t)lmax(lmax(M[left], A[left]), B[left]);
      } else if (!sc->no_gaps_in_b || bottom) {
        int64_t best = lmax(lmax(M[left] + open1, A[left] + open1), B[left] + ext);
        B[cur] = (int32_t)lmax(best, lo);
      } else {
        B[cur] = (int32_t)lo;
      }
    }
  }
  return ORC_OK;
}

/* -------------------------------------------------------------- traceback */

/* alignment.c:244-350 */
int orc_reverse_move(const orc_scoring_t *sc, const char *a, size_t len_a,
                     const char *b, size_t len_b, const int32_t *M,
                     const int32_t *A, const int32_t *B, int *matrix,
                     int32_t *score, size_t *x, size_t *y) {
  const size_t W = len_a + 1;
  int s, same;
  int rc = orc_scoring_lookup(sc, a[*x - 1], b[*y - 1], &s, &same); /* :255 */
  if (rc != ORC_OK) return rc;

  int64_t a_open = (int64_t)sc->gap_open + sc->gap_extend, a_ext = sc->gap_extend;
  int64_t b_open = a_open, b_ext = a_ext;                      /* :261-262 */
  if (sc->no_end_gap_penalty) {                                /* :265-268 */
    if (*x == len_a) a_open = a_ext = 0;
    if (*y == len_b) b_open = b_ext = 0;
  }
  if (sc->no_start_gap_penalty) {                              /* :269-272 */
    if (*x == 0) a_open = a_ext = 0;
    if (*y == 0) b_open = b_ext = 0;
  }

  int64_t from_m, from_a, from_b;
  switch (*matrix) {                                           /* :276-307 */
    case ORC_MATCH: from_m = from_a = from_b = s; (*x)--; (*y)--; break;
    case ORC_GAP_A: from_m = a_open; from_a = a_ext; from_b = a_open; (*y)--; break;
    case ORC_GAP_B: from_m = b_open; from_a = b_open; from_b = b_ext; (*x)--; break;
    default: return ORC_ERR_TRACEBACK;
  }
  const size_t at = *y * W + *x;
  const int64_t want = *score;

  /* predecessor priority GAP_A, GAP_B, MATCH: :311-327 */
  if ((!sc->no_gaps_in_a || *x == 0 || *x == len_a) && A[at] + from_a == want) {
    *matrix = ORC_GAP_A; *score = A[at];
  } else if ((!sc->no_gaps_in_b || *y == 0 || *y == len_b) && B[at] + from_b == want) {
    *matrix = ORC_GAP_B; *score = B[at];
  } else if (M[at] + from_m == want) {
    *matrix = ORC_MATCH; *score = M[at];
  } else {
    return ORC_ERR_TRACEBACK;                                  /* :328-349 */
  }
  return ORC_OK;
}

/* needleman_wunsch.c:34-146 (everything after the aligner_align call) */
int orc_nw_traceback(const orc_scoring_t *sc, const char *a, size_t len_a,
                     const char *b, size_t len_b, const int32_t *M,
                     const int32_t *A, const int32_t *B, char *out_a,
                     char *out_b, size_t *out_len, int32_t *out_score) {
  const size_t W = len_a + 1, H = len_b + 1, last = W * H - 1;
  const size_t longest = len_a + len_b;

  /* end cell: start MATCH, then GAP_B if >=, then GAP_A if >=  (:53-66) */
  int matrix = ORC_MATCH;
  int32_t score = M[last];
  if (B[last] >= score) { matrix = ORC_GAP_B; score = B[last]; }
  if (A[last] >= score) { matrix = ORC_GAP_A; score = A[last]; }
  *out_score = score;                                          /* :72 */

  /* write right-to-left into the tail of a longest-size buffer */
  size_t x = len_a, y = len_b, w = longest; /* next column goes to w-1 */
  while (x > 0 && y > 0) {                                     /* :79-114 */
    w--;
    out_a[w] = (matrix == ORC_GAP_A) ? '-' : a[x - 1];
    out_b[w] = (matrix == ORC_GAP_B) ? '-' : b[y - 1];
    int rc = orc_reverse_move(sc, a, len_a, b, len_b, M, A, B, &matrix, &score, &x, &y);
    if (rc != ORC_OK) return rc;
  }
  while (y > 0) { w--; out_a[w] = '-'; out_b[w] = b[y - 1]; y--; }   /* :117-123 */
  while (x > 0) { w--; out_a[w] = a[x - 1]; out_b[w] = '-'; x--; }   /* :126-132 */

  size_t n = longest - w;                                      /* :135-145 */
  memmove(out_a, out_a + w, n);
  memmove(out_b, out_b + w, n);
  out_a[n] = out_b[n] = '\0';
  *out_len = n;
  return ORC_OK;
}

int orc_nw_align(const orc_scoring_t *sc, const char *a, size_t len_a,
                 const char *b, size_t len_b, char *out_a, char *out_b,
                 size_t *out_len, int32_t *out_score) {
  size_t cells = (len_a + 1) * (len_b + 1);
  int32_t *buf = malloc(3 * cells * sizeof(int32_t));
  if (!buf) return ORC_ERR_CAPACITY;
  int rc = orc_fill(sc, a, len_a, b, len_b, 0, buf, buf + cells, buf + 2 * cells);
  if (rc == ORC_OK)
    rc = orc_nw_traceback(sc, a, len_a, b, len_b, buf, buf + cells,
                          buf + 2 * cells, out_a, out_b, out_len, out_score);
  free(buf);
  return rc;
}

/* ---------------------------------------------------------------- SW hits */

typedef struct { const int32_t *M; size_t W; } hit_order_t;
static __thread hit_order_t g_order; /* qsort has no context argument in C99; per thread: the CPU baseline runs one aligner per thread */

/* smith_waterman.c:71-86: score descending, then column ascending; the
 * remaining tie (upstream comparator returns 0) is fixed as index ascending. */
static int hit_cmp(const void *pa, const void *pb) {
  size_t p = *(const size_t *)pa, q = *(const size_t *)pb;
  int32_t sp = g_order.M[p], sq = g_order.M[q];
  if (sp != sq) return sp > sq ? -1 : 1;
  size_t cp = p % g_order.W, cq = q % g_order.W;
  if (cp != cq) return cp < cq ? -1 : 1;
  return p < q ? -1 : (p > q);
}

int orc_sw_hits(const orc_scoring_t *sc, const char *a, size_t len_a,
                const char *b, size_t len_b, const int32_t *M, const int32_t *A,
                const int32_t *B, int32_t min_score, size_t max_hits,
                orc_hit_t *hits, size_t *n_hits, char *str_a, char *str_b,
                size_t str_cap) {
  const size_t W = len_a + 1, cells = W * (len_b + 1);
  size_t *cand = malloc(cells * sizeof(size_t));
  unsigned char *seen = calloc(cells, 1); /* fresh mask: SURVEY A.3-2 */
  size_t n_cand = 0, used = 0, found = 0;
  int rc = ORC_OK;
  if (!cand || !seen) { free(cand); free(seen); return ORC_ERR_CAPACITY; }

  for (size_t p = 0; p < cells; p++)                           /* :152-156 */
    if (M[p] > 0) cand[n_cand++] = p;
  g_order.M = M; g_order.W = W;
  qsort(cand, n_cand, sizeof(size_t), hit_cmp);                /* :159-161 */

  for (size_t k = 0; k < n_cand && found < max_hits; k++) {    /* fetch :260-277 */
    const size_t end = cand[k];
    if (seen[end]) continue;                                   /* :269 */

    /* pass 1 (:187-199): walk to score 0, marking; abandon on a marked cell */
    size_t x = end % W, y = end / W, steps = 0;
    int matrix = ORC_MATCH, clash = 0;
    int32_t score = M[end];
    for (;; steps++) {
      size_t at = y * W + x;
      if (seen[at]) { clash = 1; break; }
      seen[at] = 1;
      if (score == 0) break;
      rc = orc_reverse_move(sc, a, len_a, b, len_b, M, A, B, &matrix, &score, &x, &y);
      if (rc != ORC_OK) goto done;
    }
    if (clash) continue;

    /* the CLI stops at the first fetched hit below min_score (sw_cmdline.c:214-217) */
    if (M[end] < min_score) break;
    if (used + steps + 1 > str_cap) { rc = ORC_ERR_CAPACITY; goto done; }

    /* pass 2 (:217-244): replay, writing columns right-to-left */
    char *ra = str_a + used, *rb = str_b + used;
    x = end % W; y = end / W; matrix = ORC_MATCH; score = M[end];
    for (size_t w = steps; score > 0;) {
      w--;
      ra[w] = (matrix == ORC_GAP_A) ? '-' : a[x - 1];
      rb[w] = (matrix == ORC_GAP_B) ? '-' : b[y - 1];
      rc = orc_reverse_move(sc, a, len_a, b, len_b, M, A, B, &matrix, &score, &x, &y);
      if (rc != ORC_OK) goto done;
    }
    ra[steps] = rb[steps] = '\0';

    orc_hit_t *h = &hits[found++];                             /* :249-255 */
    h->score = M[end];
    h->pos_a = x; h->pos_b = y;
    h->len_a = end % W - x; h->len_b = end / W - y;
    h->length = steps;
    h->str_off = used;
    used += steps + 1;
  }
done:
  *n_hits = found;
  free(cand);
  free(seen);
  return rc;
}

/* ---------------------------------------------------------------- utility */

uint64_t orc_fnv1a64(const void *data, size_t n_bytes) {
  const unsigned char *p = data;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n_bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

double orc_time_fill_batch(const orc_scoring_t *sc, const char *arena,
                           const uint64_t *off_a, const uint32_t *len_a,
                           const uint64_t *off_b, const uint32_t *len_b,
                           size_t n_pairs, int is_sw, uint64_t *checksum) {
  size_t max_cells = 0;
  for (size_t p = 0; p < n_pairs; p++) {
    size_t c = ((size_t)len_a[p] + 1) * ((size_t)len_b[p] + 1);
    if (c > max_cells) max_cells = c;
  }
  int32_t *buf = malloc(3 * max_cells * sizeof(int32_t));
  if (!buf) return -1.0;
  uint64_t acc = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (size_t p = 0; p < n_pairs; p++) {
    size_t c = ((size_t)len_a[p] + 1) * ((size_t)len_b[p] + 1);
    orc_fill(sc, arena + off_a[p], len_a[p], arena + off_b[p], len_b[p], is_sw,
             buf, buf + max_cells, buf + 2 * max_cells);
    /* touch the results so the fill cannot be elided; cheap vs the fill */
    acc += (uint32_t)buf[c - 1] + (uint32_t)buf[max_cells + c - 1] +
           (uint32_t)buf[2 * max_cells + c - 1];
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (checksum) *checksum = acc;
  free(buf);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
