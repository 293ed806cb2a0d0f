/*
 * host_sanitize.c -- the product's host C layer (scoring model, flatten, host
 * traceback) driven under AddressSanitizer + UBSan, with matrices from the oracle
 * (no GPU involved).  Built and run by tests/test_host_sanitizers.py.
 * Exit status 0 = every check passed and no sanitizer report.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "alignment.h"
#include "sa_internal.h"
#include "seqalign_oracle.h"

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd(unsigned n) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (unsigned)(rng_state % n);
}

static int check_pair(const scoring_t *sc, const char *a, size_t la, const char *b, size_t lb) {
  size_t cells = (la + 1) * (lb + 1), n1 = 0, n2 = 0;
  int32_t *M = malloc(3 * cells * sizeof(int32_t)), *A = M + cells, *B = A + cells, s1 = 0, s2 = 0;
  char *r1a = malloc(la + lb + 1), *r1b = malloc(la + lb + 1), *r2a = malloc(la + lb + 1), *r2b = malloc(la + lb + 1);
  int bad = 0;
  if (orc_fill((const orc_scoring_t *)sc, a, la, b, lb, 0, M, A, B) != ORC_OK) bad = 1;
  sa_view_t v = { sc, a, b, la, lb, M, A, B };
  if (!bad && sa_nw_traceback(&v, r1a, r1b, &n1, &s1) != SEQALIGN_OK) bad = 2;
  if (!bad && orc_nw_traceback((const orc_scoring_t *)sc, a, la, b, lb, M, A, B, r2a, r2b, &n2, &s2) != ORC_OK) bad = 3;
  if (!bad && (n1 != n2 || s1 != s2 || memcmp(r1a, r2a, n1 + 1) || memcmp(r1b, r2b, n1 + 1))) bad = 4;
  free(M); free(r1a); free(r1b); free(r2a); free(r2b);
  return bad;
}

int main(void)
{
  static scoring_t sc;      /* 271 KB: keep it off the stack */
  static const char alpha[] = "ACGTacgtN";
  int failures = 0, trial;

  if (sizeof(scoring_t) != orc_sizeof_scoring()) { fprintf(stderr, "layout mismatch\n"); return 2; }

  for (trial = 0; trial < 200; trial++) {
    int flags = (int)rnd(32), cs = (int)rnd(2);
    int both = ((flags >> 2) & 1) && ((flags >> 3) & 1);
    char a[64], b[64];
    size_t la = rnd(60), lb = rnd(60), i;
    sa_flat_scoring_t flat;
    scoring_init(&sc, 1 + (int)rnd(3), both ? -7 : -(int)rnd(4), -(int)rnd(6), -(int)rnd(2),
                 flags & 1, (flags >> 1) & 1, (flags >> 2) & 1, (flags >> 3) & 1, (flags >> 4) & 1, cs);
    if (rnd(2)) scoring_add_wildcard(&sc, 'N', (int)rnd(3) - 1);
    if (rnd(2)) { scoring_add_mutation(&sc, 'a', 'c', -2); scoring_add_mutation(&sc, 'c', 'a', 1); }
    for (i = 0; i < la; i++) a[i] = alpha[rnd(sizeof(alpha) - 1)];
    for (i = 0; i < lb; i++) b[i] = alpha[rnd(sizeof(alpha) - 1)];
    /* flatten in both modes; NW may be out of the parity domain */
    if (sa_flatten_scoring(&sc, 1, &flat) != SEQALIGN_OK) { failures++; continue; }
    sa_flat_scoring_free(&flat);
    if (sa_flatten_scoring(&sc, 0, &flat) == SEQALIGN_OK) {
      int bad = check_pair(&sc, a, la, b, lb);
      if (bad) { fprintf(stderr, "trial %d: mismatch kind %d\n", trial, bad); failures++; }
      sa_flat_scoring_free(&flat);
    }
  }
  /* presets: build, flatten, free */
  {
    void (*presets[])(scoring_t *) = { scoring_system_default, scoring_system_BLOSUM62, scoring_system_BLOSUM80,
                                       scoring_system_PAM30, scoring_system_PAM70, scoring_system_DNA_hybridization };
    size_t k;
    for (k = 0; k < sizeof(presets) / sizeof(presets[0]); k++) {
      sa_flat_scoring_t flat;
      presets[k](&sc);
      if (sa_flatten_scoring(&sc, 1, &flat) != SEQALIGN_OK) failures++;
      else sa_flat_scoring_free(&flat);
    }
  }
  /* alignment_t growth */
  {
    alignment_t *r = alignment_create(0);
    alignment_ensure_capacity(r, 5);
    alignment_ensure_capacity(r, 5000);
    memset(r->result_a, 'x', 5000); r->result_a[5000] = 0;
    alignment_free(r);
  }
  printf("host_sanitize: %d failures\n", failures);
  return failures ? 1 : 0;
}
