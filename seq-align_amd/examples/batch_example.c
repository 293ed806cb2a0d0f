/*
 * batch_example.c -- the batch C-ABI from plain C: what a caller of seq-align's
 * needleman_wunsch_align loop looks like after moving to seqalign_nw_batch
 * (INTEGRATION.md, section 2).  Build:  make -C seq-align_amd examples
 * Run (needs an MI355X):  seq-align_amd/bin/batch_example
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "seqalign_hip.h"

int main(void)
{
  /* three pairs packed into one arena, as a FASTA reader would collect them */
  static const char *a[] = {"ACAATAGAC", "ACGTGACAGAT", "CAGACGT"}, *b[] = {"ACGAATAGAT", "GTGGACGAGTA", "CGATA"};
  enum { N = 3 };
  char arena[256], *out_a, *out_b;
  uint64_t off_a[N], off_b[N], str_off[N], pos = 0, total = 0;
  uint32_t len_a[N], len_b[N], out_len[N];
  int32_t score[N];
  scoring_t scoring;
  seqalign_ctx_t *ctx = NULL;
  seqalign_batch_t batch;
  int p, rc;

  for(p = 0; p < N; p++) {
    len_a[p] = (uint32_t)strlen(a[p]); len_b[p] = (uint32_t)strlen(b[p]);
    off_a[p] = pos; memcpy(arena + pos, a[p], len_a[p]); pos += len_a[p];
    off_b[p] = pos; memcpy(arena + pos, b[p], len_b[p]); pos += len_b[p];
    str_off[p] = total; total += len_a[p] + len_b[p] + 1;     /* capacity of one alignment string */
  }
  out_a = malloc(total); out_b = malloc(total);

  scoring_system_default(&scoring);                          /* 1 / -2 / -4 / -1, as the reference tools */
  if((rc = seqalign_ctx_create(0, &ctx)) != SEQALIGN_OK) {
    fprintf(stderr, "no GPU: %s (%s)\n", seqalign_strerror(rc), seqalign_last_error());
    return EXIT_FAILURE;
  }
  memset(&batch, 0, sizeof batch);
  batch.n_pairs = N; batch.arena = arena; batch.arena_bytes = pos;
  batch.off_a = off_a; batch.len_a = len_a; batch.off_b = off_b; batch.len_b = len_b;

  rc = seqalign_nw_batch(ctx, &batch, &scoring, str_off, out_a, out_b, out_len, score);
  if(rc != SEQALIGN_OK) { fprintf(stderr, "%s %s\n", seqalign_strerror(rc), seqalign_last_error()); return EXIT_FAILURE; }
  for(p = 0; p < N; p++) printf("%s\n%s\nscore: %i\n\n", out_a + str_off[p], out_b + str_off[p], score[p]);

  seqalign_ctx_destroy(ctx);
  free(out_a); free(out_b);
  return EXIT_SUCCESS;
}
