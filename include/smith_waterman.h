/*
 * smith_waterman.h -- local alignment front-end (host side, C).
 * Mirrors reference src/smith_waterman.h:15-39.  seq_a, seq_b and scoring are
 * borrowed and must stay unchanged until the last smith_waterman_fetch().
 *
 * Deliberate difference: the visited mask is fully cleared on every
 * smith_waterman_align (the reference clears a quarter of it),
 * so hits of a re-used sw_aligner_t equal those of a fresh one.
 */
#ifndef SMITH_WATERMAN_HEADER_SEEN
#define SMITH_WATERMAN_HEADER_SEEN

#include "seq_align.h"
#include "alignment.h"

typedef struct sw_aligner_t sw_aligner_t;

#ifdef __cplusplus
extern "C" {
#endif

sw_aligner_t *smith_waterman_new();
void smith_waterman_free(sw_aligner_t *sw_aligner);

aligner_t* smith_waterman_get_aligner(sw_aligner_t *sw);

void smith_waterman_align(const char *seq_a, const char *seq_b,
                          const scoring_t *scoring, sw_aligner_t *sw);

void smith_waterman_align2(const char *seq_a, const char *seq_b,
                           size_t len_a, size_t len_b,
                           const scoring_t *scoring, sw_aligner_t *sw);

/* 1 and *result filled if another local alignment exists, else 0 */
int smith_waterman_fetch(sw_aligner_t *sw, alignment_t *result);

/* hit ordering used by smith_waterman_align2 (exported by the reference too) */
int sort_match_indices(const void *aa, const void *bb, void *arg);

#ifdef __cplusplus
}
#endif

#endif
