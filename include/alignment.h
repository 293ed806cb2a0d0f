/*
 * alignment.h -- DP core: the aligner object and its consumers (host side, C).
 *
 * Mirrors the public interface of the reference src/alignment.h:23-79.  What is
 * different underneath: aligner_align() does not fill the three matrices with a
 * scalar row-major sweep (reference src/alignment.c:28-168); it uploads the two
 * sequences, runs the gfx950 fill kernel (seq-align_amd/csrc) as a batch of
 * one, and copies the three matrices back into the aligner's own buffers, so the
 * post-condition on aligner_t is unchanged and every consumer that reads the
 * struct (traceback, printers, user code) keeps working.  There is no CPU
 * fallback: without a device aligner_align() prints an error and exits, the
 * reference's convention for fatal conditions (src/alignment.c:228-231).
 */
#ifndef ALIGNMENT_HEADER_SEEN
#define ALIGNMENT_HEADER_SEEN

#include <string.h>
#include "alignment_scoring.h"

#ifndef ROUNDUP2POW
  #define ROUNDUP2POW(x) sa_roundup2pow64(x)
  static inline size_t sa_roundup2pow64(unsigned long long v) {
    unsigned long long p = 1;
    if(v <= 1) return (size_t)v;   /* 0 -> 0, 1 -> 1 as upstream */
    while(p < v) p <<= 1;
    return (size_t)p;
  }
#endif

typedef struct
{
  const scoring_t* scoring;          /* borrowed */
  const char *seq_a, *seq_b;         /* borrowed, not NUL-dependent */
  size_t score_width, score_height;  /* len_a+1, len_b+1 */
  /* three dense row-major matrices, cell (i,j) at j*score_width+i */
  score_t *match_scores, *gap_a_scores, *gap_b_scores;
  size_t capacity;                   /* cells allocated per matrix */
} aligner_t;

typedef struct
{
  char *result_a, *result_b;
  size_t capacity, length;
  size_t pos_a, pos_b;   /* 0-based start of the aligned region (SW) */
  size_t len_a, len_b;   /* bases consumed from each sequence (SW) */
  score_t score;
} alignment_t;

enum Matrix { MATCH,GAP_A,GAP_B };
#define MATRIX_NAME(x) ((x) == MATCH ? "MATCH" : ((x) == GAP_A ? "GAP_A" : "GAP_B"))

#ifdef __cplusplus
extern "C" {
#endif

extern const char align_col_mismatch[], align_col_indel[], align_col_context[],
                  align_col_stop[];

#define aligner_init(a) (memset(a, 0, sizeof(aligner_t)))

/* Boundary of the hot path (reference src/alignment.c:170-193). */
void aligner_align(aligner_t *aligner,
                   const char *seq_a, const char *seq_b,
                   size_t len_a, size_t len_b,
                   const scoring_t *scoring, char is_sw);
void aligner_destroy(aligner_t *aligner);

alignment_t* alignment_create(size_t capacity);
void alignment_ensure_capacity(alignment_t* result, size_t strlength);
void alignment_free(alignment_t* result);

/* One traceback step: re-derives the predecessor of (*score_x,*score_y) in
 * *curr_matrix from the stored scores, priority GAP_A, GAP_B, MATCH. */
void alignment_reverse_move(enum Matrix *curr_matrix, score_t *curr_score,
                            size_t *score_x, size_t *score_y,
                            size_t *arr_index, const aligner_t *aligner);

void alignment_print_matrices(const aligner_t *aligner);
void alignment_colour_print_against(const char *alignment_a,
                                    const char *alignment_b,
                                    char case_sensitive);
void alignment_print_spacer(const char* alignment_a, const char* alignment_b,
                            const scoring_t* scoring);

#ifdef __cplusplus
}
#endif

#endif
