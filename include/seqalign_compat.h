/*
 * seqalign_compat.h -- the reference-shaped C API of this library, in one place.
 *
 * noporpoise/seq-align exposes its DP path through four small headers
 * (src/alignment_scoring.h, src/alignment.h, src/needleman_wunsch.h,
 * src/smith_waterman.h).  Code written against them keeps compiling against this
 * library: alignment_scoring.h, alignment.h, needleman_wunsch.h and
 * smith_waterman.h in this directory are one-line forwards to this file, and every
 * type, field order, macro and function signature below is the one the reference
 * declares (citations per block), because callers fill and read the structs
 * directly.  What differs is underneath: aligner_align() runs the matrix fill on
 * an MI355X (seq-align_amd/csrc), see seqalign_hip.h for the batch interface that
 * the reference does not have.
 *
 * Implementation: seq-align_amd/host/sa_{scoring,alignment,nw,sw}.c.
 */
#ifndef SEQALIGN_COMPAT_H
#define SEQALIGN_COMPAT_H

#include <inttypes.h>
#include <limits.h>
#include <stdbool.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------ versions
 * reference src/seq_align.h:13-14 */
#define SEQ_ALIGN_VERSION_STR "1.0.0"
#define SEQ_ALIGN_VERSION 0x100

/* ------------------------------------------------------------------- scoring
 * reference src/alignment_scoring.h:16-56.  A gap of N characters costs
 * gap_open + N * gap_extend.  min_penalty/max_penalty track the smallest/largest
 * penalty the builders have seen; min_penalty fixes the NW floor
 * INT_MIN + |min_penalty| (src/alignment.c:41). */
typedef int score_t;
#define SCORE_MIN INT_MIN

typedef struct
{
  int gap_open, gap_extend;
  bool no_start_gap_penalty, no_end_gap_penalty;     /* free gaps at the ends     */
  bool no_gaps_in_a, no_gaps_in_b, no_mismatches;    /* restrict the alignment    */
  bool use_match_mismatch;                           /* fallback for unknown pairs */
  int match, mismatch;
  bool case_sensitive;
  uint32_t wildcards[256/32], swap_set[256][256/32]; /* bitsets: wildcard chars; pairs with a score */
  score_t wildscores[256], swap_scores[256][256];
  int min_penalty, max_penalty;
} scoring_t;

#ifndef bitset32_get
#define bitset32_get(arr,idx)   (((arr)[(idx)>>5] >> ((idx)&31)) & 0x1)
#define bitset32_set(arr,idx)   ((arr)[(idx)>>5] |=   (1<<((idx)&31)))
#define bitset32_clear(arr,idx) ((arr)[(idx)>>5] &=  ~(1<<((idx)&31)))
#endif
#define get_wildcard_bit(scoring,c)    bitset32_get((scoring)->wildcards,c)
#define set_wildcard_bit(scoring,c)    bitset32_set((scoring)->wildcards,c)
#define get_swap_bit(scoring,a,b)      bitset32_get((scoring)->swap_set[(size_t)(a)],b)
#define set_swap_bit(scoring,a,b)      bitset32_set((scoring)->swap_set[(size_t)(a)],b)
#define scoring_is_wildcard(scoring,c) (get_wildcard_bit(scoring,c))

/* ------------------------------------------------------------------- aligner
 * reference src/alignment.h:14-45.  Three dense row-major matrices of
 * score_width * score_height cells, cell (i,j) at j*score_width + i; seq_a, seq_b
 * and scoring are borrowed. */
#ifndef ROUNDUP2POW
#define ROUNDUP2POW(x) seqalign_roundup_pow2(x)
static inline size_t seqalign_roundup_pow2(unsigned long long v)
{
  unsigned long long p = 1;
  if(v <= 1) return (size_t)v;
  while(p < v) p <<= 1;
  return (size_t)p;
}
#endif

typedef struct
{
  const scoring_t* scoring;
  const char *seq_a, *seq_b;
  size_t score_width, score_height;                  /* len_a + 1, len_b + 1 */
  score_t *match_scores, *gap_a_scores, *gap_b_scores;
  size_t capacity;                                   /* cells allocated per matrix */
} aligner_t;

typedef struct
{
  char *result_a, *result_b;
  size_t capacity, length;
  size_t pos_a, pos_b;                               /* SW: 0-based start of the hit */
  size_t len_a, len_b;                               /* SW: characters consumed      */
  score_t score;
} alignment_t;

enum Matrix { MATCH, GAP_A, GAP_B };
#define MATRIX_NAME(x) ((x) == MATCH ? "MATCH" : ((x) == GAP_A ? "GAP_A" : "GAP_B"))
#define aligner_init(a) (memset(a, 0, sizeof(aligner_t)))

typedef aligner_t nw_aligner_t;                      /* reference needleman_wunsch.h:16 */
typedef struct sw_aligner_t sw_aligner_t;            /* reference smith_waterman.h:15, opaque */

#ifdef __cplusplus
extern "C" {
#endif

/* --- scoring builders and lookup: reference src/alignment_scoring.h:58-81 --- */
void scoring_init(scoring_t *scoring, int match, int mismatch, int gap_open, int gap_extend,
                  bool no_start_gap_penalty, bool no_end_gap_penalty, bool no_gaps_in_a,
                  bool no_gaps_in_b, bool no_mismatches, bool case_sensitive);
void scoring_add_wildcard(scoring_t *scoring, char c, int s);
void scoring_add_mutation(scoring_t *scoring, char a, char b, int score);
/* scores[j*strlen(str) + i] belongs to the pair (str[i], str[j]) */
void scoring_add_mutations(scoring_t *scoring, const char *str, const int *scores, char use_match_mismatch);
void scoring_print(const scoring_t *scoring);
/* sets *score and *is_match; prints and exit(EXIT_FAILURE)s for a pair without a
 * score when use_match_mismatch is off, as the reference does */
void scoring_lookup(const scoring_t *scoring, char a, char b, int *score, bool *is_match);
void scoring_system_default(scoring_t *scoring);
void scoring_system_BLOSUM62(scoring_t *scoring);
void scoring_system_BLOSUM80(scoring_t *scoring);
void scoring_system_PAM30(scoring_t *scoring);
void scoring_system_PAM70(scoring_t *scoring);
void scoring_system_DNA_hybridization(scoring_t *scoring);
extern int blosum62[576];                            /* exported by the reference too (alignment_scoring.c:268) */

/* --- DP core: reference src/alignment.h:51-79 --- */
extern const char align_col_mismatch[], align_col_indel[], align_col_context[], align_col_stop[];

/* THE BOUNDARY of the hot path (reference src/alignment.c:170-193): records the
 * borrowed pointers, grows the matrices, fills them -- here on the GPU, as a batch
 * of one.  No CPU fallback: without a gfx950 device it prints and exits.
 * Where this call prints and exit()s although upstream would return (all are inputs
 * on which upstream's own result is undefined or impractical; the batch API reports
 * them as SEQALIGN_E_* codes instead):
 *   - (len_a+1)*(len_b+1) >= 2^31 cells (24 GB of matrices for ONE pair);
 *   - global alignment with a gap or substitution penalty below -|min_penalty|: upstream
 *     computes INT_MIN + |min_penalty| + penalty, a signed overflow.  This
 *     happens when penalties are edited after scoring_init without updating min_penalty,
 *     and for scoring_init(..., no_gaps_in_a = no_gaps_in_b = 1) with gap penalties below
 *     the mismatch score (upstream leaves them out of min_penalty, alignment_scoring.c:49-54,
 *     but still applies them in the last row and column, alignment.c:128,146).
 * Every thread that calls the legacy API gets its own device context (stream, scratch, cached
 * scoring) on first use: one aligner_t per thread runs in parallel, as with the reference
 * (src/alignment.c:170-202 mutates only its own aligner_t).  A call still costs a launch and a
 * PCIe round trip (~40 us for a tiny pair, ~0.12 ms for 150 x 150) -- batches belong in seqalign_hip.h. */
void aligner_align(aligner_t *aligner, const char *seq_a, const char *seq_b,
                   size_t len_a, size_t len_b, const scoring_t *scoring, char is_sw);
void aligner_destroy(aligner_t *aligner);

alignment_t* alignment_create(size_t capacity);
void alignment_ensure_capacity(alignment_t *result, size_t strlength);
void alignment_free(alignment_t *result);

/* one traceback step: predecessor of (*score_x,*score_y) in *curr_matrix,
 * re-derived from the stored scores with priority GAP_A, GAP_B, MATCH */
void alignment_reverse_move(enum Matrix *curr_matrix, score_t *curr_score,
                            size_t *score_x, size_t *score_y, size_t *arr_index,
                            const aligner_t *aligner);

void alignment_print_matrices(const aligner_t *aligner);
void alignment_colour_print_against(const char *alignment_a, const char *alignment_b, char case_sensitive);
void alignment_print_spacer(const char *alignment_a, const char *alignment_b, const scoring_t *scoring);

/* --- global alignment: reference src/needleman_wunsch.h:22-33 --- */
nw_aligner_t* needleman_wunsch_new();
void needleman_wunsch_free(nw_aligner_t *nw);
void needleman_wunsch_align(const char *a, const char *b, const scoring_t *scoring,
                            nw_aligner_t *nw, alignment_t *result);
void needleman_wunsch_align2(const char *a, const char *b, size_t len_a, size_t len_b,
                             const scoring_t *scoring, nw_aligner_t *nw, alignment_t *result);

/* --- local alignment: reference src/smith_waterman.h:21-39.  seq_a, seq_b and
 * scoring must stay unchanged until the last fetch.  Unlike the reference the
 * visited mask is cleared completely on every align. --- */
sw_aligner_t* smith_waterman_new();
void smith_waterman_free(sw_aligner_t *sw_aligner);
aligner_t* smith_waterman_get_aligner(sw_aligner_t *sw);
void smith_waterman_align(const char *seq_a, const char *seq_b, const scoring_t *scoring, sw_aligner_t *sw);
void smith_waterman_align2(const char *seq_a, const char *seq_b, size_t len_a, size_t len_b,
                           const scoring_t *scoring, sw_aligner_t *sw);
int smith_waterman_fetch(sw_aligner_t *sw, alignment_t *result);   /* 1 = *result filled, 0 = no more hits */
int sort_match_indices(const void *aa, const void *bb, void *arg); /* exported by the reference too */

#ifdef __cplusplus
}
#endif

#endif /* SEQALIGN_COMPAT_H */
