/* sa_internal.h -- private glue between the host C layer and the HIP layer. */
#ifndef SA_INTERNAL_H
#define SA_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "alignment.h"
#include "seqalign_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- non-exiting scoring lookup (sa_scoring.c) ---------------------------- */
/* 0 ok; 1 = pair has no score and use_match_mismatch is off. */
int sa_scoring_lookup_rc(const scoring_t *sc, int a, int b, int *score, int *is_match);
int sa_fold_char(const scoring_t *sc, int c);

/* ---- flattened scoring (sa_flatten.c) ------------------------------------- */
/* Sentinels in the class table / generic scores.  Neither can be a real score
 * inside the parity domain (floor + s must not wrap). */
#define SA_S_BLOCKED ((int32_t)INT32_MIN)       /* no_mismatches: M := floor   */
#define SA_S_UNKNOWN ((int32_t)(INT32_MIN + 1)) /* no score: reference exit()s */

enum {
  SA_F_NO_START_GAP = 1u << 0,
  SA_F_NO_END_GAP   = 1u << 1,
  SA_F_NO_GAPS_A    = 1u << 2,
  SA_F_NO_GAPS_B    = 1u << 3,
  SA_F_NO_MISMATCH  = 1u << 4,
  SA_F_IS_SW        = 1u << 5,
  SA_F_HAS_SENTINEL = 1u << 6   /* table or generic scores hold a sentinel   */
};

typedef struct {
  int32_t gap_open;     /* border cells: gap_open + k*gap_extend              */
  int32_t open1;        /* gap_open + gap_extend: first gap character         */
  int32_t ext;          /* gap_extend                                         */
  int32_t floor;        /* SW: 0; NW: INT_MIN + |min_penalty| (alignment.c:41)*/
  int32_t gen_eq;       /* generic x generic, same folded char                */
  int32_t gen_ne;       /* generic x generic, different folded chars          */
  uint32_t flags;       /* SA_F_*                                             */
  uint32_t n_classes;   /* K >= 1; class 0 = "generic" (no table entries)     */
  uint16_t code[256];   /* raw char -> folded char | class << 8               */
  int32_t *table;       /* [K*K] row = class of seq_a char; malloc'ed          */
} sa_flat_scoring_t;

/* SEQALIGN_OK / SEQALIGN_E_DOMAIN / SEQALIGN_E_NOMEM */
int sa_flatten_scoring(const scoring_t *sc, int is_sw, sa_flat_scoring_t *out);
void sa_flat_scoring_free(sa_flat_scoring_t *f);

/* ---- process-wide default context for the legacy single-pair API ----------- */
/* Created on first use on device 0 (or $SEQALIGN_DEVICE); prints + exit()s when
 * no device is present -- there is no CPU fallback. */
seqalign_ctx_t *sa_default_ctx_or_die(void);

/* One pair through the device: used by aligner_align(). */
int sa_fill_one_pair(seqalign_ctx_t *ctx, const scoring_t *sc, int is_sw,
                     const char *a, size_t len_a, const char *b, size_t len_b,
                     int32_t *M, int32_t *A, int32_t *B, uint64_t *status);

/* ---- host traceback over given matrices (sa_traceback.c) ------------------- */
typedef struct {
  const scoring_t *sc;
  const char *a, *b;
  size_t len_a, len_b;
  const int32_t *M, *A, *B;
} sa_view_t;

/* SEQALIGN_OK / SEQALIGN_E_TRACEBACK / SEQALIGN_E_UNKNOWN_PAIR */
int sa_reverse_move_rc(const sa_view_t *v, int *matrix, int32_t *score,
                       size_t *x, size_t *y);
int sa_nw_traceback(const sa_view_t *v, char *out_a, char *out_b,
                    size_t *out_len, int32_t *out_score);

/* ---- alignments as bit planes (sa_moves.c) ---------------------------------- */
/* What the device walkers on direction bytes send home instead of the two gapped strings: per walked column one bit
 * "gap in seq_a" and one bit "gap in seq_b", forward column order, right-aligned in n_words uint32 per plane. */
int sa_expand_nw_moves(const char *a, uint32_t len_a, const char *b, uint32_t len_b, const uint32_t *plane_a,
                       const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, char *out_a, char *out_b,
                       uint32_t *out_len);
int sa_expand_sw_moves(const char *a, const char *b, uint32_t end_x, uint32_t end_y, const uint32_t *plane_a,
                       const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, char *out_a, char *out_b,
                       uint32_t pos[4]);
int sa_moves_uses_simd(void);
void sa_moves_force_scalar(int on);
/* CIGAR straight from the planes (seq_a = query, seq_b = reference: a gap in result_b is I, a gap in result_a is D; format 1: M / I / D,
 * 2: = / X / I / D with the letters compared as they are or, fold != 0, case-folded) -- the same text seqalign_cigar makes of the expanded
 * strings, without expanding them.  out == NULL: only the length.  Return SEQALIGN_OK, SEQALIGN_E_TRACEBACK (the planes describe no walk
 * over these lengths) or SEQALIGN_E_NOMEM (cap, counted with the NUL, is too small); *out_len = strlen of the CIGAR. */
int sa_cigar_nw_moves(const char *a, uint32_t len_a, const char *b, uint32_t len_b, const uint32_t *plane_a,
                      const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, int format, int fold, char *out, uint64_t cap,
                      uint32_t *out_len, uint32_t *out_columns);
int sa_cigar_sw_moves(const char *a, const char *b, uint32_t end_x, uint32_t end_y, const uint32_t *plane_a,
                      const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, int format, int fold, char *out, uint64_t cap,
                      uint32_t pos[4], uint32_t *out_len);

/* SW hit enumeration over candidate cells (index list, any order; sorted here).
 * `seen` is a caller-provided zeroed bitmap of W*H bits. */
typedef int (*sa_hit_sink_t)(void *user, int32_t score, size_t end_x, size_t end_y,
                             size_t x, size_t y, size_t steps);

#ifdef __cplusplus
}
#endif
#endif
