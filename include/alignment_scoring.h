/*
 * alignment_scoring.h -- scoring model of the aligner (host side, C).
 *
 * Mirrors the public interface of the reference header
 * src/alignment_scoring.h:16-81 so that code written against seq-align keeps
 * compiling: same type names, same field order (the struct is part of the
 * boundary -- callers fill and read it directly), same function names and
 * argument meaning.  Implementation: seq-align_amd/host/sa_scoring.c.
 *
 * A gap of length N costs gap_open + N*gap_extend (src/alignment_scoring.c:28-29).
 */
#ifndef ALIGNMENT_SCORING_HEADER_SEEN
#define ALIGNMENT_SCORING_HEADER_SEEN

#include <inttypes.h>
#include <limits.h>
#include <stdbool.h>
#include <stddef.h>

typedef int score_t;
#define SCORE_MIN INT_MIN

typedef struct
{
  int gap_open, gap_extend;

  /* free gaps before the first / after the last aligned base */
  bool no_start_gap_penalty, no_end_gap_penalty;

  /* forbid gaps in a / in b / aligned mismatches */
  bool no_gaps_in_a, no_gaps_in_b, no_mismatches;

  /* fall back to match/mismatch for pairs missing from swap_scores */
  bool use_match_mismatch;
  int match, mismatch;

  bool case_sensitive;

  /* wildcards: bitset of chars that pair with anything for wildscores[c];
   * swap_set[a] : bitset of b for which swap_scores[a][b] is defined */
  uint32_t wildcards[256/32], swap_set[256][256/32];
  score_t wildscores[256], swap_scores[256][256];

  /* smallest / largest of all penalties seen by the builders; min_penalty
   * fixes the NW floor INT_MIN+|min_penalty| (src/alignment.c:41) */
  int min_penalty, max_penalty;
} scoring_t;

#ifndef bitset32_get
  #define bitset32_get(arr,idx)   (((arr)[(idx)>>5] >> ((idx)&31)) & 0x1)
  #define bitset32_set(arr,idx)   ((arr)[(idx)>>5] |=   (1<<((idx)&31)))
  #define bitset32_clear(arr,idx) ((arr)[(idx)>>5] &=  ~(1<<((idx)&31)))
#endif

#define get_wildcard_bit(scoring,c) bitset32_get((scoring)->wildcards,c)
#define set_wildcard_bit(scoring,c) bitset32_set((scoring)->wildcards,c)
#define get_swap_bit(scoring,a,b) bitset32_get((scoring)->swap_set[(size_t)(a)],b)
#define set_swap_bit(scoring,a,b) bitset32_set((scoring)->swap_set[(size_t)(a)],b)
#define scoring_is_wildcard(scoring,c) (get_wildcard_bit(scoring,c))

#ifdef __cplusplus
extern "C" {
#endif

void scoring_init(scoring_t* scoring, int match, int mismatch,
                  int gap_open, int gap_extend,
                  bool no_start_gap_penalty, bool no_end_gap_penalty,
                  bool no_gaps_in_a, bool no_gaps_in_b,
                  bool no_mismatches, bool case_sensitive);

void scoring_add_wildcard(scoring_t* scoring, char c, int s);
void scoring_add_mutation(scoring_t* scoring, char a, char b, int score);
/* scores[j*strlen(str)+i] is the score of (str[i], str[j]) */
void scoring_add_mutations(scoring_t* scoring, const char *str, const int *scores,
                           char use_match_mismatch);

void scoring_print(const scoring_t* scoring);

/* Always sets *score and *is_match; prints and exit(EXIT_FAILURE)s on a pair
 * without a score when use_match_mismatch is off (reference behaviour). */
void scoring_lookup(const scoring_t* scoring, char a, char b,
                    int *score, bool *is_match);

/* Built-in systems (reference src/alignment_scoring.c:307-392) */
void scoring_system_PAM30(scoring_t *scoring);
void scoring_system_PAM70(scoring_t *scoring);
void scoring_system_BLOSUM80(scoring_t *scoring);
void scoring_system_BLOSUM62(scoring_t *scoring);
void scoring_system_DNA_hybridization(scoring_t *scoring);
void scoring_system_default(scoring_t *scoring);

/* 24x24 BLOSUM62 over "ARNDCQEGHILKMFPSTWYVBZX*"; exported by the reference
 * library too (src/alignment_scoring.c:268). */
extern int blosum62[576];

#ifdef __cplusplus
}
#endif

#endif
