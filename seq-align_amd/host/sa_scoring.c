/*
 * sa_scoring.c -- scoring_t builders, lookup and built-in systems (host, C).
 *
 * Fresh implementation of the interface in include/alignment_scoring.h; the
 * behaviour follows the reference src/alignment_scoring.c (cited per function)
 * and is pinned by tests/golden/presets.json + fill_small.json.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "alignment_scoring.h"
#include "sa_internal.h"
#include "sa_preset_tables.h"

static int lower_ascii(int c) { return (c >= 'A' && c <= 'Z') ? c + ('a' - 'A') : c; }

int sa_fold_char(const scoring_t *sc, int c)
{
  /* reference: tolower() unless case_sensitive (alignment_scoring.c:136-140).
   * Bytes are taken as unsigned; the reference indexes its tables with a
   * (possibly negative) char, i.e. is only defined for 7-bit ASCII. */
  c &= 0xFF;
  return sc->case_sensitive ? c : lower_ascii(c);
}

static void widen_penalty_range(scoring_t *sc, int v)
{
  if(v < sc->min_penalty) sc->min_penalty = v;
  if(v > sc->max_penalty) sc->max_penalty = v;
}

/* reference alignment_scoring.c:21-55 */
void scoring_init(scoring_t *sc, int match, int mismatch,
                  int gap_open, int gap_extend,
                  bool no_start_gap_penalty, bool no_end_gap_penalty,
                  bool no_gaps_in_a, bool no_gaps_in_b,
                  bool no_mismatches, bool case_sensitive)
{
  sc->gap_open = gap_open;
  sc->gap_extend = gap_extend;
  sc->no_start_gap_penalty = no_start_gap_penalty;
  sc->no_end_gap_penalty = no_end_gap_penalty;
  sc->no_gaps_in_a = no_gaps_in_a;
  sc->no_gaps_in_b = no_gaps_in_b;
  sc->no_mismatches = no_mismatches;
  sc->use_match_mismatch = true;
  sc->match = match;
  sc->mismatch = mismatch;
  sc->case_sensitive = case_sensitive;

  memset(sc->wildcards, 0, sizeof(sc->wildcards));
  memset(sc->swap_set, 0, sizeof(sc->swap_set));

  sc->min_penalty = sc->max_penalty = match;
  widen_penalty_range(sc, mismatch);
  /* gap penalties only count when some gap is allowed (reference :51-54) */
  if(!no_gaps_in_a || !no_gaps_in_b) {
    widen_penalty_range(sc, gap_open + gap_extend);
    widen_penalty_range(sc, gap_extend);
  }
}

/* reference alignment_scoring.c:57-64 */
void scoring_add_wildcard(scoring_t *sc, char c, int s)
{
  int k = sa_fold_char(sc, c);
  set_wildcard_bit(sc, k);
  sc->wildscores[k] = s;
  widen_penalty_range(sc, s);
}

/* reference alignment_scoring.c:66-72 (no case folding at this level) */
void scoring_add_mutation(scoring_t *sc, char a, char b, int score)
{
  unsigned char ua = (unsigned char)a, ub = (unsigned char)b;
  sc->swap_scores[ua][ub] = score;
  set_swap_bit(sc, ua, ub);
  widen_penalty_range(sc, score);
}

/* reference alignment_scoring.c:74-95 */
void scoring_add_mutations(scoring_t *sc, const char *str, const int *scores,
                           char use_match_mismatch)
{
  size_t n = strlen(str), i, j;
  for(i = 0; i < n; i++)
    for(j = 0; j < n; j++)
      scoring_add_mutation(sc, (char)sa_fold_char(sc, str[i]),
                           (char)sa_fold_char(sc, str[j]), scores[j*n + i]);
  sc->use_match_mismatch = use_match_mismatch;
}

/* reference alignment_scoring.c:97-111 */
void scoring_print(const scoring_t *sc)
{
  printf("scoring:\n");
  printf("  match: %i; mismatch: %i; (use_match_mismatch: %i)\n",
         sc->match, sc->mismatch, sc->use_match_mismatch);
  printf("  gap_open: %i; gap_extend: %i;\n", sc->gap_open, sc->gap_extend);
  printf("  no_gaps_in_a: %i; no_gaps_in_b: %i; no_mismatches: %i;\n",
         sc->no_gaps_in_a, sc->no_gaps_in_b, sc->no_mismatches);
  printf("  no_start_gap_penalty: %i; no_end_gap_penalty: %i;\n",
         sc->no_start_gap_penalty, sc->no_end_gap_penalty);
}

/* min wildscore over whichever of a,b is a wildcard (reference :115-129) */
static int either_wildcard(const scoring_t *sc, int a, int b, int *score)
{
  int wa = get_wildcard_bit(sc, a), wb = get_wildcard_bit(sc, b);
  if(wa && wb) *score = sc->wildscores[a] < sc->wildscores[b] ? sc->wildscores[a] : sc->wildscores[b];
  else if(wa) *score = sc->wildscores[a];
  else if(wb) *score = sc->wildscores[b];
  else *score = 0;
  return wa || wb;
}

/* Decision order of reference alignment_scoring.c:133-182, returning a code
 * instead of exiting so the batch path can report SEQALIGN_E_UNKNOWN_PAIR. */
int sa_scoring_lookup_rc(const scoring_t *sc, int a, int b, int *score, int *is_match)
{
  a = sa_fold_char(sc, a);
  b = sa_fold_char(sc, b);
  *is_match = (a == b);

  if(sc->no_mismatches && a != b) {            /* only wildcards may pair up */
    *is_match = either_wildcard(sc, a, b, score);
    return 0;
  }
  if(get_swap_bit(sc, a, b)) { *score = sc->swap_scores[a][b]; return 0; }
  if(either_wildcard(sc, a, b, score)) { *is_match = 1; return 0; }
  if(sc->use_match_mismatch) { *score = *is_match ? sc->match : sc->mismatch; return 0; }
  *score = 0;
  return 1;
}

void scoring_lookup(const scoring_t *sc, char a, char b, int *score, bool *is_match)
{
  int same;
  if(sa_scoring_lookup_rc(sc, a, b, score, &same) != 0) {
    /* reference alignment_scoring.c:178-181 */
    fprintf(stderr, "Error: Unknown character pair (%c,%c) and "
                    "match/mismatch have not been set\n",
            sa_fold_char(sc, a), sa_fold_char(sc, b));
    exit(EXIT_FAILURE);
  }
  *is_match = same != 0;
}

/* ---- built-in systems (reference alignment_scoring.c:307-392) ------------- */

static void load_preset(scoring_t *sc, const sa_preset_t *p)
{
  size_t n = strlen(p->letters), i, j;
  /* all presets: no free end gaps, no restrictions, case-insensitive */
  scoring_init(sc, p->match, p->mismatch, p->gap_open, p->gap_extend,
               0, 0, 0, 0, 0, 0);
  for(i = 0; i < n; i++)
    for(j = 0; j < n; j++)
      scoring_add_mutation(sc, p->letters[i], p->letters[j],
                           (int)p->pairs[i*n + j] - SA_PRESET_BIAS);
  sc->use_match_mismatch = p->use_match_mismatch;
}

void scoring_system_PAM30(scoring_t *sc)    { load_preset(sc, &sa_preset_PAM30); }
void scoring_system_PAM70(scoring_t *sc)    { load_preset(sc, &sa_preset_PAM70); }
void scoring_system_BLOSUM80(scoring_t *sc) { load_preset(sc, &sa_preset_BLOSUM80); }
void scoring_system_BLOSUM62(scoring_t *sc) { load_preset(sc, &sa_preset_BLOSUM62); }
void scoring_system_DNA_hybridization(scoring_t *sc) { load_preset(sc, &sa_preset_DNA_hybridization); }
void scoring_system_default(scoring_t *sc)  { load_preset(sc, &sa_preset_default); }

/* The reference exports its BLOSUM62 table (non-static, alignment_scoring.c:268)
 * in NCBI letter order; rebuilt here from the preset data at load time. */
int blosum62[576];

__attribute__((constructor)) static void fill_exported_blosum62(void)
{
  static const char ncbi[] = "arndcqeghilkmfpstwyvbzx*";
  const sa_preset_t *p = &sa_preset_BLOSUM62;
  size_t n = strlen(p->letters), i, j;
  for(i = 0; i < 24; i++) {
    for(j = 0; j < 24; j++) {
      size_t pi = (size_t)(strchr(p->letters, ncbi[i]) - p->letters);
      size_t pj = (size_t)(strchr(p->letters, ncbi[j]) - p->letters);
      /* table[j*24+i] is the score of (letter i, letter j) */
      blosum62[j*24 + i] = (int)p->pairs[pi*n + pj] - SA_PRESET_BIAS;
    }
  }
}
