/*
 * sa_alignment.c -- aligner_t / alignment_t and the printers (host, C).
 *
 * Interface: include/alignment.h (mirror of reference src/alignment.h).
 * aligner_align() is the drop-in boundary of the hot path: it keeps the
 * reference's post-condition (reference src/alignment.c:170-193) but the fill
 * itself runs on the GPU (seq-align_amd/csrc) -- see sa_fill_one_pair().
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "alignment.h"
#include "sa_internal.h"

const char align_col_mismatch[] = "\033[92m"; /* green  */
const char align_col_indel[]    = "\033[91m"; /* red    */
const char align_col_context[]  = "\033[95m"; /* pink   */
const char align_col_stop[]     = "\033[0m";

static void die_oom(const char *what)
{
  fprintf(stderr, "seqalign: out of memory (%s)\n", what);
  exit(EXIT_FAILURE);
}

void aligner_align(aligner_t *aligner,
                   const char *seq_a, const char *seq_b,
                   size_t len_a, size_t len_b,
                   const scoring_t *scoring, char is_sw)
{
  size_t cells = (len_a + 1) * (len_b + 1);
  uint64_t status = ~(uint64_t)0;
  int rc;

  aligner->scoring = scoring;
  aligner->seq_a = seq_a;
  aligner->seq_b = seq_b;
  aligner->score_width = len_a + 1;
  aligner->score_height = len_b + 1;

  if(aligner->capacity < cells) {
    /* grow-only, power-of-two capacity like the reference (alignment.c:183-190) */
    size_t cap = ROUNDUP2POW(cells), bytes = cap * sizeof(score_t);
    aligner->match_scores = realloc(aligner->match_scores, bytes);
    aligner->gap_a_scores = realloc(aligner->gap_a_scores, bytes);
    aligner->gap_b_scores = realloc(aligner->gap_b_scores, bytes);
    if(!aligner->match_scores || !aligner->gap_a_scores || !aligner->gap_b_scores)
      die_oom("score matrices");
    aligner->capacity = cap;
  }

  rc = sa_fill_one_pair(sa_default_ctx_or_die(), scoring, is_sw != 0,
                        seq_a, len_a, seq_b, len_b,
                        aligner->match_scores, aligner->gap_a_scores,
                        aligner->gap_b_scores, &status);
  if(rc == SEQALIGN_E_UNKNOWN_PAIR) {
    /* what scoring_lookup would have printed at that cell
     * (reference alignment_scoring.c:178-181) */
    size_t i = (size_t)(status % (len_a + 1)), j = (size_t)(status / (len_a + 1));
    fprintf(stderr, "Error: Unknown character pair (%c,%c) and "
                    "match/mismatch have not been set\n",
            sa_fold_char(scoring, seq_a[i - 1]), sa_fold_char(scoring, seq_b[j - 1]));
    exit(EXIT_FAILURE);
  }
  if(rc != SEQALIGN_OK) {
    fprintf(stderr, "seqalign: GPU fill failed: %s %s\n",
            seqalign_strerror(rc), seqalign_last_error());
    exit(EXIT_FAILURE);
  }
}

void aligner_destroy(aligner_t *aligner)
{
  if(aligner->capacity > 0) {
    free(aligner->match_scores);
    free(aligner->gap_a_scores);
    free(aligner->gap_b_scores);
  }
}

alignment_t* alignment_create(size_t capacity)
{
  alignment_t *r = malloc(sizeof(alignment_t));
  if(!r) die_oom("alignment_t");
  memset(r, 0, sizeof(*r));
  capacity = ROUNDUP2POW(capacity);
  if(capacity == 0) capacity = 1;
  r->result_a = malloc(capacity);
  r->result_b = malloc(capacity);
  if(!r->result_a || !r->result_b) die_oom("alignment strings");
  r->capacity = capacity;
  r->result_a[0] = r->result_b[0] = '\0';
  return r;
}

void alignment_ensure_capacity(alignment_t* r, size_t strlength)
{
  size_t need = strlength + 1;
  if(r->capacity >= need) return;
  need = ROUNDUP2POW(need);
  r->result_a = realloc(r->result_a, need);
  r->result_b = realloc(r->result_b, need);
  r->capacity = need;
  if(!r->result_a || !r->result_b) {
    fprintf(stderr, "%s:%i: Out of memory\n", __FILE__, __LINE__);
    exit(EXIT_FAILURE);
  }
}

void alignment_free(alignment_t* r)
{
  free(r->result_a);
  free(r->result_b);
  free(r);
}

void alignment_reverse_move(enum Matrix *curr_matrix, score_t *curr_score,
                            size_t *score_x, size_t *score_y,
                            size_t *arr_index, const aligner_t *aligner)
{
  sa_view_t v = { aligner->scoring, aligner->seq_a, aligner->seq_b,
                  aligner->score_width - 1, aligner->score_height - 1,
                  aligner->match_scores, aligner->gap_a_scores, aligner->gap_b_scores };
  int matrix = (int)*curr_matrix, rc;
  int32_t score = *curr_score;
  size_t x0 = *score_x, y0 = *score_y;

  rc = sa_reverse_move_rc(&v, &matrix, &score, score_x, score_y);
  if(rc == SEQALIGN_E_UNKNOWN_PAIR) {
    bool m; int s; /* prints + exits exactly like the reference lookup */
    scoring_lookup(aligner->scoring, aligner->seq_a[x0 - 1], aligner->seq_b[y0 - 1], &s, &m);
  }
  if(rc != SEQALIGN_OK) {
    /* reference alignment.c:328-349 */
    alignment_print_matrices(aligner);
    fprintf(stderr, "[%s:%zu,%zu]: %i '%c' '%c'\n", MATRIX_NAME(*curr_matrix),
            *score_x, *score_y, *curr_score,
            aligner->seq_a[x0 - 1], aligner->seq_b[y0 - 1]);
    fprintf(stderr,
"Program error: traceback fail (get_reverse_move)\n"
"This may be due to an integer overflow if your sequences are long or scores\n"
"are large. If this is the case using smaller scores or shorter sequences may\n"
"work around this problem.\n");
    exit(EXIT_FAILURE);
  }
  *curr_matrix = (enum Matrix)matrix;
  *curr_score = score;
  *arr_index = *score_y * aligner->score_width + *score_x;
}

static void print_one_matrix(const char *title, const score_t *m, size_t w, size_t h)
{
  size_t i, j;
  printf("%s:\n", title);
  for(j = 0; j < h; j++) {
    printf("%3i:", (int)j);
    for(i = 0; i < w; i++) printf("\t%3i", (int)m[j*w + i]);
    putc('\n', stdout);
  }
}

/* text format of reference alignment.c:353-400 (README.md:118-145) */
void alignment_print_matrices(const aligner_t *aligner)
{
  size_t w = aligner->score_width, h = aligner->score_height;
  printf("seq_a: %.*s\nseq_b: %.*s\n", (int)w - 1, aligner->seq_a,
         (int)h - 1, aligner->seq_b);
  print_one_matrix("match_scores", aligner->match_scores, w, h);
  print_one_matrix("gap_a_scores", aligner->gap_a_scores, w, h);
  print_one_matrix("gap_b_scores", aligner->gap_b_scores, w, h);
  printf("match: %i mismatch: %i gapopen: %i gapexend: %i\n",
         aligner->scoring->match, aligner->scoring->mismatch,
         aligner->scoring->gap_open, aligner->scoring->gap_extend);
  printf("\n");
}

/* reference alignment.c:402-449: a, coloured against b */
void alignment_colour_print_against(const char *alignment_a,
                                    const char *alignment_b,
                                    char case_sensitive)
{
  int in_indel = 0, in_mismatch = 0;
  size_t i;
  for(i = 0; alignment_a[i] != '\0'; i++) {
    char a = alignment_a[i], b = alignment_b[i];
    int differ = case_sensitive ? (a != b) : (tolower(a) != tolower(b));
    int indel = (b == '-');
    int mismatch = differ && a != '-' && b != '-';

    if(indel && !in_indel) fputs(align_col_indel, stdout);
    if(!indel && in_indel) fputs(align_col_stop, stdout);
    in_indel = indel;

    if(mismatch && !in_mismatch) fputs(align_col_mismatch, stdout);
    if(!mismatch && in_mismatch) fputs(align_col_stop, stdout);
    in_mismatch = mismatch;

    putc(a, stdout);
  }
  if(in_indel || in_mismatch) fputs(align_col_stop, stdout);
}

/* reference alignment.c:452-474: ' ' gap, '|' match, '*' mismatch */
void alignment_print_spacer(const char* alignment_a, const char* alignment_b,
                            const scoring_t* scoring)
{
  size_t i;
  for(i = 0; alignment_a[i] != '\0'; i++) {
    char a = alignment_a[i], b = alignment_b[i];
    if(a == '-' || b == '-') putc(' ', stdout);
    else if(a == b || (!scoring->case_sensitive && tolower(a) == tolower(b))) putc('|', stdout);
    else putc('*', stdout);
  }
}

/* CIGAR of an alignment (include/seqalign_hip.h): a derived format, the reference prints the gapped strings only */
size_t seqalign_cigar(const char *ra, const char *rb, size_t length, int extended, int case_insensitive,
                      char *out, size_t cap)
{
  size_t i = 0, used = 0;
  if(!ra || !rb || !out || cap == 0) return (size_t)-1;
  while(i < length) {
    char op;
    size_t run = 0;
    char tmp[24];
    int n;
    #define SA_OP(k) ((ra[k] == '-' && rb[k] == '-') ? '?' : ra[k] == '-' ? 'D' : rb[k] == '-' ? 'I' : !extended ? 'M' : \
                      ((case_insensitive ? tolower((unsigned char)ra[k]) == tolower((unsigned char)rb[k]) : ra[k] == rb[k]) ? '=' : 'X'))
    op = SA_OP(i);
    if(op == '?') return (size_t)-1;
    while(i < length && SA_OP(i) == op) { run++; i++; }
    #undef SA_OP
    n = snprintf(tmp, sizeof tmp, "%zu%c", run, op);
    if(n < 0 || used + (size_t)n + 1 > cap) return (size_t)-1;
    memcpy(out + used, tmp, (size_t)n);
    used += (size_t)n;
  }
  out[used] = '\0';
  return used;
}
