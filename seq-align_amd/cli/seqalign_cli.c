/*
 * seqalign_cli.c -- batch command-line front-end: `seqalign_nw` (global) and
 * `seqalign_sw` (local), SURVEY 8f-3.
 *
 * Same options and the same output text as the reference tools
 * (src/tools/nw_cmdline.c:78-149, src/tools/sw_cmdline.c:125-314, option set of
 * src/alignment_cmdline.c:179-532), but pairs are not aligned one at a time:
 * they are collected and go through seqalign_nw_batch / seqalign_sw_batch in
 * batches, so the GPU sees thousands of pairs per launch.  Differences, all
 * deliberate: (1) --match/--mismatch/--gapopen/--gapextend also lower
 * min_penalty, so the NW floor stays defined (upstream leaves it stale: UB,
 * SURVEY A.3-3); (2) every pair's local hits come from a fresh visited mask
 * (SURVEY A.3-2); (3) no interactive stepping through the hits (every hit is
 * printed, as upstream does for non-interactive input).  Sequence files may be
 * gzip-compressed (zlib, like upstream's seq_file); --zam is provided for the global
 * tool.  --printmatrices uses the per-pair API (it needs the matrices on the host).
 *
 * Round 5: a PIPELINE of three threads over a ring of batches.  The reference's driver has one pair in flight
 * (src/alignment_cmdline.c:578-640: read two records, align, print, repeat); rounds 3-4 here read 65 536 pairs, aligned
 * them, printed them, one after the other, with four mallocs per record.  Now
 *     reader  : parses records straight into the batch's ONE growing text arena (names and sequences, NUL-terminated;
 *               the sequences' offsets / lengths are the seqalign_batch_t arrays) -- batch k + 1
 *     aligner : seqalign_nw_batch / seqalign_sw_batch on the GPU                 -- batch k
 *     printer : (the main thread) writes the reference's text                     -- batch k - 1
 * run side by side; a batch's buffers are reused when it comes round again.  Output order is input order.
 */
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "seqalign_hip.h"
#include "seqalign_io.h"

enum { TOOL_NW, TOOL_SW };

typedef struct {
  int tool;
  int case_sensitive, print_scores, print_seq, print_matrices, print_fasta, print_pretty, print_colour, zam;
  int cigar;   /* 0, SEQALIGN_CIGAR_M (--cigar) or SEQALIGN_CIGAR_EQX (--cigarx): an output of this tool only (the reference has no CIGAR) */
  int min_score, min_score_set;
  unsigned max_hits; int max_hits_set;
  unsigned context;
  const char *seq1, *seq2;
  const char *files1[64], *files2[64];
  int n_files;
} opts_t;

typedef struct { const char *name, *seq; size_t len; } rec_t;   /* a view into a batch's text arena */

static scoring_t scoring;   /* 271 KB */
static opts_t opt;

static void die(const char *msg, const char *arg)
{
  fprintf(stderr, "Error: ");
  fprintf(stderr, msg, arg);
  fprintf(stderr, "\nusage: %s [OPTIONS] [seq1 seq2]   (options as in seq-align's %s; --help lists them)\n",
          opt.tool == TOOL_NW ? "seqalign_nw" : "seqalign_sw",
          opt.tool == TOOL_NW ? "needleman_wunsch" : "smith_waterman");
  exit(EXIT_FAILURE);
}

static void oom(void) { fprintf(stderr, "out of memory\n"); exit(EXIT_FAILURE); }

static int parse_int(const char *s, int *out)
{
  char *end;
  long v = strtol(s, &end, 10);
  if(end == s || *end) return 0;
  *out = (int)v;
  return 1;
}

/* One batch on its way through the pipeline.  Input: `text` holds, per pair, name_a\0 seq_a\0 name_b\0 seq_b\0 in reading
 * order -- one growing arena instead of four mallocs per record -- and is the seqalign_batch_t's arena (off_a / off_b point
 * at the sequences).  Results: the aligner's output buffers.  All arrays are kept and reused when the batch comes round. */
typedef struct {
  char *text; size_t text_len, text_cap;
  size_t *name_a, *name_b;
  uint64_t *off_a, *off_b; uint32_t *len_a, *len_b;
  size_t n, cap;
  uint64_t cells;                         /* sum of (len_a + 1)(len_b + 1): a batch also ends at BATCH_CELLS */
  int last;                               /* the reader's last batch: end of input */
  /* results */
  uint64_t *str_off; uint32_t *out_len; int32_t *score;     /* NW */
  int32_t *min_score; seqalign_sw_hit_t *hits; uint64_t n_hits, hit_cap;   /* SW */
  char *out_a, *out_b; size_t out_cap, res_cap;
} batch_t;

static void *grow(void *p, size_t bytes) { p = realloc(p, bytes); if(!p) oom(); return p; }

static size_t text_add(batch_t *b, const char *s, size_t n)   /* appends s\0, returns its offset */
{
  const size_t at = b->text_len;
  if(at + n + 1 > b->text_cap) {
    b->text_cap = b->text_cap ? 2 * b->text_cap : (size_t)1 << 20;
    while(at + n + 1 > b->text_cap) b->text_cap *= 2;
    b->text = grow(b->text, b->text_cap);
  }
  memcpy(b->text + at, s, n);
  b->text[at + n] = '\0';
  b->text_len = at + n + 1;
  return at;
}

/* the first record of a pair goes in when it is read (the reader's buffers are only valid until its next call) */
static void batch_add_a(batch_t *b, const char *na, const char *sa, size_t la)
{
  if(b->n == b->cap) {
    b->cap = b->cap ? 2 * b->cap : 4096;
    b->name_a = grow(b->name_a, b->cap * sizeof(size_t)); b->name_b = grow(b->name_b, b->cap * sizeof(size_t));
    b->off_a = grow(b->off_a, b->cap * sizeof(uint64_t)); b->off_b = grow(b->off_b, b->cap * sizeof(uint64_t));
    b->len_a = grow(b->len_a, b->cap * sizeof(uint32_t)); b->len_b = grow(b->len_b, b->cap * sizeof(uint32_t));
  }
  b->name_a[b->n] = text_add(b, na, strlen(na));
  b->off_a[b->n] = text_add(b, sa, la); b->len_a[b->n] = (uint32_t)la;
}
static void batch_add_b(batch_t *b, const char *nb, const char *sb, size_t lb)
{
  b->name_b[b->n] = text_add(b, nb, strlen(nb));
  b->off_b[b->n] = text_add(b, sb, lb); b->len_b[b->n] = (uint32_t)lb;
  b->cells += ((uint64_t)b->len_a[b->n] + 1) * ((uint64_t)lb + 1);
  b->n++;
}
static void batch_drop_a(batch_t *b) { b->text_len = b->name_a[b->n]; }   /* an odd record at end of file */

static rec_t rec_a(const batch_t *b, size_t i) { rec_t r = { b->text + b->name_a[i], b->text + b->off_a[i], b->len_a[i] }; return r; }
static rec_t rec_b(const batch_t *b, size_t i) { rec_t r = { b->text + b->name_b[i], b->text + b->off_b[i], b->len_b[i] }; return r; }

/* ---------------------------------------------------------------- options */

static void parse_args(int argc, char **argv)
{
  int i, scoring_set = 0, subst_set = 0, match_set = 0, mismatch_set = 0;
  if(argc == 1) die("No input specified%s", "");

  /* pass 1: case sensitivity and the scoring system (alignment_cmdline.c:203-250) */
  for(i = 1; i < argc; i++) {
    if(!strcasecmp(argv[i], "--case_sensitive")) opt.case_sensitive = 1;
    else if(!strcasecmp(argv[i], "--scoring") && i + 1 < argc) {
      const char *n = argv[++i];
      if(scoring_set) die("More than one scoring system specified - not permitted%s", "");
      if(!strcasecmp(n, "PAM30")) scoring_system_PAM30(&scoring);
      else if(!strcasecmp(n, "PAM70")) scoring_system_PAM70(&scoring);
      else if(!strcasecmp(n, "BLOSUM80")) scoring_system_BLOSUM80(&scoring);
      else if(!strcasecmp(n, "BLOSUM62")) scoring_system_BLOSUM62(&scoring);
      else if(!strcasecmp(n, "DNA_HYBRIDIZATION")) scoring_system_DNA_hybridization(&scoring);
      else die("Unknown --scoring choice '%s'", n);
      scoring_set = 1;
    }
  }
  /* pass 2 (alignment_cmdline.c:252-485) */
  for(i = 1; i < argc; i++) {
    const char *a = argv[i];
    if(a[0] != '-' || !strcmp(a, "-")) {
      if(argc - i != 2) die("Unknown options: '%s'", a);
      opt.seq1 = argv[i]; opt.seq2 = argv[i+1];
      break;
    }
    if(!strcasecmp(a, "--case_sensitive")) continue;
    else if(!strcasecmp(a, "--scoring")) i++;
    else if(!strcasecmp(a, "--freestartgap")) { if(opt.tool != TOOL_NW) die("--freestartgap only valid with Needleman-Wunsch%s", ""); scoring.no_start_gap_penalty = 1; }
    else if(!strcasecmp(a, "--freeendgap")) { if(opt.tool != TOOL_NW) die("--freeendgap only valid with Needleman-Wunsch%s", ""); scoring.no_end_gap_penalty = 1; }
    else if(!strcasecmp(a, "--nogaps")) scoring.no_gaps_in_a = scoring.no_gaps_in_b = 1;
    else if(!strcasecmp(a, "--nogapsin1")) scoring.no_gaps_in_a = 1;
    else if(!strcasecmp(a, "--nogapsin2")) scoring.no_gaps_in_b = 1;
    else if(!strcasecmp(a, "--nomismatches")) scoring.no_mismatches = 1;
    else if(!strcasecmp(a, "--printscores")) { if(opt.tool != TOOL_NW) die("--printscores only valid with Needleman-Wunsch%s", ""); opt.print_scores = 1; }
    else if(!strcasecmp(a, "--printseq")) { if(opt.tool != TOOL_SW) die("--printseq only valid with Smith-Waterman%s", ""); opt.print_seq = 1; }
    else if(!strcasecmp(a, "--printmatrices")) opt.print_matrices = 1;
    else if(!strcasecmp(a, "--printfasta")) opt.print_fasta = 1;
    else if(!strcasecmp(a, "--pretty")) opt.print_pretty = 1;
    else if(!strcasecmp(a, "--colour")) opt.print_colour = 1;
    else if(!strcasecmp(a, "--cigar")) opt.cigar = SEQALIGN_CIGAR_M;
    else if(!strcasecmp(a, "--cigarx")) opt.cigar = SEQALIGN_CIGAR_EQX;
    else if(!strcasecmp(a, "--zam")) { if(opt.tool != TOOL_NW) die("--zam only valid with Needleman-Wunsch%s", ""); opt.zam = 1; }
    else if(!strcasecmp(a, "--stdin")) { opt.files1[opt.n_files] = "-"; opt.files2[opt.n_files++] = NULL; }
    else if(i + 1 >= argc) die("%s takes an argument", a);
    else if(!strcasecmp(a, "--match")) { if(!parse_int(argv[++i], &scoring.match)) die("Invalid --match argument ('%s') must be an int", argv[i]); match_set = 1; }
    else if(!strcasecmp(a, "--mismatch")) { if(!parse_int(argv[++i], &scoring.mismatch)) die("Invalid --mismatch argument ('%s') must be an int", argv[i]); mismatch_set = 1; }
    else if(!strcasecmp(a, "--gapopen")) { if(!parse_int(argv[++i], &scoring.gap_open)) die("Invalid --gapopen argument ('%s') must be an int", argv[i]); }
    else if(!strcasecmp(a, "--gapextend")) { if(!parse_int(argv[++i], &scoring.gap_extend)) die("Invalid --gapextend argument ('%s') must be an int", argv[i]); }
    else if(!strcasecmp(a, "--minscore")) { if(opt.tool != TOOL_SW) die("--minscore only valid with Smith-Waterman%s", ""); if(!parse_int(argv[++i], &opt.min_score)) die("Invalid --minscore '%s'", argv[i]); opt.min_score_set = 1; }
    else if(!strcasecmp(a, "--maxhits")) { int v; if(opt.tool != TOOL_SW) die("--maxhits only valid with Smith-Waterman%s", ""); if(!parse_int(argv[++i], &v) || v < 0) die("Invalid --maxhits '%s'", argv[i]); opt.max_hits = (unsigned)v; opt.max_hits_set = 1; }
    else if(!strcasecmp(a, "--context")) { int v; if(opt.tool != TOOL_SW) die("--context only valid with Smith-Waterman%s", ""); if(!parse_int(argv[++i], &v) || v < 0) die("Invalid --context '%s'", argv[i]); opt.context = (unsigned)v; }
    else if(!strcasecmp(a, "--file")) { opt.files1[opt.n_files] = argv[++i]; opt.files2[opt.n_files++] = NULL; }
    else if(!strcasecmp(a, "--files")) {
      if(i + 2 >= argc) die("--files option takes 2 arguments%s", "");
      opt.files1[opt.n_files] = argv[i+1];
      opt.files2[opt.n_files++] = (!strcmp(argv[i+1], "-") && !strcmp(argv[i+2], "-")) ? NULL : argv[i+2];
      i += 2;
    }
    else if(!strcasecmp(a, "--wildcard")) {
      int w;
      if(i + 2 >= argc || strlen(argv[i+1]) != 1 || !parse_int(argv[i+2], &w)) die("--wildcard <w> <s> takes a single character and a number%s", "");
      scoring_add_wildcard(&scoring, argv[i+1][0], w);
      i += 2;
    }
    else if(!strcasecmp(a, "--substitution_matrix") || !strcasecmp(a, "--substitution_pairs")) {
      char err[256];
      FILE *f = fopen(argv[++i], "r");
      int rc;
      if(!f) die("Couldn't read file: %s", argv[i]);
      rc = !strcasecmp(a, "--substitution_matrix") ? seqalign_scoring_load_matrix(f, &scoring, opt.case_sensitive, err, sizeof err)
                                                    : seqalign_scoring_load_pairs(f, &scoring, opt.case_sensitive, err, sizeof err);
      fclose(f);
      if(rc) { fprintf(stderr, "Error: %s\nFile: %s\n", err, argv[i]); exit(EXIT_FAILURE); }
      subst_set = 1;
    }
    else die("Unknown argument '%s'", a);
    if(opt.n_files >= 64) die("too many input files%s", "");
  }
  if((match_set && !mismatch_set && !scoring.no_mismatches) || (!match_set && mismatch_set))
    die("--match --mismatch must both be set or neither set%s", "");
  if(subst_set && !match_set) scoring.use_match_mismatch = 0;
  if(scoring.use_match_mismatch && scoring.match < scoring.mismatch) die("Match value should not be less than mismatch penalty%s", "");
  if(opt.tool == TOOL_NW && scoring.no_mismatches && (scoring.no_gaps_in_a || scoring.no_gaps_in_b))
    die("--nogaps.. --nomismatches cannot be used at together%s", "");
  if(!opt.seq1 && !opt.n_files) die("No input specified%s", "");
  if(opt.cigar && (opt.zam || opt.print_pretty || opt.print_colour || opt.print_matrices || opt.print_seq || opt.context))
    die("--cigar / --cigarx print one tab-separated line per alignment: not with --zam, --pretty, --colour, --printmatrices, --printseq or --context%s", "");
  if(opt.zam && (opt.print_pretty || opt.print_scores || opt.print_colour || opt.print_fasta))
    die("Cannot use --printscore, --printfasta, --pretty or --colour with --zam%s", "");

  /* keep min/max_penalty covering every penalty in use (see file header) */
  {
    int vals[4] = { scoring.match, scoring.mismatch, scoring.gap_open + scoring.gap_extend, scoring.gap_extend }, k;
    for(k = 0; k < 4; k++) {
      if(vals[k] < scoring.min_penalty) scoring.min_penalty = vals[k];
      if(vals[k] > scoring.max_penalty) scoring.max_penalty = vals[k];
    }
  }
}

/* ---------------------------------------------------------------- printing */

static void put_line(const char *mine, const char *other)
{
  if(opt.print_colour) alignment_colour_print_against(mine, other, scoring.case_sensitive);
  else fputs(mine, stdout);
}

/* --zam (nw_cmdline.c:36-76): gaps as '_', a spacer line (' ' indel, '*' mismatch, '|' match), then the
   number of mismatches and of indel columns */
static void print_zam(const char *res_a, const char *res_b)
{
  size_t i, mismatches = 0, indels = 0;
  fputs("Br1:", stdout);
  for(i = 0; res_a[i]; i++) putc(res_a[i] == '-' ? '_' : res_a[i], stdout);
  fputs("\n    ", stdout);
  for(i = 0; res_a[i]; i++) {
    const char x = res_a[i], y = res_b[i];
    if(x == '-' || y == '-') { putc(' ', stdout); indels++; }
    else if((scoring.case_sensitive && x != y) || tolower((unsigned char)x) != tolower((unsigned char)y)) { putc('*', stdout); mismatches++; }
    else putc('|', stdout);
  }
  fputs("\nBr2:", stdout);
  for(i = 0; res_b[i]; i++) putc(res_b[i] == '-' ? '_' : res_b[i], stdout);
  printf("\n%zu %zu\n\n", mismatches, indels);
}

/* nw_cmdline.c:78-149 */
static void print_nw(const rec_t *ra, const rec_t *rb, const char *res_a, const char *res_b, int score)
{
  if(opt.zam) { print_zam(res_a, res_b); return; }
  const char *na = ra->name[0] ? ra->name : NULL, *nb = rb->name[0] ? rb->name : NULL;
  if(opt.print_fasta && na) { fputs(na, stdout); putc('\n', stdout); }
  if(opt.print_fasta && opt.print_pretty && nb) { fputs(nb, stdout); putc('\n', stdout); }
  put_line(res_a, res_b); putc('\n', stdout);
  if(opt.print_pretty) { alignment_print_spacer(res_a, res_b, &scoring); putc('\n', stdout); }
  else if(opt.print_fasta && nb) { fputs(nb, stdout); putc('\n', stdout); }
  put_line(res_b, res_a); putc('\n', stdout);
  if(opt.print_scores) printf("score: %i\n", score);
  putc('\n', stdout);
}

/* sw_cmdline.c:60-93 */
static void print_sw_part(const char *mine, const char *other, size_t pos, size_t len, const char *whole,
                          size_t spaces_left, size_t spaces_right, size_t ctx_left, size_t ctx_right)
{
  size_t i;
  printf("  ");
  for(i = 0; i < spaces_left; i++) putc(' ', stdout);
  if(ctx_left > 0) {
    if(opt.print_colour) fputs(align_col_context, stdout);
    printf("%.*s", (int)ctx_left, whole + pos - ctx_left);
    if(opt.print_colour) fputs(align_col_stop, stdout);
  }
  put_line(mine, other);
  if(ctx_right > 0) {
    if(opt.print_colour) fputs(align_col_context, stdout);
    printf("%.*s", (int)ctx_right, whole + pos + len);
    if(opt.print_colour) fputs(align_col_stop, stdout);
  }
  for(i = 0; i < spaces_right; i++) putc(' ', stdout);
  printf("  [pos: %li; len: %lu]\n", (long)pos, (unsigned long)len);
}

#define MAX2(x,y) ((x) >= (y) ? (x) : (y))
#define MIN2(x,y) ((x) <= (y) ? (x) : (y))

/* sw_cmdline.c:152-201 (per-pair heading) */
static void print_sw_heading(size_t index, const rec_t *ra, const rec_t *rb)
{
  printf("== Alignment %zu lengths (%lu, %lu):\n", index, (unsigned long)ra->len, (unsigned long)rb->len);
  if(opt.print_fasta && ra->name[0]) { fputs(ra->name, stdout); putc('\n', stdout); }
  if(opt.print_seq) { fputs(ra->seq, stdout); putc('\n', stdout); }
  if(opt.print_fasta && rb->name[0]) { fputs(rb->name, stdout); putc('\n', stdout); }
  if(opt.print_seq) { fputs(rb->seq, stdout); putc('\n', stdout); }
  putc('\n', stdout);
}

/* sw_cmdline.c:219-305 (one hit) */
static void print_sw_hit(size_t index, size_t hit_index, const rec_t *ra, const rec_t *rb,
                         const seqalign_sw_hit_t *h, const char *res_a, const char *res_b)
{
  size_t ctx_l = 0, ctx_r = 0, ls_a = 0, ls_b = 0, rs_a = 0, rs_b = 0, k;
  printf("hit %zu.%zu score: %i\n", index, hit_index, h->score);
  if(opt.context) {
    size_t rem_a = ra->len - (h->pos_a + h->len_a), rem_b = rb->len - (h->pos_b + h->len_b);
    ctx_l = MIN2(MAX2((size_t)h->pos_a, (size_t)h->pos_b), (size_t)opt.context);
    ctx_r = MIN2(MAX2(rem_a, rem_b), (size_t)opt.context);
    ls_a = ctx_l > h->pos_a ? ctx_l - h->pos_a : 0;
    ls_b = ctx_l > h->pos_b ? ctx_l - h->pos_b : 0;
    rs_a = ctx_r > rem_a ? ctx_r - rem_a : 0;
    rs_b = ctx_r > rem_b ? ctx_r - rem_b : 0;
  }
  print_sw_part(res_a, res_b, h->pos_a, h->len_a, ra->seq, ls_a, rs_a, ctx_l - ls_a, ctx_r - rs_a);
  if(opt.print_pretty) {
    size_t max_l = MAX2(ls_a, ls_b), max_r = MAX2(rs_a, rs_b);
    fputs("  ", stdout);
    for(k = 0; k < max_l; k++) putc(' ', stdout);
    for(k = 0; k < ctx_l - max_l; k++) putc('.', stdout);
    alignment_print_spacer(res_a, res_b, &scoring);
    for(k = 0; k < ctx_r - max_r; k++) putc('.', stdout);
    for(k = 0; k < max_r; k++) putc(' ', stdout);
    putc('\n', stdout);
  }
  print_sw_part(res_b, res_a, h->pos_b, h->len_b, rb->seq, ls_b, rs_b, ctx_l - ls_b, ctx_r - rs_b);
  printf("\n");
}

/* ---------------------------------------------------------------- batches */

static void check(int rc, const char *what)
{
  if(rc != SEQALIGN_OK) {
    fprintf(stderr, "Error: %s: %s %s\n", what, seqalign_strerror(rc), seqalign_last_error());
    exit(EXIT_FAILURE);
  }
}

static size_t g_alignment_index = 0;
/* SEQALIGN_GPUS=N: one context per device 0..N-1; every batch is split over them by pair index */
static seqalign_ctx_t *g_ctxs[64];
static int g_nctx = 1;

static void as_batch(const batch_t *bt, seqalign_batch_t *b)
{
  b->n_pairs = bt->n; b->arena = bt->text; b->arena_bytes = bt->text_len;
  b->off_a = bt->off_a; b->len_a = bt->len_a; b->off_b = bt->off_b; b->len_b = bt->len_b;
}

static void reserve_out(batch_t *bt, size_t bytes)
{
  if(bytes > bt->out_cap) {
    bt->out_cap = bytes + bytes / 4;
    free(bt->out_a); free(bt->out_b);
    bt->out_a = malloc(bt->out_cap); bt->out_b = malloc(bt->out_cap);
    if(!bt->out_a || !bt->out_b) oom();
  }
}

/* ---- stage 2: the GPU */
static void align_nw(batch_t *bt)
{
  seqalign_batch_t b;
  size_t n = bt->n, i, total = 0;
  if(!n) return;
  if(n > bt->res_cap) {
    bt->res_cap = n + n / 4;
    bt->str_off = grow(bt->str_off, (bt->res_cap + 1) * sizeof(uint64_t));
    bt->out_len = grow(bt->out_len, bt->res_cap * sizeof(uint32_t));
    bt->score = grow(bt->score, bt->res_cap * sizeof(int32_t));
  }
  as_batch(bt, &b);
  if(opt.cigar) {
    /* CIGAR straight from the walks' bit planes (seqalign_nw_batch_cigar): 64-byte slots -- a read's "150M" is 5 bytes -- and once
     * more with worst-case slots (2 (len_a + len_b) + 2) for the rare batch in which some pair's does not fit */
    int pass, rc = SEQALIGN_OK;
    for(pass = 0; pass < 2; pass++) {
      for(i = 0, total = 0; i < n; i++) { bt->str_off[i] = total; total += pass ? 2 * ((size_t)bt->len_a[i] + bt->len_b[i]) + 2 : 64; }
      bt->str_off[n] = total;
      reserve_out(bt, total + 1);
      rc = g_nctx > 1 ? seqalign_nw_batch_cigar_multi(g_ctxs, g_nctx, &b, &scoring, opt.cigar, bt->str_off, bt->out_a, bt->out_len, bt->score)
                      : seqalign_nw_batch_cigar(g_ctxs[0], &b, &scoring, opt.cigar, bt->str_off, bt->out_a, bt->out_len, bt->score);
      if(rc != SEQALIGN_E_NOMEM) break;
    }
    check(rc, "seqalign_nw_batch_cigar");
    return;
  }
  for(i = 0; i < n; i++) { bt->str_off[i] = total; total += (size_t)bt->len_a[i] + bt->len_b[i] + 1; }
  reserve_out(bt, total + 1);
  if(g_nctx > 1) check(seqalign_nw_batch_multi(g_ctxs, g_nctx, &b, &scoring, bt->str_off, bt->out_a, bt->out_b, bt->out_len, bt->score), "seqalign_nw_batch_multi");
  else check(seqalign_nw_batch(g_ctxs[0], &b, &scoring, bt->str_off, bt->out_a, bt->out_b, bt->out_len, bt->score), "seqalign_nw_batch");
}

static unsigned sw_cap(void) { return opt.max_hits_set ? opt.max_hits : 16; }   /* device path; see the re-run in print_sw */

static void align_sw(batch_t *bt)
{
  seqalign_batch_t b;
  size_t n = bt->n, i;
  const unsigned cap = sw_cap();
  uint64_t str_cap = 0, hit_cap;
  bt->n_hits = 0;
  if(!n) return;
  if(n > bt->res_cap) {
    bt->res_cap = n + n / 4;
    bt->min_score = grow(bt->min_score, (bt->res_cap + 1) * sizeof(int32_t));
  }
  as_batch(bt, &b);
  for(i = 0; i < n; i++) {
    /* sw_cmdline.c:192-197 */
    bt->min_score[i] = opt.min_score_set ? opt.min_score
                                         : (int)(scoring.match * MAX2(0.2 * MIN2(bt->len_a[i], bt->len_b[i]), 2));
    str_cap += (uint64_t)(cap ? cap : 1) * ((uint64_t)bt->len_a[i] + bt->len_b[i] + 1);
  }
  hit_cap = (uint64_t)n * (cap ? cap : 1) + 16;
  if(hit_cap > bt->hit_cap) { bt->hit_cap = hit_cap + hit_cap / 4; free(bt->hits); bt->hits = malloc(bt->hit_cap * sizeof(*bt->hits)); if(!bt->hits) oom(); }
  reserve_out(bt, str_cap + 16);
  if(opt.cigar) {   /* (no CIGAR is longer than 2 x its columns: the strings' room holds it twice over, out_a and out_b are one allocation each) */
    if(cap && g_nctx > 1) check(seqalign_sw_batch_cigar_multi(g_ctxs, g_nctx, &b, &scoring, bt->min_score, cap, opt.cigar, bt->hits, hit_cap, &bt->n_hits, bt->out_a, str_cap + 16), "seqalign_sw_batch_cigar_multi");
    else if(cap) check(seqalign_sw_batch_cigar(g_ctxs[0], &b, &scoring, bt->min_score, cap, opt.cigar, bt->hits, hit_cap, &bt->n_hits, bt->out_a, str_cap + 16), "seqalign_sw_batch_cigar");
    return;
  }
  if(cap && g_nctx > 1) check(seqalign_sw_batch_multi(g_ctxs, g_nctx, &b, &scoring, bt->min_score, cap, bt->hits, hit_cap, &bt->n_hits, bt->out_a, bt->out_b, str_cap + 16), "seqalign_sw_batch_multi");
  else if(cap) check(seqalign_sw_batch(g_ctxs[0], &b, &scoring, bt->min_score, cap, bt->hits, hit_cap, &bt->n_hits, bt->out_a, bt->out_b, str_cap + 16), "seqalign_sw_batch");
}

/* ---- stage 3: the reference's text */

/* The plain form (no --pretty / --colour / --printfasta / --zam / --printmatrices): res_a \n res_b \n [score: N \n] \n per
 * pair (nw_cmdline.c:78-149 with those options off), composed in 1 MiB pieces and written with one fwrite each -- the batch
 * calls know every string's length, so nothing is scanned (a million pairs: 0.27 s of fputs / putc / printf before). */
static void print_nw_batch_plain(const batch_t *bt)
{
  enum { PIECE = 1 << 20 };
  static char *buf;
  size_t i, at = 0;
  if(!buf && !(buf = malloc(PIECE + 64))) oom();
  for(i = 0; i < bt->n; i++) {
    const size_t len = bt->out_len[i], need = 2 * len + 32;
    const char *ra = bt->out_a + bt->str_off[i], *rb = bt->out_b + bt->str_off[i];
    if(need > PIECE) {                       /* a pair longer than a piece: straight through stdio */
      fwrite(buf, 1, at, stdout); at = 0;
      fwrite(ra, 1, len, stdout); putc('\n', stdout); fwrite(rb, 1, len, stdout); putc('\n', stdout);
      if(opt.print_scores) printf("score: %i\n", bt->score[i]);
      putc('\n', stdout);
      continue;
    }
    if(at + need > PIECE) { fwrite(buf, 1, at, stdout); at = 0; }
    memcpy(buf + at, ra, len); at += len; buf[at++] = '\n';
    memcpy(buf + at, rb, len); at += len; buf[at++] = '\n';
    if(opt.print_scores) {
      char digits[12];
      long long v = bt->score[i];
      unsigned long long u = v < 0 ? (unsigned long long)-v : (unsigned long long)v;
      int nd = 0;
      memcpy(buf + at, "score: ", 7); at += 7;
      if(v < 0) buf[at++] = '-';
      do { digits[nd++] = (char)('0' + u % 10); u /= 10; } while(u);
      while(nd) buf[at++] = digits[--nd];
      buf[at++] = '\n';
    }
    buf[at++] = '\n';
  }
  fwrite(buf, 1, at, stdout);
  fflush(stdout);
}

/* --cigar / --cigarx (this tool's own; the reference prints the gapped strings only): one line per pair,
 *     <pair index>\t<name of seq 1 | *>\t<name of seq 2 | *>\t<score>\t<CIGAR>
 * names without their '>' / '@'; seq 1 is the CIGAR's query, seq 2 its reference (include/seqalign_hip.h). */
static const char *bare_name(const char *name) { return !name || !name[0] ? "*" : (name[0] == '>' || name[0] == '@') ? name + 1 : name; }
static size_t g_pair_index;

static void print_nw_batch_cigar(const batch_t *bt)
{
  size_t i;
  for(i = 0; i < bt->n; i++, g_pair_index++) {
    const rec_t ra = rec_a(bt, i), rb = rec_b(bt, i);
    printf("%zu\t%s\t%s\t%i\t", g_pair_index, bare_name(ra.name), bare_name(rb.name), bt->score[i]);
    fwrite(bt->out_a + bt->str_off[i], 1, bt->out_len[i], stdout);
    putc('\n', stdout);
  }
  fflush(stdout);
}

/* SW: one line per hit,  <pair index>\t<hit index>\t<score>\t<pos in seq 1>\t<len>\t<pos in seq 2>\t<len>\t<CIGAR>  (0-based positions, as the
 * reference's "[pos: P; len: L]", sw_cmdline.c:82-124) */
static void print_sw_batch_cigar(const batch_t *bt)
{
  const unsigned cap = sw_cap();
  size_t k = 0;
  while(k < bt->n_hits) {
    const size_t pair = (size_t)bt->hits[k].pair;
    size_t k1 = k, hit = 0;
    while(k1 < bt->n_hits && bt->hits[k1].pair == pair) k1++;
    if(!opt.max_hits_set && k1 - k == cap) {
      /* the device path stopped at its cap but the user asked for "no limit": this one pair through the per-pair API (as print_sw_batch) */
      const rec_t ra = rec_a(bt, pair), rb = rec_b(bt, pair);
      sw_aligner_t *sw = smith_waterman_new();
      alignment_t *r = alignment_create(256);
      char *text = malloc(2 * (ra.len + rb.len) + 8);
      if(!text) oom();
      smith_waterman_align2(ra.seq, rb.seq, ra.len, rb.len, &scoring, sw);
      while(smith_waterman_fetch(sw, r) && r->score >= bt->min_score[pair]) {
        if(seqalign_cigar(r->result_a, r->result_b, r->length, opt.cigar == SEQALIGN_CIGAR_EQX, !scoring.case_sensitive, text, 2 * (ra.len + rb.len) + 8) == (size_t)-1)
          die("internal error: CIGAR of a hit%s", "");
        printf("%zu\t%zu\t%i\t%zu\t%zu\t%zu\t%zu\t%s\n", g_pair_index + pair, hit++, r->score, r->pos_a, r->len_a, r->pos_b, r->len_b, text);
      }
      free(text); alignment_free(r); smith_waterman_free(sw);
    } else {
      for(; k < k1; k++) {
        const seqalign_sw_hit_t *h = &bt->hits[k];
        printf("%zu\t%zu\t%i\t%u\t%u\t%u\t%u\t%s\n", g_pair_index + pair, hit++, h->score, h->pos_a, h->len_a, h->pos_b, h->len_b, bt->out_a + h->str_off);
      }
    }
    k = k1;
  }
  g_pair_index += bt->n;
  fflush(stdout);
}

static void print_nw_batch(const batch_t *bt)
{
  size_t i;
  if(opt.cigar) { print_nw_batch_cigar(bt); return; }
  if(!opt.print_matrices && !opt.zam && !opt.print_fasta && !opt.print_pretty && !opt.print_colour) { print_nw_batch_plain(bt); return; }
  for(i = 0; i < bt->n; i++) {
    const rec_t ra = rec_a(bt, i), rb = rec_b(bt, i);
    if(opt.print_matrices) {   /* needs the matrices on the host: per-pair API */
      nw_aligner_t *nw = needleman_wunsch_new();
      alignment_t *r = alignment_create(256);
      needleman_wunsch_align2(ra.seq, rb.seq, ra.len, rb.len, &scoring, nw, r);
      alignment_print_matrices(nw);
      alignment_free(r); needleman_wunsch_free(nw);
    }
    print_nw(&ra, &rb, bt->out_a + bt->str_off[i], bt->out_b + bt->str_off[i], bt->score[i]);
  }
  fflush(stdout);
}

static void print_sw_batch(const batch_t *bt)
{
  /* pairs with an empty sequence are reported and skipped upstream (sw_cmdline.c:137-151) */
  const unsigned cap = sw_cap();
  size_t i, h0;
  if(opt.cigar) { print_sw_batch_cigar(bt); return; }
  for(i = 0, h0 = 0; i < bt->n; i++) {
    const rec_t ra = rec_a(bt, i), rb = rec_b(bt, i);
    size_t h1 = h0, k;
    while(h1 < bt->n_hits && bt->hits[h1].pair == i) h1++;
    if(ra.len == 0 || rb.len == 0) {
      fprintf(stderr, "Error: Sequences must have length > 0\n");
      if(opt.print_fasta && ra.name[0] && rb.name[0]) fprintf(stderr, "%s\n%s\n", ra.name, rb.name);
      h0 = h1;
      continue;
    }
    print_sw_heading(g_alignment_index, &ra, &rb);
    if(opt.print_matrices) {
      sw_aligner_t *sw = smith_waterman_new();
      smith_waterman_align2(ra.seq, rb.seq, ra.len, rb.len, &scoring, sw);
      alignment_print_matrices(smith_waterman_get_aligner(sw));
      smith_waterman_free(sw);
    }
    if(!opt.max_hits_set && h1 - h0 == cap) {
      /* the device path stopped at its cap but the user asked for "no limit":
       * re-run this one pair through the per-pair API, which has none */
      sw_aligner_t *sw = smith_waterman_new();
      alignment_t *r = alignment_create(256);
      size_t hit_index = 0;
      smith_waterman_align2(ra.seq, rb.seq, ra.len, rb.len, &scoring, sw);
      while(smith_waterman_fetch(sw, r) && r->score >= bt->min_score[i]) {
        seqalign_sw_hit_t h;
        h.pair = i; h.score = r->score; h.pos_a = (uint32_t)r->pos_a; h.pos_b = (uint32_t)r->pos_b;
        h.len_a = (uint32_t)r->len_a; h.len_b = (uint32_t)r->len_b; h.length = (uint32_t)r->length; h.str_off = 0;
        print_sw_hit(g_alignment_index, hit_index++, &ra, &rb, &h, r->result_a, r->result_b);
      }
      alignment_free(r); smith_waterman_free(sw);
    } else {
      for(k = h0; k < h1; k++)
        print_sw_hit(g_alignment_index, k - h0, &ra, &rb, &bt->hits[k], bt->out_a + bt->hits[k].str_off, bt->out_b + bt->hits[k].str_off);
    }
    fputs("==\n", stdout);
    g_alignment_index++;
    h0 = h1;
  }
  fflush(stdout);
}

/* ------------------------------------------------------------- the pipeline */

#define BATCH_PAIRS 65536
/* ... or this many DP cells, whichever comes first: the multi-hit SW path keeps 5 bytes per cell on the GPU (NW and --maxhits 1:
 * one), and the first batch pays for allocating that -- 65 536 reads of 150 bp against 1 000 bp windows would ask for 50 GB at once
 * (0.5-2 s in the driver); 2 G cells = 10 GB, taken once and reused by every later batch */
#define BATCH_CELLS ((uint64_t)2 << 30)
#define N_BATCHES 8          /* one on the GPU, one being printed, the rest read ahead: opening the GPU takes as long as reading
                                ~0.5 M pairs, and a reader that may only run two batches ahead sits out most of that (a batch's
                                buffers are allocated when it is first used: short inputs touch one) */

typedef struct {             /* a FIFO of batches between two stages */
  pthread_mutex_t mu; pthread_cond_t cv;
  batch_t *q[N_BATCHES + 1]; int head, count;
} fifo_t;

static void fifo_init(fifo_t *f) { pthread_mutex_init(&f->mu, NULL); pthread_cond_init(&f->cv, NULL); f->head = f->count = 0; }
static void fifo_put(fifo_t *f, batch_t *b)
{
  pthread_mutex_lock(&f->mu);
  f->q[(f->head + f->count++) % (N_BATCHES + 1)] = b;    /* (never more than N_BATCHES in circulation) */
  pthread_cond_signal(&f->cv);
  pthread_mutex_unlock(&f->mu);
}
static batch_t *fifo_get(fifo_t *f)
{
  batch_t *b;
  pthread_mutex_lock(&f->mu);
  while(!f->count) pthread_cond_wait(&f->cv, &f->mu);
  b = f->q[f->head]; f->head = (f->head + 1) % (N_BATCHES + 1); f->count--;
  pthread_mutex_unlock(&f->mu);
  return b;
}

static fifo_t g_free, g_read, g_aligned;
/* SEQALIGN_CLI_TIMING=1: how long each stage WORKED (not waited), on stderr at the end: which one bounds the tool */
static double g_busy[3], g_open_s;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* stage 1: records into batches (two records at a time, alignment_cmdline.c:611-622) */
static void *reader_main(void *arg)
{
  batch_t *bt = fifo_get(&g_free);
  int f;
  double t0 = now_s();
  (void)arg;
  bt->n = 0; bt->text_len = 0; bt->last = 0; bt->cells = 0;
  if(opt.seq1) { batch_add_a(bt, "", opt.seq1, strlen(opt.seq1)); batch_add_b(bt, "", opt.seq2, strlen(opt.seq2)); }
  for(f = 0; f < opt.n_files; f++) {
    seqalign_reader_t *r1 = seqalign_reader_open(opt.files1[f]), *r2 = NULL;
    const char *n1, *s1, *n2, *s2;
    size_t l1, l2;
    if(!r1) { fprintf(stderr, "Error: Couldn't read file: %s\n", opt.files1[f]); exit(EXIT_FAILURE); }
    if(opt.files2[f] && !(r2 = seqalign_reader_open(opt.files2[f]))) { fprintf(stderr, "Error: Couldn't read file: %s\n", opt.files2[f]); exit(EXIT_FAILURE); }
    for(;;) {
      if(!seqalign_reader_next(r1, &n1, &s1, &l1)) break;
      batch_add_a(bt, n1, s1, l1);      /* (the reader's buffers are valid until its next call) */
      if(!seqalign_reader_next(r2 ? r2 : r1, &n2, &s2, &l2)) {
        fprintf(stderr, "Odd number of sequences - I read in pairs!\n");
        batch_drop_a(bt);
        break;
      }
      batch_add_b(bt, n2, s2, l2);
      if(bt->n >= BATCH_PAIRS || bt->cells >= BATCH_CELLS) {
        g_busy[0] += now_s() - t0;
        fifo_put(&g_read, bt);
        bt = fifo_get(&g_free);         /* blocks while all batches are downstream: bounded memory */
        t0 = now_s();
        bt->n = 0; bt->text_len = 0; bt->last = 0; bt->cells = 0;
      }
    }
    seqalign_reader_close(r1);
    if(r2) seqalign_reader_close(r2);
  }
  bt->last = 1;
  g_busy[0] += now_s() - t0;
  fifo_put(&g_read, bt);
  return NULL;
}

/* stage 2.  Opening the GPU (runtime initialisation + context: 0.2-0.3 s) is this thread's first job, so that it runs beside
 * the reader's first batch instead of in front of it. */
static void open_gpus(void)
{
  int rc = seqalign_ctx_create(getenv("SEQALIGN_DEVICE") ? atoi(getenv("SEQALIGN_DEVICE")) : 0, &g_ctxs[0]);
  if(rc != SEQALIGN_OK) {
    fprintf(stderr, "seqalign: cannot open the GPU: %s (%s)\nseqalign: there is no CPU path; an MI355X (gfx950) is required\n",
            seqalign_strerror(rc), seqalign_last_error());
    exit(EXIT_FAILURE);
  }
  if(getenv("SEQALIGN_GPUS")) {
    int want = atoi(getenv("SEQALIGN_GPUS")), have = seqalign_device_count(), g;
    if(want > have) want = have;
    if(want > 64) want = 64;
    for(g = 1; g < want; g++) {
      if(seqalign_ctx_create(g, &g_ctxs[g]) != SEQALIGN_OK) {
        fprintf(stderr, "seqalign: cannot open GPU %i: %s\n", g, seqalign_last_error());
        exit(EXIT_FAILURE);
      }
      g_nctx = g + 1;
    }
  }
}

static void *aligner_main(void *arg)
{
  (void)arg;
  { const double t0 = now_s(); open_gpus(); g_open_s = now_s() - t0; }
  for(;;) {
    batch_t *bt = fifo_get(&g_read);
    const double t0 = now_s();
    if(opt.tool == TOOL_NW) align_nw(bt); else align_sw(bt);
    g_busy[1] += now_s() - t0;
    fifo_put(&g_aligned, bt);
    if(bt->last) return NULL;
  }
}

static void usage(void)
{
  const int nw = opt.tool == TOOL_NW;
  printf("usage: %s [OPTIONS] [seq1 seq2]\n"
         "  %s alignment of pairs of sequences on an MI355X -- the options and the output of seq-align's %s.\n"
         "  Pairs are read two records at a time (FASTA, FASTQ or one sequence per line; gzip accepted) and aligned in batches.\n\n"
         "  input:    --file <file> | --files <f1> <f2> | --stdin | seq1 seq2\n"
         "  scoring:  --case_sensitive  --scoring PAM30|PAM70|BLOSUM80|BLOSUM62|DNA_HYBRIDIZATION\n"
         "            --match <n> --mismatch <n> --gapopen <n> --gapextend <n>   (gap of length N: gapopen + N * gapextend)\n"
         "            --substitution_matrix <file>  --substitution_pairs <file>  --wildcard <char> <score>\n"
         "            --nogaps --nogapsin1 --nogapsin2 --nomismatches%s\n"
         "  output:   --printfasta --pretty --colour --printmatrices%s\n"
         "            --cigar | --cigarx   one tab-separated line per alignment with its CIGAR (M / I / D, or = / X / I / D; seq 1 is the\n"
         "                                 query) instead of the gapped strings -- made from the GPU walks' bit planes, no strings expanded\n"
         "  environment: SEQALIGN_DEVICE=<gpu>  SEQALIGN_GPUS=<n> (split every batch over n GPUs)\n"
         "               SEQALIGN_CLI_TIMING=1 (the stages' busy times on stderr)  SEQALIGN_CLI_EXIT=full (run the runtime's exit handlers)\n",
         nw ? "seqalign_nw" : "seqalign_sw", nw ? "Global (Needleman-Wunsch)" : "Local (Smith-Waterman)",
         nw ? "needleman_wunsch" : "smith_waterman",
         nw ? " --freestartgap --freeendgap" : "",
         nw ? " --printscores --zam" : " --printseq --minscore <n> --maxhits <n> --context <n>");
}

int main(int argc, char **argv)
{
  const char *base = strrchr(argv[0], '/');
  static batch_t batches[N_BATCHES];
  pthread_t reader, aligner;
  int k;
  const double t_main = now_s();
  base = base ? base + 1 : argv[0];
  /* a command-line run is short: skip the ~0.1-0.2 s the library would spend looking for a good
     placement of its matrix arenas (seqalign_hip.h, seqalign_arenas_alloc) unless the user asks for it */
  setenv("SEQALIGN_ARENA_SCAN_GIB", "0", 0);
  memset(&opt, 0, sizeof opt);
  opt.tool = strstr(base, "sw") ? TOOL_SW : TOOL_NW;
  for(k = 1; k < argc; k++) if(!strcasecmp(argv[k], "--help") || !strcmp(argv[k], "-h")) { usage(); return EXIT_SUCCESS; }

  scoring_system_default(&scoring);
  if(opt.tool == TOOL_SW) {   /* sw_cmdline.c:37-46 */
    scoring.match = 2; scoring.mismatch = -2; scoring.gap_open = -2; scoring.gap_extend = -1;
  }
  parse_args(argc, argv);

  setvbuf(stdout, NULL, _IOFBF, (size_t)4 << 20);   /* hundreds of MB of text: fewer, larger writes */
  fifo_init(&g_free); fifo_init(&g_read); fifo_init(&g_aligned);
  for(k = 0; k < N_BATCHES; k++) fifo_put(&g_free, &batches[k]);
  if(pthread_create(&reader, NULL, reader_main, NULL) || pthread_create(&aligner, NULL, aligner_main, NULL)) {
    fprintf(stderr, "seqalign: cannot start the pipeline's threads\n");
    return EXIT_FAILURE;
  }
  for(;;) {   /* stage 3, here: in input order */
    batch_t *bt = fifo_get(&g_aligned);
    const int last = bt->last;
    const double t0 = now_s();
    if(opt.tool == TOOL_NW) print_nw_batch(bt); else print_sw_batch(bt);
    g_busy[2] += now_s() - t0;
    if(last) break;
    fifo_put(&g_free, bt);
  }
  pthread_join(reader, NULL); pthread_join(aligner, NULL);
  {
    const double t1 = now_s();
    for(k = 0; k < g_nctx; k++) seqalign_ctx_destroy(g_ctxs[k]);
    if(getenv("SEQALIGN_CLI_TIMING"))
      fprintf(stderr, "seqalign: stages busy: read %.3f s, align %.3f s (GPU opened in %.3f s), print %.3f s; main() %.3f s to the last byte, "
              "%.3f s closing the contexts\n", g_busy[0], g_busy[1], g_open_s, g_busy[2], t1 - t_main, now_s() - t1);
  }
  /* Everything is written and the contexts are closed: leave without the HIP runtime's exit handlers (0.15 s of a 0.3-0.5 s
     run on a million pairs; SEQALIGN_CLI_EXIT=full runs them, for leak checkers) */
  if(!getenv("SEQALIGN_CLI_EXIT") || strcmp(getenv("SEQALIGN_CLI_EXIT"), "full")) {
    const int bad = fflush(NULL) != 0 || ferror(stdout);
    _exit(bad ? EXIT_FAILURE : EXIT_SUCCESS);
  }
  return EXIT_SUCCESS;
}
