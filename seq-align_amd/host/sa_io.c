/*
 * sa_io.c -- scoring-file loaders and the sequence reader (host, C99).
 * Interface and the reference functions they replace: include/seqalign_io.h.
 */
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include "seqalign_io.h"

/* ------------------------------------------------------------- line input */

typedef struct { char *b; size_t len, cap; } line_t;

/* next line without its end-of-line characters; 0 at end of file */
static int read_line(FILE *f, line_t *ln)
{
  int c;
  ln->len = 0;
  while((c = getc(f)) != EOF) {
    if(ln->len + 2 > ln->cap) {
      ln->cap = ln->cap ? 2 * ln->cap : 256;
      ln->b = realloc(ln->b, ln->cap);
      if(!ln->b) { fprintf(stderr, "seqalign: out of memory\n"); exit(EXIT_FAILURE); }
    }
    if(c == '\n') break;
    ln->b[ln->len++] = (char)c;
  }
  if(c == EOF && ln->len == 0) { if(ln->b) ln->b[0] = '\0'; return 0; }
  while(ln->len && (ln->b[ln->len-1] == '\r' || ln->b[ln->len-1] == '\n')) ln->len--;
  if(!ln->b) { ln->cap = 16; ln->b = malloc(ln->cap); }
  ln->b[ln->len] = '\0';
  return 1;
}

static int blank(const char *s)
{
  for(; *s; s++) if(!isspace((unsigned char)*s)) return 0;
  return 1;
}

static int fail(char *err, size_t cap, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  if(err && cap) vsnprintf(err, cap, fmt, ap);
  va_end(ap);
  return -1;
}

static char fold(char c, int case_sensitive)
{
  return case_sensitive ? c : (char)tolower((unsigned char)c);
}

/* ------------------------------------------------------- scoring matrices */

int seqalign_scoring_load_matrix(FILE *f, scoring_t *sc, int case_sensitive, char *err, size_t err_cap)
{
  line_t ln = {0};
  char cols[256];
  int n_cols = 0, line_no = 0, rc = 0, rows = 0;
  char sep;

  /* header: first line that is neither a comment nor blank */
  for(;;) {
    if(!read_line(f, &ln)) { rc = fail(err, err_cap, "substitution matrix: empty file"); goto out; }
    line_no++;
    if(ln.len && ln.b[0] != '#' && !blank(ln.b)) break;
  }
  if(ln.len < 2) { rc = fail(err, err_cap, "substitution matrix: too few column headings (line %d)", line_no); goto out; }
  sep = ln.b[0];
  if(isdigit((unsigned char)sep) || sep == '-') {
    rc = fail(err, err_cap, "substitution matrix: digits and '-' cannot be separators (line %d)", line_no);
    goto out;
  }

  if(isspace((unsigned char)sep)) {
    /* whitespace separated: column characters are the header's tokens */
    const char *p;
    for(p = ln.b; *p; p++)
      if(!isspace((unsigned char)*p) && n_cols < 256) cols[n_cols++] = fold(*p, case_sensitive);
    while(read_line(f, &ln)) {
      const char *q = ln.b;
      char from;
      int k;
      line_no++;
      while(*q && isspace((unsigned char)*q)) q++;
      if(!*q || ln.b[0] == '#') continue;
      from = fold(*q, case_sensitive);
      q++;
      for(k = 0; k < n_cols; k++) {
        char *end;
        long v;
        if(!isspace((unsigned char)*q)) {
          rc = fail(err, err_cap, "substitution matrix: expected whitespace between elements (line %d)", line_no);
          goto out;
        }
        v = strtol(q, &end, 10);
        if(end == q) { rc = fail(err, err_cap, "substitution matrix: missing number (line %d)", line_no); goto out; }
        scoring_add_mutation(sc, from, cols[k], (int)v);
        q = end;
      }
      if(!blank(q)) { rc = fail(err, err_cap, "substitution matrix: too many columns (line %d)", line_no); goto out; }
      rows++;
    }
  } else {
    /* single-character separator: header is <sep>c<sep>c..., rows are c<sep>n<sep>n... */
    size_t i;
    for(i = 0; i < ln.len; i += 2) {
      if(ln.b[i] != sep || i + 1 >= ln.len) {
        rc = fail(err, err_cap, "substitution matrix: separator missing from header (line %d)", line_no);
        goto out;
      }
      if(n_cols < 256) cols[n_cols++] = fold(ln.b[i+1], case_sensitive);
    }
    while(read_line(f, &ln)) {
      const char *q;
      char from;
      int k = 0;
      line_no++;
      if(!ln.len || ln.b[0] == '#' || blank(ln.b)) continue;
      from = fold(ln.b[0], case_sensitive);
      q = ln.b + 1;
      while(*q) {
        char *end;
        long v;
        if(*q != sep) { rc = fail(err, err_cap, "substitution matrix: separator missing (line %d)", line_no); goto out; }
        q++;
        v = strtol(q, &end, 10);
        if(end == q) { rc = fail(err, err_cap, "substitution matrix: missing number (line %d)", line_no); goto out; }
        if(k >= n_cols) { rc = fail(err, err_cap, "substitution matrix: too many columns (line %d)", line_no); goto out; }
        scoring_add_mutation(sc, from, cols[k++], (int)v);
        q = end;
      }
      rows++;
    }
  }
  if(rows == 0) rc = fail(err, err_cap, "substitution matrix: no rows");
out:
  free(ln.b);
  return rc;
}

int seqalign_scoring_load_pairs(FILE *f, scoring_t *sc, int case_sensitive, char *err, size_t err_cap)
{
  line_t ln = {0};
  int line_no = 0, added = 0, rc = 0;

  while(read_line(f, &ln)) {
    char a, b, *end;
    const char *num;
    long v;
    line_no++;
    if(!ln.len || ln.b[0] == '#' || blank(ln.b)) continue;
    if(ln.len < 5) { rc = fail(err, err_cap, "substitution pairs: line too short (line %d)", line_no); goto out; }
    a = ln.b[0];
    if(isspace((unsigned char)ln.b[1])) {          /* a <ws> b <ws> score */
      size_t p = 1;
      while(ln.b[p] && isspace((unsigned char)ln.b[p])) p++;
      if(p + 2 >= ln.len || !isspace((unsigned char)ln.b[p+1])) {
        rc = fail(err, err_cap, "substitution pairs: line too short (line %d)", line_no);
        goto out;
      }
      b = ln.b[p];
      num = ln.b + p + 2;
    } else {                                        /* a<sep>b<sep>score */
      if(ln.b[1] != ln.b[3]) { rc = fail(err, err_cap, "substitution pairs: inconsistent separators (line %d)", line_no); goto out; }
      b = ln.b[2];
      num = ln.b + 4;
    }
    v = strtol(num, &end, 10);
    if(end == num || !blank(end)) { rc = fail(err, err_cap, "substitution pairs: invalid number (line %d)", line_no); goto out; }
    scoring_add_mutation(sc, fold(a, case_sensitive), fold(b, case_sensitive), (int)v);
    added++;
  }
  if(!added) rc = fail(err, err_cap, "substitution pairs: no pairs in file");
out:
  free(ln.b);
  return rc;
}

/* --------------------------------------------------------- sequence files */

/* Lines come out of a block buffer filled by gzread (zlib reads plain files as they are, so one reader serves .fa and .fa.gz;
   the reference reads its sequence files through zlib too, alignment_cmdline.c via seq_file): one memchr + one memcpy per line.
   (Round 5: a gzgetc per character made this stage the slowest of the command-line tools' three, 0.6 s per 320 MB.) */
#define READER_BLOCK ((size_t)4 << 20)
struct seqalign_reader {
  gzFile f;            /* plain or gzip-compressed, zlib tells them apart */
  char *buf; size_t pos, end;
  line_t line, name, seq;
  int have_line;       /* line holds an unread line */
};

seqalign_reader_t *seqalign_reader_open(const char *path)
{
  seqalign_reader_t *r = calloc(1, sizeof(*r));
  if(!r) return NULL;
  r->buf = malloc(READER_BLOCK);
  r->f = !r->buf ? NULL : strcmp(path, "-") == 0 ? gzdopen(dup(STDIN_FILENO), "rb") : gzopen(path, "rb");
  if(!r->f) { free(r->buf); free(r); return NULL; }
  gzbuffer(r->f, 1 << 20);
  return r;
}

void seqalign_reader_close(seqalign_reader_t *r)
{
  if(!r) return;
  gzclose(r->f);
  free(r->buf); free(r->line.b); free(r->name.b); free(r->seq.b);
  free(r);
}

static void line_room(line_t *ln, size_t need)
{
  if(need > ln->cap) {
    ln->cap = 2 * need > 256 ? 2 * need : 256;
    ln->b = realloc(ln->b, ln->cap);
    if(!ln->b) { fprintf(stderr, "seqalign: out of memory\n"); exit(EXIT_FAILURE); }
  }
}

/* next line without its end-of-line characters; 0 at end of file */
static int read_line_block(seqalign_reader_t *r, line_t *ln)
{
  int got = 0;
  ln->len = 0;
  for(;;) {
    const char *s, *nl;
    size_t avail, take;
    if(r->pos == r->end) {
      const int n = gzread(r->f, r->buf, (unsigned)READER_BLOCK);
      if(n <= 0) break;
      r->pos = 0; r->end = (size_t)n;
    }
    s = r->buf + r->pos; avail = r->end - r->pos;
    nl = memchr(s, '\n', avail);
    take = nl ? (size_t)(nl - s) : avail;
    line_room(ln, ln->len + take + 2);
    memcpy(ln->b + ln->len, s, take);
    ln->len += take;
    r->pos += take + (nl ? 1 : 0);
    got = 1;
    if(nl) break;
  }
  line_room(ln, 2);
  if(!got) { ln->b[0] = '\0'; return 0; }
  while(ln->len && ln->b[ln->len-1] == '\r') ln->len--;
  ln->b[ln->len] = '\0';
  return 1;
}

static void set_line(line_t *dst, const char *s, size_t n)
{
  if(n + 1 > dst->cap) { dst->cap = 2 * (n + 1); dst->b = realloc(dst->b, dst->cap); }
  memcpy(dst->b, s, n);
  dst->b[n] = '\0';
  dst->len = n;
}

static void append_line(line_t *dst, const char *s, size_t n)
{
  if(dst->len + n + 1 > dst->cap) { dst->cap = 2 * (dst->len + n + 1); dst->b = realloc(dst->b, dst->cap); }
  memcpy(dst->b + dst->len, s, n);
  dst->len += n;
  dst->b[dst->len] = '\0';
}

/* the line just read becomes dst (the buffers change places: no copy) */
static void take_line(seqalign_reader_t *r, line_t *dst) { const line_t t = *dst; *dst = r->line; r->line = t; }

static int next_line(seqalign_reader_t *r)
{
  if(r->have_line) { r->have_line = 0; return 1; }
  return read_line_block(r, &r->line);
}

int seqalign_reader_next(seqalign_reader_t *r, const char **name, const char **seq, size_t *seq_len)
{
  do { if(!next_line(r)) return 0; } while(blank(r->line.b));

  set_line(&r->name, "", 0);
  set_line(&r->seq, "", 0);
  if(r->line.b[0] == '>') {                          /* FASTA: sequence may span lines */
    take_line(r, &r->name);
    while(next_line(r)) {
      if(r->line.b[0] == '>' || r->line.b[0] == '@') { r->have_line = 1; break; }
      if(blank(r->line.b)) continue;
      if(r->seq.len == 0) take_line(r, &r->seq);     /* (the usual case, one line: no copy) */
      else append_line(&r->seq, r->line.b, r->line.len);
    }
  } else if(r->line.b[0] == '@') {                   /* FASTQ: 4-line records */
    take_line(r, &r->name);
    if(next_line(r)) take_line(r, &r->seq);
    /* '+' line and quality line; a line that is not '+' (truncated record) starts the next record.
       At end of file there is no line to push back. */
    if(next_line(r)) {
      if(r->line.b[0] == '+') (void)next_line(r);
      else r->have_line = 1;
    }
  } else {                                           /* plain */
    take_line(r, &r->seq);
  }
  *name = r->name.b;
  *seq = r->seq.b;
  *seq_len = r->seq.len;
  return 1;
}
