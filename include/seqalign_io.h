/*
 * seqalign_io.h -- the data formats either side of the hot path:
 * substitution-matrix / pair-list scoring files and sequence files.
 *
 * Replaces, with fresh code and error CODES instead of exit():
 *   align_scoring_load_matrix   reference src/alignment_scoring_load.c:39-220
 *   align_scoring_load_pairwise reference src/alignment_scoring_load.c:223-306
 *   the seq_file based pair iteration of align_from_file
 *                               reference src/alignment_cmdline.c:578-640
 * (seq_file / string_buffer are not vendored upstream; the sequence reader goes
 * through zlib like upstream's, so files may be gzip-compressed; the scoring
 * loaders take a FILE*.)  Implementation: seq-align_amd/host/sa_io.c.
 */
#ifndef SEQALIGN_IO_H
#define SEQALIGN_IO_H

#include <stdio.h>

#include "seqalign_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Substitution matrix: first non-comment line = column characters, every later
 * line = row character + one score per column; entries separated by whitespace,
 * or by the single character that starts the header line.  '#' lines are
 * comments.  Every (row, column) score is added with scoring_add_mutation
 * (lower-cased unless case_sensitive).  Returns 0, or -1 with a message in err. */
int seqalign_scoring_load_matrix(FILE *f, scoring_t *scoring, int case_sensitive,
                                 char *err, size_t err_cap);

/* Pair list: one "a b score" (whitespace) or "a<sep>b<sep>score" per line. */
int seqalign_scoring_load_pairs(FILE *f, scoring_t *scoring, int case_sensitive,
                                char *err, size_t err_cap);

/* Sequence reader: FASTA ('>'), FASTQ ('@') or plain (one sequence per line),
 * decided per record from its first character.  Blank lines are skipped. */
typedef struct seqalign_reader seqalign_reader_t;

seqalign_reader_t *seqalign_reader_open(const char *path);   /* "-" = stdin; NULL on failure */
void seqalign_reader_close(seqalign_reader_t *r);
/* 1 = a record was read (*name: header line incl. its '>' / '@', "" for plain
 * input; *seq: the sequence; both valid until the next call), 0 = end of file. */
int seqalign_reader_next(seqalign_reader_t *r, const char **name, const char **seq, size_t *seq_len);

#ifdef __cplusplus
}
#endif
#endif
