/* sa_moves.c -- an alignment as two bit planes, and its expansion into the reference's pair of gapped strings.
 *
 * The device walkers on direction bytes (csrc/sa_traceback.hip) know an alignment as a sequence of three-way moves --
 * MATCH / GAP_A / GAP_B, the states of alignment_reverse_move (reference src/alignment.c:244-350).  Sending that home
 * instead of the two strings it stands for (reference src/needleman_wunsch.c:82-145 writes 2 x (len_a + len_b) chars
 * per pair) is 2 bits per column instead of 16: two planes of one bit per column,
 *     plane A: the column has a gap in seq_a  ('-' in result_a, the walk stood in GAP_A)
 *     plane B: the column has a gap in seq_b  ('-' in result_b, the walk stood in GAP_B)
 * in FORWARD column order, right-aligned in the pair's slot: a plane is n_words uint32, the alignment's last walked
 * column is bit 32 * n_words - 1 (bit 31 of the last word), the first is bit 32 * n_words - n_moves.  The host threads
 * that unpacked the strings anyway expand the planes against the sequences they still hold:
 *     result_x = the characters of seq_x in order, with '-' wherever plane X has a bit
 * -- which is exactly what AVX-512 VBMI2's byte expand does 64 columns at a time (vpexpandb); a scalar loop serves
 * every other CPU.  The columns a global walk does not visit (it stops at the first row or column; the reference then
 * pads, src/needleman_wunsch.c:117-132) are not in the planes: they follow from the lengths.
 */
#include <stdint.h>
#include <string.h>

#include "sa_internal.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

/* 64 plane bits starting at absolute bit `at` (words beyond n_words read as 0) */
static inline uint64_t plane_bits64(const uint32_t *plane, uint32_t n_words, uint64_t at) {
  const uint64_t w = at >> 5;
  const unsigned s = (unsigned)(at & 31);
  uint64_t lo = 0, hi = 0;
  if (w < n_words) lo = plane[w];
  if (w + 1 < n_words) lo |= (uint64_t)plane[w + 1] << 32;
  if (s == 0) return lo;
  if (w + 2 < n_words) hi = plane[w + 2];
  return (lo >> s) | (hi << (64 - s));
}

static inline uint64_t plane_popcount(const uint32_t *plane, uint32_t n_words, uint64_t first_bit) {
  /* bits [first_bit, 32 * n_words) */
  uint64_t n = 0, w = first_bit >> 5;
  if (w >= n_words) return 0;
  n += (uint64_t)__builtin_popcount(plane[w] >> (first_bit & 31));
  for (++w; w < n_words; ++w) n += (uint64_t)__builtin_popcount(plane[w]);
  return n;
}

/* does any column in [first_bit, 32 * n_words) have BOTH planes set?  No walk produces that (a column is a gap in at most
 * one of the strings); corrupt or stale plane words could, and the expansion would then write past len_a + len_b columns */
static inline int planes_overlap(const uint32_t *pa, const uint32_t *pb, uint32_t n_words, uint64_t first_bit) {
  uint64_t w = first_bit >> 5;
  if (w >= n_words) return 0;
  uint32_t any = (pa[w] & pb[w]) >> (first_bit & 31);
  for (++w; w < n_words; ++w) any |= pa[w] & pb[w];
  return any != 0;
}

typedef void (*expand_fn)(const char *src, const uint32_t *plane, uint32_t n_words, uint64_t first_bit, uint64_t n_cols,
                          char *dst);

/* dst[c] = plane bit (first_bit + c) ? '-' : next character of src, for c in [0, n_cols) */
static void expand_scalar(const char *src, const uint32_t *plane, uint32_t n_words, uint64_t first_bit, uint64_t n_cols,
                          char *dst) {
  for (uint64_t c = 0; c < n_cols; c += 64) {
    uint64_t gaps = plane_bits64(plane, n_words, first_bit + c);
    const uint64_t m = n_cols - c < 64 ? n_cols - c : 64;
    if (gaps == 0) { memcpy(dst + c, src, m); src += m; continue; }
    for (uint64_t k = 0; k < m; ++k, gaps >>= 1) {
      if (gaps & 1) dst[c + k] = '-'; else dst[c + k] = *src++;
    }
  }
}

#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw,avx512vbmi2,bmi2,popcnt")))
static void expand_vbmi2(const char *src, const uint32_t *plane, uint32_t n_words, uint64_t first_bit, uint64_t n_cols,
                         char *dst) {
  const __m512i dash = _mm512_set1_epi8('-');
  for (uint64_t c = 0; c < n_cols; c += 64) {
    const uint64_t m = n_cols - c < 64 ? n_cols - c : 64;
    const uint64_t valid = m == 64 ? ~0ull : ((1ull << m) - 1);
    const uint64_t keep = ~plane_bits64(plane, n_words, first_bit + c) & valid;   /* columns that take a character */
    if (keep == ~0ull) {   /* 64 columns without a gap: most of a read's alignment */
      _mm512_storeu_si512((void *)(dst + c), _mm512_loadu_si512((const void *)src));
      src += 64;
      continue;
    }
    const unsigned n_chars = (unsigned)__builtin_popcountll(keep);
    /* masked load: bytes past the sequence's end are never touched */
    const __m512i chars = _mm512_maskz_loadu_epi8(n_chars == 64 ? ~0ull : ((1ull << n_chars) - 1), src);
    _mm512_mask_storeu_epi8(dst + c, valid, _mm512_mask_expand_epi8(dash, keep, chars));
    src += n_chars;
  }
}
#endif

static expand_fn pick_expand(void) {
#if defined(__x86_64__)
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("avx512bw")) return expand_vbmi2;
#endif
  return expand_scalar;
}

static expand_fn g_expand;   /* benign race: every thread resolves the same pointer */

int sa_moves_uses_simd(void) {
  if (!g_expand) g_expand = pick_expand();
  return g_expand != expand_scalar;
}

/* for the tests: 0 = as detected, 1 = force the scalar loop */
void sa_moves_force_scalar(int on) { g_expand = on ? expand_scalar : pick_expand(); }

/* Global alignment (reference src/needleman_wunsch.c:82-145): n_moves walked columns in the planes, then the rest of the
 * longer sequence against gaps.  Writes result_a / result_b NUL-terminated (capacity len_a + len_b + 1 each) and the number
 * of columns.  Returns SEQALIGN_OK, or SEQALIGN_E_TRACEBACK when the planes do not describe a walk over these lengths. */
int sa_expand_nw_moves(const char *a, uint32_t len_a, const char *b, uint32_t len_b, const uint32_t *plane_a,
                       const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, char *out_a, char *out_b,
                       uint32_t *out_len) {
  if (!g_expand) g_expand = pick_expand();
  if ((uint64_t)n_moves > 32ull * n_words) return SEQALIGN_E_TRACEBACK;
  const uint64_t first = 32ull * n_words - n_moves;
  const uint64_t gaps_a = plane_popcount(plane_a, n_words, first), gaps_b = plane_popcount(plane_b, n_words, first);
  /* a walked column consumes a character of seq_a unless it is a gap in a, of seq_b unless it is a gap in b */
  const uint64_t used_a = n_moves - gaps_a, used_b = n_moves - gaps_b;
  if (used_a > len_a || used_b > len_b || planes_overlap(plane_a, plane_b, n_words, first)) return SEQALIGN_E_TRACEBACK;
  const uint64_t rest_a = len_a - used_a, rest_b = len_b - used_b;   /* where the walk stopped: (x, y) */
  /* a global walk stops at the first row OR the first column (src/needleman_wunsch.c:82-115): one of the rests is 0; and the
   * columns must fit the caller's len_a + len_b + 1 bytes (they do whenever the planes are disjoint: checked for corrupt words) */
  if ((rest_a && rest_b) || rest_a + rest_b + n_moves > (uint64_t)len_a + len_b) return SEQALIGN_E_TRACEBACK;
  uint64_t col = 0;
  /* src/needleman_wunsch.c:117-132 emits, backwards, the rest of b against gaps and then the rest of a: forwards the
   * other way round */
  if (rest_a) { memcpy(out_a, a, rest_a); memset(out_b, '-', rest_a); col = rest_a; }
  if (rest_b) { memset(out_a + col, '-', rest_b); memcpy(out_b + col, b, rest_b); col += rest_b; }
  g_expand(a + rest_a, plane_a, n_words, first, n_moves, out_a + col);
  g_expand(b + rest_b, plane_b, n_words, first, n_moves, out_b + col);
  col += n_moves;
  out_a[col] = out_b[col] = '\0';
  *out_len = (uint32_t)col;
  return SEQALIGN_OK;
}

/* Local alignment (reference src/smith_waterman.c:187-255): the hit is exactly its n_moves walked columns; it starts at
 * (pos_a, pos_b) = the end cell minus the characters the walk consumed.  end_x / end_y: the hit's end cell. */
int sa_expand_sw_moves(const char *a, const char *b, uint32_t end_x, uint32_t end_y, const uint32_t *plane_a,
                       const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, char *out_a, char *out_b,
                       uint32_t pos[4]) {
  if (!g_expand) g_expand = pick_expand();
  if ((uint64_t)n_moves > 32ull * n_words) return SEQALIGN_E_TRACEBACK;
  const uint64_t first = 32ull * n_words - n_moves;
  const uint64_t used_a = n_moves - plane_popcount(plane_a, n_words, first);
  const uint64_t used_b = n_moves - plane_popcount(plane_b, n_words, first);
  if (used_a > end_x || used_b > end_y || planes_overlap(plane_a, plane_b, n_words, first)) return SEQALIGN_E_TRACEBACK;
  pos[0] = (uint32_t)(end_x - used_a); pos[1] = (uint32_t)(end_y - used_b);   /* smith_waterman.c:251-255 */
  pos[2] = (uint32_t)used_a; pos[3] = (uint32_t)used_b;
  g_expand(a + pos[0], plane_a, n_words, first, n_moves, out_a);
  g_expand(b + pos[1], plane_b, n_words, first, n_moves, out_b);
  out_a[n_moves] = out_b[n_moves] = '\0';
  return SEQALIGN_OK;
}

/* ---- CIGAR straight from the planes (north_star: "identical CIGAR/alignment strings") -------------------
 * The reference prints the two gapped strings (src/tools/nw_cmdline.c:78-149); CIGAR is the derived format, defined in
 * include/seqalign_hip.h beside seqalign_cigar: seq_a is the query, seq_b the reference -- plane B bit ('-' in result_b) = I,
 * plane A bit ('-' in result_a) = D, neither = M (or '=' / 'X').  A run-length encoder over the bits: 64 columns per step, a run's
 * length is a count of trailing ones -- a 150-column read without gaps is three steps and the four bytes "150M", where the
 * strings are 2 x 151 bytes written and then read again by whoever wants the CIGAR. */
#include <ctype.h>

typedef struct {
  char *out;          /* NULL: count only */
  uint64_t cap, used; /* used counts every byte the CIGAR needs, written or not */
  char op;            /* the open run (0: none) */
  uint64_t run;
} cigar_sink;

static inline void cigar_flush(cigar_sink *s) {
  if (!s->op) return;
  uint64_t v = s->run;
  if (v < 100) {   /* nearly every run of a read-sized alignment: one or two digits, no division loop */
    const int n = v < 10 ? 1 : 2;
    if (s->out && s->used + (uint64_t)n + 1 < s->cap) {   /* (+ 1 op, and the NUL must still fit behind it: <) */
      char *o = s->out + s->used;
      if (n == 2) { *o++ = (char)('0' + v / 10); v %= 10; }
      o[0] = (char)('0' + v); o[1] = s->op;
    }
    s->used += (uint64_t)n + 1;
    s->op = 0; s->run = 0;
    return;
  }
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  if (s->out && s->used + (uint64_t)n + 1 < s->cap) {
    for (int k = 0; k < n; ++k) s->out[s->used + k] = tmp[n - 1 - k];
    s->out[s->used + n] = s->op;
  }
  s->used += (uint64_t)n + 1;
  s->op = 0; s->run = 0;
}

static inline void cigar_push(cigar_sink *s, char op, uint64_t n) {
  if (!n) return;
  if (s->op != op) { cigar_flush(s); s->op = op; }
  s->run += n;
}

/* the walked columns [first, first + n_cols) of the planes; a / b: the characters those columns consume, in order (format 2 only) */
static void cigar_walked(cigar_sink *s, const uint32_t *plane_a, const uint32_t *plane_b, uint32_t n_words, uint64_t first, uint64_t n_cols,
                         const char *a, const char *b, int format, int fold) {
  for (uint64_t c = 0; c < n_cols; c += 64) {
    const uint64_t m = n_cols - c < 64 ? n_cols - c : 64;
    const uint64_t valid = m == 64 ? ~0ull : ((1ull << m) - 1);
    const uint64_t ga = plane_bits64(plane_a, n_words, first + c) & valid, gb = plane_bits64(plane_b, n_words, first + c) & valid;
    const uint64_t mm = ~(ga | gb) & valid;
    uint64_t at = 0;
    while (at < m) {
      const uint64_t bit = 1ull << at;
      const uint64_t same = (ga & bit) ? ga : (gb & bit) ? gb : mm;
      const char op = (ga & bit) ? 'D' : (gb & bit) ? 'I' : 'M';
      /* trailing ones of `same` from `at` on */
      const uint64_t rest = ~(same >> at);
      uint64_t run = rest ? (uint64_t)__builtin_ctzll(rest) : 64;
      if (run > m - at) run = m - at;
      if (op == 'D') b += run;
      else if (op == 'I') a += run;
      else if (format != 2) { a += run; b += run; }
      if (op != 'M' || format != 2) cigar_push(s, op, run);
      else {
        for (uint64_t k = 0; k < run; ++k, ++a, ++b) {
          const int eq = fold ? tolower((unsigned char)*a) == tolower((unsigned char)*b) : *a == *b;
          cigar_push(s, eq ? '=' : 'X', 1);
        }
      }
      at += run;
    }
  }
}

static int cigar_finish(cigar_sink *s, uint32_t *out_len) {
  cigar_flush(s);
  *out_len = (uint32_t)s->used;
  if (!s->out) return SEQALIGN_OK;
  if (s->used + 1 > s->cap) return SEQALIGN_E_NOMEM;
  s->out[s->used] = '\0';
  return SEQALIGN_OK;
}

int sa_cigar_nw_moves(const char *a, uint32_t len_a, const char *b, uint32_t len_b, const uint32_t *plane_a,
                      const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, int format, int fold, char *out, uint64_t cap,
                      uint32_t *out_len, uint32_t *out_columns) {
  if ((uint64_t)n_moves > 32ull * n_words) return SEQALIGN_E_TRACEBACK;
  const uint64_t first = 32ull * n_words - n_moves;
  const uint64_t gaps_a = plane_popcount(plane_a, n_words, first), gaps_b = plane_popcount(plane_b, n_words, first);
  const uint64_t used_a = n_moves - gaps_a, used_b = n_moves - gaps_b;
  if (used_a > len_a || used_b > len_b || planes_overlap(plane_a, plane_b, n_words, first)) return SEQALIGN_E_TRACEBACK;
  const uint64_t rest_a = len_a - used_a, rest_b = len_b - used_b;
  if ((rest_a && rest_b) || rest_a + rest_b + n_moves > (uint64_t)len_a + len_b) return SEQALIGN_E_TRACEBACK;
  cigar_sink s = {out, cap, 0, 0, 0};
  /* forwards: the rest of seq_a against gaps (I), or the rest of seq_b against gaps (D) -- sa_expand_nw_moves,
   * src/needleman_wunsch.c:117-132 -- then the walked columns */
  cigar_push(&s, 'I', rest_a);
  cigar_push(&s, 'D', rest_b);
  cigar_walked(&s, plane_a, plane_b, n_words, first, n_moves, a + rest_a, b + rest_b, format, fold);
  if (out_columns) *out_columns = (uint32_t)(rest_a + rest_b + n_moves);
  return cigar_finish(&s, out_len);
}

int sa_cigar_sw_moves(const char *a, const char *b, uint32_t end_x, uint32_t end_y, const uint32_t *plane_a,
                      const uint32_t *plane_b, uint32_t n_words, uint32_t n_moves, int format, int fold, char *out, uint64_t cap,
                      uint32_t pos[4], uint32_t *out_len) {
  if ((uint64_t)n_moves > 32ull * n_words) return SEQALIGN_E_TRACEBACK;
  const uint64_t first = 32ull * n_words - n_moves;
  const uint64_t used_a = n_moves - plane_popcount(plane_a, n_words, first);
  const uint64_t used_b = n_moves - plane_popcount(plane_b, n_words, first);
  if (used_a > end_x || used_b > end_y || planes_overlap(plane_a, plane_b, n_words, first)) return SEQALIGN_E_TRACEBACK;
  pos[0] = (uint32_t)(end_x - used_a); pos[1] = (uint32_t)(end_y - used_b);
  pos[2] = (uint32_t)used_a; pos[3] = (uint32_t)used_b;
  cigar_sink s = {out, cap, 0, 0, 0};
  cigar_walked(&s, plane_a, plane_b, n_words, first, n_moves, a + pos[0], b + pos[1], format, fold);
  return cigar_finish(&s, out_len);
}
