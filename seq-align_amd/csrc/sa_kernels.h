/* sa_kernels.h -- launch interface between sa_device.hip and the kernels. */
#ifndef SA_KERNELS_H
#define SA_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seqalign_hip.h"
extern "C" {
#include "sa_internal.h"   /* SA_F_* */
}

/* ---- the direction byte in its LOCAL form (round 6, second half) -----------------------------------------------------
 * The direction byte of sa_fill_dirs.hip answers, per state, "in which state does a walk that leaves this cell ARRIVE" -- for MATCH
 * and GAP_A that is a fact about the NEIGHBOUR (which of M / A / B is the maximum of the cell up-left; whether B >= M in the cell
 * above), which the fills carry over from the previous row and then convert, cell by cell, from comparison masks into two-bit
 * codes: select upon select -- 15-25 of a packed cell's ~115-165 issue cycles (profiles/r06/r06_local_dirs.txt).  A walk visits
 * ~170 of a pair's 22 801 cells.  So where ONLY tile walkers read the bytes (seqalign_nw_batch's moves path and the best-hit path:
 * every BASELINE config's host-level call but the multi-hit one) the fill stores each cell's OWN five
 * comparisons as five raw bits -- the sign bit of each saturating difference shifted into place, one v_lshrrev + one v_bitop3
 * per decision, no mask, no select, no carried tags -- and the walker reads the state it arrives in from the byte of the cell it
 * arrives at (it reads that byte anyway: it is the next step's):
 *     bit 0  GA  gap_a >= max(match, gap_b)        a walk that ARRIVES here by a MATCH move is in GAP_A if GA, else GAP_B if BM,
 *     bit 1  BM  gap_b >= match                     else MATCH (alignment.c:311-327's order); a GAP_A walk that opened its gap in
 *                                                   the cell below arrives in GAP_B if BM, else MATCH
 *     bit 2  CA  gap_a(above) + extend == gap_a     a GAP_A walk LEAVING this cell stays in GAP_A
 *     bit 3  FA  gap_a(left) + open + extend == gap_b   a GAP_B walk leaving this cell arrives in GAP_A,
 *     bit 4  FB  gap_b(left) + extend == gap_b          else GAP_B if FB, else MATCH
 *     bits 5 6 7 (Smith-Waterman)  this cell's match / gap_a / gap_b score is 0: a walk standing here in that state ends
 * The lane walkers look a cell AHEAD -- where a walk goes next must not wait for the byte of the cell it arrives at -- and keep the
 * older form (they walked launches of >= SA_WALK_TILE_MAX walks until the tile walks on the local form drew level with them at every
 * size: option trace_kernel = lane, or dirs_local = 0, brings them back); so do the multi-hit path's bytes, whose sweep routes
 * arrivals by them.  SaFillParams::dirs_local /
 * SaTraceParams::dirs_local say which form a launch writes / reads (option dirs_local = 0: the older form everywhere). */
#define SA_LD_GA 1u
#define SA_LD_BM 2u
#define SA_LD_CA 4u
#define SA_LD_FA 8u
#define SA_LD_FB 16u
#define SA_LD_END0 32u   /* << state */

/* Everything one fill launch needs; passed by value as the kernarg. */
/* ---- the direction bytes in BLOCKS (round 6) --------------------------------------------------------------------------
 * Where nobody but the walkers reads them -- seqalign_nw_batch's directions-only fills and the packed SW best-hit fill; not the
 * multi-hit path, whose sweep reads rows -- a pair's direction bytes are laid out in blocks of 8 rows x 16 columns = one 128-byte
 * line, block rows first:
 *     byte of cell (x, y)  at  ((y / 8) * nbx + x / 16) * 128 + (y % 8) * 16 + x % 16,      nbx = ceil((len_a + 1) / 16)
 * Row-major at a pitch of len_a + 1 bytes, a walk that climbs a row per step pulls in a new 128-byte line per step (measured:
 * 1.27 lines per step, 278 MB for C2's 10 000 walks, the walkers bound by exactly that: profiles/r06/r06_walkers.txt); in blocks a
 * diagonal walk stays 8-16 steps in a line.  The fills write a block row (8 rows: nbx x 128 contiguous bytes) at a time out of an
 * LDS buffer that is laid out the same way.  Rows of up to 512 columns (wider: row-major as before).  SA_DIRS_BLOCKED=0 (make exp)
 * builds the row-major form for an A/B. */
#ifndef SA_DIRS_BLOCKED
#define SA_DIRS_BLOCKED 1
#endif
static inline bool sa_dirs_blocked_shape(uint32_t max_len_a) { return SA_DIRS_BLOCKED != 0 && max_len_a + 1u <= 512u; }
static inline uint64_t sa_dirs_blocked_bytes(uint32_t len_a, uint32_t len_b) {   /* a multiple of 128 */
  return (uint64_t)((len_b + 8u) >> 3) * (uint64_t)((len_a + 16u) >> 4) * 128u;
}

struct SaFillParams {
  const uint8_t *arena;
  const uint64_t *off_a;
  const uint32_t *len_a;
  const uint64_t *off_b;
  const uint32_t *len_b;
  const uint64_t *mat_off;
  int32_t *M, *A, *B;
  uint64_t *status;
  const uint16_t *code;   /* [256] raw char -> folded char | class << 8 */
  const int32_t *table;   /* [K*K] */
  uint32_t n_pairs;
  uint32_t K;
  int32_t gap_open, open1, ext, floor, gen_eq, gen_ne;
  uint32_t flags;         /* SA_F_* (sa_internal.h) */
  /* optional (stream kernel, SW): the best match_scores cell per pair in the reference's
   * hit order (score desc, column asc, index asc), as sa_reduce.hip reports it -- saves the
   * separate pass over match_scores when only the best hit is wanted */
  int32_t *best_score;
  uint64_t *best_index;
  /* optional (stream kernel, SW): what the multi-hit path needs to know about match_scores before it sweeps the
   * matrices backwards (sa_sw_sweep.hip) -- whether any cell has score >= max(cand_min[pair], 1) (the candidates of
   * smith_waterman.c:152-156; cand_count != 0) and bounds of where they are: cand_box[4*pair..] = first row, last row,
   * a column at or below the lowest, a column at or above the highest candidate column. */
  const int32_t *cand_min;
  uint32_t *cand_count;
  uint32_t *cand_box;
  uint32_t *cand_rows;          /* the multi-hit path's scratch arena (SaSweepParams::hit_keys) as uint32: pair p's part starts
                                    with its rows' candidate columns -- lowest / highest per row, rows 0..len_b (lo > hi: none) */
  const uint64_t *cand_rows_off; /* [n] where pair p's part starts, in uint64 (SaSweepParams::hit_off)                    */
  /* launch tuning (host side only; the context's options, sa_ctx.hpp SaOptions): 0 = the launcher's own choice */
  uint32_t tune_cpl, tune_wpb, tune_lds_pad;
  uint32_t tune_quad = 0;   /* host side only: the packed fills with four pairs per wave: 0 = by shape and size, 1 = never, 2 = whenever the shape allows (option quad) */
  /* != 0: every pair of the launch has the same len_a and len_b and pair k's cells start at mat_off[0] + k * uniform_stride
   * (>= its cell count); the packed two-pairs-per-wave fills (sa_fill_dirs_x2.hip) need a multiple of 256 */
  uint64_t uniform_stride;
  const uint32_t *pair_list;    /* optional (the direction fills of seqalign_nw_batch): the launch takes pairs pair_list[0 .. n_pairs)
                                   of the descriptor arrays instead of pairs 0 .. n_pairs -- a chunk whose pairs are MOSTLY of one
                                   shape goes through the packed kernel with the list of those, the others through the one-pair
                                   kernel with the list of the rest */
  int32_t table_abs_max;        /* host side only: the largest |entry| of the K x K table (0 for K <= 1): the packed fills' int16 bound */
  uint32_t dirs_local = 0;      /* host side only: the NW / best-hit direction fills write the LOCAL form of the byte (above) */
};

/* How the multi-hit path packs a match_scores cell into a 64-bit key whose ascending order IS the reference's hit
 * order (score desc, column asc, then cell index = row asc; smith_waterman.c:71-86):
 *     key = (cap - score) << (row_bits + col_bits) | column << row_bits | row
 * cap = an upper bound of every score of the chunk; row_bits + col_bits + score_bits <= 63. */
struct SaKeyLayout {
  int32_t cap;
  uint32_t row_bits, col_bits, score_bits;
};

struct SaReduceParams {
  const uint32_t *len_a, *len_b;
  const uint64_t *mat_off;
  const int32_t *M;
  int32_t min_score;
  int32_t *best_score;
  uint64_t *best_index;
  uint32_t *cand_count;
  const uint64_t *cand_off;
  const uint32_t *cand_cap;
  uint32_t *cand_index;
  int32_t *cand_score;
  uint32_t n_pairs;
  uint32_t slices;   /* > 1 (best cell only, few long pairs): that many waves per pair (sa_reduce.hip) */
  uint32_t tune_depth; /* host side only: 0 = by the launch's size, 4 / 8 = KiB a wave loads per step (option reduce_depth) */
};

/* candidate count and bounding box per pair, as the stream fill (SA_STREAM_CAND) or sa_launch_sw_box leaves them */
struct SaCandBox {
  uint32_t *cand_count;       /* [n]                                                       */
  uint32_t *cand_box;         /* [4n] rmin, rmax, cmin, cmax                               */
  uint32_t *cand_rows;        /* the multi-hit path's scratch arena, as uint32 (SaFillParams::cand_rows) */
  const uint64_t *hit_off;    /* [n + 1] where pair p's part of it starts, in uint64       */
  const int32_t *cand_min;    /* [n] per-pair min_score                                    */
  uint8_t *dirs;              /* optional: room for one byte per cell (same cell offsets as the matrices).  If the batch is in
                                 sa_fill_dirs.hip's domain the fill writes match_scores + directions there INSTEAD of the three
                                 matrices and sets *dirs_used (gap_a / gap_b are then not written at all)                     */
  bool *dirs_used;
  bool best_only;             /* dirs + dirs_used as above, but for the best-hit path: no candidate outputs (the cand_* members
                                 are NULL), the fill writes ONLY the direction bytes and reports the best cell through the
                                 best_score / best_index arguments (sa_fill_dirs_x2.hip: fill_sw_best_x2_kernel); the chunk's
                                 three matrices are not even allocated */
  uint64_t uniform_stride;    /* != 0: the chunk's layout is SaFillParams::uniform_stride's (every pair the same shape, cells
                                 k * uniform_stride apart): the packed two-pairs-per-wave fill may take it (sa_fill_dirs_x2.hip) */
  const uint32_t *pair_list;  /* != NULL (ragged chunk, every pair's cells on a multiple of 256, uniform_stride = 256): the packed fill takes
                                 list_count / 2 waves, wave u the pairs pair_list[2u], pair_list[2u + 1] of the descriptor arrays -- equal in
                                 shape, or the same pair twice (it has the wave to itself)                                               */
  uint32_t list_count;
  bool dirs_local;            /* best_only: the fill writes the LOCAL form of the direction byte (tile walkers behind it) */
};

/* SW multi-hit enumeration: the reverse sweep (sa_sw_sweep.hip) */
struct SaSweepParams {
  const uint8_t *arena;
  const uint64_t *off_a;
  const uint32_t *len_a;
  const uint64_t *off_b;
  const uint32_t *len_b;
  const uint64_t *mat_off;
  const int32_t *M, *A, *B;      /* OVER-READ (sw_sweep_dirs_ev_kernel, branchless row loads): up to CPL * 4 <= 64 bytes of M and
                                    CPL + 3 <= 19 bytes of `dirs` BEHIND the last pair's last cell are loaded (never used): the
                                    caller's arenas must extend that far (sa_host::reserve_arenas and the dirs buffers add 4 KiB) */
  const uint16_t *code;
  const int32_t *table;
  const uint32_t *cand_count;    /* [n]                                                              */
  const uint32_t *cand_box;      /* [4n]                                                             */
  const int32_t *min_score;      /* [n]                                                              */
  unsigned long long *hit_keys;  /* scratch arena; pair p's part is elements [hit_off[p], hit_off[p + 1]): first len_b + 1
                                    elements = its rows' candidate columns (in: the fill / sa_launch_sw_box), then (out) its
                                    hits' keys: ascending when hit_count <= 64, else in the order the sweep met them
                                    (SA_SWEEP_UNSORTED)                                                           */
  const uint64_t *hit_off;       /* [n + 1]; room for hits: a hit owns at least ceil(min_score / best move) + 1 cells  */
  uint32_t *hit_count;           /* [n] every hit of the pair (no max_hits here)                      */
  uint32_t *status;              /* [n] 0, SEQALIGN_E_* of a walk (see err_key), | SA_SWEEP_UNSORTED   */
  unsigned long long *err_key;   /* [n] key of the first (lowest) walk that met the error             */
                                 /* (strips: the caller zeroes hit_count / status and sets err_key to ~0)            */
  uint32_t lds_columns;          /* one wave per pair, wide rows: the winners of two rows live in LDS, sized for this
                                    many columns (>= every pair's len_a + 1)                                         */
  /* one wave per strip of strip_columns columns (few wide pairs; strip_progress != NULL selects it): */
  uint32_t *strip_progress;      /* [2 * sa_sweep_strip_blocks() + 1] zeroed: per (pair, strip) rows done | end, then
                                    the ticket counter                                                               */
  uint32_t strip_columns;        /* 64, 128 or 256                                                                   */
  uint32_t strip_interval;       /* rows between two publications of a strip's progress: 16 or 64                    */
  uint32_t strips_per_pair;      /* sa_sweep_strips_per_pair(max_len_a, strip_columns)                               */
  unsigned long long *bnd;       /* per pair, row (counted from the box's last row) and strip: the winner of the strip's
                                    first column, 2 uint64; pair p at 2 * row_off[p] * strips_per_pair               */
  const uint64_t *row_off;       /* [n] prefix of len_b + 1                                                          */
  uint32_t n_pairs, K;
  int32_t open1, ext, gen_eq, gen_ne;
  uint32_t flags;
  uint32_t max_len_a;            /* of the chunk: picks the kernel                                    */
  uint32_t max_len_b;            /* of the chunk (0: unknown): with max_len_a the radices of the sweep's mixed-radix 32-bit word  */
  SaKeyLayout layout;
  unsigned long long *trace;     /* optional [8n]: cycles, rows, active row segments, rounds, cycles in active segments (option sweep_trace) */
  uint32_t tune_ev;              /* host side only: the direction-byte sweep's second form (walks as key << 2 | state words; option sweep_ev) */
  uint32_t tune_cpl;             /* host side only: 1, 2, 4 forces the LDS form with segments of 64 * that many columns (option sweep_cpl) */
  const uint8_t *dirs;           /* != NULL: the matrices were filled by sa_fill_dirs.hip -- M holds match_scores, dirs one byte
                                    of directions per cell (same cell offsets), A / B are not used                           */
};
#define SA_SWEEP_UNSORTED 0x80000000u
#define SA_SWEEP_OVERFLOW 0x40000000u   /* more hits than the pair's part of the arena holds (cannot happen: see hit_off) */
/* widest pair (columns) whose two rows of records fit LDS (12 B per column and row with 64-bit keys) */
#define SA_SWEEP_LDS_COLUMNS 2048u
uint32_t sa_sweep_strips_per_pair(uint32_t max_len_a, uint32_t strip_columns);
uint32_t sa_sweep_strip_blocks(uint32_t n_pairs, uint32_t max_len_a, uint32_t strip_columns);

struct SaTraceParams {
  const uint8_t *arena;
  const uint64_t *off_a;
  const uint32_t *len_a;
  const uint64_t *off_b;
  const uint32_t *len_b;
  const uint64_t *mat_off;
  const int32_t *M, *A, *B;
  const uint16_t *code;
  const int32_t *table;
  const uint64_t *str_off;
  char *out_a, *out_b;
  uint32_t *out_head, *out_len;
  int32_t *out_score;
  uint32_t *trace_status;
  /* pipelined host-level calls (sa_batch.hip): when out_meta4 != NULL the four per-walk words go there instead,
   * interleaved -- head, len, score, status of walk w at out_meta4[4w..] -- so that one sub-batch's results are ONE
   * contiguous D2H; fill_status (optional, [n] per pair): a fill status != ~0 (a character pair without a score,
   * alignment_scoring.c:178-181) becomes the walk's status SEQALIGN_E_UNKNOWN_PAIR, which saves copying the fill's
   * status words back separately */
  uint32_t *out_meta4;
  const uint64_t *fill_status;
  const uint64_t *start_index; /* SW: end cell of the hit per pair; NULL = NW        */
  uint32_t *out_pos;           /* SW: [4*n] pos_a, pos_b, len_a, len_b               */
  /* SW multi-hit path: n_pairs WALKS, walk w = hit walker_rank[w] of pair walker_pair[w], ending at the cell packed in
   * hit_keys[mat_off[pair] + rank] (layout); str_off and every out_* array are indexed by the walk */
  const uint32_t *walker_pair, *walker_rank;
  const unsigned long long *hit_keys;
  const uint64_t *hit_off;     /* (SaSweepParams::hit_off: pair p's keys start at hit_off[p] + len_b + 1) */
  SaKeyLayout layout;
  uint32_t n_pairs, K;
  int32_t open1, ext, gen_eq, gen_ne;
  uint32_t flags;
  uint32_t tune_walker;        /* host side only: 0 = by batch shape, 1 = one lane per walk, 2 = one wave per walk (option trace_kernel) */
  uint32_t dirs_blocked;       /* `dirs` is laid out in blocks of 8 x 16 cells (above) instead of row-major at pitch len_a + 1 */
  uint32_t tune_group;         /* host side only: the tile walker on moves: 0 / 4 = four walks per wave in lockstep, 8 = eight, 1 = one (option walk_group) */
  uint32_t tune_tile;          /* host side only: the local tile walker's tile edge: 32 or 64 bytes; 0 = 32 for NW walks, 64 for SW ones (option walk_tile) */
  uint32_t tune_stage;         /* host side only: the local tile walker writes a wave's moves as one run out of LDS (option walk_stage) */
  uint32_t dirs_local;         /* `dirs` holds the LOCAL form of the direction byte (above): tile walkers on moves only */
  const uint8_t *dirs;         /* SW multi-hit path behind sa_fill_dirs.hip: walks follow the direction bytes (hit_keys != NULL) */
  const int32_t *nw_score;     /* NW behind the directions-only fill (dirs != NULL): per pair the end cell's score ...            */
  const uint64_t *nw_state;    /* ... and the matrix the walk starts in (0 MATCH, 1 GAP_A, 2 GAP_B)                                */
  const int32_t *start_score;  /* SW walks on direction bytes from start_index (the best-hit path behind fill_sw_best_x2_kernel, which
                                  writes no match_scores): the start cell's score                                                  */
  /* walks on direction bytes that send home MOVES instead of strings (host/sa_moves.c): per walked column one bit "gap in
   * seq_a" (the walk stood in GAP_A) and one bit "gap in seq_b", forward column order, right-aligned in the walk's slot --
   * walk w owns words [2 ((str_off[w] >> 5) + w), + 2 nw) of `moves`, nw = (len_a + len_b + 31) >> 5: plane A, then plane B
   * (str_off is the prefix of len_a + len_b, so the slots do not overlap); out_a / out_b / out_meta4 are not used, the two
   * words of walk w go to out_meta2[2w..]: score, then the number of walked columns -- or 0xFFFFFFF0 | SEQALIGN_E_* .
   * SW walks also need out_pos for nothing: the host derives the hit's position from the planes (sa_expand_sw_moves). */
  uint32_t *moves;
  uint32_t stage_words = 0;    /* >= (len_a + len_b + 31) >> 5 of every walk of the launch (0 = unknown): the one-lane-per-walk kernel keeps a walk's
                                  finished words in LDS until it is over (up to 96 words: 3 040 columns) */
  uint32_t *out_meta2;
  /* SW walks that send home moves: out_meta4[4w..] = score, walked columns, end cell x, y.  walks_per_pair = k > 0: the launch has
   * k walks per pair, walk w = hit w % k of pair w / k (its key in hit_keys), in the slot 2 k ((str_off[pair] >> 5) + pair) +
   * 2 (w % k) nw -- str_off then indexed by PAIR -- and walks beyond hit_count[pair], or of a pair with sweep_status != 0,
   * return without a trace: the launch can be enqueued before anybody has seen the sweep's counts */
  uint32_t walks_per_pair;
  const uint32_t *hit_count, *sweep_status;
};
/* direction-byte walks: one wave per walk (LDS tiles) below this many walks per launch, one lane per walk from there on (sa_traceback.hip) */
#define SA_WALK_TILE_MAX 24576u
#define SA_MOVES_ERR 0xFFFFFFF0u

/* ---- which kernels a call launched (seqalign_ctx_last_call_info, include/seqalign_hip.h: SEQALIGN_K_*) ------------------
 * Every launcher below reports what it launches and for how many pairs / walks to the calling thread's recorder
 * (sa_device.hip), which the C-ABI entry points point at their context for the duration of the call. */
void sa_record_launch(int kind, uint64_t items);

/* substitution lookup flavour */
enum { SA_SUBST_SIMPLE = 0, SA_SUBST_LDS = 1, SA_SUBST_GLOBAL = 2 };
#define SA_LDS_TABLE_MAX_K 64

/* ---- the DOMAINS of the direction-byte fills: pure functions of (flattened scoring, shape) ---------------------------
 * Which scorings and shapes a kernel family takes is a property of the scoring's numbers and the pairs' lengths alone; the
 * launchers add what depends on the launch (output pointers present, block-aligned), and the host-level entry points ask the
 * same functions BEFORE they lay a chunk out (sa_device.hip: *_applicable) -- from the flattened scoring, no launch needed. */
struct SaScoringTraits {
  uint32_t flags, K;
  int32_t gap_open, open1, ext, gen_eq, gen_ne, table_abs_max;
};
inline SaScoringTraits sa_traits_of(const SaFillParams &p) {
  return SaScoringTraits{p.flags, p.K, p.gap_open, p.open1, p.ext, p.gen_eq, p.gen_ne, p.table_abs_max};
}
/* the row sweeps' GENERAL path (free / forbidden gaps, sentinel scores, gap_open > 0): none of the direction fills takes it */
inline bool sa_scoring_needs_general(const SaScoringTraits &t) {
  return (t.flags & (SA_F_NO_END_GAP | SA_F_NO_GAPS_A | SA_F_NO_GAPS_B | SA_F_HAS_SENTINEL)) || t.open1 > t.ext;
}
/* rows the direction fills keep in one wave's registers: 16 columns per lane (round 5; 8 before).  The SW sweep behind them
 * keeps a row in registers too: up to 512 columns in either of its forms, 513 .. 1 024 in the one-word form only
 * (sw_sweep_dirs_ev_kernel, keys of <= 62 bits) -- the caller (sa_batch_sw.hip) offers a direction arena only when that holds */
/* NW, directions only: no flag at all, no sentinel, gap_open <= 0, gap_extend <= 0, a table that fits LDS */
inline bool sa_domain_nw_dirs_row(uint32_t max_len_a) { return max_len_a + 1 <= 16 * 64; }
inline bool sa_domain_nw_dirs(const SaScoringTraits &t, uint32_t max_len_a) {
  return t.flags == 0 && !sa_scoring_needs_general(t) && t.K <= SA_LDS_TABLE_MAX_K && t.ext <= 0 && sa_domain_nw_dirs_row(max_len_a);
}
/* SW, match_scores + directions / directions + best cell: the same without the start-gap / mismatch flags */
inline bool sa_domain_sw_dirs(const SaScoringTraits &t, uint32_t max_len_a) {
  return (t.flags & SA_F_IS_SW) && !sa_scoring_needs_general(t) && !(t.flags & (SA_F_NO_START_GAP | SA_F_NO_MISMATCH)) &&
         t.K <= SA_LDS_TABLE_MAX_K && t.ext <= 0 && sa_domain_nw_dirs_row(max_len_a);
}
/* two pairs per wave in packed int16: every score the recurrence can produce on pairs up to max_len_a x max_len_b, de-trended
 * or not, stays inside int16 */
inline bool sa_domain_x2_scores_fit(const SaScoringTraits &t, uint32_t max_len_a, uint32_t max_len_b) {
  auto mag = [](int64_t v) { return v < 0 ? -v : v; };
  int64_t pen = mag(t.gen_eq) > mag(t.gen_ne) ? mag(t.gen_eq) : mag(t.gen_ne);
  if (mag(t.open1) > pen) pen = mag(t.open1);
  if (mag(t.ext) > pen) pen = mag(t.ext);
  if (mag(t.gap_open) + mag(t.ext) > pen) pen = mag(t.gap_open) + mag(t.ext);
  if (t.K > 1 && (int64_t)t.table_abs_max > pen) pen = t.table_abs_max;
  return ((int64_t)max_len_a + max_len_b + 2) * pen + ((int64_t)max_len_a + 1) * mag(t.ext) <= 30000;
}
/* ... and for the packed NW fills (round 6), which keep every value V of cell (g, j) as V - (g + j) gap_extend (the recurrence of gap_a
 * loses its "+ extend", gap_b's scan its re-trend: two packed adds per cell): |V'| <= (la + lb + 2) (pen + |extend|) */
inline bool sa_domain_nw_x2_scores_fit(const SaScoringTraits &t, uint32_t max_len_a, uint32_t max_len_b) {
  auto mag = [](int64_t v) { return v < 0 ? -v : v; };
  int64_t pen = mag(t.gen_eq) > mag(t.gen_ne) ? mag(t.gen_eq) : mag(t.gen_ne);
  if (mag(t.open1) > pen) pen = mag(t.open1);
  if (mag(t.ext) > pen) pen = mag(t.ext);
  if (mag(t.gap_open) + mag(t.ext) > pen) pen = mag(t.gap_open) + mag(t.ext);
  if (t.K > 1 && (int64_t)t.table_abs_max > pen) pen = t.table_abs_max;
  return ((int64_t)max_len_a + max_len_b + 2) * (pen + mag(t.ext)) <= 30000;
}
inline bool sa_domain_nw_dirs_x2(const SaScoringTraits &t, uint32_t la, uint32_t lb) {
  return sa_domain_nw_dirs(t, la) && sa_domain_nw_x2_scores_fit(t, la, lb);
}
inline bool sa_domain_sw_dirs_x2(const SaScoringTraits &t, uint32_t la, uint32_t lb) {
  return sa_domain_sw_dirs(t, la) && sa_domain_x2_scores_fit(t, la, lb);
}
/* (the best-hit fill has no sweep behind it: like the NW fill it takes rows up to 1 024 columns -- round 5) */
inline bool sa_domain_sw_best_x2(const SaScoringTraits &t, uint32_t la, uint32_t lb) {
  return (t.flags & SA_F_IS_SW) && !sa_scoring_needs_general(t) && !(t.flags & (SA_F_NO_START_GAP | SA_F_NO_MISMATCH)) &&
         t.K <= SA_LDS_TABLE_MAX_K && t.ext <= 0 && sa_domain_nw_dirs_row(la) && lb < 32768 && sa_domain_x2_scores_fit(t, la, lb);
}

/* returns hipSuccess or the launch error; never synchronises */
hipError_t sa_launch_fill_wavefront(const SaFillParams &p, uint32_t max_len_a,
                                    hipStream_t stream);
hipError_t sa_launch_fill_rowscan(const SaFillParams &p, uint32_t max_len_a,
                                  hipStream_t stream);
/* LDS-ring stream writer; only when sa_stream_kernel_applicable() */
bool sa_stream_kernel_applicable(const SaFillParams &p, uint32_t max_len_a);
hipError_t sa_launch_fill_stream(const SaFillParams &p, uint32_t max_len_a,
                                 hipStream_t stream);
/* whether sa_launch_fill_stream would also report the candidates' count and box (p.cand_*) */
bool sa_stream_kernel_emits_candidates(const SaFillParams &p, uint32_t max_len_a);
/* whether sa_launch_fill_stream would also fill p.best_score / p.best_index */
bool sa_stream_kernel_reports_best(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b);
/* few long pairs: the column strips of a pair as a pipeline of waves (any len_a);
 * progress = 8 * ceil(n_pairs / 8) * sa_fill_strips_per_pair(max_len_a) + 1 uint32 of scratch (the last one is
 * the ticket counter, sa_fill_strips.hip) */
uint32_t sa_fill_strips_per_pair(uint32_t max_len_a);
hipError_t sa_launch_fill_strips(const SaFillParams &p, uint32_t max_len_a, uint32_t *progress, hipStream_t stream);
/* long rows (1024..4095 columns), fast-path scorings: one workgroup per pair, shared LDS ring */
bool sa_wgstream_kernel_applicable(const SaFillParams &p, uint32_t max_len_a);
hipError_t sa_launch_fill_wgstream(const SaFillParams &p, uint32_t max_len_a, hipStream_t stream);
bool sa_wgstream_kernel_reports_best(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b);
bool sa_wgstream_kernel_emits_candidates(const SaFillParams &p, uint32_t max_len_a);
/* the SW multi-hit path's own fill (sa_fill_dirs.hip): match_scores + one byte of directions per cell into `dirs`
 * (same cell offsets as the matrices), candidates reported as by the stream kernel; plain SW scorings, rows <= 1 024 columns */
bool sa_dirs_fill_applicable(const SaFillParams &p, uint32_t max_len_a, const uint8_t *dirs);
hipError_t sa_launch_fill_dirs(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream);
/* seqalign_nw_batch's own fill: ONLY the directions (1 B per cell) + per pair the end cell's score (p.best_score) and state
 * (p.best_index); plain NW scorings (no flag), rows <= 1 024 columns */
bool sa_nw_dirs_fill_applicable(const SaFillParams &p, uint32_t max_len_a, const uint8_t *dirs);
hipError_t sa_launch_fill_nw_dirs(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream);
/* the same fills with two pairs per wave in packed int16 (sa_fill_dirs_x2.hip): uniform batches, match / mismatch scorings,
 * scores inside int16 */
bool sa_x2_scores_fit(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b);
bool sa_nw_dirs_x2_applicable(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b, const uint8_t *dirs);
hipError_t sa_launch_fill_nw_dirs_x2(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream);
hipError_t sa_launch_fill_nw_dirs_mixed(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, uint32_t n_modal, uint32_t n_rest,
                                        hipStream_t stream);
bool sa_sw_best_x2_applicable(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b, const uint8_t *dirs);
hipError_t sa_launch_fill_sw_best_x2(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream);
bool sa_dirs_x2_applicable(const SaFillParams &p, uint32_t max_len_a, uint32_t max_len_b, const uint8_t *dirs);
hipError_t sa_launch_fill_dirs_x2(const SaFillParams &p, uint32_t max_len_a, uint8_t *dirs, hipStream_t stream);
hipError_t sa_launch_sw_reduce(const SaReduceParams &p, hipStream_t stream);
/* candidates' count and box from match_scores already in HBM (fills that cannot report them themselves): one
 * pass over M */
hipError_t sa_launch_sw_box(const SaReduceParams &p, const SaCandBox &c, uint32_t max_len_b, hipStream_t stream);
/* SW multi-hit enumeration (sa_sw_sweep.hip): every hit of every pair in one reverse sweep, then one traceback
 * per wanted hit (sa_launch_nw_traceback with SaTraceParams::hit_keys), then the strings packed back to back (walk
 * w's len[w] chars at head[w] of its slot to dst_off[w]) */
hipError_t sa_launch_sw_sweep(const SaSweepParams &p, hipStream_t stream);
hipError_t sa_launch_gather_hits(const char *src_a, const char *src_b, const uint64_t *walker_str, const uint32_t *head,
                                 const uint32_t *len, const uint64_t *dst_off, char *dst_a, char *dst_b, uint32_t n_walkers,
                                 hipStream_t stream);
hipError_t sa_launch_nw_traceback(const SaTraceParams &p, hipStream_t stream);
/* the three matrix arenas, placed and checked (sa_placement.hip) */
#define SA_ARENA_MAX_TRIES 96
struct SaArenaInfo {   /* = seqalign_arena_info_t (include/seqalign_hip.h) */
  float quality, target;
  int32_t vmm;
  uint32_t chunk_mib;
  float depth_gib, scanned_gib, depth_a_gib;
  uint32_t tries, second_walk_from;
  float try_quality[SA_ARENA_MAX_TRIES], try_depth_gib[SA_ARENA_MAX_TRIES];
  float kept_gib;
  float seconds;
};
struct SaPlacementOpts {
  size_t scan_bytes;     /* device memory the walk may hold transiently besides the arenas; 0 = allocate plainly */
  float quality_stop;    /* probe ratio that ends the walk */
  float free_fraction;   /* ... and never more than this share of the memory free at its start (0: the default, 0.6) */
  size_t keep_bytes;     /* how much of the walk's unused chunks stays with the process (the chunk pool) instead of going back */
};
/* the per-device chunk pool (sa_placement.hip): large scratch buffers mapped from chunks a walk left behind */
void *sa_pool_alloc(int device, size_t bytes);   /* NULL: the pool cannot serve it (allocate plainly) */
bool sa_pool_free(void *ptr);                    /* false: not a pool buffer (hipFree it) */
size_t sa_pool_bytes(int device);
void sa_pool_trim(int device, size_t keep_bytes);
void sa_pool_ref(int device);
void sa_pool_unref(int device);
struct SaArenaSet;
hipError_t sa_arenas_create(int device, size_t bytes, hipStream_t stream, const SaPlacementOpts &opt, SaArenaSet **out);
void sa_arenas_destroy(SaArenaSet *s);
SaArenaSet *sa_arenas_take(void *base0);   /* removes the set from the registry (seqalign_arenas_free); NULL if unknown */
const SaArenaInfo *sa_arenas_info(const SaArenaSet *s);
bool sa_arenas_copy_info(void *base0, SaArenaInfo *out);   /* false: not arenas of sa_arenas_create */
void *const *sa_arenas_base(const SaArenaSet *s);
size_t sa_arenas_bytes(const SaArenaSet *s);
/* DPP self-test: out[l] = value shifted in from lane l-1 (lane 0 gets `fill`) */
hipError_t sa_launch_dpp_probe(int32_t *out64, int32_t fill, hipStream_t stream);

#endif
