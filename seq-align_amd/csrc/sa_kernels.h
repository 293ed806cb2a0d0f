/* sa_kernels.h -- launch interface between sa_device.hip and the kernels. */
#ifndef SA_KERNELS_H
#define SA_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* Everything one fill launch needs; passed by value as the kernarg. */
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
  /* optional (stream kernel, SW): candidate emission for the multi-hit path.  Every match_scores cell
   * with score >= max(cand_min[pair], 1) is appended, in row-major order, to the pair's key list, which
   * starts at ELEMENT mat_off[pair] of cand_key (capacity = the pair's cell count, so no sizing pass):
   *     key = (key_cap - score) << (key_row_bits + key_col_bits) | column << key_row_bits | row
   * uint32 elements, or uint64 when key64 (the fields do not fit 32 bits).  Ascending key order IS the
   * reference's hit order (score desc, column asc, then cell index = row asc; smith_waterman.c:71-86).
   * cand_box[4*pair..] = first row, last row, lowest column, highest column holding a candidate. */
  const int32_t *cand_min;
  void *cand_key;
  uint32_t *cand_count;
  uint32_t *cand_box;
  int32_t key_cap;
  uint32_t key_row_bits, key_col_bits, key64;
};

/* how seqalign_sw_batch's multi-hit path lays out a candidate key (see SaFillParams) */
struct SaKeyLayout {
  int32_t cap;
  uint32_t row_bits, col_bits, score_bits, key64;
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
};

/* one SW hit as the enumeration kernel reports it (smith_waterman.c:249-255) */
struct SaDevHit {
  int32_t score;
  uint32_t pos_a, pos_b, len_a, len_b, length;
  uint32_t str_off;       /* into the pair's string slot */
};

/* candidate keys of a batch, as the stream fill (SA_STREAM_CAND) or sa_launch_sw_emit leaves them */
struct SaCandKeys {
  void *keys;                 /* pair p: elements [mat_off[p], mat_off[p] + cand_count[p]) */
  void *tmp;                  /* same size: the sort's second buffer                       */
  uint32_t *cand_count;       /* [n]                                                       */
  uint32_t *cand_box;         /* [4n] rmin, rmax, cmin, cmax                               */
  const int32_t *cand_min;    /* [n] per-pair min_score                                    */
  SaKeyLayout layout;
};

struct SaSortParams {
  const uint64_t *mat_off;
  const uint32_t *cand_count;
  void *keys, *tmp;
  uint32_t n_pairs, key64;
  uint32_t n_passes;          /* stable counting-sort passes, least significant first        */
  uint8_t shift[8], bits[8];  /* digit of pass k = (key >> shift[k]) & ((1 << bits[k]) - 1)  */
};

struct SaEnumParams {
  const uint8_t *arena;
  const uint64_t *off_a;
  const uint32_t *len_a;
  const uint64_t *off_b;
  const uint32_t *len_b;
  const uint64_t *mat_off;
  const int32_t *M, *A, *B;
  const uint16_t *code;
  const int32_t *table;
  const void *keys;              /* SORTED candidate keys, pair p at element mat_off[p]  */
  const uint32_t *cand_count;    /* [n]                                            */
  const uint32_t *cand_box;      /* [4n] (window kernel)                            */
  const int32_t *min_score;      /* [n]                                            */
  uint32_t *mask;                /* visited bits in HBM (lane kernel), zeroed by the caller */
  const uint64_t *mask_off;      /* [n] in 32-bit words                            */
  const uint64_t *str_off;       /* [n] slot of max_hits*(len_a+len_b) chars        */
  char *out_a, *out_b;
  SaDevHit *hits;                /* [n * max_hits]                                  */
  uint32_t *hit_count, *str_used, *enum_status;
  uint32_t n_pairs, K, max_hits;
  int32_t open1, ext, gen_eq, gen_ne;
  uint32_t flags;
  uint32_t max_mask_words;       /* largest per-pair bitmap, 32-bit words            */
  SaKeyLayout layout;
  uint32_t only_flagged;         /* generic kernels: run only pairs flagged SA_ENUM_FALLBACK / SA_ENUM_GENERIC */
  uint32_t window_bytes;         /* window kernel: LDS bytes for the direction window */
  uint32_t claim_bits;           /* window kernel: log2 of the claim slots            */
  uint32_t best_step;            /* largest score one move can add (sizes the window's margin) */
  uint8_t *dir;                  /* direction bytes of every pair's window (sw_direction_kernel -> window kernel),
                                    pair p at byte dir_offset(mat_off[p], p), sa_sw_enum_window.hip                               */
  unsigned long long *trace;     /* optional [16n]: load cycles, walk cycles, iterations, 1, 5 phase totals (SEQALIGN_ENUM_TRACE) */
  const uint32_t *pair_list;     /* window kernels: the pairs of this launch (n_list of them); NULL = all n_pairs */
  uint32_t n_list;
  uint32_t threads;              /* window kernel: threads per workgroup of this launch (256 / 512 / 1024)      */
  uint32_t retry;                /* window kernels: second attempt at flagged pairs, margin as large as LDS allows */
  uint32_t max_len_a, max_len_b; /* of the chunk (sizes the direction kernel's tiles)                              */
  uint32_t inline_steps;         /* window kernel: cells a thread walks itself before queueing (0 = default)        */
};

/* LDS configurations of the window kernel, smallest first: a pair goes to the first one whose window holds what
 * it needs; several workgroups per CU for the small ones (sa_sw_enum_window.hip) */
struct SaEnumClass {
  uint32_t threads, claim_bits, window_bytes;
};
int sa_enum_classes(uint32_t key64, SaEnumClass out[4]);
/* enum_status values besides SEQALIGN_E_*: */
#define SA_ENUM_STOPPED_AT_MAX 0x80000000u   /* top bit: stopped at max_hits with candidates left */
#define SA_ENUM_FALLBACK 0x40000000u         /* window kernel: a walk left the window -- retry with the largest one */
#define SA_ENUM_GENERIC 0x20000000u          /* window kernels cannot take this pair: the generic kernel does    */

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
  const uint64_t *start_index; /* SW: end cell of the hit per pair; NULL = NW        */
  uint32_t *out_pos;           /* SW: [4*n] pos_a, pos_b, len_a, len_b               */
  uint32_t n_pairs, K;
  int32_t open1, ext, gen_eq, gen_ne;
  uint32_t flags;
};

/* substitution lookup flavour */
enum { SA_SUBST_SIMPLE = 0, SA_SUBST_LDS = 1, SA_SUBST_GLOBAL = 2 };
#define SA_LDS_TABLE_MAX_K 64

/* returns hipSuccess or the launch error; never synchronises */
hipError_t sa_launch_fill_wavefront(const SaFillParams &p, uint32_t max_len_a,
                                    hipStream_t stream);
hipError_t sa_launch_fill_rowscan(const SaFillParams &p, uint32_t max_len_a,
                                  hipStream_t stream);
/* LDS-ring stream writer; only when sa_stream_kernel_applicable() */
bool sa_stream_kernel_applicable(const SaFillParams &p, uint32_t max_len_a);
hipError_t sa_launch_fill_stream(const SaFillParams &p, uint32_t max_len_a,
                                 hipStream_t stream);
/* whether sa_launch_fill_stream would also emit the candidate keys (p.cand_*) */
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
hipError_t sa_launch_sw_reduce(const SaReduceParams &p, hipStream_t stream);
/* candidate keys from match_scores already in HBM (fills that cannot emit them themselves): one pass over M */
hipError_t sa_launch_sw_emit(const SaReduceParams &p, const SaCandKeys &c, hipStream_t stream);
/* per-pair stable LSD radix sort of the candidate keys (sa_sort.hip); the result is in `keys` after an even
 * number of passes, else in `tmp` */
void sa_sort_plan(const SaKeyLayout &l, SaSortParams *p);
hipError_t sa_launch_sort_keys(const SaSortParams &p, hipStream_t stream);
/* SW multi-hit enumeration: the LDS-window kernel (sa_sw_enum_window.hip) for every pair, flagging the pairs it
 * cannot take; the generic kernels (sa_sw_enum.hip) for flagged pairs (only_flagged) or for all */
size_t sa_enum_window_lds_limit();
/* bytes of SaEnumParams::dir for a chunk of n pairs and `cells` matrix cells */
static inline size_t sa_dir_bytes(uint64_t cells, uint64_t n) { return (size_t)(2 * cells + 8 * n + 16); }
/* one class of pairs: direction bytes (needs matrices + boxes, not the sorted keys), then the enumeration */
hipError_t sa_launch_sw_direction(const SaEnumParams &p, hipStream_t stream);
hipError_t sa_launch_sw_enumerate_window(const SaEnumParams &p, hipStream_t stream);
hipError_t sa_launch_sw_enumerate(const SaEnumParams &p, hipStream_t stream);
/* every pair's strings (and, hits_out != NULL, its hit records) packed back to back: dst_off / hit_dst = prefixes */
hipError_t sa_launch_gather_strings(const char *src_a, const char *src_b, const uint64_t *str_off,
                                    const uint32_t *used, const uint64_t *dst_off, char *dst_a, char *dst_b,
                                    const SaDevHit *hits_in, const uint32_t *hit_count, const uint64_t *hit_dst,
                                    SaDevHit *hits_out, uint32_t max_hits, uint32_t n_pairs, hipStream_t stream);
hipError_t sa_launch_nw_traceback(const SaTraceParams &p, hipStream_t stream);
/* three device allocations of `bytes`, spread over HBM and checked (sa_placement.hip);
 * *quality (may be NULL): 3-stream / 1-stream write bandwidth ratio of the result, < 0 if not probed */
hipError_t sa_alloc_arenas_spread(size_t bytes, void *out[3], hipStream_t stream, float *quality);
/* DPP self-test: out[l] = value shifted in from lane l-1 (lane 0 gets `fill`) */
hipError_t sa_launch_dpp_probe(int32_t *out64, int32_t fill, hipStream_t stream);

#endif
