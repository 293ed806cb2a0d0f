/*
 * seqalign_hip.h -- C ABI of the MI355X (gfx950) DP-fill engine.
 *
 * This is the drop-in boundary for the ONE hot path of noporpoise/seq-align:
 * the O(n*m) affine-gap fill of the match / gap_a / gap_b score matrices
 * (reference: static alignment_fill_matrices, src/alignment.c:28-168, reached
 * only through aligner_align, src/alignment.c:170-193).  Plain pointers and
 * sizes only; no C++ or torch types.  Built by seq-align_amd/Makefile into
 * seq-align_amd/lib/libseqalign_hip.so together with the host-side mirror of the
 * reference API (alignment.h, alignment_scoring.h, needleman_wunsch.h,
 * smith_waterman.h in this directory).
 *
 * The reference aligns one pair per call (callback per pair,
 * src/alignment_cmdline.c:611-622); a 150x150 fill is ~23 k cells, far below one
 * kernel launch, so the new surface here is a BATCH of independent pairs.
 * aligner_align() itself is kept (alignment.h) and is a batch of one.
 *
 * There is NO CPU fallback: every entry point that needs the device fails with
 * SEQALIGN_E_NO_DEVICE / SEQALIGN_E_HIP when the HIP runtime or a gfx950 device
 * is missing.
 *
 * Output contract (bit-exact with the reference): for pair p the three matrices
 * hold (len_a+1)*(len_b+1) int32 each, dense pitch W=len_a+1, cell (i,j) at
 * j*W+i (ARR_2D_INDEX, src/alignment_macros.h:11), starting mat_off[p] cells
 * into each of the three arenas.
 */
#ifndef SEQALIGN_HIP_H
#define SEQALIGN_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "seqalign_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------- */
enum {
  SEQALIGN_OK = 0,
  SEQALIGN_E_NO_DEVICE = 1,   /* no HIP runtime / no gfx950 device            */
  SEQALIGN_E_HIP = 2,         /* a HIP call failed (see seqalign_last_error)  */
  SEQALIGN_E_ARG = 3,         /* bad argument                                 */
  SEQALIGN_E_NOMEM = 4,
  SEQALIGN_E_UNKNOWN_PAIR = 5,/* character pair without a score and
                                 use_match_mismatch==0; the reference exit()s
                                 here (src/alignment_scoring.c:178-181)       */
  SEQALIGN_E_DOMAIN = 6,      /* NW scoring whose gap penalties are below
                                 -|min_penalty|: signed overflow (UB) in the
                                 reference fill                */
  SEQALIGN_E_TRACEBACK = 7,   /* src/alignment.c:328-349 would exit()         */
  SEQALIGN_E_TOO_LARGE = 8    /* one pair needs >= 2^31 cells                 */
};

const char *seqalign_strerror(int code);
/* Text of the last HIP failure on the calling thread ("" if none). */
const char *seqalign_last_error(void);

/* ---- device context --------------------------------------------------------- */
typedef struct seqalign_ctx seqalign_ctx_t;

/* Number of visible gfx950 devices (0 when there is no GPU / no runtime). */
int seqalign_device_count(void);

/* One context per device (one per process in the multi-GPU layout: the batch is
 * sharded by pair index, no collective -- SURVEY 8e). */
int seqalign_ctx_create(int device, seqalign_ctx_t **out);
void seqalign_ctx_destroy(seqalign_ctx_t *ctx);
int seqalign_ctx_device(const seqalign_ctx_t *ctx);

/* ---- scoring, flattened for the device ------------------------------------- */
/* scoring_lookup (src/alignment_scoring.c:133-182) is a pure function of
 * (scoring, a, b); it is flattened ONCE per scoring into a 256-entry
 * char -> (folded char, class) map plus a dense class x class int32 table and
 * uploaded.  The handle stays valid until released or the context dies. */
typedef struct seqalign_dev_scoring seqalign_dev_scoring_t;

int seqalign_scoring_upload(seqalign_ctx_t *ctx, const scoring_t *scoring,
                            int is_sw, seqalign_dev_scoring_t **out);
void seqalign_scoring_release(seqalign_ctx_t *ctx, seqalign_dev_scoring_t *h);

/* ---- batch descriptors -------------------------------------------------------- */
/* Host-side batch: one byte arena + per-pair offsets/lengths (raw chars, not
 * NUL-dependent, exactly what aligner_align takes per pair). */
typedef struct {
  uint64_t n_pairs;
  const char *arena;       /* all sequence bytes                               */
  uint64_t arena_bytes;
  const uint64_t *off_a;   /* [n_pairs] byte offset of seq_a in arena          */
  const uint32_t *len_a;   /* [n_pairs]                                        */
  const uint64_t *off_b;
  const uint32_t *len_b;
} seqalign_batch_t;

/* Device-resident batch + outputs: every pointer is a DEVICE pointer. */
typedef struct {
  uint64_t n_pairs;
  const uint8_t *arena;
  const uint64_t *off_a;
  const uint32_t *len_a;
  const uint64_t *off_b;
  const uint32_t *len_b;
  const uint64_t *mat_off; /* [n_pairs] first cell of pair p in each arena      */
  int32_t *match_scores;   /* three arenas, same per-pair offsets               */
  int32_t *gap_a_scores;
  int32_t *gap_b_scores;
  uint64_t *status;        /* [n_pairs] ~0 = ok, else row-major index of the
                              first cell whose character pair has no score     */
  uint32_t max_len_a;      /* max over the batch (selects columns-per-lane)     */
  uint32_t max_len_b;
} seqalign_dev_batch_t;

/* Which fill kernel family to launch. */
enum {
  SEQALIGN_KERNEL_AUTO = 0,
  SEQALIGN_KERNEL_WAVEFRONT = 1, /* anti-diagonal wavefront, one wave per pair
                                    (north_star's literal schedule); 4-5x slower
                                    than the row sweeps: each lane stores to a
                                    different row.  Kept as an independently
                                    scheduled cross-check; never picked by AUTO.
                                    (Until round 3 every gap_extend > 0 call was
                                    routed here; the row sweeps now take the
                                    trend of their gap_b scan from the right end
                                    for a positive extension, sa_rowsweep.hpp) */
  SEQALIGN_KERNEL_ROWSCAN = 2,   /* row sweep + max-plus prefix scan for gap_b,
                                    rows stored straight from registers        */
  SEQALIGN_KERNEL_STREAM = 3,    /* same sweep, output through an LDS ring as
                                    aligned 1 KiB blocks (len_a <= 1023 and the
                                    three arenas congruent mod 4 KiB; otherwise
                                    the call falls back: AUTO -> STRIPS,
                                    STREAM -> ROWSCAN)                         */
  SEQALIGN_KERNEL_STRIPS = 4,    /* long rows: the 512-column strips of a pair run
                                    as a pipeline of waves, 64 rows apart      */
  SEQALIGN_KERNEL_WGSTREAM = 5   /* rows of 1024..4095 columns: one workgroup per
                                    pair, row split over four waves, shared LDS
                                    ring (fast-path scorings; else ROWSCAN)    */
};

/* THE HOT PATH.  Replaces alignment_fill_matrices (src/alignment.c:28-168) for a
 * whole batch: enqueues the fill on `stream` (a hipStream_t passed as void*,
 * NULL = the context's own stream) and returns without synchronising.
 * WHERE the three output arenas lie decides how fast it runs (it is bound by HBM writes): with arenas from
 * seqalign_arenas_alloc (placed by a measured walk) BASELINE configs[1] runs at 0.82-0.86 of the 8 TB/s peak;
 * with three buffers of the caller's own (hipMalloc, a torch tensor) at 0.65 -- or 0.83 when the allocator happens to hand out
 * memory of different classes: both were seen for the same call on one box (bench.py: roofline.frac_unplaced; round 6:
 * 0.824 / 0.650 / 0.835 in three consecutive processes).  De-phasing the three streams inside the kernel changes nothing
 * (profiles/r06/r06_dephase.txt): it is which memory, not which offsets.  Callers who own the buffers should get them from
 * seqalign_arenas_alloc. */
int seqalign_fill_batch_device(seqalign_ctx_t *ctx,
                               const seqalign_dev_scoring_t *scoring,
                               const seqalign_dev_batch_t *batch, int kernel,
                               void *stream);

/* SW local maxima (replaces the scan of src/smith_waterman.c:152-156 on the
 * device): per pair, the best match_scores cell in reference hit order (score
 * desc, column asc, index asc) and every cell with score >= min_score compacted
 * into cand[cand_off[p] .. +cand_cap[p]) (unsorted; cand_count[p] may exceed the
 * capacity, in which case only the first cand_cap[p] were stored).
 * All pointers are DEVICE pointers. */
typedef struct {
  uint64_t n_pairs;
  const uint32_t *len_a, *len_b;
  const uint64_t *mat_off;
  const int32_t *match_scores;
  int32_t min_score;
  int32_t *best_score;      /* [n_pairs]                                        */
  uint64_t *best_index;     /* [n_pairs] row-major cell index                   */
  uint32_t *cand_count;     /* [n_pairs]                                        */
  const uint64_t *cand_off; /* [n_pairs]                                        */
  const uint32_t *cand_cap; /* [n_pairs]                                        */
  uint32_t *cand_index;     /* compacted cell indices                           */
  int32_t *cand_score;      /* and their scores                                 */
} seqalign_sw_reduce_t;

int seqalign_sw_reduce_device(seqalign_ctx_t *ctx, const seqalign_sw_reduce_t *r,
                              void *stream);

/* Traceback on the device: consumes the matrices a fill left in HBM
 * and applies alignment.c:244-350 per pair, one lane per pair.
 *   NW (seqalign_nw_traceback_device): needleman_wunsch.c:53-132 from the
 *       bottom-right cell to the border.
 *   SW (seqalign_sw_traceback_device): the local alignment ending at
 *       start_index[p] (a match_scores cell, normally best_index from
 *       seqalign_sw_reduce_device), walked until the score is 0 -- what the first
 *       smith_waterman_fetch returns (smith_waterman.c:165-258); out_pos gets
 *       pos_a, pos_b, len_a, len_b (4 per pair).
 * Pair p's two strings are written RIGHT-ALIGNED into
 * out_a/out_b[str_off[p] .. str_off[p]+len_a+len_b): they start at
 * str_off[p]+out_head[p] and are out_len[p] long (no NUL).  status[p] is 0 or a
 * SEQALIGN_E_* code.  All pointers are DEVICE pointers. */
typedef struct {
  const uint64_t *str_off;
  char *out_a, *out_b;
  uint32_t *out_head, *out_len;
  int32_t *out_score;
  uint32_t *status;
  const uint64_t *start_index; /* SW only */
  uint32_t *out_pos;           /* SW only */
} seqalign_trace_t;

int seqalign_nw_traceback_device(seqalign_ctx_t *ctx,
                                 const seqalign_dev_scoring_t *scoring,
                                 const seqalign_dev_batch_t *batch,
                                 const seqalign_trace_t *trace, void *stream);
int seqalign_sw_traceback_device(seqalign_ctx_t *ctx,
                                 const seqalign_dev_scoring_t *scoring,
                                 const seqalign_dev_batch_t *batch,
                                 const seqalign_trace_t *trace, void *stream);

/* ---- host-level convenience (H2D -> fill -> D2H) ---------------------------- */
/* Fills every pair of a HOST batch and copies the matrices back into the three
 * host arenas (cell offsets mat_off[p], host pointer).  Streams the batch in
 * chunks that fit the context's device buffers.  status may be NULL. */
int seqalign_fill_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch,
                        const scoring_t *scoring, int is_sw,
                        const uint64_t *mat_off, int32_t *match_scores,
                        int32_t *gap_a_scores, int32_t *gap_b_scores,
                        uint64_t *status);

/* Global NW over a host batch: GPU fill + traceback (the reference's consumer,
 * src/needleman_wunsch.c:53-145, fresh code).  By default the traceback also runs
 * on the device and only the strings cross PCIe; the option traceback=host copies
 * the matrices back and walks them on the host (north_star's literal split --
 * identical results, PCIe-bound).  For plain scorings (no free / forbidden gaps, no
 * sentinel scores, gap_open <= 0, gap_extend <= 0) and rows up to 1 024 columns (round 5; over 768: from 384 pairs) the
 * device path does not write the matrices at all: the fill leaves one byte of
 * directions per cell -- the answers to alignment_reverse_move's equality tests,
 * src/alignment.c:311-327, taken while the operands are in registers -- and the
 * walk follows them (option nw_dirs=0: three matrices everywhere).
 * Large chunks run as a pipeline of sub-batches (upload / kernels / download on
 * three streams; option subbatches).  Results:
 * score[p], and the two alignment strings of pair p written NUL-terminated at
 * out_a + str_off[p], out_b + str_off[p] (capacity len_a+len_b+1 each). */
int seqalign_nw_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch,
                      const scoring_t *scoring, const uint64_t *str_off,
                      char *out_a, char *out_b, uint32_t *out_len,
                      int32_t *out_score);

/* Local SW over a host batch (src/smith_waterman.c:165-277 semantics, fresh
 * visited mask per pair).  Emits, per pair, successive hits with
 * score >= min_score[p], at most max_hits per pair, into the caller's hit array
 * (hit_cap entries total).
 *   max_hits == 1: GPU fill + GPU reduction + GPU traceback of the best hit.
 *   max_hits >= 2: GPU fill that also reports where the cells >= min_score are (count,
 *       bounding box, columns per row) -> one reverse sweep over each pair's matrices:
 *       a cell is marked by the lowest-ranked walk (reference order: score desc, column
 *       asc, index asc) that arrives at it, so the winners -- and with them every hit of
 *       the pair, in order -- follow from one pass over the rows, bottom to top, without
 *       the sequential procedure of smith_waterman.c:165-277 being run
 *       -> one GPU traceback per wanted hit.  Any max_hits; only the strings cross PCIe.
 *       Plain scorings, rows up to 1 024 columns (513 and up: batches of >= 128 pairs, 769 and up: >= 640; keys
 *       of <= 62 bits): the fill writes match_scores + one byte of directions per cell
 *       instead of the three matrices, the sweep and the walks read those (option
 *       sweep_dirs=0: three matrices everywhere).
 *   option traceback=host: the matrices and the compacted candidates of every
 *       pair are copied back and the hits are enumerated on the host (threaded
 *       over pairs). */
typedef struct {
  uint64_t pair;
  int32_t score;
  uint32_t pos_a, pos_b, len_a, len_b; /* smith_waterman.c:249-255 */
  uint32_t length;                     /* alignment columns */
  uint64_t str_off;                    /* into out_a / out_b */
} seqalign_sw_hit_t;

/* Errors and what has been delivered when they are returned:
 *   SEQALIGN_E_NOMEM   hit_cap or str_cap is too small: the hits that FIT have been delivered first -- a prefix of the
 *                      result in pair order, *n_hits counts them -- then the call stops and reports it;
 *   a pair's own error (SEQALIGN_E_UNKNOWN_PAIR: a character pair without a score, SEQALIGN_E_TRACEBACK): the lowest failing
 *                      pair's code is returned, it takes precedence over E_NOMEM within its chunk, and nothing of that chunk
 *                      is delivered (earlier chunks' hits stay, *n_hits says how many).
 * The packed best-hit path (max_hits == 1, direction bytes) has no separate fill status to fetch: a direction fill is only
 * admitted for scorings in which every character pair has a score, and each walk carries its pair's status home in its own
 * word (sa_batch_sw.hip). */
int seqalign_sw_batch(seqalign_ctx_t *ctx, const seqalign_batch_t *batch,
                      const scoring_t *scoring, const int32_t *min_score,
                      uint32_t max_hits, seqalign_sw_hit_t *hits,
                      uint64_t hit_cap, uint64_t *n_hits, char *out_a,
                      char *out_b, uint64_t str_cap);

/* ---- asynchronous host-level calls ----------------------------------------------- */
/* A host-level call on one batch is a serial chain (pack, upload, fill, walk, results home, expansion into the caller's strings)
 * of which the kernels are about half; ACROSS batches the chain overlaps: batch k + 1's packing and upload need nothing of batch
 * k's walk and expansion.  The reference has one pair in flight (src/alignment_cmdline.c:611-622); a caller streaming batches
 * submits them and waits for them in order:
 *     seqalign_nw_batch_submit(ctx, &batch[k], sc, ..., &job[k]);   ...   rc = seqalign_job_wait(job[k - 2]);
 * Up to `async_lanes` (option, 1..8, default 3) submitted calls of a context are in flight at once, each on a lane with streams,
 * pinned staging and device scratch of its own (so `async_lanes` times a synchronous call's scratch memory); jobs START in
 * submission order and may finish in any.  A job IS the synchronous call -- same arguments, same results bit for bit -- run with a
 * snapshot of the context's options taken at submit.  Everything passed (the batch's arrays, the scoring, the output buffers) is
 * borrowed until seqalign_job_wait returns.  Submitting is thread-safe; any thread may wait for any job, once.
 *   seqalign_job_wait   blocks until the job has run; returns its SEQALIGN_* code (its text: seqalign_last_error in the waiting
 *                       thread; what it launched: seqalign_ctx_last_call_info of the context) and frees the ticket;
 *   seqalign_job_done   1 when the job has finished (wait will not block), else 0; does not free.
 * seqalign_ctx_destroy runs every submitted job to its end first; tickets never waited for are leaked, not dangling. */
typedef struct seqalign_job seqalign_job_t;
int seqalign_nw_batch_submit(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                             const uint64_t *str_off, char *out_a, char *out_b, uint32_t *out_len, int32_t *out_score,
                             seqalign_job_t **job);
int seqalign_sw_batch_submit(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                             const int32_t *min_score, uint32_t max_hits, seqalign_sw_hit_t *hits, uint64_t hit_cap,
                             uint64_t *n_hits, char *out_a, char *out_b, uint64_t str_cap, seqalign_job_t **job);
int seqalign_job_wait(seqalign_job_t *job);
int seqalign_job_done(seqalign_job_t *job);

/* ---- several GPUs from one process -------------------------------------------- */
/* The same three calls over n_ctx contexts (normally one per GPU of the node): the
 * pairs are split into n_ctx contiguous index ranges of (nearly) equal DP cells --
 * sum of (len_a+1)*(len_b+1), so ragged batches stay balanced -- context g works on range g
 * from its own host thread, nothing is exchanged between devices (the
 * path shards by pair, no collective).  Results are exactly those of the
 * single-context call.  hits/strings of the SW call come back in pair order; each
 * range gets a share of hit_cap / str_cap proportional to its pairs
 * (SEQALIGN_E_NOMEM if a range overflows its share). */
int seqalign_fill_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx,
                              const seqalign_batch_t *batch, const scoring_t *scoring,
                              int is_sw, const uint64_t *mat_off, int32_t *match_scores,
                              int32_t *gap_a_scores, int32_t *gap_b_scores, uint64_t *status);
int seqalign_nw_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx,
                            const seqalign_batch_t *batch, const scoring_t *scoring,
                            const uint64_t *str_off, char *out_a, char *out_b,
                            uint32_t *out_len, int32_t *out_score);
int seqalign_sw_batch_multi(seqalign_ctx_t *const *ctxs, int n_ctx,
                            const seqalign_batch_t *batch, const scoring_t *scoring,
                            const int32_t *min_score, uint32_t max_hits,
                            seqalign_sw_hit_t *hits, uint64_t hit_cap, uint64_t *n_hits,
                            char *out_a, char *out_b, uint64_t str_cap);

/* ---- arena placement ----------------------------------------------------------- */
/* Three device buffers of bytes_each for seqalign_dev_batch_t's match_scores /
 * gap_a_scores / gap_b_scores (the reference's three malloc'd matrices,
 * src/alignment.c:183-190, for a whole batch), PLACED in HBM: on MI355X the fill's
 * write pattern (thousands of concurrent sequential streams into three arenas) runs
 * ~25 % slower when all three arenas lie in one "class" of physical memory than when
 * one of them lies in another; three hipMallocs in a row usually
 * land in one class.  Any 4 KiB-aligned device memory is accepted by the fill -- this
 * is only the fast way to get it.  The context's own scratch (host-level entry
 * points) is allocated the same way.
 * How: the arenas are built with the virtual-memory API (hipMemCreate / hipMemMap)
 * from uniform 512 MiB physical chunks.  M and A take the first chunks; further
 * chunks are created (not mapped, not touched: ~10 us each) and every 4 GiB a window
 * of them is mapped as a candidate third arena and timed with the fill's own store
 * pattern against one linear stream; if no candidate reaches the target a second walk
 * moves the SECOND arena the same way (three arenas in three classes are the best there
 * is: the headline kernel at the speed of a memset) (ratio ~0.75: the three disturb each other,
 * 0.98-1.02: one arena in another class, 1.04-1.05: the best there is; the probe repeats to
 * +-2 %).  The first candidate at or above the option arena_quality (default 1.045) ends the walk, otherwise the best one is kept; all other chunks go
 * straight back.  Bounded by the option arena_scan_gib (default 160; 0 = three plain
 * hipMallocs, what the command-line tools use) and by 60 % of the free device
 * memory (re-checked while the walk goes on; a context's own scratch: 25 %, and 16 GiB after its first walk; walks are
 * serialised per device, process-wide); typically 10-60 GiB are held for ~0.1 s.  Arenas under 256 MiB are
 * allocated plainly.  *quality, if not NULL, receives the probe ratio of the result
 * (< 0: not probed); seqalign_arenas_info tells how it was found.  Arenas are meant
 * to be kept and reused.  Free with seqalign_arenas_free (NOT hipFree). */
int seqalign_arenas_alloc(seqalign_ctx_t *ctx, uint64_t bytes_each, void *arenas[3], float *quality);
int seqalign_arenas_free(seqalign_ctx_t *ctx, void *arenas[3]);

#define SEQALIGN_ARENA_MAX_TRIES 96
typedef struct {
  float quality;        /* probe ratio of the arenas handed out (< 0: not probed)          */
  float target;         /* the ratio that would have ended the walk (option arena_quality)  */
  int32_t vmm;          /* 1: built from hipMemCreate chunks, 0: three hipMallocs            */
  uint32_t chunk_mib;   /* chunk size                                                        */
  float depth_gib;      /* how far behind the second arena, in allocation order, the third   */
  float scanned_gib;    /* device memory held at the end of the walks (transient)            */
  float depth_a_gib;    /* where the second walk moved the SECOND arena to (< 0: it stayed    */
                        /* right behind the first)                                           */
  uint32_t tries;       /* candidates timed, both walks                                      */
  uint32_t second_walk_from; /* try_*[second_walk_from ..] belong to the second walk         */
  float try_quality[SEQALIGN_ARENA_MAX_TRIES];    /* their ratios, in order                  */
  float try_depth_gib[SEQALIGN_ARENA_MAX_TRIES];
  float kept_gib;       /* what the process's chunk pool of this device holds after the walk  */
  float seconds;        /* wall clock of seqalign_arenas_alloc: chunk creation, mappings, every timed candidate, releases */
} seqalign_arena_info_t;
/* How the arenas returned by seqalign_arenas_alloc were placed (arenas[0] identifies them). */
int seqalign_arenas_info(seqalign_ctx_t *ctx, void *const arenas[3], seqalign_arena_info_t *info);

/* The chunk pool.  A placement walk creates up to arena_scan_gib of 512 MiB chunks and uses three arenas' worth; the driver
 * clears released VRAM before it hands it out again, so a process whose walk had just given back 160 GiB waited seconds at
 * its next large allocation (round 4: the first seqalign_nw_batch of BASELINE configs[4]'s share 3.7 s).  Up to
 * arena_keep_gib (option, default 16) of the unused chunks therefore stay with the process, per device, and the contexts'
 * large scratch buffers (direction bytes, staging, unplaced arena sets) are mapped from them; chunks of freed buffers and
 * arenas return to the pool.  The pool is emptied when the device's last context is destroyed.
 * seqalign_pool_trim releases everything beyond keep_bytes now (and lowers the cap to it; the next walk raises it again);
 * keep_bytes = UINT64_MAX only queries.  *held_bytes, if not NULL, receives what the pool holds afterwards. */
int seqalign_pool_trim(seqalign_ctx_t *ctx, uint64_t keep_bytes, uint64_t *held_bytes);

/* ---- context options ----------------------------------------------------------- */
/* Everything that steers a context's choices is an option of THAT context.  The defaults are read once, in
 * seqalign_ctx_create, from the environment (SEQALIGN_<KEY>, upper case); afterwards only this call changes
 * them -- nothing below seqalign_ctx_create reads the environment.  Keys and values:
 *   kernel          auto | wavefront | rowscan | stream | strips | wgstream   what SEQALIGN_KERNEL_AUTO means
 *   traceback       device | host           where seqalign_nw_batch / seqalign_sw_batch walk the matrices
 *   trace_kernel    auto | lane | wave      the device walker
 *   sweep_mode      auto | pair | strips    multi-hit SW: one wave per pair / per strip
 *   sweep_strip     0 | 64 | 128 | 256      columns per strip       sweep_cpl  0 | 1 | 2 | 4
 *   chunk_bytes     0 | >= 1 MiB            device memory one host-level chunk may use
 *   async_lanes     0 (= 3) | 1 .. 8        submitted calls in flight per context (seqalign_*_batch_submit)
 *   walk_group      0 | 1 | 4 | 8           walks per wave of the device walker on LDS tiles (4 / 8: in lockstep, vector state); 0: four on
 *                                           direction bytes laid out in blocks (NW, best hit; rows <= 512 columns), else one wave per walk
 *   dirs_local      1 | 0                   seqalign_nw_batch / seqalign_sw_batch(max_hits = 1) on direction bytes: the byte holds the cell's own comparisons and the
 *                                           walker resolves the state it arrives in (cheaper to write; 0: the older form everywhere)
 *   walk_tile       0 | 32 | 64             that walker's tile edge in bytes (0: 32 for global walks -- fewer lines per reload --, 64 for best-hit walks)
 *   walk_stage      1 | 0                   that walker sends a wave's moves home as one run of whole lines out of LDS (0: two pieces per walk)
 *   subbatches      0 (by size) .. 256      sub-batches a chunk of seqalign_nw_batch is pipelined in (1 = off)
 *   nw_dirs, sweep_dirs  1 | 0              direction bytes instead of the three matrices where they apply (above)
 *   pack16          1 | 0 | 2               direction-byte fills take two pairs per wave in packed int16 where scores fit int16 --
 *                                           pairs of EQUAL shape share a wave: a chunk of one shape, or a ragged chunk whose pairs
 *                                           are paired up by shape on the host (the others: one per wave in the same launch, NW;
 *                                           a wave to themselves, SW): for chunks of > 1 024 pairs (ragged: >= 2 048) | never | whatever the size
 *   quad            0 | 1 | 2               the NW and the SW best-hit fill with FOUR pairs per wave (32 lanes a couple of pairs; chunks of
 *                                           one shape, rows up to 192 columns): for chunks of >= 4 096 (NW) / 16 384 (SW) pairs | never |
 *                                           whatever the size
 *   walk_overlap    0 | 1                   seqalign_nw_batch (direction bytes): walks on their own stream beside the next fills (round 4:
 *                                           off -- beside the four-pairs-per-wave fills the walks cost more than they hide)
 *   nw_moves        1 | 0                   the walks on direction bytes send home two bits per alignment column (which string has
 *                                           a gap there) and the host expands them against the caller's sequences, instead of the
 *                                           gapped strings (seqalign_nw_batch; seqalign_sw_batch: also ONE launch + wait for all
 *                                           hits of a chunk, the walks launched before the counts are seen); DESIGN.md 3.5d, 3.6c
 *   zero_copy       auto | 0..3             that path's kernels read the packed sequences (1) / write the moves (2) in pinned host
 *                                           memory in place; auto: moves in place when the walks run one wave each
 *   sweep_ev        1 | 0                   the direction-byte sweep carries a walk as one word key << 2 | state
 *   arena_scan_gib  0 .. 1024               arena_quality  0.5 .. 1.5    (seqalign_arenas_alloc)
 *   arena_keep_gib  0 .. 1024               (the chunk pool, seqalign_pool_trim)
 *   arena_free_pct  10 .. 90                share of the memory free at its start that a walk of seqalign_arenas_alloc may hold (60)
 *   upload_slices   0 .. 16                 seqalign_nw_batch: slices a sub-batch's sequences are packed and uploaded in (0 = 1: measured,
 *                                           more slices cost what they overlap)
 *   cpl, wpb, lds_pad, reduce_depth, sweep_trace, timing   tuning experiments / development aids
 * Numbers are integers and nothing else ("abc", "1x", "" are refused, not read as 0); switches take 1 / 0, true / false, on / off,
 * yes / no.  Returns SEQALIGN_E_ARG for an unknown key or a value outside the key's range (nothing changes then). */
int seqalign_ctx_set_option(seqalign_ctx_t *ctx, const char *key, const char *value);

/* The value in force (as text: what seqalign_ctx_set_option would take) -- for callers that change an option for a
 * while and put it back.  Returns SEQALIGN_E_ARG for an unknown key or when cap is too small. */
int seqalign_ctx_get_option(const seqalign_ctx_t *ctx, const char *key, char *value, size_t cap);

/* ---- what the last call launched ------------------------------------------------ */
/* Which kernels the context's last call (any entry point above that launches) put on the device, and for how many
 * pairs / walks each: the fills of src/alignment.c:28-168 come in several forms (three matrices, direction bytes, two
 * pairs per wave in packed int16 ...) chosen per chunk from the scoring and the shapes, and results are identical
 * whichever runs -- so a test (or a user wondering where the time goes) asks HERE which one did.  launches[k] /
 * items[k] are indexed by SEQALIGN_K_*; a call that chunks or pipelines its batch adds up all its launches. */
enum {
  SEQALIGN_K_FILL_WAVEFRONT = 0,   /* three matrices, sa_fill_wavefront.hip                                        */
  SEQALIGN_K_FILL_ROWSCAN,         /* ... sa_fill_rowscan.hip                                                      */
  SEQALIGN_K_FILL_STREAM,          /* ... sa_fill_stream.hip (the headline kernel)                                 */
  SEQALIGN_K_FILL_STRIPS,          /* ... sa_fill_strips.hip                                                       */
  SEQALIGN_K_FILL_WGSTREAM,        /* ... sa_fill_wgstream.hip                                                     */
  SEQALIGN_K_FILL_NW_DIRS,         /* NW, direction bytes only, one pair per wave (items: pairs)                   */
  SEQALIGN_K_FILL_NW_DIRS_X2,      /* NW, direction bytes only, two pairs per wave in packed int16                 */
  SEQALIGN_K_FILL_SW_DIRS,         /* SW, match_scores + direction bytes, one pair per wave                        */
  SEQALIGN_K_FILL_SW_DIRS_X2,      /* SW, match_scores + direction bytes, two pairs per wave                       */
  SEQALIGN_K_FILL_SW_BEST_X2,      /* SW best hit: direction bytes + the best cell, two pairs per wave             */
  SEQALIGN_K_SW_REDUCE,            /* sa_reduce.hip: the separate max-reduction over match_scores                  */
  SEQALIGN_K_SW_BOX,               /* sa_reduce.hip: candidates' box / rows from match_scores in HBM               */
  SEQALIGN_K_SWEEP_REGS,           /* multi-hit sweep on three matrices, rows in registers                         */
  SEQALIGN_K_SWEEP_LDS,            /* ... winners of two rows in LDS (wide pairs)                                  */
  SEQALIGN_K_SWEEP_STRIPS,         /* ... one wave per strip (few wide pairs)                                      */
  SEQALIGN_K_SWEEP_DIRS,           /* multi-hit sweep on match_scores + direction bytes, one pair per wave         */
  SEQALIGN_K_SWEEP_DIRS_X2,        /* (reserved: two pairs per wave measured slower, never launched)               */
  SEQALIGN_K_WALK_LANE,            /* traceback on three matrices, one lane per walk (items: walks)                */
  SEQALIGN_K_WALK_WAVE,            /* ... one wave per walk, LDS tiles                                             */
  SEQALIGN_K_WALK_DIRS_LANE,       /* traceback on direction bytes, strings out, one lane per walk                 */
  SEQALIGN_K_WALK_DIRS_TILE,       /* ... one wave per walk, LDS tiles                                             */
  SEQALIGN_K_WALK_MOVES_LANE,      /* traceback on direction bytes, two bits per column out (host/sa_moves.c)      */
  SEQALIGN_K_WALK_MOVES_TILE,      /* ... one wave per walk                                                        */
  SEQALIGN_K_FILL_NW_DIRS_X4,      /* NW, direction bytes only, FOUR pairs per wave (32 lanes a couple of pairs)   */
  SEQALIGN_K_FILL_SW_BEST_X4,      /* SW best hit: direction bytes + the best cell, four pairs per wave            */
  SEQALIGN_K_COUNT
};
#define SEQALIGN_K_MAX 32
typedef struct {
  uint32_t launches[SEQALIGN_K_MAX];
  uint64_t items[SEQALIGN_K_MAX];
} seqalign_call_info_t;
int seqalign_ctx_last_call_info(const seqalign_ctx_t *ctx, seqalign_call_info_t *out);
/* "fill_stream", "fill_nw_dirs_x2", ... ; NULL for a kind that does not exist */
const char *seqalign_kernel_kind_name(int kind);

/* ---- CIGAR -------------------------------------------------------------------- */
/* The reference has no CIGAR output (its result is the pair of gapped strings, src/alignment.h:33-40); this
 * is the derived format: run-length encoding of the alignment's columns with seq_a as the query and seq_b as
 * the reference -- both letters: M (extended = 0) or '=' / 'X' (extended != 0; letters compared as they are,
 * or case-folded when case_insensitive), '-' in result_b: I (insertion to the reference), '-' in result_a: D.
 * Writes a NUL-terminated string; returns its length, or (size_t)-1 when cap is too small or a column has a
 * gap in both strings.  Host only, no device involved. */
size_t seqalign_cigar(const char *result_a, const char *result_b, size_t length, int extended,
                      int case_insensitive, char *out, size_t cap);

/* The batch calls with CIGAR as their output (north_star: "identical CIGAR/alignment strings").  Same alignments, same order, same
 * query convention as seqalign_cigar above -- seq_a is the QUERY, seq_b the REFERENCE: a column with '-' in result_b is I, with '-'
 * in result_a is D.  format: SEQALIGN_CIGAR_M (M / I / D) or SEQALIGN_CIGAR_EQX (= / X / I / D; letters compared case-folded unless
 * scoring->case_sensitive).  On plain scorings the device walks come home as two bits per alignment column and the
 * host run-length encodes those bits: the gapped strings are never written, and a read's "150M" is 5 bytes where its strings are 302.
 *   seqalign_nw_batch_cigar: pair p's CIGAR NUL-terminated at cigar + cigar_off[p], capacity cigar_off[p + 1] - cigar_off[p] (cigar_off
 *     has n_pairs + 1 entries; SEQALIGN_E_NOMEM when a pair's does not fit: no CIGAR is longer than 2 (len_a + len_b) + 1 bytes),
 *     cigar_len[p] = its strlen, out_score[p] as seqalign_nw_batch.  A global alignment's CIGAR covers both sequences completely
 *     (leading / trailing gaps are I / D runs, as the reference's strings have them, src/needleman_wunsch.c:117-132).
 *   seqalign_sw_batch_cigar: as seqalign_sw_batch; hit h's CIGAR NUL-terminated at cigar + hits[h].str_off, hits back to back;
 *     hits[h].length stays the number of alignment columns, pos_a / pos_b / len_a / len_b say where the hit lies. */
#define SEQALIGN_CIGAR_M 1
#define SEQALIGN_CIGAR_EQX 2
int seqalign_nw_batch_cigar(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring, int format,
                            const uint64_t *cigar_off, char *cigar, uint32_t *cigar_len, int32_t *out_score);
int seqalign_sw_batch_cigar(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                            const int32_t *min_score, uint32_t max_hits, int format, seqalign_sw_hit_t *hits,
                            uint64_t hit_cap, uint64_t *n_hits, char *cigar, uint64_t cigar_cap);
int seqalign_nw_batch_cigar_submit(seqalign_ctx_t *ctx, const seqalign_batch_t *batch, const scoring_t *scoring, int format,
                                   const uint64_t *cigar_off, char *cigar, uint32_t *cigar_len, int32_t *out_score,
                                   seqalign_job_t **job);   /* (asynchronous: seqalign_job_wait, above) */
/* ... and over several contexts (GPUs), as seqalign_nw_batch_multi / seqalign_sw_batch_multi */
int seqalign_nw_batch_cigar_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                  int format, const uint64_t *cigar_off, char *cigar, uint32_t *cigar_len, int32_t *out_score);
int seqalign_sw_batch_cigar_multi(seqalign_ctx_t *const *ctxs, int n_ctx, const seqalign_batch_t *batch, const scoring_t *scoring,
                                  const int32_t *min_score, uint32_t max_hits, int format, seqalign_sw_hit_t *hits,
                                  uint64_t hit_cap, uint64_t *n_hits, char *cigar, uint64_t cigar_cap);

/* ---- diagnostics ---------------------------------------------------------------- */
/* The host legs of seqalign_nw_batch's direction-byte path alone, no device involved: sizes, offsets and packing of the
 * sequences (pack_ms), then the expansion of synthetic all-MATCH moves into the caller's strings (expand_ms); averages
 * over `iterations`.  What one rank's CPU share sustains when N ranks do this at once (seq-align_amd/tools/host_scale.py). */
int seqalign_host_legs_nw(const seqalign_batch_t *batch, const uint64_t *str_off, char *out_a, char *out_b,
                          uint32_t *out_len, int iterations, double *pack_ms, double *expand_ms);

/* ---- misc ---------------------------------------------------------------------- */
/* The context's own stream (a hipStream_t as void*): what NULL means wherever a `stream` is passed.  For callers that
 * keep data in HBM and order their own work (events, copies) with the library's launches -- and a reason not to create
 * another stream for that: the runtime multiplexes the streams of one priority over four hardware queues, and every
 * further queue in use makes the host-level calls' own pipelines slower. */
void *seqalign_ctx_stream(seqalign_ctx_t *ctx);

/* Event pair on a stream for kernel timing (HIP events; bench.py). */
int seqalign_time_fill_ms(seqalign_ctx_t *ctx,
                          const seqalign_dev_scoring_t *scoring,
                          const seqalign_dev_batch_t *batch, int kernel,
                          void *stream, int repeats, float *ms_each);

#ifdef __cplusplus
}
#endif
#endif /* SEQALIGN_HIP_H */
