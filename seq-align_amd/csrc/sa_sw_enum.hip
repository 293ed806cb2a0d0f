// sa_sw_enum.hip -- Smith-Waterman multi-hit enumeration on the device
// (SURVEY 8f-2).
//
// Reference semantics (src/smith_waterman.c:137-277): every cell with
// match_scores > 0 is a candidate; candidates are visited in the order
// (score desc, column asc [, index asc]); a candidate that is already marked is
// skipped, otherwise it is walked back to score 0 marking every cell on the way,
// and the walk is abandoned (its marks stay) as soon as it meets a marked cell; a
// completed walk is a hit.  The CLI stops at the first hit below min_score and
// after max_hits (sw_cmdline.c:214-217).
//
// Here: the fill emits the cells >= min_score as sort keys (sa_fill_stream.hpp), sa_sort.hip
// orders each pair's keys, and the enumeration proper runs in sa_sw_enum_window.hip (one
// workgroup per pair, everything in LDS).  THIS file holds the GENERIC kernels, which walk
// the matrices in HBM and make no assumption about where a walk goes: they take the pairs
// the window kernel flags (a walk left its window, the candidates' box does not fit LDS, a
// traceback error to report) or, with SEQALIGN_SW_ENUM=wave|lane, every pair: one WAVE per
// pair with the visited bitmap in LDS and 64 speculative walks per round
// (sw_enumerate_wave_kernel), or -- bitmap over 64 KiB -- one LANE per pair with the
// bitmap in HBM (sw_enumerate_kernel, the literal procedure).  Hits' strings are written
// left-aligned into the pair's slot.
#include "sa_trace_common.hpp"

namespace sa {

// a sorted candidate key -> score and cell (SaFillParams: key layout)
struct KeyReader {
  const void *base;
  uint32_t key64, cshift, sshift, rmask, cmask, W;
  int cap;
  __device__ __forceinline__ KeyReader(const SaEnumParams &p, uint64_t mo, uint32_t W_)
      : base(p.layout.key64 ? (const void *)(static_cast<const unsigned long long *>(p.keys) + mo)
                            : (const void *)(static_cast<const uint32_t *>(p.keys) + mo)),
        key64(p.layout.key64), cshift(p.layout.row_bits), sshift(p.layout.row_bits + p.layout.col_bits),
        rmask((1u << p.layout.row_bits) - 1u), cmask((1u << p.layout.col_bits) - 1u), W(W_), cap(p.layout.cap) {}
  __device__ __forceinline__ void get(uint32_t i, int &score, uint32_t &cell) const {
    const unsigned long long k = key64 ? static_cast<const unsigned long long *>(base)[i]
                                       : (unsigned long long)static_cast<const uint32_t *>(base)[i];
    score = cap - (int)(uint32_t)(k >> sshift);
    cell = ((uint32_t)k & rmask) * W + ((uint32_t)(k >> cshift) & cmask);
  }
};

__device__ __forceinline__ bool skip_pair(const SaEnumParams &p, uint32_t pair) {
  return p.only_flagged && !(p.enum_status[pair] & (SA_ENUM_FALLBACK | SA_ENUM_GENERIC));
}

__global__ void __launch_bounds__(64) sw_enumerate_kernel(const SaEnumParams p) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= p.n_pairs || skip_pair(p, pair)) return;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const PairView v{p.arena + p.off_a[pair], p.arena + p.off_b[pair], p.M + mo, p.A + mo, p.B + mo, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  uint32_t *seen = p.mask + p.mask_off[pair];
  const KeyReader keys(p, mo, W);
  const uint32_t n_cand = p.cand_count[pair];
  const int min_score = p.min_score[pair];
  char *oa = p.out_a + p.str_off[pair];
  char *ob = p.out_b + p.str_off[pair];
  SaDevHit *hits = p.hits + (uint64_t)pair * p.max_hits;

  uint32_t emitted = 0, used = 0, err = 0, kpos = 0;
  bool exhausted = true;
  for (; kpos < n_cand; ++kpos) {
    if (emitted >= p.max_hits) { exhausted = false; break; }
    int cscore;
    uint32_t end;
    keys.get(kpos, cscore, end);
    if (cscore < min_score) break;                     // sorted: nothing later qualifies
    if ((seen[end >> 5] >> (end & 31)) & 1u) continue; // smith_waterman.c:269

    // pass 1 (:187-199): walk to score 0, marking; abandon on a marked cell
    uint32_t x = end % W, y = end / W, steps = 0;
    int matrix = MAT_MATCH, score = cscore;
    bool clash = false;
    for (;; ++steps) {
      const uint32_t at = y * W + x;
      const uint32_t word = seen[at >> 5], bit = 1u << (at & 31);
      if (word & bit) { clash = true; break; }
      seen[at >> 5] = word | bit;
      if (score == 0) break;
      if ((err = reverse_move(v, k, x, y, matrix, score))) break;
    }
    if (err) break;
    if (clash) continue;

    // pass 2 (:217-244): replay, writing the columns right to left
    x = end % W; y = end / W; matrix = MAT_MATCH; score = cscore;
    for (uint32_t w = steps; score > 0;) {
      --w;
      oa[used + w] = (matrix == MAT_GAP_A) ? '-' : (char)v.seq_a[x - 1];
      ob[used + w] = (matrix == MAT_GAP_B) ? '-' : (char)v.seq_b[y - 1];
      if ((err = reverse_move(v, k, x, y, matrix, score))) break;
    }
    if (err) break;
    SaDevHit h;                                         // smith_waterman.c:249-255
    h.score = cscore; h.pos_a = x; h.pos_b = y;
    h.len_a = end % W - x; h.len_b = end / W - y; h.length = steps; h.str_off = used;
    hits[emitted++] = h;
    used += steps;
  }
  p.hit_count[pair] = emitted;
  p.str_used[pair] = used;
  p.enum_status[pair] = err ? err : (exhausted ? 0u : SA_ENUM_STOPPED_AT_MAX);
}

// ---------------------------------------------------------------------------
// One WAVE per pair, visited bitmap in LDS, 64 candidates at a time.
//
// Numbers that shaped it (C3, 150x1000, min_score 60; oracle statistics): a pair has
// ~15 000 candidates, ~8 000 of them already marked when their turn comes, ~7 000
// walks of which all but ~1 end after 1-2 steps on a marked cell.  With one lane per
// pair that is ~30 000 dependent HBM round trips per pair on 157 waves (141 ms).
//
// The path a candidate would walk does not depend on the marks -- only where it
// stops does.  So per batch of 64 consecutive candidates:
//   speculate  every lane walks ITS candidate against the bitmap as it is now, records
//              the first kPath cells and why it stopped (start marked / met a marked cell /
//              reached score 0 / path longer than kPath).  The loads of the 64 walks
//              overlap.
//   apply      in candidate order (ballot + ctz), wave-uniformly: replay the recorded
//              cells against the CURRENT bitmap -- marks made by earlier candidates of
//              the same batch can only stop a walk earlier, exactly as in the sequential
//              procedure -- marking as the reference does.  A walk that outlives its
//              recording is continued for real (uniform serial walk); a walk that
//              reaches score 0 is a hit and is replayed once more to write its strings
//              (smith_waterman.c:217-244).
// The bitmap (one bit per cell) lives in LDS; pairs whose bitmap exceeds 64 KiB take
// the one-lane kernel above.
constexpr int kPath = 4;
enum { R_NONE = 0, R_SKIP, R_CLASH, R_ZERO, R_CAP, R_ERR };

__device__ __forceinline__ bool seen_bit(const uint32_t *seen, uint32_t at) { return (seen[at >> 5] >> (at & 31)) & 1u; }

__global__ void __launch_bounds__(64) sw_enumerate_wave_kernel(const SaEnumParams p) {
  extern __shared__ uint32_t seen[];
  __shared__ uint32_t claim[256];   // per round: which lane owns a cell (light candidates, see "apply")
  const uint32_t pair = blockIdx.x;
  const int lane = threadIdx.x;
  if (skip_pair(p, pair)) return;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const uint32_t words = (W * (lb + 1) + 31) / 32;
  const PairView v{p.arena + p.off_a[pair], p.arena + p.off_b[pair], p.M + mo, p.A + mo, p.B + mo, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  for (uint32_t i = lane; i < words; i += 64) seen[i] = 0;   // fresh mask per call
  __syncthreads();

  const KeyReader keys(p, mo, W);
  const uint32_t n_cand = p.cand_count[pair];
  const int min_score = p.min_score[pair];
  char *oa = p.out_a + p.str_off[pair];
  char *ob = p.out_b + p.str_off[pair];
  SaDevHit *hits = p.hits + (uint64_t)pair * p.max_hits;

  uint32_t emitted = 0, used = 0, err = 0;   // wave-uniform
  bool exhausted = true, done = false;
  for (uint32_t base = 0; base < n_cand && !done; base += 64) {
    const uint32_t idx = base + lane;
    bool valid = idx < n_cand;
    int cscore = INT32_MIN;
    uint32_t cell = 0u;
    if (valid) keys.get(idx, cscore, cell);
    valid = valid && cscore >= min_score;              // sorted: the valid lanes are a prefix
    const uint32_t nvalid = __popcll(__ballot(valid));
    if (nvalid == 0) break;                            // nothing later qualifies

    // ---- speculate
    uint32_t path[kPath];
    int plen = 0, reason = R_NONE, matrix = MAT_MATCH, score = cscore;
    uint32_t x = cell % W, y = cell / W, lerr = 0;
    if (valid) {
      if (seen_bit(seen, cell)) reason = R_SKIP;       // smith_waterman.c:269
#pragma unroll
      for (int s = 0; s < kPath; ++s) {
        if (reason == R_NONE) {   // (lanes that have stopped idle through the remaining steps)
          const uint32_t at = y * W + x;
          if (s > 0 && seen_bit(seen, at)) {
            reason = R_CLASH;
          } else {
            path[s] = at; plen = s + 1;
            if (score == 0) reason = R_ZERO;
            else if ((lerr = reverse_move(v, k, x, y, matrix, score))) reason = R_ERR;
          }
        }
      }
      if (reason == R_NONE) reason = R_CAP;            // kPath cells recorded, kPath moves made
    }
    __syncthreads();

    // ---- apply, in candidate order
    // Two thirds of the candidates that get this far are "light": the walk recorded one or two cells
    // and then met a marked one.  All such a candidate does is test-and-set its first cell and, if that
    // was free, set its second (the cell after it was and stays marked).  Light candidates whose cells
    // are pairwise different commute, so a run of them is applied at once with LDS atomics; overlaps are
    // found with a 256-slot claim table (the lowest lane keeps a slot; anyone else on that slot -- a true
    // overlap or a hash collision -- is taken one by one instead, which is always exact).  Everything
    // else (longer recorded walks, walks that reached 0 or ran out of recording) goes one by one, each
    // after the light ones in front of it.
    const bool light = valid && reason == R_CLASH && plen <= 2;   // (longer recordings: no gain, more collisions)
    bool bulk = light;
    {
#pragma unroll
      for (int t = 0; t < 4; ++t) claim[lane * 4 + t] = 0xffffffffu;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const uint32_t h0 = (path[0] * 2654435761u) >> 24, h1 = (path[plen == 2 ? 1 : 0] * 2654435761u) >> 24;
      if (light) { atomicMin(&claim[h0], (uint32_t)lane); atomicMin(&claim[h1], (uint32_t)lane); }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (light) bulk = claim[h0] == (uint32_t)lane && claim[h1] == (uint32_t)lane;
    }
    const unsigned long long others = __ballot(valid && reason != R_SKIP && !bulk);
    uint32_t cur = 0;
    for (;;) {
      const unsigned long long rest = cur < 64 ? (others >> cur) << cur : 0ull;
      const int i = rest ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest)) : 64;
      if (bulk && (uint32_t)lane >= cur && lane < i) {
        const uint32_t b0 = 1u << (path[0] & 31);
        const uint32_t was = atomicOr(&seen[path[0] >> 5], b0);        // smith_waterman.c:269 if it was set
        if (!(was & b0) && plen == 2) atomicOr(&seen[path[1] >> 5], 1u << (path[1] & 31));
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (i >= 64) break;
      cur = i + 1;
      const int r_i = __builtin_amdgcn_readlane(reason, i), plen_i = __builtin_amdgcn_readlane(plen, i);
      bool clash = false;   // a marked cell stops the replay; at s = 0 that is the "already marked" skip (:269)
#pragma unroll
      for (int s = 0; s < kPath; ++s) {
        if (s < plen_i && !clash) {
          const uint32_t at = __builtin_amdgcn_readlane(path[s], i);
          const uint32_t word = seen[at >> 5], bit = 1u << (at & 31);   // one read per cell: test and set
          if (word & bit) clash = true;
          else seen[at >> 5] = word | bit;               // every lane writes the same word: same value
        }
      }
      if (clash || r_i == R_CLASH) continue;
      if (r_i == R_ERR) { err = __builtin_amdgcn_readlane(lerr, i); done = true; break; }

      // the walk is still alive: score 0 reached (R_ZERO) or recording ran out (R_CAP)
      const int end_score = __builtin_amdgcn_readlane(cscore, i);
      const uint32_t end = __builtin_amdgcn_readlane(cell, i);
      uint32_t steps = plen_i - 1;
      bool zero = (r_i == R_ZERO);
      if (r_i == R_CAP) {                              // continue for real, wave-uniformly
        uint32_t cx = __builtin_amdgcn_readlane(x, i), cy = __builtin_amdgcn_readlane(y, i);
        int cm = __builtin_amdgcn_readlane(matrix, i), cs = __builtin_amdgcn_readlane(score, i);
        for (steps = kPath;; ++steps) {
          const uint32_t at = cy * W + cx;
          if (seen_bit(seen, at)) { clash = true; break; }
          seen[at >> 5] |= 1u << (at & 31);
          if (cs == 0) { zero = true; break; }
          if ((err = reverse_move(v, k, cx, cy, cm, cs))) break;
        }
        if (err) { done = true; break; }
        if (clash) continue;
      }
      if (!zero) continue;

      // a hit (smith_waterman.c:217-255): replay, writing the columns right to left
      uint32_t hx = end % W, hy = end / W;
      int hm = MAT_MATCH, hs = end_score;
      for (uint32_t w = steps; hs > 0;) {
        --w;
        if (lane == 0) {
          oa[used + w] = (hm == MAT_GAP_A) ? '-' : (char)v.seq_a[hx - 1];
          ob[used + w] = (hm == MAT_GAP_B) ? '-' : (char)v.seq_b[hy - 1];
        }
        if ((err = reverse_move(v, k, hx, hy, hm, hs))) break;
      }
      if (err) { done = true; break; }
      if (lane == 0) {
        SaDevHit h;
        h.score = end_score; h.pos_a = hx; h.pos_b = hy;
        h.len_a = end % W - hx; h.len_b = end / W - hy; h.length = steps; h.str_off = used;
        hits[emitted] = h;
      }
      ++emitted;
      used += steps;
      if (emitted >= p.max_hits) {
        exhausted = (base + (uint32_t)i + 1 >= n_cand);
        done = true;
        break;
      }
    }
    __syncthreads();
  }
  if (lane == 0) {
    p.hit_count[pair] = emitted;
    p.str_used[pair] = used;
    p.enum_status[pair] = err ? err : (exhausted ? 0u : SA_ENUM_STOPPED_AT_MAX);
  }
}

// strings and hit records of all pairs packed back to back for one D2H each: one wave per pair
__global__ void __launch_bounds__(256) gather_strings_kernel(const char *src_a, const char *src_b,
                                                             const uint64_t *str_off, const uint32_t *used,
                                                             const uint64_t *dst_off, char *dst_a, char *dst_b,
                                                             const SaDevHit *hits_in, const uint32_t *hit_count,
                                                             const uint64_t *hit_dst, SaDevHit *hits_out,
                                                             uint32_t max_hits, uint32_t n_pairs) {
  const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x & 63;
  const uint32_t n = used[pair];
  const char *sa_ = src_a + str_off[pair], *sb_ = src_b + str_off[pair];
  char *da = dst_a + dst_off[pair], *db = dst_b + dst_off[pair];
  for (uint32_t i = lane; i < n; i += 64) { da[i] = sa_[i]; db[i] = sb_[i]; }
  if (hits_out) {
    const uint32_t nh = hit_count[pair];
    for (uint32_t i = lane; i < nh; i += 64) hits_out[hit_dst[pair] + i] = hits_in[(uint64_t)pair * max_hits + i];
  }
}

}  // namespace sa

hipError_t sa_launch_sw_enumerate(const SaEnumParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  // (staging the sequences' codes in LDS next to the bitmap was tried: slower -- flat loads and one
  // workgroup less per CU -- C3 26.9 -> 36.1 ms)
  const size_t mask = (size_t)p.max_mask_words * 4;
  const char *force = getenv("SEQALIGN_SW_ENUM");   // "lane": the one-lane-per-pair kernel (experiments)
  if (mask <= 65536 && !(force && force[0] == 'l')) {
    hipLaunchKernelGGL(sa::sw_enumerate_wave_kernel, dim3(p.n_pairs), dim3(64), mask, stream, p);
  }
  else {
    hipLaunchKernelGGL(sa::sw_enumerate_kernel, dim3((p.n_pairs + 63) / 64), dim3(64), 0, stream, p);
  }
  return hipGetLastError();
}

hipError_t sa_launch_gather_strings(const char *src_a, const char *src_b, const uint64_t *str_off,
                                    const uint32_t *used, const uint64_t *dst_off, char *dst_a, char *dst_b,
                                    const SaDevHit *hits_in, const uint32_t *hit_count, const uint64_t *hit_dst,
                                    SaDevHit *hits_out, uint32_t max_hits, uint32_t n_pairs, hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(sa::gather_strings_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, stream, src_a, src_b,
                     str_off, used, dst_off, dst_a, dst_b, hits_in, hit_count, hit_dst, hits_out, max_hits, n_pairs);
  return hipGetLastError();
}
