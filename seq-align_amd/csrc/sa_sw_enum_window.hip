// sa_sw_enum_window.hip -- Smith-Waterman multi-hit enumeration, one WORKGROUP per pair, everything the
// procedure touches staged in LDS.
//
// Reference semantics (src/smith_waterman.c:137-277): candidates (cells with match_scores >= min_score) are
// visited in (score desc, column asc, index asc) order; a candidate that is already marked is skipped, else it
// is walked back (alignment_reverse_move, alignment.c:244-350) to score 0 marking every cell, and the walk is
// abandoned -- its marks stay -- when it meets a marked cell; a completed walk is a hit.  A 150x1000 pair has
// ~15 000 candidates >= --minscore, a 300x300 BLOSUM62 pair ~50 000, nearly all of them in the plume of the
// one real hit: ~half are already marked when their turn comes, the others walk 1-3 cells into a marked one.
// The procedure is sequential by definition; what follows makes it parallel without changing its result.
//
// 1. The route of a walk does not depend on the marks -- only where it stops does.  So the predecessor of every
//    state is worked out ONCE, for all cells of a window around the candidates, by all threads in parallel with
//    coalesced loads (the decision code is the traceback's own, sa_trace_common.hpp): one byte per cell in LDS,
//    2 bits per state (predecessor matrix, or 3 = "this state's score is 0": the walk ends here) + 1 visited
//    bit.  After that a walk step is ONE LDS byte read; no match_scores / gap scores / sequence / table access.
// 2. Rounds of T candidates in order, one per thread (rank = thread index).  Repeat until the round is done:
//      claim   every unfinished thread walks from where it stands until a marked cell / the end of the walk,
//              and atomicMin's its rank into a claim table slot (hash of the cell) for every cell it passes;
//      commit  it walks the same cells again and marks them for as long as the claim slot still holds ITS rank;
//              at the first cell claimed by a lower rank it stops and resumes there in the next iteration.
//    Why this is the sequential result: a thread's claimed path is a superset of its true path (it can only be
//    cut short by marks of lower ranks, and those are only ever committed when true, by induction), so cells no
//    lower rank claims are cells no lower rank will ever mark -- the sequential walk would find them unmarked
//    and mark them: committing them now is exact.  Marks of a higher rank never land on a lower rank's path
//    (the lower rank holds the claim).  The lowest unfinished rank always wins all its claims, so every
//    iteration makes progress; a walk is a hit when it has committed its way to a score-0 state.  A hash
//    collision only delays a commit by an iteration.  (Measured on C3 / C4 pairs: ~5 iterations per round of
//    1 024; checked against the sequential procedure by the tests and tools/fuzz_e2e.py.)
// 3. The hits of a round are numbered and their strings written by their own threads (block prefix sums give
//    hit ordinal and string offset in rank order), cut at max_hits.
//
// The window is the candidates' bounding box (reported by the fill) extended up/left as far as LDS allows.  A
// walk that leaves it, a pair whose box does not fit, or a traceback error: the pair is flagged
// SA_ENUM_FALLBACK and taken by the generic kernel (sa_sw_enum.hip) -- exact, only slower.
#include <algorithm>

#include "sa_trace_common.hpp"

namespace sa {

constexpr uint32_t kVis = 0x40u;            // visited bit of a window byte
constexpr size_t kWindowLdsLimit = 160u * 1024u - 2048u;   // dynamic LDS per workgroup (CDNA4: 160 KiB per CU)

template <int T>
struct BlockScan {
  uint32_t *wave_tot;   // [T/64] in LDS
  // exclusive prefix sum of v over the workgroup in thread order; *total = sum.  Two barriers.
  __device__ __forceinline__ uint32_t excl(uint32_t v, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int u = 0; u < T / 64; ++u) {
      const uint32_t t = wave_tot[u];
      if (u < w) base += t;
      tot += t;
    }
    *total = tot;
    return base + inc - v;
  }
};

template <int T, typename KeyT>
__global__ void __launch_bounds__(T) sw_enumerate_window_kernel(const SaEnumParams p) {
  extern __shared__ uint32_t lds32[];
  __shared__ uint32_t s_flag;              // != 0: hand the pair to the generic kernel
  __shared__ uint32_t s_last;              // candidate index of the hit that reached max_hits
  __shared__ uint32_t s_wave_tot[T / 64];
  const uint32_t pair = blockIdx.x;
  const uint32_t tid = threadIdx.x;

  const uint32_t n_cand = p.cand_count[pair];
  if (n_cand == 0) {
    if (tid == 0) { p.hit_count[pair] = 0; p.str_used[pair] = 0; p.enum_status[pair] = 0; }
    return;
  }
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const uint32_t *box = p.cand_box + 4ull * pair;
  const uint32_t rmin = box[0], rmax = box[1], cmin = box[2], cmax = box[3];

  // ---- window: the box, extended up / left by the largest margin that fits (wave-uniform arithmetic)
  const uint32_t nclaim = 1u << p.claim_bits;
  uint32_t *claim = lds32;
  uint8_t *win = reinterpret_cast<uint8_t *>(lds32 + nclaim);
  uint32_t *win32 = lds32 + nclaim;
  const uint64_t budget = p.window_bytes;
  auto area = [&](uint32_t m) {
    const uint32_t r0 = rmin > m ? rmin - m : 0, c0 = cmin > m ? cmin - m : 0;
    return (uint64_t)(rmax - r0 + 1) * (cmax - c0 + 1);
  };
  if (area(0) > budget) {
    if (tid == 0) p.enum_status[pair] = SA_ENUM_FALLBACK;
    return;
  }
  uint32_t lo = 0, hi = max(rmin, cmin);
  while (lo < hi) {   // largest margin with area <= budget
    const uint32_t mid = lo + (hi - lo + 1) / 2;
    if (area(mid) <= budget) lo = mid; else hi = mid - 1;
  }
  const uint32_t r0 = rmin > lo ? rmin - lo : 0, c0 = cmin > lo ? cmin - lo : 0;
  const uint32_t Ww = cmax - c0 + 1, Hw = rmax - r0 + 1, wcells = Ww * Hw;

  const PairView v{p.arena + p.off_a[pair], p.arena + p.off_b[pair], p.M + mo, p.A + mo, p.B + mo, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  if (tid == 0) { s_flag = 0; s_last = 0; }
  __syncthreads();

  // ---- 1. predecessor bits of every state in the window (fresh visited bits: SURVEY A.3-2)
  for (uint32_t wy = tid / Ww, wx = tid % Ww, idx = tid; idx < wcells; idx += T) {
    const uint32_t x = c0 + wx, y = r0 + wy;
    uint32_t byte = 0x3f;                      // border cells: every SW border score is 0
    if (x > 0 && y > 0) {
      const uint32_t at = y * W + x;
      const int s[3] = {v.M[at], v.A[at], v.B[at]};
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (s[m] > 0) {
          uint32_t px = x, py = y;
          int pm = m, ps = s[m];
          if (reverse_move(v, k, px, py, pm, ps)) atomicOr(&s_flag, 1u);   // the generic kernel reports the error
          byte = (byte & ~(3u << (2 * m))) | ((uint32_t)pm << (2 * m));
        }
      }
    }
    win[idx] = (uint8_t)byte;
    wx += T % Ww; wy += T / Ww;               // idx += T without a division
    if (wx >= Ww) { wx -= Ww; ++wy; }
  }
  __syncthreads();
  if (s_flag) {
    if (tid == 0) p.enum_status[pair] = SA_ENUM_FALLBACK;
    return;
  }

  // ---- 2. rounds of T candidates
  const KeyT *keys = static_cast<const KeyT *>(p.keys) + mo;
  const int min_score = p.min_score[pair];
  const uint32_t cshift = p.layout.row_bits, sshift = p.layout.row_bits + p.layout.col_bits;
  const uint32_t rmask = (1u << p.layout.row_bits) - 1u, cmask = (1u << p.layout.col_bits) - 1u;
  char *oa = p.out_a + p.str_off[pair];
  char *ob = p.out_b + p.str_off[pair];
  SaDevHit *hits = p.hits + (uint64_t)pair * p.max_hits;
  const uint32_t hshift = 32u - p.claim_bits;
  BlockScan<T> scan{s_wave_tot};

  uint32_t emitted = 0, used = 0;             // the same in every thread
  bool exhausted = true;
  for (uint32_t base = 0; base < n_cand; base += T) {
    const uint32_t idx = base + tid;
    bool active = idx < n_cand;
    const KeyT key = active ? keys[idx] : (KeyT)0;
    const int cscore = p.layout.cap - (int)(uint32_t)(key >> sshift);
    const uint32_t col = (uint32_t)(key >> cshift) & cmask, row = (uint32_t)key & rmask;
    active = active && cscore >= min_score;
    if (__syncthreads_count(active) == 0) break;      // sorted: nothing later qualifies

    // where this thread's walk stands: window coordinates + matrix; cells committed so far
    uint32_t wx = col - c0, wy = row - r0, at = wy * Ww + wx, done_len = 0;
    int m = MAT_MATCH;
    bool is_hit = false;
    for (;;) {
      for (uint32_t i = tid; i < nclaim; i += T) claim[i] = 0xffffffffu;
      __syncthreads();
      // claim
      uint32_t plen = 0;
      bool to_end = false;                    // the claimed path ends in a score-0 state (else: in front of a marked cell)
      if (active) {
        uint32_t cx = wx, cy = wy, cat = at, byte = win[cat];
        int cm = m;
        if (byte & kVis) {
          active = false;                     // already marked (smith_waterman.c:269), or the walk ran into a mark
        } else {
          for (;;) {
            atomicMin(&claim[(cat * 2654435761u) >> hshift], tid);
            ++plen;
            const uint32_t d = (byte >> (2 * cm)) & 3u;
            if (d == 3u) { to_end = true; break; }
            if (cm != MAT_GAP_A) { if (cx == 0) { atomicOr(&s_flag, 1u); break; } --cx; }
            if (cm != MAT_GAP_B) { if (cy == 0) { atomicOr(&s_flag, 1u); break; } --cy; }
            cat = cy * Ww + cx;
            cm = (int)d;
            byte = win[cat];
            if (byte & kVis) break;
          }
        }
      }
      __syncthreads();
      // commit the prefix nobody of lower rank claims
      if (active) {
        uint32_t t = 0;
        while (claim[(at * 2654435761u) >> hshift] == tid) {
          atomicOr(&win32[at >> 2], kVis << (8u * (at & 3u)));
          ++t;
          if (t == plen) break;
          const uint32_t d = (win[at] >> (2 * m)) & 3u;
          if (m != MAT_GAP_A) --wx;
          if (m != MAT_GAP_B) --wy;
          at = wy * Ww + wx;
          m = (int)d;
        }
        done_len += t;
        if (t == plen) {                      // walked to the end of the claimed path: this candidate is done
          active = false;
          is_hit = to_end;
        }
      }
      const int left = __syncthreads_count(active);
      if (s_flag) {                           // a walk left the window
        if (tid == 0) p.enum_status[pair] = SA_ENUM_FALLBACK;
        return;
      }
      if (left == 0) break;
    }

    // ---- 3. this round's hits, in rank order (smith_waterman.c:217-255)
    if (__syncthreads_count(is_hit)) {
      const uint32_t steps = is_hit ? done_len - 1 : 0;   // cells marked = moves + 1
      uint32_t n_hits, n_chars;
      const uint32_t ord = scan.excl(is_hit ? 1u : 0u, &n_hits);
      const bool take = is_hit && emitted + ord < p.max_hits;
      const uint32_t soff = scan.excl(take ? steps : 0u, &n_chars);
      if (take) {
        // replay from the candidate cell, writing the columns right to left; (wx, wy) is the score-0 cell
        uint32_t x = col, y = row, a2 = (row - r0) * Ww + (col - c0);
        int hm = MAT_MATCH;
        for (uint32_t w = steps; w-- > 0;) {
          oa[used + soff + w] = (hm == MAT_GAP_A) ? '-' : (char)v.seq_a[x - 1];
          ob[used + soff + w] = (hm == MAT_GAP_B) ? '-' : (char)v.seq_b[y - 1];
          const uint32_t d = (win[a2] >> (2 * hm)) & 3u;
          if (hm != MAT_GAP_A) { --x; a2 -= 1; }
          if (hm != MAT_GAP_B) { --y; a2 -= Ww; }
          hm = (int)d;
        }
        SaDevHit h;
        h.score = cscore; h.pos_a = x; h.pos_b = y; h.len_a = col - x; h.len_b = row - y;
        h.length = steps; h.str_off = used + soff;
        hits[emitted + ord] = h;
        if (emitted + ord + 1 == p.max_hits) s_last = idx;
      }
      const uint32_t taken = min(n_hits, p.max_hits - emitted);
      emitted += taken;
      used += n_chars;
      __syncthreads();
      if (emitted >= p.max_hits) {
        exhausted = (s_last + 1 >= n_cand);
        break;
      }
    }
  }
  if (tid == 0) {
    p.hit_count[pair] = emitted;
    p.str_used[pair] = used;
    p.enum_status[pair] = exhausted ? 0u : SA_ENUM_STOPPED_AT_MAX;
  }
}

template <int T>
static hipError_t launch_window(const SaEnumParams &p, size_t lds, hipStream_t stream) {
  hipError_t e;
  if (p.layout.key64) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_enumerate_window_kernel<T, unsigned long long>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sw_enumerate_window_kernel<T, unsigned long long>), dim3(p.n_pairs), dim3(T), lds, stream, p);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_enumerate_window_kernel<T, uint32_t>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sw_enumerate_window_kernel<T, uint32_t>), dim3(p.n_pairs), dim3(T), lds, stream, p);
  }
  return hipGetLastError();
}

}  // namespace sa

size_t sa_enum_window_lds_limit() { return sa::kWindowLdsLimit; }

// p.window_bytes: what the largest pair of the batch would like (the caller's estimate, capped here);
// threads per workgroup follow the window: small windows leave room for several workgroups per CU
hipError_t sa_launch_sw_enumerate_window(const SaEnumParams &p_in, hipStream_t stream) {
  if (p_in.n_pairs == 0) return hipSuccess;
  SaEnumParams p = p_in;
  if (const char *env = getenv("SEQALIGN_ENUM_WINDOW_BYTES")) {   // tests: a window too small for the walks -> fallback path
    const long v = atol(env);
    if (v >= 16) p.window_bytes = (uint32_t)std::min<long>(v, (long)p.window_bytes);
  }
  int threads = p.window_bytes <= 24u * 1024u ? 256 : p.window_bytes <= 64u * 1024u ? 512 : 1024;
  if (const char *env = getenv("SEQALIGN_ENUM_THREADS")) {   // tuning experiments
    const int t = atoi(env);
    if (t == 256 || t == 512 || t == 1024) threads = t;
  }
  p.claim_bits = threads == 256 ? 11 : threads == 512 ? 12 : 13;
  if (const char *env = getenv("SEQALIGN_ENUM_CLAIM_BITS")) {
    const int b = atoi(env);
    if (b >= 8 && b <= 14) p.claim_bits = (uint32_t)b;
  }
  const size_t claims = (size_t)4 << p.claim_bits;
  size_t window = (p.window_bytes + 15u) & ~(size_t)15u;
  if (claims + window > sa::kWindowLdsLimit) window = (sa::kWindowLdsLimit - claims) & ~(size_t)15u;
  p.window_bytes = (uint32_t)window;
  const size_t lds = claims + window;
  if (threads == 256) return sa::launch_window<256>(p, lds, stream);
  if (threads == 512) return sa::launch_window<512>(p, lds, stream);
  return sa::launch_window<1024>(p, lds, stream);
}
