// sa_traceback.hip -- traceback on the device: global NW, and the
// best local SW hit.
//
// The reference re-derives each predecessor from the three score matrices with
// equality tests, priority GAP_A, GAP_B, MATCH (src/alignment.c:244-350), after
// choosing the end matrix at the bottom-right cell (src/needleman_wunsch.c:53-66).
// This kernel does exactly that on the matrices the fill kernel left in HBM, so
// an end-to-end batch returns O(len_a+len_b) characters per pair over PCIe
// instead of 12 B per DP cell (273 KB per 150x150 pair).
//
// One LANE per pair (64 pairs per wave): the walk is a chain of dependent loads
// (~len_a+len_b steps), so the only parallelism is across pairs.  Each lane
// writes its alignment right-to-left into its slot and reports where it starts;
// the host left-aligns while copying out of the staging buffer.
//
// Further down, the walkers of the direction-byte paths (one byte per cell instead of the three matrices): on strings or on
// MOVES (two bits per column, expanded by the host), one lane per walk with the next byte asked for ahead, one wave per walk
// from LDS tiles, four / eight walks per wave in lockstep -- and (round 6) traceback_moves_group_local_kernel for the byte's
// LOCAL form (sa_kernels.h: SA_LD_*): a cell's own comparisons, the state a walk arrives in resolved by byte look-ups.
#include "sa_trace_common.hpp"

namespace sa {

// Where the direction byte of cell (x, y) lies: row-major at pitch W, or -- SaTraceParams::dirs_blocked (sa_kernels.h: the NW and
// best-hit fills of round 6) -- in blocks of 8 rows x 16 columns, nbx = ceil(W / 16) blocks per block row.
__device__ __forceinline__ uint32_t dirs_at(bool blocked, uint32_t x, uint32_t y, uint32_t W, uint32_t nbx) {
  const uint32_t blk = ((y >> 3) * nbx + (x >> 4)) * 128u + ((y & 7u) << 4) + (x & 15u);
  return blocked ? blk : y * W + x;
}
// A walker's 64 x 64 tile of direction bytes into LDS, row-major there whatever the layout in memory.
//   row-major: the 64 rows x 64 bytes that END at (x, y) -- lane r one row; ox / oy = its first column / row;
//   blocked:   the 8 x 4 BLOCKS (8 block rows of 4 blocks = 512 contiguous bytes each: 32 whole lines, all of them used, where
//              the row-major tile's 64 pieces of 64 bytes lie on ~96 lines) whose last block row / block column hold (x, y):
//              ox / oy are multiples of 16 / 8, (x, y) sits at least 48 columns and 56 rows from the tile's first.
// lane l loads 64 bytes: row-major its row; blocked piece p = 4 l + q of the tile's 256 16-byte pieces = block row p / 32, block
// p / 8 % 4, row p % 8 of the block.  Rows / block rows past the pair's last are not loaded (never looked at).
template <int kT>
__device__ __forceinline__ void load_dirs_tile(uint8_t *tile, const uint8_t *__restrict__ Dg, bool blocked, uint32_t x, uint32_t y,
                                               uint32_t W, uint32_t nbx, uint32_t lb, int lane, uint32_t &ox, uint32_t &oy) {
  static_assert(kT == 64, "tiles are 64 x 64");
  typedef uint32_t u4_u __attribute__((ext_vector_type(4), aligned(1)));
  typedef uint32_t u4_a __attribute__((ext_vector_type(4)));
  if (blocked) {
    const uint32_t bx0 = (x >> 4) >= 3u ? (x >> 4) - 3u : 0u, by0 = (y >> 3) >= 7u ? (y >> 3) - 7u : 0u;
    ox = bx0 << 4; oy = by0 << 3;
    const uint32_t nby = (lb + 8u) >> 3;                 // block rows of the pair
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t pc = 4u * (uint32_t)lane + (uint32_t)q, br = pc >> 5, bc = (pc >> 3) & 3u, r = pc & 7u;
      if (by0 + br < nby) {
        // (a block column past the pair's last: the next block row's first blocks, or the buffer's slack -- never looked at)
        const u4_a v = *reinterpret_cast<const u4_a *>(Dg + ((uint64_t)(by0 + br) * nbx + bx0 + bc) * 128u + r * 16u);
        *reinterpret_cast<u4_a *>(tile + (br * 8u + r) * kT + bc * 16u) = v;
      }
    }
  } else {
    ox = x >= (uint32_t)(kT - 1) ? x - (kT - 1) : 0; oy = y >= (uint32_t)(kT - 1) ? y - (kT - 1) : 0;
    const uint32_t r = oy + lane;
    if (r <= lb) {
      // 64 bytes of row r from column ox on; past the row's end that is the next row (or, behind the last pair, the
      // slack every directions buffer has) -- never used: the walk only reads columns <= x
      const uint8_t *src = Dg + (uint64_t)r * W + ox;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<u4_u *>(tile + lane * kT + 16 * q) = *reinterpret_cast<const u4_u *>(src + 16 * q);
    }
  }
}

// Where SW walk `w` starts: the cell start_index[w], or -- walks of the multi-hit path (sa_sw_sweep.hip) -- the cell
// packed in hit number walker_rank[w] of its pair's sorted hit keys.
__device__ __forceinline__ void sw_walk_start(const SaTraceParams &p, uint32_t w, uint32_t pair, uint32_t W, uint32_t &x, uint32_t &y) {
  if (p.hit_keys) {
    const unsigned long long key = p.hit_keys[p.hit_off[pair] + p.len_b[pair] + 1 + p.walker_rank[w]];
    y = (uint32_t)key & ((1u << p.layout.row_bits) - 1u);
    x = (uint32_t)(key >> p.layout.row_bits) & ((1u << p.layout.col_bits) - 1u);
  } else {
    const uint32_t end = (uint32_t)p.start_index[w];
    x = end % W; y = end / W;
  }
}

// where a finished walk leaves head / len / score / status (SaTraceParams::out_meta4)
__device__ __forceinline__ void write_walk_meta(const SaTraceParams &p, uint32_t w, uint32_t pair, uint32_t head, uint32_t len,
                                                int score, uint32_t err) {
  if (p.fill_status && p.fill_status[pair] != ~0ull && !err) err = 5u;   // SEQALIGN_E_UNKNOWN_PAIR
  if (p.out_meta4) {
    *reinterpret_cast<uint4 *>(p.out_meta4 + 4ull * w) = make_uint4(head, len, (uint32_t)score, err);
  } else {
    p.out_score[w] = score;
    p.out_head[w] = head;
    p.out_len[w] = len;
    p.trace_status[w] = err;
  }
}

// SW: start_index != nullptr -> local alignment ending at that match_scores cell
// (smith_waterman.c:165-258 on a fresh mask: the first fetched hit always
// succeeds), walked until the score reaches 0.
template <bool SW>
__global__ void __launch_bounds__(64) traceback_kernel(const SaTraceParams p) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;   // the walk: one per pair, or one per SW hit (walker_pair)
  if (w >= p.n_pairs) return;
  const uint32_t pair = p.walker_pair ? p.walker_pair[w] : w;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair];
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const int32_t *__restrict__ Mg = p.M + mo;
  const int32_t *__restrict__ Ag = p.A + mo;
  const int32_t *__restrict__ Bg = p.B + mo;
  const uint32_t W = la + 1;
  char *oa = p.out_a + p.str_off[w];
  char *ob = p.out_b + p.str_off[w];

  const PairView v{sa_, sb_, Mg, Ag, Bg, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};

  int matrix = MAT_MATCH;
  int score;
  uint32_t x, y, head = la + lb, err = 0;
  if constexpr (SW) {
    sw_walk_start(p, w, pair, W, x, y);
    score = Mg[y * W + x];
  } else {
    // end cell: ties resolve GAP_A > GAP_B > MATCH (needleman_wunsch.c:53-66)
    const uint32_t corner = W * (lb + 1) - 1;
    x = la; y = lb;
    score = Mg[corner];
    { const int b = Bg[corner]; if (b >= score) { matrix = MAT_GAP_B; score = b; } }
    { const int a = Ag[corner]; if (a >= score) { matrix = MAT_GAP_A; score = a; } }
  }
  const int end_score = score;
  const uint32_t end_x = x, end_y = y;

  while (SW ? (score > 0) : (x > 0 && y > 0)) {
    const uint8_t ca = sa_[x - 1], cb = sb_[y - 1];
    --head;
    oa[head] = (matrix == MAT_GAP_A) ? '-' : (char)ca;
    ob[head] = (matrix == MAT_GAP_B) ? '-' : (char)cb;
    if ((err = reverse_move(v, k, x, y, matrix, score))) break;
  }
  if constexpr (!SW) {
    if (!err) {
      for (; y > 0; --y) { --head; oa[head] = '-'; ob[head] = (char)sb_[y - 1]; }   // needleman_wunsch.c:117-123
      for (; x > 0; --x) { --head; oa[head] = (char)sa_[x - 1]; ob[head] = '-'; }   // :126-132
    }
  } else {
    // smith_waterman.c:251-255: start position and consumed lengths
    p.out_pos[4 * w + 0] = x;
    p.out_pos[4 * w + 1] = y;
    p.out_pos[4 * w + 2] = end_x - x;
    p.out_pos[4 * w + 3] = end_y - y;
  }
  (void)end_x; (void)end_y;
  write_walk_meta(p, w, pair, head, la + lb - head, end_score, err);
}

// ---------------------------------------------------------------------------
// SW hits behind sa_fill_dirs.hip: the fill left one byte of directions per cell (where a walk goes from the cell in
// each of its three states, 3 = that state's score is 0), so a walk is a chain of ONE byte load per step instead of
// three ints, and there is nothing to decide.  Same loop as traceback_kernel<true> (smith_waterman.c:187-255):
// while the state's score is positive, emit the column, step back.  One lane per walk.
__global__ void __launch_bounds__(64) traceback_dirs_kernel(const SaTraceParams p) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= p.n_pairs) return;
  const uint32_t pair = p.walker_pair ? p.walker_pair[w] : w;
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const uint8_t *__restrict__ Dg = p.dirs + mo;
  char *oa = p.out_a + p.str_off[w];
  char *ob = p.out_b + p.str_off[w];
  uint32_t x, y, head = la + lb;
  sw_walk_start(p, w, pair, W, x, y);
  const int score = p.start_score ? p.start_score[w] : p.M[mo + (uint64_t)y * W + x];
  const uint32_t end_x = x, end_y = y;
  uint32_t st = MAT_MATCH;
  const bool blk = p.dirs_blocked != 0;
  const uint32_t nbx = (W + 15u) >> 4;
  for (;;) {
    const uint32_t f = ((uint32_t)Dg[dirs_at(blk, x, y, W, nbx)] >> (2u * st)) & 3u;
    if (f == 3u) break;                               // this state's score is 0: the hit starts here
    --head;
    oa[head] = (st == MAT_GAP_A) ? '-' : (char)sa_[x - 1];
    ob[head] = (st == MAT_GAP_B) ? '-' : (char)sb_[y - 1];
    x -= (st != MAT_GAP_A);                           // MATCH: up-left, GAP_A: up, GAP_B: left (alignment.c:274-296)
    y -= (st != MAT_GAP_B);
    st = f;
  }
  p.out_pos[4 * w + 0] = x;
  p.out_pos[4 * w + 1] = y;
  p.out_pos[4 * w + 2] = end_x - x;
  p.out_pos[4 * w + 3] = end_y - y;
  write_walk_meta(p, w, pair, head, la + lb - head, score, 0u);
}

// Needleman-Wunsch behind sa_fill_dirs.hip's directions-only fill (seqalign_nw_batch): the end cell's score and matrix
// come from the fill (nw_score / nw_state), every step is one byte load.  needleman_wunsch.c:53-145.
__global__ void __launch_bounds__(64) traceback_nw_dirs_kernel(const SaTraceParams p) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= p.n_pairs) return;
  const uint32_t la = p.len_a[w], lb = p.len_b[w], W = la + 1;
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[w];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[w];
  const uint8_t *__restrict__ Dg = p.dirs + p.mat_off[w];
  char *oa = p.out_a + p.str_off[w];
  char *ob = p.out_b + p.str_off[w];
  uint32_t x = la, y = lb, head = la + lb, st = (uint32_t)p.nw_state[w];
  const bool blk = p.dirs_blocked != 0;
  const uint32_t nbx = (W + 15u) >> 4;
  while (x > 0 && y > 0) {
    const uint32_t f = ((uint32_t)Dg[dirs_at(blk, x, y, W, nbx)] >> (2u * st)) & 3u;
    --head;
    oa[head] = (st == MAT_GAP_A) ? '-' : (char)sa_[x - 1];
    ob[head] = (st == MAT_GAP_B) ? '-' : (char)sb_[y - 1];
    x -= (st != MAT_GAP_A);
    y -= (st != MAT_GAP_B);
    st = f;
  }
  for (; y > 0; --y) { --head; oa[head] = '-'; ob[head] = (char)sb_[y - 1]; }   // needleman_wunsch.c:117-123
  for (; x > 0; --x) { --head; oa[head] = (char)sa_[x - 1]; ob[head] = '-'; }   // :126-132
  write_walk_meta(p, w, w, head, la + lb - head, p.nw_score[w], 0u);
}

// ---------------------------------------------------------------------------
// The same two walks, one WAVE per walk, from 64 x 64-byte tiles of the directions staged in LDS.  With one lane per
// walk a step is one dependent byte load from HBM (~1 us), and 4 000 or 10 000 walks are only 63 or 157 waves: the chip
// idles while every lane waits (C4: 0.34 ms for 4 001 hits).  Here the wave loads the 64 rows x 64 columns that end at
// the current cell -- lane r one row: 64 contiguous bytes -- plus the sequences' characters of those rows and columns,
// and walks out of LDS until it steps outside (at least 64 steps per tile, a 300-step walk needs 5-9 tiles).  All
// lanes walk redundantly (uniform control flow), lane 0 writes the characters.  Bound by instruction issue (~15 per
// step and wave), so it is the choice while the walks do not fill the chip several times over (the launcher decides).
template <bool NW>
__global__ void __launch_bounds__(64) traceback_dirs_tile_kernel(const SaTraceParams p) {
  constexpr int kT = 64;
  __shared__ __attribute__((aligned(16))) uint8_t tile[kT * kT];
  __shared__ uint8_t ca[kT], cb[kT];   // seq_a[ox + c - 1], seq_b[oy + r - 1]
  const int lane = threadIdx.x;
  const uint32_t w = blockIdx.x;
  const uint32_t pair = NW ? w : (p.walker_pair ? p.walker_pair[w] : w);
  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  const uint8_t *__restrict__ Dg = p.dirs + mo;
  char *oa = p.out_a + p.str_off[w];
  char *ob = p.out_b + p.str_off[w];
  uint32_t x, y, head = la + lb, st;
  int score;
  if constexpr (NW) {
    x = la; y = lb; st = (uint32_t)p.nw_state[w]; score = p.nw_score[w];
  } else {
    sw_walk_start(p, w, pair, W, x, y);
    st = MAT_MATCH; score = p.start_score ? p.start_score[w] : p.M[mo + (uint64_t)y * W + x];
  }
  const uint32_t end_x = x, end_y = y;
  uint32_t ox = 0, oy = 0;
  bool loaded = false;
  for (;;) {
    if constexpr (NW) { if (x == 0 || y == 0) break; }
    if (!loaded || x < ox || y < oy) {   // (wave-uniform) make (x, y) the tile's bottom-right cell (blocked: its last blocks')
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // earlier LDS reads are done with the old tile
      load_dirs_tile<kT>(tile, Dg, p.dirs_blocked != 0, x, y, W, (W + 15u) >> 4, lb, lane, ox, oy);
      { const uint32_t i = ox + lane; ca[lane] = (i >= 1 && i <= la) ? sa_[i - 1] : (uint8_t)0; }
      { const uint32_t j = oy + lane; cb[lane] = (j >= 1 && j <= lb) ? sb_[j - 1] : (uint8_t)0; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_s_waitcnt(0);   // the tile is in LDS before anyone reads it (one wave: program order)
      loaded = true;
    }
    const uint32_t f = ((uint32_t)tile[(y - oy) * kT + (x - ox)] >> (2u * st)) & 3u;
    if constexpr (!NW) { if (f == 3u) break; }   // this state's score is 0: the hit starts here
    --head;
    if (lane == 0) {
      oa[head] = (st == MAT_GAP_A) ? '-' : (char)ca[x - ox];
      ob[head] = (st == MAT_GAP_B) ? '-' : (char)cb[y - oy];
    }
    x -= (st != MAT_GAP_A);
    y -= (st != MAT_GAP_B);
    st = f;
  }
  if (lane != 0) return;
  if constexpr (NW) {
    for (; y > 0; --y) { --head; oa[head] = '-'; ob[head] = (char)sb_[y - 1]; }   // needleman_wunsch.c:117-123
    for (; x > 0; --x) { --head; oa[head] = (char)sa_[x - 1]; ob[head] = '-'; }   // :126-132
  } else {
    p.out_pos[4 * w + 0] = x;
    p.out_pos[4 * w + 1] = y;
    p.out_pos[4 * w + 2] = end_x - x;
    p.out_pos[4 * w + 3] = end_y - y;
  }
  write_walk_meta(p, w, pair, head, la + lb - head, score, 0u);
}

// ---------------------------------------------------------------------------
// The NW walks on direction bytes, sending home MOVES instead of strings (SaTraceParams::moves, host/sa_moves.c): what a
// walk knows is one of three states per column; the two gapped strings of needleman_wunsch.c:82-145 are that plus the
// sequences the host still holds.  So the walk emits one bit per column and plane -- "gap in a" (it stood in GAP_A), "gap
// in b" (GAP_B) -- 32 columns per word, a word's bits set from the top down because the walk runs backwards; the host expands them
// (vpexpandb: 64 columns per instruction).  2 bits per column cross PCIe instead of 16, the walk reads no characters and
// stores two words per 32 steps instead of two bytes per step, and the columns the walk does not visit (the padding of
// needleman_wunsch.c:117-132) are not sent at all: they follow from the lengths.
struct MoveSlot {
  uint32_t *plane_a, *plane_b;   // nw words each
  int nw;
};
// which pair, which start cell and which slot walk `w` has.  Three ways of numbering walks:
//   NW                         : walk w = pair w, from the bottom-right cell (nw_state / nw_score from the fill)
//   SW, walks_per_pair == 0    : walk w = pair w from start_index[w] (the best-hit path), or hit walker_rank[w] of pair
//                                walker_pair[w] (lists made by the host after it has seen the hit counts)
//   SW, walks_per_pair == k    : walk w = hit w % k of pair w / k -- launched BEFORE anybody has seen the hit counts: a walk whose
//                                rank is beyond its pair's hits (or whose pair needs the host: unsorted keys, an error) returns
struct MoveWalk {
  uint32_t pair, x, y, st;
  int score;
  MoveSlot slot;
  bool valid;
};
template <bool NW>
__device__ __forceinline__ MoveWalk move_walk(const SaTraceParams &p, uint32_t w) {
  MoveWalk m{};   // (an invalid walk returns early: the look-ahead kernel reads x / y / st / slot.nw before it looks at `valid`)
  m.valid = true;
  uint32_t rank = 0, wpp = 1;
  if constexpr (NW) {
    m.pair = w;
  } else if (p.walks_per_pair) {
    wpp = p.walks_per_pair;
    m.pair = w / wpp; rank = w % wpp;
    if (rank >= p.hit_count[m.pair] || (p.sweep_status[m.pair] != 0)) { m.valid = false; return m; }
  } else {
    m.pair = p.walker_pair ? p.walker_pair[w] : w;
    if (p.walker_rank) rank = p.walker_rank[w];
  }
  const uint32_t la = p.len_a[m.pair], lb = p.len_b[m.pair], W = la + 1;
  const int nw = (int)((la + lb + 31u) >> 5);
  uint32_t *base;
  if (!NW && p.walks_per_pair) base = p.moves + 2ull * wpp * ((p.str_off[m.pair] >> 5) + m.pair) + 2ull * rank * nw;
  else base = p.moves + 2ull * ((p.str_off[w] >> 5) + w);
  m.slot = MoveSlot{base, base + nw, nw};
  if constexpr (NW) {
    m.x = la; m.y = lb; m.st = (uint32_t)p.nw_state[w]; m.score = p.nw_score[w];
  } else {
    m.st = MAT_MATCH;
    if (p.hit_keys) {
      const unsigned long long key = p.hit_keys[p.hit_off[m.pair] + lb + 1 + rank];
      m.y = (uint32_t)key & ((1u << p.layout.row_bits) - 1u);
      m.x = (uint32_t)(key >> p.layout.row_bits) & ((1u << p.layout.col_bits) - 1u);
      m.score = p.layout.cap - (int)(key >> (p.layout.row_bits + p.layout.col_bits));   // (the key's score field: cap - score)
    } else {
      const uint32_t end = (uint32_t)p.start_index[w];
      m.x = end % W; m.y = end / W;
      m.score = p.start_score[w];
    }
  }
  return m;
}
// NW: score, walked columns (or SA_MOVES_ERR | code) at out_meta2[2w..]; SW: score, walked columns, end cell at out_meta4[4w..]
template <bool NW>
__device__ __forceinline__ void write_moves_meta(const SaTraceParams &p, uint32_t w, const MoveWalk &m, uint32_t end_x, uint32_t end_y,
                                                 uint32_t n_moves) {
  if (p.fill_status && p.fill_status[m.pair] != ~0ull) n_moves = SA_MOVES_ERR | 5u;   // SEQALIGN_E_UNKNOWN_PAIR
  if constexpr (NW) *reinterpret_cast<uint2 *>(p.out_meta2 + 2ull * w) = make_uint2((uint32_t)m.score, n_moves);
  else *reinterpret_cast<uint4 *>(p.out_meta4 + 4ull * w) = make_uint4((uint32_t)m.score, n_moves, end_x, end_y);
}

// one lane per walk: a word of each plane leaves as soon as its 32 columns are known
template <bool NW>
__global__ void __launch_bounds__(64) traceback_moves_lane_kernel(const SaTraceParams p) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= p.n_pairs) return;
  const MoveWalk m = move_walk<NW>(p, w);
  if (!m.valid) return;
  const uint32_t W = p.len_a[m.pair] + 1;
  const uint8_t *__restrict__ Dg = p.dirs + p.mat_off[m.pair];
  const MoveSlot s = m.slot;
  uint32_t x = m.x, y = m.y, st = m.st, k = 0, acc_a = 0, acc_b = 0;
  const bool blk = p.dirs_blocked != 0;
  const uint32_t nbx = (W + 15u) >> 4;
  for (;;) {
    if constexpr (NW) { if (x == 0 || y == 0) break; }
    const uint32_t f = ((uint32_t)Dg[dirs_at(blk, x, y, W, nbx)] >> (2u * st)) & 3u;
    if constexpr (!NW) { if (f == 3u) break; }   // this state's score is 0: the hit starts here (smith_waterman.c:192)
    const uint32_t bit = 0x80000000u >> (k & 31u);   // the walk runs backwards: a word's columns arrive last first
    acc_a |= st == MAT_GAP_A ? bit : 0u;
    acc_b |= st == MAT_GAP_B ? bit : 0u;
    if ((++k & 31u) == 0) {
      const int j = s.nw - (int)(k >> 5);   // the walk's word number k/32 - 1 from the end
      s.plane_a[j] = acc_a; s.plane_b[j] = acc_b;
      acc_a = acc_b = 0;
    }
    x -= (st != MAT_GAP_A);
    y -= (st != MAT_GAP_B);
    st = f;
  }
  if (k & 31u) {   // (the unfinished word: its columns sit at the top, where they belong)
    const int j = s.nw - 1 - (int)(k >> 5);
    s.plane_a[j] = acc_a; s.plane_b[j] = acc_b;
  }
  write_moves_meta<NW>(p, w, m, m.x, m.y, k);
}

// The same walk with the NEXT cell's byte on its way before this cell's byte is looked at (walks of up to 32 stage_words columns).
// A step is bound by the latency of its one-byte load (scattered: 64 walks, 64 cache lines; ~0.55 us), but WHERE the walk goes
// next does not wait for that byte: it follows from the state the walk stands in (alignment.c:311-327 -- MATCH: up-left,
// GAP_A: up, GAP_B: left); the byte only says in which state it ARRIVES.  So two loads are in flight per lane, in two
// registers that take turns.  For the compiler to leave them in flight -- it waits for EVERY load before it touches a loaded
// register as soon as a younger load, or a store, MAY or MAY NOT have been issued on the way -- the loop has no branch but its
// own: the wave runs until its longest walk is over, a walk that is over holds its position, every choice is a select, the
// load is unconditional (a cell outside the matrix, asked for ahead of a walk about to end, reads cell 0 and is answered "every
// state ends here"), and the words of the planes wait in LDS (word j of lane l at [plane][j][l], rewritten every step until
// complete; row stage_words is where walks that are over write) and leave when the wave is done.
template <bool NW>
__global__ void __launch_bounds__(64) traceback_moves_lane_ahead_kernel(const SaTraceParams p) {
  extern __shared__ uint32_t stage[];   // [2][stage_words + 1][64]
  const uint32_t lane = threadIdx.x, rows = p.stage_words + 1u;
  const uint32_t w_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool exists = w_raw < p.n_pairs;
  const uint32_t w = exists ? w_raw : p.n_pairs - 1;      // (a lane without a walk shadows the last one and delivers nothing)
  const MoveWalk m = move_walk<NW>(p, w);
  const uint32_t la = p.len_a[m.pair], lb = p.len_b[m.pair], W = la + 1;
  const uint8_t *__restrict__ Dg = p.dirs + p.mat_off[m.pair];
  const MoveSlot s = m.slot;
  uint32_t x = m.x, y = m.y, st = m.st, k = 0, acc_a = 0, acc_b = 0;
  bool alive = exists && m.valid;
  if constexpr (NW) alive = alive && x != 0 && y != 0;
  struct Ahead { uint32_t raw; bool inside; };
  const bool blk = p.dirs_blocked != 0;
  const uint32_t nbx = (W + 15u) >> 4;
  auto ask = [&](uint32_t cx, uint32_t cy) __attribute__((always_inline)) -> Ahead {
    const bool inside = cx <= la && cy <= lb;
    return Ahead{(uint32_t)Dg[inside ? dirs_at(blk, cx, cy, W, nbx) : 0u], inside};
  };
  Ahead b0 = ask(x, y);
  uint32_t nx = x - (st != MAT_GAP_A), ny = y - (st != MAT_GAP_B);   // (may wrap below 0: `inside` answers, nobody stands there)
  Ahead b1 = ask(nx, ny);
  // one step on the byte `cur` of the cell the walk stands on; afterwards the walk stands on (nx, ny), whose byte is the OTHER
  // register, and `cur` is loaded again with the cell after that
  auto step = [&](Ahead &cur) __attribute__((always_inline)) {
    const uint32_t f = cur.inside ? (cur.raw >> (2u * st)) & 3u : 3u;
    const bool go = NW ? alive : (alive && f != 3u);   // (SW: this state's score is 0 -- the hit starts here, smith_waterman.c:192)
    const uint32_t bit = go ? 0x80000000u >> (k & 31u) : 0u;   // the walk runs backwards: a word's columns arrive last first
    acc_a |= st == MAT_GAP_A ? bit : 0u;
    acc_b |= st == MAT_GAP_B ? bit : 0u;
    const uint32_t row = go ? (uint32_t)s.nw - 1u - (k >> 5) : p.stage_words;   // the word column k belongs to
    stage[row * 64u + lane] = acc_a; stage[(rows + row) * 64u + lane] = acc_b;
    k += go;
    const bool full = go && (k & 31u) == 0;
    acc_a = full ? 0u : acc_a; acc_b = full ? 0u : acc_b;
    x = go ? nx : x; y = go ? ny : y; st = go ? f : st;
    alive = NW ? (go && x != 0 && y != 0) : go;
    nx = x - (st != MAT_GAP_A); ny = y - (st != MAT_GAP_B);
    cur = ask(nx, ny);
  };
  while (__any(alive)) {
    step(b0);
    step(b1);
  }
  if (!exists || !m.valid) return;
  for (int j = s.nw - (int)((k + 31u) >> 5); j < s.nw; ++j) {   // the words the walk touched: the last ceil(k / 32) of its slot
    s.plane_a[j] = stage[(uint32_t)j * 64u + lane]; s.plane_b[j] = stage[(rows + (uint32_t)j) * 64u + lane];
  }
  write_moves_meta<NW>(p, w, m, m.x, m.y, k);
}

// One wave per walk from 64 x 64-byte LDS tiles (traceback_dirs_tile_kernel's walk), all 64 lanes walking the same walk: the
// state is wave-uniform and the compiler keeps it in scalar registers.  With ~10 such waves taking turns on a SIMD the time of
// this kernel is its INSTRUCTION COUNT PER STEP (10 000 walks of 300 steps: ~28 instructions per step 108 us; with four cells
// of a diagonal run taken at once 91 us; this form, 16 instructions per step, 65 us -- profiles/r04/r04_walkers.txt; the LDS
// read's latency is not it: a read-ahead form with more bookkeeping was slower).  Inside a tile a walk cannot reach the tile's
// edge, nor the matrix border, in fewer than min(tx, ty) steps, and it completes a 32-column word every 32 steps: so the steps
// run in BURSTS of min(tx, ty, 32 - k % 32) with nothing in the loop but the byte, the state it says the walk arrives in, the
// place in the tile, and two bits of "which state" shifted into a 64-bit word -- gap-in-a is the low bit of GAP_A = 1, gap-in-b
// the high bit of GAP_B = 2, so the two planes' words are the even and the odd bits of that word, pulled apart once per 32
// steps.  Lane l keeps word l of the current block of 64 words per plane in a register; a block leaves as one coalesced store
// per plane.
// (walks_per_pair: one WAVE per pair walks the pair's hits one after the other -- nearly every pair has one; launched as one
// workgroup per walk slot, the 30 000 of C3's 40 000 slots that return at once cost 0.2 ms of workgroup dispatch, and as one
// workgroup per pair with a wave per slot, four times the LDS per workgroup held for waves that had nothing to do)
// The LOCAL form of the direction byte (sa_kernels.h: SA_LD_*; the fills of chunks whose walks are tile walks write it): the byte
// of a cell holds that cell's OWN comparisons, and the state a walk arrives in follows from the state it left, the byte of the
// cell it left and the byte of the cell it arrives at -- which the walk reads anyway, as the next step's.  alignment.c:311-327:
// GAP_A is tested first, then GAP_B, else MATCH.  Split in two so that only two instructions sit between a step's byte and the
// next step's address (the walks are a chain of dependent LDS reads): when a walk LEAVES a cell it works out which bits of the
// arrival cell's byte will decide (`am`) and what is decided already (`fx`) --
//     leaving in MATCH:  GA and BM of the arrival cell;   in GAP_A: CA ? GAP_A : BM of the arrival cell;
//     in GAP_B: FA ? GAP_A : FB ? GAP_B : MATCH, both of the cell it leaves
// -- beside the read of the next byte, and on arrival the state is one and-or and one look-up in a four-entry constant.
// (A walk that has not moved yet: am = 0, fx = the state it starts in.)
__device__ __forceinline__ void local_depart(uint32_t st, uint32_t cur, uint32_t &am, uint32_t &fx) {
  const uint32_t from_b = (0x64u >> (2u * ((cur >> 3) & 3u))) & 3u;   // FA | FB << 1  ->  GAP_A, GAP_B or MATCH (0x64: 0, 1, 2, 1)
  const bool ca = (cur & SA_LD_CA) != 0;
  am = st == MAT_MATCH ? 3u : (st == MAT_GAP_A && !ca) ? 2u : 0u;
  fx = st == MAT_GAP_B ? from_b : (st == MAT_GAP_A && ca) ? (uint32_t)MAT_GAP_A : 0u;
}
__device__ __forceinline__ uint32_t local_arrive(uint32_t cur, uint32_t am, uint32_t fx) {
  return (0x64u >> (2u * ((cur & am) | fx))) & 3u;   // 0: MATCH, 1: GAP_A, 2: GAP_B, 3 (GA and BM): GAP_A
}
__device__ __forceinline__ uint32_t even_bits(unsigned long long v) {   // bits 0, 2, 4, ... of v as a 32-bit word
  v &= 0x5555555555555555ull;
  v = (v | (v >> 1)) & 0x3333333333333333ull;
  v = (v | (v >> 2)) & 0x0f0f0f0f0f0f0f0full;
  v = (v | (v >> 4)) & 0x00ff00ff00ff00ffull;
  v = (v | (v >> 8)) & 0x0000ffff0000ffffull;
  return (uint32_t)(v | (v >> 16));
}
template <bool NW, bool LOCAL = false>
__global__ void __launch_bounds__(64) traceback_moves_tile_kernel(const SaTraceParams p) {
  constexpr int kT = 64;
  __shared__ __attribute__((aligned(16))) uint8_t tile[kT * kT];
  const int lane = threadIdx.x;
  const uint32_t reps = (!NW && p.walks_per_pair) ? p.walks_per_pair : 1u;
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  for (uint32_t rep = 0; rep < reps; ++rep) {
    const uint32_t w = blockIdx.x * reps + rep;
    if (w >= p.n_pairs) return;
    const MoveWalk m = move_walk<NW>(p, w);
    if (!m.valid) return;
    const uint32_t lb = p.len_b[m.pair], W = p.len_a[m.pair] + 1;
    const uint8_t *__restrict__ Dg = p.dirs + p.mat_off[m.pair];
    const MoveSlot s = m.slot;
    uint32_t x = uni(m.x), y = uni(m.y), st = uni(m.st), k = 0, reg_a = 0, reg_b = 0;
    uint32_t am = 0, fx = st;       // LOCAL: what decides the state at the next cell (local_depart; no step yet: the given state)
    unsigned long long codes = 0;   // the states of the current word's steps, two bits each, the first step on top
    auto flush = [&](int q, int first_slot) {
      const int at = s.nw - 64 * (q + 1) + lane;
      if (lane >= first_slot) { s.plane_a[at] = reg_a; s.plane_b[at] = reg_b; }
    };
    bool over = false;
    while (!over) {
      if constexpr (NW) { if (x == 0 || y == 0) break; }
      // ---- the tile whose bottom-right cell is (x, y) (blocked direction bytes: whose last block row / column hold it)
      uint32_t ox, oy;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      load_dirs_tile<kT>(tile, Dg, p.dirs_blocked != 0, x, y, W, (W + 15u) >> 4, lb, lane, ox, oy);
      ox = uni(ox); oy = uni(oy);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_s_waitcnt(0);
      uint32_t at = (y - oy) * kT + (x - ox);   // the place of the cell the walk stands on: ty * 64 + tx
      auto word_done = [&]() __attribute__((always_inline)) {   // after a step: the 32nd column of a word?
        if ((k & 31u) != 0) return;
        const int j = (int)(k >> 5) - 1;
        const uint32_t wa = even_bits(codes), wb = even_bits(codes >> 1);
        if (lane == 63 - (j & 63)) { reg_a = wa; reg_b = wb; }
        codes = 0;
        if ((j & 63) == 63) flush(j >> 6, 0);
      };
      for (;;) {
        const uint32_t tx = at & (kT - 1), ty = at >> 6;
        // a burst: steps that can neither leave the tile (nor reach the matrix border before its last step) nor pass the end of a word
        uint32_t n = min(min(tx, ty), 32u - (k & 31u));
        if (n == 0) {
          // on the tile's first row or column: one step with the coordinates spelled out, then the next tile
          uint32_t f = 0;
          if constexpr (LOCAL) {
            const uint32_t cur = uni((uint32_t)tile[at]);
            st = local_arrive(cur, am, fx);
            if constexpr (!NW) { if ((cur >> (5u + st)) & 1u) { over = true; break; } }
            local_depart(st, cur, am, fx);
          } else {
            f = uni(((uint32_t)tile[at] >> (2u * st)) & 3u);
            if constexpr (!NW) { if (f == 3u) { over = true; break; } }
          }
          codes = (codes << 2) | st;
          x = ox + tx - (st != MAT_GAP_A); y = oy + ty - (st != MAT_GAP_B);
          if constexpr (!LOCAL) st = f;
          ++k;
          word_done();
          break;
        }
        // (two steps per turn of the loop: its counter and branch are 3 of a step's 16 instructions)
        auto one = [&]() __attribute__((always_inline)) -> bool {
          if constexpr (LOCAL) {
            // the state the walk stands in HERE: from the step that brought it (the state and the cell it left) and this cell's byte
            const uint32_t cur = uni((uint32_t)tile[at]);
            st = local_arrive(cur, am, fx);
            if constexpr (!NW) { if ((cur >> (5u + st)) & 1u) return true; }   // this state's score is 0: the hit starts here
            codes = (codes << 2) | st;
            at -= (0x00014041u >> (8u * st)) & 0xffu;
            local_depart(st, cur, am, fx);
            ++k;
            return false;
          } else {
          const uint32_t f = uni(((uint32_t)tile[at] >> (2u * st)) & 3u);
          if constexpr (!NW) { if (f == 3u) return true; }
          codes = (codes << 2) | st;
          at -= (0x00014041u >> (8u * st)) & 0xffu;   // MATCH: one row and one column back (65), GAP_A: a row (64), GAP_B: a column (1)
          st = f;
          ++k;
          return false;
          }
        };
        bool ended = false;
        for (; n >= 2u; n -= 2u) {
          if (one() || one()) { ended = true; break; }
        }
        if (!ended && n) ended = one();
        if (ended) { over = true; break; }
        word_done();
        x = ox + (at & (kT - 1)); y = oy + (at >> 6);
        if constexpr (NW) { if (x == 0 || y == 0) { over = true; break; } }
      }
    }
    if (k & 31u) {   // the unfinished word: its columns on top
      const int j = (int)(k >> 5);
      const unsigned long long top = codes << (2u * (32u - (k & 31u)));
      const uint32_t wa = even_bits(top), wb = even_bits(top >> 1);
      if (lane == 63 - (j & 63)) { reg_a = wa; reg_b = wb; }
    }
    if (k) {
      const int j = (int)((k - 1) >> 5);
      flush(j >> 6, 63 - (j & 63));
    }
    if (lane == 0) write_moves_meta<NW>(p, w, m, m.x, m.y, k);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

// ---------------------------------------------------------------------------
// Round 6: G walks per wave, in lockstep.
// The tile walker above keeps a walk's state wave-uniform, i.e. in SCALAR registers: a step is ~16 instructions, most of them
// scalar, and a SIMD's eight waves queue for its one scalar issue slot -- C2's 10 000 walks took 63 us where the walks'
// dependent accesses alone would take 6 (profiles/r05/r05g_e2e_roofline.json: 0.05-0.12 of its latency bound).  Here a wave
// carries G walks (G = 4: 16 lanes each, a 64 x 64-byte tile per walk in LDS), every lane holds ITS group's state in vector
// registers, and one step -- the same ~12 vector instructions -- moves G walks: 3-4 instructions per walk and step instead of
// 16.  The walks step in LOCKSTEP: all live walks of the wave have taken the same number of steps k, so bursts (steps that
// can neither leave any group's tile nor pass the end of a 32-column word) have one wave-uniform length, word boundaries
// coincide, and the only scalar work is per burst: its length, and whether some group needs a new tile (a group that does
// gets one -- its 16 lanes load four rows each -- the others keep theirs).  A walk that ends (NW: the matrix border; SW: a
// state whose score is 0) goes quiet: its lanes' updates are predicated off, its step count stays.
// Same moves, same words, same meta as traceback_moves_tile_kernel (every walker test runs both: option walk_group).
// MEASURED (profiles/r06/r06_walkers.txt), on ROW-MAJOR direction bytes: correct, and NOT faster -- C2's 10 000 walks 70 us with
// four or eight walks per wave against 62 us with one; the first version, which renewed a group's tile alone and waited row by
// row, 97 / 153 us.  With 3-4 vector instructions per walk and step the walks take what they took with 16 scalar ones, so
// instructions are not what bounds them THERE.  A tile is 64 row pieces of 64 bytes at a row pitch of len_a + 1 bytes: every piece
// lies on its own 128-byte line (1.5 of them on average), and a walk that climbs a row per step pulls in at least a line per step.
// The counters say so (rocprofv3 --pmc, C2's 10 000 walks of ~171 steps): TCC_MISS 2.17 M lines = 278 MB per launch (FETCH_SIZE
// 135.8 MiB x 2, the guide's gfx950 correction: the same), 1.27 lines per step -- in 62 us that is 4.5 TB/s = 0.56 of the HBM peak
// on 64-byte pieces scattered over 228 MB.  So the direction bytes of the paths only walkers read went into BLOCKS of 8 rows x 16
// columns per line (sa_kernels.h: SA_DIRS_BLOCKED; a tile is then 32 whole lines).  On those this kernel is the faster one: C2's
// walks into a device buffer 42.9 us against 54.6 with one wave per walk, C4's best-hit walks 52 against 72 -- and when the moves
// are written in place into pinned host memory (seqalign_nw_batch's small chunks: no copy behind the kernel) both forms take
// 60-65 us: 20 000 scattered 24-byte pieces over PCIe, ~9 GB/s, are then what the kernel's end waits for.  Default (option
// walk_group = 0): four walks per wave on blocked direction bytes, one wave per walk on row-major ones.
// The tile of one group's walk: KT x KT bytes whose bottom-right cell is where the walk stands (blocked direction bytes: whose
// last block row / column hold it), L = 64 / G lanes loading KT / L rows each; all of a lane's pieces are asked for before the
// first is written to LDS: one latency per reload, not one per row.  ox / oy: the matrix cell of the tile's first byte.
template <int G, int KT = 64>
__device__ __forceinline__ void group_load_tile(const SaTraceParams &p, uint8_t *tile, const uint8_t *__restrict__ Dg, uint32_t x, uint32_t y,
                                                uint32_t W, uint32_t lb, int lg, uint32_t &ox, uint32_t &oy) {
  constexpr int kT = KT, L = 64 / G, kRows = kT / L, kQ = kT / 16;   // rows per lane; 16-byte pieces per row = block columns of a tile
  static_assert(kT % L == 0 && kT % 16 == 0, "whole rows per lane, whole blocks per row");
  typedef uint32_t u4_u __attribute__((ext_vector_type(4), aligned(1)));
  u4_u buf[kRows][kQ];
  if (p.dirs_blocked) {   // (wave-uniform) kT / 8 x kQ blocks whose last block row / column hold (x, y): load_dirs_tile, L lanes a tile
    const uint32_t nbx = (W + 15u) >> 4, nby = (lb + 8u) >> 3;
    const uint32_t bx0 = (x >> 4) >= (uint32_t)(kQ - 1) ? (x >> 4) - (kQ - 1) : 0u, by0 = (y >> 3) >= (uint32_t)(kT / 8 - 1) ? (y >> 3) - (kT / 8 - 1) : 0u;
    ox = bx0 << 4; oy = by0 << 3;
#pragma unroll
    for (int i = 0; i < kRows * kQ; ++i) {
      const uint32_t pc = (uint32_t)(lg * kRows * kQ + i), br = min(by0 + pc / (8u * kQ), nby - 1u), bc = (pc >> 3) % kQ, r = pc & 7u;
      buf[i / kQ][i % kQ] = *reinterpret_cast<const u4_u *>(Dg + ((uint64_t)br * nbx + bx0 + bc) * 128u + r * 16u);
    }
#pragma unroll
    for (int i = 0; i < kRows * kQ; ++i) {
      const uint32_t pc = (uint32_t)(lg * kRows * kQ + i), br = pc / (8u * kQ), bc = (pc >> 3) % kQ, r = pc & 7u;
      *reinterpret_cast<u4_u *>(tile + (br * 8u + r) * kT + bc * 16u) = buf[i / kQ][i % kQ];
    }
  } else {
    ox = x >= (uint32_t)(kT - 1) ? x - (kT - 1) : 0; oy = y >= (uint32_t)(kT - 1) ? y - (kT - 1) : 0;
#pragma unroll
    for (int r4 = 0; r4 < kRows; ++r4) {
      // kT bytes of row oy + tr from column ox on (past the row's end: the next row, or the buffer's slack; a row past the
      // pair's last: that last row again -- never looked at: the walk only moves up and left of (x, y))
      const uint32_t tr = (uint32_t)(lg * kRows + r4), r = min(oy + tr, lb);
      const uint8_t *src = Dg + (uint64_t)r * W + ox;
#pragma unroll
      for (int q = 0; q < kQ; ++q) buf[r4][q] = *reinterpret_cast<const u4_u *>(src + 16 * q);
    }
#pragma unroll
    for (int r4 = 0; r4 < kRows; ++r4) {
      const uint32_t tr = (uint32_t)(lg * kRows + r4);
#pragma unroll
      for (int q = 0; q < kQ; ++q) *reinterpret_cast<u4_u *>(tile + tr * kT + 16 * q) = buf[r4][q];
    }
  }
}

template <bool NW, int G>
__global__ void __launch_bounds__(64) traceback_moves_group_kernel(const SaTraceParams p) {
  constexpr int kT = 64, L = 64 / G;           // tile edge; lanes per walk
  static_assert(G == 2 || G == 4 || G == 8, "walks per wave");
  __shared__ __attribute__((aligned(16))) uint8_t tiles[G * kT * kT];
  const int lane = threadIdx.x, g = lane / L, lg = lane % L;
  uint8_t *const tile = tiles + g * (kT * kT);
  const uint32_t w_raw = blockIdx.x * G + g;
  const bool exists = w_raw < p.n_pairs;
  const uint32_t w = exists ? w_raw : p.n_pairs - 1;     // (a group without a walk shadows the last one and delivers nothing)
  const MoveWalk m = move_walk<NW>(p, w);
  const uint32_t lb = p.len_b[m.pair], W = p.len_a[m.pair] + 1;
  const uint8_t *__restrict__ Dg = p.dirs + p.mat_off[m.pair];
  const MoveSlot s = m.slot;
  uint32_t x = m.x, y = m.y, st = m.st, kk = 0, reg_a = 0, reg_b = 0, ox = 0, oy = 0, at = 0;
  unsigned long long codes = 0;                // the states of the current word's steps, two bits each, the first step on top
  bool live = exists && m.valid;
  if constexpr (NW) live = live && x != 0 && y != 0;
  bool fresh = true;                           // no tile yet
  uint32_t k = 0;                              // steps every live walk has taken (wave-uniform)
  // word j (the walk's steps 32 j .. 32 j + 31) of my group's walk: lane 15 - j % L keeps it; a block of L words leaves as one run
  auto keep_word = [&](uint32_t j, uint32_t wa, uint32_t wb, bool mine) __attribute__((always_inline)) {
    if (mine && lg == (L - 1) - (int)(j % L)) { reg_a = wa; reg_b = wb; }
  };
  auto flush = [&](uint32_t q, int first_slot, bool mine) __attribute__((always_inline)) {
    const int dst = s.nw - L * (int)(q + 1) + lg;
    if (mine && lg >= first_slot) { s.plane_a[dst] = reg_a; s.plane_b[dst] = reg_b; }
  };
  while (__any(live)) {
    // ---- tiles: when some live walk has none, or stands on its tile's first row or column, EVERY live walk gets the tile whose
    // bottom-right cell is where it stands.  (A first version gave a new tile only to the group that needed one: four walks'
    // tiles then run out at four different times, ~15 reloads per wave instead of a walk's own 3-4, each a full memory latency
    // for everybody -- 97 us for C2's walks where one wave per walk takes 63.  Walks leave a tile after 63 .. 126 steps whatever
    // their shape, so tiles renewed together run out together: max steps / 63 reloads per wave.)
    const uint32_t tx0 = at & (kT - 1), ty0 = at >> 6;
    if (__any(live && (fresh || tx0 == 0 || ty0 == 0))) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // earlier LDS reads are done with the old tiles
      group_load_tile<G>(p, tile, Dg, x, y, W, lb, lg, ox, oy);
      at = (y - oy) * kT + (x - ox);
      fresh = false;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_s_waitcnt(0);                           // the tiles are in LDS before anyone reads them (one wave)
    }
    // ---- a burst: steps during which no live walk can leave its tile (at least one: a walk on its tile's edge has just been
    // given a new tile -- or stands on the matrix border, SW only, where every state's score is 0 and the step ends it), and
    // none passes the end of a word
    uint32_t room = live ? max(min(at & (kT - 1), at >> 6), 1u) : 64u;
#pragma unroll
    for (int o = L; o < 64; o <<= 1) room = min(room, (uint32_t)__shfl_xor((int)room, o));
    const uint32_t n = min((uint32_t)__builtin_amdgcn_readfirstlane((int)room), 32u - (k & 31u));
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t f = ((uint32_t)tile[at] >> (2u * st)) & 3u;
      bool go = live;
      if constexpr (!NW) go = go && f != 3u;   // this state's score is 0: the hit starts here (smith_waterman.c:192)
      codes = go ? (codes << 2) | st : codes;
      at -= go ? (0x00014041u >> (8u * st)) & 0xffu : 0u;   // MATCH: a row and a column back (65), GAP_A: a row (64), GAP_B: a column (1)
      st = go ? f : st;
      kk += go;
      if constexpr (!NW) live = go;
    }
    k += n;
    x = ox + (at & (kT - 1)); y = oy + (at >> 6);
    // a word of 32 columns complete: for the walks that have come this far
    if ((k & 31u) == 0) {
      const bool mine = exists && m.valid && kk == k;
      const uint32_t j = (k >> 5) - 1;
      keep_word(j, even_bits(codes), even_bits(codes >> 1), mine);
      codes = mine ? 0ull : codes;
      if (j % L == L - 1) flush(j / L, 0, mine);
    }
    if constexpr (NW) live = live && x != 0 && y != 0;
  }
  const bool mine = exists && m.valid;
  if (kk & 31u) {   // the unfinished word: its columns on top
    const uint32_t j = kk >> 5;
    const unsigned long long top = codes << (2u * (32u - (kk & 31u)));
    keep_word(j, even_bits(top), even_bits(top >> 1), mine);
  }
  if (kk) {         // the last block of words: from the slot of the last word on
    const uint32_t j = (kk - 1) >> 5;
    flush(j / L, (L - 1) - (int)(j % L), mine);
  }
  if (mine && lg == 0) write_moves_meta<NW>(p, w, m, m.x, m.y, kk);
}

// The same walker on the LOCAL form of the direction byte (sa_kernels.h: SA_LD_*; local_depart / local_arrive above), its step
// written for the instruction count: with two or three such waves on a SIMD all in the same phase (tiles arrive together, bursts
// run together) every instruction of the step costs the launch 0.7 us (profiles/r06/r06_local_dirs.txt: the first form of this
// decode, 29 vector instructions per step where the older byte takes 14, ran 71 us where that takes 60).  So:
//   * every small function of two or three bits is ONE v_perm_b32 -- a byte look-up in a constant (selector bytes 1-3 = 0x0c
//     give zero bytes, entry 7 of the eight-entry tables provides that 0x0c for the next selector): the state a walk arrives in
//     and the distance to the next cell from t = (cur & am) | fx; which bits of the byte the departure looks at, from the state;
//     am and fx from those bits;
//   * the field of the byte a departure looks at (GAP_A: CA; GAP_B: FA, FB; MATCH: none) is one v_bfe_u32 whose WIDTH is the
//     state itself (0, 1, 2 bits) and whose offset comes from the table;
//   * the states of 16 steps collect in 32 bits (v_lshl_or_b32; 64-bit shifts are four instructions with their selects);
//   * Needleman-Wunsch: nothing is predicated per step -- within a burst no walk can reach the border, a walk that is over
//     gets 0 as its table of distances (it stays where it is) and its words are put back after the burst.
// Same moves, words and meta as the kernels above; tests run every form (options dirs_local, walk_group).
template <bool NW, int G, int KT = 64>
__global__ void __launch_bounds__(64) traceback_moves_group_local_kernel(const SaTraceParams p) {
  constexpr int kT = KT, L = 64 / G, kSh = KT == 64 ? 6 : 5;
  static_assert(KT == 64 || KT == 32, "tile edge");
  static_assert(G == 2 || G == 4 || G == 8, "walks per wave");
  __shared__ __attribute__((aligned(16))) uint8_t tiles[G * kT * kT];
  const int lane = threadIdx.x, g = lane / L, lg = lane % L;
  const uint32_t tbase = (uint32_t)g * (kT * kT);
  const uint32_t w_raw = blockIdx.x * G + g;
  const bool exists = w_raw < p.n_pairs;
  const uint32_t w = exists ? w_raw : p.n_pairs - 1;
  const MoveWalk m = move_walk<NW>(p, w);
  const uint32_t lb = p.len_b[m.pair], W = p.len_a[m.pair] + 1;
  const uint8_t *__restrict__ Dg = p.dirs + p.mat_off[m.pair];
  const MoveSlot s = m.slot;
  constexpr uint32_t kState = 0x01020100u;                   // t -> MATCH, GAP_A, GAP_B, GAP_A (GA and BM)
  constexpr uint32_t kDist = (uint32_t)kT << 24 | 1u << 16 | (uint32_t)kT << 8 | (uint32_t)(kT + 1);   // t -> a row and a column back (kT + 1), a row (kT), a column (1), a row
  constexpr uint32_t kField = 0x00632202u;                   // state -> offset of the departure's field | first table entry << 5
  constexpr uint32_t kAmLo = 0x00000203u, kAmHi = 0u;        // entry -> am: MATCH 3; GAP_A: !CA 2, CA 0; GAP_B 0
  constexpr uint32_t kFxLo = 0x00010000u, kFxHi = 0x0c010201u;   // entry -> fx: GAP_A & CA: GAP_A; GAP_B: FA | FB << 1 -> 0 1 2 1; [7] = 0x0c
  constexpr uint32_t kEnd = 0x00804020u;                     // state -> its "score is 0" bit of the byte (SW)
  uint32_t x = m.x, y = m.y, kk = 0, reg_a = 0, reg_b = 0, ox = 0, oy = 0, at = tbase;
  // what the LAST step left behind: the state it left in and the byte of the cell it left -- the departure's half of the decode
  // (local_depart) is worked out from them at the START of the next step, beside that step's LDS read, so that only the arrival's
  // three instructions sit between a byte and the next address.  No step yet: "left in GAP_B with FA / FB saying the given state".
  uint32_t ps = MAT_GAP_B, pc = m.st << 3;
  uint32_t codes = 0, codes_hi = 0;            // the states of the current word's steps, two bits each: steps 16-31 / steps 0-15, the first on top
  bool live = exists && m.valid;
  if constexpr (NW) live = live && x != 0 && y != 0;
  bool fresh = true;
  uint32_t k = 0;                              // steps every live walk has taken (wave-uniform)
  auto keep_word = [&](uint32_t j, uint32_t wa, uint32_t wb, bool mine) __attribute__((always_inline)) {
    if (mine && lg == (L - 1) - (int)(j % L)) { reg_a = wa; reg_b = wb; }
  };
  auto flush = [&](uint32_t q, int first_slot, bool mine) __attribute__((always_inline)) {
    const int dst = s.nw - L * (int)(q + 1) + lg;
    if (mine && lg >= first_slot) { s.plane_a[dst] = reg_a; s.plane_b[dst] = reg_b; }
  };
  // one step: the byte is asked for, the last step's departure decoded while it is on its way (am: which bits of this byte decide,
  // fx: what is decided already), then the arrival: t, the state the walk stands in here
  auto step_state = [&](uint32_t &cur, uint32_t &t) __attribute__((always_inline)) -> uint32_t {
    cur = (uint32_t)tiles[at];
    const uint32_t fld = __builtin_amdgcn_perm(0u, kField, ps);
    const uint32_t q = __builtin_amdgcn_ubfe(pc, fld, ps) + __builtin_amdgcn_ubfe(fld, 5u, 3u) + 0x07070700u;
    const uint32_t am = __builtin_amdgcn_perm(kAmHi, kAmLo, q), fx = __builtin_amdgcn_perm(kFxHi, kFxLo, q);
    t = (cur & am) | fx;
    return __builtin_amdgcn_perm(0u, kState, t);
  };
  while (__any(live)) {
    const uint32_t tx0 = at & (kT - 1), ty0 = (at >> kSh) & (kT - 1);
    if (__any(live && (fresh || tx0 == 0 || ty0 == 0))) {   // (tiles renewed together: traceback_moves_group_kernel)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      group_load_tile<G, KT>(p, tiles + tbase, Dg, x, y, W, lb, lg, ox, oy);
      at = tbase + (y - oy) * kT + (x - ox);
      fresh = false;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_s_waitcnt(0);
    }
    // ---- a burst: no live walk can leave its tile, none passes the end of a half word
    uint32_t room = live ? max(min(at & (kT - 1), (at >> kSh) & (kT - 1)), 1u) : 64u;
#pragma unroll
    for (int o = L; o < 64; o <<= 1) room = min(room, (uint32_t)__shfl_xor((int)room, o));
    const uint32_t n = min((uint32_t)__builtin_amdgcn_readfirstlane((int)room), 16u - (k & 15u));
    if constexpr (NW) {
      const uint32_t dist = live ? kDist : 0u, codes0 = codes;
      for (uint32_t i = 0; i < n; ++i) {
        uint32_t cur, t;
        const uint32_t st = step_state(cur, t);
        at -= __builtin_amdgcn_perm(0u, dist, t);
        codes = (codes << 2) | st;
        ps = st; pc = cur;
      }
      codes = live ? codes : codes0;
      kk += live ? n : 0u;
    } else {
      uint32_t dist = live ? kDist : 0u;
      for (uint32_t i = 0; i < n; ++i) {
        uint32_t cur, t;
        const uint32_t st = step_state(cur, t);
        const bool go = live && (cur & __builtin_amdgcn_perm(0u, kEnd, st)) == 0;   // else: this state's score is 0, the hit starts here
        dist = go ? dist : 0u;
        at -= __builtin_amdgcn_perm(0u, dist, t);
        codes = go ? (codes << 2) | st : codes;
        kk += go;
        ps = go ? st : ps; pc = go ? cur : pc;
        live = go;
      }
    }
    k += n;
    x = ox + (at & (kT - 1)); y = oy + ((at >> kSh) & (kT - 1));
    // half a word / a word of 32 columns complete: for the walks that have come this far
    if ((k & 15u) == 0) {
      const bool mine = exists && m.valid && kk == k;
      if (k & 16u) {
        codes_hi = mine ? codes : codes_hi;
        codes = mine ? 0u : codes;
      } else {
        const unsigned long long full = (unsigned long long)codes_hi << 32 | codes;
        const uint32_t j = (k >> 5) - 1;
        keep_word(j, even_bits(full), even_bits(full >> 1), mine);
        codes = mine ? 0u : codes; codes_hi = mine ? 0u : codes_hi;
        if (j % L == L - 1) flush(j / L, 0, mine);
      }
    }
    if constexpr (NW) live = live && x != 0 && y != 0;
  }
  const bool mine = exists && m.valid;
  if (kk & 31u) {   // the unfinished word: its columns on top
    const uint32_t j = kk >> 5, r = kk & 31u;
    const unsigned long long part = r >= 16u ? ((unsigned long long)codes_hi << (2u * (r - 16u))) | codes : (unsigned long long)codes;
    const unsigned long long top = part << (2u * (32u - r));
    keep_word(j, even_bits(top), even_bits(top >> 1), mine);
  }
  // The words' way home.  In place over PCIe (seqalign_nw_batch's small chunks: no copy behind the kernel) a wave's G walks leave
  // as 2 G pieces of ~24 bytes, each its own partial line -- ~20 000 of them for BASELINE configs[1], and the kernel's end waits
  // for them (60 us where the walk needs 43: profiles/r06/r06_walkers.txt).  The walks' slots lie one behind the other
  // (move_walk), so when every walk of the wave fits ONE block of L words per plane the wave puts its words where they belong
  // in a copy of its whole region in LDS -- the tiles are done with -- zeros elsewhere, and writes the region as one run of
  // consecutive lanes: whole lines but for its two ends.  (Words of a slot that the walk did not fill are never looked at:
  // host/sa_moves.c takes the last ceil(columns / 32) of each plane.)
  const uint32_t nwu = (uint32_t)s.nw;
  if (p.tune_stage && !__any(exists && nwu > (uint32_t)L)) {
    uint32_t *stg = reinterpret_cast<uint32_t *>(tiles);
    const unsigned long long base_me = (unsigned long long)(s.plane_a - p.moves);      // my slot, in words from p.moves
    const uint32_t n_here = min((uint32_t)G, p.n_pairs - blockIdx.x * G);               // walks of this wave (>= 1)
    const unsigned long long base0 = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base_me, 0) |
                                     (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base_me >> 32), 0) << 32;
    const uint32_t off = (uint32_t)(base_me - base0);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)(off + 2u * nwu), (int)((n_here - 1u) * L));   // (slots ascend)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (uint32_t i = (uint32_t)lane; i < total; i += 64u) stg[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (mine && kk) {
      const uint32_t j = (kk - 1) >> 5;                      // (< L: one block)
      const int dst = (int)nwu - L + lg;
      if (lg >= (L - 1) - (int)j) { stg[off + (uint32_t)dst] = reg_a; stg[off + nwu + (uint32_t)dst] = reg_b; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (uint32_t i = (uint32_t)lane; i < total; i += 64u) p.moves[base0 + i] = stg[i];
  } else if (kk) {
    const uint32_t j = (kk - 1) >> 5;
    flush(j / L, (L - 1) - (int)(j % L), mine);
  }
  if (mine && lg == 0) write_moves_meta<NW>(p, w, m, m.x, m.y, kk);
}

// ---------------------------------------------------------------------------
// One WAVE per pair, the walk's neighbourhood staged in LDS.
//
// A step needs the three matrices at ONE predecessor cell and the two sequence
// codes: with one lane per pair every step is two or three dependent HBM round
// trips (~1.5 us), so a 10 000 x 10 000 pair took 30 ms to trace after 5 ms of
// fill, and the 157 waves of a 10 k-pair batch left the chip idle.  Here the wave
// loads the kTile x kTile block of cells that ends at the current cell (one
// dwordx4 per lane and matrix) plus the codes / characters of those rows and
// columns; the walk then runs out of LDS -- it needs at least kTile steps to leave
// a tile -- and reloads when it steps outside.  All 64 lanes walk redundantly
// (uniform control flow), lane 0 writes the characters.
constexpr int kTile = 16;

struct TileAccess {
  const PairView &v;
  const uint16_t *code;
  int32_t *cells;        // [3][kTile * kTile]
  uint16_t *codes;       // [2][kTile]: code of seq_a[ox-1 + c], seq_b[oy-1 + r]
  uint8_t *chars;        // [2][kTile]
  uint32_t ox, oy;       // tile covers columns ox .. ox+kTile-1, rows oy .. oy+kTile-1
  bool loaded;
  int lane;

  // make (x, y) the tile's bottom-right cell
  __device__ __forceinline__ void refill(uint32_t x, uint32_t y) {
    ox = x >= (uint32_t)(kTile - 1) ? x - (kTile - 1) : 0;
    oy = y >= (uint32_t)(kTile - 1) ? y - (kTile - 1) : 0;
    const uint32_t r = lane >> 2, c0 = (lane & 3) * 4, H = v.lb + 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // earlier LDS reads are done with the old tile
    if (oy + r < H) {
      const uint32_t at = (oy + r) * v.W + ox + c0;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int32_t *src = (m == 0 ? v.M : m == 1 ? v.A : v.B) + at;
        int32_t *dst = cells + m * (kTile * kTile) + r * kTile + c0;
        if (ox + c0 + 4 <= v.W) {
          const v4i_u q = *reinterpret_cast<const v4i_u *>(src);
          dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = (ox + c0 + e < v.W) ? src[e] : 0;
        }
      }
    }
    // sequence characters / codes of the tile's columns (seq_a[ox-1+c]) and rows (seq_b[oy-1+r])
    if (lane < 2 * kTile) {
      const bool is_b = lane >= kTile;
      const uint32_t i = (is_b ? oy : ox) + (lane & (kTile - 1));   // matrix coordinate; sequence index i-1
      const uint32_t n = is_b ? v.lb : v.la;
      uint8_t ch = 0;
      if (i >= 1 && i <= n) ch = (is_b ? v.seq_b : v.seq_a)[i - 1];
      chars[lane] = ch;
      codes[lane] = code[ch];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);      // the tile is in LDS before anyone reads it (one wave: program order)
    loaded = true;
  }
  __device__ __forceinline__ bool inside(uint32_t x, uint32_t y) const {
    return loaded && x >= ox && x < ox + kTile && y >= oy && y < oy + kTile;
  }
  // sequence index i (matrix column i+1)
  __device__ __forceinline__ int code_a(uint32_t i) const { return codes[i + 1 - ox]; }
  __device__ __forceinline__ int code_b(uint32_t j) const { return codes[kTile + j + 1 - oy]; }
  __device__ __forceinline__ char char_a(uint32_t i) const { return (char)chars[i + 1 - ox]; }
  __device__ __forceinline__ char char_b(uint32_t j) const { return (char)chars[kTile + j + 1 - oy]; }
  __device__ __forceinline__ void cell(uint32_t x, uint32_t y, int &m, int &a, int &b) const {
    const uint32_t t = (y - oy) * kTile + (x - ox);
    m = cells[t]; a = cells[kTile * kTile + t]; b = cells[2 * kTile * kTile + t];
  }
};

// a step from (x, y) touches column x / row y (codes, characters) and the cell (x-1 | x, y-1 | y):
// all inside the tile iff x-1 >= ox and y-1 >= oy (or the coordinate is 0 and so is the origin)
__device__ __forceinline__ bool step_inside(const TileAccess &t, uint32_t x, uint32_t y) {
  if (!t.inside(x, y)) return false;
  return (x == 0 || x - 1 >= t.ox) && (y == 0 || y - 1 >= t.oy);
}

template <bool SW>
__global__ void __launch_bounds__(kWave *kWavesPerBlock) traceback_wave_kernel(const SaTraceParams p) {
  __shared__ int32_t tile_cells[kWavesPerBlock][3 * kTile * kTile];
  __shared__ uint16_t tile_codes[kWavesPerBlock][2 * kTile];
  __shared__ uint8_t tile_chars[kWavesPerBlock][2 * kTile];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const uint32_t w = blockIdx.x * kWavesPerBlock + wave;
  if (w >= p.n_pairs) return;
  const uint32_t pair = p.walker_pair ? p.walker_pair[w] : w;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair], W = la + 1;
  const uint64_t mo = p.mat_off[pair];
  const PairView v{p.arena + p.off_a[pair], p.arena + p.off_b[pair], p.M + mo, p.A + mo, p.B + mo, la, lb, W};
  const TraceConsts k{p.code, p.table, (int)p.K, p.open1, p.ext, p.gen_eq, p.gen_ne,
                      (p.flags & SA_F_NO_START_GAP) != 0, (p.flags & SA_F_NO_END_GAP) != 0,
                      (p.flags & SA_F_NO_GAPS_A) != 0, (p.flags & SA_F_NO_GAPS_B) != 0};
  char *oa = p.out_a + p.str_off[w];
  char *ob = p.out_b + p.str_off[w];
  TileAccess t{v, p.code, tile_cells[wave], tile_codes[wave], tile_chars[wave], 0, 0, false, lane};

  int matrix = MAT_MATCH, score;
  uint32_t x, y, head = la + lb, err = 0;
  if constexpr (SW) {
    sw_walk_start(p, w, pair, W, x, y);
    t.refill(x, y);
    int a_, b_;
    t.cell(x, y, score, a_, b_);
  } else {
    // end cell: ties resolve GAP_A > GAP_B > MATCH (needleman_wunsch.c:53-66)
    x = la; y = lb;
    t.refill(x, y);
    int m_, a_, b_;
    t.cell(x, y, m_, a_, b_);
    score = m_;
    if (b_ >= score) { matrix = MAT_GAP_B; score = b_; }
    if (a_ >= score) { matrix = MAT_GAP_A; score = a_; }
  }
  const int end_score = score;
  const uint32_t end_x = x, end_y = y;

  while (SW ? (score > 0) : (x > 0 && y > 0)) {
    if (!step_inside(t, x, y)) t.refill(x, y);
    --head;
    if (lane == 0) {
      oa[head] = (matrix == MAT_GAP_A) ? '-' : t.char_a(x - 1);
      ob[head] = (matrix == MAT_GAP_B) ? '-' : t.char_b(y - 1);
    }
    if ((err = reverse_move_t(t, k, la, lb, x, y, matrix, score))) break;
  }
  if (lane == 0) {
    if constexpr (!SW) {
      if (!err) {   // needleman_wunsch.c:117-132: the rest of the longer sequence against gaps
        for (; y > 0; --y) { --head; oa[head] = '-'; ob[head] = (char)v.seq_b[y - 1]; }
        for (; x > 0; --x) { --head; oa[head] = (char)v.seq_a[x - 1]; ob[head] = '-'; }
      }
    } else {
      // smith_waterman.c:251-255: start position and consumed lengths
      p.out_pos[4 * w + 0] = x;
      p.out_pos[4 * w + 1] = y;
      p.out_pos[4 * w + 2] = end_x - x;
      p.out_pos[4 * w + 3] = end_y - y;
    }
    write_walk_meta(p, w, pair, head, la + lb - head, end_score, err);
  }
}

}  // namespace sa

hipError_t sa_launch_nw_traceback(const SaTraceParams &p, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  if (p.dirs) {
    // walks on direction bytes: one wave per walk from LDS tiles for small launches (bound by its instruction count per step:
    // ~6.5 ns per 300-step walk once the chip is full -- 10 000 walks 0.065 ms; its moves leave coalesced, in place over PCIe:
    // sa_batch.hip), one lane per walk with the next cell's byte asked for ahead beyond (bound by the sectors its scattered
    // bytes pull in: ~3.3 ns per walk -- 31 250 walks 0.104 ms, 46 875 walks 0.158 ms -- behind a device buffer and a copy).
    // seqalign_nw_batch end to end, 150 x 150: the tile form 8 % ahead at 10 000 pairs, 4 % at 24 576, equal at 32 768
    // (profiles/r04/r04_walkers.txt): SA_WALK_TILE_MAX.  The option trace_kernel = lane | wave forces one.
    // (round 6: a launch on the byte's LOCAL form is a tile walk whatever its size -- the host chose the form knowing that)
    const bool tiles = p.tune_walker ? p.tune_walker == 2 : (p.dirs_local != 0 || p.n_pairs < SA_WALK_TILE_MAX);
    if (p.nw_state) {   // NW behind the directions-only fill
      if (!p.nw_score) return hipErrorInvalidValue;
      if (p.moves) {    // ... sending home moves instead of strings
        if (!p.out_meta2) return hipErrorInvalidValue;
        sa_record_launch(tiles ? SEQALIGN_K_WALK_MOVES_TILE : SEQALIGN_K_WALK_MOVES_LANE, p.n_pairs);
        // (round 6: walks per wave of the tile walker -- option walk_group; by itself: four in lockstep on BLOCKED direction bytes
        // (43 against 55 us for C2's walks into a device buffer, 52 against 72 for C4's), one wave per walk on row-major ones, where
        // both are bound by the lines the rows' pieces pull in and the lockstep form is the slower: profiles/r06/r06_walkers.txt)
        const uint32_t grp = p.tune_group ? p.tune_group : (p.dirs_blocked ? 4u : 1u);
        if (p.dirs_local) {   // the byte's local form (sa_kernels.h): read by the tile walkers only -- the host asked for it knowing that
          if (!tiles) return hipErrorInvalidValue;
          if (grp == 8) hipLaunchKernelGGL((sa::traceback_moves_group_local_kernel<true, 8>), dim3((p.n_pairs + 7) / 8), dim3(64), 0, stream, p);
          // (tile edge, option walk_tile: 32 x 32 bytes by default for these walks -- 8 lines per reload instead of 32, good for >= 16 steps
          //  instead of >= 48: C2's walks 58.7 -> 53.4 us; the best-hit walks are level, 64.5-65.1 against 65.7-66.6, and keep 64:
          //  profiles/r06/r06_local_dirs.txt)
          else if (grp == 4 && p.tune_tile != 64) hipLaunchKernelGGL((sa::traceback_moves_group_local_kernel<true, 4, 32>), dim3((p.n_pairs + 3) / 4), dim3(64), 0, stream, p);
          else if (grp == 4) hipLaunchKernelGGL((sa::traceback_moves_group_local_kernel<true, 4>), dim3((p.n_pairs + 3) / 4), dim3(64), 0, stream, p);
          else hipLaunchKernelGGL((sa::traceback_moves_tile_kernel<true, true>), dim3(p.n_pairs), dim3(64), 0, stream, p);
        } else
        if (tiles && grp == 8) hipLaunchKernelGGL((sa::traceback_moves_group_kernel<true, 8>), dim3((p.n_pairs + 7) / 8), dim3(64), 0, stream, p);
        else if (tiles && grp == 4) hipLaunchKernelGGL((sa::traceback_moves_group_kernel<true, 4>), dim3((p.n_pairs + 3) / 4), dim3(64), 0, stream, p);
        else if (tiles) hipLaunchKernelGGL(sa::traceback_moves_tile_kernel<true>, dim3(p.n_pairs), dim3(64), 0, stream, p);
        else if (p.stage_words && p.stage_words <= 95u) hipLaunchKernelGGL(sa::traceback_moves_lane_ahead_kernel<true>, dim3((p.n_pairs + 63) / 64), dim3(64), (size_t)(p.stage_words + 1) * 512, stream, p);
        else hipLaunchKernelGGL(sa::traceback_moves_lane_kernel<true>, dim3((p.n_pairs + 63) / 64), dim3(64), 0, stream, p);
      } else {
      if (p.dirs_local) return hipErrorInvalidValue;   // (the walkers that write strings read the older form)
      sa_record_launch(tiles ? SEQALIGN_K_WALK_DIRS_TILE : SEQALIGN_K_WALK_DIRS_LANE, p.n_pairs);
      if (tiles) hipLaunchKernelGGL(sa::traceback_dirs_tile_kernel<true>, dim3(p.n_pairs), dim3(64), 0, stream, p);
      else hipLaunchKernelGGL(sa::traceback_nw_dirs_kernel, dim3((p.n_pairs + 63) / 64), dim3(64), 0, stream, p);
      }
    } else if (p.moves) {   // SW hits, moves home: score, walked columns and end cell per walk in out_meta4
      if (!(p.hit_keys || (p.start_index && p.start_score)) || !p.out_meta4) return hipErrorInvalidValue;
      if (p.walks_per_pair && !(p.hit_keys && p.hit_count && p.sweep_status)) return hipErrorInvalidValue;
      // (walks_per_pair: most of the launch's walks return at once -- the choice follows the pairs, not the slots)
      const bool wtiles = p.tune_walker ? p.tune_walker == 2 : (p.dirs_local != 0 || (p.walks_per_pair ? p.n_pairs / p.walks_per_pair < SA_WALK_TILE_MAX : p.n_pairs < SA_WALK_TILE_MAX));
      sa_record_launch(wtiles ? SEQALIGN_K_WALK_MOVES_TILE : SEQALIGN_K_WALK_MOVES_LANE, p.n_pairs);
      const uint32_t wpb = p.walks_per_pair ? p.walks_per_pair : 1u;   // (<= 8: seqalign_sw_batch's one-trip path)
      if (wpb > 8) return hipErrorInvalidValue;
      // (walks_per_pair -- the one-trip multi-hit call, most slots empty -- stays one wave per pair)
      const uint32_t grp = p.tune_group ? p.tune_group : (p.dirs_blocked ? 4u : 1u);
      if (p.dirs_local) {   // the best-hit path's bytes in their local form: tile walkers, one walk per pair
        if (!wtiles || p.walks_per_pair) return hipErrorInvalidValue;
        if (grp == 8) hipLaunchKernelGGL((sa::traceback_moves_group_local_kernel<false, 8>), dim3((p.n_pairs + 7) / 8), dim3(64), 0, stream, p);
        else if (grp == 4 && p.tune_tile == 32) hipLaunchKernelGGL((sa::traceback_moves_group_local_kernel<false, 4, 32>), dim3((p.n_pairs + 3) / 4), dim3(64), 0, stream, p);
        else if (grp == 4) hipLaunchKernelGGL((sa::traceback_moves_group_local_kernel<false, 4>), dim3((p.n_pairs + 3) / 4), dim3(64), 0, stream, p);
        else hipLaunchKernelGGL((sa::traceback_moves_tile_kernel<false, true>), dim3(p.n_pairs), dim3(64), 0, stream, p);
      } else
      if (wtiles && !p.walks_per_pair && grp == 8) hipLaunchKernelGGL((sa::traceback_moves_group_kernel<false, 8>), dim3((p.n_pairs + 7) / 8), dim3(64), 0, stream, p);
      else if (wtiles && !p.walks_per_pair && grp == 4) hipLaunchKernelGGL((sa::traceback_moves_group_kernel<false, 4>), dim3((p.n_pairs + 3) / 4), dim3(64), 0, stream, p);
      else if (wtiles) hipLaunchKernelGGL(sa::traceback_moves_tile_kernel<false>, dim3((p.n_pairs + wpb - 1) / wpb), dim3(64), 0, stream, p);
      else if (p.stage_words && p.stage_words <= 95u) hipLaunchKernelGGL(sa::traceback_moves_lane_ahead_kernel<false>, dim3((p.n_pairs + 63) / 64), dim3(64), (size_t)(p.stage_words + 1) * 512, stream, p);
      else hipLaunchKernelGGL(sa::traceback_moves_lane_kernel<false>, dim3((p.n_pairs + 63) / 64), dim3(64), 0, stream, p);
    } else {            // SW hits behind sa_fill_dirs.hip
      if (!(p.hit_keys || (p.start_index && p.start_score)) || !p.out_pos || p.dirs_local) return hipErrorInvalidValue;
      sa_record_launch(tiles ? SEQALIGN_K_WALK_DIRS_TILE : SEQALIGN_K_WALK_DIRS_LANE, p.n_pairs);
      if (tiles) hipLaunchKernelGGL(sa::traceback_dirs_tile_kernel<false>, dim3(p.n_pairs), dim3(64), 0, stream, p);
      else hipLaunchKernelGGL(sa::traceback_dirs_kernel, dim3((p.n_pairs + 63) / 64), dim3(64), 0, stream, p);
    }
    return hipGetLastError();
  }
  const bool sw = p.start_index || p.hit_keys;
  // Measured (round 2 record: profiles/r02/; the tool was pruned in round 5): the tiled wave-per-pair walker wins when there are few pairs
  // (1 x 10 000^2: 9.5 -> 7.4 ms, 16 x 5 000^2: 6.6 -> 4.2 ms of traceback) and loses a little when the lanes of
  // one-lane-per-pair waves are all busy (10 k x 150^2: +0.13 ms).  The option trace_kernel = lane | wave forces one.
  // SW walks (a hit is ~the shorter sequence long): the tiled walker also wins with 10 000 walks (C3: 0.58 -> 0.47 ms,
  // C4: 0.88 -> 0.45 ms)
  const bool lane_kernel = p.tune_walker ? p.tune_walker == 1 : (!sw && p.n_pairs >= 2048);
  sa_record_launch(lane_kernel ? SEQALIGN_K_WALK_LANE : SEQALIGN_K_WALK_WAVE, p.n_pairs);
  if (lane_kernel) {
    const dim3 grid((p.n_pairs + 63) / 64), block(64);   // one wave per workgroup: spread over all CUs
    if (sw) hipLaunchKernelGGL(sa::traceback_kernel<true>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(sa::traceback_kernel<false>, grid, block, 0, stream, p);
  } else {
    const dim3 grid((p.n_pairs + sa::kWavesPerBlock - 1) / sa::kWavesPerBlock), block(sa::kWave * sa::kWavesPerBlock);
    if (sw) hipLaunchKernelGGL(sa::traceback_wave_kernel<true>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(sa::traceback_wave_kernel<false>, grid, block, 0, stream, p);
  }
  return hipGetLastError();
}
