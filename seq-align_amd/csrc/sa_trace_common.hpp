// sa_trace_common.hpp -- one traceback step on the device, shared by
// sa_traceback.hip (NW / SW hits) and sa_sw_sweep.hip (SW hit enumeration: where a walk goes from a cell).
// Decision order of alignment_reverse_move (reference src/alignment.c:244-350).
#pragma once

#include "sa_fill_common.hpp"

namespace sa {

enum { MAT_MATCH = 0, MAT_GAP_A = 1, MAT_GAP_B = 2 };

// read-only view of one pair for the walkers
struct PairView {
  const uint8_t *seq_a, *seq_b;
  const int32_t *M, *A, *B;
  uint32_t la, lb, W;
};

struct TraceConsts {
  const uint16_t *code;
  const int32_t *table;
  int K, open1, ext, gen_eq, gen_ne;
  bool no_start, no_end, no_gaps_a, no_gaps_b;
};

// Moves (x,y,matrix,score) to the predecessor.  Returns 0, or SEQALIGN_E_UNKNOWN_PAIR
// (5) / SEQALIGN_E_TRACEBACK (7).  `acc` supplies the data: code_a(i) / code_b(j) =
// code[seq_a[i]] / code[seq_b[j]], and cell(x, y, m, a, b) = the three matrices at (x, y)
// -- straight from HBM (GlobalAccess) or from a tile staged in LDS (sa_traceback.hip).
template <class Access>
__device__ __forceinline__ uint32_t reverse_move_t(Access &acc, const TraceConsts &k, uint32_t la, uint32_t lb,
                                                   uint32_t &x, uint32_t &y, int &matrix, int &score) {
  // gap costs for leaving (x,y) (alignment.c:261-272)
  long long open_a = k.open1, ext_a = k.ext, open_b = k.open1, ext_b = k.ext;
  if (k.no_end) {
    if (x == la) open_a = ext_a = 0;
    if (y == lb) open_b = ext_b = 0;
  }
  if (k.no_start) {   // x, y >= 1 for every caller; kept for symmetry with the reference
    if (x == 0) open_a = ext_a = 0;
    if (y == 0) open_b = ext_b = 0;
  }
  long long via_m, via_a, via_b;
  uint32_t nx = x, ny = y;   // the predecessor cell (locals: x / y stay in registers for every caller)
  if (matrix == MAT_MATCH) {
    const int code_a = acc.code_a(x - 1), code_b = acc.code_b(y - 1);
    int s = (k.K <= 1) ? ((code_a & 0xff) == (code_b & 0xff) ? k.gen_eq : k.gen_ne)
                       : subst_score<SA_SUBST_GLOBAL>(code_a & 0xff, (code_a >> 8) * k.K, code_b, k.table,
                                                      k.gen_eq, k.gen_ne);
    if (s == SA_S_UNKNOWN) return 5;
    // a blocked pair (no_mismatches, not a match) looks up as score 0 upstream
    // (alignment_scoring.c:148-153 with no wildcard involved)
    if (s == SA_S_BLOCKED) s = 0;
    via_m = via_a = via_b = s; nx = x - 1; ny = y - 1;
  } else if (matrix == MAT_GAP_A) {
    via_m = via_b = open_a; via_a = ext_a; ny = y - 1;
  } else {
    via_m = via_a = open_b; via_b = ext_b; nx = x - 1;
  }
  int mi, ai, bi;
  acc.cell(nx, ny, mi, ai, bi);
  x = nx; y = ny;
  const long long av = ai, bv = bi, mv = mi, cur = score;
  // (all three tests before the priority chain: the three loads behind them go out together instead of one per
  // branch -- a walker is a chain of dependent round trips as it is)
  const bool from_a = (!k.no_gaps_a || nx == 0 || nx == la) & (av + via_a == cur);
  const bool from_b = (!k.no_gaps_b || ny == 0 || ny == lb) & (bv + via_b == cur);
  const bool from_m = mv + via_m == cur;
  if (!(from_a | from_b | from_m)) return 7;
  matrix = from_a ? MAT_GAP_A : from_b ? MAT_GAP_B : MAT_MATCH;
  score = from_a ? ai : from_b ? bi : mi;
  return 0;
}

struct GlobalAccess {
  const PairView &v;
  const uint16_t *code;
  __device__ __forceinline__ int code_a(uint32_t i) const { return code[v.seq_a[i]]; }
  __device__ __forceinline__ int code_b(uint32_t j) const { return code[v.seq_b[j]]; }
  __device__ __forceinline__ void cell(uint32_t x, uint32_t y, int &m, int &a, int &b) const {
    const uint32_t at = y * v.W + x;
    m = v.M[at]; a = v.A[at]; b = v.B[at];
  }
};

__device__ __forceinline__ uint32_t reverse_move(const PairView &v, const TraceConsts &k, uint32_t &x,
                                                 uint32_t &y, int &matrix, int &score) {
  GlobalAccess acc{v, k.code};
  return reverse_move_t(acc, k, v.la, v.lb, x, y, matrix, score);
}

}  // namespace sa
