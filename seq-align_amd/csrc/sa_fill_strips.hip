// sa_fill_strips.hip -- few, LONG pairs: column strips of one pair run as a
// pipeline of waves.
//
// Replaces alignment_fill_matrices (reference src/alignment.c:28-168) for pairs
// whose rows do not fit one wave; same arithmetic as the other row-sweep kernels
// (sa_rowsweep.hpp).  sa_fill_rowscan.hip walks the 512-column strips of a pair
// one after the other inside ONE wave: a single 10 000 x 10 000 pair keeps 1 of
// the chip's 1024 SIMDs busy for 79 ms (1.3 GCUPS).  The only dependency between
// strip s and strip s-1 is the boundary column (match/gap_a/gap_b of the last
// column of s-1, row by row), so here every (pair, strip) is its own wave:
//
//   * strip s computes rows in chunks of 64; before a chunk it waits until strip
//     s-1 has published that it is done with those rows (one uint32 per strip in
//     HBM, agent-scope release/acquire), then reads the 64 boundary cells back
//     from the matrices -- exactly what the rowscan kernel's RowFeed does;
//   * strip s therefore runs 64 rows behind s-1: a pair with S strips takes
//     len_b + 64*S row steps instead of len_b * S.
//
// Workgroup = one wave.  A workgroup does NOT take its (pair, strip) from blockIdx:
// it draws a TICKET from an atomic counter when it starts running, and ticket =
// (group of 8 pairs, strip, pair in group).  A strip's ticket is therefore always
// higher than the ticket of the strip it waits for, and a ticket only exists once
// its workgroup is resident on a CU -- so a waiting wave only ever waits for waves
// that are running or finished, whatever order the hardware dispatches workgroups
// in (other contexts' kernels, CU masks, preemption).  No cooperative launch, no
// dispatch-order assumption, no watchdog.  Tickets are drawn roughly in dispatch
// order, so one pair's strips (8 tickets apart) still tend to land on one XCD.
#include "sa_rowsweep.hpp"

namespace sa {

constexpr int kStripCPL = 8;                       // 512 columns per strip (256-column strips measured: no faster)
constexpr uint32_t kStripCols = kWave * kStripCPL;

template <int SUBST, bool GENERAL>
__global__ void __launch_bounds__(kWave)
fill_strips_kernel(const SaFillParams p, uint32_t *progress, const uint32_t strips_per_pair) {
  constexpr int CPL = kStripCPL;
  extern __shared__ __attribute__((aligned(16))) int32_t lds_table[];
  const int32_t *table = p.table;
  if constexpr (SUBST == SA_SUBST_LDS) {
    for (uint32_t k = threadIdx.x; k < p.K * p.K; k += blockDim.x) lds_table[k] = p.table[k];
    __syncthreads();
    table = lds_table;
  }

  const int lane = threadIdx.x;
  // ticket = (group * strips_per_pair + strip) * 8 + pair_in_group; the counter sits behind the progress words
  uint32_t ticket = 0;
  if (lane == 0) ticket = atomicAdd(progress + (uint64_t)gridDim.x, 1u);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  const uint32_t in_group = ticket & 7u, gs = ticket >> 3;
  const uint32_t strip = gs % strips_per_pair, pair = (gs / strips_per_pair) * 8 + in_group;
  if (pair >= p.n_pairs) return;

  const uint32_t la = p.len_a[pair], lb = p.len_b[pair];
  const uint32_t i0 = strip * kStripCols;
  uint32_t *done = progress + (uint64_t)pair * strips_per_pair;   // done[s] = rows strip s has finished (+1)
  if (i0 >= la && !(strip == 0)) return;                           // this pair has fewer strips
  const uint8_t *__restrict__ sa_ = p.arena + p.off_a[pair];
  const uint8_t *__restrict__ sb_ = p.arena + p.off_b[pair];
  const uint64_t mo = p.mat_off[pair];
  int32_t *__restrict__ Mg = p.M + mo;
  int32_t *__restrict__ Ag = p.A + mo;
  int32_t *__restrict__ Bg = p.B + mo;
  const uint32_t W = la + 1;

  const SweepConsts k(p, table);
  const Border bd{p.floor, p.gap_open, p.ext, (p.flags & SA_F_IS_SW) != 0,
                  (p.flags & SA_F_NO_START_GAP) != 0};

  // ---- borders (reference alignment.c:46-81): row 0 over my columns; column 0 by strip 0
  const uint32_t cols = (i0 < la) ? min(kStripCols, la - i0) : 0;
  for (uint32_t i = i0 + 1 + lane; i <= i0 + cols; i += kWave) {
    Mg[i] = k.floor_;
    Ag[i] = k.floor_;
    Bg[i] = bd.edge_gap(i);
  }
  if (strip == 0) {
    if (lane == 0) { Mg[0] = 0; Ag[0] = 0; Bg[0] = 0; }
    for (uint32_t j = 1 + lane; j <= lb; j += kWave) {
      const size_t c = (size_t)j * W;
      Mg[c] = k.floor_;
      Ag[c] = bd.edge_gap(j);
      Bg[c] = k.floor_;
    }
  }

  unsigned long long err = ~0ull;
  if (cols) {
    RowSweep<CPL, SUBST, GENERAL> sw;
    const uint32_t col0 = i0 + lane * CPL;
    const int ncol = max(0, min(CPL, (int)cols - lane * CPL));
    sw.start_strip(p, k, bd, sa_, la, i0, col0, lane);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);   // seq_a codes landed (see RowFeed::load)
    const bool last_strip = i0 + kStripCols >= la;

    RowFeed feed;
    uint32_t off = W + col0 + 1;            // (row 1, my first column)
    for (uint32_t j = 1; j <= lb; ++j, off += W) {
      const int q = (j - 1) & (kWave - 1);
      if (q == 0) {
        if (strip > 0) {   // rows j .. j+63 of the strip to my left must be in memory
          const uint32_t need = min(j + kWave - 1, lb) + 1;
          // (the strip to my left holds a lower ticket: it is resident or done, see the header)
          while (__hip_atomic_load(done + strip - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need)
            __builtin_amdgcn_s_sleep(8);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the boundary loads below see those rows
        }
        feed.load(p, k, bd, sb_, lb, W, i0, Mg, Ag, Bg, j + lane);
      }
      int mv[CPL], av[CPL], bv[CPL];
      sw.row(k, j, lb, la, W, lane, col0, ncol, read_lane(feed.code, q), read_lane(feed.Z, q),
             read_lane(feed.B, q), mv, av, bv);
      if (ncol == CPL) {
        store_run<CPL, true>(Mg + off, mv);
        store_run<CPL, true>(Ag + off, av);
        store_run<CPL, true>(Bg + off, bv);
      } else if (ncol > 0) {
        store_partial<CPL>(Mg + off, mv, ncol);
        store_partial<CPL>(Ag + off, av, ncol);
        store_partial<CPL>(Bg + off, bv, ncol);
      }
      if (!last_strip && (q == kWave - 1 || j == lb)) {   // publish: rows <= j are written
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(done + strip, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    err = sw.reduce_err();
  }
  // status[pair]: the lowest failing cell index over the strips (~0 = none)
  if (lane == 0 && err != ~0ull) atomicMin(reinterpret_cast<unsigned long long *>(p.status + pair), err);
}

__global__ void __launch_bounds__(256) strips_init_kernel(uint32_t *progress, uint64_t n_progress, uint64_t *status,
                                                          uint32_t n_pairs) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_progress) progress[i] = 0;   // the progress words + the ticket counter behind them
  if (i < n_pairs) status[i] = ~0ull;
}

}  // namespace sa

uint32_t sa_fill_strips_per_pair(uint32_t max_len_a) {
  return max_len_a ? (max_len_a + sa::kStripCols - 1) / sa::kStripCols : 1;
}

// progress: 8 * ceil(n_pairs / 8) * sa_fill_strips_per_pair(max_len_a) + 1 uint32 of scratch
hipError_t sa_launch_fill_strips(const SaFillParams &p, uint32_t max_len_a, uint32_t *progress, hipStream_t stream) {
  if (p.n_pairs == 0) return hipSuccess;
  const uint32_t spp = sa_fill_strips_per_pair(max_len_a);
  const uint64_t groups = (p.n_pairs + 7) / 8, blocks = groups * spp * 8;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const uint64_t n_progress = blocks;   // the ticket counter is progress[gridDim.x]
  const uint64_t init_n = n_progress + 1 > p.n_pairs ? n_progress + 1 : p.n_pairs;
  hipLaunchKernelGGL(sa::strips_init_kernel, dim3((unsigned)((init_n + 255) / 256)), dim3(256), 0, stream,
                     progress, n_progress, p.status, p.n_pairs);
  const dim3 grid((unsigned)blocks), block(sa::kWave);
  const bool general = sa::needs_general(p);
  using namespace sa;
  if (p.K <= 1) {
    if (general) hipLaunchKernelGGL((fill_strips_kernel<SA_SUBST_SIMPLE, true>), grid, block, 0, stream, p, progress, spp);
    else hipLaunchKernelGGL((fill_strips_kernel<SA_SUBST_SIMPLE, false>), grid, block, 0, stream, p, progress, spp);
  } else if (p.K <= SA_LDS_TABLE_MAX_K) {
    const size_t lds = (size_t)p.K * p.K * sizeof(int32_t);
    if (general) hipLaunchKernelGGL((fill_strips_kernel<SA_SUBST_LDS, true>), grid, block, lds, stream, p, progress, spp);
    else hipLaunchKernelGGL((fill_strips_kernel<SA_SUBST_LDS, false>), grid, block, lds, stream, p, progress, spp);
  } else {
    hipLaunchKernelGGL((fill_strips_kernel<SA_SUBST_GLOBAL, true>), grid, block, 0, stream, p, progress, spp);
  }
  return hipGetLastError();
}
