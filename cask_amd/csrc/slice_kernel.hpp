// Variant SLICE (round 6): row-mapped slices for the short rows of a power-law matrix, nonzero-mapped (SCAN) blocks for
// its long rows, ONE launch.
//
// webbase-1M and its look-alikes: 65-79 % of the rows hold one nonzero, 92 % at most four -- and those 92 % hold only
// 40 % of the nonzeros.  The SCAN kernel treats all of it alike: every product is parked in LDS, every thread walks a run
// of 8 products guided by a row-end word, carries are joined by a segmented scan, ~680 row sums per block are staged and
// swept out -- four workgroup barriers and three LDS round trips behind the gathers, for rows that mostly ARE one
// product.  Here a row of at most K nonzeros belongs to ONE thread:
//   * the matrix is cut into windows of R * wg_size consecutive rows, a workgroup each; inside a window the short rows are
//     sorted by length (host, plan_host.hpp build_slice_plan) and stored as jagged planes: plane j = the j-th nonzero of
//     every row that has one = positions 0 .. cnt[j]-1 of the sorted order, consecutive in memory, nothing padded;
//   * thread t owns sorted positions t, t + wg_size, ..: its loads of a plane are a wave's consecutive addresses (the
//     address is plane base + position: no row_ptr, no dependent trip), the gathers follow, the <= K products are added
//     in stored order in a register -- no LDS park, no scan;
//   * the sums return to row order through LDS by the window's 16-bit slot map (slot[row] = sorted position, or "not
//     mine" for a long row): one barrier, y leaves in coalesced stores.
// Rows longer than K: scan_block over the plan's compacted copy of them, in the first blocks of the same grid (their
// workgroups live longer: dispatched first), writing their rows through a row map.  Every y entry has exactly one
// writer; every addition's operands are a function of the plan: bitwise reproducible.
//
// Reference: ParallelCsrReadControl.java:119-145,262-276 (several short rows per cycle under a lane mask),
// SpmvKernel.java:68-78 (empty-row skip counts); plan_host.hpp for the format.
#pragma once
#include "scan_kernel.hpp"

namespace caskhip {

template <int KM, int R, bool NT>
__device__ __forceinline__ void slice_block(const SliceDesc &d, const double *__restrict__ sval, const int *__restrict__ sci,
                                            const uint16_t *__restrict__ slot, const double *__restrict__ x,
                                            double *__restrict__ y) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *sums = reinterpret_cast<double *>(smem);            // R * wg_size doubles
  const int WG = blockDim.x, tid = threadIdx.x;
  // Every load below is UNCONDITIONAL with a clamped index -- a lane behind a plane's end re-reads the plane's last
  // element (one request per wave: the lanes share the address) -- and masks come as selects at the very end.  The first
  // version loaded under `if (position < cnt[j])`: the compiler then joins the loaded register with its default at the
  // end of every branch and waits THERE (s_waitcnt vmcnt(1) behind each plane's pair of loads), which made the eight
  // loads of a thread eight serial round trips: 18.2-19.9 us on webbase2 where SCAN takes 15.6.
  // issue order: the planes (the long pole: values + columns, then the dependent gathers), then the slot map
  double v[R][KM];
  int c[R][KM];
  int off = d.nnz_start;
#pragma unroll
  for (int j = 0; j < KM; j++) {
    const int cj = d.cnt[j];                                  // scalar: the descriptor came through the scalar cache
    const int last = off + max(cj - 1, 0);                    // (cj == 0: one element that is some plane's, or the arrays' spare one)
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int e = min(off + r * WG + tid, last);
      v[r][j] = stream_load<NT>(sval + e);
      c[r][j] = stream_load<NT>(sci + e);
    }
    off += cj;
  }
  unsigned sl[R];
#pragma unroll
  for (int r = 0; r < R; r++)
    sl[r] = (unsigned)stream_load<NT>(slot + d.row_start + min(r * WG + tid, d.n_rows - 1));
  double xv[R][KM];
#pragma unroll
  for (int j = 0; j < KM; j++)
#pragma unroll
    for (int r = 0; r < R; r++) xv[r][j] = x[c[r][j]];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int s = r * WG + tid;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < KM; j++) acc = s < (int)d.cnt[j] ? fma(v[r][j], xv[r][j], acc) : acc;   // a select: 0 * NaN never happens
    if (s < d.n_short) sums[s] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int i = r * WG + tid;
    if (i < d.n_rows && sl[r] != (unsigned)SLICE_NOT_MINE) y[d.row_start + i] = sums[sl[r]];
  }
}

// grid = n_scan + n_slice workgroups: hardware blocks [0, n_scan) are the nonzero-mapped blocks of the long rows
// (scan_block over the plan's sub-matrix arrays), the rest the row-mapped slices.
template <int KM, int IPT, int XP>
__global__ void k_spmv_slice(const BlockDesc *__restrict__ blocks, int n_scan, int remap, int nnz_long, int n_cols,
                             const int *__restrict__ rp_unused, const int *__restrict__ lci, const double *__restrict__ lval,
                             const unsigned *__restrict__ meta, const int *__restrict__ rowmap,
                             const SliceDesc *__restrict__ slices, int n_slice, const double *__restrict__ sval,
                             const int *__restrict__ sci, const uint16_t *__restrict__ slot, const double *__restrict__ x,
                             double *__restrict__ y, double *__restrict__ partials) {
  constexpr int R = slice_rows_per_thread(KM);
  if ((int)blockIdx.x < n_scan) {
    scan_block<IPT, true, XP>(blockIdx.x, blocks, n_scan, remap, nnz_long, n_cols, rp_unused, lci, lval, meta, rowmap, x, y,
                              partials);
    return;
  }
  const int lb = logical_block((int)blockIdx.x - n_scan, n_slice, remap);
  const SliceDesc d = slices[lb];
  slice_block<KM, R, true>(d, sval, sci, slot, x, y);
}

// dst[i] = src[idx[i]]: the plan's copies of the value stream in slice / sub-matrix order (once per plan)
__global__ void k_gather_f64(int64_t n, const int *__restrict__ idx, const double *__restrict__ src, double *__restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = idx[i] >= 0 ? src[idx[i]] : 0.0;                   // (a padded stream's spare slots: index -1)
}

}  // namespace caskhip
