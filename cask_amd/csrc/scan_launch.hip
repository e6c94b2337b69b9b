// k_spmv_scan / k_spmv_slice instantiations (see scan_launch.hpp).
#include "slice_kernel.hpp"

namespace caskhip {

template <int IPT, int XP>
static void launch_scan_ix(const ScanLaunch &l, const double *x, double *y, hipStream_t s) {
  const dim3 grid(l.grid), block(l.wg_size);
  if constexpr (XP > 0 && 2 * XP > IPT + 1) {                // the planner never builds a window the product area cannot hold
    (void)grid; (void)block; (void)x; (void)y; (void)s;
    return;
  } else {
    if constexpr (IPT >= 8) {                                 // padded plans exist for 8 and 16 items per thread (cask_hip.hip build_scan_plan)
      if (l.pad_val) {
        if (l.nontemporal)
          hipLaunchKernelGGL((k_spmv_scan_pad<IPT, true, XP>), grid, block, l.lds_bytes, s, l.blocks, l.n_regular, l.remap, l.n_cols,
                             l.rp, l.ci, l.val, l.pad_ci, l.pad_val, l.meta, l.rowmap, x, y, l.partials);
        else
          hipLaunchKernelGGL((k_spmv_scan_pad<IPT, false, XP>), grid, block, l.lds_bytes, s, l.blocks, l.n_regular, l.remap, l.n_cols,
                             l.rp, l.ci, l.val, l.pad_ci, l.pad_val, l.meta, l.rowmap, x, y, l.partials);
        return;
      }
    }
    if (l.nontemporal)
      hipLaunchKernelGGL((k_spmv_scan<IPT, true, XP>), grid, block, l.lds_bytes, s, l.blocks, l.grid, l.remap, l.nnz, l.n_cols,
                         l.rp, l.ci, l.val, l.meta, l.rowmap, x, y, l.partials);
    else
      hipLaunchKernelGGL((k_spmv_scan<IPT, false, XP>), grid, block, l.lds_bytes, s, l.blocks, l.grid, l.remap, l.nnz, l.n_cols,
                         l.rp, l.ci, l.val, l.meta, l.rowmap, x, y, l.partials);
  }
}

template <int IPT>
static void launch_scan_i(const ScanLaunch &l, const double *x, double *y, hipStream_t s) {
  switch (l.xp) {
    case 0:  launch_scan_ix<IPT, 0>(l, x, y, s); break;
    case 2:  launch_scan_ix<IPT, 2>(l, x, y, s); break;
    case 4:  launch_scan_ix<IPT, 4>(l, x, y, s); break;
    default: launch_scan_ix<IPT, 8>(l, x, y, s); break;
  }
}

// SLICE: items_per_thread 4 or 8 for the long rows' blocks, K rounded up to the kernel's 2 / 4 / 8 planes
template <int KM, int IPT, int XP>
static void launch_slice_kix(const ScanLaunch &l, const double *x, double *y, hipStream_t s) {
  if constexpr (XP > 0 && 2 * XP > IPT + 1) {
    (void)l; (void)x; (void)y; (void)s;
    return;
  } else {
    hipLaunchKernelGGL((k_spmv_slice<KM, IPT, XP>), dim3(l.grid + l.n_slice_blocks), dim3(l.wg_size), l.lds_bytes, s, l.blocks,
                       l.grid, l.remap, l.nnz, l.n_cols, l.rp, l.ci, l.val, l.meta, l.rowmap, l.slices, l.n_slice_blocks,
                       l.slice_val, l.slice_ci, l.slice_slot, x, y, l.partials);
  }
}
template <int KM, int IPT>
static void launch_slice_ki(const ScanLaunch &l, const double *x, double *y, hipStream_t s) {
  switch (l.xp) {
    case 0:  launch_slice_kix<KM, IPT, 0>(l, x, y, s); break;
    case 2:  launch_slice_kix<KM, IPT, 2>(l, x, y, s); break;
    default: launch_slice_kix<KM, IPT, 4>(l, x, y, s); break;
  }
}
template <int KM>
static void launch_slice_k(const ScanLaunch &l, int items_per_thread, const double *x, double *y, hipStream_t s) {
  if (items_per_thread == 4) launch_slice_ki<KM, 4>(l, x, y, s);
  else launch_slice_ki<KM, 8>(l, x, y, s);
}

void launch_scan(const ScanLaunch &l, int items_per_thread, const double *x, double *y, hipStream_t s) {
  if (l.n_slice_blocks > 0) {
    switch (slice_kernel_km(l.slice_k)) {
      case 2:  launch_slice_k<2>(l, items_per_thread, x, y, s); break;
      case 4:  launch_slice_k<4>(l, items_per_thread, x, y, s); break;
      default: launch_slice_k<8>(l, items_per_thread, x, y, s); break;
    }
    return;
  }
  switch (items_per_thread) {
    case 2:  launch_scan_i<2>(l, x, y, s); break;
    case 4:  launch_scan_i<4>(l, x, y, s); break;
    case 8:  launch_scan_i<8>(l, x, y, s); break;
    default: launch_scan_i<16>(l, x, y, s); break;
  }
}

void gather_values(int64_t n, const int *d_idx, const double *d_src, double *d_dst, hipStream_t s) {
  if (n <= 0) return;
  const int grid = (int)std::min<int64_t>(4096, (n + 255) / 256);
  hipLaunchKernelGGL(k_gather_f64, dim3(grid), dim3(256), 0, s, n, d_idx, d_src, d_dst);
}

}  // namespace caskhip
