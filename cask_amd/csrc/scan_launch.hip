// k_spmv_scan / k_far_panels instantiations (see scan_launch.hpp).
#include "scan_kernel.hpp"

namespace caskhip {

constexpr int FAR_U = SCAN_FAR_U;       // far nonzeros per lane of k_far_panels and of the fused producers

template <int IPT, int XP>
static void launch_scan_ix(const ScanLaunch &l, const ScanFar &far, const double *x, double *y, hipStream_t s) {
  const bool fused = l.farx && l.sync;
  const dim3 grid(l.grid + (fused ? far.grid : 0)), block(l.wg_size);
  if constexpr (XP > 0 && 2 * XP > IPT + 1) {                // the planner never builds a window the product area cannot hold
    (void)grid; (void)block; (void)x; (void)y; (void)s; (void)far;
    return;
  } else {
#define CASK_LAUNCH_S(NT, FARX)                                                                                      \
  hipLaunchKernelGGL((k_spmv_scan<IPT, NT, FARX, XP>), grid, block, l.lds_bytes, s, l.blocks, l.grid, l.remap, l.nnz, \
                     l.n_cols, l.rp, l.ci, l.val, l.meta, l.rowmap, x, l.farx, y, l.partials, far, ScanSync{l.sync, l.sync ? l.sync + far.grid : nullptr, l.needs})
    if (fused)       { if (l.nontemporal) CASK_LAUNCH_S(true, 2); else CASK_LAUNCH_S(false, 2); }
    else if (l.farx) { if (l.nontemporal) CASK_LAUNCH_S(true, 1); else CASK_LAUNCH_S(false, 1); }
    else             { if (l.nontemporal) CASK_LAUNCH_S(true, 0); else CASK_LAUNCH_S(false, 0); }
#undef CASK_LAUNCH_S
  }
}

template <int IPT>
static void launch_scan_i(const ScanLaunch &l, const ScanFar &far, const double *x, double *y, hipStream_t s) {
  switch (l.xp) {
    case 0:  launch_scan_ix<IPT, 0>(l, far, x, y, s); break;
    case 2:  launch_scan_ix<IPT, 2>(l, far, x, y, s); break;
    case 4:  launch_scan_ix<IPT, 4>(l, far, x, y, s); break;
    default: launch_scan_ix<IPT, 8>(l, far, x, y, s); break;
  }
}

int scan_far_chunk(int wg_size) { return FAR_U * wg_size; }

void launch_scan(const ScanLaunch &l, const ScanFar &far, int items_per_thread, const double *x, double *y, hipStream_t s) {
  if (l.farx && !l.sync && far.grid > 0)
    hipLaunchKernelGGL((k_far_panels<FAR_U>), dim3(far.grid), dim3(l.wg_size), 0, s, far.panels, far.fcol, x, l.farx);
  switch (items_per_thread) {
    case 2:  launch_scan_i<2>(l, far, x, y, s); break;
    case 4:  launch_scan_i<4>(l, far, x, y, s); break;
    case 8:  launch_scan_i<8>(l, far, x, y, s); break;
    default: launch_scan_i<16>(l, far, x, y, s); break;
  }
}

}  // namespace caskhip
