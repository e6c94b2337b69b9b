// k_spmv_merge_pair instantiations (merge_pair_kernel.hpp) and their launcher.
#include "merge_pair_kernel.hpp"
#include "merge_launch.hpp"

namespace caskhip {

template <int XU>
static void launch_pair_x(const MergeLaunch &l, const double *x, double *y, hipStream_t s) {
  const int half = ((l.grid + 1) / 2 + 7) & ~7;               // multiple of 8: blocks hw and hw + half share an XCD's run
  const dim3 grid(half), block(l.wg_size);
  if (l.one_window && XU >= 2)
    hipLaunchKernelGGL((k_spmv_merge_pair<XU, (XU >= 2)>), grid, block, l.lds_bytes, s, l.blocks, l.grid, half, l.remap, l.n_cols,
                       l.nnz, l.rp, l.ci16, l.xchunk, l.maxch, l.val, x, y);
  else
    hipLaunchKernelGGL((k_spmv_merge_pair<XU, false>), grid, block, l.lds_bytes, s, l.blocks, l.grid, half, l.remap, l.n_cols,
                       l.nnz, l.rp, l.ci16, l.xchunk, l.maxch, l.val, x, y);
}

void launch_merge_pair(const MergeLaunch &l, const double *x, double *y, hipStream_t s) {
  switch (l.xu) {
    case 1:  launch_pair_x<1>(l, x, y, s); break;
    case 2:  launch_pair_x<2>(l, x, y, s); break;
    case 4:  launch_pair_x<4>(l, x, y, s); break;
    default: launch_pair_x<8>(l, x, y, s); break;
  }
}

}  // namespace caskhip
