// k_spmv_merge instantiations for items_per_thread = 8 (see merge_launch.hpp).
#include "merge_launch_impl.hpp"

namespace caskhip {
template void launch_merge_blocks<8>(const MergeLaunch &, const double *, double *, hipStream_t);

template <int XU, bool WIDE>
static void launch_dual(const MergeOperand &a, const MergeOperand &b, int wg_size, int lds_bytes, int remap, hipStream_t s) {
  hipLaunchKernelGGL((k_spmv_merge_dual<8, XU, true, true, true, WIDE, false>), dim3(a.n_blocks + b.n_blocks), dim3(wg_size), lds_bytes, s,
                     a, b, remap);
}

bool launch_merge_dual8(const MergeLaunch &a, const double *xa, double *ya, const MergeLaunch &b, const double *xb, double *yb,
                        hipStream_t s) {
  const bool same_shape = a.wg_size == b.wg_size && a.xu == b.xu && a.remap == b.remap && a.one_window == b.one_window;
  const bool packed = a.ci16 && b.ci16 && a.packed12 && b.packed12;
  if (!same_shape || !packed || a.any_skew || b.any_skew || a.solver_pass || b.solver_pass || a.xu < 2 || a.grid <= 0 || b.grid <= 0)
    return false;
  auto operand = [](const MergeLaunch &l, const double *x, double *y) {
    return MergeOperand{l.blocks, l.grid, l.n_cols, l.nnz, l.maxch, l.rp, l.ci, l.ci16, l.xchunk, l.val, x, y, l.partials, l.halo, l.dot};
  };
  const MergeOperand oa = operand(a, xa, ya), ob = operand(b, xb, yb);
  const int lds = a.lds_bytes > b.lds_bytes ? a.lds_bytes : b.lds_bytes;
  const bool wide = a.one_window;                             // paired window loads: every tile of both plans is one window
  switch (a.xu) {
    case 2:  wide ? launch_dual<2, true>(oa, ob, a.wg_size, lds, a.remap, s) : launch_dual<2, false>(oa, ob, a.wg_size, lds, a.remap, s); break;
    case 4:  wide ? launch_dual<4, true>(oa, ob, a.wg_size, lds, a.remap, s) : launch_dual<4, false>(oa, ob, a.wg_size, lds, a.remap, s); break;
    default: wide ? launch_dual<8, true>(oa, ob, a.wg_size, lds, a.remap, s) : launch_dual<8, false>(oa, ob, a.wg_size, lds, a.remap, s); break;
  }
  return true;
}
}
