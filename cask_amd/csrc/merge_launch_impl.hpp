// Definition of launch_merge_blocks<IPT>: picks the k_spmv_merge instantiation for the plan's window
// shape, load policy and index width.  Included by merge_ipt<N>.hip only.
#pragma once
#include "merge_kernel.hpp"
#include "merge_launch.hpp"

namespace caskhip {

template <int IPT, int XU>
static void launch_merge_ix(const MergeLaunch &l, const double *x, double *y, hipStream_t s) {
  const dim3 grid(l.grid), block(l.wg_size);
#define CASK_LAUNCH_K(NT, C16, C12, WIDE, SKEW, EXT, PASS)                                                        \
  hipLaunchKernelGGL((k_spmv_merge<IPT, XU, NT, C16, C12, WIDE, SKEW, EXT>), grid, block, l.lds_bytes, s,         \
                     l.blocks, l.grid, l.remap, l.n_cols, l.nnz, l.rp, l.ci, l.ci16, l.xchunk, l.maxch, l.val, x, \
                     y, l.partials, l.halo, l.dot, PASS)
  // ordinary products run the lean kernel; halo sources or a dot epilogue select the extended one, a solver
  // pass the one that composes its operand
  const bool ext = l.halo.haddr != nullptr || l.dot.w != nullptr;
  const PassArg<2> pass2{l.pass};
#define CASK_LAUNCH_M(NT, C16, C12, WIDE, SKEW)                                              \
  do {                                                                                       \
    if (l.solver_pass) CASK_LAUNCH_K(NT, C16, C12, WIDE, SKEW, 2, pass2);                    \
    else if (ext)      CASK_LAUNCH_K(NT, C16, C12, WIDE, SKEW, 1, PassArg<1>{});             \
    else               CASK_LAUNCH_K(NT, C16, C12, WIDE, SKEW, 0, PassArg<0>{});             \
  } while (0)
  // plans with skewed blocks exist only with streaming loads (one instantiation less per shape)
  const bool nt = l.nontemporal || l.any_skew;
  constexpr bool TILED = XU > 0;
  constexpr bool CAN12 = TILED && IPT == 8;                   // 12-bit packed slots exist for 8 items per thread
  constexpr bool CANWIDE = CAN12 && XU >= 2;                  // paired window loads: packed plans whose tiles are one window
  if (TILED && l.ci16 && l.packed12 && CAN12 && l.one_window && CANWIDE) {
    if (l.any_skew) CASK_LAUNCH_M(true, TILED, CAN12, CANWIDE, true);
    else if (nt)    CASK_LAUNCH_M(true, TILED, CAN12, CANWIDE, false);
    else            CASK_LAUNCH_M(false, TILED, CAN12, CANWIDE, false);
  } else if (TILED && l.ci16 && l.packed12 && CAN12) {
    if (l.any_skew) CASK_LAUNCH_M(true, TILED, CAN12, false, true);
    else if (nt)    CASK_LAUNCH_M(true, TILED, CAN12, false, false);
    else            CASK_LAUNCH_M(false, TILED, CAN12, false, false);
  } else if (TILED && l.ci16) {
    if (l.any_skew) CASK_LAUNCH_M(true, TILED, false, false, true);
    else if (nt)    CASK_LAUNCH_M(true, TILED, false, false, false);
    else            CASK_LAUNCH_M(false, TILED, false, false, false);
  } else {
    if (l.any_skew) CASK_LAUNCH_M(true, false, false, false, true);
    else if (nt)    CASK_LAUNCH_M(true, false, false, false, false);
    else            CASK_LAUNCH_M(false, false, false, false, false);
  }
#undef CASK_LAUNCH_K
#undef CASK_LAUNCH_M
}

template <int IPT>
void launch_merge_blocks(const MergeLaunch &l, const double *x, double *y, hipStream_t s) {
  switch (l.xu) {
    case 0:  launch_merge_ix<IPT, 0>(l, x, y, s); break;
    case 1:  launch_merge_ix<IPT, 1>(l, x, y, s); break;
    case 2:  launch_merge_ix<IPT, 2>(l, x, y, s); break;
    case 4:  launch_merge_ix<IPT, 4>(l, x, y, s); break;
    default: launch_merge_ix<IPT, 8>(l, x, y, s); break;
  }
}

}  // namespace caskhip
