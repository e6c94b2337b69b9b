// Launch interface of the workgroup-level merge kernel between the translation units of
// libcask_hip.so: cask_hip.hip fills a MergeLaunch from the plan, merge_ipt<N>.hip hold the kernel
// instantiations for items_per_thread = N (compiled in parallel).  Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include "spmv_common.hpp"

namespace caskhip {

struct MergeLaunch {
  int grid, wg_size, lds_bytes;    // lds_bytes already includes the dot epilogue's slice of w when dot.w is set; a window
                                   //   that shares the product area (merge_window_aliased) has no share of its own in it
  int xu;                          // 8-byte window loads per lane: 0 (no tile), 1, 2, 4 or 8
  int remap, n_cols, nnz, maxch;
  bool nontemporal, any_skew;
  const BlockDesc *blocks;
  const int *rp, *ci;
  const unsigned *ci16;            // NULL: 32-bit indices; else 16-bit slots per nonzero, or (packed12) 12-byte records
  bool packed12;                   //   of eight 12-bit slots per thread, [block][thread]
  bool one_window;                 // every tiled block's tile is one contiguous window (paired window loads)
  const int *xchunk;
  const double *val;
  double *partials;                // long-row pieces
  XHalo halo;
  DotEpilogue dot;
  bool solver_pass;                // EXT == 2 launch: `pass` describes the composed operand and the scalars
  SolverPass pass;
};

template <int IPT>
void launch_merge_blocks(const MergeLaunch &l, const double *x, double *y, hipStream_t s);

extern template void launch_merge_blocks<2>(const MergeLaunch &, const double *, double *, hipStream_t);
extern template void launch_merge_blocks<4>(const MergeLaunch &, const double *, double *, hipStream_t);
extern template void launch_merge_blocks<8>(const MergeLaunch &, const double *, double *, hipStream_t);
extern template void launch_merge_blocks<16>(const MergeLaunch &, const double *, double *, hipStream_t);

}  // namespace caskhip
