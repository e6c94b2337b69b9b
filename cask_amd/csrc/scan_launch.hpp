// Launch interface of the segmented-scan kernel (variant SCAN) and of the row-mapped slices (variant SLICE, whose long
// rows are SCAN blocks of the same launch) between the translation units of libcask_hip.so: cask_hip.hip plans and fills
// a ScanLaunch, scan_launch.hip holds the kernel instantiations.  Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include "spmv_common.hpp"

namespace caskhip {

struct ScanLaunch {
  int grid, wg_size, lds_bytes, remap, nnz, n_cols;
  int xp;                          // 16-byte x window loads per thread: 0 (no window), 2, 4 or 8; window = 2 * xp * wg_size
                                   //   entries, parked in the product area (xp <= items_per_thread / 2)
  bool nontemporal;
  const BlockDesc *blocks;
  const int *rp, *ci;              // ci: the caller's columns, or the plan's own stream (window slots);
                                   //   rp is read by KIND_HOLES blocks only (which of their rows are empty)
  const double *val;
  const unsigned *meta;            // [block][thread] row-end words
  const int *rowmap;               // KIND_HOLES blocks: at blocks[b].aux the number of non-empty rows, then their local rows
  double *partials;                // long-row pieces of split rows
  // padded plan (r6, scan_kernel.hpp scan_block_padded): the first n_regular blocks stream the plan's padded copies
  // (block b at b * wg_size * items_per_thread), the long-row pieces behind them the arrays above; NULL = the unpadded plan
  const int *pad_ci;
  const double *pad_val;
  int n_regular;
  // SLICE (n_slice_blocks > 0): `grid` nonzero-mapped blocks over the plan's copy of the long rows (blocks / ci / val /
  // meta / rowmap / nnz above describe THAT sub-matrix), then n_slice_blocks row-mapped blocks
  int n_slice_blocks, slice_k;
  const SliceDesc *slices;
  const double *slice_val;
  const int *slice_ci;
  const uint16_t *slice_slot;      // [n_rows]: a row's position among its block's short rows, or SLICE_NOT_MINE
};

// items_per_thread in {2, 4, 8, 16} (SLICE: 4, 8)
void launch_scan(const ScanLaunch &l, int items_per_thread, const double *x, double *y, hipStream_t s);
// d_dst[i] = d_src[d_idx[i]] (a SLICE plan's copies of the value stream, once per plan)
void gather_values(int64_t n, const int *d_idx, const double *d_src, double *d_dst, hipStream_t s);

}  // namespace caskhip
