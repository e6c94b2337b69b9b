// Launch interface of the segmented-scan kernel (variant SCAN) between the translation units of libcask_hip.so:
// cask_hip.hip plans and fills a ScanLaunch, scan_launch.hip holds the kernel instantiations.  Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include "spmv_common.hpp"

namespace caskhip {

constexpr int KIND_HOLES = 0x2000;    // SCAN block that spans empty rows: rowmap + zero fill
constexpr int SCAN_NEEDS = 32;          // producer flags a product block can wait for (more: its far entries stay direct)
constexpr int SCAN_LDS_BIT = 0x40000000;   // plan-owned column stream: this reference is a slot of the block's LDS window

struct ScanLaunch {
  int grid, wg_size, lds_bytes, remap, nnz, n_cols;
  int xp;                          // 16-byte x window loads per thread: 0 (no window), 2, 4 or 8; window = 2 * xp * wg_size
                                   //   entries, parked in the product area (xp <= items_per_thread / 2)
  bool nontemporal;
  const BlockDesc *blocks;
  const int *rp, *ci;              // ci: the caller's columns, or the plan's own stream (window slots / far references);
                                   //   rp is read by KIND_HOLES blocks only (which of their rows are empty)
  const double *val;
  const unsigned *meta;            // [block][thread] row-end words
  const int *rowmap;               // KIND_HOLES blocks: at blocks[b].aux the number of non-empty rows, then their local rows
  double *farx;                    // far plans: x values of the far nonzeros, else NULL
  int *sync;                       // fused far pre-gather: producer flags [far.grid] then block epochs [grid]; else NULL
  const int *needs;                //   and per block the producers it waits for, [grid][SCAN_NEEDS]
  double *partials;                // long-row pieces of split rows
};

constexpr int SCAN_PANELS = 8;
struct ScanPanels {
  int start[SCAN_PANELS + 1];          // panel p owns far entries [start[p], start[p+1])
};

// the far pre-gather (when l.farx) and the product; items_per_thread in {2, 4, 8, 16}
struct ScanFar {
  ScanPanels panels;
  const int *fcol;                 // columns of the far nonzeros, panel-major
  int n_far, grid;                 // grid = 8 * workgroups per panel (0: no far nonzeros)
};
int scan_far_chunk(int wg_size);   // far nonzeros one workgroup of the pre-gather handles
void launch_scan(const ScanLaunch &l, const ScanFar &far, int items_per_thread, const double *x, double *y, hipStream_t s);

}  // namespace caskhip
