// Plain data of a launch plan, shared by the device kernels (spmv_common.hpp includes this) and by the host planners
// (plan_host.hpp).  No HIP here: the planners -- index arithmetic and byte images, the part of the engine a sanitizer can
// reach -- compile with plain g++ (`make asan`, tests/cpp/test_planners.cpp).
#pragma once
#include <stdint.h>

namespace caskhip {

// One workgroup's share of the merge path.  32 bytes, read through the scalar
// cache (address depends on blockIdx only).
struct BlockDesc {
  int32_t row_start;   // first row of the block
  int32_t n_rows;      // rows finished by this block (1 for a long-row piece)
  int32_t nnz_start;   // first nonzero
  int32_t nnz_count;   // nonzeros in the block
  int32_t cmin;        // smallest column referenced (x window start)
  int32_t cwidth;      // window width in doubles (0 when the block has no nonzeros)
  int32_t kind_g;      // bits 0-7: lanes per row in the reduce phase; bit 8: long-row piece; bit 9: piece writes a partial
  int32_t aux;         // long-row piece: slot in the partials buffer; other blocks: largest column referenced
};
constexpr int KIND_LONG = 0x100;
constexpr int KIND_PARTIAL = 0x200;
constexpr int KIND_SKEW = 0x800;      // block holds rows much longer than its lanes-per-row suits: second, wave-per-row pass
constexpr int KIND_CONTIG = 0x400;    // tiled block whose chunks are consecutive: chunk c starts at cmin + 64c
constexpr int KIND_HOLES = 0x2000;    // SCAN block that spans rows it has no row end for: rowmap (+ zero fill of its empty rows)
constexpr int KIND_NOFILL = 0x4000;   // ... and those other rows are NOT its own (sub-matrix blocks of a SLICE plan): no zero fill

struct SplitRow {      // a row whose pieces are summed by the fix-up kernel
  int32_t row, first_slot, n_slots, pad;
};

// One workgroup of row-mapped slices (variant SLICE, slice_kernel.hpp): `n_rows` consecutive rows from `row_start`, of which
// the `n_short` with at most K nonzeros are this block's -- sorted by length (longest first, ties in row order) they are
// positions 0 .. n_short-1, and plane j (the j-th nonzero of every row that has one: positions 0 .. cnt[j]-1) is stored
// contiguously behind plane j-1 from nonzero `nnz_start` of the plan's slice arrays.  32 bytes, scalar-cache read.
constexpr int SLICE_KMAX = 8;
struct SliceDesc {
  int32_t row_start, n_rows, nnz_start, n_short;
  uint16_t cnt[SLICE_KMAX];
};
// the slice kernel is instantiated for 2, 4 and 8 planes (K rounds up) with 2, 2 and 1 rows per thread
constexpr int slice_kernel_km(int k) { return k <= 2 ? 2 : k <= 4 ? 4 : 8; }
constexpr int slice_rows_per_thread(int km) { return km <= 4 ? 2 : 1; }
constexpr uint16_t SLICE_NOT_MINE = 0xffff;   // slot-map entry of a row the nonzero-mapped blocks of the launch own

constexpr int SCAN_LDS_BIT = 0x40000000;   // SCAN's own column stream: this reference is a slot of the block's LDS window

// MERGE: does the x window of this kernel shape share the product area's LDS?  (k_spmv_merge; the host plans the
// launch's dynamic LDS with the same rule)
constexpr bool merge_window_aliased(int xu, int ipt) { return xu == 8 && ipt >= 8; }

// Rows longer than SKEW_FACTOR * (lanes per row) products get a whole wave each in a second pass.
constexpr int SKEW_FACTOR = 32;
// MERGE blocks with 1 or 2 lanes per row (hundreds of short rows per block): rows beyond SKEW_SHORT * G products
// leave the first pass already and are summed by 16 lanes (up to SKEW_MED_MAX products) or a wave.
constexpr int SKEW_SHORT = 8;
constexpr int SKEW_MED_MAX = 512;
constexpr int skew_short_max(int g) { return g <= 2 ? SKEW_SHORT * g : SKEW_FACTOR * g; }

constexpr int LONG_PIECE_FACTOR = 16;         // a long-row piece is at most 16*CAP nonzeros

}  // namespace caskhip
