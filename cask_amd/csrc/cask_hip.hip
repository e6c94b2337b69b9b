// libcask_hip.so -- the C ABI of include/cask_hip.h: device-resident CSR, launch
// planning (merge-path cuts, x-window spans), kernel dispatch, measured DSE,
// BLAS-1 and the CG/BiCG drivers.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "blas1_kernels.hpp"
#include "cask_hip.h"
#include "host_copy.hpp"
#include "cask_hip_p2p.h"
#include "merge_launch.hpp"
#include "plan_host.hpp"
#include "scan_launch.hpp"
#include "spmv_kernels.hpp"
#ifdef CASK_UNITY   // single-translation-unit build (diagnostic builds: tools/stamps.py)
#include "scan_launch.hip"
#include "merge_launch_impl.hpp"
namespace caskhip {
template void launch_merge_blocks<2>(const MergeLaunch &, const double *, double *, hipStream_t);
template void launch_merge_blocks<4>(const MergeLaunch &, const double *, double *, hipStream_t);
template void launch_merge_blocks<8>(const MergeLaunch &, const double *, double *, hipStream_t);
template void launch_merge_blocks<16>(const MergeLaunch &, const double *, double *, hipStream_t);
}
#endif

#include "internal.hpp"

using namespace caskhip;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
}  // namespace

int caskhip::report_failure(int code, const std::string &msg) { return fail(code, msg); }

namespace {

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(CASK_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

using plan::is_pow2;
using plan::pow2_floor;

constexpr int MAX_LDS_BYTES = 64 * 1024;      // per workgroup; keeps >= 2 workgroups per CU (160 KiB LDS)
constexpr int DEFAULT_TILE = 4096;            // doubles (32 KiB)

struct Plan {
  cask_hip_params prm{};          // resolved
  int grid = 0;
  int lds_bytes = 0;
  bool ldsx = false;
  int xu = 0;                      // MERGE: 8-byte window loads per lane (window capacity = xu*wg_size)
  // MERGE
  DevBuf<BlockDesc> blocks;
  DevBuf<BlockDesc> long_blocks;   // pipelined plan: long-row pieces get their own launch
  int n_blocks = 0, n_long_blocks = 0;
  DevBuf<SplitRow> split_rows;
  DevBuf<double> partials;
  DevBuf<unsigned short> ci16;     // MERGE with an x tile: LDS slot of each nonzero's column (2 B/nnz), or
  bool packed12 = false;           //   12-byte records of eight 12-bit slots per thread, [block][thread] (1.5 B/nnz)
  double slot_bytes_per_nnz = 0.0; //   what the slot stream costs (reported)
  DevBuf<int> xchunk;              // MERGE with ci16: first column of each 16-column tile chunk, maxch per block, [t >> 4][u] (plan::build_chunk_table)
  bool one_window = false;         // every tiled block's chunks are consecutive (KIND_CONTIG): paired window loads
  int maxch = 0;
  bool any_skew = false;           // MERGE: some block is flagged KIND_SKEW (selects the kernel with the second pass)
  int n_long_rows = 0, n_split_rows = 0;
  DevBuf<double> dot_part;         // MERGE: per-block (+ per split row) shares of the fused w.y
  // VECTOR
  DevBuf<int2v> xspan;
  int vec_long_rows = 0;           // VECTOR: rows longer than this go to the long-row pieces (long_blocks); 0 = none do
  bool vec_long_wave = false;      //   ... which a wave each sums (short pieces) instead of a workgroup of 256
  // SCAN (scan_kernel.hpp): per-thread row-end words, the row map of blocks that span empty rows, and -- with an x window --
  // the plan's own column stream (LDS slots)
  DevBuf<unsigned> scan_meta;
  DevBuf<int> scan_rowmap, scan_ci;
  // padded SCAN plan (r6): the plan's own copies of the streams, block b at b * wg_size * items_per_thread; the first
  // n_regular blocks of `blocks` are the nonzero-mapped ones, the long-row pieces follow
  DevBuf<double> scan_pval;
  DevBuf<int> scan_pci;
  int n_regular = 0;
  // SLICE (slice_kernel.hpp): the row-mapped slices (descriptors, slot map, the plan's copies of the short rows' values
  // and columns in plane order) and the long rows' sub-matrix (its values; blocks / scan_meta / scan_rowmap / scan_ci above
  // describe it)
  DevBuf<SliceDesc> slices;
  DevBuf<uint16_t> slice_slot;
  DevBuf<double> slice_val, long_val;
  DevBuf<int> slice_ci;
  int n_slice_blocks = 0, slice_k = 0;
  int64_t long_nnz = 0;
};

// Buffers of a solve, kept on the handle between solves of the same shape (a solver called in a loop -- or timed
// over a few passes -- must not pay a dozen allocations per call).
// A solver's checkpoint without a gap in the GPU's work (r6).  Every 16 passes the host wants the two flags; copying them and
// waiting for the stream left the GPU idle from the copy's completion until the next pass was launched (~15 us per
// checkpoint: 0.9 us per pass).  Deferred: the flags go to PINNED memory behind an event, the host launches ONE more pass and
// only then waits for that event -- the GPU is busy with the pass while the copy's completion travels (waiting behind the
// pass's product alone -- 8-20 us -- covered it on the large systems only).  A converged solve has launched one pass more:
// its product(s) overwrite q, its update launches return on the done flag (blas1_kernels.hpp).
struct DeferredFlags {
  int *host = nullptr;                                        // pinned: [0] done, [1] iterations
  DevEvent arrived;
  int pending = 0;                                            // the number of passes launched at the checkpoint in flight; 0: none
  ~DeferredFlags() {
    if (host) (void)hipHostFree(host);
  }
  hipError_t create() {
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&host), 2 * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) { host = nullptr; return e; }
    host[0] = host[1] = 0;
    return arrived.create();
  }
  hipError_t request(const int *d_flags, int launched, hipStream_t s) {
    hipError_t e = hipMemcpyAsync(host, d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipEventRecord(arrived, s);
    if (e == hipSuccess) pending = launched;
    return e;
  }
  hipError_t wait(int h_flags[2]) {
    hipError_t e = hipEventSynchronize(arrived);
    h_flags[0] = host[0];
    h_flags[1] = host[1];
    pending = 0;
    return e;
  }
};

struct SolverWorkspace {
  int64_t n = -1, S = 0, n_full = 0;
  int n_slots = 0;
  bool shared_vec = false;
  DevBuf<double> own_vec, q, qt, part_a, part_b, part_c, scal, x_full, x_full_t;
  DevBuf<int> flags;
};

// Pinned, GPU-mapped host memory (the staging buffers of the host-vector entry point).
constexpr int HOST_DONE_DEFAULT = 1;   // 0: hipStreamSynchronize, 1: the completion word (cask_hip_spmv: -4.5 us a call, profiles/r06_host_entry.txt)

struct PinBuf {
  double *p = nullptr, *dev = nullptr;   // host address / the address the GPU uses for it
  size_t n = 0;
  ~PinBuf() { release(); }
  void release() {
    if (p) (void)hipHostFree(p);
    p = dev = nullptr;
    n = 0;
  }
  hipError_t ensure(size_t count) {
    if (p && n == count) return hipSuccess;
    release();
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(double), hipHostMallocMapped);
    if (e != hipSuccess) { p = nullptr; return e; }
    e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), p, 0);
    if (e != hipSuccess) { release(); return e; }
    n = count;
    return hipSuccess;
  }
};

}  // namespace

struct cask_hip_matrix {
  int32_t n_rows = 0, n_cols = 0;
  int64_t nnz = 0;
  int device = 0;
  const int *d_rp = nullptr, *d_ci = nullptr;
  const double *d_val = nullptr;
  DevBuf<int> own_rp, own_ci;
  DevBuf<double> own_val;
  std::vector<int> h_rp;           // row_ptr kept on the host for (re)planning
  std::vector<int> h_ci;           // col_ind on the host (filled on first use) for the x-tile planner
  int max_row = 0, empty_rows = 0;
  cask_hip_params requested{};
  Plan plan;
  // row-sharded product (cask_hip_p2p.h): columns >= halo_n_own are read from the addresses in halo_addr
  int halo_n_own = std::numeric_limits<int>::max();
  const uint64_t *halo_addr = nullptr;
  int64_t halo_shift = 0;          // bytes added to every halo address (set per launch by the sharded solvers)
  hipStream_t stream = nullptr;    // for the host-vector entry points and timing
  DevBuf<double> d_x, d_y;         // staging for cask_hip_spmv
  PinBuf pin_x, pin_y;             // ... and its pinned host side (the staged entry: host_entry below)
  PinBuf pin_done;                 // a word the GPU stores behind a product of the host entry: the CPU polls it instead of waiting for a signal
  uint64_t done_seq = 0;
  std::unique_ptr<cask_hip_matrix> transpose;
  std::unique_ptr<SolverWorkspace> solver_ws;
  std::unique_ptr<DeferredFlags> checkpoint;   // (created by the first solve that defers: a pinned allocation is ~100 us)
  int solver_load_choice = 0;      // what SolverLoadPolicy measured on this handle's last trial: 1 streaming, -1 cached loads, 0 not yet
  ~cask_hip_matrix() {
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

// ------------------------------------------------------------------ planning
int resolve_params(const cask_hip_matrix &m, const cask_hip_params &in, cask_hip_params &out) {
  out = in;
  const double mean = m.n_rows ? (double)m.nnz / m.n_rows : 0.0;
  if (out.variant == CASK_HIP_VARIANT_AUTO) {
    // Measured on all four BASELINE families (profiles/r01_dse_out.json, r03_dse_out.json): the workgroup-level merge
    // kernel wins on three; short, heavily skewed rows (webbase-1M-like: mean 3, longest 4 700) are the segmented-scan
    // kernel's (21.6 against 24.1 us).  The DSE (cask_hip_tune / cask_amd.dse) refines the choice per matrix.
    out.variant = CASK_HIP_VARIANT_MERGE;
    if (!m.halo_addr && mean < 4.0 && m.max_row > 256 && m.nnz >= (1 << 20)) out.variant = CASK_HIP_VARIANT_SCAN;
  }
  if (out.variant == CASK_HIP_VARIANT_MERGE_PAIR_REMOVED || out.xcd_remap == 2)
    return fail(CASK_HIP_ERR_INVALID, "variant MERGE_PAIR (xcd_remap = 2) was removed in ABI 6: a measured loss on every "
                                      "BASELINE family but one (docs/experiments.md); use CASK_HIP_VARIANT_MERGE");
  if (out.wg_size == 0) out.wg_size = 256;
  if (!(out.wg_size == 64 || out.wg_size == 128 || out.wg_size == 256 || out.wg_size == 512 ||
        out.wg_size == 1024))
    return fail(CASK_HIP_ERR_INVALID, "wg_size must be 64, 128, 256, 512 or 1024");
  if (out.variant != CASK_HIP_VARIANT_VECTOR && out.variant != CASK_HIP_VARIANT_MERGE &&
      out.variant != CASK_HIP_VARIANT_MERGE_WAVE && out.variant != CASK_HIP_VARIANT_SCAN && out.variant != CASK_HIP_VARIANT_SLICE)
    return fail(CASK_HIP_ERR_INVALID, "unknown variant");
  if (m.halo_addr && out.variant != CASK_HIP_VARIANT_MERGE)
    return fail(CASK_HIP_ERR_INVALID, "a matrix with halo sources runs the MERGE variant only");
  if (out.far_columns >= 1)
    return fail(CASK_HIP_ERR_INVALID, "far_columns = 1 / 2 were removed (MERGE's far slots in ABI 6, SCAN's column-panel pre-gather "
                                      "in ABI 7): the pre-gather cost more time than the line fills it saved, docs/experiments.md; "
                                      "use 0 / -1");
  if (m.nnz < 2 && !m.halo_addr) {                                          // the merge kernels stream 16-byte pairs
    if (out.variant == CASK_HIP_VARIANT_SLICE) out.lanes_per_row = 0;       // (it carried K, not lanes)
    out.variant = CASK_HIP_VARIANT_VECTOR;
  }
  if (out.variant == CASK_HIP_VARIANT_SLICE) {
    // K, the longest row a slice thread takes, travels in lanes_per_row; the long rows' blocks are SCAN blocks
    if (out.lanes_per_row == 0) out.lanes_per_row = 4;
    if (out.lanes_per_row < 1 || out.lanes_per_row > SLICE_KMAX)
      return fail(CASK_HIP_ERR_INVALID, "variant SLICE: lanes_per_row carries K, the longest row a slice thread takes: 1..8");
    if (out.items_per_thread == 0) out.items_per_thread = 8;
    if (out.items_per_thread != 4 && out.items_per_thread != 8)
      return fail(CASK_HIP_ERR_INVALID, "variant SLICE: items_per_thread (of the long rows' blocks) must be 4 or 8");
    if (m.n_cols >= SCAN_LDS_BIT) return fail(CASK_HIP_ERR_INVALID, "variant SLICE: too many columns");
  } else if (out.lanes_per_row == 0) {
    int l = pow2_floor(std::max(1, (int)std::lround(mean / 4.0)));
    out.lanes_per_row = std::min(64, std::max(out.variant == CASK_HIP_VARIANT_VECTOR ? 2 : 1, l));
  }
  if (out.variant != CASK_HIP_VARIANT_SLICE && (!is_pow2(out.lanes_per_row) || out.lanes_per_row > 64))
    return fail(CASK_HIP_ERR_INVALID, "lanes_per_row must be a power of two in 1..64");
  if (out.items_per_thread == 0) out.items_per_thread = 8;
  if (!(out.items_per_thread == 2 || out.items_per_thread == 4 || out.items_per_thread == 8 ||
        out.items_per_thread == 16))
    return fail(CASK_HIP_ERR_INVALID, "items_per_thread must be 2, 4, 8 or 16");
  if (out.tile_width == 0) out.tile_width = DEFAULT_TILE;
  if (out.tile_width < -1) return fail(CASK_HIP_ERR_INVALID, "tile_width must be -1 (off), 0 (default) or > 0");
  if (out.xcd_remap == 0) out.xcd_remap = 1;
  if (out.nontemporal == 0) out.nontemporal = 1;
  if (out.index16 == 0) out.index16 = 1;
  if (out.index16 == 3 || out.index16 == 4)
    return fail(CASK_HIP_ERR_INVALID, "index16 = 3 / 4 (run records and their A/B twin) were removed in ABI 6: the decode cost "
                                      "more than the bytes bought (docs/experiments.md); use 0 / 1 (12-bit packed slots)");
  if (out.index16 > 2) return fail(CASK_HIP_ERR_INVALID, "index16 must be -1 (off), 0/1 (compressed) or 2 (16-bit only)");
  if (out.far_columns < -1) return fail(CASK_HIP_ERR_INVALID, "far_columns must be -1 or 0");
  if (out.variant != CASK_HIP_VARIANT_MERGE) out.index16 = -1;
  if (out.variant == CASK_HIP_VARIANT_MERGE || out.variant == CASK_HIP_VARIANT_SCAN || out.variant == CASK_HIP_VARIANT_SLICE) {
    const long cap = (long)out.wg_size * out.items_per_thread;
    if (8 * (cap + 2) + 8 * out.wg_size > MAX_LDS_BYTES)
      return fail(CASK_HIP_ERR_INVALID, "wg_size*items_per_thread needs more than 64 KiB of LDS");
  }
  return CASK_HIP_OK;
}

// The planners themselves -- merge-path cuts, chunk tiles, packed slots, seam placement, SCAN words / windows, SLICE --
// are host-only code on plain arrays: plan_host.hpp (compiled and run under ASan / UBSan by `make asan`).
int ensure_host_col_ind(cask_hip_matrix &m) {
  if (m.h_ci.size() == (size_t)m.nnz) return CASK_HIP_OK;
  m.h_ci.resize((size_t)m.nnz);
  if (m.nnz) HIP_TRY(hipMemcpy(m.h_ci.data(), m.d_ci, (size_t)m.nnz * sizeof(int), hipMemcpyDeviceToHost));
  for (int64_t k = 0; k < m.nnz; k++)
    if (m.h_ci[k] < 0 || m.h_ci[k] >= m.n_cols) {
      m.h_ci.clear();
      return fail(CASK_HIP_ERR_INVALID, "column index out of range");
    }
  return CASK_HIP_OK;
}

template <int IPT>
const void *merge_wave_fn(bool nt) {
  return nt ? reinterpret_cast<const void *>(&k_spmv_merge_wave<IPT, true>)
            : reinterpret_cast<const void *>(&k_spmv_merge_wave<IPT, false>);
}
const void *merge_wave_fn(int ipt, bool nt) {
  switch (ipt) {
    case 2: return merge_wave_fn<2>(nt);
    case 4: return merge_wave_fn<4>(nt);
    case 8: return merge_wave_fn<8>(nt);
    default: return merge_wave_fn<16>(nt);
  }
}

// SCAN plan (scan_kernel.hpp): blocks of at most cap nonzeros snapped to rows; one word per thread with the row
// ends of its run; tile_width = an x window per block (the plan's own column stream then).
int build_scan_plan(cask_hip_matrix &m, const cask_hip_params &prm) {
  Plan &pl = m.plan;
  const int wg = prm.wg_size, ipt = prm.items_per_thread, cap = wg * ipt;
  std::vector<BlockDesc> blocks;
  std::vector<SplitRow> splits;
  int n_long = 0, n_slots = 0;
  // cap - 1 nonzeros per block: a block that starts on an odd nonzero streams one foreign element in front of its own
  // (16-byte loads), and the workgroup loads exactly cap elements
  plan::build_merge_blocks(m.h_rp.data(), m.n_rows, cap - 1, 1 << 30, (long)cap * LONG_PIECE_FACTOR, wg, blocks, nullptr, splits,
                           n_long, n_slots, false);
  // padded plan (scan_kernel.hpp scan_block_padded): the stream addresses of a block are a function of its index -- the
  // descriptor trip leaves the front of every workgroup's life.  CASK_HIP_SCAN_PAD=0: the unpadded plan (the A/B).
  // Measured (profiles/r06_scan_pad.txt): with 8 items per thread and an x window 15.20-15.27 -> 14.72-14.78 us on webbase2,
  // 21.11-21.18 -> 20.90-20.95 on webbase-1M-like; without a window or with 4 items per thread it LOSES 2-6 %: padded there
  // only.  CASK_HIP_SCAN_PAD=0: never (the A/B).
  static const bool pad_off = std::getenv("CASK_HIP_SCAN_PAD") && std::atoi(std::getenv("CASK_HIP_SCAN_PAD")) == 0;
  const bool windowed = m.n_cols < SCAN_LDS_BIT && plan::scan_window_xp(prm.tile_width > 0 ? prm.tile_width : 0, wg, ipt) > 0;
  const bool pad = !pad_off && ipt >= 8 && windowed && m.nnz > 0 && (int64_t)blocks.size() * cap < ((int64_t)1 << 31);
  int n_regular = (int)blocks.size();
  if (pad) n_regular = plan::regular_blocks_first(blocks);
  pl.n_long_rows = n_long;
  pl.n_split_rows = (int)splits.size();
  pl.grid = pl.n_blocks = (int)blocks.size();
  std::vector<unsigned> meta;
  std::vector<int> rowmap;
  plan::build_scan_meta(m.h_rp.data(), blocks, wg, ipt, meta, rowmap);
  HIP_TRY(pl.scan_meta.upload(meta));
  rowmap.push_back(0);                                        // never empty: the kernel forms rowmap + aux
  HIP_TRY(pl.scan_rowmap.upload(rowmap));
  if (!splits.empty()) {
    HIP_TRY(pl.split_rows.upload(splits));
    HIP_TRY(pl.partials.alloc(n_slots));
  }
  pl.prm.lanes_per_row = 0;
  pl.prm.index16 = -1;
  pl.prm.far_columns = -1;
  pl.prm.tile_width = -1;
  pl.xu = 0;
  // x window (tile_width): per block the contiguous column range of at most W entries that covers most of its nonzeros,
  // staged in LDS (in the product area: no LDS of its own); a nonzero inside it streams an LDS slot instead of a column
  int xp = (m.nnz > 0 && m.n_cols < SCAN_LDS_BIT) ? plan::scan_window_xp(prm.tile_width > 0 ? prm.tile_width : 0, wg, ipt) : 0;
  std::vector<int> sci;
  if (xp > 0 || pad) {
    int rc = ensure_host_col_ind(m);
    if (rc) return rc;
  }
  if (xp > 0) {
    sci = m.h_ci;
    sci.push_back(0);                                         // the kernel's last 8-byte pair of an odd nnz
    const int W = 2 * xp * wg;
    if (plan::build_scan_window(m.h_ci.data(), m.nnz, blocks, W, sci) > 0) {
      if (!pad) HIP_TRY(pl.scan_ci.upload(sci));              // (a padded plan streams its own copy below)
      pl.prm.tile_width = W;
      pl.xu = xp;
    } else {
      sci.clear();
      for (BlockDesc &d : blocks)
        if (!(d.kind_g & KIND_LONG)) d.cmin = d.cwidth = 0;
    }
  }
  if (pad) {
    std::vector<int> pci, src;
    plan::build_padded_streams(blocks, n_regular, cap, m.h_ci.data(), sci.empty() ? nullptr : sci.data(), pci, src);
    HIP_TRY(pl.scan_pci.upload(pci));
    HIP_TRY(pl.scan_pval.alloc(src.size()));
    DevBuf<int> dsrc;
    HIP_TRY(dsrc.upload(src));
    gather_values((int64_t)src.size(), dsrc.p, m.d_val, pl.scan_pval.p, m.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m.stream));                  // (the index array dies here)
    pl.n_regular = n_regular;
  }
  HIP_TRY(pl.blocks.upload(blocks));
  pl.lds_bytes = 8 * (cap + wg + 4) + 24 * 8;                 // (the window lives in the product area)
  pl.ldsx = pl.xu > 0;
  return CASK_HIP_OK;
}

// SLICE plan (slice_kernel.hpp, plan_host.hpp build_slice_plan): the short rows as row-mapped slices, the long rows as a
// SCAN plan over a compacted copy; the plan owns copies of the value and column streams in that order.
int build_slice_plan(cask_hip_matrix &m, const cask_hip_params &prm) {
  Plan &pl = m.plan;
  const int wg = prm.wg_size, ipt = prm.items_per_thread, cap = wg * ipt, k = prm.lanes_per_row;
  int rc = ensure_host_col_ind(m);
  if (rc) return rc;
  plan::SlicePlan sp;
  plan::build_slice_plan(m.h_rp.data(), m.n_rows, k, slice_rows_per_thread(slice_kernel_km(k)) * wg, wg, ipt, sp);
  const int64_t n_slice_nnz = (int64_t)sp.slice_src.size(), n_long_nnz = (int64_t)sp.long_src.size();
  pl.slice_k = k;
  pl.n_slice_blocks = (int)sp.slices.size();
  pl.long_nnz = n_long_nnz;
  pl.n_long_rows = sp.n_long_pieces_rows;
  pl.n_split_rows = (int)sp.splits.size();
  pl.n_blocks = (int)sp.blocks.size();
  pl.grid = pl.n_blocks + pl.n_slice_blocks;
  pl.prm.index16 = -1;
  pl.prm.far_columns = -1;
  pl.prm.tile_width = -1;
  pl.prm.nontemporal = 1;                                     // (the slice kernels stream nontemporally: one instantiation)
  pl.xu = 0;
  // columns in plan order (host), values gathered on the device (the handle keeps no host copy of them)
  std::vector<int> sci((size_t)n_slice_nnz + 2), lci((size_t)n_long_nnz + 2, 0);
  for (int64_t i = 0; i < n_slice_nnz; i++) sci[(size_t)i] = m.h_ci[(size_t)sp.slice_src[(size_t)i]];
  for (int64_t i = 0; i < n_long_nnz; i++) lci[(size_t)i] = m.h_ci[(size_t)sp.long_src[(size_t)i]];
  if (n_long_nnz > 0) {
    const int xp = plan::scan_window_xp(prm.tile_width > 0 ? prm.tile_width : 0, wg, ipt);
    if (xp > 0) {
      std::vector<int> plain(lci.begin(), lci.begin() + n_long_nnz);
      const int W = 2 * std::min(xp, 4) * wg;                 // (the slice launch has window instantiations up to 4 loads per thread)
      if (plan::build_scan_window(plain.data(), n_long_nnz, sp.blocks, W, lci) > 0) {
        pl.prm.tile_width = W;
        pl.xu = std::min(xp, 4);
      } else {
        for (BlockDesc &d : sp.blocks)
          if (!(d.kind_g & KIND_LONG)) d.cmin = d.cwidth = 0;
      }
    }
  }
  HIP_TRY(pl.slices.upload(sp.slices));
  HIP_TRY(pl.slice_slot.upload(sp.slot));
  HIP_TRY(pl.slice_ci.upload(sci));
  HIP_TRY(pl.scan_ci.upload(lci));
  HIP_TRY(pl.blocks.upload(sp.blocks));
  HIP_TRY(pl.scan_meta.upload(sp.meta));
  if (sp.rowmap.empty()) sp.rowmap.push_back(0);
  HIP_TRY(pl.scan_rowmap.upload(sp.rowmap));
  if (!sp.splits.empty()) {
    HIP_TRY(pl.split_rows.upload(sp.splits));
    HIP_TRY(pl.partials.alloc(sp.n_partial_slots));
  }
  HIP_TRY(pl.slice_val.alloc((size_t)n_slice_nnz + 2));
  HIP_TRY(pl.long_val.alloc((size_t)n_long_nnz + 2));
  HIP_TRY(hipMemsetAsync(pl.slice_val.p, 0, ((size_t)n_slice_nnz + 2) * sizeof(double), m.stream));
  HIP_TRY(hipMemsetAsync(pl.long_val.p, 0, ((size_t)n_long_nnz + 2) * sizeof(double), m.stream));
  {
    DevBuf<int> src;
    HIP_TRY(src.upload(sp.slice_src));
    gather_values(n_slice_nnz, src.p, m.d_val, pl.slice_val.p, m.stream);
    DevBuf<int> lsrc;
    HIP_TRY(lsrc.upload(sp.long_src));
    gather_values(n_long_nnz, lsrc.p, m.d_val, pl.long_val.p, m.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m.stream));                  // (the index arrays die here)
  }
  const int slice_lds = 8 * slice_rows_per_thread(slice_kernel_km(k)) * wg;
  pl.lds_bytes = std::max(8 * (cap + wg + 4) + 24 * 8, slice_lds);
  pl.ldsx = pl.xu > 0;
  return CASK_HIP_OK;
}

// VECTOR: which kernel runs L lanes per row -- from 4 lanes the pair-load kernel (16-byte value pairs, VEC_RG row groups per
// wave in flight), below it the round-1 one (one row per lane group at a time).  (r6: the pair-load kernel at L = 1 / 2 --
// webbase2 56.8 / 68.7 us against 47.4 / 55.3, G3_circuit-like 37.2 / 52.0 against 38.5 / 61.2: not adopted.)
constexpr bool vector_pair_kernel(int lanes) { return lanes >= 4; }

int build_plan(cask_hip_matrix &m, const cask_hip_params &requested) {
  cask_hip_params prm;
  int rc = resolve_params(m, requested, prm);
  if (rc) return rc;
  Plan &pl = m.plan;
  pl.prm = prm;
  m.solver_load_choice = 0;                                   // (a new plan: the solvers measure their load policy again)
  pl.blocks.release();
  pl.long_blocks.release();
  pl.ci16.release();
  pl.packed12 = false;
  pl.one_window = false;
  pl.xchunk.release();
  pl.maxch = 0;
  pl.any_skew = false;
  pl.n_blocks = pl.n_long_blocks = 0;
  pl.split_rows.release();
  pl.partials.release();
  pl.dot_part.release();
  pl.xspan.release();
  pl.n_long_rows = pl.n_split_rows = 0;
  pl.vec_long_rows = 0;
  pl.vec_long_wave = false;
  pl.scan_meta.release();
  pl.scan_rowmap.release();
  pl.scan_ci.release();
  pl.scan_pval.release();
  pl.scan_pci.release();
  pl.n_regular = 0;
  pl.slices.release();
  pl.slice_slot.release();
  pl.slice_val.release();
  pl.long_val.release();
  pl.slice_ci.release();
  pl.n_slice_blocks = pl.slice_k = 0;
  pl.long_nnz = 0;
  pl.grid = 0;
  pl.lds_bytes = 0;
  pl.ldsx = false;
  pl.xu = 0;
  if (m.n_rows == 0) return CASK_HIP_OK;

  const int tile = prm.tile_width > 0 ? prm.tile_width : 0;
  if (prm.variant == CASK_HIP_VARIANT_MERGE_WAVE) {
    // persistent pipelined waves: small per-wave blocks, long rows in their own launch
    const int cap = 64 * prm.items_per_thread;
    std::vector<BlockDesc> blocks, longs;
    std::vector<SplitRow> splits;
    int n_long = 0, n_slots = 0;
    plan::build_merge_blocks(m.h_rp.data(), m.n_rows, cap, 127, 32768, 64, blocks, &longs, splits, n_long, n_slots);
    pl.n_long_rows = n_long;
    pl.n_split_rows = (int)splits.size();
    pl.n_blocks = (int)blocks.size();
    pl.n_long_blocks = (int)longs.size();
    HIP_TRY(pl.blocks.upload(blocks));
    if (!longs.empty()) HIP_TRY(pl.long_blocks.upload(longs));
    if (!splits.empty()) {
      HIP_TRY(pl.split_rows.upload(splits));
      HIP_TRY(pl.partials.alloc(n_slots));
    }
    const int waves_per_wg = prm.wg_size / 64;
    pl.lds_bytes = waves_per_wg * ((cap + 2) * 8 + 128 * 4);
    pl.prm.tile_width = -1;                                  // x comes from L2 in this kernel
    int occ = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, merge_wave_fn(prm.items_per_thread, prm.nontemporal > 0),
                                                         prm.wg_size, pl.lds_bytes));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m.device));
    if (occ < 1) return fail(CASK_HIP_ERR_INVALID, "design point does not fit on a CU (LDS/registers)");
    const int want = (pl.n_blocks + waves_per_wg - 1) / waves_per_wg;
    pl.grid = std::max(1, std::min(want, occ * cus));
    if (pl.n_blocks == 0) pl.grid = 0;
    pl.ldsx = false;
  } else if (prm.variant == CASK_HIP_VARIANT_SCAN) {
    int rc2 = build_scan_plan(m, prm);
    if (rc2) return rc2;
  } else if (prm.variant == CASK_HIP_VARIANT_SLICE) {
    int rc2 = build_slice_plan(m, prm);
    if (rc2) return rc2;
  } else if (prm.variant == CASK_HIP_VARIANT_MERGE) {
    const int cap = prm.wg_size * prm.items_per_thread;
    std::vector<BlockDesc> blocks;
    std::vector<SplitRow> splits;
    int n_long = 0, n_slots = 0;
    plan::build_merge_blocks(m.h_rp.data(), m.n_rows, cap, 2 * prm.wg_size - 1, (long)cap * LONG_PIECE_FACTOR, prm.wg_size, blocks,
                             nullptr, splits, n_long, n_slots);
    for (const BlockDesc &b : blocks) pl.any_skew = pl.any_skew || (b.kind_g & KIND_SKEW);
    pl.n_long_rows = n_long;
    pl.n_split_rows = (int)splits.size();
    pl.grid = (int)blocks.size();
    pl.n_blocks = pl.grid;
    HIP_TRY(pl.blocks.upload(blocks));
    if (!splits.empty()) {
      HIP_TRY(pl.split_rows.upload(splits));
      HIP_TRY(pl.partials.alloc(n_slots));
    }
    const int base_lds = 8 * (cap + 2) + 8 * prm.wg_size;
    pl.xu = 0;
    pl.prm.tile_width = -1;
    if (tile > 0 && m.nnz > 0 && prm.index16 > 0) {
      // tile = set of column ranges in 64-column chunks, 16-bit slot indices (host planner)
      int rc2 = ensure_host_col_ind(m);
      if (rc2) return rc2;
      // (a window that shares the product area -- merge_window_aliased, spmv_common.hpp -- costs no LDS of its own)
      auto window_lds = [&](int xu) { return merge_window_aliased(xu, prm.items_per_thread) ? 0 : 8 * xu * prm.wg_size; };
      int xu_cap = 8;
      while (xu_cap > 0 && base_lds + window_lds(xu_cap) > MAX_LDS_BYTES) xu_cap /= 2;
      const int max_slots = std::min(tile, xu_cap * prm.wg_size);
      std::vector<std::vector<int>> chunk_starts;
      std::vector<unsigned short> ci16;
      plan::build_chunk_tiles(m.h_ci.data(), m.nnz, blocks, max_slots, chunk_starts, ci16);
      pl.prm.far_columns = -1;
      if (m.halo_addr) plan::place_seam_blocks(m.h_ci.data(), m.halo_n_own, blocks, chunk_starts, prm.xcd_remap > 0);
      int max_used = 0;                                       // slots of the fullest tile
      for (const BlockDesc &d : blocks)
        if (!(d.kind_g & KIND_LONG)) max_used = std::max(max_used, d.cwidth);
      if (max_used > 0) {
        int xu = 1;
        while (xu * prm.wg_size < max_used) xu *= 2;
        pl.xu = xu;
        pl.maxch = xu * prm.wg_size / plan::TILE_SUB;
        pl.prm.tile_width = xu * prm.wg_size;
        std::vector<int> xchunk;
        plan::build_chunk_table(blocks, chunk_starts, prm.wg_size, xu, xchunk);
        HIP_TRY(pl.xchunk.upload(xchunk));
        pl.one_window = true;
        for (const BlockDesc &d : blocks)
          if (!(d.kind_g & KIND_LONG) && d.cwidth > 0 && !(d.kind_g & KIND_CONTIG)) pl.one_window = false;
        // index16 = 1 packs the slots 12 bits each where the kernel has that layout (8 items per thread, tile of
        // at most 4096 slots): 1.5 instead of 2 bytes per nonzero; index16 = 2 keeps 16-bit slots
        pl.packed12 = prm.index16 == 1 && prm.items_per_thread == 8 && pl.prm.tile_width <= 4096;
        pl.slot_bytes_per_nnz = pl.packed12 ? 1.5 : 2.0;
        if (pl.packed12) {
          std::vector<unsigned short> packed;
          plan::pack_slots12(m.nnz, blocks, ci16, prm.wg_size, packed);
          HIP_TRY(pl.ci16.upload(packed));
        } else {
          HIP_TRY(pl.ci16.upload(ci16));
        }
        pl.prm.index16 = pl.packed12 ? 1 : 2;                 // reported: what the plan streams
      }
      HIP_TRY(pl.blocks.upload(blocks));               // cwidth now holds the slots each block uses
    } else if (tile > 0 && m.nnz > 0) {
      // 32-bit indices: one contiguous window per block
      hipLaunchKernelGGL(k_col_span_blocks, dim3(pl.grid), dim3(256), 0, m.stream, pl.blocks.p, pl.grid, m.d_ci);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(blocks.data(), pl.blocks.p, blocks.size() * sizeof(BlockDesc),
                             hipMemcpyDeviceToHost, m.stream));
      HIP_TRY(hipStreamSynchronize(m.stream));
      int max_width = 0;
      for (const BlockDesc &d : blocks)
        if (!(d.kind_g & KIND_LONG) && d.cwidth <= tile) max_width = std::max(max_width, d.cwidth);
      int xu = 0;
      if (max_width > 0) {
        xu = 1;
        while (xu < 8 && xu * prm.wg_size < max_width) xu *= 2;
        while (xu > 0 && base_lds + (merge_window_aliased(xu, prm.items_per_thread) ? 0 : 8 * xu * prm.wg_size) > MAX_LDS_BYTES) xu /= 2;
      }
      pl.xu = xu;
      if (xu > 0) pl.prm.tile_width = xu * prm.wg_size;
      if (m.halo_addr) {
        int rc2 = ensure_host_col_ind(m);
        if (rc2) return rc2;
        std::vector<std::vector<int>> none;
        plan::place_seam_blocks(m.h_ci.data(), m.halo_n_own, blocks, none, prm.xcd_remap > 0);
        HIP_TRY(pl.blocks.upload(blocks));
      }
    } else if (m.halo_addr && m.nnz > 0) {
      int rc2 = ensure_host_col_ind(m);
      if (rc2) return rc2;
      std::vector<std::vector<int>> none;
      plan::place_seam_blocks(m.h_ci.data(), m.halo_n_own, blocks, none, prm.xcd_remap > 0);
      HIP_TRY(pl.blocks.upload(blocks));
    }
    if (!pl.ci16.p) pl.prm.index16 = -1;
    // shares of a fused dot (a few KB): allocated with the plan, so that a product with the dot epilogue
    // never allocates -- it may be running under stream capture
    HIP_TRY(pl.dot_part.alloc((size_t)pl.grid + (size_t)pl.n_split_rows));
    pl.ldsx = pl.xu > 0;
    pl.lds_bytes = base_lds + (merge_window_aliased(pl.xu, prm.items_per_thread) ? 0 : 8 * pl.xu * prm.wg_size);
  } else {
    // rows far longer than the lanes suit leave the row-mapped kernel (r6): long-row pieces for k_spmv_long + the fix-up,
    // the path the merge plans have -- L lanes walking a 4 700-entry row of a power-law matrix took 260-300 us
    {
      const int long_len = plan::vector_long_row_len(prm.lanes_per_row);
      std::vector<BlockDesc> longs;
      std::vector<SplitRow> splits;
      int n_slots = 0;
      plan::build_vector_long_pieces(m.h_rp.data(), m.n_rows, long_len, longs, splits, n_slots);
      if (!longs.empty()) {
        pl.vec_long_rows = long_len;
        pl.n_long_blocks = (int)longs.size();
        pl.n_split_rows = (int)splits.size();
        HIP_TRY(pl.long_blocks.upload(longs));
        if (!splits.empty()) {
          HIP_TRY(pl.split_rows.upload(splits));
          HIP_TRY(pl.partials.alloc(n_slots));
        }
        long long piece_nnz = 0;
        for (const BlockDesc &d : longs) {
          pl.n_long_rows += !(d.kind_g & KIND_PARTIAL) || d.nnz_start == m.h_rp[d.row_start];
          piece_nnz += d.nnz_count;
        }
        pl.vec_long_wave = piece_nnz / (long long)longs.size() < 1024;
      }
    }
    // L >= 4: the pair-load kernel, VEC_RG row groups per wave (spmv_kernels.hpp)
    const int rows_per_wg = prm.wg_size / prm.lanes_per_row * (vector_pair_kernel(prm.lanes_per_row) ? VEC_RG : 1);
    pl.grid = (m.n_rows + rows_per_wg - 1) / rows_per_wg;
    int max_width = 0;
    if (tile > 0 && m.nnz > 0) {
      HIP_TRY(pl.xspan.alloc(pl.grid));
      hipLaunchKernelGGL(k_col_span_rows, dim3(pl.grid), dim3(256), 0, m.stream, pl.xspan.p, pl.grid,
                         rows_per_wg, m.n_rows, m.d_rp, m.d_ci);
      HIP_TRY(hipGetLastError());
      std::vector<int2v> spans(pl.grid);
      HIP_TRY(hipMemcpyAsync(spans.data(), pl.xspan.p, spans.size() * sizeof(int2v), hipMemcpyDeviceToHost,
                             m.stream));
      HIP_TRY(hipStreamSynchronize(m.stream));
      const int eff_tile = std::min(tile, MAX_LDS_BYTES / 8);
      pl.prm.tile_width = eff_tile;
      for (const int2v &s : spans)
        if (s.y <= eff_tile) max_width = std::max(max_width, (int)s.y);
    }
    pl.ldsx = max_width > 0;
    pl.lds_bytes = 8 * max_width;
    if (vector_pair_kernel(prm.lanes_per_row) && max_width > 0)   // the pair-load kernel parks VEC_XW entries per lane
      pl.lds_bytes = 8 * std::max(max_width, std::min(VEC_XW * prm.wg_size, MAX_LDS_BYTES / 8));
  }
  if (!pl.ldsx && pl.prm.tile_width > 0 && m.nnz > 0) {
    // no block fits the tile: run the plain-gather kernels
  }
  return CASK_HIP_OK;
}

// The plan of `src` for `dst`, a handle over the SAME matrix in other device arrays (the rotating copies of the
// DSE's cold timing): every plan array is a function of the sparsity pattern alone (a SLICE plan's value copies: of
// the matrix, which is the same), so it is copied device to device
// instead of being planned again on the host (13 of the 14 plans per design point: most of a sweep's wall time).
int clone_plan(cask_hip_matrix &dst, const cask_hip_matrix &src) {
  Plan &d = dst.plan;
  const Plan &s = src.plan;
  if (dst.n_rows != src.n_rows || dst.n_cols != src.n_cols || dst.nnz != src.nnz)
    return fail(CASK_HIP_ERR_INVALID, "clone_plan: different matrices");
  d.prm = s.prm; d.grid = s.grid; d.lds_bytes = s.lds_bytes; d.ldsx = s.ldsx; d.xu = s.xu;
  d.n_blocks = s.n_blocks; d.n_long_blocks = s.n_long_blocks; d.packed12 = s.packed12; d.one_window = s.one_window;
  d.slot_bytes_per_nnz = s.slot_bytes_per_nnz;
  d.maxch = s.maxch; d.any_skew = s.any_skew; d.n_long_rows = s.n_long_rows;
  d.n_split_rows = s.n_split_rows; d.vec_long_rows = s.vec_long_rows; d.vec_long_wave = s.vec_long_wave;
  d.n_slice_blocks = s.n_slice_blocks; d.slice_k = s.slice_k; d.long_nnz = s.long_nnz;
  HIP_TRY(d.blocks.copy_from(s.blocks)); HIP_TRY(d.long_blocks.copy_from(s.long_blocks));
  HIP_TRY(d.split_rows.copy_from(s.split_rows)); HIP_TRY(d.partials.copy_from(s.partials));
  HIP_TRY(d.ci16.copy_from(s.ci16)); HIP_TRY(d.xchunk.copy_from(s.xchunk)); HIP_TRY(d.dot_part.copy_from(s.dot_part));
  HIP_TRY(d.xspan.copy_from(s.xspan));
  HIP_TRY(d.scan_meta.copy_from(s.scan_meta)); HIP_TRY(d.scan_rowmap.copy_from(s.scan_rowmap));
  HIP_TRY(d.scan_ci.copy_from(s.scan_ci));
  HIP_TRY(d.scan_pval.copy_from(s.scan_pval)); HIP_TRY(d.scan_pci.copy_from(s.scan_pci));
  d.n_regular = s.n_regular;
  // SLICE: the plan's copies of the value stream are the same bits in every copy of the matrix
  HIP_TRY(d.slices.copy_from(s.slices)); HIP_TRY(d.slice_slot.copy_from(s.slice_slot)); HIP_TRY(d.slice_ci.copy_from(s.slice_ci));
  HIP_TRY(d.slice_val.copy_from(s.slice_val)); HIP_TRY(d.long_val.copy_from(s.long_val));
  return CASK_HIP_OK;
}

// ------------------------------------------------------------------ dispatch
template <int L>
int launch_vector_l(const cask_hip_matrix &m, const double *x, double *y, hipStream_t s) {
  const Plan &pl = m.plan;
  const dim3 grid(pl.grid), block(pl.prm.wg_size);
  const int remap = pl.prm.xcd_remap > 0, tile = std::max(0, pl.prm.tile_width);
  const bool nt = pl.prm.nontemporal > 0;
#define CASK_LAUNCH_V(LDSX, NT)                                                                                \
  do {                                                                                                         \
    if (vector_pair_kernel(L))                                                                                 \
      hipLaunchKernelGGL((k_spmv_vector2<(L >= 4 ? L : 4), LDSX, NT>), grid, block, pl.lds_bytes, s, m.n_rows, pl.grid, \
                         remap, tile, (int)m.nnz, pl.vec_long_rows, pl.xspan.p, m.d_rp, m.d_ci, m.d_val, x, y); \
    else                                                                                                       \
      hipLaunchKernelGGL((k_spmv_vector<L, LDSX, NT>), grid, block, pl.lds_bytes, s, m.n_rows, pl.grid, remap, \
                         tile, pl.vec_long_rows, pl.xspan.p, m.d_rp, m.d_ci, m.d_val, x, y);                   \
  } while (0)
  if (pl.ldsx) { if (nt) CASK_LAUNCH_V(true, true); else CASK_LAUNCH_V(true, false); }
  else         { if (nt) CASK_LAUNCH_V(false, true); else CASK_LAUNCH_V(false, false); }
#undef CASK_LAUNCH_V
  if (pl.n_long_blocks > 0) {                                 // the rows the row-mapped kernel left alone
    // a wave per piece when the pieces are short (a power-law matrix has thousands of rows of 40-200 nonzeros: a
    // workgroup of 256 for each kept the chip busy with 8 blocks per CU of mostly idle lanes), a workgroup when they are long
    const dim3 lb(pl.vec_long_wave ? 64 : 256);
    if (nt)
      hipLaunchKernelGGL((k_spmv_long<true>), dim3(pl.n_long_blocks), lb, 0, s, pl.long_blocks.p, pl.n_long_blocks, m.d_ci,
                         m.d_val, x, y, pl.partials.p);
    else
      hipLaunchKernelGGL((k_spmv_long<false>), dim3(pl.n_long_blocks), lb, 0, s, pl.long_blocks.p, pl.n_long_blocks, m.d_ci,
                         m.d_val, x, y, pl.partials.p);
    if (pl.n_split_rows > 0)
      hipLaunchKernelGGL(k_spmv_fixup, dim3((pl.n_split_rows + 63) / 64), dim3(64), 0, s, pl.split_rows.p, pl.n_split_rows,
                         pl.partials.p, y, DotEpilogue{nullptr, nullptr}, (const int *)nullptr);
  }
  return CASK_HIP_OK;
}

// a launch with the dot epilogue parks the block's slice of w (2*wg_size doubles) and its per-wave sums
// (16 doubles) in dynamic LDS behind the x tile
int dot_lds_bytes(int wg_size) { return 16 * wg_size + 128; }

// What a merge launch needs from the handle and its plan.
MergeLaunch merge_launch_of(const cask_hip_matrix &m, const DotEpilogue &dot, const SolverPass *pass) {
  const Plan &pl = m.plan;
  MergeLaunch l{};
  l.grid = pl.grid;
  l.wg_size = pl.prm.wg_size;
  // a solver pass always carries the dot area: its first 16 doubles also serve the sums of the pass scalars
  l.lds_bytes = pl.lds_bytes + ((dot.w || pass) ? dot_lds_bytes(pl.prm.wg_size) : 0);
  l.solver_pass = pass != nullptr;
  if (pass) l.pass = *pass;
  l.xu = pl.xu;
  l.remap = pl.prm.xcd_remap > 0;
  l.n_cols = m.n_cols;
  l.nnz = (int)m.nnz;
  l.maxch = pl.maxch;
  l.nontemporal = pl.prm.nontemporal > 0;
  l.any_skew = pl.any_skew;
  l.blocks = pl.blocks.p;
  l.rp = m.d_rp;
  l.ci = m.d_ci;
  l.ci16 = reinterpret_cast<const unsigned *>(pl.ci16.p);
  l.packed12 = pl.packed12;
  l.one_window = pl.one_window;
  l.xchunk = pl.xchunk.p;
  l.val = m.d_val;
  l.partials = pl.partials.p;
  l.halo = XHalo{m.halo_n_own, m.halo_addr, m.halo_shift};
  l.dot = dot;
  return l;
}

// the pieces of split long rows, summed in piece order behind the product launch
void launch_merge_fixup(const cask_hip_matrix &m, double *y, hipStream_t s, const DotEpilogue &dot, const SolverPass *pass) {
  const Plan &pl = m.plan;
  if (pl.n_split_rows <= 0 || (pass && pass->final_only)) return;   // (a final_only launch runs no product: nothing to fix up)
  // a solver pass's dot operand is a stored vector by the time the fix-up runs: the direction this very
  // launch stored (b_new), or the plain vector the pass names (wa without wb)
  const double *fw = pass ? (dot.dot_part ? (pass->wa && !pass->wb ? pass->wa : pass->b_new) : nullptr) : dot.w;
  const DotEpilogue fix{fw, fw ? dot.dot_part + pl.grid : nullptr};
  hipLaunchKernelGGL(k_spmv_fixup, dim3((pl.n_split_rows + 63) / 64), dim3(64), 0, s, pl.split_rows.p,
                     pl.n_split_rows, pl.partials.p, y, fix, pass ? (const int *)pass->done : (const int *)nullptr);
}

template <int IPT>
int launch_merge_i(const cask_hip_matrix &m, const double *x, double *y, hipStream_t s, const DotEpilogue &dot,
                   const SolverPass *pass) {
  launch_merge_blocks<IPT>(merge_launch_of(m, dot, pass), x, y, s);
  launch_merge_fixup(m, y, s, dot, pass);
  return CASK_HIP_OK;
}

template <int IPT>
int launch_merge_wave_i(const cask_hip_matrix &m, const double *x, double *y, hipStream_t s) {
  const Plan &pl = m.plan;
  const int remap = pl.prm.xcd_remap > 0 && pl.grid >= 8;   // every XCD needs at least one workgroup
  const bool nt = pl.prm.nontemporal > 0;
  if (pl.grid > 0) {
    if (nt)
      hipLaunchKernelGGL((k_spmv_merge_wave<IPT, true>), dim3(pl.grid), dim3(pl.prm.wg_size), pl.lds_bytes, s,
                         pl.blocks.p, pl.n_blocks, remap, (int)m.nnz, m.d_rp, m.d_ci, m.d_val, x, y);
    else
      hipLaunchKernelGGL((k_spmv_merge_wave<IPT, false>), dim3(pl.grid), dim3(pl.prm.wg_size), pl.lds_bytes, s,
                         pl.blocks.p, pl.n_blocks, remap, (int)m.nnz, m.d_rp, m.d_ci, m.d_val, x, y);
  }
  if (pl.n_long_blocks > 0) {
    if (nt)
      hipLaunchKernelGGL((k_spmv_long<true>), dim3(pl.n_long_blocks), dim3(256), 0, s, pl.long_blocks.p,
                         pl.n_long_blocks, m.d_ci, m.d_val, x, y, pl.partials.p);
    else
      hipLaunchKernelGGL((k_spmv_long<false>), dim3(pl.n_long_blocks), dim3(256), 0, s, pl.long_blocks.p,
                         pl.n_long_blocks, m.d_ci, m.d_val, x, y, pl.partials.p);
  }
  if (pl.n_split_rows > 0)
    hipLaunchKernelGGL(k_spmv_fixup, dim3((pl.n_split_rows + 63) / 64), dim3(64), 0, s, pl.split_rows.p,
                       pl.n_split_rows, pl.partials.p, y, DotEpilogue{nullptr, nullptr}, (const int *)nullptr);
  return CASK_HIP_OK;
}

// The merge kernel can leave the shares of w.y behind (one per block + one per split row).
bool plan_fuses_dot(const Plan &pl) {
  return pl.prm.variant == CASK_HIP_VARIANT_MERGE && pl.grid > 0 && pl.dot_part.p != nullptr &&
         pl.lds_bytes + dot_lds_bytes(pl.prm.wg_size) <= MAX_LDS_BYTES;
}
int dot_part_count(const Plan &pl) { return pl.grid + pl.n_split_rows; }

// y = A x; with w != NULL (MERGE plans only) also plan.dot_part[0 .. dot_part_count) = shares of w.y.
// pass != NULL (MERGE plans with the dot epilogue only): a solver pass -- operand composed on the fly, see
// SolverPass; `want_dot` then says whether the shares of the dot are wanted.
int launch_spmv(cask_hip_matrix &m, const double *x, double *y, hipStream_t s, const double *w = nullptr,
                const SolverPass *pass = nullptr, bool want_dot = false) {
  Plan &pl = m.plan;
  if (w || pass) {
    if (!plan_fuses_dot(pl)) return fail(CASK_HIP_ERR_INVALID, "this design point has no fused dot epilogue");
  }
  const DotEpilogue dot{w, (w || (pass && want_dot)) ? pl.dot_part.p : nullptr};
  if (m.n_rows == 0 || (pl.grid == 0 && pl.n_long_blocks == 0)) return CASK_HIP_OK;
  if (pl.prm.variant == CASK_HIP_VARIANT_MERGE_WAVE) {
    switch (pl.prm.items_per_thread) {
      case 2:  launch_merge_wave_i<2>(m, x, y, s); break;
      case 4:  launch_merge_wave_i<4>(m, x, y, s); break;
      case 8:  launch_merge_wave_i<8>(m, x, y, s); break;
      default: launch_merge_wave_i<16>(m, x, y, s); break;
    }
    HIP_TRY(hipGetLastError());
    return CASK_HIP_OK;
  }
  if (pl.prm.variant == CASK_HIP_VARIANT_SCAN || pl.prm.variant == CASK_HIP_VARIANT_SLICE) {
    const bool slice = pl.prm.variant == CASK_HIP_VARIANT_SLICE;
    ScanLaunch l{};
    l.grid = slice ? pl.n_blocks : pl.grid;
    l.wg_size = pl.prm.wg_size;
    l.lds_bytes = pl.lds_bytes;
    l.remap = pl.prm.xcd_remap > 0;
    l.nnz = slice ? (int)pl.long_nnz : (int)m.nnz;
    l.n_cols = m.n_cols;
    l.xp = pl.xu;
    l.nontemporal = pl.prm.nontemporal > 0;
    l.blocks = pl.blocks.p;
    l.rp = m.d_rp;
    l.ci = pl.scan_ci.p ? pl.scan_ci.p : m.d_ci;
    l.val = slice ? pl.long_val.p : m.d_val;
    l.meta = pl.scan_meta.p;
    l.rowmap = pl.scan_rowmap.p;
    l.partials = pl.partials.p;
    if (!slice && pl.scan_pval.p) {
      l.pad_val = pl.scan_pval.p;
      l.pad_ci = pl.scan_pci.p;
      l.n_regular = pl.n_regular;
    }
    if (slice) {
      l.n_slice_blocks = pl.n_slice_blocks;
      l.slice_k = pl.slice_k;
      l.slices = pl.slices.p;
      l.slice_val = pl.slice_val.p;
      l.slice_ci = pl.slice_ci.p;
      l.slice_slot = pl.slice_slot.p;
    }
    if (slice) {                                              // development only: time the two kinds of workgroup apart
      static const int diag = std::getenv("CASK_HIP_SLICE_DIAG") ? std::atoi(std::getenv("CASK_HIP_SLICE_DIAG")) : 0;
      if (diag == 1) l.n_slice_blocks = 0;                    // the long rows' blocks alone (the result is incomplete)
      if (diag == 2) l.grid = 0;                              // the slices alone
    }
    launch_scan(l, pl.prm.items_per_thread, x, y, s);
    if (pl.n_split_rows > 0)
      hipLaunchKernelGGL(k_spmv_fixup, dim3((pl.n_split_rows + 63) / 64), dim3(64), 0, s, pl.split_rows.p,
                         pl.n_split_rows, pl.partials.p, y, DotEpilogue{nullptr, nullptr}, (const int *)nullptr);
    HIP_TRY(hipGetLastError());
    return CASK_HIP_OK;
  }
  if (pl.prm.variant == CASK_HIP_VARIANT_VECTOR) {
    switch (pl.prm.lanes_per_row) {
      case 1:  launch_vector_l<1>(m, x, y, s); break;
      case 2:  launch_vector_l<2>(m, x, y, s); break;
      case 4:  launch_vector_l<4>(m, x, y, s); break;
      case 8:  launch_vector_l<8>(m, x, y, s); break;
      case 16: launch_vector_l<16>(m, x, y, s); break;
      case 32: launch_vector_l<32>(m, x, y, s); break;
      default: launch_vector_l<64>(m, x, y, s); break;
    }
  } else {
    switch (pl.prm.items_per_thread) {
      case 2:  launch_merge_i<2>(m, x, y, s, dot, pass); break;
      case 4:  launch_merge_i<4>(m, x, y, s, dot, pass); break;
      case 8:  launch_merge_i<8>(m, x, y, s, dot, pass); break;
      default: launch_merge_i<16>(m, x, y, s, dot, pass); break;
    }
  }
  HIP_TRY(hipGetLastError());
  return CASK_HIP_OK;
}

int check_csr(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *row_ptr) {
  if (n_rows < 0 || n_cols < 0 || nnz < 0) return fail(CASK_HIP_ERR_INVALID, "negative dimension");
  if (nnz >= (int64_t)std::numeric_limits<int32_t>::max())
    return fail(CASK_HIP_ERR_INVALID, "nnz must fit int32 (CsrMatrix uses int indices)");
  if (row_ptr[0] != 0) return fail(CASK_HIP_ERR_INVALID, "row_ptr[0] must be 0");
  for (int32_t r = 0; r < n_rows; r++)
    if (row_ptr[r + 1] < row_ptr[r]) return fail(CASK_HIP_ERR_INVALID, "row_ptr must be non-decreasing");
  if (row_ptr[n_rows] != nnz) return fail(CASK_HIP_ERR_INVALID, "row_ptr[n_rows] must equal nnz");
  return CASK_HIP_OK;
}

int finish_create(std::unique_ptr<cask_hip_matrix> &m, const cask_hip_params *params, cask_hip_matrix **out) {
  m->max_row = 0;
  m->empty_rows = 0;
  for (int r = 0; r < m->n_rows; r++) {
    const int len = m->h_rp[r + 1] - m->h_rp[r];
    m->max_row = std::max(m->max_row, len);
    m->empty_rows += len == 0;
  }
  HIP_TRY(hipGetDevice(&m->device));
  HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  if (params) m->requested = *params;
  int rc = build_plan(*m, m->requested);
  if (rc) return rc;
  *out = m.release();
  return CASK_HIP_OK;
}

int ensure_device() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(CASK_HIP_ERR_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
  return CASK_HIP_OK;
}

int blas_grid(int64_t n) {
  const int64_t want = (n / 2 + BLAS_WG - 1) / BLAS_WG;
  return (int)std::max<int64_t>(1, std::min<int64_t>(BLAS_MAX_PARTIALS, want));
}
// ... of a solver's launches (blas1_kernels.hpp: the BIG and the SMALL shape)
struct SolverShape { int grid, wg; bool big; };
// sharded: the scalars arrive all-reduced (no partial sums to add up: what BIG is for), and ranks that share a device in a
// rehearsal are time-sliced against each other -- SMALL there.  CASK_HIP_SOLVER_SHAPE=big|small: development A/B.
SolverShape solver_shape(int64_t n, bool sharded) {
  const int64_t pairs = n / 2;
  bool big = !sharded && pairs > (int64_t)SOLVER_GRID_MAX * SOLVER_WG;
  if (const char *e = std::getenv("CASK_HIP_SOLVER_SHAPE")) big = std::strcmp(e, "big") == 0 ? true : std::strcmp(e, "small") == 0 ? false : big;
  if (big) return SolverShape{SOLVER_GRID_MAX, SOLVER_WG, true};
  return SolverShape{blas_grid(n), BLAS_WG, false};
}
// an update launch in the instantiation of the shape
#define CASK_LAUNCH_UPD(big, kernel, ...)                                      \
  do {                                                                          \
    if (big) hipLaunchKernelGGL(kernel<true>, __VA_ARGS__);                     \
    else     hipLaunchKernelGGL(kernel<false>, __VA_ARGS__);                    \
  } while (0)
#define CASK_LAUNCH_UPD_J(big, kernel, jac, ...)                               \
  do {                                                                          \
    if (big) hipLaunchKernelGGL((kernel<jac, true>), __VA_ARGS__);              \
    else     hipLaunchKernelGGL((kernel<jac, false>), __VA_ARGS__);             \
  } while (0)

// A solver re-reads the matrix every iteration.  When matrix + vectors fit the 256 MiB Infinity Cache,
// cached loads can beat the streaming ("nt") ones the one-shot product prefers -- or lose to them: on
// the G3_circuit-like system the cached product is 8 % faster, on the cant-like one 20 % slower.  So
// the solver MEASURES (the DSE idea applied at run time): unless the caller pinned `nontemporal`,
// iterations 16-31 run with streaming loads, 32-47 with cached loads, and from 48 on the faster of the
// two.  The load policy changes no arithmetic, so the iterates are the same either way; the plan is
// restored when the solver returns.
// Completion word of the host entry: stored (system scope) by a one-thread launch BEHIND the product on the handle's stream.
// The product's y went to pinned host memory before this launch began (stream order, the kernel-end release), and posted
// writes of one device reach host memory in order: a CPU that sees the word sees y.
__global__ void k_done_word(unsigned long long *word, unsigned long long seq) {
  __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// CASK_HIP_HOST_DONE=sync|word (development A/B): how the CPU waits for something the GPU leaves in host memory
bool host_done_word() {
  static const int mode = [] {
    const char *e = std::getenv("CASK_HIP_HOST_DONE");
    return e && std::strcmp(e, "sync") == 0 ? 0 : e && std::strcmp(e, "word") == 0 ? 1 : HOST_DONE_DEFAULT;
  }();
  return mode == 1;
}
// Poll a word in pinned host memory until it holds seq; false after `limit_ms` (the caller then synchronises: a launch that
// faulted never stores the word, and the synchronisation is what reports it).
bool poll_word(const volatile unsigned long long *word, unsigned long long seq, int limit_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (*word == seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return true;
    }
    if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(limit_ms)) return false;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}
bool ensure_checkpoint(cask_hip_matrix *m) {
  if (m->checkpoint) return true;
  std::unique_ptr<DeferredFlags> c(new DeferredFlags);
  if (c->create() != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  m->checkpoint = std::move(c);
  return true;
}

struct SolverLoadPolicy {
  cask_hip_matrix *m, *mt;
  int saved = 0, saved_t = 0;
  bool choosing = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  SolverLoadPolicy(cask_hip_matrix *m_, cask_hip_matrix *mt_, int n_vectors) : m(m_), mt(mt_) {
    saved = m->plan.prm.nontemporal;
    if (mt) saved_t = mt->plan.prm.nontemporal;
    const int64_t per_matrix = 12 * m->nnz + 4 * ((int64_t)m->n_rows + 1);
    const int64_t working_set = per_matrix * (mt ? 2 : 1) + 8 * (int64_t)m->n_rows * n_vectors;
    choosing = m->requested.nontemporal == 0 && working_set < (int64_t)(224 << 20) && !m->plan.any_skew &&
               !(mt && mt->plan.any_skew);
    if (choosing && m->solver_load_choice != 0) {             // measured by an earlier solve on this handle (r6): no second trial
      set(m->solver_load_choice);
      choosing = false;
    }
    if (choosing)
      for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) choosing = false;
  }
  void set(int nt) {
    m->plan.prm.nontemporal = nt;
    if (mt) mt->plan.prm.nontemporal = nt;
  }
  // call at every 16-iteration checkpoint, before the stream is synchronised
  void record(int launched, hipStream_t s) {
    if (!choosing) return;
    if (launched == 16) (void)hipEventRecord(ev[0], s);
    if (launched == 32) { (void)hipEventRecord(ev[1], s); set(-1); }
    if (launched == 48) (void)hipEventRecord(ev[2], s);
  }
  // ... and after it
  void decide(int launched) {
    if (!choosing || launched != 48) return;
    float t_stream = 0.f, t_cached = 0.f;
    if (hipEventElapsedTime(&t_stream, ev[0], ev[1]) == hipSuccess &&
        hipEventElapsedTime(&t_cached, ev[1], ev[2]) == hipSuccess && t_cached < t_stream)
      set(-1);
    else
      set(1);
    m->solver_load_choice = m->plan.prm.nontemporal < 0 ? -1 : 1;
    choosing = false;
  }
  ~SolverLoadPolicy() {
    m->plan.prm.nontemporal = saved;
    if (mt) mt->plan.prm.nontemporal = saved_t;
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};

// Scratch owned by a solver run.
struct SolverScratch {
  DevBuf<double> partials_a, partials_b, scalars;   // scalars: [0..7]
  DevBuf<int> flags;                                // [0] done, [1] iterations
};

}  // namespace

// ======================================================================= C ABI
extern "C" {

#ifdef CASK_STAMPS
// diagnostic build only: point the kernels' stamp buffer at device memory (8 x u64 per workgroup)
int cask_hip_debug_set_stamps(unsigned long long *d_buf) {
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_buf, sizeof(d_buf)));
  return CASK_HIP_OK;
}
#endif

const char *cask_hip_last_error(void) { return g_err.c_str(); }
int cask_hip_abi_version(void) { return CASK_HIP_ABI_VERSION; }

int cask_hip_device_count(int32_t *count) {
  if (!count) return fail(CASK_HIP_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  *count = n;
  return CASK_HIP_OK;
}

int cask_hip_device_props_get(int32_t device, cask_hip_device_props *out) {
  if (!out) return fail(CASK_HIP_ERR_INVALID, "out is NULL");
  int rc = ensure_device();
  if (rc) return rc;
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  std::memset(out, 0, sizeof(*out));
  std::snprintf(out->name, sizeof(out->name), "%s", p.name);
  std::snprintf(out->arch, sizeof(out->arch), "%s", p.gcnArchName);
  out->compute_units = p.multiProcessorCount;
  out->lds_bytes_per_cu = (int32_t)p.maxSharedMemoryPerMultiProcessor;
  out->wavefront_size = p.warpSize;
  out->clock_mhz = p.clockRate / 1000;
  out->hbm_bytes = (int64_t)p.totalGlobalMem;
  out->l2_bytes = p.l2CacheSize;
  return CASK_HIP_OK;
}

int cask_hip_device_pci_bus_id(int32_t device, char *out, int32_t len) {
  if (!out || len < 16) return fail(CASK_HIP_ERR_INVALID, "out must hold at least 16 bytes");
  out[0] = 0;
  HIP_TRY(hipDeviceGetPCIBusId(out, len, device));
  return CASK_HIP_OK;
}

int cask_hip_csr_create(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *row_ptr,
                        const int32_t *col_ind, const double *values, const cask_hip_params *params,
                        cask_hip_matrix **out) {
  if (!out) return fail(CASK_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!row_ptr || (nnz > 0 && (!col_ind || !values))) return fail(CASK_HIP_ERR_INVALID, "NULL array");
  int rc = check_csr(n_rows, n_cols, nnz, row_ptr);
  if (rc) return rc;
  for (int64_t k = 0; k < nnz; k++)
    if (col_ind[k] < 0 || col_ind[k] >= n_cols) return fail(CASK_HIP_ERR_INVALID, "column index out of range");
  rc = ensure_device();
  if (rc) return rc;
  std::unique_ptr<cask_hip_matrix> m(new cask_hip_matrix);
  m->n_rows = n_rows;
  m->n_cols = n_cols;
  m->nnz = nnz;
  m->h_rp.assign(row_ptr, row_ptr + n_rows + 1);
  if (nnz) m->h_ci.assign(col_ind, col_ind + nnz);
  HIP_TRY(m->own_rp.upload(row_ptr, (size_t)n_rows + 1));
  HIP_TRY(m->own_ci.upload(col_ind, (size_t)nnz));
  HIP_TRY(m->own_val.upload(values, (size_t)nnz));
  m->d_rp = m->own_rp.p;
  m->d_ci = m->own_ci.p;
  m->d_val = m->own_val.p;
  return finish_create(m, params, out);
}

int cask_hip_csr_create_device(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *d_row_ptr,
                               const int32_t *d_col_ind, const double *d_values,
                               const cask_hip_params *params, cask_hip_matrix **out) {
  if (!out) return fail(CASK_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!d_row_ptr || (nnz > 0 && (!d_col_ind || !d_values))) return fail(CASK_HIP_ERR_INVALID, "NULL array");
  if (n_rows < 0 || n_cols < 0 || nnz < 0) return fail(CASK_HIP_ERR_INVALID, "negative dimension");
  if ((reinterpret_cast<uintptr_t>(d_values) & 15) || (reinterpret_cast<uintptr_t>(d_col_ind) & 7))
    return fail(CASK_HIP_ERR_INVALID, "device arrays must be 16-byte (values) / 8-byte (col_ind) aligned");
  int rc = ensure_device();
  if (rc) return rc;
  std::unique_ptr<cask_hip_matrix> m(new cask_hip_matrix);
  m->n_rows = n_rows;
  m->n_cols = n_cols;
  m->nnz = nnz;
  m->h_rp.resize((size_t)n_rows + 1);
  HIP_TRY(hipMemcpy(m->h_rp.data(), d_row_ptr, ((size_t)n_rows + 1) * sizeof(int), hipMemcpyDeviceToHost));
  rc = check_csr(n_rows, n_cols, nnz, m->h_rp.data());
  if (rc) return rc;
  if (nnz > 0) {                                              // columns must lie in [0, n_cols): one small reduction
    DevBuf<int> range;
    const int init[2] = {std::numeric_limits<int>::max(), std::numeric_limits<int>::min()};
    HIP_TRY(range.upload(init, 2));
    const int grid = (int)std::min<int64_t>(256, (nnz + 255) / 256);
    hipLaunchKernelGGL(k_col_range, dim3(grid), dim3(256), 0, nullptr, nnz, d_col_ind, range.p);
    HIP_TRY(hipGetLastError());
    int got[2];
    HIP_TRY(hipMemcpy(got, range.p, sizeof(got), hipMemcpyDeviceToHost));
    if (got[0] < 0 || got[1] >= n_cols) return fail(CASK_HIP_ERR_INVALID, "column index out of range");
  }
  m->d_rp = d_row_ptr;
  m->d_ci = d_col_ind;
  m->d_val = d_values;
  return finish_create(m, params, out);
}

int cask_hip_csr_destroy(cask_hip_matrix *m) {
  delete m;
  return CASK_HIP_OK;
}

int cask_hip_csr_set_params(cask_hip_matrix *m, const cask_hip_params *params) {
  if (!m) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
  cask_hip_params p{};
  if (params) p = *params;
  HIP_TRY(hipSetDevice(m->device));
  int rc = build_plan(*m, p);
  if (rc) {
    (void)build_plan(*m, m->requested);       // keep the handle usable with its previous design point
    return rc;
  }
  m->requested = p;
  if (m->transpose) return cask_hip_csr_set_params(m->transpose.get(), params);
  return CASK_HIP_OK;
}

int cask_hip_csr_get_params(const cask_hip_matrix *m, cask_hip_params *out) {
  if (!m || !out) return fail(CASK_HIP_ERR_INVALID, "NULL argument");
  *out = m->plan.prm;
  return CASK_HIP_OK;
}

int cask_hip_csr_get_info(const cask_hip_matrix *m, cask_hip_csr_info *out) {
  if (!m || !out) return fail(CASK_HIP_ERR_INVALID, "NULL argument");
  std::memset(out, 0, sizeof(*out));
  out->n_rows = m->n_rows;
  out->n_cols = m->n_cols;
  out->nnz = m->nnz;
  out->grid = m->plan.grid;
  out->lds_bytes = m->plan.lds_bytes;
  out->n_long_rows = m->plan.n_long_rows;
  out->n_split_rows = m->plan.n_split_rows;
  out->max_row_nnz = m->max_row;
  out->empty_rows = m->empty_rows;
  out->mean_row_nnz = m->n_rows ? (double)m->nnz / m->n_rows : 0.0;
  out->algorithmic_bytes = 12 * m->nnz + 4 * ((int64_t)m->n_rows + 1) + 8 * (int64_t)m->n_cols + 8 * (int64_t)m->n_rows;
  out->fuses_dot = plan_fuses_dot(m->plan) ? 1 : 0;
  return CASK_HIP_OK;
}

int cask_hip_spmv_device(cask_hip_matrix *m, const double *d_x, double *d_y, void *stream) {
  if (!m) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
  if ((m->n_cols > 0 && !d_x) || (m->n_rows > 0 && !d_y)) return fail(CASK_HIP_ERR_INVALID, "NULL vector");
  if (reinterpret_cast<uintptr_t>(d_x) & 15)                  // the x tile is staged in 16-byte pairs
    return fail(CASK_HIP_ERR_INVALID, "x must be 16-byte aligned");
  return launch_spmv(*m, d_x, d_y, static_cast<hipStream_t>(stream));
}

int cask_hip_spmv_sequence_device(cask_hip_matrix *const *mats, int32_t n_mats, const double *d_x, double *d_y,
                                  int32_t k, void *stream) {
  if (!mats || n_mats <= 0 || k < 0) return fail(CASK_HIP_ERR_INVALID, "bad argument");
  for (int i = 0; i < n_mats; i++) {
    if (!mats[i]) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
    if (mats[i]->n_rows != mats[0]->n_rows || mats[i]->n_cols != mats[0]->n_cols)
      return fail(CASK_HIP_ERR_INVALID, "the handles of a sequence must have one shape");
  }
  if ((mats[0]->n_cols > 0 && !d_x) || (mats[0]->n_rows > 0 && !d_y)) return fail(CASK_HIP_ERR_INVALID, "NULL vector");
  if (reinterpret_cast<uintptr_t>(d_x) & 15) return fail(CASK_HIP_ERR_INVALID, "x must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int i = 0; i < k; i++) {
    int rc = launch_spmv(*mats[i % n_mats], d_x, d_y, s);
    if (rc) return rc;
  }
  return CASK_HIP_OK;
}

int cask_hip_spmv_windows_device(cask_hip_matrix *const *mats, int32_t n_mats, const double *d_x, double *d_y,
                                 int32_t k, int32_t windows, int32_t as_graph, double *usec, void *stream) {
  if (!mats || n_mats <= 0 || k <= 0 || windows <= 0 || !usec) return fail(CASK_HIP_ERR_INVALID, "bad argument");
  for (int i = 0; i < n_mats; i++) {
    if (!mats[i]) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
    if (mats[i]->n_rows != mats[0]->n_rows || mats[i]->n_cols != mats[0]->n_cols)
      return fail(CASK_HIP_ERR_INVALID, "the handles of a sequence must have one shape");
  }
  if ((mats[0]->n_cols > 0 && !d_x) || (mats[0]->n_rows > 0 && !d_y)) return fail(CASK_HIP_ERR_INVALID, "NULL vector");
  if (reinterpret_cast<uintptr_t>(d_x) & 15) return fail(CASK_HIP_ERR_INVALID, "x must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  struct Events {
    std::vector<hipEvent_t> e;
    ~Events() { for (hipEvent_t ev : e) (void)hipEventDestroy(ev); }
  } evs;
  evs.e.reserve((size_t)windows + 1);
  for (int r = 0; r <= windows; r++) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, as_graph ? hipEventDefault : hipEventDisableSystemFence));
    evs.e.push_back(ev);
  }
  if (as_graph) {
    // ONE graph: windows x k kernel nodes with an event-record node at every window boundary (hipEventRecordWithFlags
    // ..External inside a stream capture).  Graph nodes are the cheapest launches the runtime has (a stream launch
    // measures 0.14 us more per product, and a graph START ~9 us -- so no graph per window); the event nodes make the
    // windows visible inside it.  Launched twice: the events keep the second launch's times, the first is the lead-in.
    struct Capture {
      hipStream_t cs = nullptr;
      hipGraph_t g = nullptr;
      hipGraphExec_t x = nullptr;
      ~Capture() {
        if (x) (void)hipGraphExecDestroy(x);
        if (g) (void)hipGraphDestroy(g);
        if (cs) (void)hipStreamDestroy(cs);
      }
    } cap;
    // ONE capture; at every window boundary an event-record node is added to the graph being captured, behind the
    // capture's current dependency set (hipStreamGetCaptureInfo_v2 + hipGraphAddEventRecordNode), and made the new
    // dependency set (hipStreamUpdateCaptureDependencies) so that the next window's first kernel node follows it.
    // (hipEventRecordWithFlags(.., hipEventRecordExternal) inside a capture returns "invalid argument" on ROCm 7.2 --
    // which is why torch refuses external events there; a child graph per window works but costs ~2.6 us per window.)
    HIP_TRY(hipStreamCreateWithFlags(&cap.cs, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(cap.cs, hipStreamCaptureModeThreadLocal));
    int rc = CASK_HIP_OK;
    hipError_t ge = hipSuccess;
    auto record_node = [&](hipEvent_t ev) -> hipError_t {
      hipStreamCaptureStatus st;
      hipGraph_t g = nullptr;
      const hipGraphNode_t *deps = nullptr;
      size_t n_deps = 0;
      hipError_t e = hipStreamGetCaptureInfo_v2(cap.cs, &st, nullptr, &g, &deps, &n_deps);
      if (e != hipSuccess) return e;
      hipGraphNode_t node = nullptr;
      e = hipGraphAddEventRecordNode(&node, g, deps, n_deps, ev);
      if (e != hipSuccess) return e;
      return hipStreamUpdateCaptureDependencies(cap.cs, &node, 1, hipStreamSetCaptureDependencies);
    };
    ge = record_node(evs.e[0]);
    int64_t i = 0;
    for (int r = 0; r < windows && ge == hipSuccess && rc == CASK_HIP_OK; r++) {
      for (int j = 0; j < k && rc == CASK_HIP_OK; j++, i++) rc = launch_spmv(*mats[i % n_mats], d_x, d_y, cap.cs);
      if (rc == CASK_HIP_OK) ge = record_node(evs.e[(size_t)r + 1]);
    }
    hipError_t ee = hipStreamEndCapture(cap.cs, &cap.g);      // always ended, whatever happened inside
    if (rc) return rc;
    if (ge != hipSuccess) {
      (void)hipGetLastError();
      return fail(CASK_HIP_ERR_RUNTIME, std::string("event-record node inside a capture: ") + hipGetErrorString(ge));
    }
    HIP_TRY(ee);
    HIP_TRY(hipGraphInstantiate(&cap.x, cap.g, nullptr, nullptr, 0));
    HIP_TRY(hipGraphLaunch(cap.x, s));
    HIP_TRY(hipGraphLaunch(cap.x, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else {
    HIP_TRY(hipEventRecord(evs.e[0], s));
    int64_t i = 0;
    for (int r = 0; r < windows; r++) {
      for (int j = 0; j < k; j++, i++) {
        int rc = launch_spmv(*mats[i % n_mats], d_x, d_y, s);
        if (rc) return rc;
      }
      HIP_TRY(hipEventRecord(evs.e[(size_t)r + 1], s));
    }
    HIP_TRY(hipEventSynchronize(evs.e[(size_t)windows]));
  }
  for (int r = 0; r < windows; r++) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, evs.e[(size_t)r], evs.e[(size_t)r + 1]));
    usec[r] = 1e3 * (double)ms;
  }
  return CASK_HIP_OK;
}

int cask_hip_spmv_dot_device(cask_hip_matrix *m, const double *d_x, double *d_y, const double *d_w, double *d_result,
                             void *stream) {
  if (!m) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
  if ((m->n_cols > 0 && !d_x) || (m->n_rows > 0 && (!d_y || !d_w)) || !d_result)
    return fail(CASK_HIP_ERR_INVALID, "NULL vector");
  if (reinterpret_cast<uintptr_t>(d_x) & 15) return fail(CASK_HIP_ERR_INVALID, "x must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plan_fuses_dot(m->plan)) {
    int rc = launch_spmv(*m, d_x, d_y, s, d_w);
    if (rc) return rc;
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(BLAS_WG), 0, s, dot_part_count(m->plan), m->plan.dot_part.p, d_result,
                       0, 0.0, (int *)nullptr, (int *)nullptr, 0);
    HIP_TRY(hipGetLastError());
    return CASK_HIP_OK;
  }
  int rc = launch_spmv(*m, d_x, d_y, s);
  if (rc) return rc;
  return cask_hip_ddot_device(m->n_rows, d_w, d_y, d_result, stream);
}

int cask_hip_csr_set_halo_sources(cask_hip_matrix *m, int32_t n_own, const uint64_t *d_src_addr) {
  if (!m) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
  if (!d_src_addr) {                                          // back to the plain layout
    m->halo_n_own = std::numeric_limits<int>::max();
    m->halo_addr = nullptr;
    return CASK_HIP_OK;
  }
  if (n_own < 1 || n_own > m->n_cols) return fail(CASK_HIP_ERR_INVALID, "n_own must lie in [1, n_cols]");
  // A handle created with AUTO may have resolved to another family (SCAN for short, heavily skewed rows): with halo
  // sources AUTO resolves to MERGE (resolve_params), so only an EXPLICITLY requested other variant is refused.
  const bool auto_variant = m->requested.variant == CASK_HIP_VARIANT_AUTO;
  if (m->nnz < 2 || (!auto_variant && m->plan.prm.variant != CASK_HIP_VARIANT_MERGE))
    return fail(CASK_HIP_ERR_INVALID, "halo sources need the MERGE variant (and at least 2 nonzeros)");
  m->halo_n_own = n_own;
  m->halo_addr = d_src_addr;
  HIP_TRY(hipSetDevice(m->device));
  int rc = build_plan(*m, m->requested);                      // seam blocks get flagged and dispatched first
  if (rc) {
    m->halo_n_own = std::numeric_limits<int>::max();
    m->halo_addr = nullptr;
    (void)build_plan(*m, m->requested);
  }
  return rc;
}

// ---- the host-vector entry point (row a6: the function the reference's clients call) ---------------------------------
// ABI <= 6: hipMemcpyAsync from the caller's pageable x, launch, hipMemcpyAsync into the caller's pageable y,
// synchronise -- 77-83 us per call on the cant-like matrix for 1 MB of vectors and an 8.5 us product (the runtime stages
// pageable copies through its own pinned buffers with one thread).  Round 2 tried "memcpy into pinned buffers + two DMAs"
// and measured 83: the two single-threaded host copies ARE the cost.  Now (include/cask_hip.h above cask_hip_spmv):
//   staged      host copies by a few threads (host_copy.hpp), a copy kernel that pulls x from the pinned buffer (no SDMA
//               start-up), y written by the product straight into pinned host memory;
//   registered  vectors inside a range the caller registered: no host copy, the same two kernels on the caller's memory.
}  // extern "C"
namespace {

__global__ void k_pull_f64(int64_t n, const double *__restrict__ src, double *__restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = __builtin_nontemporal_load(src + i);
}

std::atomic<int> g_host_entry_mode{-1};                       // -1: not decided yet (environment on first use)

int host_entry_mode() {
  int mode = g_host_entry_mode.load(std::memory_order_relaxed);
  if (mode >= 0) return mode;
  mode = CASK_HIP_HOST_ENTRY_AUTO;
  if (const char *e = std::getenv("CASK_HIP_HOST_ENTRY")) {
    const std::string v(e);
    if (v == "pageable") mode = CASK_HIP_HOST_ENTRY_PAGEABLE;
    else if (v == "staged") mode = CASK_HIP_HOST_ENTRY_STAGED;
    else if (v == "register_cache") mode = CASK_HIP_HOST_ENTRY_REGISTER_CACHE;
    else if (v != "auto" && !v.empty())
      std::fprintf(stderr, "cask_hip: CASK_HIP_HOST_ENTRY=%s is not a mode (auto | pageable | staged | register_cache): auto\n", e);
  }
  g_host_entry_mode.store(mode, std::memory_order_relaxed);
  return mode;
}

HostCopyPool &copy_pool() {
  static HostCopyPool pool([] {
    int helpers = 3;
    if (const char *e = std::getenv("CASK_HIP_HOST_THREADS")) helpers = std::max(0, std::min(15, std::atoi(e) - 1));
    const int cores = (int)std::thread::hardware_concurrency();
    return std::max(0, std::min(helpers, cores - 1));
  }());
  return pool;
}

// Host ranges the GPU may access in place: the caller's (cask_hip_host_register: reference-counted) and, in
// register_cache mode, the engine's own (keyed by pointer + length, at most REG_CACHE ranges, least recently used out).
struct HostRange {
  const char *base;
  size_t bytes;
  char *dev;
  int refs;                                                   // > 0: the caller's; 0: a cache entry
  uint64_t used;
};
constexpr size_t REG_CACHE = 16;
std::mutex g_reg_mu;
std::vector<HostRange> g_ranges;
uint64_t g_reg_clock = 0;

// (Nothing is unregistered at process exit: the runtime may be gone by the time static destructors run, and the
// caller's memory certainly may be.)

// the GPU's address for [p, p + bytes) if it lies inside a registered range (NULL if not).  callers_only: ranges the
// CALLER declared (cask_hip_host_register) -- a cache entry is only ever trusted in register_cache mode.
char *registered_device_pointer(const void *p, size_t bytes, bool callers_only) {
  const char *c = static_cast<const char *>(p);
  for (HostRange &r : g_ranges)
    if ((!callers_only || r.refs > 0) && c >= r.base && c + bytes <= r.base + r.bytes) {
      r.used = ++g_reg_clock;
      return r.dev + (c - r.base);
    }
  return nullptr;
}

int register_range(const void *ptr, size_t bytes, int refs) {
  char *dev = nullptr;
  hipError_t e = hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterMapped);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(CASK_HIP_ERR_RUNTIME, std::string("hipHostRegister: ") + hipGetErrorString(e));
  }
  e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), const_cast<void *>(ptr), 0);
  if (e != hipSuccess) {
    (void)hipHostUnregister(const_cast<void *>(ptr));
    return fail(CASK_HIP_ERR_RUNTIME, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
  }
  g_ranges.push_back(HostRange{static_cast<const char *>(ptr), bytes, dev, refs, ++g_reg_clock});
  return CASK_HIP_OK;
}

// register_cache mode: the GPU's address for this vector, registering it (and evicting the least recently used cache
// entry) if need be; NULL if the registration failed (the call then takes the staged path)
char *cached_device_pointer(const void *p, size_t bytes) {
  if (char *d = registered_device_pointer(p, bytes, false)) return d;
  size_t n_cache = 0, lru = g_ranges.size();
  for (size_t i = 0; i < g_ranges.size(); i++)
    if (g_ranges[i].refs == 0) {
      n_cache++;
      if (lru == g_ranges.size() || g_ranges[i].used < g_ranges[lru].used) lru = i;
    }
  // a range that overlaps a cache entry without lying inside it (the same buffer seen with another length): drop the old one
  const char *c = static_cast<const char *>(p);
  for (size_t i = 0; i < g_ranges.size();)
    if (g_ranges[i].refs == 0 && c < g_ranges[i].base + g_ranges[i].bytes && g_ranges[i].base < c + bytes) {
      (void)hipHostUnregister(const_cast<char *>(g_ranges[i].base));
      g_ranges.erase(g_ranges.begin() + (long)i);
      n_cache--;
      lru = g_ranges.size();
      for (size_t k = 0; k < g_ranges.size(); k++)
        if (g_ranges[k].refs == 0 && (lru == g_ranges.size() || g_ranges[k].used < g_ranges[lru].used)) lru = k;
    } else {
      i++;
    }
  if (n_cache >= REG_CACHE && lru < g_ranges.size()) {
    (void)hipHostUnregister(const_cast<char *>(g_ranges[lru].base));
    g_ranges.erase(g_ranges.begin() + (long)lru);
  }
  if (register_range(p, bytes, 0) != CASK_HIP_OK) return nullptr;
  return g_ranges.back().dev;
}

}  // namespace
extern "C" {

int cask_hip_host_entry_mode(int mode) {
  const int prev = host_entry_mode();
  if (mode < CASK_HIP_HOST_ENTRY_AUTO || mode > CASK_HIP_HOST_ENTRY_REGISTER_CACHE) return prev;
  if (prev == CASK_HIP_HOST_ENTRY_REGISTER_CACHE && mode != prev) {
    // leaving the cache mode: its entries go (unregistered while the caller's vectors are, by that mode's contract,
    // still mapped) -- no other mode may ever meet a range the caller did not declare itself
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (size_t i = 0; i < g_ranges.size();)
      if (g_ranges[i].refs == 0) {
        (void)hipHostUnregister(const_cast<char *>(g_ranges[i].base));
        g_ranges.erase(g_ranges.begin() + (long)i);
      } else {
        i++;
      }
  }
  g_host_entry_mode.store(mode, std::memory_order_relaxed);
  return prev;
}

int cask_hip_host_register(const void *ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail(CASK_HIP_ERR_INVALID, "cask_hip_host_register: empty range");
  int rc = ensure_device();
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  for (HostRange &r : g_ranges)
    if (r.base == ptr) {
      if (r.bytes < bytes) return fail(CASK_HIP_ERR_INVALID, "cask_hip_host_register: this address is registered with a shorter length");
      r.refs++;                                               // (a cache entry becomes the caller's)
      return CASK_HIP_OK;
    }
  return register_range(ptr, bytes, 1);
}

int cask_hip_host_unregister(const void *ptr) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  for (size_t i = 0; i < g_ranges.size(); i++)
    if (g_ranges[i].base == ptr && g_ranges[i].refs > 0) {
      if (--g_ranges[i].refs == 0) {
        HIP_TRY(hipHostUnregister(const_cast<void *>(ptr)));
        g_ranges.erase(g_ranges.begin() + (long)i);
      }
      return CASK_HIP_OK;
    }
  return fail(CASK_HIP_ERR_INVALID, "cask_hip_host_unregister: not a registered range");
}

int cask_hip_spmv(cask_hip_matrix *m, const double *x, double *y) {
  if (!m) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
  if ((m->n_cols > 0 && !x) || (m->n_rows > 0 && !y)) return fail(CASK_HIP_ERR_INVALID, "NULL vector");
  HIP_TRY(hipSetDevice(m->device));
  if (m->d_x.n != (size_t)m->n_cols || !m->d_x.p) HIP_TRY(m->d_x.alloc(m->n_cols));
  if (m->d_y.n != (size_t)m->n_rows || !m->d_y.p) HIP_TRY(m->d_y.alloc(m->n_rows));
  const size_t xb = (size_t)m->n_cols * sizeof(double), yb = (size_t)m->n_rows * sizeof(double);
  int mode = host_entry_mode();
  // vectors the GPU may touch in place: inside a range the caller registered, or (register_cache) any vector
  char *x_dev = nullptr, *y_dev = nullptr;
  if (mode != CASK_HIP_HOST_ENTRY_PAGEABLE && mode != CASK_HIP_HOST_ENTRY_STAGED) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (mode == CASK_HIP_HOST_ENTRY_REGISTER_CACHE) {
      if (xb) x_dev = cached_device_pointer(x, xb);
      if (yb) y_dev = cached_device_pointer(y, yb);
    } else if (!g_ranges.empty()) {
      if (xb) x_dev = registered_device_pointer(x, xb, true);
      if (yb) y_dev = registered_device_pointer(y, yb, true);
    }
  }
  // Measured (profiles/r06_host_entry.txt): 1 MB of vectors (cant-like) 70 us pageable, 44 staged with 4 host threads, 38
  // in place; 25 MB (G3_circuit-like) 504 pageable -- the runtime pipelines its staging copies with the DMA, and that much
  // is PCIe time whoever moves it -- against 636-1051 staged.  So: staged from 64 KiB to 4 MiB of vectors, pageable outside.
  if (mode == CASK_HIP_HOST_ENTRY_AUTO || mode == CASK_HIP_HOST_ENTRY_REGISTER_CACHE)
    mode = (xb + yb >= 64 * 1024 && xb + yb <= 4 * 1024 * 1024) ? CASK_HIP_HOST_ENTRY_STAGED : CASK_HIP_HOST_ENTRY_PAGEABLE;
  auto pull = [&](const double *src) {                        // x over PCIe into the device operand (16-byte aligned: the kernels' need)
    const int grid = (int)std::min<int64_t>(512, ((int64_t)m->n_cols + 255) / 256);
    hipLaunchKernelGGL(k_pull_f64, dim3(grid), dim3(256), 0, m->stream, (int64_t)m->n_cols, src, m->d_x.p);
  };
  // ---- x
  if (xb) {
    if (x_dev) {
      pull(reinterpret_cast<const double *>(x_dev));
    } else if (mode == CASK_HIP_HOST_ENTRY_STAGED) {
      HIP_TRY(m->pin_x.ensure((size_t)m->n_cols));
      copy_pool().copy(m->pin_x.p, x, xb);
      pull(m->pin_x.dev);
    } else {
      HIP_TRY(hipMemcpyAsync(m->d_x.p, x, xb, hipMemcpyHostToDevice, m->stream));
    }
  }
  // ---- the product; y goes where the CPU will read it
  double *y_target = m->d_y.p;
  if (yb && y_dev) y_target = reinterpret_cast<double *>(y_dev);
  else if (yb && mode == CASK_HIP_HOST_ENTRY_STAGED) {
    HIP_TRY(m->pin_y.ensure((size_t)m->n_rows));
    y_target = m->pin_y.dev;
  }
  int rc = launch_spmv(*m, m->d_x.p, y_target, m->stream);
  if (rc) return rc;
  if (yb && y_target == m->d_y.p) HIP_TRY(hipMemcpyAsync(y, m->d_y.p, yb, hipMemcpyDeviceToHost, m->stream));
  // How the CPU learns that y is there.  hipStreamSynchronize waits for the queue's completion signal; when y already
  // lands in host memory (staged / in place) a word stored behind the product and polled by the caller gets there
  // earlier (profiles/r06_host_entry.txt).  The poll is bounded: a launch that faulted never stores the word, and
  // hipStreamSynchronize is what reports it.  CASK_HIP_HOST_DONE=sync|word: development A/B.
  bool waited = false;
  if (host_done_word() && yb && y_target != m->d_y.p && m->pin_done.ensure(1) == hipSuccess) {
    const unsigned long long seq = ++m->done_seq;
    hipLaunchKernelGGL(k_done_word, dim3(1), dim3(1), 0, m->stream, reinterpret_cast<unsigned long long *>(m->pin_done.dev), seq);
    if (hipGetLastError() == hipSuccess)
      waited = poll_word(reinterpret_cast<const volatile unsigned long long *>(m->pin_done.p), seq, 20);
  }
  if (!waited) HIP_TRY(hipStreamSynchronize(m->stream));
  if (yb && !y_dev && y_target != m->d_y.p) copy_pool().copy(y, m->pin_y.p, yb, 1);
  return CASK_HIP_OK;
}

static int ensure_transpose(cask_hip_matrix *m) {
  if (m->transpose) return CASK_HIP_OK;
  // Explicit transpose by counting sort on the host (once); rows of A^T come
  // out with ascending column (= original row) order.
  std::vector<int> ci((size_t)m->nnz);
  std::vector<double> val((size_t)m->nnz);
  if (m->nnz) {
    HIP_TRY(hipMemcpy(ci.data(), m->d_ci, (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(val.data(), m->d_val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost));
  }
  std::vector<int> trp((size_t)m->n_cols + 1, 0), tci((size_t)m->nnz);
  std::vector<double> tval((size_t)m->nnz);
  for (int64_t k = 0; k < m->nnz; k++) trp[ci[k] + 1]++;
  for (int c = 0; c < m->n_cols; c++) trp[c + 1] += trp[c];
  std::vector<int> fill(trp.begin(), trp.end() - 1);
  for (int r = 0; r < m->n_rows; r++)
    for (int k = m->h_rp[r]; k < m->h_rp[r + 1]; k++) {
      const int dst = fill[ci[k]]++;
      tci[dst] = r;
      tval[dst] = val[k];
    }
  cask_hip_matrix *t = nullptr;
  int rc = cask_hip_csr_create(m->n_cols, m->n_rows, m->nnz, trp.data(), tci.data(), tval.data(), &m->requested, &t);
  if (rc) return rc;
  m->transpose.reset(t);
  return CASK_HIP_OK;
}

int cask_hip_spmv_transpose_device(cask_hip_matrix *m, const double *d_x, double *d_y, void *stream) {
  if (!m) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
  HIP_TRY(hipSetDevice(m->device));
  int rc = ensure_transpose(m);
  if (rc) return rc;
  return cask_hip_spmv_device(m->transpose.get(), d_x, d_y, stream);
}

int cask_hip_spmv_time(cask_hip_matrix *m, const double *d_x, double *d_y, int32_t warmup, int32_t iters,
                       double *usec_median, double *usec_min) {
  if (!m || iters <= 0) return fail(CASK_HIP_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(m->device));
  for (int i = 0; i < warmup; i++) {
    int rc = launch_spmv(*m, d_x, d_y, m->stream);
    if (rc) return rc;
  }
  std::vector<DevEvent> ev((size_t)iters + 1);
  for (auto &e : ev) HIP_TRY(e.create());
  HIP_TRY(hipEventRecord(ev[0], m->stream));
  for (int i = 0; i < iters; i++) {
    int rc = launch_spmv(*m, d_x, d_y, m->stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev[i + 1], m->stream));
  }
  HIP_TRY(hipStreamSynchronize(m->stream));
  std::vector<double> us((size_t)iters);
  for (int i = 0; i < iters; i++) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    us[i] = ms * 1e3;
  }
  std::sort(us.begin(), us.end());
  if (usec_median) *usec_median = us[us.size() / 2];
  if (usec_min) *usec_min = us[0];
  return CASK_HIP_OK;
}

// Best-of-`reps` microseconds per launch of a HIP graph of `k` launches rotating over `mats` (one handle: warm).
static int time_graph(std::vector<cask_hip_matrix *> &mats, const double *x, double *y, int k, int warm_replays, int reps,
                      double *usec) {
  cask_hip_matrix *m0 = mats[0];
  hipStream_t s = m0->stream;
  for (size_t i = 0; i < mats.size(); i++) {                  // eager once: first-use work stays out of the capture
    int rc = launch_spmv(*mats[i], x, y, s);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(s));
  struct GraphPair {                                          // destroyed on every way out
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ~GraphPair() {
      if (exec) (void)hipGraphExecDestroy(exec);
      if (graph) (void)hipGraphDestroy(graph);
    }
  } g;
  HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = CASK_HIP_OK;
  for (int i = 0; i < k && rc == CASK_HIP_OK; i++) rc = launch_spmv(*mats[(size_t)i % mats.size()], x, y, s);
  hipError_t e = hipStreamEndCapture(s, &g.graph);
  if (rc) return rc;
  if (e != hipSuccess) return fail(CASK_HIP_ERR_RUNTIME, std::string("graph capture: ") + hipGetErrorString(e));
  e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
  if (e != hipSuccess) return fail(CASK_HIP_ERR_RUNTIME, std::string("graph instantiate: ") + hipGetErrorString(e));
  DevEvent e0, e1;
  HIP_TRY(e0.create()); HIP_TRY(e1.create());
  double best = 0.0;
  for (int r = 0; r < warm_replays + reps; r++) {
    HIP_TRY(hipEventRecord(e0, s));
    HIP_TRY(hipGraphLaunch(g.exec, s));
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / k;
    if (r >= warm_replays && (best == 0.0 || us < best)) best = us;
  }
  *usec = best;
  return CASK_HIP_OK;
}

int cask_hip_tune(cask_hip_matrix *m, const int32_t *variants, int32_t n_variants, const int32_t *lanes,
                  int32_t n_lanes, const int32_t *tiles, int32_t n_tiles, const int32_t *wg_sizes,
                  int32_t n_wg_sizes, const int32_t *items, int32_t n_items, int32_t warmup, int32_t iters,
                  cask_hip_tune_point *results, int32_t max_results, int32_t *n_results, int32_t *best_index) {
  if (!m) return fail(CASK_HIP_ERR_INVALID, "matrix is NULL");
  static const int32_t def_variants[] = {CASK_HIP_VARIANT_VECTOR, CASK_HIP_VARIANT_MERGE, CASK_HIP_VARIANT_MERGE_WAVE,
                                         CASK_HIP_VARIANT_SCAN, CASK_HIP_VARIANT_SLICE};
  static const int32_t def_lanes[] = {4, 8, 16, 32};
  static const int32_t def_tiles[] = {-1, 1024, 4096};
  static const int32_t def_wg[] = {256, 512};
  static const int32_t def_items[] = {4, 8};
  if (!variants || n_variants <= 0) { variants = def_variants; n_variants = 5; }
  if (!lanes || n_lanes <= 0) { lanes = def_lanes; n_lanes = 4; }
  if (!tiles || n_tiles <= 0) { tiles = def_tiles; n_tiles = 3; }
  if (!wg_sizes || n_wg_sizes <= 0) { wg_sizes = def_wg; n_wg_sizes = 2; }
  if (!items || n_items <= 0) { items = def_items; n_items = 2; }
  if (warmup < 0) warmup = 0;
  warmup = std::min(warmup, 3);
  HIP_TRY(hipSetDevice(m->device));
  // A block with halo sources is tuned with them detached: its extended columns then index the caller-sized x of the
  // sweep (own entries + a local halo copy), the shapes are ranked without the remote loads, and the halo comes back
  // on the winner.
  const int saved_n_own = m->halo_n_own;
  const uint64_t *saved_halo = m->halo_addr;
  struct HaloGuard {
    cask_hip_matrix *m; int n_own; const uint64_t *addr;
    ~HaloGuard() {
      if (addr && !m->halo_addr) {
        m->halo_n_own = n_own;
        m->halo_addr = addr;
        (void)build_plan(*m, m->requested);
      }
    }
  } halo_guard{m, saved_n_own, saved_halo};
  if (saved_halo) {
    m->halo_n_own = std::numeric_limits<int>::max();
    m->halo_addr = nullptr;
  }
  DevBuf<double> x, y;
  HIP_TRY(x.alloc(m->n_cols));
  HIP_TRY(y.alloc(m->n_rows));
  {
    std::vector<double> hx((size_t)m->n_cols);
    for (int i = 0; i < m->n_cols; i++) hx[i] = 0.25 * i;     // the reference's test operand (test_spmv.cpp:27-28)
    if (m->n_cols) HIP_TRY(hipMemcpy(x.p, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  const cask_hip_params saved = m->requested;
  cask_hip_csr_info info;
  cask_hip_csr_get_info(m, &info);
  // cold timing: device copies of the matrix that together exceed 2x the Infinity Cache (as cask_amd/dse.py did in
  // round 1 -- now the one implementation); a matrix under 4 MB is not an HBM workload, it is timed warm only
  const int64_t matrix_bytes = 12 * m->nnz + 4 * ((int64_t)m->n_rows + 1);
  int copies = 1;
  if (matrix_bytes >= (4 << 20)) copies = (int)std::min<int64_t>(64, std::max<int64_t>(2, (2 * (int64_t)(256 << 20) + matrix_bytes - 1) / matrix_bytes + 1));
  if (const char *e = std::getenv("CASK_HIP_TUNE_COPIES")) copies = std::max(1, std::atoi(e));
  std::vector<DevBuf<int>> ci_copy((size_t)copies - 1);
  std::vector<DevBuf<double>> val_copy((size_t)copies - 1);
  std::vector<std::unique_ptr<cask_hip_matrix>> extra;
  std::vector<cask_hip_matrix *> rot{m}, one{m};
  for (int c = 0; c + 1 < copies; c++) {
    HIP_TRY(ci_copy[c].alloc((size_t)m->nnz));
    HIP_TRY(val_copy[c].alloc((size_t)m->nnz));
    HIP_TRY(hipMemcpy(ci_copy[c].p, m->d_ci, (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(val_copy[c].p, m->d_val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToDevice));
    cask_hip_matrix *h = nullptr;
    int rc = cask_hip_csr_create_device(m->n_rows, m->n_cols, m->nnz, m->d_rp, ci_copy[c].p, val_copy[c].p, &saved, &h);
    if (rc) return rc;
    extra.emplace_back(h);
    rot.push_back(h);
  }
  const int k_cold = iters > 0 ? std::max(iters, copies) : std::max(24, 4 * copies);
  const int k_warm = iters > 0 ? iters : 48;
  int count = 0, best = -1;
  double best_us = 0.0;
  cask_hip_params best_params{};
  // Pruning (r4).  The RESULTS keep the reference's odometer order (first range fastest, Utils.hpp:173-192; unlike
  // Dse.cpp:40-47 the last point is evaluated too) -- the MEASUREMENTS do not follow it.  Every family's points are
  // ranked by a prior (VECTOR: the lane count nearest the mean row length; the merge families: AUTO's shape), the
  // two best guesses of every family are measured first, and only then may a family be dropped: when even those two
  // are more than 1.5x behind the incumbent AND its times are not still improving (its latest point was not its best
  // by > 10 %).  Round 3 pruned in odometer order, i.e. after L = 4 and L = 8 for the row-mapped VECTOR family --
  // its worst points on 64-nonzero rows -- and never saw L = 32.  Pruned points are reported with valid = 0,
  // usec = -1; CASK_HIP_TUNE_NO_PRUNE=1 measures everything.
  const bool prune = std::getenv("CASK_HIP_TUNE_NO_PRUNE") == nullptr;
  struct Cand { cask_hip_tune_point pt; int family; double prior; int rank; bool done; };
  std::vector<Cand> cands;
  const double mean_row = m->n_rows ? std::max(1.0, (double)m->nnz / m->n_rows) : 1.0;
  auto l2 = [](double v) { return std::log2(std::max(v, 1.0)); };
  // SLICE has something to offer only where short rows exist: with fewer than a quarter of the rows at <= 8 nonzeros its
  // plan is a SCAN plan with a second copy of the streams (and costs a host sort per point): not a candidate there
  int64_t short_rows = 0;
  for (int r = 0; r < m->n_rows; r++) short_rows += m->h_rp[(size_t)r + 1] - m->h_rp[(size_t)r] <= SLICE_KMAX;
  const bool slice_worth_a_look = 4 * short_rows >= (int64_t)m->n_rows;
  // lanes only matter for VECTOR and items only for the merge families: irrelevant repeats are skipped.
  for (int iv = 0; iv < n_items; iv++)
    for (int iw = 0; iw < n_wg_sizes; iw++)
      for (int it = 0; it < n_tiles; it++)
        for (int il = 0; il < n_lanes; il++)
          for (int iva = 0; iva < n_variants; iva++) {
            const int variant = variants[iva];
            // SCAN: items per thread like the merge kernels, its tile axis is its LDS x window; SLICE (r6): the lanes axis
            // carries K (1..8: the longest row a slice thread takes), items / tile / workgroup those of its SCAN blocks
            const bool slice = variant == CASK_HIP_VARIANT_SLICE;
            const bool is_merge = variant == CASK_HIP_VARIANT_MERGE || variant == CASK_HIP_VARIANT_MERGE_WAVE ||
                                  variant == CASK_HIP_VARIANT_SCAN || slice;
            // a block with halo sources runs the MERGE variant only: a winner from another family could not be applied
            if (saved_halo && variant != CASK_HIP_VARIANT_MERGE) continue;
            if (variant == CASK_HIP_VARIANT_VECTOR && iv != 0) continue;
            if (is_merge && !slice && il != 0) continue;
            if (slice && (!slice_worth_a_look || lanes[il] < 1 || lanes[il] > SLICE_KMAX || (items[iv] != 4 && items[iv] != 8))) continue;
            if (variant == CASK_HIP_VARIANT_MERGE_WAVE && it != 0) continue;   // no x tile in that kernel
            Cand c{};
            c.pt.params = saved;
            c.pt.params.variant = variant;
            c.pt.params.lanes_per_row = (variant == CASK_HIP_VARIANT_VECTOR || slice) ? lanes[il] : 0;
            c.pt.params.items_per_thread = is_merge ? items[iv] : 0;
            c.pt.params.tile_width = tiles[it];
            c.pt.params.wg_size = wg_sizes[iw];
            c.pt.valid = 0;
            c.pt.usec = -1.0;                                 // "pruned, not measured" until it is
            c.family = variant & 7;
            // the prior: distance from the family's best guess (ties: odometer order)
            const double d_wg = std::fabs(l2(wg_sizes[iw]) - l2(256));
            const double tile_ref = variant == CASK_HIP_VARIANT_SCAN ? 4096 : (mean_row >= 16 ? 1024 : 2048);
            const double d_tile = tiles[it] <= 0 ? 1.5 : std::fabs(l2(tiles[it]) - l2(tile_ref)) * 0.5;
            if (variant == CASK_HIP_VARIANT_VECTOR)
              c.prior = 4.0 * std::fabs(l2(lanes[il]) - l2(std::min(mean_row, 64.0) / 2.0)) + d_wg + d_tile;
            else
              c.prior = 2.0 * std::fabs(l2(items[iv]) - l2(8)) + d_wg + d_tile + (slice ? std::fabs(l2(lanes[il]) - l2(2)) : 0.0);
            cands.push_back(c);
          }
  count = (int)cands.size();
  // rank within the family
  for (int f = 0; f < 8; f++) {
    std::vector<int> idx;
    for (int i = 0; i < count; i++)
      if (cands[i].family == f) idx.push_back(i);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cands[a].prior < cands[b].prior; });
    for (size_t r = 0; r < idx.size(); r++) cands[idx[r]].rank = (int)r;
  }
  // schedule: rank 0 of every family, rank 1 of every family, then the rest in odometer order
  std::vector<int> order;
  for (int r = 0; r < 2; r++)
    for (int i = 0; i < count; i++)
      if (cands[i].rank == r) order.push_back(i);
  const size_t n_prior = order.size();
  for (int i = 0; i < count; i++)
    if (cands[i].rank >= 2) order.push_back(i);
  int fam_points[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double fam_best[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fam_last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double fam_prev_best[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t oi = 0; oi < order.size(); oi++) {
    Cand &c = cands[order[oi]];
    const int f = c.family;
    // (SLICE -- a measured loss on every matrix of round 6, and a host sort per point -- goes when its two best guesses are 10 %
    // behind, the other families at 50 %)
    const double behind = f == (CASK_HIP_VARIANT_SLICE & 7) ? 1.10 : 1.5;
    if (prune && oi >= n_prior && fam_points[f] >= 2 && best >= 0 && fam_best[f] > behind * best_us) {
      // ... unless the family is still on its way down: its latest point beat everything it had by > 10 %
      const bool improving = fam_last[f] == fam_best[f] && fam_prev_best[f] > 0 && fam_last[f] < 0.9 * fam_prev_best[f];
      if (!improving) continue;                               // stays valid = 0, usec = -1
    }
    cask_hip_tune_point &pt = c.pt;
    int rc = build_plan(*m, pt.params);
    for (size_t k = 1; k < rot.size() && rc == CASK_HIP_OK; k++) rc = clone_plan(*rot[k], *m);
    if (rc == CASK_HIP_OK && m->plan.prm.variant == CASK_HIP_VARIANT_VECTOR &&
        m->plan.prm.wg_size < m->plan.prm.lanes_per_row)
      rc = CASK_HIP_ERR_INVALID;
    pt.usec = 0.0;                                            // measured (or rejected), not pruned
    if (rc == CASK_HIP_OK) {
      double cold = 0, warm = 0;
      rc = time_graph(one, x.p, y.p, k_warm, warmup, 3, &warm);
      if (rc == CASK_HIP_OK) rc = copies > 1 ? time_graph(rot, x.p, y.p, k_cold, warmup, 3, &cold) : CASK_HIP_OK;
      if (rc == CASK_HIP_OK) {
        if (copies == 1) cold = warm;
        pt.params = m->plan.prm;
        pt.usec = cold;
        pt.usec_warm = warm;
        pt.copies = copies;
        pt.gflops = cold > 0 ? 2.0 * m->nnz / cold * 1e-3 : 0.0;
        pt.gbytes_per_s = cold > 0 ? info.algorithmic_bytes / cold * 1e-3 : 0.0;
        pt.valid = 1;
        if (best < 0 || cold < best_us) { best = order[oi]; best_us = cold; best_params = pt.params; }
        fam_points[f]++;
        fam_prev_best[f] = fam_best[f];
        if (fam_points[f] == 1 || cold < fam_best[f]) fam_best[f] = cold;
        fam_last[f] = cold;
      }
    }
  }
  if (results)
    for (int i = 0; i < count && i < max_results; i++) results[i] = cands[i].pt;
  if (n_results) *n_results = std::min(count, max_results);
  if (best_index) *best_index = (best < max_results) ? best : -1;
  if (saved_halo) {                                           // back on, before the winner's plan is built
    m->halo_n_own = saved_n_own;
    m->halo_addr = saved_halo;
  }
  int rc;
  if (best >= 0) {
    rc = build_plan(*m, best_params);
    if (rc == CASK_HIP_OK) m->requested = best_params;
  } else {
    rc = build_plan(*m, saved);
  }
  return rc;
}

// ---------------------------------------------------------------------- BLAS-1
int cask_hip_ddot_device(int64_t n, const double *d_x, const double *d_y, double *d_result, void *stream) {
  if (n < 0 || !d_result || (n > 0 && (!d_x || !d_y))) return fail(CASK_HIP_ERR_INVALID, "bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // partial buffers: one per (device, stream) of the calling thread, kept for reuse -- two dots in flight on
  // different streams must not share their partials, and a change of device must not leak the old buffer.
  // (A stream handle reused after hipStreamDestroy simply reuses the buffer.)
  struct Scratch { int dev; hipStream_t s; double *p; };
  static thread_local std::vector<Scratch> scratches;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  double *scratch = nullptr;
  for (const Scratch &sc : scratches)
    if (sc.dev == dev && sc.s == s) scratch = sc.p;
  if (!scratch) {
    if (scratches.size() >= 64) {                             // bound the cache: drop this device's oldest entries
      for (auto it = scratches.begin(); it != scratches.end();)
        if (it->dev == dev) { HIP_TRY(hipStreamSynchronize(it->s)); (void)hipFree(it->p); it = scratches.erase(it); } else ++it;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&scratch), BLAS_MAX_PARTIALS * sizeof(double)));
    scratches.push_back(Scratch{dev, s, scratch});
  }
  const int grid = blas_grid(n);
  hipLaunchKernelGGL(k_dot_partial, dim3(grid), dim3(BLAS_WG), 0, s, n, d_x, d_y, scratch, (const int *)nullptr);
  hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(BLAS_WG), 0, s, grid, scratch, d_result, 0, 0.0, (int *)nullptr,
                     (int *)nullptr, 0);
  HIP_TRY(hipGetLastError());
  return CASK_HIP_OK;
}

int cask_hip_daxpy_device(int64_t n, double sign, const double *d_num, const double *d_den, const double *d_x,
                          double *d_y, void *stream) {
  if (n < 0 || (n > 0 && (!d_x || !d_y))) return fail(CASK_HIP_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_axpy, dim3(blas_grid(n)), dim3(BLAS_WG), 0, static_cast<hipStream_t>(stream), n, sign, d_num,
                     d_den, d_x, d_y, (const int *)nullptr);
  HIP_TRY(hipGetLastError());
  return CASK_HIP_OK;
}

int cask_hip_daxpby_device(int64_t n, double alpha, const double *d_x, double sign, double beta,
                           const double *d_num, const double *d_den, double *d_y, void *stream) {
  if (n < 0 || (n > 0 && (!d_x || !d_y))) return fail(CASK_HIP_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_axpby, dim3(blas_grid(n)), dim3(BLAS_WG), 0, static_cast<hipStream_t>(stream), n, alpha, d_x,
                     sign, beta, d_num, d_den, d_y, (const int *)nullptr);
  HIP_TRY(hipGetLastError());
  return CASK_HIP_OK;
}

// ---------------------------------------------------------------------- solvers
static int solver_common_checks(cask_hip_matrix *m, const double *rhs, double *x, int32_t maxiters, double tol) {
  if (!m || !rhs || !x) return fail(CASK_HIP_ERR_INVALID, "NULL argument");
  if (m->n_rows != m->n_cols) return fail(CASK_HIP_ERR_INVALID, "solver needs a square matrix");
  if (maxiters < 0 || !(tol >= 0)) return fail(CASK_HIP_ERR_INVALID, "bad maxiters/tol");
  return CASK_HIP_OK;
}

// ---- CG / BiCG on device vectors, single GPU or one rank of a row-sharded solve -------------------------------
// Composed mode (MERGE plans): a pass is the product launch(es) -- which compose the direction p = r + beta p_old
// while staging their x windows, store it for the rows they own, apply the solution update the previous pass owes
// and leave the shares of p.Ap behind (SolverPass) -- plus ONE update launch (r -= alpha Ap and the shares of
// r.r): 2 launches per CG pass, 3 per BiCG pass.  Classic mode (any plan; sharded: with an operand-exchange
// callback): product, x/r update, p update as separate launches.  Row-sharded: the partial sums are added to
// one scalar per rank and all-reduced through the caller's callback between the launches; those collectives are
// also what orders a rank's stores to its vector slices against the peers' in-kernel halo loads.
//
// Cross-rank hazards of a composed pass i with in-kernel halos (K1_i = product launch(es), K2_i = update launch,
// AR1_i = all-reduce of p.Ap / pt.q, AR2_i = all-reduce of r.r (and rt.r); an all-reduce completes on a rank only
// after every rank has enqueued its contribution behind its own preceding kernels):
//   * K1_i reads the peers' r (rt): written by their K2_{i-1}, which precedes their contribution to AR2_{i-1},
//     which K1_i waits for (it consumes the reduced scalars).               [pass 0: r is final before the r.r
//     all-reduce of the set-up]
//   * K1_i reads the peers' p_old = P[(i-1)&1]: written by their K1_{i-1}, two collectives earlier.
//   * K1_i WRITES its own P[i&1], which peers read as p_old in THEIR K1_{i-1}: those launches precede the peers'
//     contributions to AR1_{i-1}, which completed before this rank's K2_{i-1}, hence before K1_i.
//   * K2_i WRITES this rank's r: peers read it in their K1_i, before their contribution to AR1_i, which this rank's
//     K2_i waits for.
// A classic pass with in-kernel halos has no collective between its p update and the next product: it spends a
// third one as a fence.  Vector slots peers read are written with system-scope write-through stores and read with
// system-scope loads.
namespace {

struct SolveSetup {
  cask_hip_matrix *A = nullptr, *At = nullptr;
  int kind = CASK_HIP_SOLVER_CG;
  bool composed = false, sharded = false;
  cask_hip_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  cask_hip_exchange_fn exchange = nullptr;
  void *exchange_user = nullptr;
  int64_t n = 0, S = 0, n_full = 0;
  int sys_scope = 0;
};

// vector slots of the (shared or private) allocation, `S` doubles apart
enum { SLOT_R = 0, SLOT_P0 = 1, SLOT_P1 = 2, SLOT_RT = 3, SLOT_PT0 = 4, SLOT_PT1 = 5 };
// device scalars
enum { SC_RS0 = 0, SC_RS1 = 1, SC_ALPHA = 2, SC_DOT = 4, SC_RR = 5, SC_RHO = 6, SC_COUNT = 8 };

int run_allreduce(const SolveSetup &st, double *d, int count, hipStream_t s) {
  if (!st.allreduce) return CASK_HIP_OK;
  if (st.allreduce(d, count, s, st.allreduce_user) != 0)
    return fail(CASK_HIP_ERR_RUNTIME, "the all-reduce callback of the sharded solver failed");
  return CASK_HIP_OK;
}

}  // namespace

int cask_hip_solve_device(cask_hip_matrix *m, cask_hip_matrix *mt_in, const cask_hip_solver_config *cfg_in,
                          const double *d_rhs, double *d_x, int32_t maxiters, double tol, int32_t *iterations,
                          int32_t *converged, double *usec_per_iteration, void *stream) {
  if (!m || !d_rhs || !d_x) return fail(CASK_HIP_ERR_INVALID, "NULL argument");
  if (maxiters < 0 || !(tol >= 0)) return fail(CASK_HIP_ERR_INVALID, "bad maxiters/tol");
  // the update kernels read and write x and the right-hand side in 16-byte pairs
  if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_rhs)) & 15)
    return fail(CASK_HIP_ERR_INVALID, "d_x and d_rhs must be 16-byte aligned");
  cask_hip_solver_config cfg{};
  if (cfg_in) cfg = *cfg_in;
  if (cfg.kind == 0) cfg.kind = CASK_HIP_SOLVER_CG;
  if (cfg.kind != CASK_HIP_SOLVER_CG && cfg.kind != CASK_HIP_SOLVER_BICG)
    return fail(CASK_HIP_ERR_INVALID, "unknown solver kind");
  HIP_TRY(hipSetDevice(m->device));
  SolveSetup st;
  st.A = m;
  st.kind = cfg.kind;
  st.allreduce = cfg.allreduce;
  st.allreduce_user = cfg.allreduce_user;
  st.exchange = cfg.exchange;
  st.exchange_user = cfg.exchange_user;
  st.sharded = cfg.allreduce != nullptr;
  st.n = m->n_rows;
  const bool bicg = cfg.kind == CASK_HIP_SOLVER_BICG;
  const bool has_halo = m->halo_addr != nullptr;
  // the operand of a product: n_rows own entries (+ halo columns behind them, or the gathered vector)
  if (!has_halo && !st.exchange && m->n_rows != m->n_cols) return fail(CASK_HIP_ERR_INVALID, "solver needs a square matrix");
  if (has_halo && m->halo_n_own != m->n_rows)
    return fail(CASK_HIP_ERR_INVALID, "a sharded solver block owns as many columns as rows");
  if (st.exchange && (cfg.n_full < m->n_cols)) return fail(CASK_HIP_ERR_INVALID, "n_full must cover the block's columns");
  if (bicg) {
    if (mt_in) {
      st.At = mt_in;
    } else {
      if (has_halo || st.exchange) return fail(CASK_HIP_ERR_INVALID, "a sharded BiCG needs the row block of A^T");
      int rc = ensure_transpose(m);
      if (rc) return rc;
      st.At = m->transpose.get();
    }
    if (st.At->n_rows != m->n_rows) return fail(CASK_HIP_ERR_INVALID, "A and A^T blocks differ in their row count");
  }
  const bool can_compose = plan_fuses_dot(m->plan) && (!bicg || plan_fuses_dot(st.At->plan)) && !st.exchange;
  if (cfg.mode == CASK_HIP_SOLVER_COMPOSED && !can_compose)
    return fail(CASK_HIP_ERR_INVALID, "composed passes need MERGE plans with the dot epilogue (and no exchange callback)");
  // AUTO: measured on one GPU (profiles/r02_solver_modes.txt) the composed pass LOSES to the classic one --
  // G3_circuit-like CG 51.5 vs 46.7 us, atmosmodd-like BiCG 120 vs 80 us per pass: the bytes of the p update do
  // not go away, they move into the product launch (own-row reads and writes +4.8 us, the second x window
  // +2.2 us) and every workgroup pays for summing the previous launch's partials first (+4.6 us), which is more
  // than the launch boundary saved.  A row-sharded solve with in-kernel halos is where it pays: the scalars
  // arrive all-reduced (no sums), and the two all-reduces of a pass are the only cross-rank ordering it needs,
  // whereas a classic pass needs a third collective as a fence behind its p update.
  st.composed = cfg.mode == CASK_HIP_SOLVER_COMPOSED   ? true
                : cfg.mode == CASK_HIP_SOLVER_CLASSIC ? false
                                                      : (can_compose && st.sharded && cfg.d_shared_base != nullptr);
  // (the ranks of a sharded solve must all take the same form: the caller agrees on `mode` collectively when
  // their plans may differ -- cask_amd/dist.py does)
  const int64_t n = st.n;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n_slots = bicg ? 6 : 3;
  // vectors peers may read live in the caller's shared allocation, `stride` doubles per slot
  double *base = cfg.d_shared_base;
  st.S = cfg.stride;
  if (!base) {
    // private vectors: rows rounded up to 32 -- or, with an exchange callback, the caller's stride: the padded
    // all-gather (cask_hip_rccl_comm_set_stride) reads `stride` doubles from every operand slot
    if (st.exchange && cfg.stride > 0) {
      if (cfg.stride < n || (cfg.stride & 1)) return fail(CASK_HIP_ERR_INVALID, "stride must be even and at least the block's row count");
    } else {
      st.S = (n + 31) / 32 * 32;
      if (st.S == 0) st.S = 32;
    }
  } else {
    if (st.S < n || (st.S & 1)) return fail(CASK_HIP_ERR_INVALID, "stride must be even and at least the block's row count");
    if (reinterpret_cast<uintptr_t>(base) & 15) return fail(CASK_HIP_ERR_INVALID, "shared base must be 16-byte aligned");
    st.sys_scope = 1;
  }
  if (!m->solver_ws) m->solver_ws.reset(new SolverWorkspace);
  SolverWorkspace &ws = *m->solver_ws;
  const int64_t want_full = st.exchange ? cfg.n_full : 0;
  if (ws.n != n || ws.S != st.S || ws.n_slots != n_slots || ws.shared_vec != (cfg.d_shared_base != nullptr) ||
      ws.n_full != want_full) {
    ws.n = -1;                                                // (re)build; a failed allocation leaves it invalid
    if (!cfg.d_shared_base) {
      HIP_TRY(ws.own_vec.alloc((size_t)(st.S * n_slots)));
      HIP_TRY(hipMemsetAsync(ws.own_vec.p, 0, (size_t)(st.S * n_slots) * sizeof(double), static_cast<hipStream_t>(stream)));   // padded tails stay finite
    } else {
      ws.own_vec.release();
    }
    HIP_TRY(ws.q.alloc(n));
    if (bicg) HIP_TRY(ws.qt.alloc(n)); else ws.qt.release();
    HIP_TRY(ws.part_a.alloc(BLAS_MAX_PARTIALS)); HIP_TRY(ws.part_b.alloc(BLAS_MAX_PARTIALS));
    HIP_TRY(ws.part_c.alloc(BLAS_MAX_PARTIALS));
    HIP_TRY(ws.scal.alloc(SC_COUNT)); HIP_TRY(ws.flags.alloc(2));
    if (want_full) {
      HIP_TRY(ws.x_full.alloc((size_t)want_full));
      if (bicg) HIP_TRY(ws.x_full_t.alloc((size_t)want_full)); else ws.x_full_t.release();
    } else {
      ws.x_full.release();
      ws.x_full_t.release();
    }
    ws.n = n; ws.S = st.S; ws.n_slots = n_slots; ws.shared_vec = cfg.d_shared_base != nullptr; ws.n_full = want_full;
  }
  if (!base) base = ws.own_vec.p;
  auto slot = [&](int k) { return base + (int64_t)k * st.S; };
  double *r = slot(SLOT_R), *rt = bicg ? slot(SLOT_RT) : nullptr;
  DevBuf<double> &q = ws.q, &qt = ws.qt, &part_a = ws.part_a, &part_b = ws.part_b, &part_c = ws.part_c, &scal = ws.scal,
                 &x_full = ws.x_full, &x_full_t = ws.x_full_t;
  DevBuf<int> &flags = ws.flags;
  int *done = flags.p, *iters = flags.p + 1;                   // (zeroed, with the scalars, by the first launch below: k_copy2)
  const SolverShape shape = solver_shape(n, st.sharded);
  const int g = shape.grid;
  const dim3 bg(g), bw(shape.wg);
  const double tol2 = tol * tol;
  int rc;

  // the product y = Op * v for a vector v in slot k (or, classic sharded mode, gathered through the callback)
  auto product = [&](cask_hip_matrix *op, int k, double *y, const double *w, DevBuf<double> &full) -> int {
    const double *operand = slot(k);
    if (st.exchange) {
      if (st.exchange(slot(k), full.p, s, st.exchange_user) != 0)
        return fail(CASK_HIP_ERR_RUNTIME, "the operand-exchange callback of the sharded solver failed");
      operand = full.p;
    }
    op->halo_shift = (int64_t)k * st.S * 8;
    const int rc2 = launch_spmv(*op, operand, y, s, w);
    op->halo_shift = 0;
    return rc2;
  };

  // ---- r = b - A x0 ; p = r ; rsold = r.r   (SparseLinearSolvers.hpp:189-198); BiCG: rt = r, rho = rt.r
  // x0 travels through slot P1 (peers read it there); in composed mode that slot is pass 0's "old direction",
  // which beta = 0 wipes out, so any finite content does.
  // (r6: the set-up is two launches besides the product -- k_copy2, k_solver_residual -- instead of three to five runtime copies,
  //  an axpby and a dot launch: ~30-60 us of a solve, profiles/r06_update_launches.txt (6))
  hipLaunchKernelGGL(k_copy2, bg, bw, 0, s, n, d_x, slot(SLOT_P1), bicg ? slot(SLOT_PT1) : (double *)nullptr, scal.p, (int)SC_COUNT,
                     flags.p, 2);
  if (st.sharded && !st.exchange) {                           // every rank's x0 is in place before anyone's halo loads
    rc = run_allreduce(st, scal.p + SC_DOT, 1, s);
    if (rc) return rc;
  }
  rc = product(m, SLOT_P1, q.p, nullptr, x_full);
  if (rc) return rc;
  // r = b - q ; rt = r (BiCG) ; classic passes keep p (and pt) materialised in slot P0 ; the shares of r.r
  hipLaunchKernelGGL(k_solver_residual, bg, bw, 0, s, n, d_rhs, (const double *)q.p, r, bicg ? rt : (double *)nullptr,
                     !st.composed ? slot(SLOT_P0) : (double *)nullptr,
                     (!st.composed && bicg) ? slot(SLOT_PT0) : (double *)nullptr, part_a.p);
  hipLaunchKernelGGL(k_dot_final, dim3(1), bw, 0, s, g, part_a.p, scal.p + SC_RS0, 0, 0.0, (int *)nullptr, (int *)nullptr, 0);
  HIP_TRY(hipGetLastError());
  rc = run_allreduce(st, scal.p + SC_RS0, 1, s);              // also: every rank's r (and p) is final before the first pass
  if (rc) return rc;

  // the caller reduces by peer stores (cask_hip_push_allreduce): the local sum and the exchange become one launch
  const bool peer_reduce = st.allreduce == &cask_hip_push_allreduce;
  DevEvent e0, e1;
  HIP_TRY(e0.create()); HIP_TRY(e1.create());
  HIP_TRY(hipEventRecord(e0, s));
  const int check_every = 16;
  int h_flags[2] = {0, 0};
  const bool can_defer = !st.composed && !st.sharded && ensure_checkpoint(m);
  DeferredFlags none;
  DeferredFlags &deferred = can_defer ? *m->checkpoint : none;
  deferred.pending = 0;
  int launched = 0;
  double clean_us = 0.0;
  SolverLoadPolicy load_policy(m, bicg ? st.At : nullptr, bicg ? 8 : 5);
  double *rs[2] = {scal.p + SC_RS0, scal.p + SC_RS1};

  // scalars of the previous pass as the composed product launches consume them
  auto fill_pass = [&](SolverPass &sp, int iter, int p_slot0, int a_slot) {
    const int old_slot = p_slot0 + ((iter & 1) ^ 1), new_slot = p_slot0 + (iter & 1);
    sp = SolverPass{};
    sp.b_off = (int64_t)(old_slot - a_slot) * st.S;
    sp.b_new = slot(new_slot);
    sp.den = rs[(iter + 1) & 1];
    sp.num_out = rs[iter & 1];
    sp.alpha_prev = scal.p + SC_ALPHA;
    sp.done = done;
    sp.iters = iters;
    sp.tol2 = tol2;
    sp.iter = iter;
    sp.first = iter == 0;
    sp.sys_scope = st.sys_scope;
    if (st.sharded) {                                         // all-reduced scalars
      sp.part_chk = scal.p + SC_RR;
      sp.n_chk = 0;
      sp.part_num = bicg ? scal.p + SC_RHO : sp.part_chk;
      sp.n_num = 0;
    } else {
      sp.part_chk = part_a.p;
      sp.n_chk = g;
      sp.part_num = bicg ? part_b.p : part_a.p;
      sp.n_num = g;
    }
  };
  // product launches of pass `iter` (final_only != 0: only the test / the owed solution update)
  auto composed_products = [&](int iter, int final_only) -> int {
    SolverPass sp;
    int rc2;
    if (bicg) {
      fill_pass(sp, iter, SLOT_PT0, SLOT_RT);                 // qt = A^T pt ; stores pt_new ; records nothing
      sp.secondary = 1;
      sp.final_only = final_only;
      st.At->halo_shift = (int64_t)SLOT_RT * st.S * 8;
      rc2 = final_only ? CASK_HIP_OK : launch_spmv(*st.At, rt, qt.p, s, nullptr, &sp, false);
      st.At->halo_shift = 0;
      if (rc2) return rc2;
    }
    fill_pass(sp, iter, SLOT_P0, SLOT_R);
    sp.xsol = d_x;
    sp.final_only = final_only;
    if (bicg) sp.wa = slot(SLOT_PT0 + (iter & 1));            // pt_new, stored by the launch above
    return launch_spmv(*m, r, q.p, s, nullptr, &sp, true);
  };

  for (int i = 0; i < maxiters; i++) {
    const double *dot_part = part_c.p;
    int n_dot = g;
    if (st.composed) {
      rc = composed_products(i, 0);
      if (rc) return rc;
      dot_part = m->plan.dot_part.p;
      n_dot = dot_part_count(m->plan);
    } else {
      // classic: q = A p with the shares of p.q (pt.q) from the same launch when the plan has the epilogue
      const bool fused = plan_fuses_dot(m->plan);
      const double *w = bicg ? slot(SLOT_PT0) : slot(SLOT_P0);
      rc = product(m, SLOT_P0, q.p, fused ? w : nullptr, x_full);
      if (rc) return rc;
      if (bicg) {
        // (q = A p and qt = A^T pt are independent; as ONE launch -- r6, k_spmv_merge_dual -- the pass measured 72.67
        // against 72.84 us: nothing, profiles/r06_bicg_dual.txt; on a second stream it lost, docs/experiments.md)
        rc = product(st.At, SLOT_PT0, qt.p, nullptr, x_full_t);
        if (rc) return rc;
      }
      if (fused) {
        dot_part = m->plan.dot_part.p;
        n_dot = dot_part_count(m->plan);
      } else {
        hipLaunchKernelGGL(k_dot_partial, bg, bw, 0, s, n, w, q.p, part_c.p, (const int *)done);
      }
    }
    if (st.sharded) {                                         // p.Ap (pt.q): this rank's share -> scalar -> all ranks
      if (peer_reduce) {                                      // sum + peer-store all-reduce: one launch
        rc = cask_hip_push_sum_allreduce(dot_part, n_dot, nullptr, 0, scal.p + SC_DOT, done, s, st.allreduce_user);
      } else {
        hipLaunchKernelGGL(k_sum_to_scalars, dim3(1), bw, 0, s, dot_part, n_dot, scal.p + SC_DOT, (const double *)nullptr, 0,
                           (double *)nullptr, (const int *)done);
        rc = run_allreduce(st, scal.p + SC_DOT, 1, s);
      }
      if (rc) return rc;
      dot_part = scal.p + SC_DOT;
      n_dot = 0;
    }
    double *rsold = rs[i & 1], *rsnew = rs[(i + 1) & 1];
    // r -= alpha q (and rt -= alpha qt) with the shares of r.r (and rt.r); alpha stays on the device for the launch
    // that applies x += alpha p: the next product (composed passes) or the p update below (classic passes)
    if (bicg)
      CASK_LAUNCH_UPD(shape.big, k_bicg_update_r, bg, bw, 0, s, n, rsold, dot_part, n_dot, q.p, qt.p, r, rt, part_a.p, part_b.p,
                         scal.p + SC_ALPHA, (const int *)done, st.sys_scope);
    else
      CASK_LAUNCH_UPD_J(shape.big, k_cg_update_r, false, bg, bw, 0, s, n, rsold, dot_part, n_dot, q.p, r, part_a.p, scal.p + SC_ALPHA,
                         (const int *)done, st.sys_scope, (const double *)nullptr);     // :208, :212, :218
    const double *chk_part = part_a.p, *rho_part = part_b.p;
    int n_chk = g;
    if (st.sharded) {                                         // r.r (and rt.r): one collective
      if (peer_reduce) {
        rc = cask_hip_push_sum_allreduce(part_a.p, g, bicg ? (const double *)part_b.p : (const double *)nullptr, g,
                                         scal.p + SC_RR, done, s, st.allreduce_user);
      } else {
        hipLaunchKernelGGL(k_sum_to_scalars, dim3(1), bw, 0, s, part_a.p, g, scal.p + SC_RR,
                           bicg ? (const double *)part_b.p : (const double *)nullptr, g, scal.p + SC_RHO, (const int *)done);
        rc = run_allreduce(st, scal.p + SC_RR, bicg ? 2 : 1, s);
      }
      if (rc) return rc;
      chk_part = scal.p + SC_RR;
      rho_part = scal.p + SC_RHO;
      n_chk = 0;
    }
    if (!st.composed) {
      if (bicg)
        CASK_LAUNCH_UPD(shape.big, k_bicg_update_px, bg, bw, 0, s, n, chk_part, rho_part, n_chk, rsold, rsnew, scal.p + SC_ALPHA,
                           tol2, i, r, rt, slot(SLOT_P0), slot(SLOT_PT0), d_x, done, iters, st.sys_scope);
      else
        CASK_LAUNCH_UPD_J(shape.big, k_cg_update_px, false, bg, bw, 0, s, n, chk_part, n_chk, rsold, rsnew, scal.p + SC_ALPHA, tol2, i, r,
                           slot(SLOT_P0), d_x, done, iters, st.sys_scope, (const double *)nullptr);   // :210, :220-231
    }
    if (!st.composed && st.sharded && !st.exchange) {
      // classic passes with in-kernel halos: the p update above must be complete on every rank before any
      // rank's next product reads it -- no collective sits between the two, so one is spent as a fence
      rc = run_allreduce(st, scal.p + SC_COUNT - 1, 1, s);
      if (rc) return rc;
    }
    launched = i + 1;
    if (deferred.pending) {                                   // the checkpoint of the pass before: its flags travelled while this pass was launched
      const int at = deferred.pending;
      HIP_TRY(deferred.wait(h_flags));
      load_policy.decide(at);
      if (h_flags[0]) break;                                  // (the pass just launched: its product(s) overwrote q, its updates return on the flag)
      float ms_so_far = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms_so_far, e0, e1));
      clean_us = ms_so_far * 1e3 / at;
    }
    if ((i + 1) % check_every == 0 || i + 1 == maxiters) {
      if (st.composed) {                                      // the test of pass i belongs to the next product launch:
        rc = composed_products(i + 1, i + 1 == maxiters ? 1 : 2);   // run it now, without the product
        if (rc) return rc;
      }
      HIP_TRY(hipEventRecord(e1, s));
      load_policy.record(launched, s);
      // classic passes on one GPU: deferred (DeferredFlags); a composed pass tests in its NEXT product launch and a sharded
      // one is paced by its collectives: those wait here
      if (can_defer && i + 1 < maxiters) {
        HIP_TRY(deferred.request(flags.p, launched, s));
        continue;
      }
      HIP_TRY(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      load_policy.decide(launched);
      if (h_flags[0]) break;
      // a checkpoint reached without convergence: every pass so far did real work
      float ms_so_far = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms_so_far, e0, e1));
      clean_us = ms_so_far * 1e3 / launched;
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e1, s));
  if (can_defer) {                                            // (the pinned buffer: no staging copy; the event behind it is the last thing queued)
    HIP_TRY(deferred.request(flags.p, launched, s));
    HIP_TRY(deferred.wait(h_flags));
  } else {
    HIP_TRY(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (iterations) *iterations = h_flags[1];
  if (converged) *converged = h_flags[0] != 0;          // the flag carries the pass that set it
  // passes launched after the converged one are no-ops: prefer the rate measured up to the last
  // checkpoint that had not converged yet
  if (usec_per_iteration) *usec_per_iteration = clean_us > 0 ? clean_us : (launched ? ms * 1e3 / launched : 0.0);
  return CASK_HIP_OK;
}

static int solve_host(int kind, cask_hip_matrix *m, const double *rhs, double *x, int32_t maxiters, double tol,
                      int32_t *iterations, int32_t *converged, double *usec_per_iteration) {
  int rc = solver_common_checks(m, rhs, x, maxiters, tol);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(m->device));
  const int64_t n = m->n_rows;
  DevBuf<double> dx, db;
  HIP_TRY(dx.upload(x, n)); HIP_TRY(db.upload(rhs, n));
  cask_hip_solver_config cfg{};
  cfg.kind = kind;
  if (const char *e = std::getenv("CASK_HIP_SOLVER_MODE")) cfg.mode = std::atoi(e);   // development A/B: 1 composed, 2 classic
  rc = cask_hip_solve_device(m, nullptr, &cfg, db.p, dx.p, maxiters, tol, iterations, converged, usec_per_iteration, m->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(x, dx.p, n * sizeof(double), hipMemcpyDeviceToHost));
  return CASK_HIP_OK;
}

int cask_hip_cg(cask_hip_matrix *m, const double *rhs, double *x, int32_t maxiters, double tol,
                int32_t *iterations, int32_t *converged, double *usec_per_iteration) {
  return solve_host(CASK_HIP_SOLVER_CG, m, rhs, x, maxiters, tol, iterations, converged, usec_per_iteration);
}

int cask_hip_bicg(cask_hip_matrix *m, const double *rhs, double *x, int32_t maxiters, double tol,
                  int32_t *iterations, int32_t *converged, double *usec_per_iteration) {
  return solve_host(CASK_HIP_SOLVER_BICG, m, rhs, x, maxiters, tol, iterations, converged, usec_per_iteration);
}

// Preconditioned CG, pcg<double, Precon> of the reference (SparseLinearSolvers.hpp:162-239) with the
// preconditioner applied on the device: r = b - A x ; z = M^-1 r ; p = z ; rsold = r.z ; then per pass
// Ap = A p ; alpha = rsold / p.Ap ; x += alpha p ; r -= alpha Ap ; z = M^-1 r ; rsnew = r.z ;
// stop if rsnew <= tol^2 ; p = z + (rsnew/rsold) p.  `iterations` as the reference counts them.
int cask_hip_pcg(cask_hip_matrix *m, cask_hip_precond *precond, const double *rhs, double *x, int32_t maxiters,
                 double tol, int32_t *iterations, int32_t *converged, double *usec_per_iteration) {
  int rc = solver_common_checks(m, rhs, x, maxiters, tol);
  if (rc) return rc;
  if (!precond) return cask_hip_cg(m, rhs, x, maxiters, tol, iterations, converged, usec_per_iteration);
  if (cask_hip_precond_rows(precond) != m->n_rows)
    return fail(CASK_HIP_ERR_INVALID, "the preconditioner was built for a matrix of a different order");
  HIP_TRY(hipSetDevice(m->device));
  const int64_t n = m->n_rows;
  hipStream_t s = m->stream;
  DevBuf<double> dx, db, r, z, p, Ap, partials, partials_rz, scal;
  DevBuf<int> flags;
  HIP_TRY(dx.upload(x, n)); HIP_TRY(db.upload(rhs, n));
  HIP_TRY(r.alloc(n)); HIP_TRY(z.alloc(n)); HIP_TRY(p.alloc(n)); HIP_TRY(Ap.alloc(n));
  HIP_TRY(partials.alloc(BLAS_MAX_PARTIALS)); HIP_TRY(partials_rz.alloc(BLAS_MAX_PARTIALS));
  HIP_TRY(scal.alloc(4)); HIP_TRY(flags.alloc(2));
  HIP_TRY(hipMemsetAsync(flags.p, 0, 2 * sizeof(int), s));
  double *rs[2] = {scal.p, scal.p + 1};
  int *done = flags.p, *iters = flags.p + 1;
  const SolverShape shape = solver_shape(n, false);
  const int g = shape.grid;
  const dim3 bg(g), bw(shape.wg);
  rc = launch_spmv(*m, dx.p, r.p, s);                                                   // :189-190
  if (rc) return rc;
  hipLaunchKernelGGL(k_axpby, bg, bw, 0, s, n, 1.0, db.p, 1.0, -1.0, (const double *)nullptr, (const double *)nullptr,
                     r.p, (const int *)nullptr);
  rc = cask_hip_precond_apply_device(precond, r.p, z.p, s);                             // :193
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(p.p, z.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));    // :195
  hipLaunchKernelGGL(k_dot_partial, bg, bw, 0, s, n, r.p, z.p, partials.p, (const int *)nullptr);
  hipLaunchKernelGGL(k_dot_final, dim3(1), bw, 0, s, g, partials.p, rs[0], 0, 0.0, (int *)nullptr, (int *)nullptr, 0);
  HIP_TRY(hipGetLastError());

  DevEvent e0, e1;
  HIP_TRY(e0.create()); HIP_TRY(e1.create());
  HIP_TRY(hipEventRecord(e0, s));
  const int check_every = 16;
  int h_flags[2] = {0, 0};
  int launched = 0;
  double clean_us = 0.0;
  const bool fused = plan_fuses_dot(m->plan);
  const double *jacobi = cask_hip_precond_jacobi_scale(precond);   // 1/diag on the device, or NULL for the ILU kinds
  // (a Jacobi pass is launches that return on the done flag; an ILU application is not)
  const bool can_defer = jacobi != nullptr && ensure_checkpoint(m);
  DeferredFlags none;
  DeferredFlags &deferred = can_defer ? *m->checkpoint : none;
  deferred.pending = 0;
  SolverLoadPolicy load_policy(m, nullptr, 6);
  for (int i = 0; i < maxiters; i++) {
    double *rsold = rs[i & 1], *rsnew = rs[(i + 1) & 1];
    const double *pAp_part = partials.p;
    int n_pAp = g;
    if (fused) {
      rc = launch_spmv(*m, p.p, Ap.p, s, p.p);                                          // :206-208
      pAp_part = m->plan.dot_part.p;
      n_pAp = dot_part_count(m->plan);
    } else {
      rc = launch_spmv(*m, p.p, Ap.p, s);
      hipLaunchKernelGGL(k_dot_partial, bg, bw, 0, s, n, p.p, Ap.p, partials.p, (const int *)done);
    }
    if (rc) return rc;
    if (jacobi) {
      // Jacobi: z = dinv * r is never stored.  r -= alpha Ap with the shares of r.z (:212-218), then rsnew, the test,
      // p = z + beta p with z recomputed and the x update this pass owes (:210, :220-231): 3 launches and 10 vector
      // passes per iteration instead of 4 and 12
      CASK_LAUNCH_UPD_J(shape.big, k_cg_update_r, true, bg, bw, 0, s, n, rsold, pAp_part, n_pAp, Ap.p, r.p, partials_rz.p, scal.p + 2,
                         (const int *)done, 0, jacobi);
      CASK_LAUNCH_UPD_J(shape.big, k_cg_update_px, true, bg, bw, 0, s, n, partials_rz.p, g, rsold, rsnew, scal.p + 2, tol * tol, i, r.p,
                         p.p, dx.p, done, iters, 0, jacobi);
    } else {
      // x += alpha p ; r -= alpha Ap  (:210-212; the r.r shares this kernel also leaves are not used here)
      hipLaunchKernelGGL(k_cg_update_xr, bg, bw, 0, s, n, rsold, pAp_part, n_pAp, p.p, Ap.p, dx.p, r.p, partials_rz.p,
                         (const int *)done);
      // z = M^-1 r (:215) and the shares of r.z (:218)
      rc = cask_hip_precond_apply_device(precond, r.p, z.p, s);
      if (rc) return rc;
      hipLaunchKernelGGL(k_dot_partial, bg, bw, 0, s, n, r.p, z.p, partials_rz.p, (const int *)done);
      // rsnew = r.z ; converged? ; p = z + (rsnew/rsold) p   (:220-231)
      hipLaunchKernelGGL(k_cg_update_p, bg, bw, 0, s, n, partials_rz.p, g, rsold, rsnew, tol * tol, i, z.p, p.p, done,
                         iters);
    }
    launched = i + 1;
    if (deferred.pending) {                                   // the checkpoint of the pass before: its flags travelled while this pass was launched
      const int at = deferred.pending;
      HIP_TRY(deferred.wait(h_flags));
      load_policy.decide(at);
      if (h_flags[0]) break;                                  // (the pass just launched: its product overwrote Ap, its updates return on the flag)
      float ms_so_far = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms_so_far, e0, e1));
      clean_us = ms_so_far * 1e3 / at;
    }
    if ((i + 1) % check_every == 0 || i + 1 == maxiters) {
      HIP_TRY(hipEventRecord(e1, s));
      load_policy.record(launched, s);
      if (can_defer && i + 1 < maxiters) {
        HIP_TRY(deferred.request(flags.p, launched, s));
        continue;
      }
      HIP_TRY(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      load_policy.decide(launched);
      if (h_flags[0]) break;
      float ms_so_far = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms_so_far, e0, e1));
      clean_us = ms_so_far * 1e3 / launched;
    }
  }
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(x, dx.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (iterations) *iterations = h_flags[1];
  if (converged) *converged = h_flags[0] != 0;          // the flag carries the pass that set it
  if (usec_per_iteration) *usec_per_iteration = clean_us > 0 ? clean_us : (launched ? ms * 1e3 / launched : 0.0);
  return CASK_HIP_OK;
}

}  // extern "C"
