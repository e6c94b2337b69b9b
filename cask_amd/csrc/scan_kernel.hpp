// The segmented-scan kernel (variant SCAN): thread-owned runs of products + a segmented scan of the carries.
//
// Built for matrices whose rows are short and uneven (webbase-1M: mean 3 nonzeros, longest 4 700), where the
// merge kernel's row phase -- lanes mapped to ROWS -- leaves most lanes idle behind the few that walk a long
// row, needs the row offsets of ~700 rows per block in LDS, and keeps a workgroup alive for 10 us
// (profiles/r02_webbase_anatomy.txt).  Here work is mapped to NONZEROS throughout:
//
//   * a block is a run of at most CAP = wg_size * items_per_thread nonzeros snapped to row boundaries (row ends
//     are not items, so a block of 1-nonzero rows still carries CAP nonzeros);
//   * phase 1 is the merge kernel's: the value / column streams in nonzero order with 16-byte loads, x gathered
//     from L2 (or the window), products parked in LDS, one pad double per run (the transpose from the striped load
//     order to thread-owned runs);
//   * phase 2: thread t owns products [t*IPT, (t+1)*IPT).  One 32-bit word per thread, made by the planner, says
//     which of them end a row (bits 0-15) and the ordinal of the thread's first row end among the block's
//     non-empty rows (bits 16-31).  The thread adds its run in order and stores every row that ends AND starts
//     inside the run directly; the sum in front of its first row end ("head") needs what earlier threads left
//     ("carry"), the sum behind its last one ("tail") is what it leaves;
//   * carries: a segmented inclusive scan of (has a row end, tail) over the workgroup -- six DPP steps per wave
//     (row_shr 1/2/4/8, row_bcast15, row_bcast31: no LDS), the wave aggregates through LDS; row sums are staged in
//     LDS by ordinal and leave for y in one coalesced sweep.  A row of 500 nonzeros is 62 threads' tails joined by the scan in
//     log steps: no second pass, no lanes-per-row choice, no skew flag.
//
// row_ptr is not read at all (4 bytes per row less than the CSR stream; the per-thread words are 0.5 byte per
// nonzero at 8 items per thread), and without a window the block needs (CAP + wg_size + 4) * 8 bytes of LDS -- 18 KB
// at 256 x 8 -- so eight workgroups fit a CU and the whole grid of a 3 M-nonzero matrix is resident at once.
//
// Every floating-point addition's operands are a function of the plan alone: results are bitwise reproducible.
// The order differs from the merge kernel's (and from the sequential oracle's), like every parallel row sum here.
//
// Empty rows: a block that spans empty rows (KIND_HOLES) maps row-end ordinals to rows through `rowmap` and
// zero-fills its empty rows from row_ptr; runs of empty rows between blocks are zero-fill pieces (as in the merge
// plan).  Rows longer than CAP/2 are long-row pieces summed by the whole workgroup (+ the fix-up kernel when a row
// has several).
//
// x window (XP > 0): the plan then keeps its own column stream.  A nonzero inside its block's x window (per block the
// densest column range of <= 2 * XP * wg_size entries, loaded in 16-byte pairs in front of the streams and parked in LDS)
// carries an LDS slot.  (Rounds 3-5 also had "far columns" here -- the scattered nonzeros served from a column-panel
// pre-gather, as its own launch or by producer workgroups of the product launch; both cut the fabric traffic of the
// webbase-like matrix from 2.3x to 1.28x the algorithmic bytes and both cost more time than they saved: removed in
// ABI 7, docs/experiments.md.)
//
// Sub-matrix blocks (r6, the long rows of a SLICE plan: slice_kernel.hpp): KIND_HOLES | KIND_NOFILL -- the row map names
// the rows the block owns, every other row of its span belongs to somebody else and is left alone.
//
// Reference: the always-streaming multiply / reduce pipeline of src/spmv/src/SpmvKernel.java:18-309 and the
// row-length driven read control of ParallelCsrReadControl.java:148-208 -- whose per-cycle "this entry ends a row"
// control word is the same idea as the per-thread row-end word here.
#pragma once
#include "scan_launch.hpp"

namespace caskhip {

// One piece of one long row (or a run of empty rows): the whole workgroup strides over it.
template <bool NT>
__device__ __forceinline__ void scan_long_piece(const BlockDesc &d, const int *__restrict__ ci,
                                                const double *__restrict__ val, const double *__restrict__ x,
                                                double *__restrict__ y, double *__restrict__ partials, double *red) {
  const int WG = blockDim.x, tid = threadIdx.x;
  if (d.nnz_count == 0) {                                     // a run of empty rows
    for (int r = tid; r < d.n_rows; r += WG) y[d.row_start + r] = 0.0;
    return;
  }
  const int end = d.nnz_start + d.nnz_count;
  double acc = 0.0;
  for (int k = d.nnz_start + tid; k < end; k += 4 * WG) {
    int c[4];
    double v[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int kk = min(k + u * WG, end - 1);
      c[u] = stream_load<NT>(ci + kk);
      v[u] = stream_load<NT>(val + kk);
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k + u * WG < end) acc = fma(v[u], xv[u], acc);
  }
  acc = group_sum<64>(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < (WG >> 6); w++) s += red[w];
    if (d.kind_g & KIND_PARTIAL) partials[d.aux] = s;
    else y[d.row_start] = s;
  }
}

// Phase 2 of a SCAN block (the products are parked in LDS, one pad double in front of every thread-owned run of IPT): thread-owned
// runs guided by the row-end word, carries joined by a segmented scan in DPP, row sums staged by ordinal and swept out.
template <int IPT>
__device__ __forceinline__ void scan_row_sums(const BlockDesc &d, int lead, unsigned mw, double *prod, double *wval, int *wflag,
                                              const int *__restrict__ rp, const int *__restrict__ rowmap, double *__restrict__ y) {
  const int WG = blockDim.x, tid = threadIdx.x;
  // ---- phase 2: thread-owned runs ---------------------------------------------------------------------------
  const unsigned flags = mw & 0xffffu;
  const int ord = (int)(mw >> 16);
  const int k0 = tid * IPT;
  const bool holes = (d.kind_g & KIND_HOLES) != 0;            // workgroup-uniform
  const int *rmap = rowmap + d.aux;                           // holes: [number of non-empty rows, their local rows ...]
  double p[IPT];
#pragma unroll
  for (int j = 0; j < IPT; j++) p[j] = prod[lead + 1 + (IPT + 1) * tid + j];
#pragma unroll
  for (int j = 0; j < IPT; j++)
    if (k0 + j >= d.nnz_count) p[j] = 0.0;                    // behind the block's last nonzero
  __syncthreads();                                            // every run is in registers: the product area is free
  // row sums go to rsum[ordinal] (the product area again) and leave for y in one coalesced sweep at the end:
  // stored from here, lane by lane, a block's ~700 row sums were ~90 partly filled store instructions per wave
  // (3.8 of the launch's 27 us on the webbase-like matrix)
  double *rsum = prod;
  double acc = 0.0, head = 0.0;
  int i = 0;
#pragma unroll
  for (int j = 0; j < IPT; j++) {
    acc += p[j];
    if ((flags >> j) & 1u) {
      if (i == 0) head = acc;                                 // completes a row earlier threads may have begun
      else rsum[ord + i] = acc;
      i++;
      acc = 0.0;
    }
  }
  // segmented inclusive scan over the wave, (f, s) = (a row ends in lanes [.., lane], sum of the tails since), all
  // in DPP: row_shr 1, 2, 4, 8 inside the rows of 16 lanes (lanes without a source read the identity: bound_ctrl),
  // then lane 15 of each row to the next row (row_bcast15, rows 1 and 3) and lane 31 to the upper half (row_bcast31)
  const int lane = tid & 63, wave = tid >> 6;
  int f = flags != 0;
  double s = acc;
#define CASK_SCAN_STEP(CTRL, ROWMASK)                                                                     \
  do {                                                                                                    \
    int lo = __double2loint(s), hi = __double2hiint(s);                                                   \
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, true);                                    \
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, true);                                    \
    const int f2 = __builtin_amdgcn_update_dpp(0, f, CTRL, ROWMASK, 0xf, true);                           \
    const double s2 = __hiloint2double(hi, lo);                                                           \
    if (!f) s += s2;                                                                                      \
    f |= f2;                                                                                              \
  } while (0)
  CASK_SCAN_STEP(0x111, 0xf);                                 // row_shr:1
  CASK_SCAN_STEP(0x112, 0xf);                                 // row_shr:2
  CASK_SCAN_STEP(0x114, 0xf);                                 // row_shr:4
  CASK_SCAN_STEP(0x118, 0xf);                                 // row_shr:8
  CASK_SCAN_STEP(0x142, 0xa);                                 // row_bcast15 -> rows 1, 3
  CASK_SCAN_STEP(0x143, 0xc);                                 // row_bcast31 -> rows 2, 3
#undef CASK_SCAN_STEP
  if (lane == 63) {
    wval[wave] = s;
    wflag[wave] = f;
  }
  // what the lanes in front of this one leave (exclusive): lane - 1's inclusive pair (wave_shr:1; lane 0: identity)
  const double es = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(s), 0x138, 0xf, 0xf, true),
                                     __builtin_amdgcn_update_dpp(0, __double2loint(s), 0x138, 0xf, 0xf, true));
  const int ef = __builtin_amdgcn_update_dpp(0, f, 0x138, 0xf, 0xf, true);
  __syncthreads();
  if (flags) {                                                // this thread closes a row begun before its run
    double carry = 0.0;                                       // what the waves in front leave
    for (int w = 0; w < wave; w++) carry = wflag[w] ? wval[w] : carry + wval[w];
    rsum[ord] = (ef ? es : carry + es) + head;
  }
  __syncthreads();
  if (holes) {
    const int n_ends = rmap[0];
    for (int r = tid; r < n_ends; r += WG) y[d.row_start + rmap[1 + r]] = rsum[r];
    if (!(d.kind_g & KIND_NOFILL))                            // (sub-matrix blocks: the other rows are somebody else's)
      for (int r = tid; r < d.n_rows; r += WG)                // the block's empty rows (every y entry is written once)
        if (rp[d.row_start + r] == rp[d.row_start + r + 1]) y[d.row_start + r] = 0.0;
  } else {
    for (int r = tid; r < d.n_rows; r += WG) y[d.row_start + r] = rsum[r];
  }
  CASK_STAMP(4);
}

// Column references of the plan's own column stream (plans with an x window; otherwise the caller's col_ind is
// streamed as it is):  without SCAN_LDS_BIT: x[c];  with it: slot c & 0xffff of the block's x window in LDS.
// r5: the x window SHARES the product area's LDS (it is dead once every thread holds its x values: one more barrier) --
// a 2 048-entry window costs no LDS at 256 x 8 and the grid stays at 8 workgroups per CU; with 16 KB of its own the same
// window took the kernel to 4 per CU, which is what made windows lose on short rows (profiles/r05_merge_forms.txt).
template <int IPT, bool NT, int XP>
__device__ __forceinline__ void scan_block(int hw_block, const BlockDesc *__restrict__ blocks, int n_blocks, int remap,
                                           int nnz, int n_cols, const int *__restrict__ rp, const int *__restrict__ ci,
                                           const double *__restrict__ val, const unsigned *__restrict__ meta,
                                           const int *__restrict__ rowmap, const double *__restrict__ x,
                                           double *__restrict__ y, double *__restrict__ partials) {
  static_assert(IPT % 2 == 0 && IPT <= 16, "items per thread: even (16-byte loads), at most 16 (row-end bits)");
  extern __shared__ __align__(16) unsigned char smem[];
  const int WG = blockDim.x, CAP = WG * IPT, tid = threadIdx.x;
  double *prod = reinterpret_cast<double *>(smem);            // CAP + WG + 4 doubles (padded runs, see below)
  double *wval = prod + CAP + WG + 4;                              // 16 wave aggregates: value ...
  int *wflag = reinterpret_cast<int *>(wval + 16);            // ... and "holds a row end"
  static_assert(XP == 0 || 2 * XP <= IPT + 1, "the window must fit the product area");
  double *xs = prod;                                          // XP > 0: the block's x window, 2 * XP * WG doubles, in the product area

  CASK_STAMP(0);
  const int lb = logical_block(hw_block, n_blocks, remap);
  const BlockDesc d = blocks[lb];
  if (d.kind_g & KIND_LONG) {
    scan_long_piece<NT>(d, ci, val, x, y, partials, prod);
    return;
  }

  // ---- phase 1: streams -> products in LDS (element base + i of the arrays lands in prod[i]) ----------------
  // 16-byte loads need an even element index: start one element early if the block starts on an odd nonzero
  const int base = d.nnz_start & ~1, lead = d.nnz_start - base, total = d.nnz_count + lead;
  const int npairs = (total + 1) >> 1, first = base >> 1;
  const int last = min(first + max(npairs - 1, 0), ((nnz + 1) >> 1) - 1);
  // the x window [cmin, cmin + cwidth) (cmin even) goes out first: vmcnt retires in order, so parking it in LDS
  // waits for these loads only while the streams behind them are still in flight (merge_kernel.hpp, issue order)
  dbl2 xw[XP > 0 ? XP : 1];
  if (XP > 0) {
    const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x);
    const int p0 = d.cmin >> 1, plim = (n_cols - 1) >> 1;     // the last pair may reach one element past an odd n_cols:
#pragma unroll                                                //   x is 16-byte aligned, the load cannot cross a page
    for (int u = 0; u < XP; u++)
      if ((u * WG + tid) * 2 < d.cwidth) xw[u] = x2[min(p0 + u * WG + tid, plim)];
  }
  const unsigned mw = stream_load<NT>(meta + (size_t)lb * WG + tid);
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *ci2 = reinterpret_cast<const int2v *>(ci);
  dbl2 v[IPT / 2];
  int2v c[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int p = min(first + u * WG + tid, last);           // clamped: redundant loads hit the same line
    v[u] = stream_load<NT>(val2 + p);
    c[u] = stream_load<NT>(ci2 + p);
  }
  CASK_STAMP(1);
  if (XP > 0) {
    dbl2 *xs2 = reinterpret_cast<dbl2 *>(xs);
#pragma unroll
    for (int u = 0; u < XP; u++)
      if ((u * WG + tid) * 2 < d.cwidth) xs2[u * WG + tid] = xw[u];
    __syncthreads();
  }
  // foreign elements (the lead element of an odd start; the second half of the last pair of an odd total, which
  // for the very last nonzero of an odd-nnz matrix is the 8 bytes behind the arrays): give them a column this
  // block owns; their products land in slots no thread's run covers
  if (lead && tid == 0) c[0].x = c[0].y;
  if (total & 1) {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++)
      if (u * WG + tid >= npairs - 1) c[u].y = c[u].x;
  }
  dbl2 xv[IPT / 2];
  if (XP > 0) {
    // Global sources first (all of a lane's loads go out back to back), then the window slots from LDS -- every load
    // UNCONDITIONAL, the choice made by selects afterwards.  (r5 loaded under `if (!(c & SCAN_LDS_BIT))` / `if (c &
    // SCAN_LDS_BIT)`: the compiler joins a conditionally loaded register with its default at the end of the branch and
    // waits there -- s_waitcnt vmcnt(0) between the gathers: three dependent trips where one was meant.)  A lane whose
    // reference is a window slot gathers the window's first entry: the lanes share that address, one request per wave.
    int gx[IPT / 2], gy[IPT / 2];
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      gx[u] = (c[u].x & SCAN_LDS_BIT) ? d.cmin : c[u].x;
      gy[u] = (c[u].y & SCAN_LDS_BIT) ? d.cmin : c[u].y;
    }
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = x[gx[u]];
      xv[u].y = x[gy[u]];
    }
    dbl2 xl[IPT / 2];
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xl[u].x = xs[(c[u].x & SCAN_LDS_BIT) ? (c[u].x & 0xffff) : 0];
      xl[u].y = xs[(c[u].y & SCAN_LDS_BIT) ? (c[u].y & 0xffff) : 0];
    }
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = (c[u].x & SCAN_LDS_BIT) ? xl[u].x : xv[u].x;
      xv[u].y = (c[u].y & SCAN_LDS_BIT) ? xl[u].y : xv[u].y;
    }
    __syncthreads();                                          // every thread has its x values: the window's LDS is the products' now
  } else {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = x[c[u].x];
      xv[u].y = x[c[u].y];
    }
  }
#ifdef CASK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CASK_STAMP(2);
#endif
  // Products -> LDS, one pad double in front of every thread-owned run of IPT: element e (relative to `base`) sits in
  // slot e + (e + IPT - lead) / IPT, so that thread t's run starts at lead + 1 + (IPT + 1) t -- the runs of a wave's
  // lanes then start on different banks (stride 9 doubles at IPT = 8: no two lanes of a half-wave share a bank pair)
  // and the phase-2 reads cost 2 cycles each instead of 16.
  constexpr int SH = IPT == 2 ? 1 : IPT == 4 ? 2 : IPT == 8 ? 3 : 4;
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int e = 2 * (u * WG + tid);
    const dbl2 pr = v[u] * xv[u];
    prod[e + ((e + IPT - lead) >> SH)] = pr.x;
    prod[e + 1 + ((e + 1 + IPT - lead) >> SH)] = pr.y;
  }
  __syncthreads();
  CASK_STAMP(3);

  scan_row_sums<IPT>(d, lead, mw, prod, wval, wflag, rp, rowmap, y);
}

// ---- the descriptor out of the stream's way (r6) -----------------------------------------------------------------------
// A block's first act used to be a cold 32-byte descriptor read, and its streams could not be requested before that read
// had returned -- behind the streams every OTHER block had already queued: 3.2 us for the median workgroup of the
// webbase2 launch, 7 us for the slowest tenth, of a 10.4 us life (profiles/r06_scan_pad.txt).  Here the plan owns PADDED
// copies of the value and column streams: block b's nonzeros live at b * CAP .. (the rest of its CAP slots: value +0.0, a
// column of its own -- blocks are cut at <= CAP - 1 nonzeros and snapped to rows of ~3, so 0.1 % is padding), every
// stream address is a function of the block index, and the workgroup requests its streams and its row-end words AT
// ENTRY, the descriptor alongside.  The descriptor is first needed for the x window (requested when it arrives, parked
// when the streams ahead of it have landed: loads return in order) and for the rows at the very end.  No lead element, no
// clamped pairs, no foreign halves: padded streams start on an even element and end inside the block's own slots.
// Long-row pieces keep their descriptor-first path and sit BEHIND the regular blocks in the block list (hardware blocks
// >= n_regular), so that "is this a long piece" is a comparison of the block index, not a field of the descriptor.
// (The same idea LOST on the merge kernel in round 2 -- 8.21 -> 8.61 us on cant-like: there the descriptor trip is ~1 us
// and the window / row-offset loads it gates are on the critical path.)
template <int IPT, bool NT, int XP>
__device__ __forceinline__ void scan_block_padded(int hw_block, const BlockDesc *__restrict__ blocks, int n_regular, int remap,
                                                  int n_cols, const int *__restrict__ rp, const int *__restrict__ ci,
                                                  const double *__restrict__ val, const int *__restrict__ pci,
                                                  const double *__restrict__ pval, const unsigned *__restrict__ meta,
                                                  const int *__restrict__ rowmap, const double *__restrict__ x,
                                                  double *__restrict__ y, double *__restrict__ partials) {
  static_assert(IPT % 2 == 0 && IPT <= 16, "items per thread: even (16-byte loads), at most 16 (row-end bits)");
  extern __shared__ __align__(16) unsigned char smem[];
  const int WG = blockDim.x, CAP = WG * IPT, tid = threadIdx.x;
  double *prod = reinterpret_cast<double *>(smem);
  double *wval = prod + CAP + WG + 4;
  int *wflag = reinterpret_cast<int *>(wval + 16);
  static_assert(XP == 0 || 2 * XP <= IPT + 1, "the window must fit the product area");
  double *xs = prod;
  CASK_STAMP(0);
  if (hw_block >= n_regular) {                                // a long-row piece (or a run of empty rows): descriptor first
    const BlockDesc d = blocks[hw_block];
    scan_long_piece<NT>(d, ci, val, x, y, partials, prod);
    return;
  }
  const int lb = logical_block(hw_block, n_regular, remap);
  // The descriptor as the FIRST VECTOR load of the wave (every lane the same 32 bytes: one request).  A scalar load would
  // be the natural thing, but the compiler sinks it to its first use and waits there (`s_waitcnt lgkmcnt(0)` in the
  // middle of the stream requests: scalar loads return out of order, so it cannot wait for less) -- a vector load issued
  // first is simply the first to come back (vmcnt retires in order), and using it waits for nothing behind it.
  typedef int int4v __attribute__((ext_vector_type(4)));
  int zero = 0;
  asm volatile("" : "+v"(zero));                              // (opaque: keeps the address per-lane, i.e. the load a vector load)
  const int4v *dp = reinterpret_cast<const int4v *>(blocks + lb) + zero;
  const int4v dlo = dp[0], dhi = dp[1];
  // streams and row-end words: addresses from the block index alone
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(pval) + (size_t)lb * (CAP / 2);
  const int2v *ci2 = reinterpret_cast<const int2v *>(pci) + (size_t)lb * (CAP / 2);
  dbl2 v[IPT / 2];
  int2v c[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    v[u] = stream_load<NT>(val2 + u * WG + tid);
    c[u] = stream_load<NT>(ci2 + u * WG + tid);
  }
  const unsigned mw = stream_load<NT>(meta + (size_t)lb * WG + tid);
  CASK_STAMP(1);
  BlockDesc d;
  d.row_start = __builtin_amdgcn_readfirstlane(dlo.x);
  d.n_rows = __builtin_amdgcn_readfirstlane(dlo.y);
  d.nnz_start = __builtin_amdgcn_readfirstlane(dlo.z);
  d.nnz_count = __builtin_amdgcn_readfirstlane(dlo.w);
  d.cmin = __builtin_amdgcn_readfirstlane(dhi.x);
  d.cwidth = __builtin_amdgcn_readfirstlane(dhi.y);
  d.kind_g = __builtin_amdgcn_readfirstlane(dhi.z);
  d.aux = __builtin_amdgcn_readfirstlane(dhi.w);
  if (XP > 0) {
    const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x);
    const int p0 = d.cmin >> 1, plim = (n_cols - 1) >> 1;
    dbl2 xw[XP > 0 ? XP : 1];
#pragma unroll
    for (int u = 0; u < XP; u++) xw[u] = x2[min(p0 + min(u * WG + tid, max(d.cwidth / 2 - 1, 0)), plim)];   // unconditional, clamped
    dbl2 *xs2 = reinterpret_cast<dbl2 *>(xs);
#pragma unroll
    for (int u = 0; u < XP; u++)
      if ((u * WG + tid) * 2 < d.cwidth) xs2[u * WG + tid] = xw[u];
    __syncthreads();
  }
  dbl2 xv[IPT / 2];
  if (XP > 0) {
    int gx[IPT / 2], gy[IPT / 2];
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      gx[u] = (c[u].x & SCAN_LDS_BIT) ? d.cmin : c[u].x;
      gy[u] = (c[u].y & SCAN_LDS_BIT) ? d.cmin : c[u].y;
    }
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = x[gx[u]];
      xv[u].y = x[gy[u]];
    }
    dbl2 xl[IPT / 2];
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xl[u].x = xs[(c[u].x & SCAN_LDS_BIT) ? (c[u].x & 0xffff) : 0];
      xl[u].y = xs[(c[u].y & SCAN_LDS_BIT) ? (c[u].y & 0xffff) : 0];
    }
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = (c[u].x & SCAN_LDS_BIT) ? xl[u].x : xv[u].x;
      xv[u].y = (c[u].y & SCAN_LDS_BIT) ? xl[u].y : xv[u].y;
    }
    __syncthreads();                                          // every thread has its x values: the window's LDS is the products' now
  } else {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = x[c[u].x];
      xv[u].y = x[c[u].y];
    }
  }
#ifdef CASK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CASK_STAMP(2);
#endif
  constexpr int SH = IPT == 2 ? 1 : IPT == 4 ? 2 : IPT == 8 ? 3 : 4;
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int e = 2 * (u * WG + tid);
    const dbl2 pr = v[u] * xv[u];
    prod[e + ((e + IPT) >> SH)] = pr.x;
    prod[e + 1 + ((e + 1 + IPT) >> SH)] = pr.y;
  }
  __syncthreads();
  CASK_STAMP(3);
  scan_row_sums<IPT>(d, 0, mw, prod, wval, wflag, rp, rowmap, y);
}

template <int IPT, bool NT, int XP>
__global__ void k_spmv_scan_pad(const BlockDesc *__restrict__ blocks, int n_regular, int remap, int n_cols,
                                const int *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                                const int *__restrict__ pci, const double *__restrict__ pval,
                                const unsigned *__restrict__ meta, const int *__restrict__ rowmap,
                                const double *__restrict__ x, double *__restrict__ y, double *__restrict__ partials) {
  scan_block_padded<IPT, NT, XP>(blockIdx.x, blocks, n_regular, remap, n_cols, rp, ci, val, pci, pval, meta, rowmap, x, y, partials);
}

template <int IPT, bool NT, int XP>
__global__ void k_spmv_scan(const BlockDesc *__restrict__ blocks, int n_blocks, int remap, int nnz, int n_cols,
                            const int *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                            const unsigned *__restrict__ meta, const int *__restrict__ rowmap,
                            const double *__restrict__ x, double *__restrict__ y, double *__restrict__ partials) {
  scan_block<IPT, NT, XP>(blockIdx.x, blocks, n_blocks, remap, nnz, n_cols, rp, ci, val, meta, rowmap, x, y, partials);
}

}  // namespace caskhip
