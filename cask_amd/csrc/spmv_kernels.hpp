// SpMV device kernels for MI355X (gfx950, CDNA4): wave-64, DPP reductions,
// LDS-staged x tiles, XCD-aware workgroup mapping.  fp64, bandwidth bound: the
// design goal is coalesced 16-byte streams of values/col_ind with many bytes in
// flight per CU, and a cheap x gather (LDS window when the row block is banded,
// L2 otherwise).  No MFMA: arithmetic intensity is ~0.16 flop/byte.
//
// What these kernels replace in the reference: the MaxJ dataflow design
// src/spmv/src/SpmvKernel.java:18-309 (multiply lanes + adder tree + per-row
// accumulate + cross-block reduction), ParallelCsrReadControl.java:6-316 (row
// -> lane scheduling) and SpmvCacheKernel (SpmvKernel.java:107-196, the x tile
// cache, here a single LDS copy instead of input_width BRAM replicas).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace caskhip {

// Diagnostic build only (-DCASK_STAMPS, tools/stamps.py): wave 0 of every merge workgroup records
// s_memrealtime (100 MHz) at phase boundaries into a side buffer no other code reads.
#ifdef CASK_STAMPS
__device__ unsigned long long *g_stamps = nullptr;
#define CASK_STAMP(i)                                                                     \
  do {                                                                                    \
    if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CASK_STAMP(i) do {} while (0)
#endif

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int    int2v __attribute__((ext_vector_type(2)));

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

struct SplitRow {      // a row whose pieces are summed by the fix-up kernel
  int32_t row, first_slot, n_slots, pad;
};

// Where a launch reads x from.  Without a halo every column comes from x[] (n_own = INT_MAX,
// haddr = NULL).  With one (row-sharded product, include/cask_hip_p2p.h) columns >= n_own are
// halo columns: column n_own + j is read from the absolute device address haddr[j], which may
// lie in a peer GPU's shared slice -- the remote load over xGMI happens inside the product
// kernel, so the sharded product is one launch with no exchange step in front of it.
struct XHalo {
  int n_own;
  const uint64_t *haddr;
};
// Two steps so that a lane's loads stay batched: first the table entries of all its columns (entry 0
// for own columns: harmless, one cached line), then the values from wherever they live.
__device__ __forceinline__ uint64_t halo_entry(int col, const XHalo &h) { return h.haddr[max(col - h.n_own, 0)]; }
__device__ __forceinline__ uint64_t halo_source(const double *x, int col, uint64_t entry, const XHalo &h) {
  return col < h.n_own ? reinterpret_cast<uint64_t>(x + col) : entry;
}

// Optional epilogue of the merge kernel: dot_part[block] = sum over the block's rows of
// w[row] * y[row] (fixed order => reproducible), so that the p.Ap of a CG iteration costs no
// extra pass over the vectors.  w = NULL switches it off.
struct DotEpilogue {
  const double *w;
  double *dot_part;
};

// ---------------------------------------------------------------- cross-lane
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap16_f64(double v) {      // lane i <-> i^16
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_halves_f64(double v) {  // v[i%32] + v[i%32+32] in every lane
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// Sum over aligned groups of L consecutive lanes; every lane of a group ends
// with the group's total.  Fixed butterfly order => deterministic.  Must be
// called with all 64 lanes active (DPP reads neighbours' registers).
template <int L>
__device__ __forceinline__ double group_sum(double v) {
  if (L >= 2)  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
  if (L >= 4)  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
  if (L >= 8)  v += dpp_f64<0x141>(v);   // row_half_mirror      (other quad of 8)
  if (L >= 16) v += dpp_f64<0x140>(v);   // row_mirror           (other half of 16)
  if (L >= 32) v += swap16_f64(v);       // ds_swizzle SWAP,16
  if (L >= 64) v = sum_halves_f64(v);    // v_permlane32_swap
  return v;
}

// Contiguous row blocks per XCD: hardware deals workgroups round-robin over the
// 8 XCDs (MI355X_MICROARCH "Workgroup dispatch"), so hardware block b lands on
// XCD b%8.  Map it to a logical block so that each XCD walks one contiguous
// eighth of the matrix: neighbouring row blocks share x lines and the partial
// cache lines at their seams in the same 4 MiB L2.  Bijective for any grid.
__device__ __forceinline__ int logical_block(int hw, int n, int remap) {
  if (!remap) return hw;
  const int xcd = hw & 7, idx = hw >> 3;
  const int q = n >> 3, rem = n & 7;
  return xcd * q + (xcd < rem ? xcd : rem) + idx;
}

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// One row's dot product over lanes j, j+L, ... with up to four L-chunks in
// flight; XLDS selects the x source at compile time so the gathers of all four
// chunks issue back to back (a runtime select inside the unrolled loop makes
// hipcc serialise them behind per-chunk waits).
template <int L, bool XLDS, bool NT>
__device__ __forceinline__ double row_dot(int s, int e, int j, const int *__restrict__ ci,
                                          const double *__restrict__ val, const double *__restrict__ x,
                                          const double *xs, int cmin) {
  double acc = 0.0;
  int k = (s & ~(L - 1)) + j;
  if (k < s) k += L;
  for (; k < e; k += 4 * L) {
    int c[4];
    double v[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int kk = min(k + u * L, e - 1);               // clamped: always a valid element of this row
      c[u] = stream_load<NT>(ci + kk);
      v[u] = stream_load<NT>(val + kk);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) xv[u] = XLDS ? xs[c[u] - cmin] : x[c[u]];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k + u * L < e) acc = fma(v[u], xv[u], acc);
  }
  return acc;
}

// ------------------------------------------------------------ vector variant
// "Wavefront per row" family: L lanes of a wave own one row (rows per wavefront
// = 64/L; L = 1 is thread-per-row).  Loads start on an L-aligned element so a
// wave-instruction covers whole 128-byte lines.  Up to 4 L-chunks of a row are
// in flight per lane.  x comes from an LDS window staged per workgroup when
// the workgroup's column span fits the tile (LDSX), else from L2.
template <int L, bool LDSX, bool NT>
__global__ void k_spmv_vector(int n_rows, int n_wg, int remap, int tile_width,
                              const int2v *__restrict__ xspan,
                              const int *__restrict__ rp, const int *__restrict__ ci,
                              const double *__restrict__ val, const double *__restrict__ x,
                              double *__restrict__ y) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *xs = reinterpret_cast<double *>(smem);
  const int tid = threadIdx.x;
  const int wg = logical_block(blockIdx.x, n_wg, remap);
  const int rows_per_wg = blockDim.x / L;
  const int row = wg * rows_per_wg + tid / L;
  const int j = tid & (L - 1);

  int cmin = 0;
  bool in_lds = false;
  if (LDSX) {
    const int2v span = xspan[wg];
    cmin = span.x;
    in_lds = span.y > 0 && span.y <= tile_width;          // workgroup-uniform
    if (in_lds) {
      for (int i = tid; i < span.y; i += blockDim.x) xs[i] = x[cmin + i];
      __syncthreads();
    }
  }

  double acc = 0.0;
  if (row < n_rows) {
    const int s = rp[row], e = rp[row + 1];
    if (LDSX && in_lds) acc = row_dot<L, true, NT>(s, e, j, ci, val, x, xs, cmin);
    else                acc = row_dot<L, false, NT>(s, e, j, ci, val, x, xs, cmin);
  }
  acc = group_sum<L>(acc);
  if (j == 0 && row < n_rows) y[row] = acc;
}

// ------------------------------------------------------------- merge variant
// Merge-based family.  The host cuts the merge path of (row ends) x (nonzero
// indices) into pieces of at most CAP = blockDim.x*IPT items and snaps each
// cut to a row boundary, so every workgroup owns whole rows and the same
// amount of work whatever the row-length distribution; rows longer than a
// threshold become their own "long row" pieces.  Phase 1 streams the block's
// nonzeros with 16-byte loads in nonzero order (perfectly coalesced, IPT
// elements in flight per lane, independent of row structure) and parks the
// products in LDS; phase 2 sums each row's run of products with G lanes per
// row (G chosen per block from its mean row length) and a DPP butterfly.
// Rows longer than SKEW_FACTOR*G products are left to a second pass in which a whole wave sums one row
// (power-law blocks: a 500-nonzero row among 3-nonzero rows would otherwise keep one lane busy for
// microseconds while 255 idle).  The host flags such blocks (KIND_SKEW); others pay one compare.
constexpr int SKEW_FACTOR = 32;

template <int G>
__device__ __forceinline__ double reduce_rows_plain(const BlockDesc &d, const double *prod, const int *roff,
                                                    double *__restrict__ y, const double *__restrict__ w) {
  const int tid = threadIdx.x;
  const int rows_per_pass = blockDim.x / G;
  const int j = tid & (G - 1);
  double dsum = 0.0;
  for (int r0 = 0; r0 < d.n_rows; r0 += rows_per_pass) {
    const int r = r0 + tid / G;
    double acc = 0.0;
    if (r < d.n_rows) {
      const int s = roff[r], e = roff[r + 1];
#pragma unroll 4
      for (int k = s + j; k < e; k += G) acc += prod[k];
    }
    acc = group_sum<G>(acc);
    if (j == 0 && r < d.n_rows) {
      y[d.row_start + r] = acc;
      if (w) dsum = fma(w[d.row_start + r], acc, dsum);       // workgroup-uniform test
    }
  }
  return dsum;
}

template <int G, bool SKEW>
__device__ __forceinline__ double reduce_rows(const BlockDesc &d, const double *prod, const int *roff,
                                              double *__restrict__ y, const double *__restrict__ w) {
  const bool skew = SKEW && (d.kind_g & KIND_SKEW);           // workgroup-uniform
  if (!skew)                                                  // the common case keeps the lean loop
    return reduce_rows_plain<G>(d, prod, roff, y, w);
  const int tid = threadIdx.x;
  const int rows_per_pass = blockDim.x / G;
  const int j = tid & (G - 1);
  double dsum = 0.0;
  // pass 1: rows of ordinary length, G lanes each; long rows are left to pass 2
  for (int r0 = 0; r0 < d.n_rows; r0 += rows_per_pass) {
    const int r = r0 + tid / G;
    double acc = 0.0;
    bool mine = r < d.n_rows;
    if (mine) {
      const int s = roff[r], e = roff[r + 1];
      if (e - s > SKEW_FACTOR * G) {
        mine = false;
      } else {
#pragma unroll 4
        for (int k = s + j; k < e; k += G) acc += prod[k];
      }
    }
    acc = group_sum<G>(acc);
    if (j == 0 && mine) {
      y[d.row_start + r] = acc;
      if (w) dsum = fma(w[d.row_start + r], acc, dsum);
    }
  }
  // pass 2: a whole wave per long row.  Every wave scans the row lengths 64 at a time (one ballot per
  // chunk) and takes the long rows round-robin IN ROW ORDER, so which wave sums which row -- and with it
  // the order of every floating-point addition -- is a function of the matrix alone (no queue, no atomics).
  const int lane = tid & 63, wave = tid >> 6, wave_mask = (blockDim.x >> 6) - 1;
  int seen = 0;
  for (int c0 = 0; c0 < d.n_rows; c0 += 64) {                 // workgroup-uniform
    const int r = c0 + lane;
    const bool is_long = r < d.n_rows && roff[r + 1] - roff[r] > SKEW_FACTOR * G;
    unsigned long long todo = __ballot(is_long);
    while (todo) {                                            // wave-uniform
      const int b = __builtin_ctzll(todo);
      todo &= todo - 1;
      if (((seen++) & wave_mask) != wave) continue;
      const int row = c0 + b;
      const int s = roff[row], e = roff[row + 1];
      double a0 = 0.0, a1 = 0.0;
      int k = s + lane;
      for (; k + 64 < e; k += 128) {
        a0 += prod[k];
        a1 += prod[k + 64];
      }
      if (k < e) a0 += prod[k];
      const double acc = group_sum<64>(a0 + a1);
      if (lane == 0) {
        y[d.row_start + row] = acc;
        if (w) dsum = fma(w[d.row_start + row], acc, dsum);
      }
    }
  }
  return dsum;
}

// One block of the merge kernel, straight-line so that hipcc can count the
// outstanding loads exactly.  Issue order is the point (membench2, stage 3 vs
// 4: 11.0 -> 9.2 us on the cant payload):
//   1. the block's x window (XU 8-byte loads per lane)      -- oldest
//   2. its row offsets (2 loads per lane)
//   3. the value/index stream (IPT/2 16-byte + 8-byte loads) -- youngest
// vmcnt retires in order, so parking the window and the offsets in LDS waits
// only for (1) and (2) while the stream is still in flight; when the stream
// lands the gathers are LDS reads (~100 ns, no TA traffic) instead of a second
// dependent trip to L2.  XU = 0 gathers from L2 (window wider than the tile).
//
// C16 (only with an x window): the block's column indices are read from the
// 16-bit side array (col_ind - cmin, built at plan time), 2 instead of 4 bytes
// per nonzero and already LDS offsets -- the stream shrinks from 12 to 10
// bytes per nonzero, which on a bandwidth-bound kernel is the whole game.
//
// Sharded product (SEAM = true, only for blocks that read halo columns; cask_hip_p2p.h): the block
// fetches the address-table entries of its window first, issues its stream like any block, then the
// window loads themselves -- some of them remote: one local and one xGMI round trip, overlapped with
// the stream.  A separate instantiation, so the code of every other block is exactly the one above.
typedef __attribute__((address_space(1))) const double gdouble;
__device__ __forceinline__ double load_at(uint64_t addr) { return *reinterpret_cast<gdouble *>(addr); }

template <int IPT, int XU, bool NT, bool C16, bool SEAM>
__device__ __forceinline__ void merge_load(const BlockDesc &d, int n_cols, int xlim, int max_gpair,
                                           const int *__restrict__ rp, const int *__restrict__ ci,
                                           const unsigned *__restrict__ ci16, const int *__restrict__ xchunk,
                                           const double *__restrict__ val, const double *__restrict__ x,
                                           double *prod, int *roff, double *xs, const XHalo &halo) {
  const int WG = blockDim.x, tid = threadIdx.x;
  // 16-byte loads need an even element index: start one element early if the
  // block starts on an odd nonzero (that element belongs to the previous block;
  // its product lands in prod[0] and no row of this block references it).
  const int base = d.nnz_start & ~1;
  const int lead = d.nnz_start - base;
  const int total = d.nnz_count + lead;
  // An odd total ends in a pair whose second element is foreign (the next
  // block's first nonzero, or -- for the very last nonzero of an odd-nnz matrix
  // -- the 8 bytes after the array: a 16-byte-aligned 16-byte load that holds
  // one valid element cannot cross a page, so it is memory-safe).
  const int npairs = (total + 1) >> 1;

  // x tile.  With 16-bit indices the tile is a SET of column ranges cut into 64-column chunks
  // (xchunk[c] = first column of chunk c; built on the host): chunk c = u*(WG/64) + wave lands in
  // LDS slots [64c, 64c+64), and a nonzero's 16-bit index is its slot.  One contiguous window is
  // the special case of consecutive chunks; stencil-like matrices (a few narrow bands far apart)
  // fit the same way.  Without 16-bit indices the tile is the contiguous window [cmin, cmin+cwidth).
  double xw[XU > 0 ? XU : 1];
  uint64_t xsrc[XU > 0 ? XU : 1];
  if (XU > 0) {
    if (SEAM) {
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, wpw = WG >> 6;
      const bool chunked = C16 && !(d.kind_g & KIND_CONTIG);
      int col[XU > 0 ? XU : 1];
#pragma unroll
      for (int u = 0; u < XU; u++) {
        col[u] = min(chunked ? xchunk[u * wpw + wave] + lane : d.cmin + u * WG + tid, n_cols - 1);
        xsrc[u] = halo_entry(col[u], halo);
      }
#pragma unroll
      for (int u = 0; u < XU; u++) xsrc[u] = halo_source(x, col[u], xsrc[u], halo);
    } else if (C16 && !(d.kind_g & KIND_CONTIG)) {            // workgroup-uniform
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, wpw = WG >> 6;
#pragma unroll
      for (int u = 0; u < XU; u++) xw[u] = x[min(xchunk[u * wpw + wave] + lane, xlim)];
    } else {                                                  // one window: no chunk table on the critical path
#pragma unroll
      for (int u = 0; u < XU; u++) xw[u] = x[min(d.cmin + u * WG + tid, xlim)];
    }
  }
  const int ro0 = rp[d.row_start + min(tid, d.n_rows)] - base;
  const int ro1 = rp[d.row_start + min(tid + WG, d.n_rows)] - base;

  dbl2 v[IPT / 2];
  int2v c[IPT / 2];
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *ci2 = reinterpret_cast<const int2v *>(ci);
  const int first = base >> 1;
  const int last = min(first + max(npairs - 1, 0), max_gpair);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int p = min(first + u * WG + tid, last);           // clamped: redundant loads hit the same line
    v[u] = stream_load<NT>(val2 + p);
    if (C16) {
      const unsigned w = stream_load<NT>(ci16 + p);
      c[u].x = (int)(w & 0xffffu);
      c[u].y = (int)(w >> 16);
    } else {
      c[u] = stream_load<NT>(ci2 + p);
    }
  }
  if (XU > 0 && SEAM) {
#pragma unroll
    for (int u = 0; u < XU; u++) xw[u] = load_at(xsrc[u]);
  }

  CASK_STAMP(1);
  if (XU > 0) {
#pragma unroll
    for (int u = 0; u < XU; u++) xs[u * WG + tid] = xw[u];
  }
  roff[tid] = ro0;
  roff[tid + WG] = ro1;
  if (XU > 0) __syncthreads();
  CASK_STAMP(2);

  // foreign elements: give them a column this block owns, so their gather stays
  // inside the x window / inside x (their products land in slots no row uses)
  if (lead && tid == 0) c[0].x = c[0].y;
  if (total & 1) {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++)
      if (u * WG + tid >= npairs - 1) c[u].y = c[u].x;
  }
  dbl2 xv[IPT / 2];
  if (XU > 0) {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = xs[C16 ? c[u].x : c[u].x - d.cmin];
      xv[u].y = xs[C16 ? c[u].y : c[u].y - d.cmin];
    }
  } else if (SEAM) {                                          // gathers, some of them from peers
    uint64_t ex[IPT / 2], ey[IPT / 2];
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      ex[u] = halo_entry(c[u].x, halo);
      ey[u] = halo_entry(c[u].y, halo);
    }
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = load_at(halo_source(x, c[u].x, ex[u], halo));
      xv[u].y = load_at(halo_source(x, c[u].y, ey[u], halo));
    }
  } else {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = x[c[u].x];
      xv[u].y = x[c[u].y];
    }
  }
  // every lane stores: lanes past the last pair hold a duplicate of it and land
  // in slots no row offset points to
  dbl2 *prod2 = reinterpret_cast<dbl2 *>(prod);
#ifdef CASK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CASK_STAMP(3);
#endif
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) prod2[u * WG + tid] = v[u] * xv[u];
  __syncthreads();
  CASK_STAMP(4);
}

template <int IPT, int XU, bool NT, bool C16, bool SKEW>
__device__ __forceinline__ void merge_block(const BlockDesc &d, int n_cols, int xlim, int max_gpair,
                                            const int *__restrict__ rp, const int *__restrict__ ci,
                                            const unsigned *__restrict__ ci16, const int *__restrict__ xchunk,
                                            const double *__restrict__ val, const double *__restrict__ x,
                                            double *__restrict__ y, double *prod, int *roff, double *xs,
                                            const XHalo &halo, const DotEpilogue &dot, int lb) {
  const int WG = blockDim.x, tid = threadIdx.x;
  // a seam block's largest column (d.aux, set by the planner when there is a halo) is a halo column
  if (halo.haddr != nullptr && d.aux >= halo.n_own)           // workgroup-uniform; never taken without a halo
    merge_load<IPT, XU, NT, C16, true>(d, n_cols, xlim, max_gpair, rp, ci, ci16, xchunk, val, x, prod, roff, xs, halo);
  else
    merge_load<IPT, XU, NT, C16, false>(d, n_cols, xlim, max_gpair, rp, ci, ci16, xchunk, val, x, prod, roff, xs, halo);

  double dsum;
  switch (d.kind_g & 0xff) {
    case 1:  dsum = reduce_rows<1, SKEW>(d, prod, roff, y, dot.w); break;
    case 2:  dsum = reduce_rows<2, SKEW>(d, prod, roff, y, dot.w); break;
    case 4:  dsum = reduce_rows<4, SKEW>(d, prod, roff, y, dot.w); break;
    case 8:  dsum = reduce_rows<8, SKEW>(d, prod, roff, y, dot.w); break;
    case 16: dsum = reduce_rows<16, SKEW>(d, prod, roff, y, dot.w); break;
    case 32: dsum = reduce_rows<32, SKEW>(d, prod, roff, y, dot.w); break;
    default: dsum = reduce_rows<64, SKEW>(d, prod, roff, y, dot.w); break;
  }
  if (dot.w) {                                                // launch-uniform: the block's share of w.y
    dsum = group_sum<64>(dsum);
    __syncthreads();                                          // every wave is done reading prod
    if ((tid & 63) == 0) prod[tid >> 6] = dsum;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int wv = 0; wv < (WG >> 6); wv++) s += prod[wv];
      dot.dot_part[lb] = s;
    }
  }
}

// SKEW: the plan holds blocks flagged KIND_SKEW (matrices without any run the instantiation that
// carries no second-pass code at all: 0.13 us per launch on cant).
template <int IPT, int XU, bool NT, bool C16, bool SKEW>
__global__ void k_spmv_merge(const BlockDesc *__restrict__ blocks, int n_blocks, int remap, int n_cols, int nnz,
                             const int *__restrict__ rp, const int *__restrict__ ci,
                             const unsigned *__restrict__ ci16, const int *__restrict__ xchunk, int maxch,
                             const double *__restrict__ val, const double *__restrict__ x,
                             double *__restrict__ y, double *__restrict__ partials, XHalo halo, DotEpilogue dot) {
  static_assert(IPT % 2 == 0, "items per thread must be even (16-byte loads)");
  extern __shared__ __align__(16) unsigned char smem[];
  const int WG = blockDim.x, CAP = WG * IPT, tid = threadIdx.x;
  double *prod = reinterpret_cast<double *>(smem);            // CAP + 2 doubles
  int *roff = reinterpret_cast<int *>(prod + CAP + 2);        // 2*WG ints (a block has < 2*WG rows)
  double *xs = reinterpret_cast<double *>(roff + 2 * WG);     // XU*WG doubles

  CASK_STAMP(0);
  const int lb = logical_block(blockIdx.x, n_blocks, remap);
  const BlockDesc d = blocks[lb];
  const int *my_chunks = C16 ? xchunk + (size_t)lb * maxch : nullptr;

  if (d.kind_g & KIND_LONG) {
    // One piece of one long row: the whole workgroup strides over it.
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
      if (halo.haddr) {                                       // launch-uniform
        uint64_t ent[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ent[u] = halo_entry(c[u], halo);
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = load_at(halo_source(x, c[u], ent[u], halo));
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = x[c[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (k + u * WG < end) acc = fma(v[u], xv[u], acc);
    }
    acc = group_sum<64>(acc);
    if ((tid & 63) == 0) prod[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < (WG >> 6); w++) s += prod[w];
      if (d.kind_g & KIND_PARTIAL) {
        partials[d.aux] = s;
        if (dot.w) dot.dot_part[lb] = 0.0;                    // the fix-up kernel owns this row's share
      } else {
        y[d.row_start] = s;
        if (dot.w) dot.dot_part[lb] = dot.w[d.row_start] * s;
      }
    }
    return;
  }

  const int max_gpair = ((nnz + 1) >> 1) - 1;
  // last entry of x[] a block without halo columns may touch: its window is padded to whole chunks and
  // may reach past its largest column, but never past the caller's n_own entries
  const int xlim = min(n_cols, halo.n_own) - 1;
  if (XU > 0 && d.cwidth > 0 && d.cwidth <= XU * WG)          // workgroup-uniform
    merge_block<IPT, XU, NT, C16, SKEW>(d, n_cols, xlim, max_gpair, rp, ci, ci16, my_chunks, val, x, y, prod, roff, xs,
                                        halo, dot, lb);
  else
    merge_block<IPT, 0, NT, false, SKEW>(d, n_cols, xlim, max_gpair, rp, ci, ci16, my_chunks, val, x, y, prod, roff, xs,
                                         halo, dot, lb);
  CASK_STAMP(5);
}

// --------------------------------------------- merge variant, pipelined waves
// Same merge-path decomposition, but the unit of work is a WAVE and the waves
// are persistent: each wave walks a strided list of small blocks (at most
// 64*IPT items, at most 127 rows) and software-pipelines them -- while block b
// is gathered, multiplied and row-reduced, the 16-byte value/index stream and
// the row offsets of block b+1 are already in flight, and the descriptor of
// b+2 is being fetched.  A one-shot workgroup pays the whole dependent chain
// descriptor -> row_ptr -> stream -> gather -> reduce (~6 us on MI355X) once
// per ~2 rounds; here it is paid once per wave and hidden afterwards.  No
// workgroup barrier: a wave only talks to itself through its own LDS slice
// (DS operations of one wave execute in order).
struct WaveStream {            // one block's worth of loads held in registers
  int ro0, ro1;                // row offsets for rows lane, lane+64 (relative to base)
};

template <int G>
__device__ __forceinline__ void wave_reduce_rows(int n_rows, int row_start, bool skew, const double *prod,
                                                 const int *roff, double *__restrict__ y) {
  const int lane = threadIdx.x & 63;
  constexpr int rows_per_pass = 64 / G;
  const int j = lane & (G - 1);
  for (int r0 = 0; r0 < n_rows; r0 += rows_per_pass) {
    const int r = r0 + lane / G;
    double acc = 0.0;
    bool mine = r < n_rows;
    if (mine) {
      const int s = roff[r], e = roff[r + 1];
      if (skew && e - s > SKEW_FACTOR * G) {
        mine = false;
      } else {
#pragma unroll 4
        for (int k = s + j; k < e; k += G) acc += prod[k];
      }
    }
    acc = group_sum<G>(acc);
    if (j == 0 && mine) y[row_start + r] = acc;
  }
  if (skew) {
    for (int r = 0; r < n_rows; r++) {                        // wave-uniform
      const int s = roff[r], e = roff[r + 1];
      if (e - s <= SKEW_FACTOR * G) continue;
      double a = 0.0;
      for (int k = s + lane; k < e; k += 64) a += prod[k];
      a = group_sum<64>(a);
      if (lane == 0) y[row_start + r] = a;
    }
  }
}

template <int IPT, bool NT>
__device__ __forceinline__ void wave_issue(const BlockDesc &d, int max_gpair, const int *__restrict__ rp,
                                           const int *__restrict__ ci, const double *__restrict__ val,
                                           dbl2 (&v)[IPT / 2], int2v (&c)[IPT / 2], int &ro0, int &ro1) {
  const int lane = threadIdx.x & 63;
  const int base = d.nnz_start & ~1;
  const int npairs = (d.nnz_count + d.nnz_start - base + 1) >> 1;   // an odd total ends in a half-foreign pair
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *ci2 = reinterpret_cast<const int2v *>(ci);
  // Lanes past the block's last pair re-read that pair (same cache line); the
  // clamp to max_gpair keeps an empty block at the very end of the arrays
  // inside the allocation.
  const int first = base >> 1;
  const int last = min(first + max(npairs - 1, 0), max_gpair);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int p = min(first + u * 64 + lane, last);
    v[u] = stream_load<NT>(val2 + p);
    c[u] = stream_load<NT>(ci2 + p);
  }
  ro0 = rp[d.row_start + min(lane, d.n_rows)] - base;
  ro1 = rp[d.row_start + min(lane + 64, d.n_rows)] - base;
}

template <int IPT, bool NT, bool HAS_NEXT>
__device__ __forceinline__ void wave_block(const BlockDesc &d, const BlockDesc &dn, int max_gpair,
                                           const int *__restrict__ rp, const int *__restrict__ ci,
                                           const double *__restrict__ val, const double *__restrict__ x,
                                           double *__restrict__ y, double *prod, int *roff,
                                           dbl2 (&v)[IPT / 2], int2v (&c)[IPT / 2], int &ro0, int &ro1) {
  const int lane = threadIdx.x & 63;
  const int base = d.nnz_start & ~1;
  const int total = d.nnz_count + d.nnz_start - base;
  // 1. gathers of the current block (their indices arrived while the previous block was reduced);
  //    the half-foreign tail pair of an odd block gathers a column this block owns
  if (total & 1) {
    const int npairs = (total + 1) >> 1;
#pragma unroll
    for (int u = 0; u < IPT / 2; u++)
      if (u * 64 + lane >= npairs - 1) c[u].y = c[u].x;
  }
  dbl2 xv[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    xv[u].x = x[c[u].x];
    xv[u].y = x[c[u].y];
  }
  const int cur_ro0 = ro0, cur_ro1 = ro1;
  dbl2 pv[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) pv[u] = v[u];
  // 2. next block's stream goes out BEHIND the gathers (vmcnt retires in order:
  //    waiting for the gathers must not wait for these)
  if (HAS_NEXT) wave_issue<IPT, NT>(dn, max_gpair, rp, ci, val, v, c, ro0, ro1);
  // 3. row offsets and products into this wave's LDS slice
  roff[lane] = cur_ro0;
  roff[lane + 64] = cur_ro1;
  // Every lane stores (lanes past the block's last pair hold a duplicate of it
  // and land in slots no row offset points to): a conditional store would let
  // hipcc sink the gathers into the branch, behind the prefetch, and the wait
  // for them would then drain the prefetch too.
  dbl2 *prod2 = reinterpret_cast<dbl2 *>(prod);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) prod2[u * 64 + lane] = pv[u] * xv[u];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // 4. per-row sums
  switch (d.kind_g & 0xff) {
    case 1:  wave_reduce_rows<1>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 2:  wave_reduce_rows<2>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 4:  wave_reduce_rows<4>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 8:  wave_reduce_rows<8>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 16: wave_reduce_rows<16>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 32: wave_reduce_rows<32>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    default: wave_reduce_rows<64>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int IPT, bool NT>
__global__ void k_spmv_merge_wave(const BlockDesc *__restrict__ blocks, int n_blocks, int remap, int nnz,
                                  const int *__restrict__ rp, const int *__restrict__ ci,
                                  const double *__restrict__ val, const double *__restrict__ x,
                                  double *__restrict__ y) {
  static_assert(IPT % 2 == 0, "items per thread must be even (16-byte loads)");
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int CAP = 64 * IPT;
  const int wave = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
  double *prod = reinterpret_cast<double *>(smem) + wave * (CAP + 2);
  int *roff = reinterpret_cast<int *>(reinterpret_cast<double *>(smem) + waves_per_wg * (CAP + 2)) + wave * 128;

  // Which blocks does this wave walk?  With the XCD remap every XCD owns one
  // contiguous eighth of the block list and its resident waves sweep it
  // together (stride = waves on that XCD), so concurrently processed blocks are
  // neighbours: shared x lines and seam cache lines stay in that XCD's L2.
  int b, b_end, stride;
  if (remap) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int gq = gridDim.x >> 3, gr = gridDim.x & 7;
    const int wgs_here = gq + (xcd < gr ? 1 : 0);
    const int nq = n_blocks >> 3, nr = n_blocks & 7;
    const int xb0 = xcd * nq + min(xcd, nr);
    b_end = xb0 + nq + (xcd < nr ? 1 : 0);
    stride = wgs_here * waves_per_wg;
    b = xb0 + idx * waves_per_wg + wave;
  } else {
    stride = gridDim.x * waves_per_wg;
    b = blockIdx.x * waves_per_wg + wave;
    b_end = n_blocks;
  }
  b = __builtin_amdgcn_readfirstlane(b);
  if (b >= b_end) return;

  dbl2 v[IPT / 2];
  int2v c[IPT / 2];
  int ro0, ro1;
  const int max_gpair = ((nnz + 1) >> 1) - 1;
  BlockDesc d = blocks[b];
  wave_issue<IPT, NT>(d, max_gpair, rp, ci, val, v, c, ro0, ro1);
  int nb = b + stride;
  BlockDesc dn = blocks[min(nb, b_end - 1)];
  while (nb < b_end) {
    const int nnb = nb + stride;
    const BlockDesc dnn = blocks[min(nnb, b_end - 1)];       // descriptor two blocks ahead
    wave_block<IPT, NT, true>(d, dn, max_gpair, rp, ci, val, x, y, prod, roff, v, c, ro0, ro1);
    d = dn;
    dn = dnn;
    nb = nnb;
  }
  wave_block<IPT, NT, false>(d, d, max_gpair, rp, ci, val, x, y, prod, roff, v, c, ro0, ro1);
}

// Long-row pieces as their own launch (the pipelined kernel handles only
// whole-row blocks): one workgroup per piece.
template <bool NT>
__global__ void k_spmv_long(const BlockDesc *__restrict__ blocks, int n_blocks,
                            const int *__restrict__ ci, const double *__restrict__ val,
                            const double *__restrict__ x, double *__restrict__ y,
                            double *__restrict__ partials) {
  __shared__ double red[16];
  const int tid = threadIdx.x, WG = blockDim.x;
  const BlockDesc d = blocks[blockIdx.x];
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
    for (int u = 0; u < 4; u++) xv[u] = x[c[u]];
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
    if (d.kind_g & KIND_PARTIAL) partials[d.aux] = s; else y[d.row_start] = s;
  }
}

// Sums the pieces of rows that were split over several workgroups, in piece
// order (deterministic; no float atomics anywhere in the engine).
__global__ void k_spmv_fixup(const SplitRow *__restrict__ rows, int n, const double *__restrict__ partials,
                             double *__restrict__ y, DotEpilogue dot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SplitRow r = rows[i];
  double s = 0.0;
  for (int k = 0; k < r.n_slots; k++) s += partials[r.first_slot + k];
  y[r.row] = s;
  if (dot.w) dot.dot_part[i] = dot.w[r.row] * s;              // slots behind the blocks' (caller offsets the pointer)
}

// ------------------------------------------------------------ plan helpers
// Column span [min,max] of a run of nonzeros, one workgroup per run; fills the
// x-window fields used by the LDSX paths.
__global__ void k_col_span_blocks(BlockDesc *blocks, int n_blocks, const int *__restrict__ ci) {
  __shared__ int smin[64], smax[64];
  const int b = blockIdx.x;
  if (b >= n_blocks) return;
  const int s = blocks[b].nnz_start, e = s + blocks[b].nnz_count;
  int lo = INT32_MAX, hi = -1;
  for (int k = s + threadIdx.x; k < e; k += blockDim.x) {
    const int c = ci[k];
    lo = min(lo, c); hi = max(hi, c);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) { lo = min(lo, smin[w]); hi = max(hi, smax[w]); }
    blocks[b].cmin = hi < 0 ? 0 : lo;
    blocks[b].cwidth = hi < 0 ? 0 : hi - lo + 1;
  }
}

__global__ void k_col_span_rows(int2v *xspan, int n_wg, int rows_per_wg, int n_rows,
                                const int *__restrict__ rp, const int *__restrict__ ci) {
  __shared__ int smin[64], smax[64];
  const int b = blockIdx.x;
  if (b >= n_wg) return;
  const int r0 = b * rows_per_wg, r1 = min(n_rows, r0 + rows_per_wg);
  const int s = rp[r0], e = rp[r1];
  int lo = INT32_MAX, hi = -1;
  for (int k = s + threadIdx.x; k < e; k += blockDim.x) {
    const int c = ci[k];
    lo = min(lo, c); hi = max(hi, c);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) { lo = min(lo, smin[w]); hi = max(hi, smax[w]); }
    int2v out;
    out.x = hi < 0 ? 0 : lo;
    out.y = hi < 0 ? 0 : hi - lo + 1;
    xspan[b] = out;
  }
}

}  // namespace caskhip
