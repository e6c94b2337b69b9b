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
  int32_t aux;         // long-row piece: slot in the partials buffer
};
constexpr int KIND_LONG = 0x100;
constexpr int KIND_PARTIAL = 0x200;

struct SplitRow {      // a row whose pieces are summed by the fix-up kernel
  int32_t row, first_slot, n_slots, pad;
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
template <int G>
__device__ __forceinline__ void reduce_rows(const BlockDesc &d, const double *prod, const int *roff,
                                            double *__restrict__ y) {
  const int tid = threadIdx.x;
  const int rows_per_pass = blockDim.x / G;
  const int j = tid & (G - 1);
  for (int r0 = 0; r0 < d.n_rows; r0 += rows_per_pass) {
    const int r = r0 + tid / G;
    double acc = 0.0;
    if (r < d.n_rows) {
      const int s = roff[r], e = roff[r + 1];
#pragma unroll 4
      for (int k = s + j; k < e; k += G) acc += prod[k];
    }
    acc = group_sum<G>(acc);
    if (j == 0 && r < d.n_rows) y[d.row_start + r] = acc;
  }
}

// Phase 1 of the merge kernel: products of the block's nonzeros, in nonzero
// order, into LDS.  All IPT/2 16-byte value loads and 8-byte index loads of a
// lane are issued before the first gather; XLDS is a template parameter for
// the same reason as in row_dot.
template <int IPT, bool XLDS, bool NT>
__device__ __forceinline__ void merge_stream(const BlockDesc &d, int base, int lead, int total_even,
                                             const int *__restrict__ ci, const double *__restrict__ val,
                                             const double *__restrict__ x, const double *xs, double *prod) {
  const int WG = blockDim.x, tid = threadIdx.x;
  dbl2 v[IPT / 2];
  int2v c[IPT / 2];
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val + base);
  const int2v *ci2 = reinterpret_cast<const int2v *>(ci + base);
  const int last_pair = (total_even >> 1) - 1;
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int p = min(u * WG + tid, last_pair);              // clamped: redundant loads hit the same line
    v[u] = stream_load<NT>(val2 + p);
    c[u] = stream_load<NT>(ci2 + p);
  }
  if (lead && tid == 0) c[0].x = d.cmin;                     // foreign element: keep its gather inside the window
  dbl2 xv[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    if (XLDS) {
      xv[u].x = xs[c[u].x - d.cmin];
      xv[u].y = xs[c[u].y - d.cmin];
    } else {
      xv[u].x = x[c[u].x];
      xv[u].y = x[c[u].y];
    }
  }
  dbl2 *prod2 = reinterpret_cast<dbl2 *>(prod);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int p = u * WG + tid;
    if (p <= last_pair) prod2[p] = v[u] * xv[u];
  }
}

template <int IPT, bool LDSX, bool NT>
__global__ void k_spmv_merge(const BlockDesc *__restrict__ blocks, int n_blocks, int remap, int tile_width,
                             const int *__restrict__ rp, const int *__restrict__ ci,
                             const double *__restrict__ val, const double *__restrict__ x,
                             double *__restrict__ y, double *__restrict__ partials) {
  static_assert(IPT % 2 == 0, "items per thread must be even (16-byte loads)");
  extern __shared__ __align__(16) unsigned char smem[];
  const int WG = blockDim.x, CAP = WG * IPT, tid = threadIdx.x;
  double *prod = reinterpret_cast<double *>(smem);            // CAP + 2 doubles
  int *roff = reinterpret_cast<int *>(prod + CAP + 2);        // CAP + 2 ints
  double *xs = reinterpret_cast<double *>(roff + CAP + 2);    // tile_width doubles (LDSX)

  const BlockDesc d = blocks[logical_block(blockIdx.x, n_blocks, remap)];

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
#pragma unroll
      for (int u = 0; u < 4; u++) xv[u] = x[c[u]];
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
      if (d.kind_g & KIND_PARTIAL) partials[d.aux] = s; else y[d.row_start] = s;
    }
    return;
  }

  // 16-byte loads need an even element index: start one element early if the
  // block starts on an odd nonzero (that element belongs to the previous block;
  // its product lands in prod[0] and no row of this block references it).
  const int base = d.nnz_start & ~1;
  const int lead = d.nnz_start - base;
  const int total = d.nnz_count + lead;
  const int total_even = total & ~1;

  for (int i = tid; i <= d.n_rows; i += WG) roff[i] = rp[d.row_start + i] - base;

  bool in_lds = false;
  if (LDSX) {
    in_lds = d.cwidth > 0 && d.cwidth <= tile_width;        // workgroup-uniform
    if (in_lds) {
      for (int i = tid; i < d.cwidth; i += WG) xs[i] = x[d.cmin + i];
      __syncthreads();
    }
  }

  if (total_even > 0) {
    if (LDSX && in_lds) merge_stream<IPT, true, NT>(d, base, lead, total_even, ci, val, x, xs, prod);
    else                merge_stream<IPT, false, NT>(d, base, lead, total_even, ci, val, x, xs, prod);
  }
  if ((total & 1) && tid == 0) {                              // odd tail element
    const int k = total - 1;
    const int cc = ci[base + k];
    const double xx = (LDSX && in_lds) ? xs[cc - d.cmin] : x[cc];
    prod[k] = val[base + k] * xx;
  }
  __syncthreads();

  switch (d.kind_g & 0xff) {
    case 1:  reduce_rows<1>(d, prod, roff, y); break;
    case 2:  reduce_rows<2>(d, prod, roff, y); break;
    case 4:  reduce_rows<4>(d, prod, roff, y); break;
    case 8:  reduce_rows<8>(d, prod, roff, y); break;
    case 16: reduce_rows<16>(d, prod, roff, y); break;
    case 32: reduce_rows<32>(d, prod, roff, y); break;
    default: reduce_rows<64>(d, prod, roff, y); break;
  }
}

// Sums the pieces of rows that were split over several workgroups, in piece
// order (deterministic; no float atomics anywhere in the engine).
__global__ void k_spmv_fixup(const SplitRow *__restrict__ rows, int n, const double *__restrict__ partials,
                             double *__restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SplitRow r = rows[i];
  double s = 0.0;
  for (int k = 0; k < r.n_slots; k++) s += partials[r.first_slot + k];
  y[r.row] = s;
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
