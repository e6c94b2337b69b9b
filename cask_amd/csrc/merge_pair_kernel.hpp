// Two merge blocks per workgroup, software-pipelined: an A/B against k_spmv_merge for plans whose
// every block is a tiled block with 12-bit packed slots (the cant-like and cant3-like plans: IPT = 8, nontemporal
// streams, no skewed / long-row / untiled blocks, no halo, no dot epilogue).
//
// What it is after: in k_spmv_merge every workgroup starts with a dependent ~1 us round trip for its 32-byte block
// descriptor before it can issue a single vector load, and a 2 021-block grid on 1 536 resident slots runs 1.3 rounds
// (phase stamps).  Here the grid is half the blocks (one round, everything resident) and a
// workgroup owns blocks hw and hw + G: both descriptors are requested at entry, and the second block's x window, row
// offsets, slot record and value stream are ISSUED as soon as the first block's products are in LDS -- they fly while
// the first block's rows are reduced and stored -- so the second block pays neither the descriptor trip nor most of
// its load latency.  Cost: the second block's loads live in ~30 more VGPRs across the first block's reduction.
//
// Same arithmetic in the same order as k_spmv_merge<8, XU, true, true, true, WIDE, false, 0, false>: results are
// bit-identical to the MERGE variant on the same plan (tested).
#pragma once
#include "merge_kernel.hpp"

namespace caskhip {

template <int XU>
struct PairLoads {
  double xw[XU];
  int ro0, ro1;
  dbl2 v[4];
  unsigned c12[3];
};

template <int XU, bool WIDE>
__device__ __forceinline__ void pair_issue(const BlockDesc &d, int lb, int xlim, int max_gpair, const int *__restrict__ rp,
                                           const unsigned *__restrict__ ci16, const int *__restrict__ chunks,
                                           const double *__restrict__ val, const double *__restrict__ x, PairLoads<XU> &r) {
  const int WG = blockDim.x, tid = threadIdx.x;
  const int base = d.nnz_start & ~1, lead = d.nnz_start - base, total = d.nnz_count + lead;
  const int npairs = (total + 1) >> 1;
  if (WIDE) {                                                 // one contiguous window, 16-byte pairs
    const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x);
    const int plim = xlim >> 1;
#pragma unroll
    for (int u = 0; u < XU / 2; u++) {
      const dbl2 pr = x2[min((d.cmin >> 1) + u * WG + tid, plim)];
      r.xw[2 * u] = pr.x;
      r.xw[2 * u + 1] = pr.y;
    }
  } else if (!(d.kind_g & KIND_CONTIG)) {                     // chunked tile (workgroup-uniform)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, wpw = WG >> 6;
#pragma unroll
    for (int u = 0; u < XU; u++) r.xw[u] = x[min(chunks[u * wpw + wave] + lane, xlim)];
  } else {
#pragma unroll
    for (int u = 0; u < XU; u++) r.xw[u] = x[min(d.cmin + u * WG + tid, xlim)];
  }
  r.ro0 = rp[d.row_start + min(tid, d.n_rows)] - base;
  r.ro1 = rp[d.row_start + min(tid + WG, d.n_rows)] - base;
  const unsigned *rec = ci16 + ((size_t)lb * WG + tid) * 3;
  r.c12[0] = stream_load<true>(rec);
  r.c12[1] = stream_load<true>(rec + 1);
  r.c12[2] = stream_load<true>(rec + 2);
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val);
  const int first = base >> 1, last = min(first + max(npairs - 1, 0), max_gpair);
#pragma unroll
  for (int u = 0; u < 4; u++) r.v[u] = stream_load<true>(val2 + min(first + u * WG + tid, last));
}

// window + row offsets -> LDS, barrier, gathers from the window, products -> LDS, barrier
template <int XU, bool WIDE>
__device__ __forceinline__ void pair_finish(const PairLoads<XU> &r, double *prod, int *roff, double *xs) {
  const int WG = blockDim.x, tid = threadIdx.x;
  if (WIDE) {
    dbl2 *xs2 = reinterpret_cast<dbl2 *>(xs);
#pragma unroll
    for (int u = 0; u < XU / 2; u++) {
      dbl2 pr;
      pr.x = r.xw[2 * u];
      pr.y = r.xw[2 * u + 1];
      xs2[u * WG + tid] = pr;
    }
  } else {
#pragma unroll
    for (int u = 0; u < XU; u++) xs[u * WG + tid] = r.xw[u];
  }
  roff[tid] = r.ro0;
  roff[tid + WG] = r.ro1;
  __syncthreads();
  const unsigned q0 = r.c12[0], q1 = r.c12[1], q2 = r.c12[2];
  int s[8];
  s[0] = (int)(q0 & 0xfffu);
  s[1] = (int)((q0 >> 12) & 0xfffu);
  s[2] = (int)((q0 >> 24) | ((q1 & 0xfu) << 8));
  s[3] = (int)((q1 >> 4) & 0xfffu);
  s[4] = (int)((q1 >> 16) & 0xfffu);
  s[5] = (int)((q1 >> 28) | ((q2 & 0xffu) << 4));
  s[6] = (int)((q2 >> 8) & 0xfffu);
  s[7] = (int)(q2 >> 20);
  dbl2 *prod2 = reinterpret_cast<dbl2 *>(prod);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    dbl2 xv;
    xv.x = xs[s[2 * u]];
    xv.y = xs[s[2 * u + 1]];
    prod2[u * WG + tid] = r.v[u] * xv;
  }
  __syncthreads();
}

__device__ __forceinline__ void pair_reduce(const BlockDesc &d, const double *prod, const int *roff, double *__restrict__ y) {
  switch (d.kind_g & 0xff) {
    case 1:  reduce_rows_plain<1, 0>(d, prod, roff, y, nullptr); break;
    case 2:  reduce_rows_plain<2, 0>(d, prod, roff, y, nullptr); break;
    case 4:  reduce_rows_plain<4, 0>(d, prod, roff, y, nullptr); break;
    case 8:  reduce_rows_plain<8, 0>(d, prod, roff, y, nullptr); break;
    case 16: reduce_rows_plain<16, 0>(d, prod, roff, y, nullptr); break;
    case 32: reduce_rows_plain<32, 0>(d, prod, roff, y, nullptr); break;
    default: reduce_rows_plain<64, 0>(d, prod, roff, y, nullptr); break;
  }
}

template <int XU, bool WIDE>
__global__ void k_spmv_merge_pair(const BlockDesc *__restrict__ blocks, int n_blocks, int half, int remap, int n_cols, int nnz,
                                  const int *__restrict__ rp, const unsigned *__restrict__ ci16,
                                  const int *__restrict__ xchunk, int maxch, const double *__restrict__ val,
                                  const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int WG = blockDim.x, CAP = WG * 8;
  double *prod = reinterpret_cast<double *>(smem);            // CAP + 2 doubles
  int *roff = reinterpret_cast<int *>(prod + CAP + 2);        // 2*WG ints
  double *xs = reinterpret_cast<double *>(roff + 2 * WG);     // XU*WG doubles
  const int hw0 = blockIdx.x, hw1 = hw0 + half;               // half is a multiple of 8: both blocks on this XCD's share
  if (hw0 >= n_blocks) return;                                // (half is rounded up to 8: a small grid has spare workgroups)
  const bool two = hw1 < n_blocks;                            // workgroup-uniform
  const int lb0 = logical_block(hw0, n_blocks, remap), lb1 = logical_block(two ? hw1 : hw0, n_blocks, remap);
  const BlockDesc d0 = blocks[lb0], d1 = blocks[lb1];         // both descriptors requested at entry
  const int max_gpair = ((nnz + 1) >> 1) - 1, xlim = n_cols - 1;

  PairLoads<XU> a, b;
  pair_issue<XU, WIDE>(d0, lb0, xlim, max_gpair, rp, ci16, xchunk + (size_t)lb0 * maxch, val, x, a);
  pair_finish<XU, WIDE>(a, prod, roff, xs);
  if (two) pair_issue<XU, WIDE>(d1, lb1, xlim, max_gpair, rp, ci16, xchunk + (size_t)lb1 * maxch, val, x, b);   // in flight ...
  pair_reduce(d0, prod, roff, y);                                                                                 // ... across this
  if (!two) return;
  __syncthreads();                                            // block 0's products and row offsets are no longer read
  pair_finish<XU, WIDE>(b, prod, roff, xs);
  pair_reduce(d1, prod, roff, y);
}

}  // namespace caskhip
