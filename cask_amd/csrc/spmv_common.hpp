// Shared pieces of the SpMV device kernels for MI355X (gfx950, CDNA4): wave-64, DPP reductions,
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

#include "plan_types.hpp"

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

// Where a launch reads x from.  Without a halo every column comes from x[] (n_own = INT_MAX,
// haddr = NULL).  With one (row-sharded product, include/cask_hip_p2p.h) columns >= n_own are
// halo columns: column n_own + j is read from the absolute device address haddr[j], which may
// lie in a peer GPU's shared slice -- the remote load over xGMI happens inside the product
// kernel, so the sharded product is one launch with no exchange step in front of it.
struct XHalo {
  int n_own;
  const uint64_t *haddr;
  int64_t shift;           // bytes added to every halo address: the table points at slot 0 of the peers' shared
                           //   vector allocations, a product on the vector in slot k reads k*stride*8 further on
};
// Two steps so that a lane's loads stay batched: first the table entries of all its columns (entry 0
// for own columns: harmless, one cached line), then the values from wherever they live.
__device__ __forceinline__ uint64_t halo_entry(int col, const XHalo &h) { return h.haddr[max(col - h.n_own, 0)]; }
__device__ __forceinline__ uint64_t halo_source(const double *x, int col, uint64_t entry, const XHalo &h) {
  return col < h.n_own ? reinterpret_cast<uint64_t>(x + col) : entry + (uint64_t)h.shift;
}

// Optional epilogue of the merge kernel: dot_part[block] = sum over the block's rows of
// w[row] * y[row] (fixed order => reproducible), so that the p.Ap of a CG iteration costs no
// extra pass over the vectors.  w = NULL switches it off.
struct DotEpilogue {
  const double *w;
  double *dot_part;
};

// Solver pass (EXT == 2 instantiations of the merge kernel; cask_hip_cg / cask_hip_bicg and their sharded forms).
// The reference's CG pass is  Ap = A p ; alpha = rsold/(p.Ap) ; x += alpha p ; r -= alpha Ap ; rsnew = r.r ;
// p = r + (rsnew/rsold) p  (src/runtime/SparseLinearSolvers.hpp:206-229).  Two of those steps need a sum over
// the whole vector before anything else can proceed (p.Ap and r.r), so a pass is at least two launches -- and it
// is exactly two when the p update rides on the product: the product kernel composes its operand on the fly,
//     x[c] = a[c] + beta * b[c]              (CG: a = r, b = p_old: the value IS p_new[c], same fma everywhere)
// while staging its x window, stores b_new[row] = a[row] + beta*b[row] for the rows it owns (the next pass's
// p_old; ping-pong buffers, so nobody's window sees a half-updated vector), applies the solution update the
// previous pass still owes, xsol[row] += alpha_prev * b[row], and leaves the shares of w.y behind.  beta, the
// convergence test and `iterations` come from the partial sums (or all-reduced scalars) the previous update
// kernel left: every workgroup adds them in the same order and takes the same decision.
struct SolverPass {
  int64_t b_off;             // doubles from an entry of a (the kernel's x argument) to the same entry of b
  double *b_new;             // own rows: b_new[row] = a[row] + beta*b[row]; NULL = not stored
  double *xsol;              // own rows: xsol[row] += alpha_prev * b[row]; NULL = none
  const double *wa, *wb;     // dot operand of own rows: wa[row] + beta*wb[row]; wa == NULL: the composed operand
  const double *part_chk;    // convergence quantity (r.r): n_chk partial sums, or one scalar if n_chk == 0
  const double *part_num;    // beta numerator: n_num partials / one scalar; may equal part_chk (CG)
  const double *den;         // beta denominator (rsold / rho_old), one scalar
  const double *alpha_prev;  // step length of the previous pass, one scalar
  double *num_out;           // workgroup 0 records the numerator here (the next pass's denominator)
  int *done, *iters;         // convergence flag / the reference's `iterations` (workgroup 0 writes)
  double tol2;
  int n_chk, n_num;
  int iter;                  // index of this pass
  int first;                 // pass 0: no test, beta = 0, no solution update
  int final_only;            // no product.  1: end of the solve -- test, `iterations`, the owed solution update;
                             //   2: probe at a host checkpoint -- the same, but the update only if converged
  int sys_scope;             // b_new lives in a slice peers read: store it write-through at system scope
  int secondary;             // second product of the same pass (BiCG's A^T): same scalars, records nothing, and
                             //   leaves the convergence flag to the primary launch that follows it
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

// Sum of an L2-resident partials array (the <= 1024 shares of a BLAS-1 reduction, or the per-block
// shares the product kernel's dot epilogue leaves: a few thousand), computed redundantly by every
// workgroup in the same order => every workgroup sees the bit-identical value.  This replaces a
// separate single-workgroup "final" kernel (4.3 us + a launch boundary per dot in the first version).
// The loads go out eight 16-byte pairs per lane at a time: one L2 round trip for up to 4096 shares with
// 256 threads, not one per share.  Result in all threads.
__device__ __forceinline__ double sum_partials(const double *__restrict__ partials, int n, double *red) {
  const dbl2 *p2 = reinterpret_cast<const dbl2 *>(partials);
  const int n2 = n >> 1, tid = threadIdx.x, wg = blockDim.x;
  double acc = 0.0;
  for (int i0 = 0; i0 < n2; i0 += 8 * wg) {
    dbl2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p2[min(i0 + u * wg + tid, n2 - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (i0 + u * wg + tid < n2) acc += v[u].x + v[u].y;
  }
  if ((n & 1) && tid == 0) acc += partials[n - 1];
  acc = group_sum<64>(acc);
  __syncthreads();                                  // red may still be read by a previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w];
  return s;
}

// A reduced quantity arrives either as n > 0 partial sums (single GPU: every workgroup adds them itself) or,
// n == 0, as one scalar (row-sharded solvers: the partials were summed and all-reduced between the launches).
__device__ __forceinline__ double partials_or_scalar(const double *__restrict__ p, int n, double *red) {
  return n > 0 ? sum_partials(p, n, red) : *p;
}

// The same in two steps, for a launch that has other loads to issue while the sums are on their way: loads return in
// the order they were issued, so whatever everything else waits for is REQUESTED FIRST (PartialsAhead<U>: the first U
// pairs per lane = 2 U blockDim.x partial sums, or the scalar), the bulk loads follow, and partials_finish then waits for
// exactly these (sums beyond the first 2 U blockDim.x are fetched there).  Same additions in the same order as
// sum_partials: same bits.
template <int U>
struct PartialsAhead {
  dbl2 v[U];
};
template <int U>
__device__ __forceinline__ PartialsAhead<U> partials_request(const double *__restrict__ p, int n) {
  PartialsAhead<U> a;
  const dbl2 *p2 = reinterpret_cast<const dbl2 *>(p);
  const int n2 = n >> 1, tid = threadIdx.x, wg = blockDim.x;
  if (n > 1) {                                                // launch-uniform
#pragma unroll
    for (int u = 0; u < U; u++) a.v[u] = p2[min(u * wg + tid, n2 - 1)];
  } else {
    a.v[0].x = p[0];                                          // n == 0: the scalar; n == 1: the only partial sum
  }
  return a;
}
template <int U>
__device__ __forceinline__ double partials_finish(const PartialsAhead<U> &a, const double *__restrict__ p, int n, double *red) {
  if (n <= 0) return a.v[0].x;
  const dbl2 *p2 = reinterpret_cast<const dbl2 *>(p);
  const int n2 = n >> 1, tid = threadIdx.x, wg = blockDim.x;
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; u++)
    if (u * wg + tid < n2) acc += a.v[u].x + a.v[u].y;
  for (int i0 = U * wg; i0 < n2; i0 += 8 * wg) {
    dbl2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p2[min(i0 + u * wg + tid, n2 - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (i0 + u * wg + tid < n2) acc += v[u].x + v[u].y;
  }
  if ((n & 1) && tid == 0) acc += n > 1 ? p[n - 1] : a.v[0].x;
  acc = group_sum<64>(acc);
  __syncthreads();                                  // red may still be read by a previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w];
  return s;
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

// The convergence flag is written by a workgroup of the launch that detects convergence while other workgroups
// of the SAME launch may not have started yet (a grid is not guaranteed to be co-resident) -- and those still owe
// their share of the pass (the solution update).  The flag therefore carries the pass that set it: a launch
// returns early only on a flag some EARLIER launch wrote.  Non-zero = converged for everybody else.
__device__ __forceinline__ int done_tag(int iter) { return iter + 1; }
__device__ __forceinline__ bool done_by_earlier_launch(const int *done, int iter) {
  const int d = *done;
  return d != 0 && d != done_tag(iter);
}

}  // namespace caskhip
