// BLAS-1 and fused CG/BiCG update kernels (fp64, HBM bound, 16-byte accesses).
// They stand in for the MKL calls of the reference's CG
// (cblas_ddot/daxpy/daxpby, src/runtime/SparseLinearSolvers.hpp:190-229).
// Scalars (alpha/beta numerators and denominators, the convergence flag) live
// in device memory so a solver iteration never waits for the host; once the
// flag is set every update kernel becomes a no-op, which freezes x at exactly
// the iterate the reference would return (SparseLinearSolvers.hpp:220-226).
// Reductions are two-stage with a fixed grid => bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spmv_common.hpp"

namespace caskhip {

constexpr int BLAS_WG = 256;
constexpr int BLAS_MAX_PARTIALS = 1024;
// The launches of a solver come in two shapes (r6), chosen by the vector length (solver_shape, cask_hip.hip):
//   BIG   (more than SOLVER_GRID_MAX x SOLVER_WG pairs: n > 524 288) -- SOLVER_GRID_MAX workgroups of SOLVER_WG threads: one
//         workgroup per CU of an MI355X, 16 waves each.  Every workgroup of an update launch starts by adding up ALL the
//         partial sums the launch before it left (fixed order: no atomics, no extra launch), i.e. every workgroup reads
//         the same few KB through the same L2 channels: with 1 024 workgroups of 256 threads the 3 888 sums of a
//         G3_circuit-like product launch were 31 KB x 1 024 readers (~2 us at the head of k_cg_update_r: 9.2 us for bytes
//         that stream in 6.8); a quarter of the readers, and a quarter of the sums for the launch that follows.
//   SMALL (anything shorter) -- up to BLAS_MAX_PARTIALS workgroups of BLAS_WG threads, a pair per lane: a cant-like system
//         (31 225 pairs) on 31 workgroups of 1 024 was 7 % slower per CG pass than on 122 of 256.
// UpdShape<BIG>: pairs a lane requests before it waits for the scalars, by the number of vectors the launch walks (every
// kernel within 128 VGPRs: 4 waves per SIMD, the whole BIG grid resident at once), and the partial sums it requests ahead
// of everything, in 16-byte pairs per lane (an update launch leaves one sum per workgroup, a product launch one per
// workgroup of the plan: 1 958 ... 4 314 on the BASELINE matrices).
constexpr int SOLVER_WG = 1024, SOLVER_GRID_MAX = 256;
template <bool BIG> struct UpdShape;
template <> struct UpdShape<true>  { static constexpr int AHEAD = 4, AHEAD_4V = 3, AHEAD_5V = 3, SUMS_OF_UPDATE = 1, SUMS_OF_PRODUCT = 3; };
template <> struct UpdShape<false> { static constexpr int AHEAD = 1, AHEAD_4V = 1, AHEAD_5V = 1, SUMS_OF_UPDATE = 2, SUMS_OF_PRODUCT = 8; };

__device__ __forceinline__ double wg_sum(double v, double *red) {
  v = group_sum<64>(v);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  if (tid == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w];
  return s;   // valid in thread 0
}

__global__ void k_dot_partial(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                              double *__restrict__ partials, const int *done) {
  __shared__ double red[16];
  if (done && *done) return;
  const int64_t n2 = n >> 1;
  const dbl2 *a2 = reinterpret_cast<const dbl2 *>(a);
  const dbl2 *b2 = reinterpret_cast<const dbl2 *>(b);
  double acc0 = 0.0, acc1 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const dbl2 u = a2[i], w = b2[i];
    acc0 = fma(u.x, w.x, acc0);
    acc1 = fma(u.y, w.y, acc1);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc0 = fma(a[n - 1], b[n - 1], acc0);
  const double s = wg_sum(acc0 + acc1, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// Final stage of a dot; optionally the CG convergence test:
// if (*result <= tol2) *done = 1 else *iters = iter  (SparseLinearSolvers.hpp:220-231).
__global__ void k_dot_final(int n_partials, const double *__restrict__ partials, double *__restrict__ result,
                            int check, double tol2, int *done, int *iters, int iter) {
  __shared__ double red[16];
  if (done && *done) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += blockDim.x) acc += partials[i];
  const double s = wg_sum(acc, red);
  if (threadIdx.x == 0) {
    *result = s;
    if (check) {
      if (s <= tol2) *done = 1; else *iters = iter;
    }
  }
}

__device__ __forceinline__ double scalar_ratio(double sign, const double *num, const double *den) {
  double a = sign;
  if (num) a *= *num;
  if (den) a /= *den;
  return a;
}

// y += sign*(num/den)*x
__global__ void k_axpy(int64_t n, double sign, const double *num, const double *den,
                       const double *__restrict__ x, double *__restrict__ y, const int *done) {
  if (done && *done) return;
  const double a = scalar_ratio(sign, num, den);
  const int64_t n2 = n >> 1;
  const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x);
  dbl2 *y2 = reinterpret_cast<dbl2 *>(y);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    dbl2 u = x2[i], w = y2[i];
    w.x = fma(a, u.x, w.x);
    w.y = fma(a, u.y, w.y);
    y2[i] = w;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = fma(a, x[n - 1], y[n - 1]);
}

// y = alpha*x + b*y, b = beta (host) or sign*(num/den) (device) when num != NULL
__global__ void k_axpby(int64_t n, double alpha, const double *__restrict__ x, double sign, double beta,
                        const double *num, const double *den, double *__restrict__ y, const int *done) {
  if (done && *done) return;
  const double b = num ? scalar_ratio(sign, num, den) : beta;
  const int64_t n2 = n >> 1;
  const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x);
  dbl2 *y2 = reinterpret_cast<dbl2 *>(y);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    dbl2 u = x2[i], w = y2[i];
    w.x = alpha * u.x + b * w.x;
    w.y = alpha * u.y + b * w.y;
    y2[i] = w;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = alpha * x[n - 1] + b * y[n - 1];
}

// Set-up of a CG / BiCG solve in one launch (r6; it was three to five device-to-device copies, an axpby and a dot launch, each
// copy a runtime blit with ~7 us around it): r = b - q ; copies of r where the solver wants them (rt, p, pt: NULL = not
// wanted) ; shares of r.r.  r = -q + b and the shares are formed exactly as k_axpby / k_dot_partial formed them (same
// expressions, same grid-stride order): same bits.
__global__ void k_solver_residual(int64_t n, const double *__restrict__ b, const double *__restrict__ q, double *__restrict__ r,
                                  double *__restrict__ c0, double *__restrict__ c1, double *__restrict__ c2,
                                  double *__restrict__ partials) {
  __shared__ double red[16];
  const int64_t n2 = n >> 1;
  const dbl2 *b2 = reinterpret_cast<const dbl2 *>(b), *q2 = reinterpret_cast<const dbl2 *>(q);
  dbl2 *r2 = reinterpret_cast<dbl2 *>(r), *c02 = reinterpret_cast<dbl2 *>(c0), *c12 = reinterpret_cast<dbl2 *>(c1),
       *c22 = reinterpret_cast<dbl2 *>(c2);
  double acc0 = 0.0, acc1 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const dbl2 u = q2[i];
    dbl2 w = b2[i];
    w.x = -1.0 * u.x + 1.0 * w.x;
    w.y = -1.0 * u.y + 1.0 * w.y;
    r2[i] = w;
    if (c0) c02[i] = w;
    if (c1) c12[i] = w;
    if (c2) c22[i] = w;
    acc0 = fma(w.x, w.x, acc0);
    acc1 = fma(w.y, w.y, acc1);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double w = -1.0 * q[n - 1] + 1.0 * b[n - 1];
    r[n - 1] = w;
    if (c0) c0[n - 1] = w;
    if (c1) c1[n - 1] = w;
    if (c2) c2[n - 1] = w;
    acc0 = fma(w, w, acc0);
  }
  const double sum = wg_sum(acc0 + acc1, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = sum;
}
// ... and the initial guess into the one or two slots the first product reads it from
// (also zeroes the solver's scalars and flags: first launch of a solve, in front of everything that reads them -- two runtime
//  memsets less)
__global__ void k_copy2(int64_t n, const double *__restrict__ src, double *__restrict__ d0, double *__restrict__ d1,
                        double *zero_scalars, int n_scalars, int *zero_flags, int n_flags) {
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < n_scalars; i += blockDim.x) zero_scalars[i] = 0.0;
    for (int i = threadIdx.x; i < n_flags; i += blockDim.x) zero_flags[i] = 0;
  }
  const int64_t n2 = n >> 1;
  const dbl2 *s2 = reinterpret_cast<const dbl2 *>(src);
  dbl2 *a2 = reinterpret_cast<dbl2 *>(d0), *b2 = reinterpret_cast<dbl2 *>(d1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const dbl2 v = s2[i];
    a2[i] = v;
    if (d1) b2[i] = v;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    d0[n - 1] = src[n - 1];
    if (d1) d1[n - 1] = src[n - 1];
  }
}

// The update kernels walk the vectors in 16-byte pairs (vectors are the solver's own 256-byte
// aligned allocations); element n-1 of an odd n is handled by the workgroup/lane that owns the
// slot after the last pair, so the summation order is a function of n and the grid only.
#define CASK_PAIR_LOOP(n2)                                                                     \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n2);                    \
       i += (int64_t)gridDim.x * blockDim.x)
__device__ __forceinline__ bool owns_tail(int64_t n) {
  const int64_t slot = n >> 1;
  return (n & 1) && (int64_t)blockIdx.x == (slot / blockDim.x) % gridDim.x && (int64_t)threadIdx.x == slot % blockDim.x;
}

// CG: alpha = rsold / (p.Ap) ; x += alpha p ; r -= alpha Ap ; partials of r.r
// (SparseLinearSolvers.hpp:208-218 in one pass over the vectors).  p.Ap arrives as partials: of
// k_dot_partial, or of the product kernel's fused epilogue (DotEpilogue); the r.r partials go to a
// second array.
__global__ void k_cg_update_xr(int64_t n, const double *rsold, const double *__restrict__ part_pAp, int n_part,
                               const double *__restrict__ p, const double *__restrict__ Ap,
                               double *__restrict__ x, double *__restrict__ r,
                               double *__restrict__ part_rr, const int *done) {
  __shared__ double red[16];
  if (done && *done) return;
  const double alpha = *rsold / partials_or_scalar(part_pAp, n_part, red);
  const dbl2 *p2 = reinterpret_cast<const dbl2 *>(p), *Ap2 = reinterpret_cast<const dbl2 *>(Ap);
  dbl2 *x2 = reinterpret_cast<dbl2 *>(x), *r2 = reinterpret_cast<dbl2 *>(r);
  double acc0 = 0.0, acc1 = 0.0;
  CASK_PAIR_LOOP(n >> 1) {
    const dbl2 pv = p2[i], av = Ap2[i];
    dbl2 xv = x2[i], rv = r2[i];
    xv.x = fma(alpha, pv.x, xv.x);
    xv.y = fma(alpha, pv.y, xv.y);
    rv.x = fma(-alpha, av.x, rv.x);
    rv.y = fma(-alpha, av.y, rv.y);
    x2[i] = xv;
    r2[i] = rv;
    acc0 = fma(rv.x, rv.x, acc0);
    acc1 = fma(rv.y, rv.y, acc1);
  }
  if (owns_tail(n)) {
    x[n - 1] = fma(alpha, p[n - 1], x[n - 1]);
    const double rn = fma(-alpha, Ap[n - 1], r[n - 1]);
    r[n - 1] = rn;
    acc0 = fma(rn, rn, acc0);
  }
  __syncthreads();
  const double s = wg_sum(acc0 + acc1, red);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = s;
}

// CG: rsnew = r.r (from its partials); converged if rsnew <= tol^2 (then nothing else happens:
// SparseLinearSolvers.hpp:220-226), else iterations = iter and p = r + (rsnew/rsold) p (:229-231).
// Every workgroup takes the same decision from the same sum; workgroup 0 records it.
__global__ void k_cg_update_p(int64_t n, const double *__restrict__ part_rr, int n_part, const double *rsold,
                              double *rsnew_out, double tol2, int iter, const double *__restrict__ r,
                              double *__restrict__ p, int *done, int *iters) {
  __shared__ double red[16];
  if (done_by_earlier_launch(done, iter)) return;
  const double rsnew = partials_or_scalar(part_rr, n_part, red);
  if (rsnew <= tol2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { *rsnew_out = rsnew; *done = done_tag(iter); }
    return;
  }
  const double beta = rsnew / *rsold;
  const dbl2 *r2 = reinterpret_cast<const dbl2 *>(r);
  dbl2 *p2 = reinterpret_cast<dbl2 *>(p);
  CASK_PAIR_LOOP(n >> 1) {
    const dbl2 rv = r2[i];
    dbl2 pv = p2[i];
    pv.x = fma(beta, pv.x, rv.x);
    pv.y = fma(beta, pv.y, rv.y);
    p2[i] = pv;
  }
  if (owns_tail(n)) p[n - 1] = fma(beta, p[n - 1], r[n - 1]);
  if (blockIdx.x == 0 && threadIdx.x == 0) { *rsnew_out = rsnew; *iters = iter; }
}

// ---- the update launches of cask_hip_solve_device (classic and composed passes) ---------------------------
// Vectors that peers read over xGMI (row-sharded solvers) are stored write-through at system scope.
__device__ __forceinline__ void store_pair(dbl2 *p, dbl2 v, int sys_scope) {
  if (sys_scope) {
    double *q = reinterpret_cast<double *>(p);
    __hip_atomic_store(q, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    *p = v;
  }
}
__device__ __forceinline__ void store_one(double *p, double v, int sys_scope) {
  if (sys_scope) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else *p = v;
}

// The four update launches of a classic pass share one shape (r6):
//   * a lane's first AHEAD pairs (UpdShape) are requested BEFORE the scalars are waited for.  The scalars are sums of partial
//     sums (a dependent L2 round trip + a workgroup reduction at the head of every workgroup, ~1 us); the vector loads
//     do not depend on them -- and a pair requested only once alpha is known is one more dependent trip: with 1 024
//     workgroups a lane of the G3_circuit-like system owns 3 pairs, and with only the first one ahead (r4)
//     k_cg_update_r took 9.2 us for bytes that stream in 6.8.  Loads are unconditional with clamped indices (a load
//     under a condition whose result is merged with a default makes the compiler wait at the merge), all inside the
//     one launch-uniform branch `n2 > 0`: nothing is loaded from a vector that has no pair.
//   * same pairs in the same order as the plain grid-stride loop: the sums have the same bits.
#define CASK_UPD_INDEX                                                                                                \
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x, last = n2 - 1;                                 \
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x

// CG, second launch of a pass: alpha = rsold / (p.Ap) ; r -= alpha Ap ; shares of r.r
// (SparseLinearSolvers.hpp:208, 212, 218).  x += alpha p (:210) is applied by the next product launch, which
// reads p anyway; alpha is left in *alpha_out for it.
template <bool JAC, bool BIG>
__global__ void k_cg_update_r(int64_t n, const double *rsold, const double *__restrict__ part_pAp, int n_part,
                              const double *__restrict__ Ap, double *__restrict__ r, double *__restrict__ part_rr,
                              double *alpha_out, const int *done, int sys_scope, const double *__restrict__ dinv) {
  __shared__ double red[16];
  if (*done) return;
  const dbl2 *Ap2 = reinterpret_cast<const dbl2 *>(Ap);
  dbl2 *r2 = reinterpret_cast<dbl2 *>(r);
  const dbl2 *d2 = reinterpret_cast<const dbl2 *>(dinv);      // JAC: the shares are of r.z with z = dinv * r (never stored)
  CASK_UPD_INDEX;
  double alpha, acc0 = 0.0, acc1 = 0.0;
  const PartialsAhead<UpdShape<BIG>::SUMS_OF_PRODUCT> sums = partials_request<UpdShape<BIG>::SUMS_OF_PRODUCT>(part_pAp, n_part);
  if (n2 > 0) {
    dbl2 av[UpdShape<BIG>::AHEAD], rv[UpdShape<BIG>::AHEAD], dv[JAC ? UpdShape<BIG>::AHEAD : 1];
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD; k++) {
      const int64_t j = min(i + k * stride, last);
      av[k] = Ap2[j];
      rv[k] = r2[j];
      if constexpr (JAC) dv[k] = d2[j];
    }
    alpha = *rsold / partials_finish(sums, part_pAp, n_part, red);
    auto one = [&](int64_t j, dbl2 a, dbl2 rr, dbl2 d) {
      rr.x = fma(-alpha, a.x, rr.x);
      rr.y = fma(-alpha, a.y, rr.y);
      store_pair(r2 + j, rr, sys_scope);
      if constexpr (JAC) {
        acc0 = fma(rr.x, rr.x * d.x, acc0);
        acc1 = fma(rr.y, rr.y * d.y, acc1);
      } else {
        acc0 = fma(rr.x, rr.x, acc0);
        acc1 = fma(rr.y, rr.y, acc1);
      }
    };
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD; k++)
      if (i + k * stride < n2) one(i + k * stride, av[k], rv[k], dv[JAC ? k : 0]);
    for (int64_t j = i + UpdShape<BIG>::AHEAD * stride; j < n2; j += stride) one(j, Ap2[j], r2[j], JAC ? d2[j] : dbl2{0.0, 0.0});
  } else {
    alpha = *rsold / partials_finish(sums, part_pAp, n_part, red);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_out = alpha;
  if (owns_tail(n)) {
    const double rn = fma(-alpha, Ap[n - 1], r[n - 1]);
    store_one(r + (n - 1), rn, sys_scope);
    acc0 = fma(rn, JAC ? rn * dinv[n - 1] : rn, acc0);
  }
  __syncthreads();
  const double s = wg_sum(acc0 + acc1, red);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = s;
}

// BiCG, last launch of a pass: alpha = rho / (pt.q) ; r -= alpha q ; rt -= alpha qt ; shares of r.r and rt.r
template <bool BIG>
__global__ void k_bicg_update_r(int64_t n, const double *rho, const double *__restrict__ part_ptq, int n_part,
                                const double *__restrict__ q, const double *__restrict__ qt,
                                double *__restrict__ r, double *__restrict__ rt, double *__restrict__ part_rr,
                                double *__restrict__ part_rho, double *alpha_out, const int *done, int sys_scope) {
  __shared__ double red[16];
  if (*done) return;
  const dbl2 *q2 = reinterpret_cast<const dbl2 *>(q), *qt2 = reinterpret_cast<const dbl2 *>(qt);
  dbl2 *r2 = reinterpret_cast<dbl2 *>(r), *rt2 = reinterpret_cast<dbl2 *>(rt);
  CASK_UPD_INDEX;
  double alpha, a_rr0 = 0.0, a_rr1 = 0.0, a_rho0 = 0.0, a_rho1 = 0.0;
  const PartialsAhead<UpdShape<BIG>::SUMS_OF_PRODUCT> sums = partials_request<UpdShape<BIG>::SUMS_OF_PRODUCT>(part_ptq, n_part);
  if (n2 > 0) {
    dbl2 qv[UpdShape<BIG>::AHEAD_4V], qtv[UpdShape<BIG>::AHEAD_4V], rv[UpdShape<BIG>::AHEAD_4V], rtv[UpdShape<BIG>::AHEAD_4V];
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD_4V; k++) {
      const int64_t j = min(i + k * stride, last);
      qv[k] = q2[j];
      qtv[k] = qt2[j];
      rv[k] = r2[j];
      rtv[k] = rt2[j];
    }
    alpha = *rho / partials_finish(sums, part_ptq, n_part, red);
    auto one = [&](int64_t j, dbl2 qa, dbl2 qta, dbl2 ra, dbl2 rta) {
      ra.x = fma(-alpha, qa.x, ra.x);
      ra.y = fma(-alpha, qa.y, ra.y);
      rta.x = fma(-alpha, qta.x, rta.x);
      rta.y = fma(-alpha, qta.y, rta.y);
      store_pair(r2 + j, ra, sys_scope);
      store_pair(rt2 + j, rta, sys_scope);
      a_rr0 = fma(ra.x, ra.x, a_rr0);
      a_rr1 = fma(ra.y, ra.y, a_rr1);
      a_rho0 = fma(rta.x, ra.x, a_rho0);
      a_rho1 = fma(rta.y, ra.y, a_rho1);
    };
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD_4V; k++)
      if (i + k * stride < n2) one(i + k * stride, qv[k], qtv[k], rv[k], rtv[k]);
    for (int64_t j = i + UpdShape<BIG>::AHEAD_4V * stride; j < n2; j += stride) one(j, q2[j], qt2[j], r2[j], rt2[j]);
  } else {
    alpha = *rho / partials_finish(sums, part_ptq, n_part, red);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_out = alpha;
  if (owns_tail(n)) {
    const double rn = fma(-alpha, q[n - 1], r[n - 1]);
    const double rtn = fma(-alpha, qt[n - 1], rt[n - 1]);
    store_one(r + (n - 1), rn, sys_scope);
    store_one(rt + (n - 1), rtn, sys_scope);
    a_rr0 = fma(rn, rn, a_rr0);
    a_rho0 = fma(rtn, rn, a_rho0);
  }
  __syncthreads();
  const double s1 = wg_sum(a_rr0 + a_rr1, red);
  __syncthreads();
  const double s2 = wg_sum(a_rho0 + a_rho1, red);
  if (threadIdx.x == 0) { part_rr[blockIdx.x] = s1; part_rho[blockIdx.x] = s2; }
}

// Classic passes, last launch: the solution update the pass owes, the convergence test, the direction update --
//   x += alpha p (SparseLinearSolvers.hpp:210; alpha as k_cg_update_r left it) ; rsnew = r.r ; converged if
//   rsnew <= tol^2 (:220-226, nothing after it) ; else iterations = iter, p = r + (rsnew/rsold) p (:229-231).
// x is updated here and not next to r because this launch reads p anyway: one pass over p less per iteration.
template <bool JAC, bool BIG>
__global__ void k_cg_update_px(int64_t n, const double *__restrict__ part_rr, int n_part, const double *rsold,
                               double *rsnew_out, const double *alpha, double tol2, int iter,
                               const double *__restrict__ r, double *__restrict__ p, double *__restrict__ x,
                               int *done, int *iters, int sys_scope, const double *__restrict__ dinv) {
  __shared__ double red[16];
  if (done_by_earlier_launch(done, iter)) return;
  const dbl2 *r2 = reinterpret_cast<const dbl2 *>(r);
  dbl2 *p2 = reinterpret_cast<dbl2 *>(p), *x2 = reinterpret_cast<dbl2 *>(x);
  const dbl2 *d2 = reinterpret_cast<const dbl2 *>(dinv);      // JAC: p = z + beta p with z = dinv * r recomputed here
  CASK_UPD_INDEX;
  double rsnew, a, beta;
  bool stop;
  const PartialsAhead<UpdShape<BIG>::SUMS_OF_UPDATE> sums = partials_request<UpdShape<BIG>::SUMS_OF_UPDATE>(part_rr, n_part);
  if (n2 > 0) {
    dbl2 pv[UpdShape<BIG>::AHEAD], xv[UpdShape<BIG>::AHEAD], rv[UpdShape<BIG>::AHEAD], dv[JAC ? UpdShape<BIG>::AHEAD : 1];
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD; k++) {
      const int64_t j = min(i + k * stride, last);
      pv[k] = p2[j];
      xv[k] = x2[j];
      rv[k] = r2[j];
      if constexpr (JAC) dv[k] = d2[j];
    }
    rsnew = partials_finish(sums, part_rr, n_part, red);
    stop = rsnew <= tol2;
    a = *alpha;
    beta = stop ? 0.0 : rsnew / *rsold;
    auto one = [&](int64_t j, dbl2 pp, dbl2 xx, dbl2 rr, dbl2 d) {
      xx.x = fma(a, pp.x, xx.x);
      xx.y = fma(a, pp.y, xx.y);
      x2[j] = xx;
      if (!stop) {                                            // launch-uniform
        if constexpr (JAC) {
          rr.x *= d.x;
          rr.y *= d.y;
        }
        pp.x = fma(beta, pp.x, rr.x);
        pp.y = fma(beta, pp.y, rr.y);
        store_pair(p2 + j, pp, sys_scope);
      }
    };
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD; k++)
      if (i + k * stride < n2) one(i + k * stride, pv[k], xv[k], rv[k], dv[JAC ? k : 0]);
    for (int64_t j = i + UpdShape<BIG>::AHEAD * stride; j < n2; j += stride) one(j, p2[j], x2[j], r2[j], JAC ? d2[j] : dbl2{0.0, 0.0});
  } else {
    rsnew = partials_finish(sums, part_rr, n_part, red);
    stop = rsnew <= tol2;
    a = *alpha;
    beta = stop ? 0.0 : rsnew / *rsold;
  }
  if (owns_tail(n)) {
    const double pn = p[n - 1];
    x[n - 1] = fma(a, pn, x[n - 1]);
    if (!stop) store_one(p + (n - 1), fma(beta, pn, JAC ? r[n - 1] * dinv[n - 1] : r[n - 1]), sys_scope);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *rsnew_out = rsnew;
    if (stop) *done = done_tag(iter); else *iters = iter;
  }
}

// BiCG: x += alpha p ; converged if r.r <= tol^2 ; else beta = rho_new/rho, p = r + beta p, pt = rt + beta pt
template <bool BIG>
__global__ void k_bicg_update_px(int64_t n, const double *__restrict__ part_rr, const double *__restrict__ part_rho,
                                 int n_part, const double *rho, double *rho_out, const double *alpha, double tol2,
                                 int iter, const double *__restrict__ r, const double *__restrict__ rt,
                                 double *__restrict__ p, double *__restrict__ pt, double *__restrict__ x, int *done,
                                 int *iters, int sys_scope) {
  __shared__ double red[16];
  if (done_by_earlier_launch(done, iter)) return;
  const dbl2 *r2 = reinterpret_cast<const dbl2 *>(r), *rt2 = reinterpret_cast<const dbl2 *>(rt);
  dbl2 *p2 = reinterpret_cast<dbl2 *>(p), *pt2 = reinterpret_cast<dbl2 *>(pt), *x2 = reinterpret_cast<dbl2 *>(x);
  CASK_UPD_INDEX;
  double rho_new = 0.0, a, beta;
  bool stop;
  const PartialsAhead<UpdShape<BIG>::SUMS_OF_UPDATE> sums = partials_request<UpdShape<BIG>::SUMS_OF_UPDATE>(part_rr, n_part), sums_rho = partials_request<UpdShape<BIG>::SUMS_OF_UPDATE>(part_rho, n_part);
  auto scalars = [&]() {
    const double rr = partials_finish(sums, part_rr, n_part, red);
    stop = rr <= tol2;
    if (!stop) rho_new = partials_finish(sums_rho, part_rho, n_part, red);
    a = *alpha;
    beta = stop ? 0.0 : rho_new / *rho;
  };
  if (n2 > 0) {
    dbl2 pv[UpdShape<BIG>::AHEAD_5V], xv[UpdShape<BIG>::AHEAD_5V], rv[UpdShape<BIG>::AHEAD_5V], rtv[UpdShape<BIG>::AHEAD_5V], ptv[UpdShape<BIG>::AHEAD_5V];
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD_5V; k++) {
      const int64_t j = min(i + k * stride, last);
      pv[k] = p2[j];
      xv[k] = x2[j];
      rv[k] = r2[j];
      rtv[k] = rt2[j];
      ptv[k] = pt2[j];
    }
    scalars();
    auto one = [&](int64_t j, dbl2 pp, dbl2 xx, dbl2 ra, dbl2 rta, dbl2 pta) {
      xx.x = fma(a, pp.x, xx.x);
      xx.y = fma(a, pp.y, xx.y);
      x2[j] = xx;
      if (!stop) {
        pp.x = fma(beta, pp.x, ra.x);
        pp.y = fma(beta, pp.y, ra.y);
        pta.x = fma(beta, pta.x, rta.x);
        pta.y = fma(beta, pta.y, rta.y);
        store_pair(p2 + j, pp, sys_scope);
        store_pair(pt2 + j, pta, sys_scope);
      }
    };
#pragma unroll
    for (int k = 0; k < UpdShape<BIG>::AHEAD_5V; k++)
      if (i + k * stride < n2) one(i + k * stride, pv[k], xv[k], rv[k], rtv[k], ptv[k]);
    for (int64_t j = i + UpdShape<BIG>::AHEAD_5V * stride; j < n2; j += stride) one(j, p2[j], x2[j], r2[j], rt2[j], pt2[j]);
  } else {
    scalars();
  }
  if (owns_tail(n)) {
    const double pn = p[n - 1];
    x[n - 1] = fma(a, pn, x[n - 1]);
    if (!stop) {
      store_one(p + (n - 1), fma(beta, pn, r[n - 1]), sys_scope);
      store_one(pt + (n - 1), fma(beta, pt[n - 1], rt[n - 1]), sys_scope);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (stop) *done = done_tag(iter); else { *rho_out = rho_new; *iters = iter; }
  }
}
#undef CASK_UPD_INDEX

// Row-sharded solvers: this rank's partial sums -> one scalar each (fixed order), ready for the all-reduce.
// One workgroup; up to two quantities per launch (BiCG's r.r and rt.r travel in one collective).
__global__ void k_sum_to_scalars(const double *__restrict__ pa, int na, double *out_a,
                                 const double *__restrict__ pb, int nb, double *out_b, const int *done) {
  __shared__ double red[16];
  if (done && *done) return;
  const double a = sum_partials(pa, na, red);
  if (threadIdx.x == 0) *out_a = a;
  if (pb) {
    const double b = sum_partials(pb, nb, red);
    if (threadIdx.x == 0) *out_b = b;
  }
}
#undef CASK_PAIR_LOOP

}  // namespace caskhip
