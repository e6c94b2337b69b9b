// Preconditioners of the CASK solver surface on the GPU (include/cask_hip.h, "preconditioning"):
// Jacobi and ILU(0) with level-scheduled triangular solves, and the stand-alone triangular solve
// behind cask::mkl::unittrsolve.
//
// Reference: ILUPreconditioner (src/runtime/SparseLinearSolvers.hpp:77-156) -- an IKJ incomplete
// factorisation on the pattern of the matrix, L and U extracted WITH the diagonal
// (DokMatrix::getLowerTriangular/getUpperTriangular, SparseMatrix.hpp:227-253) and applied with two
// mkl_dcsrtrsv calls that divide by the stored diagonal (MklLayer.hpp:29-85, diag = 'N') -- so the
// "L" solve divides by U's diagonal as well; that is what the known answers of
// test/LinearSolvers.cpp:116-146 pin and what is reproduced here.
//
// The factorisation is a setup step and runs on the host (as in the reference: its constructor is
// sequential CPU code); every application runs on the device.  A triangular solve is a chain of
// dependency levels: rows of one level are independent.  Wide levels get a launch of their own, runs
// of narrow levels are walked by ONE workgroup with a barrier between levels (a stencil matrix in
// natural order has thousands of levels of a few hundred rows: a launch per level would cost 2 us
// each, a barrier costs 0.2).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cask_hip.h"
#include "internal.hpp"

using caskhip::DevBuf;
using caskhip::report_failure;

namespace {

#define PC_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return report_failure(CASK_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int TRSV_WG = 1024;          // the workgroup that walks runs of narrow levels
constexpr int WIDE_LEVEL = 16384;      // rows from which a level gets a launch of its own

// x[r] = (b[r] - sum_{c != r, c in the triangle} val * x[c]) / val[r][r], entries in stored order
// flags: bit 0 = lower triangle, bit 1 = unit diagonal (the stored diagonal is ignored)
__device__ __forceinline__ void solve_row(int r, int flags, const int *__restrict__ rp, const int *__restrict__ ci,
                                          const double *__restrict__ val, const double *__restrict__ b, double *x) {
  const bool lower = flags & 1, unit = flags & 2;
  double s = b[r], diag = 0.0;
  for (int k = rp[r]; k < rp[r + 1]; k++) {
    const int c = ci[k];
    if (c == r) diag = val[k];
    else if (lower ? c < r : c > r) s -= val[k] * x[c];
  }
  x[r] = unit ? s : s / diag;
}

// one wide level: a thread per row
__global__ void k_trsv_level(int lo, int hi, int lower, const int *__restrict__ order, const int *__restrict__ rp,
                             const int *__restrict__ ci, const double *__restrict__ val,
                             const double *__restrict__ b, double *x) {
  const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hi) solve_row(order[i], lower, rp, ci, val, b, x);
}

// levels [l0, l1) by one workgroup: the barrier orders a level's stores before the next level's loads
// (all waves of a workgroup share the CU's L1)
__global__ void k_trsv_levels(int l0, int l1, int lower, const int *__restrict__ level_ptr,
                              const int *__restrict__ order, const int *__restrict__ rp, const int *__restrict__ ci,
                              const double *__restrict__ val, const double *__restrict__ b, double *x) {
  for (int l = l0; l < l1; l++) {
    const int lo = level_ptr[l], hi = level_ptr[l + 1];
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) solve_row(order[i], lower, rp, ci, val, b, x);
    __syncthreads();
  }
}

__global__ void k_scale(int64_t n, const double *__restrict__ dinv, const double *__restrict__ r, double *__restrict__ z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    z[i] = r[i] * dinv[i];
}

// Jacobi inside PCG: z = r / diag and the workgroup's share of r.z in one pass (16-byte accesses; the vectors
// are the solver's own aligned allocations).  Fixed grid and butterfly order => reproducible.
__global__ void k_scale_dot(int64_t n, const double *__restrict__ dinv, const double *__restrict__ r,
                            double *__restrict__ z, double *__restrict__ partials, const int *done) {
  __shared__ double red[16];
  if (done && *done) return;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  const dbl2 *d2 = reinterpret_cast<const dbl2 *>(dinv), *r2 = reinterpret_cast<const dbl2 *>(r);
  dbl2 *z2 = reinterpret_cast<dbl2 *>(z);
  double a0 = 0.0, a1 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 1); i += (int64_t)gridDim.x * blockDim.x) {
    const dbl2 rv = r2[i], dv = d2[i];
    dbl2 zv;
    zv.x = rv.x * dv.x;
    zv.y = rv.y * dv.y;
    z2[i] = zv;
    a0 = fma(rv.x, zv.x, a0);
    a1 = fma(rv.y, zv.y, a1);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double zl = r[n - 1] * dinv[n - 1];
    z[n - 1] = zl;
    a0 = fma(r[n - 1], zl, a0);
  }
  double v = a0 + a1;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w];
    partials[blockIdx.x] = s;
  }
}

// One triangular factor (diagonal included) on the device with its level schedule.
struct TriFactor {
  int n = 0;
  bool lower = true, unit = false;
  DevBuf<int> rp, ci, order, level_ptr;
  DevBuf<double> val;
  int n_levels = 0;
  struct Step { int l0, l1, lo, hi; bool wide; };     // levels [l0,l1) = positions [lo,hi) of `order`
  std::vector<Step> steps;

  int build(int n_, bool lower_, const std::vector<int> &h_rp, const std::vector<int> &h_ci,
            const std::vector<double> &h_val) {
    n = n_;
    lower = lower_;
    // level of a row = 1 + max level of the rows it depends on
    std::vector<int> level(n, 0);
    n_levels = n > 0 ? 1 : 0;
    auto visit = [&](int r) {
      int lv = 0;
      for (int k = h_rp[r]; k < h_rp[r + 1]; k++) {
        const int c = h_ci[k];
        if (lower ? c < r : c > r) lv = std::max(lv, level[c] + 1);
      }
      level[r] = lv;
      n_levels = std::max(n_levels, lv + 1);
    };
    if (lower) for (int r = 0; r < n; r++) visit(r);
    else       for (int r = n - 1; r >= 0; r--) visit(r);
    std::vector<int> lp(n_levels + 1, 0), ord(n);
    for (int r = 0; r < n; r++) lp[level[r] + 1]++;
    for (int l = 0; l < n_levels; l++) lp[l + 1] += lp[l];
    std::vector<int> fill(lp.begin(), lp.end() - 1);
    for (int r = 0; r < n; r++) ord[fill[level[r]]++] = r;
    steps.clear();
    for (int l = 0; l < n_levels;) {
      if (lp[l + 1] - lp[l] >= WIDE_LEVEL) {
        steps.push_back(Step{l, l + 1, lp[l], lp[l + 1], true});
        l++;
        continue;
      }
      int e = l;
      while (e < n_levels && lp[e + 1] - lp[e] < WIDE_LEVEL) e++;
      steps.push_back(Step{l, e, lp[l], lp[e], false});
      l = e;
    }
    PC_TRY(rp.upload(h_rp));
    PC_TRY(ci.upload(h_ci));
    PC_TRY(val.upload(h_val));
    PC_TRY(order.upload(ord));
    PC_TRY(level_ptr.upload(lp));
    return CASK_HIP_OK;
  }

  int solve(const double *d_b, double *d_x, hipStream_t s) const {
    const int flags = (lower ? 1 : 0) | (unit ? 2 : 0);
    for (const Step &st : steps) {
      if (st.wide)
        hipLaunchKernelGGL(k_trsv_level, dim3((st.hi - st.lo + 255) / 256), dim3(256), 0, s, st.lo, st.hi, flags,
                           order.p, rp.p, ci.p, val.p, d_b, d_x);
      else
        hipLaunchKernelGGL(k_trsv_levels, dim3(1), dim3(TRSV_WG), 0, s, st.l0, st.l1, flags, level_ptr.p, order.p,
                           rp.p, ci.p, val.p, d_b, d_x);
    }
    PC_TRY(hipGetLastError());
    return CASK_HIP_OK;
  }
};

// rows [keep entries with (lower ? c <= r : c >= r)] of a CSR matrix
void extract_triangle(int n, const int *rp, const int *ci, const double *val, bool lower, std::vector<int> &trp,
                      std::vector<int> &tci, std::vector<double> &tval) {
  trp.assign((size_t)n + 1, 0);
  tci.clear();
  tval.clear();
  for (int r = 0; r < n; r++) {
    for (int k = rp[r]; k < rp[r + 1]; k++)
      if (lower ? ci[k] <= r : ci[k] >= r) {
        tci.push_back(ci[k]);
        tval.push_back(val[k]);
      }
    trp[r + 1] = (int)tci.size();
  }
}

int check_sorted_csr(int32_t n, int64_t nnz, const int32_t *rp, const int32_t *ci) {
  if (n < 0 || nnz < 0 || !rp || (nnz > 0 && !ci)) return report_failure(CASK_HIP_ERR_INVALID, "bad CSR arguments");
  if (rp[0] != 0 || rp[n] != nnz) return report_failure(CASK_HIP_ERR_INVALID, "row_ptr does not span the nonzeros");
  for (int r = 0; r < n; r++) {
    if (rp[r + 1] < rp[r]) return report_failure(CASK_HIP_ERR_INVALID, "row_ptr must be non-decreasing");
    for (int k = rp[r]; k < rp[r + 1]; k++) {
      if (ci[k] < 0 || ci[k] >= n) return report_failure(CASK_HIP_ERR_INVALID, "column index out of range");
      if (k > rp[r] && ci[k] <= ci[k - 1])
        return report_failure(CASK_HIP_ERR_INVALID, "columns must be strictly ascending within a row");
    }
  }
  return CASK_HIP_OK;
}

}  // namespace

struct cask_hip_precond {
  int kind = 0, n = 0;
  int device = 0;
  std::vector<double> factored;        // ILU0: the factored values in the input pattern (pc of the reference)
  TriFactor L, U;
  DevBuf<double> dinv, tmp, d_r, d_z;  // Jacobi: 1/diag ; ILU0: the intermediate vector ; staging for host vectors
};

extern "C" {

int cask_hip_precond_create(int32_t kind, int32_t n, int64_t nnz, const int32_t *row_ptr, const int32_t *col_ind,
                            const double *values, cask_hip_precond **out) {
  if (!out) return report_failure(CASK_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (kind != CASK_HIP_PRECOND_JACOBI && kind != CASK_HIP_PRECOND_ILU0 && kind != CASK_HIP_PRECOND_ILU0_UNIT)
    return report_failure(CASK_HIP_ERR_INVALID, "unknown preconditioner kind");
  int rc = check_sorted_csr(n, nnz, row_ptr, col_ind);
  if (rc) return rc;
  if (nnz > 0 && !values) return report_failure(CASK_HIP_ERR_INVALID, "values is NULL");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return report_failure(CASK_HIP_ERR_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
  std::unique_ptr<cask_hip_precond> p(new cask_hip_precond);
  p->kind = kind;
  p->n = n;
  PC_TRY(hipGetDevice(&p->device));
  if (kind == CASK_HIP_PRECOND_JACOBI) {
    std::vector<double> dinv((size_t)n, 1.0);               // a row without a diagonal entry is left unscaled
    for (int r = 0; r < n; r++)
      for (int k = row_ptr[r]; k < row_ptr[r + 1]; k++)
        if (col_ind[k] == r && values[k] != 0.0) dinv[r] = 1.0 / values[k];
    PC_TRY(p->dinv.upload(dinv));
    *out = p.release();
    return CASK_HIP_OK;
  }
  // ILU(0), the IKJ sweep of ILUPreconditioner (SparseLinearSolvers.hpp:92-113): for every row i and
  // every stored (i,k) with k < i in ascending k: if (k,k) is stored, a[i][k] /= a[k][k], and every
  // stored (i,j), j > k, with (k,j) stored gets a[i][j] -= a[k][j] * a[i][k].
  std::vector<double> &a = p->factored;
  a.assign(values, values + nnz);
  std::vector<int> diag((size_t)n, -1);
  for (int r = 0; r < n; r++)
    for (int k = row_ptr[r]; k < row_ptr[r + 1]; k++)
      if (col_ind[k] == r) diag[r] = k;
  for (int i = 1; i < n; i++) {
    for (int kk = row_ptr[i]; kk < row_ptr[i + 1]; kk++) {
      const int k = col_ind[kk];
      if (k >= i) break;
      if (diag[k] < 0) continue;
      a[kk] = a[kk] / a[diag[k]];
      const double beta = a[kk];
      int pi = kk + 1, pk = diag[k] + 1;                     // (i, j > k) against (k, j > k), both ascending
      const int ei = row_ptr[i + 1], ek = row_ptr[k + 1];
      while (pi < ei && pk < ek) {
        if (col_ind[pi] < col_ind[pk]) pi++;
        else if (col_ind[pi] > col_ind[pk]) pk++;
        else {
          a[pi] = a[pi] - a[pk] * beta;
          pi++;
          pk++;
        }
      }
    }
  }
  std::vector<int> trp, tci;
  std::vector<double> tval;
  extract_triangle(n, row_ptr, col_ind, a.data(), true, trp, tci, tval);
  p->L.unit = kind == CASK_HIP_PRECOND_ILU0_UNIT;            // the textbook L has a unit diagonal; the reference's does not
  rc = p->L.build(n, true, trp, tci, tval);
  if (rc) return rc;
  extract_triangle(n, row_ptr, col_ind, a.data(), false, trp, tci, tval);
  rc = p->U.build(n, false, trp, tci, tval);
  if (rc) return rc;
  PC_TRY(p->tmp.alloc((size_t)n));
  *out = p.release();
  return CASK_HIP_OK;
}

int cask_hip_precond_destroy(cask_hip_precond *p) {
  delete p;
  return CASK_HIP_OK;
}

int cask_hip_precond_factor_values(const cask_hip_precond *p, double *values_out) {
  if (!p || !values_out) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (p->kind == CASK_HIP_PRECOND_JACOBI) return report_failure(CASK_HIP_ERR_INVALID, "only ILU0 keeps factor values");
  if (!p->factored.empty()) std::memcpy(values_out, p->factored.data(), p->factored.size() * sizeof(double));
  return CASK_HIP_OK;
}

int cask_hip_precond_info(const cask_hip_precond *p, int32_t *levels_lower, int32_t *levels_upper,
                          int32_t *launches_per_apply) {
  if (!p) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  const bool ilu = p->kind != CASK_HIP_PRECOND_JACOBI;
  if (levels_lower) *levels_lower = ilu ? p->L.n_levels : 0;
  if (levels_upper) *levels_upper = ilu ? p->U.n_levels : 0;
  if (launches_per_apply) *launches_per_apply = ilu ? (int32_t)(p->L.steps.size() + p->U.steps.size()) : 1;
  return CASK_HIP_OK;
}

int cask_hip_precond_apply_device(cask_hip_precond *p, const double *d_r, double *d_z, void *stream) {
  if (!p || (p->n > 0 && (!d_r || !d_z))) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (p->n == 0) return CASK_HIP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p->kind == CASK_HIP_PRECOND_JACOBI) {
    const int grid = (int)std::min<int64_t>(1024, ((int64_t)p->n + 255) / 256);
    hipLaunchKernelGGL(k_scale, dim3(grid), dim3(256), 0, s, (int64_t)p->n, p->dinv.p, d_r, d_z);
    PC_TRY(hipGetLastError());
    return CASK_HIP_OK;
  }
  // z = U^-1 (L^-1 r)   (ILUPreconditioner::apply, SparseLinearSolvers.hpp:143-151)
  int rc = p->L.solve(d_r, p->tmp.p, s);
  if (rc) return rc;
  return p->U.solve(p->tmp.p, d_z, s);
}

int cask_hip_precond_apply(cask_hip_precond *p, const double *r, double *z) {
  if (!p || (p->n > 0 && (!r || !z))) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (p->n == 0) return CASK_HIP_OK;
  PC_TRY(hipSetDevice(p->device));
  if (!p->d_r.p) PC_TRY(p->d_r.alloc((size_t)p->n));
  if (!p->d_z.p) PC_TRY(p->d_z.alloc((size_t)p->n));
  PC_TRY(hipMemcpy(p->d_r.p, r, (size_t)p->n * sizeof(double), hipMemcpyHostToDevice));
  int rc = cask_hip_precond_apply_device(p, p->d_r.p, p->d_z.p, nullptr);
  if (rc) return rc;
  PC_TRY(hipMemcpy(z, p->d_z.p, (size_t)p->n * sizeof(double), hipMemcpyDeviceToHost));
  return CASK_HIP_OK;
}

int cask_hip_trsolve(int32_t n, int64_t nnz, const int32_t *row_ptr, const int32_t *col_ind, const double *values,
                     int32_t lower, const double *rhs, double *x) {
  int rc = check_sorted_csr(n, nnz, row_ptr, col_ind);
  if (rc) return rc;
  if ((nnz > 0 && !values) || (n > 0 && (!rhs || !x))) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (n == 0) return CASK_HIP_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return report_failure(CASK_HIP_ERR_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
  std::vector<int> trp, tci;
  std::vector<double> tval;
  extract_triangle(n, row_ptr, col_ind, values, lower != 0, trp, tci, tval);
  TriFactor t;
  rc = t.build(n, lower != 0, trp, tci, tval);
  if (rc) return rc;
  DevBuf<double> b, out;
  PC_TRY(b.upload(rhs, (size_t)n));
  PC_TRY(out.alloc((size_t)n));
  rc = t.solve(b.p, out.p, nullptr);
  if (rc) return rc;
  PC_TRY(hipMemcpy(x, out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return CASK_HIP_OK;
}

}  // extern "C"

int cask_hip_precond_rows(const cask_hip_precond *p) { return p ? p->n : -1; }

int cask_hip_precond_apply_dot(cask_hip_precond *p, const double *d_r, double *d_z, double *d_partials,
                               int max_partials, int *n_partials, const int *d_done, hipStream_t stream) {
  if (!p || p->kind != CASK_HIP_PRECOND_JACOBI || p->n == 0) return 0;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(max_partials, ((int64_t)p->n / 2 + 255) / 256));
  hipLaunchKernelGGL(k_scale_dot, dim3(grid), dim3(256), 0, stream, (int64_t)p->n, p->dinv.p, d_r, d_z, d_partials,
                     d_done);
  if (hipGetLastError() != hipSuccess) return -1;
  *n_partials = grid;
  return 1;
}

