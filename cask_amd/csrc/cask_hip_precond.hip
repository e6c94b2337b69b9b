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
#include <cstdlib>
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

// ---- synchronisation-free triangular solve ---------------------------------------------------------------------
// Level scheduling pays a barrier (or a launch) per dependency level: 57 436 levels on the G3_circuit-like factor in
// natural order = 245 ms per application (round 1).  Here ONE launch solves the whole triangle: waves take chunks of
// 64 consecutive rows (in dependency order: ascending for the lower, descending for the upper solve) from a global
// counter, one row per lane, and a lane walks its row entry by entry IN STORED ORDER -- the arithmetic of solve_row,
// bit for bit -- waiting for each x[c] it needs:
//  * c in the wave's own chunk: the producing lane's result comes by a cross-lane read (ds_bpermute) as soon as
//    that lane has finished -- a chain of consecutive rows costs a few dozen cycles per row, not a memory round trip;
//  * c in an earlier chunk: x[] is pre-filled with a NaN sentinel and every result is ONE 8-byte write-through store
//    (sc1), so "is it ready" and "what is it" are the same 8-byte sc1 load: the value is its own flag, nothing can be
//    observed half-done (MI355X_MICROARCH: 8-byte granules, hand-off without a separate flag).
// Progress: a wave only ever waits for rows of chunks handed out BEFORE its own, i.e. held by waves that are already
// running, whatever order the hardware dispatched workgroups in; the earliest unfinished chunk never waits for anybody.
// Every wait is bounded (TRSV_MAX_POLLS, far beyond any real wait): on overflow the kernel raises *err and finishes
// with the values it has, so a launch always drains.
constexpr unsigned long long TRSV_SENTINEL = 0x7FF8C0DECA5C0DE5ull;   // a quiet NaN no computation produces
constexpr int TRSV_MAX_POLLS = 1 << 22;

__global__ void k_fill_sentinel(int64_t n, double *x, int *counter) {
  const double sent = __longlong_as_double((long long)TRSV_SENTINEL);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = sent;
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter = 0;
}

__device__ __forceinline__ double shfl_f64(double v, int src_lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256)
k_trsv_syncfree(int n, int flags, const int *__restrict__ rp, const int *__restrict__ ci, const double *__restrict__ val,
                const double *__restrict__ b, double *x, int *counter, int *err) {
  const bool lower = flags & 1, unit = flags & 2;
  const int lane = threadIdx.x & 63;
  const int n_chunks = (n + 63) >> 6;
  while (true) {
    int chunk = 0;
    if (lane == 0) chunk = atomicAdd(counter, 1);
    chunk = __builtin_amdgcn_readfirstlane(chunk);
    if (chunk >= n_chunks) break;
    const int base = chunk << 6, i = base + lane;              // position in dependency order
    const bool active = i < n;
    const int r = active ? (lower ? i : n - 1 - i) : 0;
    int k = active ? rp[r] : 0;
    const int end = active ? rp[r + 1] : 0;
    double s = active ? b[r] : 0.0, diag = 0.0, xr = 0.0;
    int fin = active ? 0 : 1, polls = 0;
    // Two kinds of step alternate.  FAST steps (no global load in them): diagonal entries and entries whose producer
    // is a lane of this wave that has finished -- a run of consecutive dependent rows advances at cross-lane speed.
    // When no lane can take a fast step any more, ONE poll of x[c] for the lanes that wait on an earlier chunk.
    int c = 0, kind = 0, src = lane;                            // kind: 0 none, 1 diagonal/skip, 2 in-wave, 3 earlier chunk
    double v = 0.0;
    bool fetched = false;                                       // (c, v, kind, src) describe entry k
    while (__ballot(!fin) != 0ull) {                            // wave-uniform
      bool progress;
      do {
        progress = false;
        if (!fin && !fetched && k < end) {
          c = ci[k];
          v = val[k];
          if (c == r) kind = 1;
          else if (lower ? c > r : c < r) kind = 1;             // not in this triangle (extract_triangle leaves none)
          else {
            const int pos = lower ? c : n - 1 - c;              // the producer's position in dependency order (< i)
            if (pos >= base) { kind = 2; src = pos - base; }
            else kind = 3;
          }
          fetched = true;
        }
        const int want = (!fin && fetched && kind == 2) ? src : lane;
        const double x_in = shfl_f64(xr, want);                 // all lanes take part
        const int fin_in = __builtin_amdgcn_ds_bpermute(want << 2, fin);
        if (!fin && fetched) {
          if (kind == 1) {
            if (c == r) diag = v;
            k++; fetched = false; progress = true;
          } else if (kind == 2 && fin_in) {
            s -= v * x_in;
            k++; fetched = false; progress = true;
          }
        }
        if (!fin && !fetched && k == end) {
          xr = unit ? s : s / diag;
          __hip_atomic_store(x + r, xr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one 8-byte write-through store
          fin = 1;
          progress = true;
        }
      } while (__ballot(progress) != 0ull);
      if (!fin && fetched && kind == 3) {                       // one poll, then back to the fast steps
        const double xv = __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned long long)__double_as_longlong(xv) != TRSV_SENTINEL) { s -= v * xv; k++; fetched = false; }
        else if (++polls > TRSV_MAX_POLLS) { *err = 1; s -= v * xv; k++; fetched = false; }
        else __builtin_amdgcn_s_sleep(1);
      }
    }
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
  DevBuf<int> sync;                                   // [0] chunk counter, [1] error flag of the sync-free solve
  int sf_grid = 0;

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
    PC_TRY(sync.alloc(2));
    PC_TRY(hipMemset(sync.p, 0, 2 * sizeof(int)));
    int occ = 0, cus = 0, dev = 0;
    PC_TRY(hipGetDevice(&dev));
    PC_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_trsv_syncfree, 256, 0));
    PC_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int want = ((n + 63) / 64 + 3) / 4;
    sf_grid = std::max(1, std::min(want, std::max(1, occ) * cus));
    return CASK_HIP_OK;
  }

  // Level-scheduled by default; CASK_HIP_TRSV=syncfree selects the one-launch synchronisation-free solve.  Measured
  // on the G3_circuit-like ILU(0) factors (57 436 levels, profiles/r02_trsv.txt): levels 225 ms, sync-free 210-270 ms
  // per PCG pass -- no better.  The critical path is 57 K DEPENDENT rows either way, and a dependent step costs
  // ~2 us in both schedules: in the sync-free kernel every step of a lane re-reads its row entry from L1/L2 and the
  // wave's loop iteration is as slow as its slowest lane, which is usually one polling a remote x[c].  Getting the
  // in-wave chain down to cross-lane speed needs the row entries in registers and polls that do not block the wave
  // (loads issued a loop iteration ahead); until then the sync-free solve is a tested option, not the default.
  bool use_syncfree() const {
    static const char *force = std::getenv("CASK_HIP_TRSV");
    return force && std::string(force) == "syncfree";
  }

  int solve(const double *d_b, double *d_x, hipStream_t s) const {
    const int flags = (lower ? 1 : 0) | (unit ? 2 : 0);
    if (n > 0 && use_syncfree()) {
      const int fill_grid = (int)std::min<int64_t>(1024, ((int64_t)n + 255) / 256);
      hipLaunchKernelGGL(k_fill_sentinel, dim3(fill_grid), dim3(256), 0, s, (int64_t)n, d_x, sync.p);
      hipLaunchKernelGGL(k_trsv_syncfree, dim3(sf_grid), dim3(256), 0, s, n, flags, rp.p, ci.p, val.p, d_b, d_x, sync.p,
                         sync.p + 1);
      PC_TRY(hipGetLastError());
      return CASK_HIP_OK;
    }
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
  return cask_hip_precond_check(p);
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
  int sync_err = 0;
  PC_TRY(hipMemcpy(&sync_err, t.sync.p + 1, sizeof(int), hipMemcpyDeviceToHost));
  if (sync_err) return report_failure(CASK_HIP_ERR_RUNTIME, "triangular solve: a dependency never arrived (poll limit reached)");
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


// After a host synchronisation: did a sync-free triangular solve of this preconditioner give up on a dependency?
int cask_hip_precond_check(cask_hip_precond *p) {
  if (!p || p->kind == CASK_HIP_PRECOND_JACOBI) return CASK_HIP_OK;
  int e[2] = {0, 0};
  if (p->L.sync.p) PC_TRY(hipMemcpy(&e[0], p->L.sync.p + 1, sizeof(int), hipMemcpyDeviceToHost));
  if (p->U.sync.p) PC_TRY(hipMemcpy(&e[1], p->U.sync.p + 1, sizeof(int), hipMemcpyDeviceToHost));
  if (e[0] || e[1]) return report_failure(CASK_HIP_ERR_RUNTIME, "triangular solve: a dependency never arrived (poll limit reached)");
  return CASK_HIP_OK;
}
