// Shared between the translation units of libcask_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

namespace caskhip {
// records the thread-local message behind cask_hip_last_error() and returns `code`
int report_failure(int code, const std::string &msg);

// Owning device buffer.
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    return hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T));
  }
  hipError_t upload(const T *h, size_t count) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice);
  }
  hipError_t upload(const std::vector<T> &h) { return upload(h.data(), h.size()); }
  // a device copy of another buffer (an empty source leaves this one empty)
  hipError_t copy_from(const DevBuf<T> &src) {
    if (!src.p) {
      release();
      return hipSuccess;
    }
    hipError_t e = alloc(src.n);
    if (e != hipSuccess || src.n == 0) return e;
    return hipMemcpy(p, src.p, src.n * sizeof(T), hipMemcpyDeviceToDevice);
  }
};

// Owning event (the timing paths return early on errors).
struct DevEvent {
  hipEvent_t e = nullptr;
  DevEvent() = default;
  DevEvent(const DevEvent &) = delete;
  DevEvent &operator=(const DevEvent &) = delete;
  ~DevEvent() {
    if (e) (void)hipEventDestroy(e);
  }
  hipError_t create() { return hipEventCreate(&e); }
  operator hipEvent_t() const { return e; }
};

}  // namespace caskhip

// Between cask_hip.hip (the PCG driver) and cask_hip_precond.hip.
struct cask_hip_precond;
int cask_hip_precond_rows(const cask_hip_precond *p);      // order of the matrix the preconditioner was built from
// Jacobi: the device vector of 1/diag (the PCG driver folds the scaling into its update kernels); NULL otherwise
const double *cask_hip_precond_jacobi_scale(const cask_hip_precond *p);

// Content fingerprint of a CSR matrix: a wrapping 64-bit sum of one mixed word per row pointer and per nonzero
// (position, column, value bits), so that it can be computed in any order -- on the host over the arrays a
// preconditioner is built from, on the device over a handle's arrays -- and compared.  It answers "is this the matrix
// the multicolour preconditioner cached a permuted copy of?" (same pattern with other values must NOT match).
__host__ __device__ inline uint64_t csr_fp_mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t csr_fp_row(uint64_t r, uint64_t rp) { return csr_fp_mix((r << 32) ^ rp ^ 0xA5A5A5A500000000ull); }
__host__ __device__ inline uint64_t csr_fp_entry(uint64_t k, uint64_t col, uint64_t value_bits) {
  return csr_fp_mix(csr_fp_mix((k << 32) ^ col) ^ value_bits);
}

// Multicolour ILU(0): the colour-ordered PCG of cask_hip_pcg.  The view gives the driver the permutation (device), the
// permuted matrix P A P^T (host CSR, to build its product handle once; the handle is owned by the preconditioner) and the
// number of r.z shares a pass leaves; the sweeps are one application with the pass's x / r update and r.z shares fused.
struct cask_hip_matrix;
struct cask_hip_mc_view {
  int n, n_colors, n_part_rz;
  uint64_t fingerprint;               // of the CSR arrays the preconditioner was built from (csr_fp_*)
  const int *d_perm;
  const int *h_rp, *h_ci;
  const double *h_va;
  cask_hip_matrix **product;
};
struct cask_hip_mc_sweep_args {
  const double *rsold, *part_pAp;     // rsold != NULL: x += alpha p ; r -= alpha Ap with alpha = *rsold / sum(part_pAp[0..n_pAp))
  int n_pAp;
  const double *p, *Ap;
  double *x, *r;
  double *z;                          // out, n + 1 entries: [n] must be 0.0 (the slot padded factor entries read)
  double *part_rz;                    // out or NULL: the shares of r.z, cask_hip_mc_view::n_part_rz of them
  const int *done;
};
extern "C" int cask_hip_precond_mc_view(cask_hip_precond *p, cask_hip_mc_view *v);          // 0 unless p is a multicolour ILU(0)
extern "C" int cask_hip_precond_mc_sweeps(cask_hip_precond *p, const cask_hip_mc_sweep_args *a, void *stream);

// Between cask_hip.hip (the solver passes) and cask_hip_p2p.hip: "sum this rank's partial sums, then all-reduce" in one
// launch, when the all-reduce callback IS cask_hip_push_allreduce.
extern "C" int cask_hip_push_sum_allreduce(const double *d_pa, int na, const double *d_pb, int nb, double *d_out,
                                           const int *d_done, void *stream, void *push);
