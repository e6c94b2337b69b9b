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

// Between cask_hip.hip (the solver passes) and cask_hip_p2p.hip: "sum this rank's partial sums, then all-reduce" in one
// launch, when the all-reduce callback IS cask_hip_push_allreduce.
extern "C" int cask_hip_push_sum_allreduce(const double *d_pa, int na, const double *d_pb, int nb, double *d_out,
                                           const int *d_done, void *stream, void *push);
