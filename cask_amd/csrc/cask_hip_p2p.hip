// Peer-to-peer exchange step of the sharded SpMV (include/cask_hip_p2p.h): shared x slices
// (hipIpc handles between the per-GPU processes) and the halo pull kernel.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "cask_hip.h"
#include "cask_hip_p2p.h"
#include "internal.hpp"

namespace {

#define P2P_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return caskhip::report_failure(CASK_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

static_assert(sizeof(hipIpcMemHandle_t) == CASK_HIP_SHARED_HANDLE_BYTES, "handle size is part of the ABI");

// A halo is a few hundred to a few hundred thousand scattered 8-byte entries; each lane issues
// its remote loads back to back (4 in flight) so one round trip over xGMI covers them.
__global__ void k_halo_pull(int64_t n, const uint64_t *__restrict__ src, double *__restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; j + 3 * stride < n; j += 4 * stride) {
    const double a = *reinterpret_cast<const double *>(src[j]);
    const double b = *reinterpret_cast<const double *>(src[j + stride]);
    const double c = *reinterpret_cast<const double *>(src[j + 2 * stride]);
    const double d = *reinterpret_cast<const double *>(src[j + 3 * stride]);
    dst[j] = a;
    dst[j + stride] = b;
    dst[j + 2 * stride] = c;
    dst[j + 3 * stride] = d;
  }
  for (; j < n; j += stride) dst[j] = *reinterpret_cast<const double *>(src[j]);
}

}  // namespace

extern "C" {

int cask_hip_shared_alloc(int64_t bytes, void **d_ptr_out, unsigned char *handle_out) {
  if (bytes <= 0 || !d_ptr_out || !handle_out) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  void *p = nullptr;
  P2P_TRY(hipMalloc(&p, (size_t)bytes));
  hipError_t e = hipMemset(p, 0, (size_t)bytes);
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return caskhip::report_failure(CASK_HIP_ERR_RUNTIME, std::string("shared allocation: ") + hipGetErrorString(e));
  }
  std::memcpy(handle_out, &h, sizeof(h));
  *d_ptr_out = p;
  return CASK_HIP_OK;
}

int cask_hip_shared_free(void *d_ptr) {
  if (!d_ptr) return CASK_HIP_OK;
  P2P_TRY(hipFree(d_ptr));
  return CASK_HIP_OK;
}

int cask_hip_shared_open(const unsigned char *handle, void **d_ptr_out) {
  if (!handle || !d_ptr_out) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void *p = nullptr;
  P2P_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *d_ptr_out = p;
  return CASK_HIP_OK;
}

int cask_hip_shared_close(void *d_ptr) {
  if (!d_ptr) return CASK_HIP_OK;
  P2P_TRY(hipIpcCloseMemHandle(d_ptr));
  return CASK_HIP_OK;
}

int cask_hip_copy_to_device(void *d_dst, const void *h_src, int64_t bytes) {
  if (bytes < 0 || (bytes > 0 && (!d_dst || !h_src))) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  if (bytes) P2P_TRY(hipMemcpy(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice));
  return CASK_HIP_OK;
}

int cask_hip_copy_to_host(void *h_dst, const void *d_src, int64_t bytes) {
  if (bytes < 0 || (bytes > 0 && (!h_dst || !d_src))) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  if (bytes) P2P_TRY(hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
  return CASK_HIP_OK;
}

int cask_hip_halo_pull_device(int64_t n_halo, const uint64_t *d_src_addr, double *d_dst, void *stream) {
  if (n_halo < 0 || (n_halo > 0 && (!d_src_addr || !d_dst))) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  if (n_halo == 0) return CASK_HIP_OK;
  const int wg = 256;
  const int64_t want = (n_halo + wg - 1) / wg;
  const int grid = (int)(want < 1024 ? want : 1024);
  hipLaunchKernelGGL(k_halo_pull, dim3(grid), dim3(wg), 0, static_cast<hipStream_t>(stream), n_halo, d_src_addr, d_dst);
  P2P_TRY(hipGetLastError());
  return CASK_HIP_OK;
}

}  // extern "C"
