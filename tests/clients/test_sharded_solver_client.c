/* A row-sharded solve through the C ABI alone, the way INTEGRATION.md section 6 shows it -- plain C, no Python, no C++
 * host layer: device CSR (cask_hip.h), the engine's own RCCL collectives (cask_hip_rccl.h: communicator bootstrapped
 * from a unique id, here with the one rank a single GPU allows), cask_hip_solve_device with the all-reduce and the
 * operand all-gather as callbacks.  The system is a 2-D 5-point Laplacian (the reference's CG harness:
 * b = A x0, expect x0 back, test/test_utils.hpp:61-70); BiCG runs on the same matrix with its transpose block.
 *   test_sharded_solver_hip                 (exit status 0 and "Test passed!" on success) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "cask_hip.h"
#include "cask_hip_rccl.h"

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ != 0) {                                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cask_hip_last_error());            \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)
#define HIP(call)                                                                            \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                      \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

int main(void) {
  const int nx = 120, n = nx * nx;
  int *rp = (int *)malloc(sizeof(int) * (n + 1)), *ci = (int *)malloc(sizeof(int) * 5 * n);
  double *va = (double *)malloc(sizeof(double) * 5 * n), *x0 = (double *)malloc(sizeof(double) * n),
         *b = (double *)malloc(sizeof(double) * n), *x = (double *)malloc(sizeof(double) * n);
  int nnz = 0;
  rp[0] = 0;
  for (int i = 0; i < n; i++) {                       /* columns ascending within a row */
    const int gx = i % nx, gy = i / nx;
    if (gy > 0) { ci[nnz] = i - nx; va[nnz++] = -1.0; }
    if (gx > 0) { ci[nnz] = i - 1; va[nnz++] = -1.0; }
    ci[nnz] = i; va[nnz++] = 4.0 + 1e-3;
    if (gx < nx - 1) { ci[nnz] = i + 1; va[nnz++] = -1.0; }
    if (gy < nx - 1) { ci[nnz] = i + nx; va[nnz++] = -1.0; }
    rp[i + 1] = nnz;
    x0[i] = 0.25 * (i % 17);
  }
  cask_hip_matrix *A = NULL, *At = NULL;
  CHECK(cask_hip_csr_create(n, n, nnz, rp, ci, va, NULL, &A));
  CHECK(cask_hip_csr_create(n, n, nnz, rp, ci, va, NULL, &At));          /* symmetric: A^T = A, its own handle */
  CHECK(cask_hip_spmv(A, x0, b));

  /* the collectives: rank 0 creates the id, every rank the communicator (here: world = 1) */
  unsigned char id[CASK_HIP_RCCL_ID_BYTES];
  const int64_t bounds[2] = {0, n};
  cask_hip_comm *comm = NULL;
  CHECK(cask_hip_rccl_unique_id(id));
  CHECK(cask_hip_rccl_comm_create(id, 0, 1, bounds, &comm));

  double *d_b = NULL, *d_x = NULL;
  HIP(hipMalloc((void **)&d_b, sizeof(double) * n));
  HIP(hipMalloc((void **)&d_x, sizeof(double) * n));
  HIP(hipMemcpy(d_b, b, sizeof(double) * n, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  int status = 0;
  for (int kind = CASK_HIP_SOLVER_CG; kind <= CASK_HIP_SOLVER_BICG; kind++) {
    cask_hip_solver_config cfg = {0};
    cfg.kind = kind;
    cfg.allreduce = cask_hip_rccl_allreduce;          /* ncclAllReduce of the dot products, on `stream` */
    cfg.allreduce_user = comm;
    cfg.exchange = cask_hip_rccl_allgather;           /* blocks with global columns: ncclAllGather of the operand */
    cfg.exchange_user = comm;
    cfg.n_full = n;
    HIP(hipMemset(d_x, 0, sizeof(double) * n));
    int32_t iters = 0, conv = 0;
    double usec = 0.0;
    CHECK(cask_hip_solve_device(A, kind == CASK_HIP_SOLVER_BICG ? At : NULL, &cfg, d_b, d_x, 2000, 1e-10, &iters, &conv, &usec,
                                stream));
    HIP(hipMemcpy(x, d_x, sizeof(double) * n, hipMemcpyDeviceToHost));
    double err = 0.0;
    for (int i = 0; i < n; i++) err = fmax(err, fabs(x[i] - x0[i]));
    printf("%s: %d passes, converged %d, %.1f us per pass, max error %.3g\n", kind == CASK_HIP_SOLVER_CG ? "CG" : "BiCG",
           iters + 1, conv, usec, err);
    if (!conv || err > 1e-6) status = 1;
  }
  CHECK(cask_hip_rccl_comm_destroy(comm));
  CHECK(cask_hip_csr_destroy(A));
  CHECK(cask_hip_csr_destroy(At));
  puts(status ? "Test FAILED" : "Test passed!");
  return status;
}
