// Development microbenchmark: what does ONE launch that streams a cant-sized
// CSR payload (8 B value + 4 B index per nonzero, 4.0 M nonzeros = 48 MB) cost
// on MI355X when the data is cold (rotating copies > 2x Infinity Cache)?
// Gives the practical ceiling the SpMV kernels are judged against.
//   hipcc --offload-arch=gfx950 -O3 -o build/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one-shot: each workgroup reads IPT elements per thread, all loads issued up front
template <int IPT, bool NT>
__global__ void k_oneshot(const double *__restrict__ val, const int *__restrict__ ci, long nnz, double *out) {
  const dbl2 *v2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *c2 = reinterpret_cast<const int2v *>(ci);
  const long pairs = nnz / 2;
  const long base = (long)blockIdx.x * blockDim.x * (IPT / 2);
  dbl2 v[IPT / 2]; int2v c[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    long p = base + u * blockDim.x + threadIdx.x;
    if (p >= pairs) p = pairs - 1;
    v[u] = NT ? __builtin_nontemporal_load(v2 + p) : v2[p];
    c[u] = NT ? __builtin_nontemporal_load(c2 + p) : c2[p];
  }
  double acc = 0;
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) acc += v[u].x * c[u].x + v[u].y * c[u].y;
  if (acc == 1.2345e-300) out[0] = acc;    // keep the loads alive
}

// persistent grid-stride with one tile of prefetch
template <int IPT, bool NT>
__global__ void k_persist(const double *__restrict__ val, const int *__restrict__ ci, long nnz, double *out) {
  const dbl2 *v2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *c2 = reinterpret_cast<const int2v *>(ci);
  const long pairs = nnz / 2;
  const long tile = (long)blockDim.x * (IPT / 2);
  const long ntiles = (pairs + tile - 1) / tile;
  double acc = 0;
  dbl2 v[IPT / 2]; int2v c[IPT / 2];
  long t = blockIdx.x;
  if (t >= ntiles) return;
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    long p = t * tile + u * blockDim.x + threadIdx.x;
    if (p >= pairs) p = pairs - 1;
    v[u] = NT ? __builtin_nontemporal_load(v2 + p) : v2[p];
    c[u] = NT ? __builtin_nontemporal_load(c2 + p) : c2[p];
  }
  for (;;) {
    const long tn = t + gridDim.x;
    dbl2 vn[IPT / 2]; int2v cn[IPT / 2];
    if (tn < ntiles) {
#pragma unroll
      for (int u = 0; u < IPT / 2; u++) {
        long p = tn * tile + u * blockDim.x + threadIdx.x;
        if (p >= pairs) p = pairs - 1;
        vn[u] = NT ? __builtin_nontemporal_load(v2 + p) : v2[p];
        cn[u] = NT ? __builtin_nontemporal_load(c2 + p) : c2[p];
      }
    }
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) acc += v[u].x * c[u].x + v[u].y * c[u].y;
    if (tn >= ntiles) break;
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) { v[u] = vn[u]; c[u] = cn[u]; }
    t = tn;
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

__global__ void k_empty() {}

int main(int argc, char **argv) {
  const long nnz = argc > 1 ? atol(argv[1]) : 4010891;
  const int copies = 13, steps = 100, reps = 5;
  std::vector<double *> vals(copies);
  std::vector<int *> cis(copies);
  for (int i = 0; i < copies; i++) {
    CK(hipMalloc((void **)&vals[i], nnz * 8 + 64));
    CK(hipMalloc((void **)&cis[i], nnz * 4 + 64));
    CK(hipMemset(vals[i], 0, nnz * 8));
    CK(hipMemset(cis[i], 0, nnz * 4));
  }
  double *out;
  CK(hipMalloc((void **)&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 12.0 * nnz;

  auto timeit = [&](const char *name, auto launch) {
    double best = 1e30;
    for (int r = 0; r < reps + 1; r++) {
      CK(hipEventRecord(e0, 0));
      for (int s = 0; s < steps; s++) launch(s % copies);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms * 1e3 / steps < best) best = ms * 1e3 / steps;
    }
    printf("%-44s %8.3f us/launch  %8.1f GB/s\n", name, best, bytes / best * 1e-3);
  };

  timeit("empty kernel (launch floor)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0); });

#define ONESHOT(IPT, WG, NT) { char nm[96]; snprintf(nm, 96, "oneshot IPT=%d WG=%d NT=%d", IPT, WG, NT); \
    long tile = (long)WG * (IPT / 2); int grid = (int)((nnz / 2 + tile - 1) / tile); \
    timeit(nm, [&](int c) { hipLaunchKernelGGL((k_oneshot<IPT, NT>), dim3(grid), dim3(WG), 0, 0, vals[c], cis[c], nnz, out); }); }
#define PERSIST(IPT, WG, NT, GRID) { char nm[96]; snprintf(nm, 96, "persist IPT=%d WG=%d NT=%d grid=%d", IPT, WG, NT, GRID); \
    timeit(nm, [&](int c) { hipLaunchKernelGGL((k_persist<IPT, NT>), dim3(GRID), dim3(WG), 0, 0, vals[c], cis[c], nnz, out); }); }

  ONESHOT(4, 256, true) ONESHOT(8, 256, true) ONESHOT(16, 256, true) ONESHOT(8, 512, true) ONESHOT(4, 512, true)
  ONESHOT(8, 256, false) ONESHOT(4, 1024, true) ONESHOT(2, 256, true) ONESHOT(8, 1024, true)
  PERSIST(4, 256, true, 2048) PERSIST(8, 256, true, 2048) PERSIST(8, 256, true, 1024) PERSIST(4, 256, true, 4096)
  PERSIST(8, 512, true, 1024) PERSIST(4, 512, true, 1024) PERSIST(16, 256, true, 1024) PERSIST(8, 256, false, 2048)
  PERSIST(4, 1024, true, 512) PERSIST(8, 1024, true, 512) PERSIST(8, 256, true, 512) PERSIST(16, 256, true, 512)
  PERSIST(16, 512, true, 512) PERSIST(16, 512, true, 256) PERSIST(16, 1024, true, 256)
  return 0;
}
