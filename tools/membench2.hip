// Development microbenchmark 2: starting from the pure 12-byte/nnz stream,
// add the SpMV ingredients one at a time (x gather, LDS products, row reduce)
// to see what each costs on a cant-shaped payload (rows of exactly 64 nnz,
// columns random within +-400 of the diagonal), cold (13 rotating copies).
//   hipcc --offload-arch=gfx950 -O3 -o build/membench2 tools/membench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// STAGE 0: stream only. 1: + gather x, register accumulate. 2: + products to LDS and read back own.
// 3: + per-row sums (rows of 64 nnz => with IPT=8, WG=256: 2048 nnz = 32 rows, 8 lanes per row) and y store.
template <int IPT, int STAGE>
__global__ void k_stage(const double *__restrict__ val, const int *__restrict__ ci, const double *__restrict__ x,
                        double *__restrict__ y, long nnz, double *out) {
  extern __shared__ double prod[];
  const dbl2 *v2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *c2 = reinterpret_cast<const int2v *>(ci);
  const long pairs = nnz / 2;
  const int WG = blockDim.x, tid = threadIdx.x;
  const long base = (long)blockIdx.x * WG * (IPT / 2);
  dbl2 v[IPT / 2]; int2v c[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    long p = base + u * WG + tid;
    if (p >= pairs) p = pairs - 1;
    v[u] = __builtin_nontemporal_load(v2 + p);
    c[u] = __builtin_nontemporal_load(c2 + p);
  }
  double acc = 0;
  if (STAGE == 0) {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) acc += v[u].x * c[u].x + v[u].y * c[u].y;
  } else {
    dbl2 xv[IPT / 2];
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) { xv[u].x = x[c[u].x]; xv[u].y = x[c[u].y]; }
    if (STAGE == 1) {
#pragma unroll
      for (int u = 0; u < IPT / 2; u++) acc += v[u].x * xv[u].x + v[u].y * xv[u].y;
    } else {
      dbl2 *p2 = reinterpret_cast<dbl2 *>(prod);
#pragma unroll
      for (int u = 0; u < IPT / 2; u++) p2[u * WG + tid] = v[u] * xv[u];
      __syncthreads();
      if (STAGE == 2) {
#pragma unroll
        for (int u = 0; u < IPT; u++) acc += prod[u * WG + tid];
      } else {
        // rows of 64 products; rows per block = WG*IPT/64; lanes per row G = 64*... = WG / rows = 64/IPT
        constexpr int G = 64 / IPT;
        const int r = tid / G, j = tid % G;
        double a = 0;
#pragma unroll
        for (int k = 0; k < 64 / G; k++) a += prod[r * 64 + j + k * G];
        if (G >= 2) a += dpp<0xB1>(a);
        if (G >= 4) a += dpp<0x4E>(a);
        if (G >= 8) a += dpp<0x141>(a);
        if (G >= 16) a += dpp<0x140>(a);
        if (j == 0) y[(long)blockIdx.x * (WG / G) + r] = a;
      }
    }
  }
  if (STAGE != 3 && acc == 1.2345e-300) out[0] = acc;
}

// STAGE 4: like 3, but x comes from an LDS window [r0-400, r0+rows+400) that is requested BEFORE the
// stream (vmcnt retires in order, so waiting for the window does not wait for the stream).
template <int IPT, int XU>
__global__ void k_stage4(const double *__restrict__ val, const int *__restrict__ ci, const double *__restrict__ x,
                         double *__restrict__ y, long nnz, int n, double *out) {
  extern __shared__ double smem[];
  const int WG = blockDim.x, tid = threadIdx.x;
  double *prod = smem;                       // WG*IPT
  double *xs = smem + WG * IPT;              // XU*WG*2
  const dbl2 *v2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *c2 = reinterpret_cast<const int2v *>(ci);
  const long pairs = nnz / 2;
  const long base = (long)blockIdx.x * WG * (IPT / 2);
  const int rows = WG * IPT / 64;
  const int r0 = blockIdx.x * rows;
  const int w0 = max(0, r0 - 400) & ~1;
  const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x + w0);
  const int maxp = (n - w0) / 2 - 1;
  dbl2 xw[XU];
#pragma unroll
  for (int u = 0; u < XU; u++) xw[u] = x2[min(u * WG + tid, maxp)];
  dbl2 v[IPT / 2]; int2v c[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    long p = base + u * WG + tid;
    if (p >= pairs) p = pairs - 1;
    v[u] = __builtin_nontemporal_load(v2 + p);
    c[u] = __builtin_nontemporal_load(c2 + p);
  }
  dbl2 *xs2 = reinterpret_cast<dbl2 *>(xs);
#pragma unroll
  for (int u = 0; u < XU; u++) xs2[u * WG + tid] = xw[u];
  __syncthreads();
  dbl2 xv[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) { xv[u].x = xs[c[u].x - w0]; xv[u].y = xs[c[u].y - w0]; }
  dbl2 *p2 = reinterpret_cast<dbl2 *>(prod);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) p2[u * WG + tid] = v[u] * xv[u];
  __syncthreads();
  constexpr int G = 64 / IPT;
  const int r = tid / G, j = tid % G;
  double a = 0;
#pragma unroll
  for (int k = 0; k < 64 / G; k++) a += prod[r * 64 + j + k * G];
  if (G >= 2) a += dpp<0xB1>(a);
  if (G >= 4) a += dpp<0x4E>(a);
  if (G >= 8) a += dpp<0x141>(a);
  if (G >= 16) a += dpp<0x140>(a);
  if (j == 0) y[(long)blockIdx.x * (WG / G) + r] = a;
}

// STAGE 5: like 4 (x window requested first, gathered from LDS), but every WAVE owns a contiguous run of
// 64*IPT nonzeros (= IPT whole rows) and reduces its own rows without a second workgroup barrier.
template <int IPT, int XU>
__global__ void k_stage5(const double *__restrict__ val, const int *__restrict__ ci, const double *__restrict__ x,
                         double *__restrict__ y, long nnz, int n, double *out) {
  extern __shared__ double smem[];
  const int WG = blockDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *prod = smem + wave * (64 * IPT);   // per wave
  double *xs = smem + WG * IPT;              // XU*WG*2 shared
  const dbl2 *v2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *c2 = reinterpret_cast<const int2v *>(ci);
  const long pairs = nnz / 2;
  const long base = (long)blockIdx.x * WG * (IPT / 2) + (long)wave * 64 * (IPT / 2);
  const int rows = WG * IPT / 64;
  const int r0 = blockIdx.x * rows;
  const int w0 = max(0, r0 - 400) & ~1;
  const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x + w0);
  const int maxp = (n - w0) / 2 - 1;
  dbl2 xw[XU];
#pragma unroll
  for (int u = 0; u < XU; u++) xw[u] = x2[min(u * WG + tid, maxp)];
  dbl2 v[IPT / 2]; int2v c[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    long p = base + u * 64 + lane;
    if (p >= pairs) p = pairs - 1;
    v[u] = __builtin_nontemporal_load(v2 + p);
    c[u] = __builtin_nontemporal_load(c2 + p);
  }
  dbl2 *xs2 = reinterpret_cast<dbl2 *>(xs);
#pragma unroll
  for (int u = 0; u < XU; u++) xs2[u * WG + tid] = xw[u];
  __syncthreads();
  dbl2 *p2 = reinterpret_cast<dbl2 *>(prod);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    dbl2 xv; xv.x = xs[c[u].x - w0]; xv.y = xs[c[u].y - w0];
    p2[u * 64 + lane] = v[u] * xv;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  constexpr int G = 64 / IPT;                // lanes per row: the wave's IPT rows in one pass
  const int r = lane / G, j = lane % G;
  double a = 0;
#pragma unroll
  for (int k = 0; k < 64 / G; k++) a += prod[r * 64 + j + k * G];
  if (G >= 2) a += dpp<0xB1>(a);
  if (G >= 4) a += dpp<0x4E>(a);
  if (G >= 8) a += dpp<0x141>(a);
  if (G >= 16) a += dpp<0x140>(a);
  if (j == 0) y[(long)blockIdx.x * rows + wave * IPT + r] = a;
}

__global__ void k_empty() {}

int main() {
  const long n = 62451, per = 64, nnz = n * per;
  const int copies = 13, steps = 100, reps = 5;
  std::vector<double> hv(nnz); std::vector<int> hc(nnz);
  std::mt19937 rng(1);
  for (long r = 0; r < n; r++)
    for (int k = 0; k < per; k++) {
      long c = r + (long)(rng() % 801) - 400;
      hc[r * per + k] = (int)(c < 0 ? 0 : (c >= n ? n - 1 : c));
      hv[r * per + k] = 1.0 + (rng() % 100) * 0.01;
    }
  std::vector<double *> vals(copies); std::vector<int *> cis(copies);
  for (int i = 0; i < copies; i++) {
    CK(hipMalloc((void **)&vals[i], nnz * 8 + 64)); CK(hipMalloc((void **)&cis[i], nnz * 4 + 64));
    CK(hipMemcpy(vals[i], hv.data(), nnz * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(cis[i], hc.data(), nnz * 4, hipMemcpyHostToDevice));
  }
  double *x, *y, *out;
  CK(hipMalloc((void **)&x, n * 8)); CK(hipMalloc((void **)&y, n * 8 + 4096)); CK(hipMalloc((void **)&out, 64));
  std::vector<double> hx(n, 1.0);
  CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 12.0 * nnz + 16.0 * n;
  auto timeit = [&](const char *name, auto launch) {
    double best = 1e30;
    for (int r = 0; r < reps + 1; r++) {
      CK(hipEventRecord(e0, 0));
      for (int s = 0; s < steps; s++) launch(s % copies);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms * 1e3 / steps < best) best = ms * 1e3 / steps;
    }
    printf("%-40s %8.3f us/launch  %8.1f GB/s\n", name, best, bytes / best * 1e-3);
  };
#define RUN(IPT, WG, STAGE) { char nm[96]; snprintf(nm, 96, "stage %d IPT=%d WG=%d", STAGE, IPT, WG); \
    long tile = (long)WG * (IPT / 2); int grid = (int)((nnz / 2 + tile - 1) / tile); \
    timeit(nm, [&](int c) { hipLaunchKernelGGL((k_stage<IPT, STAGE>), dim3(grid), dim3(WG), WG * IPT * 8, 0, vals[c], cis[c], x, y, nnz, out); }); }
  RUN(8, 256, 0) RUN(8, 256, 1) RUN(8, 256, 2) RUN(8, 256, 3)
  RUN(4, 256, 0) RUN(4, 256, 1) RUN(4, 256, 2) RUN(4, 256, 3)
  RUN(4, 512, 0) RUN(4, 512, 1) RUN(4, 512, 2) RUN(4, 512, 3)
  RUN(16, 256, 0) RUN(16, 256, 1) RUN(16, 256, 2) RUN(16, 256, 3)
  RUN(8, 128, 1) RUN(8, 128, 3) RUN(8, 64, 1) RUN(8, 64, 3)
#define RUN4(IPT, WG, XU) { char nm[96]; snprintf(nm, 96, "stage 4 (LDS x) IPT=%d WG=%d XU=%d", IPT, WG, XU); \
    long tile = (long)WG * (IPT / 2); int grid = (int)((nnz / 2 + tile - 1) / tile); \
    timeit(nm, [&](int c) { hipLaunchKernelGGL((k_stage4<IPT, XU>), dim3(grid), dim3(WG), (WG * IPT + XU * WG * 2) * 8, 0, vals[c], cis[c], x, y, nnz, (int)n, out); }); }
#define RUN5(IPT, WG, XU) { char nm[96]; snprintf(nm, 96, "stage 5 (wave rows) IPT=%d WG=%d XU=%d", IPT, WG, XU); \
    long tile = (long)WG * (IPT / 2); int grid = (int)((nnz / 2 + tile - 1) / tile); \
    timeit(nm, [&](int c) { hipLaunchKernelGGL((k_stage5<IPT, XU>), dim3(grid), dim3(WG), (WG * IPT + XU * WG * 2) * 8, 0, vals[c], cis[c], x, y, nnz, (int)n, out); }); }
  RUN5(8, 256, 2) RUN5(4, 256, 2) RUN5(4, 512, 1) RUN5(8, 512, 1) RUN5(16, 256, 2)
  RUN4(8, 256, 2) RUN4(4, 256, 2) RUN4(4, 512, 1) RUN4(8, 512, 1) RUN4(16, 256, 2) RUN4(16, 512, 1) RUN4(8, 1024, 1) RUN4(4, 1024, 1)
  return 0;
}
