// How fast can ONE launch stream a cant-sized matrix (39.8 MB, cold) on gfx950, whatever the kernel does with it?
// Every thread sums 16-byte nontemporal loads; a workgroup writes `out_per_wg` doubles.  Geometry is the variable:
//   streamfloor <MB*10> <wg_size> <loads per thread per trip U: 2|4|6|8|12|16> <trips per workgroup: 1 = one chunk per
//               workgroup; >1 = that many chunks, grid shrinks, next trip's loads issued before this trip's sums>
// 14 rotating copies, 280 back-to-back launches, best of 5; prints us per launch and TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));

template <int U>
__global__ void k_stream(const v2 *__restrict__ src, double *__restrict__ out, long n_pairs, int trips) {
  const int tid = threadIdx.x, W = blockDim.x;
  const long chunk = (long)U * W;
  long base = (long)blockIdx.x * trips * chunk;
  v2 cur[U], nxt[U];
#pragma unroll
  for (int u = 0; u < U; u++) cur[u] = __builtin_nontemporal_load(src + min(base + u * W + tid, n_pairs - 1));
  double acc = 0.0;
  for (int t = 0; t < trips; t++) {
    if (t + 1 < trips) {
#pragma unroll
      for (int u = 0; u < U; u++) nxt[u] = __builtin_nontemporal_load(src + min(base + chunk + u * W + tid, n_pairs - 1));
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += cur[u].x + cur[u].y;
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
    base += chunk;
  }
  __shared__ double part[1024];
  part[tid] = acc;
  __syncthreads();
  if (tid < 32) out[(size_t)blockIdx.x * 32 + tid] = part[tid] + part[tid + 32];
}

int main(int argc, char **argv) {
  const double mb = (argc > 1 ? std::atoi(argv[1]) : 398) / 10.0;
  const int wg = argc > 2 ? std::atoi(argv[2]) : 256, U = argc > 3 ? std::atoi(argv[3]) : 6;
  const int trips = argc > 4 ? std::atoi(argv[4]) : 1;
  const long n_pairs = (long)(mb * 1e6 / 16);
  const long chunk = (long)U * wg * trips;
  const int grid = (int)((n_pairs + chunk - 1) / chunk), copies = 14, steps = 280;
  std::vector<v2 *> src(copies); double *out;
  for (int k = 0; k < copies; k++) { CK(hipMalloc(&src[k], n_pairs * 16 + 4096)); CK(hipMemset(src[k], 0, n_pairs * 16 + 4096)); }
  CK(hipMalloc(&out, (size_t)grid * 32 * 8 + 4096));
  auto launch = [&](int k) {
    switch (U) {
      case 2: hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(wg), 0, 0, src[k], out, n_pairs, trips); break;
      case 4: hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(wg), 0, 0, src[k], out, n_pairs, trips); break;
      case 6: hipLaunchKernelGGL(k_stream<6>, dim3(grid), dim3(wg), 0, 0, src[k], out, n_pairs, trips); break;
      case 8: hipLaunchKernelGGL(k_stream<8>, dim3(grid), dim3(wg), 0, 0, src[k], out, n_pairs, trips); break;
      case 12: hipLaunchKernelGGL(k_stream<12>, dim3(grid), dim3(wg), 0, 0, src[k], out, n_pairs, trips); break;
      default: hipLaunchKernelGGL(k_stream<16>, dim3(grid), dim3(wg), 0, 0, src[k], out, n_pairs, trips);
    }
  };
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 28; i++) launch(i % copies);
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(a));
    for (int i = 0; i < steps; i++) launch(i % copies);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  std::printf("{\"MB\": %.1f, \"wg_size\": %d, \"loads_per_thread_per_trip\": %d, \"trips\": %d, \"grid\": %d, \"usec\": %.2f, \"TBs\": %.2f}\n",
              mb, wg, U, trips, grid, best * 1e3 / steps, mb * 1e6 / (best * 1e-3 / steps) / 1e12);
  return 0;
}
