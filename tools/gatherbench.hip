// Development microbenchmark: what does a random 8-byte gather that misses L2 cost in fabric traffic, by load
// flavour?  N random indices into an 8 MB table (like the far columns of the webbase-like matrix).
//   hipcc -O3 --offload-arch=gfx950 -o build/gatherbench tools/gatherbench.hip ; rocprofv3 --pmc ... -- build/gatherbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

typedef __attribute__((address_space(1))) const double gdouble;

template <int MODE>
__device__ __forceinline__ double ld(const double *p) {
  if (MODE == 0) return *p;
  if (MODE == 1) return __builtin_nontemporal_load(p);
  if (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  double v;
  if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 5) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int MODE>
__global__ void k_gather(int n, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int k = i; k + 3 * stride < n; k += 4 * stride) {
    const int a = idx[k], b = idx[k + stride], c = idx[k + 2 * stride], d = idx[k + 3 * stride];
    const double va = ld<MODE>(x + a), vb = ld<MODE>(x + b), vc = ld<MODE>(x + c), vd = ld<MODE>(x + d);
    acc += va + vb + vc + vd;
  }
  out[i] = acc;
}

int main(int argc, char **argv) {
  const int alloc = argc > 1 ? atoi(argv[1]) : 0;   // 0 hipMalloc, 1 uncached, 2 fine-grained (hipExtMallocWithFlags)
  const int n_x = 1 << 20, n = 1 << 20;   // 8 MB table, 1M gathers
  std::vector<int> h(n);
  std::mt19937 rng(1);
  for (auto &v : h) v = rng() % n_x;
  int *idx; double *x, *out, *flush;
  hipMalloc(&idx, n * 4);
  if (alloc == 0) hipMalloc(&x, n_x * 8);
  else if (hipExtMallocWithFlags((void **)&x, n_x * 8, alloc == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained) != hipSuccess) { printf("allocation mode %d refused\n", alloc); return 1; }
  printf("table allocation: %s\n", alloc == 0 ? "hipMalloc" : alloc == 1 ? "uncached" : "fine-grained");
  hipMalloc(&out, 1 << 22); hipMalloc(&flush, 512 << 20);
  hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(x, 0, n_x * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"plain", "nt", "agent(sc1)", "system(sc0 sc1)", "sc0 sc1 nt", "sc1 nt"};
  for (int mode = 0; mode < 6; mode++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipMemsetAsync(flush, rep, 512 << 20, 0);      // evict x from L2 and the Infinity Cache
      hipEventRecord(e0, 0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_gather<0>, dim3(1024), dim3(256), 0, 0, n, idx, x, out); break;
        case 1: hipLaunchKernelGGL(k_gather<1>, dim3(1024), dim3(256), 0, 0, n, idx, x, out); break;
        case 2: hipLaunchKernelGGL(k_gather<2>, dim3(1024), dim3(256), 0, 0, n, idx, x, out); break;
        case 3: hipLaunchKernelGGL(k_gather<3>, dim3(1024), dim3(256), 0, 0, n, idx, x, out); break;
        case 4: hipLaunchKernelGGL(k_gather<4>, dim3(1024), dim3(256), 0, 0, n, idx, x, out); break;
        default: hipLaunchKernelGGL(k_gather<5>, dim3(1024), dim3(256), 0, 0, n, idx, x, out); break;
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-16s %8.2f us\n", names[mode], best * 1e3);
  }
  return 0;
}
