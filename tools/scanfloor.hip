// The streaming floor of the SCAN kernel's block shape on a webbase-1M-sized problem, without the kernel: every
// workgroup of 256 threads streams E consecutive (value, column) pairs in 16-byte loads plus one 4-byte word per
// thread, parks a product per element in LDS and writes R consecutive doubles of y.  Cold: COPIES rotating sets.
//   scanfloor <elems/block> <rows/block> <nt 0|1> <y 0|1> <meta 0|1> <compute 0..2> <ipt> <descriptor 0|1> <gather flavour 0..6> <p_local>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

struct dbl2 { double x, y; };
struct int2v { int x, y; };
template <bool NT, class T> __device__ __forceinline__ T ld(const T *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ dbl2 ld2(const dbl2 *p) {
  dbl2 r; const double *q = reinterpret_cast<const double *>(p);
  typedef double v2 __attribute__((ext_vector_type(2)));
  v2 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2 *>(q)) : *reinterpret_cast<const v2 *>(q);
  r.x = t.x; r.y = t.y; return r;
}
template <bool NT> __device__ __forceinline__ int2v ldi2(const int2v *p) {
  typedef int v2 __attribute__((ext_vector_type(2)));
  v2 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2 *>(p)) : *reinterpret_cast<const v2 *>(p);
  int2v r; r.x = t.x; r.y = t.y; return r;
}

// GATHER: 0 none (the column index itself is the factor), 1 plain x[col], 2 nontemporal, 3 workgroup scope (sc0),
// 4 agent scope (sc1), 5 system scope (sc0 sc1), 7 / 8 plain for near columns and sc1 / nt for far ones, 6 returning atomic add of `zero` at L2 (diagnostic only: writes x)
template <int G> __device__ __forceinline__ double gather(const double *x, int c, unsigned long long zero, int row = 0) {
  if (G == 7 || G == 8) {                              // far columns (not within +-1000 of the row) bypass the L1: sc1 (7) / nt (8)
    const bool far = c < row - 1000 || c > row + 1000;
    if (!far) return x[c];
    return G == 7 ? __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __builtin_nontemporal_load(x + c);
  }
  if (G == 1) return x[c];
  if (G == 2) return __builtin_nontemporal_load(x + c);
  if (G == 3) return __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (G == 4) return __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (G == 5) return __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (G == 6) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(const_cast<double *>(x + c));
    return __longlong_as_double((long long)__hip_atomic_fetch_add(q, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  return (double)c;
}

template <int IPT, bool NT, int G>
__global__ __launch_bounds__(256) void k_floor(const double *__restrict__ val, const int *__restrict__ ci,
                                               const unsigned *__restrict__ meta, double *__restrict__ y, int elems,
                                               int rows, int total, int use_y, int use_meta, int compute,
                                               const int4 *__restrict__ desc, const double *x, unsigned long long zero) {
  extern __shared__ double prod[];
  const int tid = threadIdx.x, b = blockIdx.x;
  // desc: the block's first element comes from a cold 16-byte descriptor (through the scalar cache), as in the engine
  const int start = desc ? desc[b].x : b * elems, base = start & ~1, first = base >> 1;
  const int npairs = (elems + (start - base) + 1) >> 1, last = min(first + npairs - 1, (total >> 1) - 1);
  unsigned mw = use_meta ? ld<NT>(meta + (size_t)b * 256 + tid) : 0u;
  dbl2 v[IPT / 2]; int2v c[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int p = min(first + u * 256 + tid, last);
    v[u] = ld2<NT>(reinterpret_cast<const dbl2 *>(val) + p);
    c[u] = ldi2<NT>(reinterpret_cast<const int2v *>(ci) + p);
  }
  dbl2 xv[IPT / 2];                                    // every gather of the thread goes out before the first use
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int e = base + 2 * (u * 256 + tid), row = (int)((long long)e * 1000005 / total);
    xv[u].x = gather<G>(x, c[u].x + (int)mw, zero, row);
    xv[u].y = gather<G>(x, c[u].y, zero, row);
  }
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int e = 2 * (u * 256 + tid);
    prod[e + (e >> 3)] = v[u].x * xv[u].x;
    prod[e + 1 + ((e + 1) >> 3)] = v[u].y * xv[u].y;
  }
  __syncthreads();
  if (compute) {                                       // a run of IPT products per thread, then a wave scan (shape only)
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < IPT; j++) s += prod[(IPT + 1) * tid + j];
    if (compute > 1)
      for (int o = 1; o < 64; o <<= 1) s += __shfl_up(s, o);
    __syncthreads();
    prod[tid * 9] = s;
    __syncthreads();
  }
  if (use_y)
    for (int i = tid; i < rows; i += 256) y[(size_t)b * rows + i] = prod[i + (i >> 3)];
  else if (tid == 0) y[b] = prod[0];
}

int main(int argc, char **argv) {
  const int elems = argc > 1 ? std::atoi(argv[1]) : 2047, rows = argc > 2 ? std::atoi(argv[2]) : 682;
  const int nt = argc > 3 ? std::atoi(argv[3]) : 1, use_y = argc > 4 ? std::atoi(argv[4]) : 1;
  const int use_meta = argc > 5 ? std::atoi(argv[5]) : 1, compute = argc > 6 ? std::atoi(argv[6]) : 0;
  const int ipt = argc > 7 ? std::atoi(argv[7]) : 8;
  const int use_desc = argc > 8 ? std::atoi(argv[8]) : 0;
  const int g = argc > 9 ? std::atoi(argv[9]) : 0;                 // gather flavour
  const double p_local = argc > 10 ? std::atof(argv[10]) : 0.7;    // share of columns within +-1000 of the row
  const int nnz = 3105536, copies = 14, steps = 140;
  const int blocks = (nnz + elems - 1) / elems;
  const int n = 1000005;
  std::vector<int> hci(nnz + 64, 0);                            // webbase-like columns: ~3.1 per row, p_local near the row
  { unsigned long long st = 88172645463325252ull; auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int e = 0; e < nnz; e++) {
      const int row = (int)((long long)e * n / nnz);
      const bool local = (rnd() % 1000) < (unsigned long long)(p_local * 1000);
      long long c = local ? row + (long long)(rnd() % 2001) - 1000 : (long long)(rnd() % n);
      hci[e] = (int)std::min<long long>(std::max<long long>(c, 0), n - 1);
    } }
  double *x; CK(hipMalloc(&x, (size_t)(n + 64) * 8)); CK(hipMemset(x, 0, (size_t)(n + 64) * 8));
  std::vector<double *> val(copies), y(copies); std::vector<int *> ci(copies); std::vector<unsigned *> meta(copies); std::vector<int4 *> desc(copies);
  for (int k = 0; k < copies; k++) {
    CK(hipMalloc(&val[k], (size_t)(nnz + 64) * 8)); CK(hipMemset(val[k], 0, (size_t)(nnz + 64) * 8));
    CK(hipMalloc(&ci[k], (size_t)(nnz + 64) * 4)); CK(hipMemcpy(ci[k], hci.data(), (size_t)(nnz + 64) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&meta[k], (size_t)blocks * 256 * 4)); CK(hipMemset(meta[k], 0, (size_t)blocks * 256 * 4));
    CK(hipMalloc(&y[k], (size_t)blocks * (rows + 8) * 8));
    std::vector<int4> h(blocks); for (int i = 0; i < blocks; i++) h[i] = make_int4(i * elems, 0, 0, 0);
    CK(hipMalloc(&desc[k], (size_t)blocks * 16)); CK(hipMemcpy(desc[k], h.data(), (size_t)blocks * 16, hipMemcpyHostToDevice));
  }
  const size_t lds = (size_t)(256 * ipt + 256 * ipt / 8 + 2600) * 8;
  auto launch = [&](int k) {
#define GO(I, N, G) hipLaunchKernelGGL((k_floor<I, N, G>), dim3(blocks), dim3(256), lds, 0, val[k], ci[k], meta[k], y[k], elems, rows, nnz, use_y, use_meta, compute, use_desc ? desc[k] : nullptr, x, 0ull)
#define GG(I, N) do { switch (g) { case 0: GO(I, N, 0); break; case 1: GO(I, N, 1); break; case 2: GO(I, N, 2); break; case 3: GO(I, N, 3); break; \
                                   case 4: GO(I, N, 4); break; case 5: GO(I, N, 5); break; case 7: GO(I, N, 7); break; case 8: GO(I, N, 8); break; default: GO(I, N, 6); } } while (0)
    if (ipt == 16) GG(16, true);
    else if (ipt == 4) GG(4, true);
    else if (nt) GG(8, true);
    else GG(8, false);
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
  const double bytes = (double)nnz * 12 + (use_meta ? blocks * 1024.0 : 0) + (use_y ? (double)blocks * rows * 8 : 0);
  std::printf("{\"elems_per_block\": %d, \"rows_per_block\": %d, \"ipt\": %d, \"nt\": %d, \"y\": %d, \"meta\": %d, \"compute\": %d, \"descriptor\": %d, \"gather\": %d, \"p_local\": %.2f, \"blocks\": %d, \"usec\": %.2f, \"MB\": %.1f, \"TBs\": %.2f}\n",
              elems, rows, ipt, nt, use_y, use_meta, compute, use_desc, g, p_local, blocks, best * 1e3 / steps, bytes / 1e6, bytes / (best * 1e-3 / steps) / 1e12);
  return 0;
}
