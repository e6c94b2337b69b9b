// Development probe: at which dynamic-LDS sizes does the number of resident 256-thread workgroups per CU change?
//   hipcc --offload-arch=gfx950 -O2 -o build/lds_granule tools/lds_granule.hip && build/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out) {
  extern __shared__ double s[];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[(threadIdx.x + 1) % blockDim.x];
}
int main() {
  int prev = -1;
  for (int bytes = 16 * 1024; bytes <= 40 * 1024; bytes += 16) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, bytes) != hipSuccess) return 1;
    if (occ != prev) printf("%6d bytes: %d workgroups per CU\n", bytes, occ);
    prev = occ;
  }
  return 0;
}
