// The reference's device function triple on MI355X (include/cask_hip_dfe.h): LMem arenas in HBM,
// dramWrite/dramRead as offset copies, and a kernel that decodes the DFE stream format.
// Compatibility path -- correctness first; the tuned path is cask_hip.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cask_hip_dfe.h"

namespace {

#define DFE_CHECK(expr)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      std::fprintf(stderr, "cask_hip_dfe: %s failed: %s\n", #expr, hipGetErrorString(e_));       \
      std::abort(); /* SLiC functions return void and abort the process on failure */            \
    }                                                                                            \
  } while (0)

struct Arena {
  uint8_t *p = nullptr;
  size_t cap = 0;
};
std::vector<Arena> g_arenas;

uint8_t *arena(int ctrl, size_t need) {
  if ((int)g_arenas.size() <= ctrl) g_arenas.resize(ctrl + 1);
  Arena &a = g_arenas[ctrl];
  if (need > a.cap) {
    const size_t chunk = (size_t)64 << 20;
    const size_t cap = ((need + chunk - 1) / chunk) * chunk;
    uint8_t *np = nullptr;
    DFE_CHECK(hipMalloc(reinterpret_cast<void **>(&np), cap));
    DFE_CHECK(hipMemset(np, 0, cap));
    if (a.p) {
      DFE_CHECK(hipMemcpy(np, a.p, a.cap, hipMemcpyDeviceToDevice));
      DFE_CHECK(hipFree(a.p));
    }
    a.p = np;
    a.cap = cap;
  }
  return a.p;
}

// Only the addressed controller's entry is non-zero (Spmv.cpp:109-140); the routing string
// ("split -> tomem<c>" / "frommem<c> -> join") names it too and breaks ties for zero-size calls.
int pick_controller(const cask_hip_dfe_config *cfg, const int64_t *sizes, const char *routing) {
  for (int c = 0; c < cfg->num_controllers; c++)
    if (sizes[c] != 0) return c;
  if (routing) {
    const char *m = std::strstr(routing, "mem");
    if (m && m[3] >= '0' && m[3] <= '9') return std::atoi(m + 3);
  }
  return 0;
}

// records of block b start at the sum of the (input_width-padded) record counts of the blocks before it
__global__ void k_dfe_block_starts(const int *__restrict__ colptr, int n, int n_blocks, int input_width,
                                   long long *__restrict__ starts) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  long long run = 0;
  for (int b = 0; b < n_blocks; b++) {
    starts[b] = run;
    const long long cnt = n > 0 ? colptr[(long long)b * n + n - 1] : 0;
    run += (cnt + input_width - 1) / input_width * input_width;
  }
}

// Run-length-encoded column pointers (SkipEmptyRowsSpmv, Spmv.hpp:213-250): in every block but the first and the
// last, a run of k empty rows is ONE entry k | 1<<31 and a non-empty row keeps its cumulative end, so a block's
// stream is shorter than n entries and where block b starts depends on all blocks before it -- on the DFE the read
// control counts rows as it goes (ParallelCsrReadControl.java:175-189).  Here: does the stream carry any bit-31
// entry at all (cumulative ends never do)? if so find every block's first entry by one counting walk over the
// stream, then expand the blocks in parallel (one thread per block) into the plain n-entries-per-block form the
// row kernel reads.
__global__ void k_dfe_has_rle(const unsigned *__restrict__ colptr, long long len, int *flag) {
  bool any = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
    any = any || (colptr[i] & 0x80000000u);
  if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void k_dfe_rle_block_offsets(const unsigned *__restrict__ colptr, long long len, int n, int n_blocks,
                                        long long *__restrict__ first_entry, int *err) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  long long i = 0;
  for (int b = 0; b < n_blocks; b++) {
    first_entry[b] = i;
    long long rows = 0;
    while (rows < n) {
      if (i >= len) { *err = 1; return; }
      const unsigned e = colptr[i++];
      rows += (e & 0x80000000u) ? (long long)(e & 0x7fffffffu) : 1;
    }
    if (rows != n) { *err = 1; return; }
  }
  first_entry[n_blocks] = i;
}
__global__ void k_dfe_rle_expand(const unsigned *__restrict__ colptr, const long long *__restrict__ first_entry, int n,
                                 int n_blocks, int *__restrict__ expanded) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  int *out = expanded + (long long)b * n;
  int row = 0, prev = 0;
  for (long long i = first_entry[b]; i < first_entry[b + 1] && row < n; i++) {
    const unsigned e = colptr[i];
    if (e & 0x80000000u) {
      int k = (int)(e & 0x7fffffffu);
      for (; k > 0 && row < n; k--) out[row++] = prev;
    } else {
      prev = (int)e;
      out[row++] = prev;
    }
  }
}

// One thread per row; per column block: the row's run of packed {double value, int32 index} records
// (12 bytes, Spmv.hpp:14-20) against that block's slice of x, partial sums added in block order
// (SpmvKernel.java:61-78 + BramSpmvReductionKernel :250-309).
__global__ void k_dfe_rows(const int *__restrict__ colptr, const unsigned *__restrict__ records,
                           const long long *__restrict__ starts, const double *__restrict__ x, int n, int n_blocks,
                           int cache_size, double *__restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double y = 0.0;
  for (int b = 0; b < n_blocks; b++) {
    const int *cp = colptr + (long long)b * n;
    const int lo = r ? cp[r - 1] : 0, hi = cp[r];
    const double *xb = x + (long long)b * cache_size;
    double acc = 0.0;
    for (int k = lo; k < hi; k++) {
      const unsigned *rec = records + 3 * (starts[b] + k);
      const double v = __hiloint2double((int)rec[1], (int)rec[0]);
      acc += xb[(int)rec[2]] * v;
    }
    y += acc;
  }
  out[r] = y;
}

}  // namespace

extern "C" {

void cask_hip_dfe_dram_write(const cask_hip_dfe_config *cfg, int64_t size_bytes_cpu, const int64_t *sizes,
                             const int64_t *starts, const uint8_t *in, const char *routing) {
  const int c = pick_controller(cfg, sizes, routing);
  const int64_t bytes = sizes[c] ? sizes[c] : size_bytes_cpu;
  if (bytes <= 0) return;
  uint8_t *base = arena(c, (size_t)(starts[c] + bytes));
  DFE_CHECK(hipMemcpy(base + starts[c], in, (size_t)bytes, hipMemcpyHostToDevice));
}

void cask_hip_dfe_dram_read(const cask_hip_dfe_config *cfg, int64_t size_bytes_cpu, const int64_t *sizes,
                            const int64_t *starts, uint8_t *out, const char *routing) {
  const int c = pick_controller(cfg, sizes, routing);
  const int64_t bytes = sizes[c] ? sizes[c] : size_bytes_cpu;
  if (bytes <= 0) return;
  uint8_t *base = arena(c, (size_t)(starts[c] + bytes));
  DFE_CHECK(hipMemcpy(out, base + starts[c], (size_t)bytes, hipMemcpyDeviceToHost));
}

void cask_hip_dfe_run(const cask_hip_dfe_config *cfg, int64_t nIterations, int64_t nBlocks, int64_t, const int64_t *colPtrStart,
                      const int32_t *colptrSizes, const int64_t *recordsStart, const int32_t *, const int32_t *nrows,
                      const int64_t *outStart, const int32_t *, const int32_t *, const int64_t *vStart) {
  if (cfg->num_pipes <= 0 || cfg->num_controllers <= 0 || cfg->num_pipes % cfg->num_controllers != 0) {
    std::fprintf(stderr, "cask_hip_dfe: numPipes should be a multiple of numControllers\n");
    std::abort();
  }
  const int per_ctrl = cfg->num_pipes / cfg->num_controllers;
  long long *d_starts = nullptr;
  DFE_CHECK(hipMalloc(reinterpret_cast<void **>(&d_starts), sizeof(long long) * (size_t)(nBlocks > 0 ? nBlocks : 1)));
  for (int64_t it = 0; it < (nIterations > 0 ? nIterations : 1); it++)
    for (int p = 0; p < cfg->num_pipes; p++) {
      const int n = nrows[p];
      if (n <= 0) continue;
      const int c = p / per_ctrl;
      if (c >= (int)g_arenas.size() || !g_arenas[c].p) {
        std::fprintf(stderr, "cask_hip_dfe: run before dramWrite on controller %d\n", c);
        std::abort();
      }
      uint8_t *base = arena(c, (size_t)(outStart[p] + (int64_t)n * 8));
      const int *colptr = reinterpret_cast<const int *>(base + colPtrStart[p]);
      const unsigned *records = reinterpret_cast<const unsigned *>(base + recordsStart[p]);
      const double *x = reinterpret_cast<const double *>(base + vStart[p]);
      double *out = reinterpret_cast<double *>(base + outStart[p]);
      // run-length-encoded column pointers? (only the first iteration needs to look)
      int *expanded = nullptr;
      if (nBlocks > 2 && colptrSizes && colptrSizes[p] > 0) {
        const long long len = colptrSizes[p] / 4;
        int *d_flag = nullptr, h_flag[2] = {0, 0};
        DFE_CHECK(hipMalloc(reinterpret_cast<void **>(&d_flag), 2 * sizeof(int)));
        DFE_CHECK(hipMemset(d_flag, 0, 2 * sizeof(int)));
        hipLaunchKernelGGL(k_dfe_has_rle, dim3((unsigned)std::min<long long>(1024, (len + 255) / 256)), dim3(256), 0, 0,
                           reinterpret_cast<const unsigned *>(colptr), len, d_flag);
        DFE_CHECK(hipMemcpy(h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost));
        if (h_flag[0]) {
          long long *d_first = nullptr;
          DFE_CHECK(hipMalloc(reinterpret_cast<void **>(&d_first), sizeof(long long) * (size_t)(nBlocks + 1)));
          DFE_CHECK(hipMalloc(reinterpret_cast<void **>(&expanded), sizeof(int) * (size_t)n * (size_t)nBlocks));
          hipLaunchKernelGGL(k_dfe_rle_block_offsets, dim3(1), dim3(64), 0, 0, reinterpret_cast<const unsigned *>(colptr),
                             len, n, (int)nBlocks, d_first, d_flag + 1);
          hipLaunchKernelGGL(k_dfe_rle_expand, dim3(((int)nBlocks + 63) / 64), dim3(64), 0, 0,
                             reinterpret_cast<const unsigned *>(colptr), d_first, n, (int)nBlocks, expanded);
          DFE_CHECK(hipMemcpy(h_flag, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost));
          DFE_CHECK(hipFree(d_first));
          if (h_flag[1]) {
            std::fprintf(stderr, "cask_hip_dfe: run-length-encoded column pointers do not add up to %d rows per block\n", n);
            std::abort();
          }
          colptr = expanded;
        }
        DFE_CHECK(hipFree(d_flag));
      }
      hipLaunchKernelGGL(k_dfe_block_starts, dim3(1), dim3(64), 0, 0, colptr, n, (int)nBlocks, cfg->input_width, d_starts);
      hipLaunchKernelGGL(k_dfe_rows, dim3((n + 255) / 256), dim3(256), 0, 0, colptr, records, d_starts, x, n,
                         (int)nBlocks, cfg->cache_size, out);
      DFE_CHECK(hipGetLastError());
      if (expanded) {
        DFE_CHECK(hipDeviceSynchronize());
        DFE_CHECK(hipFree(expanded));
      }
    }
  DFE_CHECK(hipDeviceSynchronize());
  DFE_CHECK(hipFree(d_starts));
}

void cask_hip_dfe_reset(void) {
  for (Arena &a : g_arenas)
    if (a.p) (void)hipFree(a.p);
  g_arenas.clear();
}

}  // extern "C"
