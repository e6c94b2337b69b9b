// What a grid-wide hand-off costs inside one launch on gfx950, against the ~1.45 us boundary between two dependent
// launches of a graph.  G co-resident workgroups of 256 threads run R rounds of a two-level
// arrival (workgroup -> one of NG group counters -> top counter, all monotonically increasing, agent scope) followed
// by a bounded poll of the top counter; with WORK > 0 each workgroup also streams WORK KiB between two barriers (a
// stand-in for the phases of a CG pass).  Prints microseconds per round.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

// MODE 0: every workgroup polls the top counter with acquire loads (the textbook barrier).  MODE 1: relaxed polls of
// the top counter, one acquire fence after.  MODE 2: the last arrival writes one release word per group; a workgroup
// polls its group's word (relaxed), then fences.
template <int MODE, bool FENCES>
__global__ __launch_bounds__(256) void k_rounds(int *grp, int *top, int *failed, int rounds, int n_groups, int per_group,
                                                const double *src, double *dst, int work_elems, int sleep) {
  const int g = blockIdx.x / per_group;
  const int members = min(per_group, (int)gridDim.x - g * per_group);
  int *release = top + 64;                    // [64 groups] x 32 ints apart
  double acc = 0.0;
  for (int r = 0; r < rounds; ++r) {
    if (work_elems) {
      const double *s = src + (size_t)blockIdx.x * work_elems;
      for (int i = threadIdx.x; i < work_elems; i += 256) acc += s[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // FENCES: release/acquire at agent scope (an L2 write-back and an invalidate per workgroup and round on gfx950,
      // whose 8 L2s are not coherent with each other); without: relaxed atomics only -- the kernel then has to publish
      // its data with write-through (sc1) stores and read it with sc1 loads itself, as the engine's hand-offs do
      const int before = FENCES ? __hip_atomic_fetch_add(&grp[g * 32], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
                                : __hip_atomic_fetch_add(&grp[g * 32], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (r + 1) * members - 1) {
        const int t = FENCES ? __hip_atomic_fetch_add(top, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
                             : __hip_atomic_fetch_add(top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2 && t == (r + 1) * n_groups - 1)
          for (int k = 0; k < n_groups; ++k)
            __hip_atomic_store(&release[k * 32], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      int spins = 0;
      if (MODE == 0) {
        while (__hip_atomic_load(top, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (r + 1) * n_groups) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 22)) { *failed = 1; break; }
        }
      } else {
        const int *word = MODE == 1 ? top : &release[g * 32];
        const int want = MODE == 1 ? (r + 1) * n_groups : r + 1;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 22)) { *failed = 1; break; }
        }
        if (FENCES) __atomic_thread_fence(__ATOMIC_ACQUIRE);
      }
    }
    __syncthreads();
  }
  if (work_elems) dst[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 400;
  int *d; CK(hipMalloc(&d, (64 * 32 + 64 + 64 * 32 + 64) * sizeof(int)));
  double *src, *dst; const int max_g = 2048, max_work = 4096;
  CK(hipMalloc(&src, (size_t)max_g * max_work * 8)); CK(hipMemset(src, 0, (size_t)max_g * max_work * 8));
  CK(hipMalloc(&dst, (size_t)max_g * 256 * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t n_ints = 64 * 32 + 64 + 64 * 32 + 64;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  for (int fences : {1, 0})
  for (int mode : {0, 1, 2})
    for (int sleep : {1})
      for (int work : {0, 2048})
        for (int G : {256, 1024, 1466})
          for (int NG : {1, 8, 64}) {
            if (mode == 0 && !fences) continue;
            if (mode != 2 && NG == 64) continue;
            auto kern = mode == 0 ? k_rounds<0, true> : mode == 1 ? (fences ? k_rounds<1, true> : k_rounds<1, false>) : (fences ? k_rounds<2, true> : k_rounds<2, false>);
            int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0));
            if (G > occ * p.multiProcessorCount) continue;           // all workgroups must be resident
            const int per_group = (G + NG - 1) / NG, n_groups = (G + per_group - 1) / per_group;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
              CK(hipMemset(d, 0, n_ints * sizeof(int)));
              CK(hipEventRecord(a));
              hipLaunchKernelGGL(kern, dim3(G), dim3(256), 0, 0, d, d + 64 * 32, d + n_ints - 1, rounds, n_groups, per_group, src, dst, work, sleep);
              CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
              float ms; CK(hipEventElapsedTime(&ms, a, b));
              if (ms < best) best = ms;
            }
            int failed; CK(hipMemcpy(&failed, d + n_ints - 1, 4, hipMemcpyDeviceToHost));
            std::printf("{\"fences\": %d, \"poll\": \"%s\", \"sleep\": %d, \"workgroups\": %d, \"group_counters\": %d, \"stream_KiB_per_workgroup_per_round\": %d, \"usec_per_round\": %.3f, \"poll_gave_up\": %d}\n",
                        fences, mode == 0 ? "top, acquire loads" : mode == 1 ? "top, relaxed loads" : "per-group release word", sleep * 8,
                        G, n_groups, work * 8 / 1024, best * 1e3f / rounds, failed);
            std::fflush(stdout);
          }
  return 0;
}
