// Peer-to-peer exchange step of the sharded SpMV (include/cask_hip_p2p.h): shared x slices
// (hipIpc handles between the per-GPU processes) and the halo pull kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "cask_hip.h"
#include "cask_hip_p2p.h"
#include "internal.hpp"
#include "spmv_common.hpp"

namespace {

#define P2P_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return caskhip::report_failure(CASK_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

static_assert(sizeof(hipIpcMemHandle_t) == CASK_HIP_SHARED_HANDLE_BYTES, "handle size is part of the ABI");

// A halo is a few hundred to a few hundred thousand scattered 8-byte entries; each lane issues
// its remote loads back to back (4 in flight) so one round trip over xGMI covers them.  The entries live in a peer
// GPU's memory and change between products: they are read with system-scope loads (sc0 sc1, like the seam loads of
// the product kernel, merge_kernel.hpp load_at) -- served by the owner's memory, never by a line this GPU's L2 kept
// from the previous product.
__device__ __forceinline__ double pull_at(uint64_t addr) {
  typedef __attribute__((address_space(1))) const double gdouble;
  return __hip_atomic_load(reinterpret_cast<gdouble *>(addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_halo_pull(int64_t n, const uint64_t *__restrict__ src, double *__restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; j + 3 * stride < n; j += 4 * stride) {
    const double a = pull_at(src[j]);
    const double b = pull_at(src[j + stride]);
    const double c = pull_at(src[j + 2 * stride]);
    const double d = pull_at(src[j + 3 * stride]);
    dst[j] = a;
    dst[j + stride] = b;
    dst[j + 2 * stride] = c;
    dst[j + 3 * stride] = d;
  }
  for (; j < n; j += stride) dst[j] = pull_at(src[j]);
}

// ---- push all-gather ------------------------------------------------------------------------------------------
// The exchange SURVEY section 5 asked for at latency-bound sizes: every rank stores its slice of x straight into
// every peer's gathered vector through the IPC mappings (xGMI is point-to-point: one hop per peer), then one flag
// per peer; a rank's product may start when all world flags of this exchange have arrived.  ONE launch per
// exchange, no collective library on the data path.
//   * data: 16-byte write-through stores at system scope (sc0 sc1): the bytes are in the owner's memory when the
//     storing wave's vmcnt drains;
//   * the workgroup that arrives last at the launch's local counter knows every store of the launch has drained; it
//     takes the next sequence number (a device-resident counter: the same launch may be replayed from a graph),
//     stores it into slot [rank] of every peer's flag array (system scope), then polls its own flag array until
//     every slot has reached that number, and leaves through a system-scope acquire.  Sequence numbers only grow,
//     so a flag never needs a reset and a peer that is a whole exchange ahead reads as "arrived";
//   * gathered vectors are double-buffered by the parity of the sequence number: a peer may push exchange k+1 while
//     this rank's product still reads the buffer of exchange k; it cannot push k+2 before this rank has pushed
//     k+1, i.e. after that product (stream order);
//   * every poll is bounded: on a timeout the launch raises the error word (cask_hip_push_check) and ends.
struct PushTables {
  double *full[2][CASK_HIP_PUSH_MAX_WORLD];     // every rank's two gathered vectors as mapped in this process
  int *flags[CASK_HIP_PUSH_MAX_WORLD];          // every rank's flag array as mapped in this process
};
constexpr long long PUSH_POLL_LIMIT = 1ll << 24;              // x ~0.1 us sleep: about two seconds

__device__ __forceinline__ void store16_sys(void *p, caskhip::dbl2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

__global__ void k_push_allgather(const double *__restrict__ src, int64_t stride, int rank, int world, PushTables t,
                                 int *state /* [0] arrivals, [1] sequence number, [2] error */, int parity_of_next) {
  __shared__ int s_last, s_seq;
  const caskhip::dbl2 *s2 = reinterpret_cast<const caskhip::dbl2 *>(src);
  const int64_t n2 = stride >> 1, step = (int64_t)gridDim.x * blockDim.x;
  // few workgroups, many bytes in flight each: every workgroup ends with an atomic arrival on one counter, and those
  // serialise at ~12 ns apiece (977 workgroups: 12 us of arrivals for a 2 us copy)
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n2; i0 += 4 * step) {
    caskhip::dbl2 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = s2[min(i0 + u * step, n2 - 1)];
    for (int g = 0; g < world; g++) {
      caskhip::dbl2 *dst = reinterpret_cast<caskhip::dbl2 *>(t.full[parity_of_next][g] + (int64_t)rank * stride);
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (i0 + u * step < n2) {
          // the own copy is read by this GPU's next launch only: plain stores -- and none at all when the caller
          // keeps its slice in place (d_local IS slot [rank] of the vector being filled: cask_hip_push_own_slot)
          if (g == rank) { if (dst != s2) dst[i0 + u * step] = v[u]; }
          else store16_sys(dst + i0 + u * step, v[u]);
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave, before the barrier
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) {
    __hip_atomic_store(state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_seq = state[1] + 1;
    state[1] = s_seq;
  }
  __syncthreads();
  const int seq = s_seq;
  if ((int)threadIdx.x < world && (int)threadIdx.x != rank) {   // (the own slice needs no flag: same launch, same memory)
    __hip_atomic_store(t.flags[threadIdx.x] + rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int *mine = t.flags[rank] + threadIdx.x;
    long long spins = 0;
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
      if (++spins > PUSH_POLL_LIMIT) {
        state[2] = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  // No fence here: the gathered vector is read by the NEXT launch on this stream, which starts behind a kernel
  // boundary -- its acquire is what keeps lines of the previous exchange out of the way (a system-scope acquire at this
  // point cost 1.7 us per exchange and protected nothing this launch reads).
}

// ---- push all-reduce of a few scalars (the dot products of a row-sharded solver pass) --------------------------
// Same transport as the push all-gather, for 1-4 doubles, and ONE trip per reduction: a contribution travels as a
// 16-byte granule {value, sequence number} written by one 16-byte system-scope store, so the value's arrival IS its
// flag (a separate flag costs a second dependent round trip: stores, drain, flag, poll, table loads measured 8 us
// per reduction on one GPU, more than the 6 us of an RCCL all-reduce).  Thread (g, c) stores this rank's value c
// into slot [rank][c] of peer g's table and polls slot [g][c] of its own table until the tag is this reduction's
// number; then one thread per value adds the world contributions IN RANK ORDER -- every rank adds the same numbers
// in the same order, so every rank holds the same bits and takes the same decisions.  Tables are double-buffered by
// sequence parity: a peer that is one reduction ahead writes the other half.  (16-byte stores arriving untorn is what
// MI355X_MICROARCH observes for gfx950; the tag is the second 8 bytes and is bound to the value -- granule_tag --, so a
// granule torn either way reads as "not yet".)
constexpr int PUSH_SCALARS = 4;
struct alignas(16) PushGranule {
  double value;
  long long tag;
};
// The tag binds the sequence number to the value it travels with: tag = seq * K ^ bits(value), K odd.  Should the two
// 8-byte halves of a granule ever land apart (16-byte xGMI stores arriving whole is an observation, not a guarantee --
// ADVICE r3), a new tag next to the old value, or the old tag next to the new value, fails the check unless the two
// values are equal (then either is right) or their bit difference happens to equal (seq_old ^ seq_new) * K -- a fixed
// pseudo-random 64-bit pattern, not something two dot products differ by.  A torn granule therefore reads as "not yet".
__device__ __forceinline__ long long granule_tag(long long seq, double value) {
  return (long long)((unsigned long long)seq * 0x9E3779B97F4A7C15ull) ^ __double_as_longlong(value);
}
// The exchange proper, for a workgroup of 4 * CASK_HIP_PUSH_MAX_WORLD threads: vals[c] (c < count) of this rank in,
// the rank-order sum over all ranks out (through `got`, LDS).
__device__ __forceinline__ void push_reduce(const double *mine_vals, double *out, int count, int rank, int world,
                                            const PushTables &t, int *state, double (*got)[PUSH_SCALARS]) {
  const int g = threadIdx.x >> 2, c = threadIdx.x & 3;
  const long long seq = (long long)state[3] + 1;
  const int parity = (int)(seq & 1);
  // a rank's flag region: int vec_flags[64]; int reserved[64]; PushGranule table[2][64][PUSH_SCALARS]
  if (g < world && c < count) {
    const double mine = mine_vals[c];
    if (g == rank) {
      got[g][c] = mine;
    } else {
      PushGranule *peer = reinterpret_cast<PushGranule *>(t.flags[g] + 2 * CASK_HIP_PUSH_MAX_WORLD) +
                          ((size_t)parity * CASK_HIP_PUSH_MAX_WORLD + rank) * PUSH_SCALARS + c;
      caskhip::dbl2 gr;
      gr.x = mine;
      gr.y = __longlong_as_double(granule_tag(seq, mine));
      store16_sys(peer, gr);
      const PushGranule *own = reinterpret_cast<const PushGranule *>(t.flags[rank] + 2 * CASK_HIP_PUSH_MAX_WORLD) +
                               ((size_t)parity * CASK_HIP_PUSH_MAX_WORLD + g) * PUSH_SCALARS + c;
      long long spins = 0;
      while (true) {
        caskhip::dbl2 in;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(in) : "v"(own) : "memory");
        if (__double_as_longlong(in.y) == granule_tag(seq, in.x)) {
          got[g][c] = in.x;
          break;
        }
        if (++spins > PUSH_POLL_LIMIT) {
          state[2] = 1;
          got[g][c] = 0.0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < count) {
    double sum = 0.0;
    for (int r = 0; r < world; r++) sum += got[r][threadIdx.x];
    out[threadIdx.x] = sum;
  }
  if (threadIdx.x == 0) state[3] = (int)seq;
}

__global__ void k_push_allreduce(double *__restrict__ vals, int count, int rank, int world, PushTables t, int *state) {
  __shared__ double got[CASK_HIP_PUSH_MAX_WORLD][PUSH_SCALARS];
  __shared__ double mine[PUSH_SCALARS];
  if ((int)threadIdx.x < count) mine[threadIdx.x] = vals[threadIdx.x];
  __syncthreads();
  push_reduce(mine, vals, count, rank, world, t, state, got);
}

// The solver's "sum this rank's partial sums, then all-reduce" as ONE launch (it was k_sum_to_scalars + the reduction):
// out[0] = all-rank sum of sum(pa[0..na)), out[1] likewise for pb when given.  Nothing happens once *done is set (every
// rank sees the same flag, so every rank skips the same exchanges).
__global__ void k_push_sum_allreduce(const double *__restrict__ pa, int na, const double *__restrict__ pb, int nb,
                                     double *__restrict__ out, const int *done, int rank, int world, PushTables t, int *state) {
  __shared__ double got[CASK_HIP_PUSH_MAX_WORLD][PUSH_SCALARS];
  __shared__ double mine[PUSH_SCALARS];
  __shared__ double red[16];
  if (done && *done) return;
  const double a = caskhip::sum_partials(pa, na, red);
  if (threadIdx.x == 0) mine[0] = a;
  if (pb) {
    const double b = caskhip::sum_partials(pb, nb, red);
    if (threadIdx.x == 0) mine[1] = b;
  }
  __syncthreads();
  push_reduce(mine, out, pb ? 2 : 1, rank, world, t, state, got);
}

}  // namespace

struct cask_hip_push {
  int rank = 0, world = 1;
  int64_t stride = 0;
  PushTables tables{};
  int *d_state = nullptr;
  int parity = 0;                  // buffer the NEXT exchange fills (host mirror of the device sequence number)
};

extern "C" {

int cask_hip_push_create(int32_t rank, int32_t world, int64_t stride, const uint64_t *full_addr, const uint64_t *flag_addr,
                         cask_hip_push **out) {
  if (!out || !full_addr || !flag_addr || world < 1 || world > CASK_HIP_PUSH_MAX_WORLD || rank < 0 || rank >= world ||
      stride <= 0 || (stride & 1))
    return caskhip::report_failure(CASK_HIP_ERR_INVALID, "bad argument (stride must be even, world <= 64)");
  cask_hip_push *p = new cask_hip_push;
  p->rank = rank;
  p->world = world;
  p->stride = stride;
  for (int g = 0; g < world; g++) {
    p->tables.full[0][g] = reinterpret_cast<double *>(full_addr[g]);
    p->tables.full[1][g] = reinterpret_cast<double *>(full_addr[world + g]);
    p->tables.flags[g] = reinterpret_cast<int *>(flag_addr[g]);
    if (!full_addr[g] || !full_addr[world + g] || !flag_addr[g] || (full_addr[g] & 15) || (full_addr[world + g] & 15)) {
      delete p;
      return caskhip::report_failure(CASK_HIP_ERR_INVALID, "NULL or misaligned peer mapping");
    }
  }
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->d_state), 4 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(p->d_state, 0, 4 * sizeof(int));
  if (e != hipSuccess) {
    delete p;
    return caskhip::report_failure(CASK_HIP_ERR_RUNTIME, std::string("push state: ") + hipGetErrorString(e));
  }
  *out = p;
  return CASK_HIP_OK;
}

int cask_hip_push_destroy(cask_hip_push *p) {
  if (!p) return CASK_HIP_OK;
  if (p->d_state) (void)hipFree(p->d_state);
  delete p;
  return CASK_HIP_OK;
}

int cask_hip_push_allgather(cask_hip_push *p, const double *d_local, double **d_full_out, void *stream) {
  if (!p || !d_local || !d_full_out) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (reinterpret_cast<uintptr_t>(d_local) & 15) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "d_local must be 16-byte aligned");
  // stride*8 bytes to each of world destinations (1 MB each at webbase-1M / 8), four 16-byte pairs per lane in flight
  const int parity = p->parity;
  // one rank whose slice already sits in place: nothing to move (the launch still advances the sequence number)
  const bool nothing = p->world == 1 && d_local == p->tables.full[parity][0];
  const int64_t pairs = nothing ? 0 : p->stride >> 1;
  const int grid = (int)std::min<int64_t>(128, std::max<int64_t>(1, (pairs + 1023) / 1024));
  hipLaunchKernelGGL(k_push_allgather, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), d_local, 2 * pairs, p->rank,
                     p->world, p->tables, p->d_state, parity);
  P2P_TRY(hipGetLastError());
  *d_full_out = p->tables.full[parity][p->rank];
  p->parity ^= 1;
  return CASK_HIP_OK;
}

int cask_hip_push_allreduce(double *d_values, int32_t count, void *stream, void *push) {
  cask_hip_push *p = static_cast<cask_hip_push *>(push);
  if (!p || !d_values || count < 1 || count > PUSH_SCALARS)
    return caskhip::report_failure(CASK_HIP_ERR_INVALID, "push all-reduce takes 1 to 4 doubles");
  hipLaunchKernelGGL(k_push_allreduce, dim3(1), dim3(4 * CASK_HIP_PUSH_MAX_WORLD), 0, static_cast<hipStream_t>(stream), d_values, (int)count, p->rank,
                     p->world, p->tables, p->d_state);
  P2P_TRY(hipGetLastError());
  return CASK_HIP_OK;
}

int cask_hip_push_own_slot(cask_hip_push *p, double **d_slot_out) {
  if (!p || !d_slot_out) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  *d_slot_out = p->tables.full[p->parity][p->rank] + (int64_t)p->rank * p->stride;
  return CASK_HIP_OK;
}

// between cask_hip.hip (the solver) and this file: the fused form of k_sum_to_scalars + cask_hip_push_allreduce
int cask_hip_push_sum_allreduce(const double *d_pa, int na, const double *d_pb, int nb, double *d_out, const int *d_done,
                                void *stream, void *push) {
  cask_hip_push *p = static_cast<cask_hip_push *>(push);
  if (!p || !d_pa || !d_out) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  hipLaunchKernelGGL(k_push_sum_allreduce, dim3(1), dim3(4 * CASK_HIP_PUSH_MAX_WORLD), 0, static_cast<hipStream_t>(stream), d_pa,
                     na, d_pb, nb, d_out, d_done, p->rank, p->world, p->tables, p->d_state);
  P2P_TRY(hipGetLastError());
  return CASK_HIP_OK;
}

int cask_hip_push_check(cask_hip_push *p) {
  if (!p) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  int st[4] = {0, 0, 0, 0};
  P2P_TRY(hipMemcpy(st, p->d_state, sizeof(st), hipMemcpyDeviceToHost));
  if (st[2]) return caskhip::report_failure(CASK_HIP_ERR_RUNTIME, "push all-gather: a peer's flag did not arrive within the poll limit");
  return CASK_HIP_OK;
}

int cask_hip_shared_alloc(int64_t bytes, void **d_ptr_out, unsigned char *handle_out) {
  if (bytes <= 0 || !d_ptr_out || !handle_out) return caskhip::report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  void *p = nullptr;
  // FINE-GRAINED device memory: peers store into this region and its owner polls flags in it INSIDE a kernel.  Ordinary
  // (coarse-grained) memory is coherent across devices at kernel boundaries only -- a polled flag line may sit stale in
  // the owner's L2 -- whereas fine-grained memory is coherent at system scope.  (CASK_HIP_SHARED_COARSE=1: plain hipMalloc,
  // for comparison on one GPU.)
  static const bool coarse = std::getenv("CASK_HIP_SHARED_COARSE") != nullptr;
  if (coarse) P2P_TRY(hipMalloc(&p, (size_t)bytes));
  else P2P_TRY(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained));
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
