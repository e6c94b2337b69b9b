// The lane-group walk (k_trsv_lanes, round 5): runs of narrow levels whose rows are LONG (an FEM factor: the cant-like
// ILU(0) factors have 32 entries a row in levels of ~7 rows).
//
// k_trsv_walk2 gives a row to ONE lane: 7 of the walker's 64 lanes then chain 32 dependent multiply-adds and 32 LDS
// reads per level -- ~1 us a level, 16.8 ms per ILU(0) application on the cant-like factors, where one CPU core running
// the reference's mkl_dcsrtrsv needs 2.0 ms (profiles/r05_trsv.txt).  Here a row gets a GROUP of G = 1 .. 64 lanes
// (a power of two; E = 4, 8 or 16 entries per lane, one E per run of levels), the group's partial sums are added by
// log2 G DPP steps, and the group's last lane subtracts from the right-hand side, divides and writes the ring:
//   * SLABS.  The host cuts every level into slabs of consecutive rows whose lane groups -- 2^lg lanes for a row of up
//     to E 2^lg entries, a size per row, widest first so that every group starts on a multiple of its size -- fit 64
//     lanes, and writes, per slab and lane, E values and E LDS byte addresses (the ring slots of their x values; an
//     absent entry is value +0.0 at the address of a constant 0.0) -- stored 16-byte unit by unit across the lanes, so
//     that the wave's 16-byte reads are consecutive -- and a word: where the lane's result goes (the row's ring slot in
//     a group's last lane, a dump slot elsewhere), its row, its group's log2 size.  32 / E slabs are a chunk: 24 KB of
//     records, the lane words, and the right-hand sides and reciprocal diagonals of its <= 512 positions.  The stream
//     is the LDS image itself.  E is the one of the three that makes the run cheapest (fewest slabs: a level is a
//     dependent step whatever its width; at E = 8 a cant-like level -- 7 rows of 25-45 entries -- is ONE slab).
//   * STAGERS (waves 1-9): three groups of three waves.  A group copies its chunk into the other half of a double
//     buffer while the walker is on the chunk before it, requests its next chunk (three further on) in the following
//     phase -- three chunks are in flight, each with two phases to land: ONE chunk in flight streamed 26 GB/s, a
//     load's latency, and was the limit -- and writes its chunk's results from the ring to memory in the phase after
//     the walk (position space: the gather / scatter kernels of walk2 surround the solve).  Every phase sees one group
//     staging, one requesting, one writing back.  One barrier per chunk; a chunk's header carries its own span of
//     positions and that of the chunk three behind it, so no address waits for a load of the same phase.
//   * WALKER (wave 0): per slab E x reads, E multiply-adds in two or four chains, <= 6 DPP steps (an inclusive scan
//     whose total lands in a group's last lane; a lane of a narrower group multiplies what crosses its group's
//     boundary by 0), a multiplication by the reciprocal diagonal, the ring write -- LDS operations of one wave execute
//     in order, so a level's reads go out right behind the previous level's write, without a barrier.  The records of
//     the slabs ahead are requested in the shadow of that chain (behind the x reads: the compiler is kept from hoisting
//     them).  It issues no scalar or vector memory operation and computes no address: ~65 instructions a slab, ~480
//     cycles (LDS round trip ~130, three DPP steps ~120, the x reads ~90: profiles/r05_trsv.txt).
//   * READ-AHEAD (workgroup 8 of 9: the same XCD, the same L2) touches the lines of the chunks ten ahead, as in walk2.
// A run of levels qualifies if no row has more than 1 024 entries and every source is inside the LDS ring (the producer
// within LN_RING positions of the consumer's level end, in the same run); anything else stays with walk2.
//
// Arithmetic: a row's products are added lane group by lane group, not in stored order -- the result differs from the
// other schedules' in the last bits, and the division is a multiplication by the reciprocal the host rounded (within an
// ulp of the quotient) -- same plan, same bits: reproducible run to run; tests compare it with the other schedules and
// with the oracle under a tolerance.  Reference: the solves of ILUPreconditioner::apply
// (src/runtime/SparseLinearSolvers.hpp:143-151, MklLayer.hpp:29-85).
#pragma once
#include <hip/hip_runtime.h>

#include "trsv_lanes_plan.hpp"

namespace caskhip_lanes {

typedef double ln_dbl2 __attribute__((ext_vector_type(2)));
typedef int ln_int4 __attribute__((ext_vector_type(4)));

struct LanesTri {
  const char *lanes;                   // [chunks][LN_CHUNK_BYTES]
  const int *hdr;                      // [chunks][LN_HDR_INTS]
  const double *diag;                  // position space: the RECIPROCALS of the diagonal entries
};

__device__ __forceinline__ void ln_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);                         // vmcnt 63, expcnt 7, lgkmcnt 0 (the compiler's bookkeeping sees it)
  __builtin_amdgcn_s_barrier();
}
// a as the DPP control moves it (lanes without a source and rows outside ROWMASK: +0.0)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double ln_dpp_moved(double a) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), CTRL, ROWMASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), CTRL, ROWMASK, 0xf, true);
  return __hiloint2double(hi, lo);
}

// LDS by byte address.  The kernel has no static LDS, so its dynamic LDS starts at address 0 and the addresses the host
// wrote into the stream are used as they stand (adding the array's base would be a VALU instruction per access, and the
// walker is bound by the instructions it issues).
#define LN_AS3 __attribute__((address_space(3)))
template <typename T>
__device__ __forceinline__ T ln_ld(int addr) { return *(const LN_AS3 T *)(uintptr_t)(unsigned)addr; }
template <typename T>
__device__ __forceinline__ void ln_st(int addr, T v) { *(LN_AS3 T *)(uintptr_t)(unsigned)addr = v; }

// One chunk by the walker wave: its slabs back to back, fully unrolled (straight-line code: the compiler counts the LDS
// operations in flight).  base: the LDS byte address of the chunk's buffer.
template <bool UNIT, int E>
__device__ __forceinline__ void ln_walk_chunk(int base, int lane) {
  constexpr int C = ln_slabs_per_chunk(E), LANE_BYTES = 12 * E, SLAB_BYTES = 64 * LANE_BYTES;
  constexpr int AHEAD = C >= 8 ? 3 : C >= 4 ? 2 : 1;          // slabs whose records are held ahead of the one being solved
  constexpr int NCH = E >= 8 ? 4 : 2;                         // independent multiply-add chains per lane
  ln_dbl2 v[C][E / 2];
  ln_int4 a[C][E / 4];
  int tw[C];
  // a slab's records are stored 16-byte unit by unit (unit q of all 64 lanes, then unit q + 1): a wave's 16-byte reads
  // are then consecutive (lane by lane at a stride of 12 E bytes they were 4- to 8-way bank conflicts -- six such reads
  // per slab kept the LDS busy for longer than the whole dependent chain)
  const int rec0 = base + lane * 16, tab0 = base + LN_REC_BYTES + 4 * lane;
  auto preload = [&](int s) {
#pragma unroll
    for (int q = 0; q < E / 2; q++) v[s][q] = ln_ld<ln_dbl2>(rec0 + s * SLAB_BYTES + 1024 * q);
#pragma unroll
    for (int q = 0; q < E / 4; q++) a[s][q] = ln_ld<ln_int4>(rec0 + s * SLAB_BYTES + 1024 * (E / 2 + q));
    tw[s] = ln_ld<int>(tab0 + s * LN_TAB_BYTES);
  };
#pragma unroll
  for (int s = 0; s < AHEAD; s++) preload(s);
#pragma unroll
  for (int s = 0; s < C; s++) {
    double x[E];
#pragma unroll
    for (int t = 0; t < E; t++) x[t] = ln_ld<double>(a[s][t >> 2][t & 3]);
    const int brow = base + LN_OFF_B + ((tw[s] >> 14) & 0x1ff8);
    const double rb = ln_ld<double>(brow);
    double rd = 1.0;
    if constexpr (!UNIT) rd = ln_ld<double>(brow + (LN_OFF_D - LN_OFF_B));
    const int lg = (__builtin_amdgcn_readfirstlane(tw[s]) >> 27) & 7, lgl = (tw[s] >> 27) & 7;
    asm volatile("" ::: "memory");                            // the x reads go out first: the chain below waits for them
    if (s + AHEAD < C) preload(s + AHEAD);
    double ch[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) ch[c] = v[s][c >> 1][c & 1] * x[c];
#pragma unroll
    for (int t = NCH; t < E; t++) ch[t % NCH] = fma(v[s][t >> 1][t & 1], x[t], ch[t % NCH]);
    double acc = NCH == 4 ? (ch[0] + ch[1]) + (ch[2] + ch[3]) : ch[0] + ch[1];
    // A group's sum ends up in its LAST lane: an inclusive scan over windows of 2, 4, .. lanes.  Groups start on multiples
    // of their sizes, widest first: a step of 2^(k-1) lanes reaches across a group boundary only in lanes whose own group
    // is narrower than 2^k, and those must ignore what came across.  r6: by a SELECT on the moved value (the comparison of
    // the lane word's log2 G with k is made while the x values are on their way) -- round 5 multiplied it by a 0.0 / 1.0
    // factor, and 0 x Inf = NaN put one overflowing row's NaN into slab neighbours that depend on nothing bad, where
    // mkl_dcsrtrsv (MklLayer.hpp:29-85) and the other schedules contaminate true dependents only
    // (tests/test_nonfinite_gpu.py; the A/B: +0.4 %, inside the noise -- profiles/r06_lanes_mask.txt).
    // lg: the widest group's log2 (lane 0's), a scalar.
    auto step = [&](int k, double moved) { acc += lgl > k ? moved : 0.0; };
    if (lg >= 1) {
      step(0, ln_dpp_moved<0x111, 0xf>(acc));                 // row_shr:1
      if (lg >= 2) {
        step(1, ln_dpp_moved<0x112, 0xf>(acc));               // row_shr:2
        if (lg >= 3) {
          step(2, ln_dpp_moved<0x114, 0xf>(acc));             // row_shr:4
          if (lg >= 4) {
            step(3, ln_dpp_moved<0x118, 0xf>(acc));           // row_shr:8
            if (lg >= 5) {
              step(4, ln_dpp_moved<0x142, 0xa>(acc));         // row_bcast15 -> rows 1, 3
              if (lg >= 6) acc += ln_dpp_moved<0x143, 0xc>(acc);               // row_bcast31 -> rows 2, 3 (one group of 64)
            }
          }
        }
      }
    }
    double xn = rb - acc;
    if constexpr (!UNIT) xn = xn * rd;                        // rd = 1 / diagonal (host): a multiplication instead of ~15 instructions in the chain
    ln_st<double>(tw[s] & 0x1ffff, xn);
  }
}

template <bool UNIT, int E>
__global__ void __launch_bounds__(LN_T)
k_trsv_lanes(LanesTri t, int c0, int c1, const double *__restrict__ bp, double *xp, int *progress,
             unsigned long long *dbg) {
  constexpr int C = ln_slabs_per_chunk(E), UNITS = (LN_REC_BYTES + C * LN_TAB_BYTES) / 16;   // what a chunk of this E holds
  if (blockIdx.x != 0) {
    if (blockIdx.x != 8) return;
    // read-ahead: one word per 128-byte line of the chunks the stagers will ask for, paced by the progress word.
    // Nothing depends on it: if it falls behind or lands on another XCD the solve is only slower.
    constexpr int FETCH_AHEAD = 10;
    unsigned acc = 0;
    const int ft = threadIdx.x;
    for (int f = c0 + 3; f < c1; f++) {
      for (int polls = 0; polls < (1 << 14); polls++) {       // bounded
        if (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + FETCH_AHEAD >= f) break;
        __builtin_amdgcn_s_sleep(20);
      }
      const int p0 = t.hdr[LN_HDR_INTS * f + 8], rows = t.hdr[LN_HDR_INTS * f + 9];
      for (int u = ft; u < UNITS / 8 + 64; u += LN_T) {
        if (u < UNITS / 8)
          acc += *reinterpret_cast<const unsigned *>(t.lanes + (size_t)f * LN_CHUNK_BYTES + 128 * u);
        else {
          const int q = u - UNITS / 8;                        // 32 lines of b, 32 of the diagonal (LN_ROWS * 8 / 128)
          const long a = ((8L * p0) & ~127L) + 128L * (q & 31);
          if (a < 8L * (p0 + rows)) {
            if (q < 32) acc += *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(bp) + a);
            else if (!UNIT) acc += *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(t.diag) + a);
          }
        }
      }
    }
    if (acc == 0x9e3779b9u) *progress = -1;                   // (keeps the loads alive)
    return;
  }
  extern __shared__ double ln_lds[];
  char *lds = reinterpret_cast<char *>(ln_lds);
  double *ring = ln_lds;
  // The host wrote ABSOLUTE LDS byte addresses into the stream (ring at 0, LN_ZERO, LN_BUF0): that holds only while this
  // kernel's dynamic LDS starts at address 0 -- no static __shared__ anywhere in it, no runtime reservation.  Checked on
  // every launch (a scalar compare): a build or runtime that moves the base gets progress = LN_BAD_LDS_BASE and no solve
  // (the host side reads the word after the first application of a factor and falls back to walk2: cask_hip_precond.hip).
  if ((unsigned)(uintptr_t)(LN_AS3 char *)lds != 0u) {
    if (threadIdx.x == 0) __hip_atomic_store(progress, LN_BAD_LDS_BASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (c0 >= c1) return;                                       // (the host's one-time probe of the check above: no chunks)
  const int tid = threadIdx.x, lane = tid & 63, sg = tid - 64;
  auto buf_base = [&](int which) { return LN_BUF0 + which * LN_BUF_BYTES; };

  if (tid < 64) {
    // ------------------------------------------------------------------------------------------------ the walker
    unsigned long long d_wait = 0, d_walk = 0;
    for (int k = c0; k < c1; k++) {
      const unsigned long long q0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
      ln_barrier();                                           // chunk k is staged
      const unsigned long long q1 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
      ln_walk_chunk<UNIT, E>(buf_base((k - c0) & 1), lane);
      if (dbg) {                                              // CASK_HIP_TRSV_STATS: cycles at the barrier / walking
        __builtin_amdgcn_s_waitcnt(0xC07F);
        d_wait += q1 - q0;
        d_walk += __builtin_amdgcn_s_memtime() - q1;
      }
    }
    if (dbg && lane == 0) { atomicAdd(dbg + 0, d_wait); atomicAdd(dbg + 1, d_walk); atomicAdd(dbg + 3, (unsigned long long)(c1 - c0)); }
    ln_barrier();                                             // the last chunk's x values are in the ring
    return;
  }

  // -------------------------------------------------------------------------------------------------- the stagers
  // LN_NG = 3 groups of three waves: chunk r (counted from c0) belongs to group r % 3, which stages it in phase r - 1,
  // requests its next chunk (r + 3) in phase r -- two phases to land; three chunks are in flight (one CU streams ~26 GB/s
  // with one chunk in flight: a load's latency, not the memory, was the limit) -- and writes chunk r's results back in
  // phase r + 1, when the walker is through with it.  So every phase sees one group staging (~800 cycles), one requesting
  // (~600) and one writing back (~300) side by side; all three in one group's turn (~1 800) was as long as the walker's
  // phase.  hq: unit (st & 3) of a chunk's header: lane 2 of every wave holds {first position, positions, first position
  // and positions of the chunk three behind}.
  if (tid == 64) {
    *reinterpret_cast<double *>(lds + LN_ZERO) = 0.0;
    *reinterpret_cast<double *>(lds + LN_DUMP) = 0.0;
  }
  const int grp = __builtin_amdgcn_readfirstlane(sg / LN_ST), st = sg % LN_ST;
  ln_int4 u[LN_UJ], hq;
  double lb[LN_RJ], ld[LN_RJ];
  const ln_int4 *lanes4 = reinterpret_cast<const ln_int4 *>(t.lanes);
  const ln_int4 *hdr4 = reinterpret_cast<const ln_int4 *>(t.hdr);
  auto load = [&](int k, int p0, int rows) {                  // chunk k (clamped: loads past the end repeat the last chunk)
    const int kk = min(k, c1 - 1);
#pragma unroll
    for (int j = 0; j < LN_UJ; j++) u[j] = lanes4[(size_t)kk * LN_UNITS + min(st + LN_ST * j, UNITS - 1)];
#pragma unroll
    for (int j = 0; j < LN_RJ; j++) {
      const int p = p0 + min(st + LN_ST * j, max(rows - 1, 0));
      lb[j] = bp[p];
      if constexpr (!UNIT) ld[j] = t.diag[p];
    }
    hq = hdr4[(size_t)kk * (LN_HDR_INTS / 4) + (st & 3)];
  };
  auto stage = [&](int which, int rows) {                     // registers -> buffer `which`
    char *buf = lds + buf_base(which);
#pragma unroll
    for (int j = 0; j < LN_UJ; j++)
      if (st + LN_ST * j < UNITS) *reinterpret_cast<ln_int4 *>(buf + 16 * (st + LN_ST * j)) = u[j];
#pragma unroll
    for (int j = 0; j < LN_RJ; j++) {
      const int r = st + LN_ST * j;
      if (r < rows) {
        *reinterpret_cast<double *>(buf + LN_OFF_B + 8 * r) = lb[j];
        if constexpr (!UNIT) *reinterpret_cast<double *>(buf + LN_OFF_D + 8 * r) = ld[j];
      }
    }
  };
  auto write_back = [&](int p0, int rows) {                   // ring -> xp
    if (rows > 0) {
#pragma unroll
      for (int j = 0; j < LN_RJ; j++) {
        const int p = p0 + min(st + LN_ST * j, rows - 1);
        xp[p] = ring[p & (LN_RING - 1)];
      }
    }
  };
  int pP = 0, rP = 0;                                         // the chunk this group staged last: written back at its next turn
  {
    const int kk = min(c0 + grp, c1 - 1);                     // the group's first chunk (scalar loads: once)
    const int p0 = __builtin_amdgcn_readfirstlane(t.hdr[LN_HDR_INTS * kk + 8]);
    const int rows = __builtin_amdgcn_readfirstlane(t.hdr[LN_HDR_INTS * kk + 9]);
    load(c0 + grp, p0, rows);
    if (grp == 0) {
      stage(0, rows);                                         // chunk c0 (waits for its loads); chunk c0 + 3 is requested in phase c0
      pP = p0; rP = rows;
    }
  }
  unsigned long long e_wait = 0, e_work = 0, e_a = 0, e_b = 0, e_c = 0, e_d = 0;   // (per turn: loads awaited, write-back, staging, requests)
  for (int k = c0; k < c1; k++) {
    const unsigned long long q0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    ln_barrier();                                             // phase k: the walker is on chunk k
    const unsigned long long q1 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    // A group's three jobs take three phases, so that a phase sees one group staging, one requesting, one writing back:
    // what it does in phase k is d = (k + 1 - c0 - its number) mod 3.
    const int d = (k + 1 - c0 + 2 * grp) % LN_NG;             // (+ 2 grp = - grp mod 3)
    if (d == 0) {                                             // chunk k + 1 is in its registers: stage it
      const int rN = __builtin_amdgcn_readlane(hq.y, 2);
      const unsigned long long ta = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
      if (k + 1 < c1) {
        stage((k + 1 - c0) & 1, rN);
        pP = __builtin_amdgcn_readlane(hq.x, 2);
        rP = rN;
      }
      if (dbg) { __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long tb = __builtin_amdgcn_s_memtime(); e_a += ta - q1; e_c += tb - ta; }
    } else if (d == 1) {                                      // request the chunk of its next turn (k + 3): two phases to land
      const int pn = __builtin_amdgcn_readlane(hq.z, 2), rn = __builtin_amdgcn_readlane(hq.w, 2);   // (the header staged last phase)
      load(k + LN_NG, pn, rn);
      // the progress word (paces the read-ahead workgroup) behind the loads: a store is waited for with the loads at the
      // wave's next vmcnt(0) -- stored at the top of a staging phase it held the wave, and with it the barrier, for the
      // ~1 us a write-through store takes
      if (st == 0) __hip_atomic_store(progress, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dbg) e_d += __builtin_amdgcn_s_memtime() - q1;
    } else {                                                  // the chunk it staged two phases ago has been walked: results to memory
      write_back(pP, rP);
      rP = 0;
      if (dbg) { __builtin_amdgcn_s_waitcnt(0xC07F); e_b += __builtin_amdgcn_s_memtime() - q1; }
    }
    if (dbg) { __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long q2 = __builtin_amdgcn_s_memtime(); e_wait += q1 - q0; e_work += q2 - q1; }
  }
  if (dbg && sg == 0) { atomicAdd(dbg + 4, e_wait); atomicAdd(dbg + 6, e_work); atomicAdd(dbg + 8, e_a); atomicAdd(dbg + 11, e_b); atomicAdd(dbg + 9, e_c); atomicAdd(dbg + 10, e_d); }
  ln_barrier();                                               // the last chunk is walked
  write_back(pP, rP);
}

}  // namespace caskhip_lanes
