// Preconditioners of the CASK solver surface on the GPU (include/cask_hip.h, "preconditioning"):
// Jacobi and ILU(0) with level-scheduled triangular solves, and the stand-alone triangular solve
// behind cask::mkl::unittrsolve.
//
// Reference: ILUPreconditioner (src/runtime/SparseLinearSolvers.hpp:77-156) -- an IKJ incomplete
// factorisation on the pattern of the matrix, L and U extracted WITH the diagonal
// (DokMatrix::getLowerTriangular/getUpperTriangular, SparseMatrix.hpp:227-253) and applied with two
// mkl_dcsrtrsv calls that divide by the stored diagonal (MklLayer.hpp:29-85, diag = 'N') -- so the
// "L" solve divides by U's diagonal as well; that is what the known answers of
// test/LinearSolvers.cpp:116-146 pin and what is reproduced here.
//
// The factorisation is a setup step and runs on the host (as in the reference: its constructor is
// sequential CPU code); every application runs on the device.  A triangular solve is a chain of
// dependency levels: rows of one level are independent.  Wide levels (>= WIDE_LEVEL rows) get a launch
// of their own; runs of narrow levels are walked by ONE workgroup with a barrier between levels, out of
// a level-ordered copy of the factor staged through LDS a chunk ahead (k_trsv_packed, round 2).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "cask_hip.h"
#include "internal.hpp"
#include "spmv_common.hpp"
#include "trsv_lanes.hpp"

using caskhip::DevBuf;
using caskhip::report_failure;

namespace {

#define PC_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return report_failure(CASK_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int TRSV_WG = 1024;          // the workgroup that walks runs of narrow levels (row-indexed fallback kernel)
constexpr int WIDE_LEVEL = 256;        // rows from which a level gets a launch of its own (the packed walk gives a
                                       // level to ONE wave: beyond a few strides of 64 rows a launch is cheaper)
constexpr int WIDE_LEVEL_ROWWALK = 16384;  // the same for the row-indexed walk of round 1 (1024 threads per level)

// ---- the packed walk (k_trsv_packed): runs of narrow levels by ONE workgroup, everything but x off the critical path
constexpr int PK_T = 256;              // threads
constexpr int PK_CH = 1024;            // positions (rows in level order) staged per chunk
constexpr int PK_ECAP = 3072;          // off-diagonal entries staged per chunk
constexpr int PK_RING = 4096;          // x values of the most recent positions kept in LDS; >= 3 PK_CH + WIDE_LEVEL
constexpr int PK_PJ = PK_CH / PK_T, PK_EJ = PK_ECAP / PK_T;
constexpr unsigned PK_NEAR = 0x80000000u;
static_assert(PK_RING >= 3 * PK_CH + WIDE_LEVEL && (PK_RING & (PK_RING - 1)) == 0, "ring covers what staging cannot prefetch");
constexpr size_t PK_LDS_BYTES = sizeof(double) * (PK_RING + 2 * PK_ECAP + 2 * PK_CH) +
                                sizeof(int) * (PK_ECAP + (PK_CH + 1) + (PK_CH + 1));

// x[r] = (b[r] - sum_{c != r, c in the triangle} val * x[c]) / val[r][r], entries in stored order
// flags: bit 0 = lower triangle, bit 1 = unit diagonal (the stored diagonal is ignored)
__device__ __forceinline__ void solve_row(int r, int flags, const int *__restrict__ rp, const int *__restrict__ ci,
                                          const double *__restrict__ val, const double *__restrict__ b, double *x) {
  const bool lower = flags & 1, unit = flags & 2;
  double s = b[r], diag = 0.0;
  for (int k = rp[r]; k < rp[r + 1]; k++) {
    const int c = ci[k];
    if (c == r) diag = val[k];
    else if (lower ? c < r : c > r) s -= val[k] * x[c];
  }
  x[r] = unit ? s : s / diag;
}

// one wide level: a thread per row
__global__ void k_trsv_level(int lo, int hi, int lower, const int *__restrict__ order, const int *__restrict__ rp,
                             const int *__restrict__ ci, const double *__restrict__ val,
                             const double *__restrict__ b, double *x) {
  const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hi) solve_row(order[i], lower, rp, ci, val, b, x);
}

// levels [l0, l1) by one workgroup: the barrier orders a level's stores before the next level's loads
// (all waves of a workgroup share the CU's L1)
__global__ void k_trsv_levels(int l0, int l1, int lower, const int *__restrict__ level_ptr,
                              const int *__restrict__ order, const int *__restrict__ rp, const int *__restrict__ ci,
                              const double *__restrict__ val, const double *__restrict__ b, double *x) {
  for (int l = l0; l < l1; l++) {
    const int lo = level_ptr[l], hi = level_ptr[l + 1];
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) solve_row(order[i], lower, rp, ci, val, b, x);
    __syncthreads();
  }
}

// ---- packed walk of a run of narrow levels -------------------------------------------------------------------------
// k_trsv_levels pays, per level, a chain of dependent global loads (level_ptr -> order -> row_ptr -> entry -> x[c]):
// ~2 us a level, 57 436 levels on the G3_circuit-like factors.  Only x[c] really depends on the previous level.  Here
// the factor is stored a second time in LEVEL ORDER ("positions"; diagonal split off, the other entries of a row in
// their stored order, so the arithmetic is solve_row's bit for bit) and cut into chunks of <= PK_CH positions /
// PK_ECAP entries.  One workgroup walks the chunks: while chunk k is solved out of LDS, the registers of the 256
// threads already hold chunk k+1 (rows, diagonals, b[row], entries) -- streaming loads issued a whole chunk ahead.
// A dependency x[c] comes from one of two places, decided when the factor is built:
//   near   the producer is among the last PK_RING positions of this launch: every solved x is also kept in an LDS
//          ring indexed by position, so the value is one LDS read away;
//   early  the producer belongs to a chunk at least three before the consumer's: x[c] is loaded from memory with the
//          rest of the chunk, a chunk ahead (PK_RING >= 3 PK_CH + the widest level makes the two cases cover everything).
// Per level the dependent work is LDS reads, the row's FMAs, an LDS write and a barrier that waits for LDS only (the
// prefetch stays in flight across it; x goes to memory once per chunk, out of the ring).
// Measured, one ILU(0) application (two solves), MI355X, against the row-indexed walk (profiles/r02_trsv.txt):
// G3_circuit-like (57 436 levels of ~28 rows) 32.2 vs 208 ms, atmosmodd-like (322 levels, most of them wide: launches)
// 3.3 vs 38.8 ms, cant-like (8 548 levels of ~7 rows x 32 entries) 31.7 vs 177 ms.  Still ~0.21 us per level on the
// first: the waves' turns and preparation steps largely cost what they cost one after the other (ablations in the same
// file), and a chunk costs ~3.5 us of staging on top.
struct PackedTri {
  const int *row, *eptr, *seg;
  const int4 *hdr;                     // per chunk: {first position, positions, first entry, entries}, {first segment, segments, -, -}
  const unsigned *code;
  const double *diag, *val;
};

// (the wait as a builtin, not asm: the compiler's wait-count bookkeeping sees it -- see w2_barrier)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);                         // vmcnt 63, expcnt 7, lgkmcnt 0
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LONG: the rows of the step have many entries (an FEM factor): the entries beyond the prepared ones are read PK_U at a
// time.  Short-row steps keep the one-by-one loop -- the unrolled one costs them 10 % in code they never run.
template <bool LONG>
__global__ void __launch_bounds__(PK_T)
k_trsv_packed(PackedTri t, int c0, int c1, int unit, const double *__restrict__ b, double *x) {
  extern __shared__ double pk_lds[];
  double *ring = pk_lds;
  double *s_val = ring + PK_RING, *s_x = s_val + PK_ECAP, *s_b = s_x + PK_ECAP, *s_diag = s_b + PK_CH;
  int *s_src = reinterpret_cast<int *>(s_diag + PK_CH);       // per entry: where its x value is (byte offset in LDS)
  int *s_e = s_src + PK_ECAP;                                 // PK_CH + 1 entry offsets, chunk-relative
  int *s_seg = s_e + PK_CH + 1;                               // segment ends, chunk-relative
  const char *lds_bytes = reinterpret_cast<const char *>(pk_lds);
  const int tid = threadIdx.x;

  // the chunk in flight (position idx = tid + PK_T * j, entry tid + PK_T * j)
  int r_row[PK_PJ], r_e[PK_PJ + 1], r_seg[PK_PJ + 1];
  double r_diag[PK_PJ], r_b[PK_PJ], r_val[PK_EJ], r_x[PK_EJ];
  unsigned r_code[PK_EJ];
  int4 h0, h1;                                                // header of the chunk after the one in the registers
  int w_row[PK_PJ];                                           // rows of the chunk being solved (for writing x back)

  // Loads are unconditional (indices clamped into the chunk; the arrays carry one spare element) so that all of a
  // chunk's streaming loads are in flight together, then the gathers that need a loaded index.  The chunk's header
  // arrives with the chunk before it, so nothing here waits for a scalar.
  auto load_chunk = [&](int cs, int cnt, int ebase, int ecnt, int sbase, int scnt) {
#pragma unroll
    for (int j = 0; j < PK_PJ; j++) r_row[j] = t.row[cs + min(tid + PK_T * j, cnt - 1)];
#pragma unroll
    for (int j = 0; j < PK_EJ; j++) r_code[j] = t.code[ebase + min(tid + PK_T * j, max(ecnt - 1, 0))];
#pragma unroll
    for (int j = 0; j < PK_PJ; j++) r_diag[j] = t.diag[cs + min(tid + PK_T * j, cnt - 1)];
#pragma unroll
    for (int j = 0; j < PK_PJ + 1; j++) {
      r_e[j] = t.eptr[cs + min(tid + PK_T * j, cnt)] - ebase;
      r_seg[j] = t.seg[sbase + min(tid + PK_T * j, scnt - 1)] - cs;
    }
#pragma unroll
    for (int j = 0; j < PK_EJ; j++) r_val[j] = t.val[ebase + min(tid + PK_T * j, max(ecnt - 1, 0))];
#pragma unroll
    for (int j = 0; j < PK_PJ; j++) r_b[j] = b[r_row[j]];
#pragma unroll
    for (int j = 0; j < PK_EJ; j++) {
      const bool early = tid + PK_T * j < ecnt && !(r_code[j] & PK_NEAR);
      r_x[j] = __hip_atomic_load(x + (early ? r_code[j] : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto store_chunk = [&](int cnt, int ecnt, int scnt) {
#pragma unroll
    for (int j = 0; j < PK_PJ; j++) {
      const int idx = tid + PK_T * j;
      if (idx < cnt) {
        s_diag[idx] = r_diag[j];
        s_b[idx] = r_b[j];
      }
      w_row[j] = r_row[j];
    }
#pragma unroll
    for (int j = 0; j < PK_PJ + 1; j++) {
      const int idx = tid + PK_T * j;
      if (idx <= cnt) s_e[idx] = r_e[j];
      if (idx < scnt) s_seg[idx] = r_seg[j];
    }
#pragma unroll
    for (int j = 0; j < PK_EJ; j++) {
      const int idx = tid + PK_T * j;
      if (idx < ecnt) {
        // a near value is read from the ring, an early one from s_x: one address either way, no select in the walk
        s_src[idx] = (r_code[j] & PK_NEAR) ? (int)((r_code[j] & (PK_RING - 1)) * 8u)
                                           : (int)((PK_RING + PK_ECAP + idx) * 8);
        s_val[idx] = r_val[j];
        s_x[idx] = r_x[j];
      }
    }
  };

  h0 = t.hdr[2 * c0];
  h1 = t.hdr[2 * c0 + 1];
  int cs = uniform(h0.x), cnt = uniform(h0.y), ebase = uniform(h0.z), ecnt = uniform(h0.w), sbase = uniform(h1.x),
      scnt = uniform(h1.y);
  load_chunk(cs, cnt, ebase, ecnt, sbase, scnt);
  if (c0 + 1 < c1) {
    h0 = t.hdr[2 * c0 + 2];
    h1 = t.hdr[2 * c0 + 3];
  }
  for (int k = c0; k < c1; k++) {
    // Chunk k-1 is finished.  Its x stores (the youngest four memory operations of every thread) may still be on their
    // way: only the stores of chunk k-2 and everything older must have landed before anybody issues chunk k+1's early
    // loads below -- the factor's builder lets a chunk's early loads rely on chunks at least three back.  (Waiting for
    // the write-through stores themselves cost 1.2 us per chunk.)  Both barriers order LDS traffic only.
    static_assert(PK_PJ == 4, "the vmcnt immediate below counts the PK_PJ = 4 x stores per thread of the previous chunk");
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    lds_barrier();
    store_chunk(cnt, ecnt, scnt);
    const int cur_cs = cs, n_pos = cnt, n_seg = scnt;
    lds_barrier();
    if (k + 1 < c1) {
      cs = uniform(h0.x); cnt = uniform(h0.y); ebase = uniform(h0.z); ecnt = uniform(h0.w);
      sbase = uniform(h1.x); scnt = uniform(h1.y);
      load_chunk(cs, cnt, ebase, ecnt, sbase, scnt);
      if (k + 2 < c1) {
        h0 = t.hdr[2 * k + 4];
        h1 = t.hdr[2 * k + 5];
      }
    }
    // The walk touches LDS only (no global access inside: the prefetch above stays in flight across it).  The four
    // waves take the levels in turn (level L belongs to wave L % 4, lane i solves its i-th row).  A level's only
    // dependent work is: x reads, the row's FMAs, the ring write.  Everything else -- the level's bounds, the rows'
    // right-hand sides and entry ranges, the entries themselves (three dependent LDS reads) -- its wave does one
    // step per barrier interval while the other three waves have their turns.
    constexpr int PK_Q = 2;
    constexpr int PK_U = 16;                                  // LONG: entries per pair of dependent LDS reads
    const int wave = tid >> 6, lane = tid & 63;
    int my_level = wave, l_lo = 0, l_hi = 0, idx = 0, e0 = 0, ne = 0;
    double acc = 0.0, diag = 1.0, q_val[PK_Q];
    int q_src[PK_Q];
    auto bounds = [&]() {                                     // step 1
      l_lo = my_level > 0 ? uniform(s_seg[my_level - 1]) : 0;
      l_hi = uniform(s_seg[my_level]);
    };
    auto rows = [&]() {                                       // step 2 (idx: this lane's row of the level)
      ne = 0;
      if (idx < l_hi) {
        acc = s_b[idx];
        diag = s_diag[idx];
        e0 = s_e[idx];
        ne = s_e[idx + 1] - e0;
      }
    };
    auto entries = [&]() {                                    // step 3
      if (idx < l_hi) {
#pragma unroll
        for (int q = 0; q < PK_Q; q++) {
          const int e = e0 + min(q, max(ne - 1, 0));
          q_val[q] = s_val[e];
          q_src[q] = s_src[e];
        }
      }
    };
    auto solve = [&]() {
      if (idx < l_hi) {
        double xv[PK_Q];
#pragma unroll
        for (int q = 0; q < PK_Q; q++) xv[q] = *reinterpret_cast<const double *>(lds_bytes + q_src[q]);
        double s = acc;
#pragma unroll
        for (int q = 0; q < PK_Q; q++)
          if (q < ne) s -= q_val[q] * xv[q];
        if constexpr (!LONG) {
          for (int e = e0 + PK_Q; e < e0 + ne; e++)
            s -= s_val[e] * *reinterpret_cast<const double *>(lds_bytes + s_src[e]);
        } else
        for (int e = e0 + PK_Q; e < e0 + ne; e += PK_U) {     // longer rows: the rest straight from LDS, PK_U entries
          double v4[PK_U], x4[PK_U];                          // per pair of dependent reads, still one FMA after the other
          int a4[PK_U];
#pragma unroll
          for (int u = 0; u < PK_U; u++) {
            const int eu = min(e + u, e0 + ne - 1);
            a4[u] = s_src[eu];
            v4[u] = s_val[eu];
          }
#pragma unroll
          for (int u = 0; u < PK_U; u++) x4[u] = *reinterpret_cast<const double *>(lds_bytes + a4[u]);
#pragma unroll
          for (int u = 0; u < PK_U; u++)
            if (e + u < e0 + ne) s -= v4[u] * x4[u];
        }
        ring[(cur_cs + idx) & (PK_RING - 1)] = unit ? s : s / diag;
      }
    };
    if (my_level < n_seg) {                                   // the first level of every wave: all three steps now
      bounds();
      idx = l_lo + lane;
      rows();
      entries();
    }
    // One barrier interval of this wave; `turn` is a compile-time constant: every wave runs its own unrolled copy of
    // the loop (four intervals per trip, no dispatch on the turn inside it).
    auto interval = [&](auto turn_c) {
      constexpr int turn = decltype(turn_c)::value;
      if constexpr (turn == 0) {
        solve();
        for (idx += 64; idx < l_hi; idx += 64) {              // a level wider than a wave: the other rows one by one
          rows();
          entries();
          solve();
        }
        my_level += 4;
      } else if (my_level < n_seg) {
        if constexpr (turn == 1) {
          bounds();
          idx = l_lo + lane;
        } else if constexpr (turn == 2) {
          rows();
        } else {
          entries();
        }
      }
      lds_barrier();                                          // the level's x values are in the ring
    };
    auto walk = [&](auto first_c) {                           // first: this wave's turn in interval 0
      constexpr int f = decltype(first_c)::value;
      int sg = 0;
      for (; sg + 4 <= n_seg; sg += 4) {
        interval(std::integral_constant<int, f>{});
        interval(std::integral_constant<int, (f + 1) & 3>{});
        interval(std::integral_constant<int, (f + 2) & 3>{});
        interval(std::integral_constant<int, (f + 3) & 3>{});
      }
      if (sg < n_seg) interval(std::integral_constant<int, f>{});
      if (sg + 1 < n_seg) interval(std::integral_constant<int, (f + 1) & 3>{});
      if (sg + 2 < n_seg) interval(std::integral_constant<int, (f + 2) & 3>{});
    };
    switch (wave) {                                           // turn = (interval - wave) & 3
      case 0: walk(std::integral_constant<int, 0>{}); break;
      case 1: walk(std::integral_constant<int, 3>{}); break;
      case 2: walk(std::integral_constant<int, 2>{}); break;
      default: walk(std::integral_constant<int, 1>{}); break;
    }
    // the chunk's results: ring -> x (PK_CH <= PK_RING: all of them are still there)
#pragma unroll
    for (int j = 0; j < PK_PJ; j++) {                         // always four stores per thread (see the wait above): lanes past
      const int idx = min(tid + PK_T * j, n_pos - 1);         // the end repeat the chunk's last row (w_row is clamped alike)
      __hip_atomic_store(x + w_row[j], ring[(cur_cs + idx) & (PK_RING - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

typedef double pk_dbl2 __attribute__((ext_vector_type(2)));
typedef int pk_int4 __attribute__((ext_vector_type(4)));

// ---- the walk of narrow levels, round 3 (k_trsv_walk2, the default) -------------------------------------------------
// A first form with ONE walker wave that issued a level's bookkeeping itself while the other waves idled lost to the
// four-wave packed walk (36.5 vs 32.2 ms, docs/experiments.md; removed in round 5).  Here the work is split by kind, and everything that does not depend on x leaves the dependent chain:
//   * POSITION SPACE.  The solve runs on bp[i] = b[order[i]] and xp[i] (gathered / scattered by the whole chip around it:
//     two ~10 us launches), so the one workgroup that walks streams contiguous right-hand sides and results.  A single
//     CU sustains few misses at a time (~15-30 GB/s from memory): the row-indexed b[row] / x[row] of a chunk were two
//     thirds of its cache lines.
//   * HOST-BUILT RECORDS (TriFactor::build_walk2).  Per position 48 bytes: the first W2_Q = 3 entries' values, a "source
//     word" each -- the LDS byte address its x value will have: a ring slot, a slot of the chunk's s_x area, or a
//     constant 0.0 for an absent entry (fma(-0, 0, s) = s for every s: the walk needs no masks) -- and the entry range.
//     x values the ring no longer holds ("early": 4 % of the G3_circuit-like factor's entries) are a list per chunk
//     {producer position, s_x slot}.  Chunks hold <= W2_SUB sub-levels of <= 64 rows; the header carries their widths.
//   * STAGERS (waves 1-3).  During the walk of chunk k they write chunk k+1's records into the other half of a
//     double-buffered LDS area, request the early x values of chunk k+2 and the records of chunk k+3, and write chunk
//     k-1's results from the ring to xp.  Two register sets swap roles every phase; no load is waited for in the phase
//     that issued it; no stager reads what another wrote: the only synchronisation is the barrier that ends a phase.
//   * WALKER (wave 0).  Holds the chunk's records in registers (the workgroup is alone on its CU: 512 VGPRs) and runs
//     the levels back to back, fully unrolled: per level the x reads, the FMAs in stored order, (the division,) the ring
//     write -- and, in the shadow of that chain, the LDS reads of the records four levels ahead.  A chunk without longer
//     rows is straight-line code: the compiler counts the LDS operations in flight and a level's x reads go out right
//     behind the previous level's ring write (LDS executes a wave's operations in order).
//   * READ-AHEAD (workgroup 8 of a 9-workgroup launch: same XCD, hence the same L2).  Touches one word per 128-byte
//     line of the chunks ten ahead of the team, paced by a progress word: the stagers' loads become L2 hits.
// Early sources must be W2_DIST chunks back: a chunk's results leave for memory one phase after its walk, the stores
// drain during the next phase (never waited for directly), and loads issued two phases later see them.
// Measured (one ILU(0) application, profiles/r03_trsv.txt): G3_circuit-like 14.8 ms against 32.2 for the four-wave walk
// of round 2 -- walker 2 870 / 4 360 cycles per chunk of ~15 levels (lower / upper: the upper solve divides), stagers
// 3 700-3 800 -- ; the stencil system 2.1 against 3.3 (its wide levels gain from position space); cant-like (32 entries
// a row) 16.3 against 31.7: in a step of long rows a row's tail is padded to W2_TAIL entries of +0.0 x (constant 0.0) and
// taken W2_TAIL at a time, no index clamped, no product masked (with clamps and masks: 32-40 ms).
// Arithmetic and order are solve_row's: bit-identical to the other schedules (tested).
constexpr int W2_ST = 192, W2_T = 64 + W2_ST;            // wave 0 walks, waves 1-3 stage
constexpr int W2_GRID = 9;             // workgroup 0 solves, workgroup 8 (same XCD: same L2) reads ahead, the others leave
constexpr int W2_CH = 640;             // positions per chunk
constexpr int W2_ECAP = 1280;          // off-diagonal entries per chunk
constexpr int W2_SUB = 16;             // sub-levels (<= 64 rows each) per chunk: 12-14 VGPRs of records apiece in the walker
constexpr int W2_HAS_OVF = 1 << 24;    // in a chunk header's second word (positions | early sources << 10): some row has more than W2_Q entries
constexpr int W2_EARLY_PT = 2;         // early sources (x values read from memory) per stager thread and chunk
constexpr unsigned W2_IN_BUF = 0x80000000u;   // a source word: LDS byte address, relative to the chunk's buffer when this bit is set
constexpr int W2_TAIL = 32;            // long rows: the entries beyond the record come W2_TAIL at a time (tails padded to it)
constexpr int W2_Q = 3;                // entries of a row held in its record; a longer row reads the rest from the entry arrays
constexpr int W2_DIST = 6;             // early sources: producer chunk <= consumer chunk - W2_DIST
constexpr int W2_PJ = (W2_CH + W2_ST - 1) / W2_ST, W2_EJ = (W2_ECAP + W2_ST - 1) / W2_ST;
static_assert(PK_RING >= W2_DIST * W2_CH + WIDE_LEVEL, "everything that is not an early source is still in the ring");
static_assert(W2_ECAP < 2048 && W2_SUB % 4 == 0 && W2_SUB <= 24, "meta word: 11 + 11 bits; header: 6 words of counts");
// a buffer: recA {b, v0} | recB {v1, v2} | recC {byte addresses of x0, x1, x2, meta} | recD {diagonal}, each with 64 spare
// records (the walker reads 64 records from a sub-level's first, whatever its width) | s_val | s_x | s_src
constexpr int W2_RECS = W2_CH + 64;
constexpr size_t W2_BUF_BYTES = 56 * (size_t)W2_RECS + 20 * (size_t)W2_ECAP;
constexpr int W2_ZERO = (int)(sizeof(double) * PK_RING + 2 * W2_BUF_BYTES);   // LDS byte address of a constant 0.0 ...
constexpr int W2_DUMP = W2_ZERO + 8;                                           // ... and of a slot nobody reads
constexpr size_t W2_LDS_BYTES = W2_DUMP + 8;
static_assert(W2_LDS_BYTES <= 160 * 1024, "one workgroup's LDS on gfx950");

struct W2Pos {                         // 48 bytes per position (level order): three 16-byte loads
  double v0, v1, v2, diag;             //   values of the first three entries (0.0 when absent), the diagonal
  unsigned s0, s1, s2, meta;           //   where their x values are (source words); entries | first entry (chunk-relative) << 11
};
struct W2Ent {                         // 16 bytes per entry, level order (read for chunks with longer rows only)
  double val;
  unsigned src, pad;                   //   source word
};
struct W2Early {                       // an x value a chunk reads from memory: producer position -> s_x slot (chunk-relative entry)
  int pos, slot;
};
struct Walk2Tri {
  const W2Pos *pos;
  const W2Ent *ent;
  const W2Early *early;
  const int4 *hdr;                     // per chunk: {first position, positions | early sources << 10 | flag, first entry, entries}, {first early source, ...
};                                     //   {sub-levels, widths 0-3, 4-7, 8-11 (a byte each)}, {12-15, 16-19, 20-23, sub-levels with longer rows}

// walk2 works in POSITION space: bp[i] = b[order[i]] is gathered by the whole chip before the solve and xp scattered
// back after it (two launches of ~10 us), so that the one workgroup that walks the levels streams contiguous right-hand
// sides and results instead of one 128-byte line per row -- a single CU moves ~30 GB/s, and the scattered b[row] and
// x[row] of a chunk were two thirds of its lines.
__global__ void k_w2_gather(int n, const int *__restrict__ order, const double *__restrict__ b, double *__restrict__ bp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bp[i] = b[order[i]];
}
__global__ void k_w2_scatter(int n, const int *__restrict__ order, const double *__restrict__ xp, double *__restrict__ x) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[order[i]] = xp[i];
}
// position i: xp[i] = (bp[i] - sum val * xp[producer position]) / diagonal, entries in stored order (solve_row's arithmetic)
__device__ __forceinline__ void solve_position(int i, int unit, const int *__restrict__ eptr, const int *__restrict__ epos,
                                               const double *__restrict__ eval, const double *__restrict__ pdiag,
                                               const double *__restrict__ bp, double *xp) {
  double s = bp[i];
  for (int e = eptr[i]; e < eptr[i + 1]; e++) s -= eval[e] * xp[epos[e]];
  xp[i] = unit ? s : s / pdiag[i];
}
__global__ void k_trsv_level_p(int lo, int hi, int unit, const int *__restrict__ eptr, const int *__restrict__ epos,
                               const double *__restrict__ eval, const double *__restrict__ pdiag,
                               const double *__restrict__ bp, double *xp) {
  const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hi) solve_position(i, unit, eptr, epos, eval, pdiag, bp, xp);
}
__global__ void k_trsv_levels_p(int l0, int l1, int unit, const int *__restrict__ level_ptr, const int *__restrict__ eptr,
                                const int *__restrict__ epos, const double *__restrict__ eval,
                                const double *__restrict__ pdiag, const double *__restrict__ bp, double *xp) {
  for (int l = l0; l < l1; l++) {
    const int lo = level_ptr[l], hi = level_ptr[l + 1];
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) solve_position(i, unit, eptr, epos, eval, pdiag, bp, xp);
    __syncthreads();
  }
}

// lgkmcnt(0) as an instruction the compiler's own wait-count bookkeeping sees (an asm statement is opaque to it: it
// would keep every LDS read of the last phase "pending" and serialise the next phase's reads behind waits)
__device__ __forceinline__ void w2_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);                         // vmcnt 63, expcnt 7, lgkmcnt 0
  __builtin_amdgcn_s_barrier();
}

// One chunk by the walker wave.  Records of sub-level s: {b, v0} {v1, v2} {addresses of x0, x1, x2, meta} (and the
// diagonal).  An absent entry has v = +0.0 and the address of a constant 0.0 -- fma(-0, 0, s) = s for every s -- so the
// walk needs no masks; lanes past a level's width hold some other row's record and write to a dump slot.  The records
// of sub-level s + W2_AHEAD are requested behind the x reads of sub-level s: they fill the LDS pipeline in the shadow
// of the dependent chain (x reads -> FMAs -> ring write), which is all a level costs.  OVF = false is the chunk without
// any longer row: straight-line code, so the compiler counts what is in flight and a level's x reads go out right
// behind the previous level's ring write (the LDS executes a wave's operations in order).
template <bool LONG, bool UNIT, bool OVF>
__device__ __forceinline__ void w2_walk_chunk(char *lds_bytes, int base, int cs, const int (&cw)[6], int ovf, int lane) {
  constexpr int OFF_B = 16 * W2_RECS, OFF_C = 32 * W2_RECS, OFF_D = 48 * W2_RECS, OFF_VAL = 56 * W2_RECS,
                OFF_SRC = OFF_VAL + 16 * W2_ECAP;
  constexpr int W2_AHEAD = 4;
  pk_dbl2 ra[W2_SUB], rb[W2_SUB];
  pk_int4 rc[W2_SUB];
  double rd[W2_SUB];
  int lo = 0, lo_solve = 0;
  auto preload = [&](int s) {                                 // unconditional: the compiler can count what is in flight
    const char *rec = lds_bytes + base + 16 * (lo + lane);
    ra[s] = *reinterpret_cast<const pk_dbl2 *>(rec);
    rb[s] = *reinterpret_cast<const pk_dbl2 *>(rec + OFF_B);
    rc[s] = *reinterpret_cast<const pk_int4 *>(rec + OFF_C);
    if constexpr (!UNIT) rd[s] = *reinterpret_cast<const double *>(lds_bytes + base + OFF_D + 8 * (lo + lane));
    lo += (cw[s >> 2] >> (8 * (s & 3))) & 0xff;
  };
#pragma unroll
  for (int s = 0; s < W2_AHEAD; s++) preload(s);
  const double *s_val = reinterpret_cast<const double *>(lds_bytes + base + OFF_VAL);
  const int *s_src = reinterpret_cast<const int *>(lds_bytes + base + OFF_SRC);
#pragma unroll
  for (int s = 0; s < W2_SUB; s++) {
    // the x reads first, the records of sub-level s + W2_AHEAD behind them in the LDS queue
    const double x0 = *reinterpret_cast<const double *>(lds_bytes + rc[s].x);
    const double x1 = *reinterpret_cast<const double *>(lds_bytes + rc[s].y);
    const double x2 = *reinterpret_cast<const double *>(lds_bytes + rc[s].z);
    if (s + W2_AHEAD < W2_SUB) preload(s + W2_AHEAD);
    const int width = (cw[s >> 2] >> (8 * (s & 3))) & 0xff;
    const bool mine = lane < width;                           // lanes past the level's width hold some other row's record
    const int dst = mine ? ((cs + lo_solve + lane) & (PK_RING - 1)) * 8 : W2_DUMP;
    lo_solve += width;
    double sacc = ra[s].x - ra[s].y * x0;                     // (contracted to one fma, like `s -= v * x` everywhere else)
    sacc -= rb[s].x * x1;
    sacc -= rb[s].y * x2;
    if constexpr (OVF) {
      if ((ovf >> s) & 1) {                                   // (scalar) some row of this sub-level has more entries
        const int ne = mine ? (int)(rc[s].w & 0x7ff) : 0;
        if (W2_Q < ne) {
          const int e0 = (rc[s].w >> 11) & 0x7ff;
          if constexpr (!LONG) {
            for (int e = e0 + W2_Q; e < e0 + ne; e++)
              sacc -= s_val[e] * *reinterpret_cast<const double *>(lds_bytes + s_src[e]);
          } else {
            for (int e = e0 + W2_Q; e < e0 + ne; e += W2_TAIL) {   // (ne: the padded count -- no clamp, no mask)
              // straight-line: all W2_TAIL sources and values requested at once (they do not depend on x), then the x
              // values, then the FMAs in stored order
              double v4[W2_TAIL], x4[W2_TAIL];
              int a4[W2_TAIL];
#pragma unroll
              for (int u = 0; u < W2_TAIL; u++) a4[u] = s_src[e + u];
#pragma unroll
              for (int u = 0; u < W2_TAIL; u++) v4[u] = s_val[e + u];
#pragma unroll
              for (int u = 0; u < W2_TAIL; u++) x4[u] = *reinterpret_cast<const double *>(lds_bytes + a4[u]);
#pragma unroll
              for (int u = 0; u < W2_TAIL; u++) sacc -= v4[u] * x4[u];
            }
          }
        }
      }
    }
    *reinterpret_cast<double *>(lds_bytes + dst) = UNIT ? sacc : sacc / rd[s];
  }
}

template <bool LONG, bool UNIT>
__global__ void __launch_bounds__(W2_T)
k_trsv_walk2(Walk2Tri t, int c0, int c1, const double *__restrict__ bp, double *xp, int *progress,
             unsigned long long *dbg) {
  // Workgroup 0 is the team that solves; workgroup 8 -- dispatched to the same XCD, hence the same L2 -- only reads
  // ahead: one word per 128-byte line of the chunks the team will stage next, paced by the team's progress word.  A CU
  // sustains few misses at a time (this workgroup alone streams ~15 GB/s from memory); with the lines already in the
  // L2 a stager's load costs an L2 hit.  Nothing depends on the read-ahead: if it falls behind or lands elsewhere, the
  // solve is only slower.
  if (blockIdx.x != 0) {
    if (blockIdx.x != 8) return;
    constexpr int FETCH_AHEAD = 10;
    unsigned acc = 0;
    const int ft = threadIdx.x;
    for (int f = c0 + 3; f < c1; f++) {
      for (int polls = 0; polls < (1 << 14); polls++) {       // bounded: a stuck progress word only costs the pacing
        if (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + FETCH_AHEAD >= f) break;
        __builtin_amdgcn_s_sleep(20);
      }
      const int4 h = t.hdr[3 * f];
      const int fcnt = h.y & 0x3ff;
      const long p0 = 48L * h.x, p1 = 48L * (h.x + fcnt), e0 = 16L * h.z, e1 = (h.y & W2_HAS_OVF) ? 16L * (h.z + h.w) : e0,
                 b0 = 8L * h.x, b1 = 8L * (h.x + fcnt);
      for (long a = (p0 & ~127L) + 128L * ft; a < p1; a += 128L * W2_T)
        acc += *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(t.pos) + a);
      for (long a = (e0 & ~127L) + 128L * ft; a < e1; a += 128L * W2_T)
        acc += *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(t.ent) + a);
      for (long a = (b0 & ~127L) + 128L * ft; a < b1; a += 128L * W2_T)
        acc += *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(bp) + a);
      if (ft == 0) acc += (unsigned)t.hdr[3 * min(f + 1, c1 - 1)].x;
    }
    if (acc == 0x9e3779b9u) *progress = -1;                   // (keeps the loads alive)
    return;
  }
  extern __shared__ double pk_lds[];
  double *ring = pk_lds;
  char *lds_bytes = reinterpret_cast<char *>(pk_lds);
  const int tid = threadIdx.x, lane = tid & 63, st = tid - 64;
  auto buf_base = [&](int which) { return (int)(sizeof(double) * PK_RING + (size_t)which * W2_BUF_BYTES); };
  constexpr int OFF_B = 16 * W2_RECS, OFF_C = 32 * W2_RECS, OFF_D = 48 * W2_RECS, OFF_VAL = 56 * W2_RECS,
                OFF_X = OFF_VAL + 8 * W2_ECAP, OFF_SRC = OFF_X + 8 * W2_ECAP;

  if (tid < 64) {
    // ------------------------------------------------------------------------------------------------ the walker
    // The level widths of a chunk come by a VECTOR load (lane i: word i of the header's last 32 bytes), a phase ahead:
    // a scalar load in flight would share lgkmcnt with the LDS reads and, returning out of order, force every wait
    // for an LDS read to wait for all of them.
    const int *hw = reinterpret_cast<const int *>(t.hdr);
    int gw = hw[12 * c0 + 4 + (lane & 7)], gcs = hw[12 * c0];
    unsigned long long d_wait = 0, d_walk = 0;
    for (int k = c0; k < c1; k++) {
      const unsigned long long q0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
      w2_barrier();                                           // chunk k is staged
      const unsigned long long q1 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
      const int w = gw, cs = uniform(gcs);
      gw = hw[12 * min(k + 1, c1 - 1) + 4 + (lane & 7)];
      gcs = hw[12 * min(k + 1, c1 - 1)];
      const int cw[6] = {__builtin_amdgcn_readlane(w, 1), __builtin_amdgcn_readlane(w, 2), __builtin_amdgcn_readlane(w, 3),
                         __builtin_amdgcn_readlane(w, 4), __builtin_amdgcn_readlane(w, 5), __builtin_amdgcn_readlane(w, 6)};
      const int ovf = __builtin_amdgcn_readlane(w, 7);        // sub-levels with a row of more than W2_Q entries
      const int base = buf_base((k - c0) & 1);
      if (ovf) w2_walk_chunk<LONG, UNIT, true>(lds_bytes, base, cs, cw, ovf, lane);
      else w2_walk_chunk<LONG, UNIT, false>(lds_bytes, base, cs, cw, 0, lane);
      if (dbg) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        d_wait += q1 - q0;
        d_walk += __builtin_amdgcn_s_memtime() - q1;
      }
    }
    if (dbg && lane == 0) { atomicAdd(dbg + 0, d_wait); atomicAdd(dbg + 1, d_walk); atomicAdd(dbg + 3, (unsigned long long)(c1 - c0)); }
    w2_barrier();                                             // the last chunk's x values are in the ring
    w2_barrier();                                             // (the stagers' extra phase: the last write-back)
    return;
  }

  // -------------------------------------------------------------------------------------------------- the stagers
  // Three chunks are in flight per thread: `cur` (all loaded: staged this phase), `nxt` (its x-independent loads
  // landed during the last phase: b[row] and the early x values are requested this phase), and the chunk after it,
  // whose x-independent loads go out at the end of the phase.  No load is waited for in the phase that issued it.
  struct Indep {                                              // what does not depend on x
    int4 q0[W2_PJ], q1[W2_PJ], q2[W2_PJ];                     // {v0, v1} {v2, diagonal} {source words, meta} of a position
    double b[W2_PJ];                                          // its right-hand side
    int4 ent[W2_EJ];                                          // {value, source word, -} of an entry (chunks with longer rows only)
    int2 early[W2_EARLY_PT];                                  // {producer position, s_x slot} of an x value read from memory
  };
  Indep P, Q;
  double l_x[W2_EARLY_PT];                                    // the early x values of the chunk staged next
  int cs1 = 0, cnt1 = 0, cs2 = 0, cnt2 = 0, cs3 = 0, cnt3 = 0;   // the last three chunks staged (3: written back next)
  auto rotate_chunks = [&]() {
    cs3 = cs2; cnt3 = cnt2;
    cs2 = cs1; cnt2 = cnt1;
  };
  auto as_double = [](int lo, int hi) { return __hiloint2double(hi, lo); };

  auto load_indep = [&](Indep &d, const int4 &h, int ebase_early) {
    const int cs = uniform(h.x), cnt = uniform(h.y) & 0x3ff, n_early = (uniform(h.y) >> 10) & 0x3ff, ebase = uniform(h.z),
              ecnt = uniform(h.w);
    const bool longer = uniform(h.y) & W2_HAS_OVF;
#pragma unroll
    for (int j = 0; j < W2_PJ; j++) {
      // (unconditional, clamped: a conditional load would be merged with the register's old value through a copy, and
      // the copy waits for the load -- 700 cycles each, one after the other)
      const int p = cs + min(st + W2_ST * j, cnt - 1);
      const int4 *rec = reinterpret_cast<const int4 *>(t.pos + p);
      d.q0[j] = rec[0];
      if constexpr (UNIT) {                                   // (v2 only: one 8-byte load, not half of a 16-byte one)
        const int2 v2 = *reinterpret_cast<const int2 *>(rec + 1);
        d.q1[j].x = v2.x;
        d.q1[j].y = v2.y;
      } else d.q1[j] = rec[1];
      d.q2[j] = rec[2];
      d.b[j] = bp[p];
    }
    if (longer) {                                             // (the entry arrays repeat what the records hold)
#pragma unroll
      for (int j = 0; j < W2_EJ; j++)
        d.ent[j] = *reinterpret_cast<const int4 *>(t.ent + ebase + min(st + W2_ST * j, max(ecnt - 1, 0)));
    }
#pragma unroll
    for (int j = 0; j < W2_EARLY_PT; j++)
      d.early[j] = *reinterpret_cast<const int2 *>(t.early + ebase_early + min(st + W2_ST * j, max(n_early - 1, 0)));
  };
  auto load_dep = [&](const Indep &d, const int4 &h) {        // xp[producer position] of the early sources
    const int n_early = (uniform(h.y) >> 10) & 0x3ff;
#pragma unroll
    for (int j = 0; j < W2_EARLY_PT; j++)
      l_x[j] = __hip_atomic_load(xp + (st + W2_ST * j < n_early ? d.early[j].x : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto stage = [&](const Indep &d, const int4 &h, int which) {  // registers -> buffer `which`
    const int cs = uniform(h.x), cnt = uniform(h.y) & 0x3ff, n_early = (uniform(h.y) >> 10) & 0x3ff, ecnt = uniform(h.w);
    const bool longer = uniform(h.y) & W2_HAS_OVF;
    const int base = buf_base(which);
    // a source word is an LDS byte address: of a ring slot or the constant 0.0 as it stands, of an s_x slot of this
    // chunk's buffer when W2_IN_BUF is set (the host cannot know which of the two buffers the chunk gets)
    auto address = [&](unsigned word) { return (int)(word & ~W2_IN_BUF) + ((word & W2_IN_BUF) ? base : 0); };
#pragma unroll
    for (int j = 0; j < W2_PJ; j++) {
      const int p = st + W2_ST * j;
      if (W2_ST * j >= cnt) continue;
      if (p < cnt) {
        pk_dbl2 a, bb;
        pk_int4 c;
        a.x = d.b[j]; a.y = as_double(d.q0[j].x, d.q0[j].y);
        bb.x = as_double(d.q0[j].z, d.q0[j].w); bb.y = as_double(d.q1[j].x, d.q1[j].y);
        c.x = address((unsigned)d.q2[j].x);
        c.y = address((unsigned)d.q2[j].y);
        c.z = address((unsigned)d.q2[j].z);
        c.w = d.q2[j].w;
        *reinterpret_cast<pk_dbl2 *>(lds_bytes + base + 16 * p) = a;
        *reinterpret_cast<pk_dbl2 *>(lds_bytes + base + OFF_B + 16 * p) = bb;
        *reinterpret_cast<pk_int4 *>(lds_bytes + base + OFF_C + 16 * p) = c;
        if constexpr (!UNIT) *reinterpret_cast<double *>(lds_bytes + base + OFF_D + 8 * p) = as_double(d.q1[j].z, d.q1[j].w);
      }
    }
    if (longer) {
#pragma unroll
      for (int j = 0; j < W2_EJ; j++) {
        const int e = st + W2_ST * j;
        if (e < ecnt) {
          *reinterpret_cast<double *>(lds_bytes + base + OFF_VAL + 8 * e) = as_double(d.ent[j].x, d.ent[j].y);
          *reinterpret_cast<int *>(lds_bytes + base + OFF_SRC + 4 * e) = address((unsigned)d.ent[j].z);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < W2_EARLY_PT; j++)                     // the x values read from memory, where the source words point
      if (st + W2_ST * j < n_early) *reinterpret_cast<double *>(lds_bytes + base + OFF_X + 8 * d.early[j].y) = l_x[j];
    rotate_chunks();
    cs1 = cs; cnt1 = cnt;
  };
  auto write_back3 = [&]() {                                  // ring -> xp for the chunk behind the one being walked:
    if (cnt3 > 0) {                                           //   always W2_PJ stores per thread (lanes past the end repeat the
#pragma unroll                                                //   chunk's last position)
      for (int j = 0; j < W2_PJ; j++) {
        const int p = cs3 + min(st + W2_ST * j, cnt3 - 1);
        __hip_atomic_store(xp + p, ring[p & (PK_RING - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };

  if (tid == 64) {
    *reinterpret_cast<double *>(lds_bytes + W2_ZERO) = 0.0;
    *reinterpret_cast<double *>(lds_bytes + W2_DUMP) = 0.0;
  }
  // Three chunks are in flight per thread, in two register sets that swap roles every phase (no copies): in phase k set
  // X holds chunk k + 1 complete (staged now, then reloaded with the x-independent part of chunk k + 3), set Y the
  // x-independent part of chunk k + 2 (its b[row] and early x values are requested now).  No load is waited for in the
  // phase that issued it.  Chunk headers travel ahead of their loads.
  auto hdr_at = [&](int k) { return t.hdr[3 * min(k, c1 - 1)]; };
  auto early_at = [&](int k) { return uniform(t.hdr[3 * min(k, c1 - 1) + 1].x); };   // first early source of a chunk
  int4 hA = hdr_at(c0), hB = hdr_at(c0 + 1), hC = hdr_at(c0 + 2), hD = hdr_at(c0 + 3);
  int eC = early_at(c0 + 2), eD = early_at(c0 + 3);
  load_indep(P, hA, early_at(c0));
  load_indep(Q, hB, early_at(c0 + 1));
  load_dep(P, hA);                                            // (waits for P's early list)
  stage(P, hA, 0);                                            // chunk c0 (waits for everything)
  load_dep(Q, hB);                                            // chunk c0 + 1
  load_indep(P, hC, eC);                                      // chunk c0 + 2
  hA = hB; hB = hC; hC = hD; hD = hdr_at(c0 + 4);
  eC = eD; eD = early_at(c0 + 4);
  unsigned long long e_wait = 0, e_work = 0, e_a = 0, e_b = 0, e_c = 0, e_d = 0;
  auto phase = [&](Indep &X, Indep &Y, int k) {               // X = chunk k + 1, Y = chunk k + 2 (x-independent part)
    const unsigned long long q0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    w2_barrier();                                             // phase k: the walker is on chunk k
    const unsigned long long q1 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    // Memory operations retire in order.  The youngest W2_PJ of this thread are the write-back stores of the last
    // phase: everything older -- every load (each had that phase to land) and the stores of the phase before it -- is
    // complete after this wait, and the write-through stores themselves (1.2 us) are never waited for.  By the
    // barrier that ends THIS phase every thread has passed this point, so the results written back during phase k - 2
    // (chunk k - 3) are in memory for every load issued from phase k + 1 on: those are for chunk k + 3 = the
    // producer's chunk + W2_DIST.
    static_assert(W2_PJ == 4, "the vmcnt immediate below counts the W2_PJ = 4 write-back stores per thread");
    __builtin_amdgcn_s_waitcnt(0x0F74);                       // vmcnt 4, expcnt 7, lgkmcnt 15
    const unsigned long long q2 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    if (tid == 64) __hip_atomic_store(progress, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (paces the read-ahead workgroup)
    if (k + 1 < c1) stage(X, hA, (k + 1 - c0) & 1);           // chunk k + 1
    else rotate_chunks();
    if (dbg) __builtin_amdgcn_s_waitcnt(0xC07F);
    const unsigned long long q3 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    if (k + 2 < c1) load_dep(Y, hB);                          // chunk k + 2
    if (k + 3 < c1) load_indep(X, hC, eC);                    // chunk k + 3
    hA = hB; hB = hC; hC = hD; hD = hdr_at(k + 5);
    eC = eD; eD = early_at(k + 5);
    const unsigned long long q4 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    write_back3();                                            // chunk k - 1 (nothing in the first phase): the youngest operations
    if (dbg) { const unsigned long long q5 = __builtin_amdgcn_s_memtime(); e_wait += q1 - q0; e_work += q5 - q1; e_a += q2 - q1; e_b += q3 - q2; e_c += q4 - q3; e_d += q5 - q4; }
  };
  for (int k = c0; k < c1; k += 2) {
    phase(Q, P, k);
    if (k + 1 < c1) phase(P, Q, k + 1);
  }
  if (dbg && tid == 64) { atomicAdd(dbg + 4, e_wait); atomicAdd(dbg + 6, e_work); atomicAdd(dbg + 8, e_a); atomicAdd(dbg + 9, e_b); atomicAdd(dbg + 10, e_c); atomicAdd(dbg + 11, e_d); }
  w2_barrier();                                               // the last chunk is walked
  rotate_chunks();
  write_back3();
  w2_barrier();
}

__global__ void k_scale(int64_t n, const double *__restrict__ dinv, const double *__restrict__ r, double *__restrict__ z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    z[i] = r[i] * dinv[i];
}

// One triangular factor (diagonal included) on the device with its level schedule.
struct TriFactor {
  int n = 0;
  bool lower = true, unit = false;
  int mode = 0;                                                 // CASK_HIP_TRSV when the factor was built (forced_mode)
  DevBuf<int> rp, ci, order, level_ptr;
  DevBuf<double> val;
  int n_levels = 0;
  struct Step { int l0, l1, lo, hi; bool wide; int c0, c1; bool long_rows; int d0 = -1, d1 = -1, g0 = -1, g1 = -1, ge = 0; };   // levels [l0,l1) = positions [lo,hi) of `order`;
  std::vector<Step> steps;                                      // chunks [c0,c1) of the packed form (c0 < 0: none), [d0,d1) of walk2's, [g0,g1) of the lane-group walk's
  // the factor once more in level order for k_trsv_packed (see there)
  DevBuf<int> pk_row, pk_eptr, pk_seg;
  DevBuf<int4> pk_hdr;
  DevBuf<unsigned> pk_code;
  DevBuf<double> pk_diag, pk_val;
  // ... and once more for k_trsv_walk2 (CASK_HIP_TRSV=walk2): per-position records, its own chunks [d0,d1) per step
  DevBuf<W2Pos> w2_pos;
  DevBuf<W2Ent> w2_ent;
  DevBuf<W2Early> w2_early;
  DevBuf<int> w2_eptr, w2_epos;
  DevBuf<double> w2_eval, w2_pdiag;
  mutable DevBuf<double> w2_bp, w2_xp;             // the right-hand side and the result in position space
  mutable DevBuf<int> w2_progress;                 // the chunk the team is on (paces the read-ahead workgroup)
  bool w2_ok = false, walk2 = false;                // walk2: this factor is solved by k_trsv_walk2 (position space)
  DevBuf<int4> w2_hdr;
  mutable DevBuf<unsigned long long> w2_dbg;
  // ... and the slabs of k_trsv_lanes (trsv_lanes.hpp) for the runs of narrow levels with long rows
  DevBuf<char> ln_lanes;
  DevBuf<int> ln_hdr;
  DevBuf<double> ln_rdiag;                          // 1 / diagonal, position space

  int build(int n_, bool lower_, const std::vector<int> &h_rp, const std::vector<int> &h_ci,
            const std::vector<double> &h_val) {
    n = n_;
    lower = lower_;
    // level of a row = 1 + max level of the rows it depends on
    std::vector<int> level(n, 0);
    n_levels = n > 0 ? 1 : 0;
    auto visit = [&](int r) {
      int lv = 0;
      for (int k = h_rp[r]; k < h_rp[r + 1]; k++) {
        const int c = h_ci[k];
        if (lower ? c < r : c > r) lv = std::max(lv, level[c] + 1);
      }
      level[r] = lv;
      n_levels = std::max(n_levels, lv + 1);
    };
    if (lower) for (int r = 0; r < n; r++) visit(r);
    else       for (int r = n - 1; r >= 0; r--) visit(r);
    std::vector<int> lp(n_levels + 1, 0), ord(n);
    for (int r = 0; r < n; r++) lp[level[r] + 1]++;
    for (int l = 0; l < n_levels; l++) lp[l + 1] += lp[l];
    std::vector<int> fill(lp.begin(), lp.end() - 1);
    for (int r = 0; r < n; r++) ord[fill[level[r]]++] = r;
    steps.clear();
    mode = forced_mode();                                       // (read when the factor is built: a process may build factors under several)
    const int wide_from = mode == 2 ? WIDE_LEVEL_ROWWALK : WIDE_LEVEL;
    for (int l = 0; l < n_levels;) {
      if (lp[l + 1] - lp[l] >= wide_from) {
        steps.push_back(Step{l, l + 1, lp[l], lp[l + 1], true, -1, -1, false});
        l++;
        continue;
      }
      int e = l;
      while (e < n_levels && lp[e + 1] - lp[e] < wide_from) e++;
      steps.push_back(Step{l, e, lp[l], lp[e], false, -1, -1, false});
      l = e;
    }
    bool any_long = false;
    for (Step &st : steps) {
      if (st.wide) continue;
      int64_t ents = 0;
      for (int i = st.lo; i < st.hi; i++) ents += h_rp[ord[i] + 1] - h_rp[ord[i]] - 1;
      st.long_rows = ents > 6 * (int64_t)(st.hi - st.lo);      // more than 6 entries a row on average
      any_long = any_long || st.long_rows;
    }
    (void)any_long;
    walk2 = mode == 4 || mode == 0 || mode == 6;
    int rc = walk2 ? build_walk2(h_rp, h_ci, h_val, level, lp, ord) : build_packed(h_rp, h_ci, h_val, level, lp, ord);
    if (rc) return rc;
    PC_TRY(rp.upload(h_rp));
    PC_TRY(ci.upload(h_ci));
    PC_TRY(val.upload(h_val));
    PC_TRY(order.upload(ord));
    PC_TRY(level_ptr.upload(lp));
    return CASK_HIP_OK;
  }

  // Level-ordered copy of the factor, its chunks and the source of every dependency (near / early), per narrow step.
  // A step that cannot be packed (a row with more than PK_ECAP entries) keeps c0 = -1 and runs k_trsv_levels.
  int build_packed(const std::vector<int> &h_rp, const std::vector<int> &h_ci, const std::vector<double> &h_val,
                   const std::vector<int> &level, const std::vector<int> &lp, const std::vector<int> &ord) {
    std::vector<int> pos_of((size_t)n), prow(ord), peptr((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) pos_of[ord[i]] = i;
    std::vector<unsigned> pcode;
    std::vector<double> pval, pdiag((size_t)n, 0.0);
    pcode.reserve(h_ci.size());
    pval.reserve(h_ci.size());
    for (int i = 0; i < n; i++) {
      const int r = ord[i];
      for (int k = h_rp[r]; k < h_rp[r + 1]; k++) {
        const int c = h_ci[k];
        if (c == r) pdiag[i] = h_val[k];
        else if (lower ? c < r : c > r) {
          pcode.push_back((unsigned)c);                       // the column for now; near ones become positions below
          pval.push_back(h_val[k]);
        }
      }
      peptr[i + 1] = (int)pcode.size();
    }
    std::vector<int> chunk, segptr, seg;
    for (Step &st : steps) {
      if (st.wide) continue;
      const size_t chunk0 = chunk.size(), segptr0 = segptr.size(), seg0 = seg.size();
      bool ok = true;
      // what a chunk's early loads may rely on: positions before the start of the chunk TWO before it (the kernel
      // lets the x stores of a chunk drain during the next chunk's walk)
      int prev_start = st.lo, prev2_start = st.lo;
      for (int cs = st.lo; cs < st.hi && ok;) {
        int ce = cs;
        while (ce < st.hi && ce - cs < PK_CH && peptr[ce + 1] - peptr[cs] <= PK_ECAP) ce++;
        if (ce == cs) { ok = false; break; }
        chunk.push_back(cs);
        segptr.push_back((int)seg.size());
        for (int i = cs; i < ce;) {                           // level ends inside the chunk, and the chunk's own end
          const int end = std::min(lp[level[ord[i]] + 1], ce);
          seg.push_back(end);
          i = end;
        }
        for (int i = cs; i < ce && ok; i++) {
          const int level_end = lp[level[ord[i]] + 1];
          for (int e = peptr[i]; e < peptr[i + 1]; e++) {
            const int pp = pos_of[pcode[e]];
            if (pp >= st.lo && pp >= level_end - PK_RING) pcode[e] = PK_NEAR | (unsigned)pp;
            else if (pp >= prev2_start) { ok = false; break; } // cannot happen while PK_RING >= 3 PK_CH + WIDE_LEVEL
          }
        }
        prev2_start = prev_start;
        prev_start = cs;
        cs = ce;
      }
      if (!ok) {                                              // restore the columns of what was already rewritten
        chunk.resize(chunk0);
        segptr.resize(segptr0);
        seg.resize(seg0);
        for (int i = st.lo; i < st.hi; i++)
          for (int e = peptr[i]; e < peptr[i + 1]; e++)
            if (pcode[e] & PK_NEAR) pcode[e] = (unsigned)ord[pcode[e] & ~PK_NEAR];
        continue;
      }
      st.c0 = (int)chunk0;
      st.c1 = (int)chunk.size();
      int64_t ents = peptr[st.hi] - peptr[st.lo];
      st.long_rows = ents > 6 * (int64_t)(st.hi - st.lo);      // more than 6 entries a row on average
      chunk.push_back(st.hi);                                 // every step's list ends with its own sentinel
      segptr.push_back((int)seg.size());
    }
    // chunk headers (the step lists in `chunk` / `segptr` end with a sentinel each, so entry k+1 closes chunk k)
    std::vector<int4> hdr(2 * chunk.size());
    for (size_t k = 0; k + 1 < chunk.size(); k++) {
      const int cs = chunk[k], ce = chunk[k + 1];
      if (ce <= cs) continue;                                 // a sentinel followed by the next step's first chunk
      hdr[2 * k] = make_int4(cs, ce - cs, peptr[cs], peptr[ce] - peptr[cs]);
      hdr[2 * k + 1] = make_int4(segptr[k], segptr[k + 1] - segptr[k], 0, 0);
    }
    pcode.push_back(0u);                                      // one spare element each: the kernel's clamped loads
    pval.push_back(0.0);
    seg.push_back(0);
    PC_TRY(pk_row.upload(prow));
    PC_TRY(pk_eptr.upload(peptr));
    PC_TRY(pk_code.upload(pcode));
    PC_TRY(pk_val.upload(pval));
    PC_TRY(pk_diag.upload(pdiag));
    PC_TRY(pk_hdr.upload(hdr));
    PC_TRY(pk_seg.upload(seg));
    // The packed walk needs PK_LDS_BYTES (116 KB) of dynamic LDS: fine on gfx950 (160 KB per CU).  On a device -- or
    // under an ARCH override -- that cannot give a workgroup that much, every step falls back to the row-indexed walk
    // (c0 = -1: k_trsv_levels, 5-10x slower, same bits) instead of failing the factorisation.
    int dev = 0, lds_max = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess &&
              hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
              lds_max >= (int)PK_LDS_BYTES;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_trsv_packed<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)PK_LDS_BYTES) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_trsv_packed<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)PK_LDS_BYTES) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      for (Step &st : steps) st.c0 = -1;
    }
    return CASK_HIP_OK;
  }

  // The level-ordered copy for k_trsv_walk2 (see there): chunks of <= W2_SUB sub-levels (<= 64 consecutive rows of one
  // level) / W2_CH positions / W2_ECAP entries; per position the first two entries and the entry range; a source is
  // NEAR (the ring) when its producer is within PK_RING positions of the consumer's level end, else EARLY (memory),
  // which needs the producer W2_DIST chunks back.  A step that does not fit keeps d0 = -1.
  int build_walk2(const std::vector<int> &h_rp, const std::vector<int> &h_ci, const std::vector<double> &h_val,
                  const std::vector<int> &level, const std::vector<int> &lp, const std::vector<int> &ord) {
    std::vector<int> pos_of((size_t)n), peptr((size_t)n + 1, 0), ppos;
    for (int i = 0; i < n; i++) pos_of[ord[i]] = i;
    std::vector<double> pval, pdiag((size_t)n + 1, 1.0);
    ppos.reserve(h_ci.size());
    pval.reserve(h_ci.size());
    for (int i = 0; i < n; i++) {
      const int r = ord[i];
      for (int k = h_rp[r]; k < h_rp[r + 1]; k++) {
        const int c = h_ci[k];
        if (c == r) pdiag[i] = h_val[k];
        else if (lower ? c < r : c > r) {
          ppos.push_back(pos_of[c]);                          // the producer's POSITION: walk2 works in position space
          pval.push_back(h_val[k]);
        }
      }
      peptr[i + 1] = (int)ppos.size();
    }
    // where an entry's x value is: the ring (near: the producer is within PK_RING positions of the consumer's level end,
    // in this step) or memory (early: it must then be W2_DIST chunks back, checked below)
    std::vector<unsigned char> is_early(ppos.size() + 1, 0);
    std::vector<int> early_in_row((size_t)n + 1, 0);
    std::vector<int> step_lo((size_t)n, 0);
    for (const Step &st : steps)
      for (int i = st.lo; i < st.hi; i++) step_lo[i] = st.lo;
    for (int i = 0; i < n; i++) {
      const int level_end = lp[level[ord[i]] + 1];
      for (int e = peptr[i]; e < peptr[i + 1]; e++) {
        const int pp = ppos[e];
        is_early[e] = !(pp >= step_lo[i] && pp >= level_end - PK_RING);
        early_in_row[i] += is_early[e];
      }
    }
    // Slots of the entry arrays.  In a step of long rows a row's tail (what its record does not hold) is padded to a
    // multiple of W2_TAIL entries of value +0.0 reading the constant 0.0: the walker then takes a tail W2_TAIL entries at
    // a time without clamping an index or masking a product (fma(-0, 0, s) = s).
    std::vector<unsigned char> long_step((size_t)n, 0);
    for (const Step &st : steps)
      if (!st.wide && st.long_rows)
        for (int i = st.lo; i < st.hi; i++) long_step[i] = 1;
    std::vector<int> qptr((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) {
      const int ne = peptr[i + 1] - peptr[i];
      const int slots = (long_step[i] && ne > W2_Q) ? W2_Q + (ne - W2_Q + W2_TAIL - 1) / W2_TAIL * W2_TAIL : ne;
      qptr[i + 1] = qptr[i] + slots;
    }
    constexpr int OFF_X_IN_BUF = 56 * W2_RECS + 8 * W2_ECAP;   // (k_trsv_walk2's OFF_X)
    std::vector<int4> hdr;
    std::vector<W2Early> early_list;
    std::vector<W2Pos> apos((size_t)n + 1);
    std::vector<W2Ent> aent((size_t)qptr[n] + 1);
    for (int i = 0; i <= n; i++) {
      apos[i] = W2Pos{0.0, 0.0, 0.0, pdiag[i], (unsigned)W2_ZERO, (unsigned)W2_ZERO, (unsigned)W2_ZERO, 0u};
    }
    for (size_t e = 0; e < aent.size(); e++) aent[e] = W2Ent{0.0, (unsigned)W2_ZERO, 0u};
    std::vector<int> chunk_of((size_t)n, 0);
    for (Step &st : steps) {
      if (st.wide) continue;
      struct Chunk { int cs, ce, n_sub, n_early; unsigned char width[W2_SUB]; };
      std::vector<Chunk> chunks;
      bool ok = true;
      Chunk cur{st.lo, st.lo, 0, 0, {}};
      for (int i = st.lo; i < st.hi && ok;) {
        // the next sub-level: rows of i's level, at most 64, at most W2_ECAP entries
        const int level_end = lp[level[ord[i]] + 1];
        int e = i, early = 0;
        while (e < level_end && e - i < 64 && qptr[e + 1] - qptr[i] <= W2_ECAP &&
               early + early_in_row[e] <= W2_EARLY_PT * W2_ST) {
          early += early_in_row[e];
          e++;
        }
        if (e == i) { ok = false; break; }                    // one row with more than W2_ECAP entries (or early sources)
        const bool fits = cur.n_sub < W2_SUB && e - cur.cs <= W2_CH && qptr[e] - qptr[cur.cs] <= W2_ECAP &&
                          cur.n_early + early <= W2_EARLY_PT * W2_ST;
        if (!fits) {
          chunks.push_back(cur);
          cur = Chunk{i, i, 0, 0, {}};
        }
        cur.width[cur.n_sub++] = (unsigned char)(e - i);
        cur.n_early += early;
        cur.ce = e;
        i = e;
      }
      if (ok && cur.n_sub > 0) chunks.push_back(cur);
      if (!ok || chunks.empty()) continue;
      for (size_t k = 0; k < chunks.size(); k++)
        for (int i = chunks[k].cs; i < chunks[k].ce; i++) chunk_of[i] = (int)k;
      const size_t early0 = early_list.size();
      for (size_t k = 0; k < chunks.size() && ok; k++) {
        const Chunk &ch = chunks[k];
        for (int i = ch.cs; i < ch.ce && ok; i++) {
          const int ne = peptr[i + 1] - peptr[i], e0 = qptr[i] - qptr[ch.cs], ne_slots = qptr[i + 1] - qptr[i];
          unsigned word[W2_Q] = {(unsigned)W2_ZERO, (unsigned)W2_ZERO, (unsigned)W2_ZERO};
          for (int e = peptr[i]; e < peptr[i + 1]; e++) {
            const int pp = ppos[e], slot = e0 + (e - peptr[i]);
            unsigned src;
            if (!is_early[e]) src = (unsigned)(pp & (PK_RING - 1)) * 8u;
            else {
              if (pp >= st.lo && chunk_of[pp] > (int)k - W2_DIST) { ok = false; break; }   // (the static_assert rules it out)
              src = W2_IN_BUF | (unsigned)(OFF_X_IN_BUF + 8 * slot);
              early_list.push_back(W2Early{pp, slot});
            }
            aent[(size_t)qptr[ch.cs] + slot] = W2Ent{pval[e], src, 0u};
            if (e - peptr[i] < W2_Q) word[e - peptr[i]] = src;
          }
          W2Pos &q = apos[i];
          q.s0 = word[0]; q.s1 = word[1]; q.s2 = word[2];
          q.meta = (unsigned)ne_slots | ((unsigned)e0 << 11);   // (slots: the padded count in a step of long rows)
          if (ne > 0) q.v0 = pval[peptr[i]];
          if (ne > 1) q.v1 = pval[peptr[i] + 1];
          if (ne > 2) q.v2 = pval[peptr[i] + 2];
        }
      }
      if (!ok) {                                              // (cannot happen; the step runs k_trsv_levels_p)
        early_list.resize(early0);
        continue;
      }
      st.d0 = (int)(hdr.size() / 3);
      size_t early_at = early0;
      for (const Chunk &ch : chunks) {
        int w[6] = {0, 0, 0, 0, 0, 0}, ovf = 0, at = ch.cs;
        for (int q = 0; q < ch.n_sub; q++) {
          w[q >> 2] |= (int)ch.width[q] << (8 * (q & 3));
          for (int i = at; i < at + ch.width[q]; i++)
            if (peptr[i + 1] - peptr[i] > W2_Q) ovf |= 1 << q;   // a row with more entries than its record holds
          at += ch.width[q];
        }
        hdr.push_back(make_int4(ch.cs, (ch.ce - ch.cs) | (ch.n_early << 10) | (ovf ? W2_HAS_OVF : 0), qptr[ch.cs],
                                qptr[ch.ce] - qptr[ch.cs]));
        hdr.push_back(make_int4((int)early_at, w[0], w[1], w[2]));
        hdr.push_back(make_int4(w[3], w[4], w[5], ovf));
        early_at += ch.n_early;
      }
      st.d1 = (int)(hdr.size() / 3);
    }
    // The lane-group walk (trsv_lanes.hpp) for the runs with long rows -- every run that qualifies under
    // CASK_HIP_TRSV=lanes, none under CASK_HIP_TRSV=walk2.  Position space as above: the gather / scatter around the solve,
    // b, x and the diagonals are shared with walk2, which remains the run's fallback.
    std::vector<char> ln_bytes;
    std::vector<int> ln_words;
    if (mode != 4)
      for (Step &st : steps) {
        if (st.wide || !(st.long_rows || mode == 6)) continue;
        const size_t g0 = ln_words.size() / caskhip_lanes::LN_HDR_INTS;
        st.ge = caskhip_lanes::build_lanes_run(st.l0, st.l1, st.lo, lp, peptr, ppos, pval, ln_bytes, ln_words);
        if (st.ge) {
          st.g0 = (int)g0;
          st.g1 = (int)(ln_words.size() / caskhip_lanes::LN_HDR_INTS);
        }
      }
    pval.push_back(0.0);                                      // spare elements: the kernels' clamped loads
    ppos.push_back(0);
    early_list.push_back(W2Early{0, 0});
    if (hdr.empty()) hdr.push_back(make_int4(0, 0, 0, 0));
    PC_TRY(w2_pos.upload(apos));
    PC_TRY(w2_ent.upload(aent));
    PC_TRY(w2_early.upload(early_list));
    PC_TRY(w2_eptr.upload(peptr));                            // the position-space CSR of the wide levels' kernel
    PC_TRY(w2_epos.upload(ppos));
    PC_TRY(w2_eval.upload(pval));
    PC_TRY(w2_pdiag.upload(pdiag));
    PC_TRY(w2_bp.alloc((size_t)n + 1));
    PC_TRY(w2_xp.alloc((size_t)n + 1));
    PC_TRY(w2_progress.alloc(1));
    PC_TRY(w2_hdr.upload(hdr));
    if (std::getenv("CASK_HIP_TRSV_STATS")) {               // cycles per chunk by role, printed after every solve
      PC_TRY(w2_dbg.alloc(12));
      PC_TRY(hipMemset(w2_dbg.p, 0, 96));
      std::fprintf(stderr, "walk2 %s: %zu chunks, %zu x values read from memory\n", lower ? "L" : "U", hdr.size() / 3, early_list.size() - 1);
    }
    int dev = 0, lds_max = 0;
    w2_ok = hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
            lds_max >= (int)W2_LDS_BYTES;
    w2_ok = w2_ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_trsv_walk2<false, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES) == hipSuccess;
    w2_ok = w2_ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_trsv_walk2<false, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES) == hipSuccess;
    w2_ok = w2_ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_trsv_walk2<true, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES) == hipSuccess;
    w2_ok = w2_ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_trsv_walk2<true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES) == hipSuccess;
    if (!w2_ok) {
      (void)hipGetLastError();
      for (Step &st : steps) st.d0 = -1;
    }
    bool ln_ok = !ln_words.empty() && lds_max >= (int)caskhip_lanes::LN_LDS_BYTES && lanes_kernels_usable(w2_progress.p);
    if (ln_ok) {
      std::vector<double> rdiag(pdiag.size());
      for (size_t i = 0; i < pdiag.size(); i++) rdiag[i] = 1.0 / pdiag[i];
      PC_TRY(ln_lanes.upload(ln_bytes));
      PC_TRY(ln_hdr.upload(ln_words));
      PC_TRY(ln_rdiag.upload(rdiag));
    } else {
      (void)hipGetLastError();
      for (Step &st : steps) st.g0 = -1;
    }
    if (std::getenv("CASK_HIP_TRSV_STATS")) {
      size_t runs = 0, chunks = 0;
      int e_max = 0;
      for (const Step &st : steps)
        if (st.g0 >= 0) { runs++; chunks += st.g1 - st.g0; e_max = std::max(e_max, st.ge); }
      std::fprintf(stderr, "lanes %s: %zu runs of narrow levels, %zu chunks (%.1f MB), up to %d entries per lane\n", lower ? "L" : "U", runs,
                   chunks, ln_bytes.size() / 1e6, e_max);
    }
    return CASK_HIP_OK;
  }

  // Every instantiation of the lane-group walk, by (unit diagonal, entries per lane).
  static const void *lanes_kernel(bool unit, int e) {
    using namespace caskhip_lanes;
#define CASK_LN_K(U, E) reinterpret_cast<const void *>(&k_trsv_lanes<U, E>)
#define CASK_LN_E(U) (e == 4 ? CASK_LN_K(U, 4) : e == 8 ? CASK_LN_K(U, 8) : CASK_LN_K(U, 16))
    return unit ? CASK_LN_E(true) : CASK_LN_E(false);
#undef CASK_LN_E
#undef CASK_LN_K
  }
  // Once per process: may the lane-group kernels be used at all?  Their LDS attribute must be settable and -- the walker
  // uses ABSOLUTE LDS byte addresses the host wrote (trsv_lanes.hpp: ring at 0, LN_ZERO, LN_BUF0) -- their dynamic LDS must
  // start at address 0: every instantiation is launched once with an empty chunk range, in which it only compares its LDS
  // base with 0 and reports through the progress word.  Anything else sends every run to walk2.
  static bool lanes_kernels_usable(int *d_progress) {
    static int state = -1;                                      // (factors are built under the caller's serialisation)
    if (state >= 0) return state == 1;
    using namespace caskhip_lanes;
    bool ok = true;
    const LanesTri none{nullptr, nullptr, nullptr};
    for (int unit = 0; unit < 2 && ok; unit++)
      for (int e : {4, 8, 16}) {
        if (!ok) break;
        {
          const void *fn = lanes_kernel(unit != 0, e);
          ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LN_LDS_BYTES) == hipSuccess;
          if (!ok) break;
          int word = 0, zero = 0;
          const double *bp = nullptr;
          double *xp = nullptr;
          unsigned long long *dbg = nullptr;
          LanesTri t = none;
          void *args[] = {&t, &zero, &zero, &bp, &xp, &d_progress, &dbg};
          ok = hipMemset(d_progress, 0, sizeof(int)) == hipSuccess &&
               hipLaunchKernel(fn, dim3(1), dim3(LN_T), args, LN_LDS_BYTES, nullptr) == hipSuccess &&
               hipMemcpy(&word, d_progress, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && word != LN_BAD_LDS_BASE;
          if (!ok && word == LN_BAD_LDS_BASE)
            std::fprintf(stderr, "cask_hip: k_trsv_lanes' dynamic LDS does not start at address 0 in this build: the lane-group "
                                 "walk is off, runs of long-row levels use walk2\n");
        }
      }
    if (!ok) (void)hipGetLastError();
    state = ok ? 1 : 0;
    return ok;
  }

  // Schedules of a run of narrow levels.  walk2 (walker + stagers + read-ahead in position space, r3), packed (the
  // four-wave walk, r2) and levels (the row-indexed walk, r1) walk every row in stored order and are bit-identical to
  // each other (tested); lanes (r5, the lane-group walk for runs with long rows) adds a row's products lane group by
  // lane group and multiplies by a reciprocal diagonal: the same solution to rounding.  The default is lanes where a
  // run's rows average more than 6 entries, walk2 elsewhere.  CASK_HIP_TRSV = levels | walk2 | packed | lanes forces
  // one; any other value is an error on stderr and the default schedule (r6, ADVICE r5: `walk1` and `syncfree`, removed
  // in round 5 as measured losses -- docs/experiments.md -- used to select the default without a word).
  static int forced_mode() {
    const char *force = std::getenv("CASK_HIP_TRSV");
    if (!force || !*force) return 0;
    const std::string f(force);
    const int mode = f == "levels" ? 2 : f == "walk2" ? 4 : f == "packed" || f == "packed4" ? 5 : f == "lanes" ? 6 : f == "default" ? 0 : -1;
    if (mode < 0) {
      static bool warned = false;
      if (!warned)
        std::fprintf(stderr, "cask_hip: CASK_HIP_TRSV=%s is not a schedule (levels | packed | walk2 | lanes | default; walk1 and "
                             "syncfree were removed in round 5): running the default schedule\n", force);
      warned = true;
      return 0;
    }
    return mode;
  }

  int solve(const double *d_b, double *d_x, hipStream_t s) const {
    const int flags = (lower ? 1 : 0) | (unit ? 2 : 0);
    if (walk2 && n > 0 && w2_bp.p) {                            // the whole solve in position space
      const Walk2Tri w2{w2_pos.p, w2_ent.p, w2_early.p, w2_hdr.p};
      const caskhip_lanes::LanesTri ln{ln_lanes.p, ln_hdr.p, ln_rdiag.p};
      const int pg = (int)std::min<int64_t>(2048, ((int64_t)n + 255) / 256), u = unit ? 1 : 0;
      PC_TRY(hipMemsetAsync(w2_progress.p, 0, sizeof(int), s));
      hipLaunchKernelGGL(k_w2_gather, dim3(pg), dim3(256), 0, s, n, order.p, d_b, w2_bp.p);
      for (const Step &st : steps) {
        if (st.wide)
          hipLaunchKernelGGL(k_trsv_level_p, dim3((st.hi - st.lo + 255) / 256), dim3(256), 0, s, st.lo, st.hi, u, w2_eptr.p,
                             w2_epos.p, w2_eval.p, w2_pdiag.p, w2_bp.p, w2_xp.p);
        else if (st.g0 >= 0) {
          using namespace caskhip_lanes;
          const LanesTri t = ln;
          const int c0 = st.g0, c1 = st.g1;
          const double *bp = w2_bp.p;
          double *xp = w2_xp.p;
          int *progress = w2_progress.p;
          unsigned long long *dbg = w2_dbg.p;
          void *args[] = {const_cast<LanesTri *>(&t), const_cast<int *>(&c0), const_cast<int *>(&c1), &bp, &xp, &progress, &dbg};
          PC_TRY(hipLaunchKernel(lanes_kernel(unit, st.ge), dim3(LN_GRID), dim3(LN_T), args, LN_LDS_BYTES, s));
        }
        else if (st.d0 < 0)
          hipLaunchKernelGGL(k_trsv_levels_p, dim3(1), dim3(TRSV_WG), 0, s, st.l0, st.l1, u, level_ptr.p, w2_eptr.p, w2_epos.p,
                             w2_eval.p, w2_pdiag.p, w2_bp.p, w2_xp.p);
        else if (st.long_rows && unit)
          hipLaunchKernelGGL((k_trsv_walk2<true, true>), dim3(W2_GRID), dim3(W2_T), W2_LDS_BYTES, s, w2, st.d0, st.d1, w2_bp.p, w2_xp.p, w2_progress.p, w2_dbg.p);
        else if (st.long_rows)
          hipLaunchKernelGGL((k_trsv_walk2<true, false>), dim3(W2_GRID), dim3(W2_T), W2_LDS_BYTES, s, w2, st.d0, st.d1, w2_bp.p, w2_xp.p, w2_progress.p, w2_dbg.p);
        else if (unit)
          hipLaunchKernelGGL((k_trsv_walk2<false, true>), dim3(W2_GRID), dim3(W2_T), W2_LDS_BYTES, s, w2, st.d0, st.d1, w2_bp.p, w2_xp.p, w2_progress.p, w2_dbg.p);
        else
          hipLaunchKernelGGL((k_trsv_walk2<false, false>), dim3(W2_GRID), dim3(W2_T), W2_LDS_BYTES, s, w2, st.d0, st.d1, w2_bp.p, w2_xp.p, w2_progress.p, w2_dbg.p);
      }
      hipLaunchKernelGGL(k_w2_scatter, dim3(pg), dim3(256), 0, s, n, order.p, w2_xp.p, d_x);
      PC_TRY(hipGetLastError());
      if (w2_dbg.p) {
        unsigned long long h[12];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, w2_dbg.p, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipMemset(w2_dbg.p, 0, sizeof(h));
        if (h[3])
          std::fprintf(stderr, "walk %s: chunks %llu | cycles/chunk: walker at the barrier %.0f, walking %.0f | stagers at the barrier %.0f, working %.0f (load wait %.0f, stage %.0f, issue loads %.0f, write back %.0f)\n",
                       lower ? "L" : "U", h[3], (double)h[0] / h[3], (double)h[1] / h[3], (double)h[4] / h[3], (double)h[6] / h[3],
                       (double)h[8] / h[3], (double)h[9] / h[3], (double)h[10] / h[3], (double)h[11] / h[3]);
      }
      return CASK_HIP_OK;
    }
    // the packed walk reads b a chunk ahead of the x it writes: not for an in-place solve
    const bool packed_ok = mode != 2 && d_b != d_x;
    const PackedTri pk{pk_row.p, pk_eptr.p, pk_seg.p, pk_hdr.p, pk_code.p, pk_diag.p, pk_val.p};
    for (const Step &st : steps) {
      if (st.wide)
        hipLaunchKernelGGL(k_trsv_level, dim3((st.hi - st.lo + 255) / 256), dim3(256), 0, s, st.lo, st.hi, flags,
                           order.p, rp.p, ci.p, val.p, d_b, d_x);
      else if (packed_ok && st.c0 >= 0 && st.long_rows)
        hipLaunchKernelGGL(k_trsv_packed<true>, dim3(1), dim3(PK_T), PK_LDS_BYTES, s, pk, st.c0, st.c1, unit ? 1 : 0, d_b, d_x);
      else if (packed_ok && st.c0 >= 0)
        hipLaunchKernelGGL(k_trsv_packed<false>, dim3(1), dim3(PK_T), PK_LDS_BYTES, s, pk, st.c0, st.c1, unit ? 1 : 0, d_b, d_x);
      else
        hipLaunchKernelGGL(k_trsv_levels, dim3(1), dim3(TRSV_WG), 0, s, st.l0, st.l1, flags, level_ptr.p, order.p,
                           rp.p, ci.p, val.p, d_b, d_x);
    }
    PC_TRY(hipGetLastError());
    return CASK_HIP_OK;
  }
};

// rows [keep entries with (lower ? c <= r : c >= r)] of a CSR matrix
void extract_triangle(int n, const int *rp, const int *ci, const double *val, bool lower, std::vector<int> &trp,
                      std::vector<int> &tci, std::vector<double> &tval) {
  trp.assign((size_t)n + 1, 0);
  tci.clear();
  tval.clear();
  for (int r = 0; r < n; r++) {
    for (int k = rp[r]; k < rp[r + 1]; k++)
      if (lower ? ci[k] <= r : ci[k] >= r) {
        tci.push_back(ci[k]);
        tval.push_back(val[k]);
      }
    trp[r + 1] = (int)tci.size();
  }
}

int check_sorted_csr(int32_t n, int64_t nnz, const int32_t *rp, const int32_t *ci) {
  if (n < 0 || nnz < 0 || !rp || (nnz > 0 && !ci)) return report_failure(CASK_HIP_ERR_INVALID, "bad CSR arguments");
  if (rp[0] != 0 || rp[n] != nnz) return report_failure(CASK_HIP_ERR_INVALID, "row_ptr does not span the nonzeros");
  for (int r = 0; r < n; r++) {
    if (rp[r + 1] < rp[r]) return report_failure(CASK_HIP_ERR_INVALID, "row_ptr must be non-decreasing");
    for (int k = rp[r]; k < rp[r + 1]; k++) {
      if (ci[k] < 0 || ci[k] >= n) return report_failure(CASK_HIP_ERR_INVALID, "column index out of range");
      if (k > rp[r] && ci[k] <= ci[k - 1])
        return report_failure(CASK_HIP_ERR_INVALID, "columns must be strictly ascending within a row");
    }
  }
  return CASK_HIP_OK;
}

}  // namespace

// (Rounds 3-5 had a multicolour ILU(0) here -- CASK_HIP_PRECOND_ILU0_MC: greedy colouring, 2 x colours wide launches per
// application, the whole PCG in colour order.  NOT the reference's factors, and behind Jacobi end to end on the system it
// was built for (138 passes x 112 us = 15.5 ms against 257 x 48 us = 12.5 ms on G3_circuit-like): removed in ABI 7,
// docs/experiments.md.)

struct cask_hip_precond {
  int kind = 0, n = 0;
  int device = 0;
  std::vector<double> factored;        // ILU0: the factored values in the input pattern (pc of the reference)
  TriFactor L, U;
  DevBuf<double> dinv, tmp, d_r, d_z;  // Jacobi: 1/diag ; ILU0: the intermediate vector ; staging for host vectors
};

extern "C" {

int cask_hip_precond_create(int32_t kind, int32_t n, int64_t nnz, const int32_t *row_ptr, const int32_t *col_ind,
                            const double *values, cask_hip_precond **out) {
  if (!out) return report_failure(CASK_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (kind == CASK_HIP_PRECOND_ILU0_MC_REMOVED)
    return report_failure(CASK_HIP_ERR_INVALID, "the multicolour ILU(0) (kind 4) was removed in ABI 7: not the reference's factors and "
                                                "behind Jacobi end to end (docs/experiments.md); use CASK_HIP_PRECOND_JACOBI or _ILU0_UNIT");
  if (kind != CASK_HIP_PRECOND_JACOBI && kind != CASK_HIP_PRECOND_ILU0 && kind != CASK_HIP_PRECOND_ILU0_UNIT)
    return report_failure(CASK_HIP_ERR_INVALID, "unknown preconditioner kind");
  int rc = check_sorted_csr(n, nnz, row_ptr, col_ind);
  if (rc) return rc;
  if (nnz > 0 && !values) return report_failure(CASK_HIP_ERR_INVALID, "values is NULL");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return report_failure(CASK_HIP_ERR_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
  std::unique_ptr<cask_hip_precond> p(new cask_hip_precond);
  p->kind = kind;
  p->n = n;
  PC_TRY(hipGetDevice(&p->device));
  if (kind == CASK_HIP_PRECOND_JACOBI) {
    std::vector<double> dinv((size_t)n, 1.0);               // a row without a diagonal entry is left unscaled
    for (int r = 0; r < n; r++)
      for (int k = row_ptr[r]; k < row_ptr[r + 1]; k++)
        if (col_ind[k] == r && values[k] != 0.0) dinv[r] = 1.0 / values[k];
    PC_TRY(p->dinv.upload(dinv));
    *out = p.release();
    return CASK_HIP_OK;
  }
  // ILU(0), the IKJ sweep of ILUPreconditioner (SparseLinearSolvers.hpp:92-113): for every row i and
  // every stored (i,k) with k < i in ascending k: if (k,k) is stored, a[i][k] /= a[k][k], and every
  // stored (i,j), j > k, with (k,j) stored gets a[i][j] -= a[k][j] * a[i][k].
  std::vector<double> &a = p->factored;
  a.assign(values, values + nnz);
  std::vector<int> diag((size_t)n, -1);
  for (int r = 0; r < n; r++)
    for (int k = row_ptr[r]; k < row_ptr[r + 1]; k++)
      if (col_ind[k] == r) diag[r] = k;
  for (int i = 1; i < n; i++) {
    for (int kk = row_ptr[i]; kk < row_ptr[i + 1]; kk++) {
      const int k = col_ind[kk];
      if (k >= i) break;
      if (diag[k] < 0 || a[diag[k]] == 0.0) continue;        // !isNnz(k, k): absent or a stored zero (SparseMatrix.hpp:219-225)
      a[kk] = a[kk] / a[diag[k]];
      const double beta = a[kk];
      int pi = kk + 1, pk = diag[k] + 1;                     // (i, j > k) against (k, j > k), both ascending
      const int ei = row_ptr[i + 1], ek = row_ptr[k + 1];
      while (pi < ei && pk < ek) {
        if (col_ind[pi] < col_ind[pk]) pi++;
        else if (col_ind[pi] > col_ind[pk]) pk++;
        else {
          if (a[pk] != 0.0) a[pi] = a[pi] - a[pk] * beta;     // isNnz(k, j) tests the value, not the pattern (:107)
          pi++;
          pk++;
        }
      }
    }
  }
  std::vector<int> trp, tci;
  std::vector<double> tval;
  extract_triangle(n, row_ptr, col_ind, a.data(), true, trp, tci, tval);
  p->L.unit = kind == CASK_HIP_PRECOND_ILU0_UNIT;            // the textbook L has a unit diagonal; the reference's does not
  rc = p->L.build(n, true, trp, tci, tval);
  if (rc) return rc;
  extract_triangle(n, row_ptr, col_ind, a.data(), false, trp, tci, tval);
  rc = p->U.build(n, false, trp, tci, tval);
  if (rc) return rc;
  PC_TRY(p->tmp.alloc((size_t)n));
  *out = p.release();
  return CASK_HIP_OK;
}

int cask_hip_precond_destroy(cask_hip_precond *p) {
  delete p;
  return CASK_HIP_OK;
}

int cask_hip_precond_factor_values(const cask_hip_precond *p, double *values_out) {
  if (!p || !values_out) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (p->kind == CASK_HIP_PRECOND_JACOBI)
    return report_failure(CASK_HIP_ERR_INVALID, "only the ILU0 kinds keep factor values in the input pattern");
  if (!p->factored.empty()) std::memcpy(values_out, p->factored.data(), p->factored.size() * sizeof(double));
  return CASK_HIP_OK;
}

int cask_hip_precond_info(const cask_hip_precond *p, int32_t *levels_lower, int32_t *levels_upper,
                          int32_t *launches_per_apply) {
  if (!p) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  const bool ilu = p->kind != CASK_HIP_PRECOND_JACOBI;
  if (levels_lower) *levels_lower = ilu ? p->L.n_levels : 0;
  if (levels_upper) *levels_upper = ilu ? p->U.n_levels : 0;
  if (launches_per_apply) *launches_per_apply = ilu ? (int32_t)(p->L.steps.size() + p->U.steps.size()) : 1;
  return CASK_HIP_OK;
}

int cask_hip_precond_apply_device(cask_hip_precond *p, const double *d_r, double *d_z, void *stream) {
  if (!p || (p->n > 0 && (!d_r || !d_z))) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (p->n == 0) return CASK_HIP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p->kind == CASK_HIP_PRECOND_JACOBI) {
    const int grid = (int)std::min<int64_t>(1024, ((int64_t)p->n + 255) / 256);
    hipLaunchKernelGGL(k_scale, dim3(grid), dim3(256), 0, s, (int64_t)p->n, p->dinv.p, d_r, d_z);
    PC_TRY(hipGetLastError());
    return CASK_HIP_OK;
  }
  // z = U^-1 (L^-1 r)   (ILUPreconditioner::apply, SparseLinearSolvers.hpp:143-151)
  int rc = p->L.solve(d_r, p->tmp.p, s);
  if (rc) return rc;
  return p->U.solve(p->tmp.p, d_z, s);
}

int cask_hip_precond_apply(cask_hip_precond *p, const double *r, double *z) {
  if (!p || (p->n > 0 && (!r || !z))) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (p->n == 0) return CASK_HIP_OK;
  PC_TRY(hipSetDevice(p->device));
  if (!p->d_r.p) PC_TRY(p->d_r.alloc((size_t)p->n));
  if (!p->d_z.p) PC_TRY(p->d_z.alloc((size_t)p->n));
  PC_TRY(hipMemcpy(p->d_r.p, r, (size_t)p->n * sizeof(double), hipMemcpyHostToDevice));
  int rc = cask_hip_precond_apply_device(p, p->d_r.p, p->d_z.p, nullptr);
  if (rc) return rc;
  PC_TRY(hipMemcpy(z, p->d_z.p, (size_t)p->n * sizeof(double), hipMemcpyDeviceToHost));
  return CASK_HIP_OK;
}

int cask_hip_trsolve(int32_t n, int64_t nnz, const int32_t *row_ptr, const int32_t *col_ind, const double *values,
                     int32_t lower, const double *rhs, double *x) {
  int rc = check_sorted_csr(n, nnz, row_ptr, col_ind);
  if (rc) return rc;
  if ((nnz > 0 && !values) || (n > 0 && (!rhs || !x))) return report_failure(CASK_HIP_ERR_INVALID, "NULL argument");
  if (n == 0) return CASK_HIP_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return report_failure(CASK_HIP_ERR_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
  std::vector<int> trp, tci;
  std::vector<double> tval;
  extract_triangle(n, row_ptr, col_ind, values, lower != 0, trp, tci, tval);
  TriFactor t;
  rc = t.build(n, lower != 0, trp, tci, tval);
  if (rc) return rc;
  DevBuf<double> b, out;
  PC_TRY(b.upload(rhs, (size_t)n));
  PC_TRY(out.alloc((size_t)n));
  rc = t.solve(b.p, out.p, nullptr);
  if (rc) return rc;
  PC_TRY(hipMemcpy(x, out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return CASK_HIP_OK;
}

}  // extern "C"

int cask_hip_precond_rows(const cask_hip_precond *p) { return p ? p->n : -1; }

const double *cask_hip_precond_jacobi_scale(const cask_hip_precond *p) {
  return p && p->kind == CASK_HIP_PRECOND_JACOBI && p->n > 0 ? p->dinv.p : nullptr;
}
