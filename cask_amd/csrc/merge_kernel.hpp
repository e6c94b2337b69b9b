// The workgroup-level merge kernel (variant MERGE), the headline kernel of the engine.  Kept in its own
// header because its instantiations are what takes time to compile: merge_ipt*.hip instantiate one
// items-per-thread value each, in parallel.
#pragma once
#include "diag.hpp"
#include "spmv_common.hpp"

namespace caskhip {

// ------------------------------------------------------------- merge variant
// Merge-based family.  The host cuts the merge path of (row ends) x (nonzero
// indices) into pieces of at most CAP = blockDim.x*IPT items and snaps each
// cut to a row boundary, so every workgroup owns whole rows and the same
// amount of work whatever the row-length distribution; rows longer than a
// threshold become their own "long row" pieces.  Phase 1 streams the block's
// nonzeros with 16-byte loads in nonzero order (perfectly coalesced, IPT
// elements in flight per lane, independent of row structure) and parks the
// products in LDS; phase 2 sums each row's run of products with G lanes per
// row (G chosen per block from its mean row length) and a DPP butterfly.
// Rows longer than skew_short_max(G) products are left to a second pass in which 16 lanes or a whole wave sum one row
// (power-law blocks: a 500-nonzero row among 3-nonzero rows would otherwise keep one lane busy for
// microseconds while 255 idle).  The host flags such blocks (KIND_SKEW); others pay one compare.

template <int G, int EXT>
__device__ __forceinline__ double reduce_rows_plain(const BlockDesc &d, const double *prod, const int *roff,
                                                    double *__restrict__ y, const double *w) {
  const int tid = threadIdx.x;
  const int rows_per_pass = blockDim.x / G;
  const int j = tid & (G - 1);
  double dsum = 0.0;
  for (int r0 = 0; r0 < d.n_rows; r0 += rows_per_pass) {
    const int r = r0 + tid / G;
    double acc = 0.0;
    if (r < d.n_rows) {
      const int s = roff[r], e = roff[r + 1];
#pragma unroll 4
      for (int k = s + j; k < e; k += G) acc += prod[k];
    }
    acc = group_sum<G>(acc);
    if (j == 0 && r < d.n_rows) {
      y[d.row_start + r] = acc;
      if (EXT && w) dsum = fma(w[r], acc, dsum);              // launch-uniform; w = the block's slice, in LDS
    }
  }
  return dsum;
}

template <int G, bool SKEW, int EXT>
__device__ __forceinline__ double reduce_rows(const BlockDesc &d, const double *prod, const int *roff,
                                              double *__restrict__ y, const double *w) {
  const bool skew = SKEW && (d.kind_g & KIND_SKEW);           // workgroup-uniform
  if (!skew)                                                  // the common case keeps the lean loop
    return reduce_rows_plain<G, EXT>(d, prod, roff, y, w);
  const int tid = threadIdx.x;
  const int rows_per_pass = blockDim.x / G;
  const int j = tid & (G - 1);
  double dsum = 0.0;
  // Three classes of row (power-law blocks: ~680 rows on 256 threads, most with 1-2 products, a few with hundreds):
  //   short  (<= skew_short_max(G) products)  G lanes each, all rows of the block in parallel -- pass 1;
  //   medium (<= SKEW_MED_MAX, only when G <= 2)  16 lanes each, four rows per wave at a time   -- pass 2;
  //   long   a whole wave each                                                                  -- pass 2.
  // One lane walking a 30-product row while its 63 neighbours wait for it was 2 of the 3 us this phase took on the
  // webbase-like matrix (profiles/r02_webbase_anatomy.txt).
  constexpr int SHORT_MAX = skew_short_max(G);
  constexpr int MED_MAX = G <= 2 ? SKEW_MED_MAX : SHORT_MAX;
  for (int r0 = 0; r0 < d.n_rows; r0 += rows_per_pass) {
    const int r = r0 + tid / G;
    double acc = 0.0;
    bool mine = r < d.n_rows;
    if (mine) {
      const int s = roff[r], e = roff[r + 1];
      if (e - s > SHORT_MAX) {
        mine = false;
      } else {
#pragma unroll 4
        for (int k = s + j; k < e; k += G) acc += prod[k];
      }
    }
    acc = group_sum<G>(acc);
    if (j == 0 && mine) {
      y[d.row_start + r] = acc;
      if (EXT && w) dsum = fma(w[r], acc, dsum);
    }
  }
  // pass 2: every wave takes the 64-row chunks c = wave, wave + n_waves, ... ; one ballot per class and chunk tells
  // it which rows are left.  Which lanes sum which row -- and with it the order of every floating-point addition --
  // is a function of the matrix alone (no queue, no atomics).
  const int lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  for (int c0 = wave * 64; c0 < d.n_rows; c0 += n_waves * 64) {   // wave-uniform
    const int r = c0 + lane;
    const int len = r < d.n_rows ? roff[r + 1] - roff[r] : 0;
    unsigned long long med = __ballot(len > SHORT_MAX && len <= MED_MAX);
    unsigned long long todo = __ballot(len > MED_MAX);
    while (med) {                                             // four medium rows at a time, 16 lanes each
      int pick[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        pick[g] = med ? __builtin_ctzll(med) : -1;
        med &= med - 1;                                       // (0 stays 0)
      }
      const int gi = lane >> 4, gl = lane & 15;
      const int b = gi == 0 ? pick[0] : gi == 1 ? pick[1] : gi == 2 ? pick[2] : pick[3];
      double acc = 0.0;
      if (b >= 0) {
        const int s = roff[c0 + b], e = roff[c0 + b + 1];
#pragma unroll 4
        for (int k = s + gl; k < e; k += 16) acc += prod[k];
      }
      acc = group_sum<16>(acc);
      if (gl == 0 && b >= 0) {
        y[d.row_start + c0 + b] = acc;
        if (EXT && w) dsum = fma(w[c0 + b], acc, dsum);
      }
    }
    while (todo) {                                            // a whole wave per long row
      const int b = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int row = c0 + b;
      const int s = roff[row], e = roff[row + 1];
      double a0 = 0.0, a1 = 0.0;
      int k = s + lane;
      for (; k + 64 < e; k += 128) {
        a0 += prod[k];
        a1 += prod[k + 64];
      }
      if (k < e) a0 += prod[k];
      const double acc = group_sum<64>(a0 + a1);
      if (lane == 0) {
        y[d.row_start + row] = acc;
        if (EXT && w) dsum = fma(w[row], acc, dsum);
      }
    }
  }
  return dsum;
}

// One block of the merge kernel, straight-line so that hipcc can count the
// outstanding loads exactly.  Issue order is the point (membench2, stage 3 vs
// 4: 11.0 -> 9.2 us on the cant payload):
//   1. the block's x window (XU 8-byte loads per lane)      -- oldest
//   2. its row offsets (2 loads per lane)
//   3. the value/index stream (IPT/2 16-byte + 8-byte loads) -- youngest
// vmcnt retires in order, so parking the window and the offsets in LDS waits
// only for (1) and (2) while the stream is still in flight; when the stream
// lands the gathers are LDS reads (~100 ns, no TA traffic) instead of a second
// dependent trip to L2.  XU = 0 gathers from L2 (window wider than the tile).
//
// WIDE (with C12, XU >= 2): all tiled blocks of the plan have one contiguous window; it is loaded in pairs.
// C12 (with C16, IPT = 8): the same slots, 12 bits each, packed per thread by the planner (see the load).
// C16 (only with an x window): the block's column indices are read from the
// 16-bit side array (col_ind - cmin, built at plan time), 2 instead of 4 bytes
// per nonzero and already LDS offsets -- the stream shrinks from 12 to 10
// bytes per nonzero, which on a bandwidth-bound kernel is the whole game.
//
// Sharded product (SEAM = true, only for blocks that read halo columns; cask_hip_p2p.h): the block
// fetches the address-table entries of its window first, issues its stream like any block, then the
// window loads themselves -- some of them remote: one local and one xGMI round trip, overlapped with
// the stream.  A separate instantiation, so the code of every other block is exactly the one above.
// Halo entries live in a peer GPU's memory and change between products of a solver: they are read with
// system-scope loads (sc0 sc1: served by the owner's memory, never by a line this GPU's L2 kept from the
// previous product), the owner writes them through at system scope (SolverPass::sys_scope, blas1 kernels) and
// a collective orders the two -- see DESIGN.md section 7 for the whole argument.
typedef __attribute__((address_space(1))) const double gdouble;
__device__ __forceinline__ double load_at(uint64_t addr) {
  return __hip_atomic_load(reinterpret_cast<gdouble *>(addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_own(double *p, double v, int sys_scope) {
  if (sys_scope) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else *p = v;
}

// The SolverPass travels only with the kernels that use it.
template <int EXT> struct PassArg {};
template <> struct PassArg<2> { SolverPass sp; };
template <int EXT> __device__ __forceinline__ SolverPass pass_of(const PassArg<EXT> &) { return SolverPass{}; }
template <> __device__ __forceinline__ SolverPass pass_of<2>(const PassArg<2> &a) { return a.sp; }

// What a solver pass (EXT == 2) knows once the scalars of the previous pass are summed.
struct PassScalars {
  double beta, alpha_prev;
  bool stop;                 // converged in the previous pass, or a final_only launch: no product
  bool update_x;             // the solution update of the previous pass is still owed
};

// Every workgroup derives the same scalars from the same partial sums (fixed order => identical bits), so every
// workgroup takes the same decision; workgroup 0 records it.  `red` = 16 doubles of LDS.
__device__ __forceinline__ PassScalars pass_scalars(const SolverPass &sp, int lb, double *red) {
  PassScalars ps;
  ps.beta = 0.0;
  ps.alpha_prev = 0.0;
  ps.stop = false;
  ps.update_x = false;
  if (sp.first) return ps;
  if (diag::NO_PARTIALS) {                                    // diagnostic build: what do the partial sums cost?
    ps.beta = 0.5;
    ps.update_x = sp.xsol != nullptr;
    return ps;
  }
  const double chk = partials_or_scalar(sp.part_chk, sp.n_chk, red);
  ps.alpha_prev = *sp.alpha_prev;
  ps.update_x = sp.xsol != nullptr;
  if (chk <= sp.tol2) {                                       // SparseLinearSolvers.hpp:220-226: nothing after the test
    if (lb == 0 && threadIdx.x == 0 && !sp.secondary) {
      if (sp.part_num == sp.part_chk && sp.num_out) *sp.num_out = chk;
      *sp.done = done_tag(sp.iter);                          // tagged: see blas1_kernels.hpp, done_by_earlier_launch
    }
    ps.stop = true;
    return ps;
  }
  const double num = sp.part_num == sp.part_chk ? chk : partials_or_scalar(sp.part_num, sp.n_num, red);
  ps.beta = num / *sp.den;
  if (lb == 0 && threadIdx.x == 0 && !sp.secondary) {
    if (sp.num_out) *sp.num_out = num;
    *sp.iters = sp.iter - 1;                                  // :231, the previous pass did not converge
  }
  ps.stop = sp.final_only != 0;                               // 1: the solve ends here (the update is still applied);
  if (sp.final_only == 2) ps.update_x = false;                // 2: a probe at a checkpoint -- the next pass applies it
  return ps;
}

// Own-row work of a solver pass for the rows [row0, row0 + n) of a piece that does not go through merge_load
// (long-row pieces, zero-fill pieces, and every block when the pass stops): the solution update the previous
// pass owes and, unless the pass stops, the new direction.  Returns nothing; w for long rows is recomputed.
__device__ __forceinline__ void pass_own_rows(const SolverPass &sp, const PassScalars &ps, const double *a, int row0,
                                              int n, bool store_dir) {
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    const int row = row0 + r;
    const double bv = a[row + sp.b_off];
    if (ps.update_x) sp.xsol[row] = fma(ps.alpha_prev, bv, sp.xsol[row]);
    if (store_dir && sp.b_new) store_own(sp.b_new + row, fma(ps.beta, bv, a[row]), sp.sys_scope);
  }
}

// Chunked tiles: the first columns of the 16-column chunks a WAVE loads -- 4 lane groups x XU turns, consecutive in the
// block's table ([tid >> 4][u]: plan::build_chunk_table).  merge_workgroup fetches them with scalar loads NEXT TO the
// descriptor's (one wait for both: behind the descriptor they were a second dependent trip at the head of every
// workgroup, +5-7 % on the atmosmodd-like launch); a lane picks its group's entry with masks, not selects -- selects of
// loaded values made the compiler branch per lane group and wait in every branch.
template <int XU>
struct ChunkTab {
  static constexpr int N = XU > 0 ? XU : 1;
  int e[4 * N];
};
template <int XU>
__device__ __forceinline__ void chunk_table_fetch(const int *__restrict__ my_chunks, int tid, ChunkTab<XU> &tab) {
  constexpr int N = ChunkTab<XU>::N;
  const int *t = my_chunks + __builtin_amdgcn_readfirstlane(tid >> 6) * 4 * N;   // wave-uniform: scalar loads
#pragma unroll
  for (int i = 0; i < 4 * N; i++) tab.e[i] = t[i];
}
// (keeps the fetch where it was written: the values exist here, i.e. the loads were issued above)
template <int XU>
__device__ __forceinline__ void chunk_table_pin(ChunkTab<XU> &tab) {
  constexpr int N = ChunkTab<XU>::N;
#pragma unroll
  for (int i = 0; i < 4 * N; i += 4)
    asm volatile("" : "+s"(tab.e[i]), "+s"(tab.e[i + 1]), "+s"(tab.e[i + 2]), "+s"(tab.e[i + 3]));
}
template <int XU>
__device__ __forceinline__ void chunk_starts_of(const ChunkTab<XU> &tab, int tid, int (&st)[XU > 0 ? XU : 1]) {
  constexpr int N = ChunkTab<XU>::N;
  const int g = (tid >> 4) & 3;
  const int m0 = -(int)(g == 0), m1 = -(int)(g == 1), m2 = -(int)(g == 2), m3 = -(int)(g == 3);
#pragma unroll
  for (int u = 0; u < N; u++)
    st[u] = (tab.e[u] & m0) | (tab.e[N + u] & m1) | (tab.e[2 * N + u] & m2) | (tab.e[3 * N + u] & m3);
}

template <int IPT, int XU, bool NT, bool C16, bool C12, bool WIDE, bool SEAM, int EXT, bool ALIAS>
__device__ __forceinline__ void merge_load(const BlockDesc &d, int n_cols, int xlim, int max_gpair,
                                           const int *__restrict__ rp, const int *__restrict__ ci,
                                           const unsigned *__restrict__ ci16, const ChunkTab<XU> &xchunk,
                                           const double *__restrict__ val, const double *__restrict__ x,
                                           double *prod, int *roff, double *xs, const XHalo &halo,
                                           const double *__restrict__ w, double *wl, int lb,
                                           const SolverPass &sp, const PassScalars &ps) {
  const int WG = blockDim.x, tid = threadIdx.x;
  constexpr bool COMP = EXT == 2 && !diag::NO_SECOND_WINDOW;  // operand composed on the fly: x[c] + beta * x[c + b_off]
  constexpr bool OWN = EXT == 2 && !diag::NO_OWN_ROWS;        // own-row updates of a solver pass
  // 16-byte loads need an even element index: start one element early if the
  // block starts on an odd nonzero (that element belongs to the previous block;
  // its product lands in prod[0] and no row of this block references it).
  const int base = d.nnz_start & ~1;
  const int lead = d.nnz_start - base;
  const int total = d.nnz_count + lead;
  // An odd total ends in a pair whose second element is foreign (the next
  // block's first nonzero, or -- for the very last nonzero of an odd-nnz matrix
  // -- the 8 bytes after the array: a 16-byte-aligned 16-byte load that holds
  // one valid element cannot cross a page, so it is memory-safe).
  const int npairs = (total + 1) >> 1;

  // x tile.  With 16-bit indices the tile is either ONE window [cmin, cmin + cwidth) (KIND_CONTIG) or the SET of
  // 128-byte lines of x the block touches: 16-column chunks (their first columns in the block's table, built on the
  // host: plan::build_chunk_tiles / build_chunk_table); chunk s = u*(WG/16) + (tid >> 4) lands in LDS slots
  // [16 s, 16 s + 16), and a nonzero's 16-bit index is its slot -- stencil-like matrices (a few narrow bands far apart)
  // and stray columns fit that way.  Without 16-bit indices the tile is the contiguous window [cmin, cmin+cwidth).
  double xw[XU > 0 ? XU : 1];
  double xb[(COMP && XU > 0) ? XU : 1];                       // COMP: the same entries of the second operand
  uint64_t xsrc[XU > 0 ? XU : 1];
  if (XU > 0) {
    if (SEAM) {
      const bool chunked = C16 && !(d.kind_g & KIND_CONTIG);
      int col[XU > 0 ? XU : 1], st[XU > 0 ? XU : 1];
      if (chunked) chunk_starts_of<XU>(xchunk, tid, st);      // workgroup-uniform
#pragma unroll
      for (int u = 0; u < XU; u++) {
        col[u] = min(chunked ? st[u] + (tid & 15) : d.cmin + u * WG + tid, n_cols - 1);
        xsrc[u] = halo_entry(col[u], halo);
      }
#pragma unroll
      for (int u = 0; u < XU; u++) xsrc[u] = halo_source(x, col[u], xsrc[u], halo);
    } else if (WIDE) {
      // WIDE plans: every tiled block has ONE window (the planner starts it on an even column) and loads it in
      // 16-byte pairs, half the load instructions: 8.31 -> 8.19 us on the cant payload.  (Chunked tiles measured
      // 1-3 % SLOWER with paired loads -- per-lane or scalar chunk lookups alike -- and a per-block choice inside
      // one kernel loses the gain to the merged wait states, so this is a plan-level instantiation.)  Pair
      // j = u*WG + tid covers slots 2j, 2j+1.  The last pair may reach one element past the entries the block may
      // use (past the end of x when n_cols is odd): x is 16-byte aligned, so the aligned 16-byte load cannot
      // cross a page, and its slot is one no nonzero refers to.
      const dbl2 *x2 = reinterpret_cast<const dbl2 *>(x);
      const int plim = xlim >> 1;
#pragma unroll
      for (int u = 0; u < XU / 2; u++) {
        const int pi = min((d.cmin >> 1) + u * WG + tid, plim);
        const dbl2 pr = x2[pi];
        xw[2 * u] = pr.x;
        xw[2 * u + 1] = pr.y;
        if (COMP) {                                           // b_off is even (16-byte aligned operands)
          const dbl2 pb = x2[pi + (sp.b_off >> 1)];
          xb[COMP ? 2 * u : 0] = pb.x;
          xb[COMP ? 2 * u + 1 : 0] = pb.y;
        }
      }
    } else if (C16 && !(d.kind_g & KIND_CONTIG)) {            // workgroup-uniform
      int st[XU > 0 ? XU : 1];
      chunk_starts_of<XU>(xchunk, tid, st);
#pragma unroll
      for (int u = 0; u < XU; u++) {
        const int c = min(st[u] + (tid & 15), xlim);
        xw[u] = x[c];
        if (COMP) xb[COMP ? u : 0] = x[c + sp.b_off];
      }
    } else {                                                  // one window: no chunk table on the critical path
#pragma unroll
      for (int u = 0; u < XU; u++) {
        const int c = min(d.cmin + u * WG + tid, xlim);
        xw[u] = x[c];
        if (COMP) xb[COMP ? u : 0] = x[c + sp.b_off];
      }
    }
  }
  const int ro0 = rp[d.row_start + min(tid, d.n_rows)] - base;
  const int ro1 = rp[d.row_start + min(tid + WG, d.n_rows)] - base;

  dbl2 v[IPT / 2];
  int2v c[IPT / 2];
  unsigned c16[IPT / 2];                                      // C16: two 16-bit slots per word, unpacked after the park
  unsigned c12[3] = {0u, 0u, 0u};                             // C12: this thread's eight 12-bit slots, one 12-byte record
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *ci2 = reinterpret_cast<const int2v *>(ci);
  const int first = base >> 1;
  const int last = min(first + max(npairs - 1, 0), max_gpair);
  // C12 (IPT = 8): the planner packed the eight slots THIS thread needs -- foreign elements already replaced
  // by a slot the block owns -- into record lb*WG + tid: one 12-byte load instead of four 4-byte ones, 1.5
  // instead of 2 bytes per nonzero
  if (C12) {
    // (diagnostic build FREE_SLOTS: every block reads block 0's records -- what would a free slot stream buy?)
    const unsigned *rec = ci16 + ((size_t)(diag::FREE_SLOTS ? 0 : lb) * WG + tid) * 3;  // (a 3-vector type would be padded to 16 bytes)
    c12[0] = stream_load<NT>(rec);
    c12[1] = stream_load<NT>(rec + 1);
    c12[2] = stream_load<NT>(rec + 2);
  }
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    // clamped: redundant loads hit the same line  (diagnostic build FREE_VALUES: every block streams block 0's values)
    const int p = min((diag::FREE_VALUES ? 0 : first) + u * WG + tid, last);
    v[u] = stream_load<NT>(val2 + p);
    if (C12) continue;
    if (C16) c16[u] = stream_load<NT>(ci16 + p);
    else     c[u] = stream_load<NT>(ci2 + p);
  }
  if (XU > 0 && SEAM) {
#pragma unroll
    for (int u = 0; u < XU; u++) {
      xw[u] = load_at(xsrc[u]);
      if (COMP) xb[COMP ? u : 0] = load_at(xsrc[u] + 8 * (uint64_t)sp.b_off);
    }
  }
  // dot epilogue (EXT kernels, launch-uniform test): the block's slice of w, requested behind the stream
  // (youngest loads: nothing waits for them until the products are stored), parked in LDS for the row sums
  double w0 = 0.0, w1 = 0.0;
  if (EXT == 1 && w) {
    w0 = w[d.row_start + min(tid, d.n_rows - 1)];
    w1 = w[d.row_start + min(tid + WG, d.n_rows - 1)];
  }
  // solver pass: the rows this block owns (tid and tid + WG; a block has < 2*WG rows) -- new direction, the
  // solution update the previous pass owes, and the dot operand
  const int orow0 = d.row_start + min(tid, d.n_rows - 1), orow1 = d.row_start + min(tid + WG, d.n_rows - 1);
  if (OWN) {
    const double a0 = x[orow0], a1 = x[orow1], b0 = x[orow0 + sp.b_off], b1 = x[orow1 + sp.b_off];
    double s0 = 0.0, s1 = 0.0, wa0 = 0.0, wa1 = 0.0, wb0 = 0.0, wb1 = 0.0;
    if (ps.update_x) {
      s0 = sp.xsol[orow0];
      s1 = sp.xsol[orow1];
    }
    if (sp.wa) {                                              // launch-uniform
      wa0 = sp.wa[orow0]; wa1 = sp.wa[orow1];
      if (sp.wb) { wb0 = sp.wb[orow0]; wb1 = sp.wb[orow1]; }
    }
    const double n0 = fma(ps.beta, b0, a0), n1 = fma(ps.beta, b1, a1);
    if (sp.b_new) {
      if (tid < d.n_rows) store_own(sp.b_new + orow0, n0, sp.sys_scope);
      if (tid + WG < d.n_rows) store_own(sp.b_new + orow1, n1, sp.sys_scope);
    }
    if (ps.update_x) {
      if (tid < d.n_rows) sp.xsol[orow0] = fma(ps.alpha_prev, b0, s0);
      if (tid + WG < d.n_rows) sp.xsol[orow1] = fma(ps.alpha_prev, b1, s1);
    }
    w0 = sp.wa ? (sp.wb ? fma(ps.beta, wb0, wa0) : wa0) : n0;
    w1 = sp.wa ? (sp.wb ? fma(ps.beta, wb1, wa1) : wa1) : n1;
  }

  CASK_STAMP(1);
  if (WIDE && !SEAM) {                                        // pairs: slots 2j, 2j+1 for j = u*WG + tid
    dbl2 *xs2 = reinterpret_cast<dbl2 *>(xs);
#pragma unroll
    for (int u = 0; u < XU / 2; u++) {
      dbl2 pr;
      pr.x = COMP ? fma(ps.beta, xb[COMP ? 2 * u : 0], xw[2 * u]) : xw[2 * u];
      pr.y = COMP ? fma(ps.beta, xb[COMP ? 2 * u + 1 : 0], xw[2 * u + 1]) : xw[2 * u + 1];
      xs2[u * WG + tid] = pr;
    }
  } else if (XU > 0) {
#pragma unroll
    for (int u = 0; u < XU; u++) xs[u * WG + tid] = COMP ? fma(ps.beta, xb[COMP ? u : 0], xw[u]) : xw[u];
  }
  roff[tid] = ro0;
  roff[tid + WG] = ro1;
  if (XU > 0) __syncthreads();
  CASK_STAMP(2);
  if (C12) {                                                  // unpack only now: the stream is still landing
    static_assert(!C12 || IPT == 8, "12-bit packed slots are laid out for 8 items per thread");
    const unsigned q0 = c12[0], q1 = c12[1], q2 = c12[2];
    c[0].x = (int)(q0 & 0xfffu);
    c[0].y = (int)((q0 >> 12) & 0xfffu);
    c[1].x = (int)((q0 >> 24) | ((q1 & 0xfu) << 8));
    c[1].y = (int)((q1 >> 4) & 0xfffu);
    if (IPT / 2 > 2) {
      c[IPT / 2 > 2 ? 2 : 0].x = (int)((q1 >> 16) & 0xfffu);
      c[IPT / 2 > 2 ? 2 : 0].y = (int)((q1 >> 28) | ((q2 & 0xffu) << 4));
      c[IPT / 2 > 3 ? 3 : 0].x = (int)((q2 >> 8) & 0xfffu);
      c[IPT / 2 > 3 ? 3 : 0].y = (int)(q2 >> 20);
    }
  } else if (C16) {                                           // unpack the 16-bit slots only now
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      c[u].x = (int)(c16[u] & 0xffffu);
      c[u].y = (int)(c16[u] >> 16);
    }
  }

  // foreign elements: give them a column this block owns, so their gather stays
  // inside the x window / inside x (their products land in slots no row uses); the packed records
  // come with that done
  if (!C12) {
    if (lead && tid == 0) c[0].x = c[0].y;
    if (total & 1) {
#pragma unroll
      for (int u = 0; u < IPT / 2; u++)
        if (u * WG + tid >= npairs - 1) c[u].y = c[u].x;
    }
  }
  dbl2 xv[IPT / 2];
  if (XU > 0) {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = xs[C16 ? c[u].x : c[u].x - d.cmin];
      xv[u].y = xs[C16 ? c[u].y : c[u].y - d.cmin];
    }
    if (ALIAS) __syncthreads();                               // aliased window: its LDS becomes the products' (k_spmv_merge)
  } else if (SEAM) {                                          // gathers, some of them from peers
    uint64_t ex[IPT / 2], ey[IPT / 2];
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      ex[u] = halo_entry(c[u].x, halo);
      ey[u] = halo_entry(c[u].y, halo);
    }
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      ex[u] = halo_source(x, c[u].x, ex[u], halo);
      ey[u] = halo_source(x, c[u].y, ey[u], halo);
      xv[u].x = load_at(ex[u]);
      xv[u].y = load_at(ey[u]);
    }
    if (COMP) {
#pragma unroll
      for (int u = 0; u < IPT / 2; u++) {
        xv[u].x = fma(ps.beta, load_at(ex[u] + 8 * (uint64_t)sp.b_off), xv[u].x);
        xv[u].y = fma(ps.beta, load_at(ey[u] + 8 * (uint64_t)sp.b_off), xv[u].y);
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < IPT / 2; u++) {
      xv[u].x = x[c[u].x];
      xv[u].y = x[c[u].y];
    }
    if (COMP) {
      dbl2 xq[IPT / 2];
#pragma unroll
      for (int u = 0; u < IPT / 2; u++) {
        xq[u].x = x[c[u].x + sp.b_off];
        xq[u].y = x[c[u].y + sp.b_off];
      }
#pragma unroll
      for (int u = 0; u < IPT / 2; u++) {
        xv[u].x = fma(ps.beta, xq[u].x, xv[u].x);
        xv[u].y = fma(ps.beta, xq[u].y, xv[u].y);
      }
    }
  }
  // every lane stores: lanes past the last pair hold a duplicate of it and land
  // in slots no row offset points to
  dbl2 *prod2 = reinterpret_cast<dbl2 *>(prod);
#ifdef CASK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CASK_STAMP(3);
#endif
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) prod2[u * WG + tid] = v[u] * xv[u];
  if ((EXT == 1 && w) || (EXT == 2 && w)) {                   // solver pass: w != NULL means "leave the dot shares behind"
    wl[tid] = w0;
    wl[tid + WG] = w1;
  }
  __syncthreads();
  CASK_STAMP(4);
}

template <int IPT, int XU, bool NT, bool C16, bool C12, bool WIDE, bool SKEW, int EXT, bool ALIAS>
__device__ __forceinline__ void merge_block(const BlockDesc &d, int n_cols, int xlim, int max_gpair,
                                            const int *__restrict__ rp, const int *__restrict__ ci,
                                            const unsigned *__restrict__ ci16, const ChunkTab<XU> &xchunk,
                                            const double *__restrict__ val, const double *__restrict__ x,
                                            double *__restrict__ y, double *prod, int *roff, double *xs, double *wl,
                                            const XHalo &halo, const DotEpilogue &dot, int lb,
                                            const SolverPass &sp, const PassScalars &ps) {
  const int WG = blockDim.x, tid = threadIdx.x;
  // Launches with a dot epilogue carry 2*WG + 16 doubles more of dynamic LDS: the block's slice of w and
  // the per-wave sums.  Deliberately no static LDS: 256 bytes of it made the ordinary product measurably
  // slower (8.75 -> 8.88 us per launch in an interleaved A/B) although the occupancy calculator still
  // reports 6 workgroups per CU for 26 896 bytes (tools/lds_granule.hip); the cause was not established.
  // (wl: behind the window's own LDS, or -- a window that shares the products' space -- right behind the row offsets;
  // the kernel places it, the same for the tiled and the untiled blocks of a launch)
  double *dot_red = wl + 2 * WG;
  // the dot operand: EXT == 1 a vector (dot.w); a solver pass composes it (dot.dot_part != NULL asks for the shares)
  const bool want_dot = EXT == 2 ? dot.dot_part != nullptr : (EXT == 1 && dot.w != nullptr);   // launch-uniform
  const double *wsrc = EXT == 2 ? (want_dot ? x : nullptr) : dot.w;
  // a seam block's largest column (d.aux, set by the planner when there is a halo) is a halo column: its
  // load phase is a copy of its own, so the one every other block runs has no halo code in it
  if (EXT && halo.haddr != nullptr && d.aux >= halo.n_own)    // workgroup-uniform
    merge_load<IPT, XU, NT, C16, C12, WIDE, true, EXT, ALIAS>(d, n_cols, xlim, max_gpair, rp, ci, ci16, xchunk, val, x, prod, roff, xs,
                                                              halo, wsrc, wl, lb, sp, ps);
  else
    merge_load<IPT, XU, NT, C16, C12, WIDE, false, EXT, ALIAS>(d, n_cols, xlim, max_gpair, rp, ci, ci16, xchunk, val, x, prod, roff, xs,
                                                               halo, wsrc, wl, lb, sp, ps);
  const double *wrow = want_dot ? wl : nullptr;               // w[row_start + r] sits in wl[r]
  if (diag::NO_TAIL && !EXT) {                                // diagnostic build: what does the tail (row sums + y) cost?
    if (tid == 0 && prod[0] == 1.2345e300) y[d.row_start] = 0.0;
    return;
  }

  double dsum;
  switch (d.kind_g & 0xff) {
    case 1:  dsum = reduce_rows<1, SKEW, EXT>(d, prod, roff, y, wrow); break;
    case 2:  dsum = reduce_rows<2, SKEW, EXT>(d, prod, roff, y, wrow); break;
    case 4:  dsum = reduce_rows<4, SKEW, EXT>(d, prod, roff, y, wrow); break;
    case 8:  dsum = reduce_rows<8, SKEW, EXT>(d, prod, roff, y, wrow); break;
    case 16: dsum = reduce_rows<16, SKEW, EXT>(d, prod, roff, y, wrow); break;
    case 32: dsum = reduce_rows<32, SKEW, EXT>(d, prod, roff, y, wrow); break;
    default: dsum = reduce_rows<64, SKEW, EXT>(d, prod, roff, y, wrow); break;
  }
  if (EXT && want_dot) {                                      // launch-uniform: the block's share of w.y
    dsum = group_sum<64>(dsum);
    if ((tid & 63) == 0) dot_red[tid >> 6] = dsum;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int wv = 0; wv < (WG >> 6); wv++) s += dot_red[wv];
      dot.dot_part[lb] = s;
    }
  }
}

// SKEW: the plan holds blocks flagged KIND_SKEW (matrices without any run the instantiation that
// carries no second-pass code at all: 0.13 us per launch on cant).
// EXT: 0 = the lean kernel of ordinary products; 1 = the launch may carry halo sources and/or a dot epilogue;
// 2 = a solver pass (SolverPass: composed operand, own-row updates, scalars from partial sums) with or without
// halo sources.  Ordinary products run EXT = 0, which contains none of that code: a kernel this close to the
// memory system's limits pays for every extra branch, register and byte of LDS (measured while adding them:
// +1 to +6 %).
// Aliased window (r5, merge_window_aliased(): the widest window of the >= 8-items kernels): the products are parked OVER
// the x window -- dead once every wave has gathered its x values; one more barrier, in the shadow of the stream -- so a
// 256 x 8 block with a 2 048-entry window needs 18.4 instead of 34.8 KB of LDS: 8 instead of 4 workgroups per CU.
// Measured (profiles/r05_merge_forms.txt): G3_circuit-like 21.5 -> 20.6 us, atmosmodd-like 21.2 -> 19.6; on the
// 1 024-entry window of the cant-like plan (6 -> 8 workgroups per CU) it is 1 % SLOWER, which is why narrower windows
// keep their own LDS.
// One workgroup of a merge launch: hardware block `hw_block` of a grid of `n_blocks`.  (Factored out of k_spmv_merge in
// round 6 for a launch that carried the workgroups of TWO plans -- the A and A^T products of a BiCG pass: bit-identical,
// -0.2 %, removed, profiles/r06_bicg_dual.txt; the EXT = 0 kernels compile to the same instructions as before.)
template <int IPT, int XU, bool NT, bool C16, bool C12, bool WIDE, bool SKEW, int EXT>
__device__ __forceinline__ void merge_workgroup(int hw_block, const BlockDesc *__restrict__ blocks, int n_blocks, int remap,
                                                int n_cols, int nnz, const int *__restrict__ rp, const int *__restrict__ ci,
                                                const unsigned *__restrict__ ci16, const int *__restrict__ xchunk, int maxch,
                                                const double *__restrict__ val, const double *__restrict__ x,
                                                double *__restrict__ y, double *__restrict__ partials, const XHalo &halo,
                                                const DotEpilogue &dot, const PassArg<EXT> &pass_arg) {
  static_assert(IPT % 2 == 0, "items per thread must be even (16-byte loads)");
  extern __shared__ __align__(16) unsigned char smem[];
  const int WG = blockDim.x, CAP = WG * IPT, tid = threadIdx.x;
  constexpr bool ALIAS = merge_window_aliased(XU, IPT);
  double *prod = reinterpret_cast<double *>(smem);            // CAP + 2 doubles
  int *roff = reinterpret_cast<int *>(prod + CAP + 2);        // 2*WG ints (a block has < 2*WG rows)
  double *xs = ALIAS ? prod : reinterpret_cast<double *>(roff + 2 * WG);   // XU*WG doubles (ALIAS: XU <= IPT)
  double *wl = reinterpret_cast<double *>(roff + 2 * WG) + (ALIAS ? 0 : XU * WG);   // EXT: the block's slice of w, 2*WG + 16 doubles
  static_assert(!ALIAS || XU <= IPT, "aliased window: it must fit the products' space");

  CASK_STAMP(0);
  const int lb = logical_block(hw_block, n_blocks, remap);
  ChunkTab<XU> tab = {};
  constexpr bool CHUNKED = C16 && !WIDE && XU > 0;            // WIDE plans: every tiled block is a window
  if (CHUNKED) chunk_table_fetch<XU>(xchunk + (size_t)lb * maxch, tid, tab);
  const BlockDesc d = blocks[lb];
  if (CHUNKED) chunk_table_pin<XU>(tab);
  const ChunkTab<0> no_tab = {};

  const SolverPass sp = pass_of(pass_arg);                    // EXT < 2: all zeros, every use folds away
  PassScalars ps{0.0, 0.0, false, false};
  if (EXT == 2) {
    if (done_by_earlier_launch(sp.done, sp.iter)) return;     // a pass after the converged one: nothing happens (a flag set by a
                                                              // workgroup of THIS launch must not stop the others: they owe x)
    ps = pass_scalars(sp, lb, prod);                          // LDS not in use yet
    __syncthreads();
    if (ps.stop) {                                            // workgroup-uniform (and the same in every workgroup)
      // converged in the previous pass (or a final_only launch): the solution update that pass owes, nothing else.
      // A split long row is owned by its first piece.
      if (ps.update_x && (!(d.kind_g & KIND_LONG) || d.nnz_count == 0 || d.nnz_start == rp[d.row_start]))
        pass_own_rows(sp, ps, x, d.row_start, d.n_rows, false);
      return;
    }
  }

  if (d.kind_g & KIND_LONG) {
    if (d.nnz_count == 0) {                                   // a run of empty rows (planner: zero-fill piece)
      for (int r = tid; r < d.n_rows; r += WG) y[d.row_start + r] = 0.0;
      if (EXT == 2) pass_own_rows(sp, ps, x, d.row_start, d.n_rows, true);
      if (EXT && dot.dot_part && (EXT == 2 || dot.w) && tid == 0) dot.dot_part[lb] = 0.0;
      return;
    }
    // One piece of one long row: the whole workgroup strides over it.
    const int end = d.nnz_start + d.nnz_count;
    double acc = 0.0;
    for (int k = d.nnz_start + tid; k < end; k += 4 * WG) {
      int c[4];
      double v[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int kk = min(k + u * WG, end - 1);
        c[u] = stream_load<NT>(ci + kk);
        v[u] = stream_load<NT>(val + kk);
      }
      if (EXT && halo.haddr) {                                // launch-uniform
        uint64_t ent[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ent[u] = halo_entry(c[u], halo);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          ent[u] = halo_source(x, c[u], ent[u], halo);
          xv[u] = load_at(ent[u]);
        }
        if (EXT == 2) {
#pragma unroll
          for (int u = 0; u < 4; u++) xv[u] = fma(ps.beta, load_at(ent[u] + 8 * (uint64_t)sp.b_off), xv[u]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = x[c[u]];
        if (EXT == 2) {
#pragma unroll
          for (int u = 0; u < 4; u++) xv[u] = fma(ps.beta, x[c[u] + sp.b_off], xv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (k + u * WG < end) acc = fma(v[u], xv[u], acc);
    }
    acc = group_sum<64>(acc);
    if ((tid & 63) == 0) prod[tid >> 6] = acc;
    __syncthreads();
    // solver pass: the row's own entries (its first piece owns them) and its dot operand
    const bool owner = EXT == 2 && d.nnz_start == rp[d.row_start];
    double wrow = 0.0;
    if (EXT == 2 && tid == 0) {
      const int row = d.row_start;
      const double bv = x[row + sp.b_off], nv = fma(ps.beta, bv, x[row]);
      if (owner) {
        if (ps.update_x) sp.xsol[row] = fma(ps.alpha_prev, bv, sp.xsol[row]);
        if (sp.b_new) store_own(sp.b_new + row, nv, sp.sys_scope);
      }
      wrow = sp.wa ? (sp.wb ? fma(ps.beta, sp.wb[row], sp.wa[row]) : sp.wa[row]) : nv;
    }
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < (WG >> 6); w++) s += prod[w];
      if (d.kind_g & KIND_PARTIAL) {
        partials[d.aux] = s;
        if (EXT && dot.dot_part && (EXT == 2 || dot.w)) dot.dot_part[lb] = 0.0;   // the fix-up kernel owns this row's share
      } else {
        y[d.row_start] = s;
        if (EXT == 2 && dot.dot_part) dot.dot_part[lb] = wrow * s;
        if (EXT == 1 && dot.w) dot.dot_part[lb] = dot.w[d.row_start] * s;
      }
    }
    return;
  }

  const int max_gpair = ((nnz + 1) >> 1) - 1;
  // last entry of x[] a block without halo columns may touch: its window is padded to whole chunks and
  // may reach past its largest column, but never past the caller's n_own entries
  const int xlim = (EXT ? min(n_cols, halo.n_own) : n_cols) - 1;
  const bool tiled = XU > 0 && d.cwidth > 0 && d.cwidth <= XU * WG;   // workgroup-uniform
  if (tiled) {
    merge_block<IPT, XU, NT, C16, C12, WIDE, SKEW, EXT, ALIAS>(d, n_cols, xlim, max_gpair, rp, ci, ci16, tab, val, x, y, prod, roff,
                                                               xs, wl, halo, dot, lb, sp, ps);
  } else {
    merge_block<IPT, 0, NT, false, false, false, SKEW, EXT, false>(d, n_cols, xlim, max_gpair, rp, ci, ci16, no_tab, val, x, y, prod, roff,
                                                            xs, wl, halo, dot, lb, sp, ps);
  }
  CASK_STAMP(5);
}

template <int IPT, int XU, bool NT, bool C16, bool C12, bool WIDE, bool SKEW, int EXT>
__global__ void k_spmv_merge(const BlockDesc *__restrict__ blocks, int n_blocks, int remap, int n_cols, int nnz,
                             const int *__restrict__ rp, const int *__restrict__ ci,
                             const unsigned *__restrict__ ci16, const int *__restrict__ xchunk, int maxch,
                             const double *__restrict__ val, const double *__restrict__ x,
                             double *__restrict__ y, double *__restrict__ partials, XHalo halo, DotEpilogue dot,
                             PassArg<EXT> pass_arg) {
  merge_workgroup<IPT, XU, NT, C16, C12, WIDE, SKEW, EXT>(blockIdx.x, blocks, n_blocks, remap, n_cols, nnz, rp, ci, ci16, xchunk,
                                                          maxch, val, x, y, partials, halo, dot, pass_arg);
}

}  // namespace caskhip
