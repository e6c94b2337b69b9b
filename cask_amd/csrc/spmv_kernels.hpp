// The other SpMV kernels: wavefront-per-row (VECTOR), pipelined persistent waves (MERGE_WAVE), long-row
// pieces, the fix-up of split rows and the plan helpers.  The workgroup-level merge kernel lives in
// merge_kernel.hpp.  What these kernels replace in the reference: spmv_common.hpp.
#pragma once
#include "spmv_common.hpp"

namespace caskhip {

// One row's dot product over lanes j, j+L, ... with up to four L-chunks in
// flight; XLDS selects the x source at compile time so the gathers of all four
// chunks issue back to back (a runtime select inside the unrolled loop makes
// hipcc serialise them behind per-chunk waits).
template <int L, bool XLDS, bool NT>
__device__ __forceinline__ double row_dot(int s, int e, int j, const int *__restrict__ ci,
                                          const double *__restrict__ val, const double *__restrict__ x,
                                          const double *xs, int cmin) {
  double acc = 0.0;
  int k = (s & ~(L - 1)) + j;
  if (k < s) k += L;
  for (; k < e; k += 4 * L) {
    int c[4];
    double v[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int kk = min(k + u * L, e - 1);               // clamped: always a valid element of this row
      c[u] = stream_load<NT>(ci + kk);
      v[u] = stream_load<NT>(val + kk);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) xv[u] = XLDS ? xs[c[u] - cmin] : x[c[u]];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k + u * L < e) acc = fma(v[u], xv[u], acc);
  }
  return acc;
}

// ------------------------------------------------------------ vector variant
// "Wavefront per row" family: L lanes of a wave own one row (rows per wavefront
// = 64/L; L = 1 is thread-per-row).  Loads start on an L-aligned element so a
// wave-instruction covers whole 128-byte lines.  Up to 4 L-chunks of a row are
// in flight per lane.  x comes from an LDS window staged per workgroup when
// the workgroup's column span fits the tile (LDSX), else from L2.
// long_len > 0 (r6): rows of more than long_len nonzeros are NOT this kernel's -- a row-mapped kernel lets L lanes walk a
// 4 700-entry row of a power-law matrix while the rest of the chip has finished (260-300 us on the webbase look-alikes);
// the plan lists them as long-row pieces for k_spmv_long (+ the fix-up), as the merge plans do.
template <int L, bool LDSX, bool NT>
__global__ void k_spmv_vector(int n_rows, int n_wg, int remap, int tile_width, int long_len,
                              const int2v *__restrict__ xspan,
                              const int *__restrict__ rp, const int *__restrict__ ci,
                              const double *__restrict__ val, const double *__restrict__ x,
                              double *__restrict__ y) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *xs = reinterpret_cast<double *>(smem);
  const int tid = threadIdx.x;
  const int wg = logical_block(blockIdx.x, n_wg, remap);
  const int rows_per_wg = blockDim.x / L;
  const int row = wg * rows_per_wg + tid / L;
  const int j = tid & (L - 1);

  int cmin = 0;
  bool in_lds = false;
  if (LDSX) {
    const int2v span = xspan[wg];
    cmin = span.x;
    in_lds = span.y > 0 && span.y <= tile_width;          // workgroup-uniform
    if (in_lds) {
      for (int i = tid; i < span.y; i += blockDim.x) xs[i] = x[cmin + i];
      __syncthreads();
    }
  }

  double acc = 0.0;
  bool mine = row < n_rows;
  if (mine) {
    const int s = rp[row], e = rp[row + 1];
    if (long_len > 0 && e - s > long_len) mine = false;       // (the same answer in all L lanes of the row)
    else if (LDSX && in_lds) acc = row_dot<L, true, NT>(s, e, j, ci, val, x, xs, cmin);
    else                     acc = row_dot<L, false, NT>(s, e, j, ci, val, x, xs, cmin);
  }
  acc = group_sum<L>(acc);
  if (j == 0 && mine) y[row] = acc;
}

// ------------------------------------------------ vector variant, pair loads (L >= 4)
// The same family -- L lanes of a wave per row -- built the way the merge kernel streams: 16-byte value pairs and
// 8-byte index pairs (a lane owns elements 2p, 2p+1 of its row), and VEC_RG row groups per wave in flight at once, so
// a lane has VEC_RG * VEC_U pair loads outstanding before it consumes the first (the round-1 kernel had four 8+4-byte
// loads).  A row group is the 64/L consecutive rows one wave covers at a time; a workgroup owns
// (wg_size/L) * VEC_RG consecutive rows.  Rows longer than 2*L*VEC_U elements loop.  Partial sums stay in
// registers (DPP butterfly over the L lanes): no product staging in LDS, one barrier (the x window).
constexpr int VEC_RG = 4;   // row groups per wave
constexpr int VEC_U = 2;    // pair loads per lane and row before the first use
constexpr int VEC_XW = 4;   // x-window loads per lane (windows of up to VEC_XW * wg_size entries are staged in LDS)
template <int L, bool LDSX, bool NT>
__global__ void k_spmv_vector2(int n_rows, int n_wg, int remap, int tile_width, int nnz, int long_len,
                               const int2v *__restrict__ xspan,
                               const int *__restrict__ rp, const int *__restrict__ ci,
                               const double *__restrict__ val, const double *__restrict__ x,
                               double *__restrict__ y) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *xs = reinterpret_cast<double *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = logical_block(blockIdx.x, n_wg, remap);
  constexpr int RPW = 64 / L;                                 // rows of one row group
  const int rows_per_wg = (blockDim.x / L) * VEC_RG;
  const int j = lane & (L - 1), q = lane / L;
  const int row0 = wg * rows_per_wg + wave * (RPW * VEC_RG) + q;
  CASK_STAMP(0);
  if (nnz == 0) {                                             // nothing to stream (and no valid pair to clamp to)
#pragma unroll
    for (int g = 0; g < VEC_RG; g++)
      if (j == 0 && row0 + g * RPW < n_rows) y[row0 + g * RPW] = 0.0;
    return;
  }

  // Issue order as in the merge kernel: row bounds (oldest), then the x window into registers, then the stream --
  // the stream waits for the row bounds only, the window is parked in LDS while the stream is in flight.
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *ci2 = reinterpret_cast<const int2v *>(ci);
  const int max_gpair = ((nnz + 1) >> 1) - 1;                 // last valid pair of the arrays
  int s[VEC_RG], e[VEC_RG];
  bool mine[VEC_RG];
#pragma unroll
  for (int g = 0; g < VEC_RG; g++) {
    const int row = min(row0 + g * RPW, n_rows - 1);
    s[g] = rp[row];
    e[g] = rp[row + 1];
    mine[g] = row0 + g * RPW < n_rows;
    if (long_len > 0 && e[g] - s[g] > long_len) mine[g] = false;   // a long-row piece's (k_spmv_long): not summed, not stored here
    if (!mine[g]) e[g] = s[g];                                // past the matrix / somebody else's: an empty row
  }
  int cmin = 0, span_hi = 0;                                  // span_hi: last staged window entry (gathers are clamped to it)
  bool in_lds = false;
  double xw[VEC_XW];
  if (LDSX) {
    const int2v span = xspan[wg];
    cmin = span.x;
    span_hi = max(span.y - 1, 0);
    in_lds = span.y > 0 && span.y <= tile_width && span.y <= VEC_XW * (int)blockDim.x;   // workgroup-uniform
    if (in_lds) {
#pragma unroll
      for (int u = 0; u < VEC_XW; u++) xw[u] = x[cmin + min(u * (int)blockDim.x + tid, span.y - 1)];
    }
  }
  dbl2 v[VEC_RG][VEC_U];
  int2v c[VEC_RG][VEC_U];
#pragma unroll
  for (int g = 0; g < VEC_RG; g++) {
    const int plast = min(max(e[g] - 1, s[g]) >> 1, max_gpair);
#pragma unroll
    for (int u = 0; u < VEC_U; u++) {
      const int p = min((s[g] >> 1) + u * L + j, plast);     // clamped: a valid pair of (or next to) the row
      v[g][u] = stream_load<NT>(val2 + p);
      c[g][u] = stream_load<NT>(ci2 + p);
    }
  }
  CASK_STAMP(1);                                              // the row bounds have arrived: the stream is requested
  if (LDSX && in_lds) {
#pragma unroll
    for (int u = 0; u < VEC_XW; u++) xs[u * blockDim.x + tid] = xw[u];
    __syncthreads();
  }
  CASK_STAMP(2);
#ifdef CASK_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CASK_STAMP(3);                                              // the stream has landed
#endif
  double acc[VEC_RG];
#pragma unroll
  for (int g = 0; g < VEC_RG; g++) {
    acc[g] = 0.0;
    const int plast = min(max(e[g] - 1, s[g]) >> 1, max_gpair);
    dbl2 xv[VEC_U];
#pragma unroll
    for (int u = 0; u < VEC_U; u++) {
      const int p = (s[g] >> 1) + u * L + j;
      // a clamped lane re-read a pair some other lane owns (its products are masked below); the second element of
      // the very last pair of an odd-nnz matrix lies behind col_ind: give it the first one's column
      const int cx = c[g][u].x, cy = 2 * min(p, plast) + 1 < nnz ? c[g][u].y : c[g][u].x;
      if (LDSX && in_lds) {
        xv[u].x = xs[min(max(cx - cmin, 0), span_hi)];
        xv[u].y = xs[min(max(cy - cmin, 0), span_hi)];
      } else {
        xv[u].x = x[cx];
        xv[u].y = x[cy];
      }
      if (p <= plast) {
        if (2 * p >= s[g] && 2 * p < e[g]) acc[g] = fma(v[g][u].x, xv[u].x, acc[g]);
        if (2 * p + 1 >= s[g] && 2 * p + 1 < e[g]) acc[g] = fma(v[g][u].y, xv[u].y, acc[g]);
      }
    }
    // the rest of a long row (wave-uniform per row group only through the lanes' own bounds: a plain loop)
    for (int p = (s[g] >> 1) + VEC_U * L + j; p <= plast && 2 * p < e[g]; p += L) {
      const dbl2 vv = stream_load<NT>(val2 + p);
      const int2v cc = stream_load<NT>(ci2 + p);
      const int c1 = 2 * p + 1 < nnz ? cc.y : cc.x;
      const double x0 = (LDSX && in_lds) ? xs[min(max(cc.x - cmin, 0), span_hi)] : x[cc.x];
      const double x1 = (LDSX && in_lds) ? xs[min(max(c1 - cmin, 0), span_hi)] : x[c1];
      if (2 * p >= s[g]) acc[g] = fma(vv.x, x0, acc[g]);
      if (2 * p + 1 < e[g]) acc[g] = fma(vv.y, x1, acc[g]);
    }
  }
  CASK_STAMP(4);
#pragma unroll
  for (int g = 0; g < VEC_RG; g++) {
    const double r = group_sum<L>(acc[g]);
    if (j == 0 && mine[g]) y[row0 + g * RPW] = r;
  }
  CASK_STAMP(5);
}

// --------------------------------------------- merge variant, pipelined waves
// Same merge-path decomposition, but the unit of work is a WAVE and the waves
// are persistent: each wave walks a strided list of small blocks (at most
// 64*IPT items, at most 127 rows) and software-pipelines them -- while block b
// is gathered, multiplied and row-reduced, the 16-byte value/index stream and
// the row offsets of block b+1 are already in flight, and the descriptor of
// b+2 is being fetched.  A one-shot workgroup pays the whole dependent chain
// descriptor -> row_ptr -> stream -> gather -> reduce (~6 us on MI355X) once
// per ~2 rounds; here it is paid once per wave and hidden afterwards.  No
// workgroup barrier: a wave only talks to itself through its own LDS slice
// (DS operations of one wave execute in order).
struct WaveStream {            // one block's worth of loads held in registers
  int ro0, ro1;                // row offsets for rows lane, lane+64 (relative to base)
};

template <int G>
__device__ __forceinline__ void wave_reduce_rows(int n_rows, int row_start, bool skew, const double *prod,
                                                 const int *roff, double *__restrict__ y) {
  const int lane = threadIdx.x & 63;
  constexpr int rows_per_pass = 64 / G;
  const int j = lane & (G - 1);
  for (int r0 = 0; r0 < n_rows; r0 += rows_per_pass) {
    const int r = r0 + lane / G;
    double acc = 0.0;
    bool mine = r < n_rows;
    if (mine) {
      const int s = roff[r], e = roff[r + 1];
      if (skew && e - s > SKEW_FACTOR * G) {
        mine = false;
      } else {
#pragma unroll 4
        for (int k = s + j; k < e; k += G) acc += prod[k];
      }
    }
    acc = group_sum<G>(acc);
    if (j == 0 && mine) y[row_start + r] = acc;
  }
  if (skew) {
    for (int r = 0; r < n_rows; r++) {                        // wave-uniform
      const int s = roff[r], e = roff[r + 1];
      if (e - s <= SKEW_FACTOR * G) continue;
      double a = 0.0;
      for (int k = s + lane; k < e; k += 64) a += prod[k];
      a = group_sum<64>(a);
      if (lane == 0) y[row_start + r] = a;
    }
  }
}

template <int IPT, bool NT>
__device__ __forceinline__ void wave_issue(const BlockDesc &d, int max_gpair, const int *__restrict__ rp,
                                           const int *__restrict__ ci, const double *__restrict__ val,
                                           dbl2 (&v)[IPT / 2], int2v (&c)[IPT / 2], int &ro0, int &ro1) {
  const int lane = threadIdx.x & 63;
  const int base = d.nnz_start & ~1;
  const int npairs = (d.nnz_count + d.nnz_start - base + 1) >> 1;   // an odd total ends in a half-foreign pair
  const dbl2 *val2 = reinterpret_cast<const dbl2 *>(val);
  const int2v *ci2 = reinterpret_cast<const int2v *>(ci);
  // Lanes past the block's last pair re-read that pair (same cache line); the
  // clamp to max_gpair keeps an empty block at the very end of the arrays
  // inside the allocation.
  const int first = base >> 1;
  const int last = min(first + max(npairs - 1, 0), max_gpair);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    const int p = min(first + u * 64 + lane, last);
    v[u] = stream_load<NT>(val2 + p);
    c[u] = stream_load<NT>(ci2 + p);
  }
  ro0 = rp[d.row_start + min(lane, d.n_rows)] - base;
  ro1 = rp[d.row_start + min(lane + 64, d.n_rows)] - base;
}

template <int IPT, bool NT, bool HAS_NEXT>
__device__ __forceinline__ void wave_block(const BlockDesc &d, const BlockDesc &dn, int max_gpair,
                                           const int *__restrict__ rp, const int *__restrict__ ci,
                                           const double *__restrict__ val, const double *__restrict__ x,
                                           double *__restrict__ y, double *prod, int *roff,
                                           dbl2 (&v)[IPT / 2], int2v (&c)[IPT / 2], int &ro0, int &ro1) {
  const int lane = threadIdx.x & 63;
  const int base = d.nnz_start & ~1;
  const int total = d.nnz_count + d.nnz_start - base;
  // 1. gathers of the current block (their indices arrived while the previous block was reduced);
  //    the half-foreign tail pair of an odd block gathers a column this block owns
  if (total & 1) {
    const int npairs = (total + 1) >> 1;
#pragma unroll
    for (int u = 0; u < IPT / 2; u++)
      if (u * 64 + lane >= npairs - 1) c[u].y = c[u].x;
  }
  dbl2 xv[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) {
    xv[u].x = x[c[u].x];
    xv[u].y = x[c[u].y];
  }
  const int cur_ro0 = ro0, cur_ro1 = ro1;
  dbl2 pv[IPT / 2];
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) pv[u] = v[u];
  // 2. next block's stream goes out BEHIND the gathers (vmcnt retires in order:
  //    waiting for the gathers must not wait for these)
  if (HAS_NEXT) wave_issue<IPT, NT>(dn, max_gpair, rp, ci, val, v, c, ro0, ro1);
  // 3. row offsets and products into this wave's LDS slice
  roff[lane] = cur_ro0;
  roff[lane + 64] = cur_ro1;
  // Every lane stores (lanes past the block's last pair hold a duplicate of it
  // and land in slots no row offset points to): a conditional store would let
  // hipcc sink the gathers into the branch, behind the prefetch, and the wait
  // for them would then drain the prefetch too.
  dbl2 *prod2 = reinterpret_cast<dbl2 *>(prod);
#pragma unroll
  for (int u = 0; u < IPT / 2; u++) prod2[u * 64 + lane] = pv[u] * xv[u];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // 4. per-row sums
  switch (d.kind_g & 0xff) {
    case 1:  wave_reduce_rows<1>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 2:  wave_reduce_rows<2>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 4:  wave_reduce_rows<4>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 8:  wave_reduce_rows<8>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 16: wave_reduce_rows<16>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    case 32: wave_reduce_rows<32>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
    default: wave_reduce_rows<64>(d.n_rows, d.row_start, d.kind_g & KIND_SKEW, prod, roff, y); break;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int IPT, bool NT>
__global__ void k_spmv_merge_wave(const BlockDesc *__restrict__ blocks, int n_blocks, int remap, int nnz,
                                  const int *__restrict__ rp, const int *__restrict__ ci,
                                  const double *__restrict__ val, const double *__restrict__ x,
                                  double *__restrict__ y) {
  static_assert(IPT % 2 == 0, "items per thread must be even (16-byte loads)");
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int CAP = 64 * IPT;
  const int wave = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
  double *prod = reinterpret_cast<double *>(smem) + wave * (CAP + 2);
  int *roff = reinterpret_cast<int *>(reinterpret_cast<double *>(smem) + waves_per_wg * (CAP + 2)) + wave * 128;

  // Which blocks does this wave walk?  With the XCD remap every XCD owns one
  // contiguous eighth of the block list and its resident waves sweep it
  // together (stride = waves on that XCD), so concurrently processed blocks are
  // neighbours: shared x lines and seam cache lines stay in that XCD's L2.
  int b, b_end, stride;
  if (remap) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int gq = gridDim.x >> 3, gr = gridDim.x & 7;
    const int wgs_here = gq + (xcd < gr ? 1 : 0);
    const int nq = n_blocks >> 3, nr = n_blocks & 7;
    const int xb0 = xcd * nq + min(xcd, nr);
    b_end = xb0 + nq + (xcd < nr ? 1 : 0);
    stride = wgs_here * waves_per_wg;
    b = xb0 + idx * waves_per_wg + wave;
  } else {
    stride = gridDim.x * waves_per_wg;
    b = blockIdx.x * waves_per_wg + wave;
    b_end = n_blocks;
  }
  b = __builtin_amdgcn_readfirstlane(b);
  if (b >= b_end) return;

  dbl2 v[IPT / 2];
  int2v c[IPT / 2];
  int ro0, ro1;
  const int max_gpair = ((nnz + 1) >> 1) - 1;
  BlockDesc d = blocks[b];
  wave_issue<IPT, NT>(d, max_gpair, rp, ci, val, v, c, ro0, ro1);
  int nb = b + stride;
  BlockDesc dn = blocks[min(nb, b_end - 1)];
  while (nb < b_end) {
    const int nnb = nb + stride;
    const BlockDesc dnn = blocks[min(nnb, b_end - 1)];       // descriptor two blocks ahead
    wave_block<IPT, NT, true>(d, dn, max_gpair, rp, ci, val, x, y, prod, roff, v, c, ro0, ro1);
    d = dn;
    dn = dnn;
    nb = nnb;
  }
  wave_block<IPT, NT, false>(d, d, max_gpair, rp, ci, val, x, y, prod, roff, v, c, ro0, ro1);
}

// Long-row pieces as their own launch (the pipelined kernel handles only
// whole-row blocks): one workgroup per piece.
template <bool NT>
__global__ void k_spmv_long(const BlockDesc *__restrict__ blocks, int n_blocks,
                            const int *__restrict__ ci, const double *__restrict__ val,
                            const double *__restrict__ x, double *__restrict__ y,
                            double *__restrict__ partials) {
  __shared__ double red[16];
  const int tid = threadIdx.x, WG = blockDim.x;
  const BlockDesc d = blocks[blockIdx.x];
  if (d.nnz_count == 0) {                                     // a run of empty rows (planner: zero-fill piece)
    for (int r = tid; r < d.n_rows; r += WG) y[d.row_start + r] = 0.0;
    return;
  }
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
#pragma unroll
    for (int u = 0; u < 4; u++) xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k + u * WG < end) acc = fma(v[u], xv[u], acc);
  }
  acc = group_sum<64>(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < (WG >> 6); w++) s += red[w];
    if (d.kind_g & KIND_PARTIAL) partials[d.aux] = s; else y[d.row_start] = s;
  }
}

// Sums the pieces of rows that were split over several workgroups, in piece
// order (deterministic; no float atomics anywhere in the engine).
__global__ void k_spmv_fixup(const SplitRow *__restrict__ rows, int n, const double *__restrict__ partials,
                             double *__restrict__ y, DotEpilogue dot, const int *done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (done && *done) return;                                  // solver pass after convergence: the product did not run
  const SplitRow r = rows[i];
  double s = 0.0;
  for (int k = 0; k < r.n_slots; k++) s += partials[r.first_slot + k];
  y[r.row] = s;
  if (dot.w) dot.dot_part[i] = dot.w[r.row] * s;              // slots behind the blocks' (caller offsets the pointer)
}

// ------------------------------------------------------------ plan helpers
// Column span [min,max] of a run of nonzeros, one workgroup per run; fills the
// x-window fields used by the LDSX paths.
__global__ void k_col_span_blocks(BlockDesc *blocks, int n_blocks, const int *__restrict__ ci) {
  __shared__ int smin[64], smax[64];
  const int b = blockIdx.x;
  if (b >= n_blocks) return;
  const int s = blocks[b].nnz_start, e = s + blocks[b].nnz_count;
  int lo = INT32_MAX, hi = -1;
  for (int k = s + threadIdx.x; k < e; k += blockDim.x) {
    const int c = ci[k];
    lo = min(lo, c); hi = max(hi, c);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) { lo = min(lo, smin[w]); hi = max(hi, smax[w]); }
    blocks[b].cmin = hi < 0 ? 0 : lo;
    blocks[b].cwidth = hi < 0 ? 0 : hi - lo + 1;
  }
}

// min / max of a column-index array (create_device validates borrowed arrays with it); out[0] = min, out[1] = max,
// both preset by the host.
__global__ void k_col_range(int64_t n, const int *__restrict__ ci, int *out) {
  int lo = INT32_MAX, hi = INT32_MIN;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const int c = ci[k];
    lo = min(lo, c); hi = max(hi, c);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  __shared__ int slo[16], shi[16];                            // one atomic pair per workgroup, not per wave
  if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) { lo = min(lo, slo[w]); hi = max(hi, shi[w]); }
    atomicMin(out, lo);
    atomicMax(out + 1, hi);
  }
}

__global__ void k_col_span_rows(int2v *xspan, int n_wg, int rows_per_wg, int n_rows,
                                const int *__restrict__ rp, const int *__restrict__ ci) {
  __shared__ int smin[64], smax[64];
  const int b = blockIdx.x;
  if (b >= n_wg) return;
  const int r0 = b * rows_per_wg, r1 = min(n_rows, r0 + rows_per_wg);
  const int s = rp[r0], e = rp[r1];
  int lo = INT32_MAX, hi = -1;
  for (int k = s + threadIdx.x; k < e; k += blockDim.x) {
    const int c = ci[k];
    lo = min(lo, c); hi = max(hi, c);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) { lo = min(lo, smin[w]); hi = max(hi, smax[w]); }
    int2v out;
    out.x = hi < 0 ? 0 : lo;
    out.y = hi < 0 ? 0 : hi - lo + 1;
    xspan[b] = out;
  }
}

}  // namespace caskhip
