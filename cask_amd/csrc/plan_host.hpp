// Host planners of the SpMV launch plans: merge-path cuts, x tiles as 64-column chunks, 12-bit packed slots, seam-block
// placement, SCAN row-end words / row maps / x windows, and the row-mapped slices of variant SLICE.  Everything here is
// index arithmetic on plain arrays and byte images written with shifts and memcpy -- no HIP, no device pointers -- so that
// it compiles with plain g++ and runs under AddressSanitizer / UBSan on the CPU (`make asan`,
// tests/cpp/test_planners.cpp builds every plan for the reference's fixtures and checks the invariants the kernels
// rely on).  cask_hip.hip calls these and uploads what they return.
//
// What this replaces in the reference: Spmv::preprocess / do_blocking / sliceColumns (src/runtime/Spmv.cpp:42-107,
// 329-365) -- the host-side packing of the matrix into what the device design streams.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "plan_types.hpp"

namespace caskhip {
namespace plan {

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

// Blocks are independent (disjoint nonzero ranges): plan them on a few host threads.  work(b0, b1) handles [b0, b1).
template <typename F>
inline void for_block_ranges(size_t nb, int64_t nnz, F work) {
  size_t n_threads = nnz < 200000 ? 1 : std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16);
  n_threads = std::min(n_threads, std::max<size_t>(nb, 1));
  if (n_threads <= 1) {
    work((size_t)0, nb);
    return;
  }
  std::vector<std::thread> pool;
  for (size_t t = 0; t < n_threads; t++) pool.emplace_back(work, nb * t / n_threads, nb * (t + 1) / n_threads);
  for (auto &th : pool) th.join();
}

// Cut the merge path of (row ends) against (nonzero indices) into shares of at
// most `cap` items (and at most `max_rows` rows), snapped to row boundaries;
// rows longer than cap/2 become long-row pieces of at most `piece` nonzeros.
// Long pieces go to `longs` when it is given (pipelined plan), else inline.
// `rows_are_items` = false (SCAN plans): only nonzeros count against `cap`, and a block that spans empty rows is
// flagged KIND_HOLES.
inline void build_merge_blocks(const int *rp, int n_rows, int cap, int max_rows, long piece, int threads,
                               std::vector<BlockDesc> &blocks, std::vector<BlockDesc> *longs, std::vector<SplitRow> &splits,
                               int &n_long, int &n_partial_slots, bool rows_are_items = true) {
  const int long_t = cap / 2;
  n_long = 0;
  n_partial_slots = 0;
  int cur_start = 0, cur_rows = 0, cur_nnz = 0, cur_max = 0, cur_empty = 0;
  auto close = [&](int next_row) {
    if (cur_rows == 0) return;
    BlockDesc d{};
    d.row_start = cur_start;
    d.n_rows = cur_rows;
    d.nnz_start = rp[cur_start];
    d.nnz_count = cur_nnz;
    // lanes per row in the reduce phase: as many as keep the phase to ONE pass over the block's rows
    // (threads / rows), but no more than the mean row length can feed
    const int mean = std::max(1, cur_nnz / cur_rows);
    const int by_rows = pow2_floor(std::max(1, threads / cur_rows));
    int by_len = 1;
    while (by_len < mean && by_len < 64) by_len *= 2;
    d.kind_g = std::min(64, std::max(1, std::min(by_rows, by_len)));
    if (cur_max > skew_short_max(d.kind_g & 0xff)) d.kind_g |= KIND_SKEW;   // longer rows: 16 lanes or a wave each
    if (cur_nnz == 0) {
      // a run of empty rows: nothing to stream.  The stream path would still issue its clamped 16-byte pair
      // loads, and for a block at the (odd) end of the arrays the pair's second element lies past col_ind --
      // an uninitialised column fed to an x gather.  Such a block is a zero-fill piece instead (long-row path,
      // nnz_count == 0, n_rows rows), which touches neither the stream nor x.
      d.kind_g = KIND_LONG;
      d.aux = 0;
      (longs ? *longs : blocks).push_back(d);
    } else {
      if (!rows_are_items) d.kind_g = cur_empty ? KIND_HOLES : 0;
      blocks.push_back(d);
    }
    cur_rows = 0;
    cur_nnz = 0;
    cur_max = 0;
    cur_empty = 0;
    cur_start = next_row;
  };
  for (int r = 0; r < n_rows; r++) {
    const int len = rp[r + 1] - rp[r];
    if (len > long_t) {
      close(r);
      n_long++;
      const int n_pieces = (int)((len + piece - 1) / piece);
      if (n_pieces > 1) splits.push_back(SplitRow{r, n_partial_slots, n_pieces, 0});
      for (int pc = 0; pc < n_pieces; pc++) {
        BlockDesc d{};
        d.row_start = r;
        d.n_rows = 1;
        d.nnz_start = rp[r] + (int)(pc * piece);
        d.nnz_count = (int)std::min<long>(piece, len - pc * piece);
        d.kind_g = KIND_LONG | (n_pieces > 1 ? KIND_PARTIAL : 0);
        d.aux = n_pieces > 1 ? n_partial_slots++ : 0;
        (longs ? *longs : blocks).push_back(d);
      }
      cur_start = r + 1;
      continue;
    }
    if (cur_rows > 0 && ((rows_are_items ? cur_rows + 1 : 0) + cur_nnz + len > cap || cur_rows + 1 > max_rows)) close(r);
    if (cur_rows == 0) cur_start = r;
    cur_rows++;
    cur_nnz += len;
    cur_empty += len == 0;
    cur_max = std::max(cur_max, len);
  }
  close(n_rows);
}

// x tile as a set of column ranges: for every block collect the distinct columns it references, join
// columns closer than GAP into ranges, cut the ranges into 64-column chunks.  A block whose chunks are ONE run of
// consecutive chunks is a window (KIND_CONTIG: the kernel computes its addresses from cmin; chunk_starts holds the
// 64-column starts).  Any other block (r6) gets the set of 128-byte LINES of x it touches instead: 16-column chunks on
// multiples of 16 (TILE_SUB), chunk_starts holds their starts -- a stray column then costs one line of x, not the four
// of a 64-column chunk: on the G3_circuit-like matrix (a 5-point grid + 1 % random long-range edges: ~8 stray columns
// in a block of ~400 rows) the windows of a launch shrink from 28.6 to 16.3 MB of distinct lines, 12 of the launch's
// 128 MB (tools/chunk_model.py; measured: DESIGN.md section 5).  A block whose chunks fit `max_slots` is "tiled": its
// nonzeros get 16-bit LDS slot indices (chunk * width + offset).  d.cwidth = slots used (0 = not tiled).
constexpr int TILE_SUB = 16;
inline void build_chunk_tiles(const int *ci, int64_t nnz, std::vector<BlockDesc> &blocks, int max_slots,
                              std::vector<std::vector<int>> &chunk_starts, std::vector<unsigned short> &ci16) {
  constexpr int GAP = 32;
  ci16.assign((size_t)nnz + 8, 0);
  chunk_starts.assign(blocks.size(), {});
  auto work = [&](size_t b0, size_t b1) {
    std::vector<int> uniq, starts;
    for (size_t b = b0; b < b1; b++) {
      BlockDesc &d = blocks[b];
      d.cwidth = 0;
      if ((d.kind_g & KIND_LONG) || d.nnz_count == 0) continue;
      const int k0 = d.nnz_start, k1 = d.nnz_start + d.nnz_count;
      uniq.assign(ci + k0, ci + k1);
      std::sort(uniq.begin(), uniq.end());
      uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
      starts.clear();
      size_t i = 0;
      while (i < uniq.size()) {
        size_t j = i;
        while (j + 1 < uniq.size() && uniq[j + 1] - uniq[j] <= GAP) j++;
        for (int c = uniq[i] & ~1; c <= uniq[j]; c += 64) starts.push_back(c);   // even starts: the kernel loads the tile in 16-byte pairs
        i = j + 1;
      }
      bool contiguous = !starts.empty();
      for (size_t c = 1; c < starts.size(); c++) contiguous = contiguous && starts[c] == starts[c - 1] + 64;
      int width = 64;
      if (!contiguous) {                                      // the lines of x the block touches
        width = TILE_SUB;
        starts.clear();
        for (int c : uniq)
          if (starts.empty() || (c & ~(TILE_SUB - 1)) != starts.back()) starts.push_back(c & ~(TILE_SUB - 1));
      }
      if ((int)starts.size() * width > max_slots) continue;   // does not fit: the block gathers from L2
      d.cmin = starts.empty() ? 0 : starts.front();
      d.cwidth = (int)starts.size() * width;
      if (contiguous) d.kind_g |= KIND_CONTIG;
      for (int k = k0; k < k1; k++) {
        const int c = ci[k];
        const int idx = (int)(std::upper_bound(starts.begin(), starts.end(), c) - starts.begin()) - 1;
        ci16[k] = (unsigned short)(idx * width + (c - starts[idx]));
      }
      chunk_starts[b] = starts;
    }
  };
  for_block_ranges(blocks.size(), nnz, work);
}

// The chunk table of a tiled merge plan as the kernel reads it (merge_load): slot u * wg + t of a block's window belongs
// to 16-column chunk s = u * (wg / 16) + (t >> 4); a thread fetches the starts of ITS xu chunks with one or two 16-byte
// loads, so the table is stored [t >> 4][u].  Chunks a block does not use repeat its first one (their loads hit a line
// the block reads anyway).  Windows (KIND_CONTIG) and untiled blocks leave their part of the table unread.
inline void build_chunk_table(const std::vector<BlockDesc> &blocks, const std::vector<std::vector<int>> &chunk_starts, int wg,
                              int xu, std::vector<int> &table) {
  const int groups = wg / TILE_SUB, per_block = groups * xu;
  table.assign(blocks.size() * (size_t)per_block, 0);
  for (size_t b = 0; b < blocks.size(); b++) {
    const BlockDesc &d = blocks[b];
    if ((d.kind_g & (KIND_LONG | KIND_CONTIG)) || d.cwidth <= 0) continue;
    const std::vector<int> &st = chunk_starts[b];
    int *t = table.data() + b * (size_t)per_block;
    for (int s = 0; s < per_block; s++) {
      const int u = s / groups, g = s % groups;
      t[g * xu + u] = s < (int)st.size() ? st[(size_t)s] : st[0];
    }
  }
}

// 12-bit packed slots for the IPT = 8 merge kernel: record (b*wg + t) holds the eight slots thread t of block b
// needs, in the order the kernel consumes them -- pair u (u = 0..3) is elements 2p, 2p+1 of pair index
// p = min(first + u*wg + t, last) exactly as merge_load computes it.  Elements that are not the block's own
// (the lead element of an odd start, the half-foreign last pair, clamped duplicates) get the slot of a
// neighbouring own element, which is what the kernel's fix-up does for the 16-bit layout at run time.
// Stored as 6 unsigned shorts (3 dwords, little endian bit stream) per record.
inline void pack_slots12(int64_t nnz, const std::vector<BlockDesc> &blocks, const std::vector<unsigned short> &ci16, int wg,
                         std::vector<unsigned short> &packed) {
  packed.assign(blocks.size() * (size_t)wg * 6, 0);
  const int max_gpair = (int)((nnz + 1) / 2) - 1;
  auto work = [&](size_t b0, size_t b1) {
    for (size_t b = b0; b < b1; b++) {
      const BlockDesc &d = blocks[b];
      if ((d.kind_g & KIND_LONG) || d.cwidth <= 0) continue;   // long pieces and untiled blocks read 32-bit indices
      const int base = d.nnz_start & ~1, lead = d.nnz_start - base, total = d.nnz_count + lead;
      const int npairs = (total + 1) >> 1, first = base >> 1;
      const int last = std::min(first + std::max(npairs - 1, 0), max_gpair);
      const int own0 = d.nnz_start, own1 = d.nnz_start + d.nnz_count;      // own elements [own0, own1)
      for (int t = 0; t < wg; t++) {
        unsigned slots[8];
        for (int u = 0; u < 4; u++) {
          const int pr = std::min(first + u * wg + t, last);
          int e0 = 2 * pr, e1 = 2 * pr + 1;
          // replace foreign elements by the pair's own element (or the block's first nonzero)
          const bool f0 = e0 < own0 || e0 >= own1, f1 = e1 < own0 || e1 >= own1;
          if (f0 && !f1) e0 = e1;
          if (f1 && !f0) e1 = e0;
          if (f0 && f1) e0 = e1 = own0;
          slots[2 * u] = ci16[e0];
          slots[2 * u + 1] = ci16[e1];
        }
        unsigned w0 = slots[0] | (slots[1] << 12) | (slots[2] << 24);
        unsigned w1 = (slots[2] >> 8) | (slots[3] << 4) | (slots[4] << 16) | (slots[5] << 28);
        unsigned w2 = (slots[5] >> 4) | (slots[6] << 8) | (slots[7] << 20);
        unsigned short *rec = packed.data() + (b * (size_t)wg + t) * 6;
        rec[0] = (unsigned short)(w0 & 0xffff); rec[1] = (unsigned short)(w0 >> 16);
        rec[2] = (unsigned short)(w1 & 0xffff); rec[3] = (unsigned short)(w1 >> 16);
        rec[4] = (unsigned short)(w2 & 0xffff); rec[5] = (unsigned short)(w2 >> 16);
      }
    }
  };
  for_block_ranges(blocks.size(), nnz, work);
}

// Host twin of logical_block() (spmv_common.hpp).
inline int logical_block_host(int hw, int n, bool remap) {
  if (!remap) return hw;
  const int xcd = hw & 7, idx = hw >> 3, q = n >> 3, rem = n & 7;
  return xcd * q + std::min(xcd, rem) + idx;
}

// Sharded product: record every block's largest column in aux (the kernel's seam test) and move the
// seam blocks -- the ones that read halo columns, i.e. wait for a round trip over xGMI -- to the slots
// that are dispatched first, so that their longer life overlaps the rest of the launch instead of
// extending its tail.  chunk_starts (may be empty) is permuted alongside.
inline void place_seam_blocks(const int *ci, int halo_n_own, std::vector<BlockDesc> &blocks,
                              std::vector<std::vector<int>> &chunk_starts, bool remap) {
  const int nb = (int)blocks.size();
  std::vector<int> seam;
  for (int b = 0; b < nb; b++) {
    BlockDesc &d = blocks[b];
    if (d.kind_g & KIND_LONG) continue;                       // long-row pieces test every column themselves
    int cmax = -1;
    for (int k = d.nnz_start; k < d.nnz_start + d.nnz_count; k++) cmax = std::max(cmax, ci[k]);
    d.aux = cmax;
    if (cmax >= halo_n_own) seam.push_back(b);
  }
  if (seam.empty() || (int)seam.size() > nb / 4) return;      // halo everywhere: no order helps
  std::vector<char> is_seam(nb, 0), is_target(nb, 0);
  for (int b : seam) is_seam[b] = 1;
  std::vector<int> targets;
  for (int hw = 0; hw < (int)seam.size(); hw++) {
    const int lb = logical_block_host(hw, nb, remap);
    targets.push_back(lb);
    is_target[lb] = 1;
  }
  size_t ti = 0;
  for (int b : seam) {
    if (is_target[b]) continue;                               // already in an early slot
    while (ti < targets.size() && is_seam[targets[ti]]) ti++; // that slot holds a seam block: leave it
    if (ti == targets.size()) break;
    const int t = targets[ti++];
    std::swap(blocks[b], blocks[t]);
    if (!chunk_starts.empty()) std::swap(chunk_starts[b], chunk_starts[t]);
  }
}

// -------------------------------------------------------------------------------------------------------- VECTOR
// Rows a row-mapped kernel with L lanes per row should not walk itself: more than 16 L nonzeros (at least 32) -- the
// lanes would loop 4+ times over a dependent load -> gather chain (~2 us a turn) while the rest of their wave idles.  They
// become long-row pieces (a wave, or a workgroup of 256 when the pieces are long, strides over <= 4 096 nonzeros; several
// pieces of one row meet in the fix-up).  (First cut, 32 L / at least 64: webbase2 48.8 us at L = 2.)
inline int vector_long_row_len(int lanes_per_row) { return std::max(32, 16 * lanes_per_row); }
constexpr int VECTOR_LONG_PIECE = 4096;
inline void build_vector_long_pieces(const int *rp, int n_rows, int long_len, std::vector<BlockDesc> &longs,
                                     std::vector<SplitRow> &splits, int &n_partial_slots) {
  n_partial_slots = 0;
  for (int r = 0; r < n_rows; r++) {
    const int len = rp[r + 1] - rp[r];
    if (len <= long_len) continue;
    const int n_pieces = (len + VECTOR_LONG_PIECE - 1) / VECTOR_LONG_PIECE;
    if (n_pieces > 1) splits.push_back(SplitRow{r, n_partial_slots, n_pieces, 0});
    for (int pc = 0; pc < n_pieces; pc++) {
      BlockDesc d{};
      d.row_start = r;
      d.n_rows = 1;
      d.nnz_start = rp[r] + pc * VECTOR_LONG_PIECE;
      d.nnz_count = std::min(VECTOR_LONG_PIECE, len - pc * VECTOR_LONG_PIECE);
      d.kind_g = KIND_LONG | (n_pieces > 1 ? KIND_PARTIAL : 0);
      d.aux = n_pieces > 1 ? n_partial_slots++ : 0;
      longs.push_back(d);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------- SCAN
// One word per thread: which of its `ipt` products end a row (bits 0-15) and the ordinal of its first row end among the
// block's non-empty rows (bits 16-31).  Blocks that span empty rows (KIND_HOLES) get a row map at rowmap[d.aux]:
// [number of non-empty rows, their local rows ...].  Clears every block's window (cmin = cwidth = 0).
inline void build_scan_meta(const int *rp, std::vector<BlockDesc> &blocks, int wg, int ipt, std::vector<unsigned> &meta,
                            std::vector<int> &rowmap) {
  meta.assign(blocks.size() * (size_t)wg, 0u);
  std::vector<int> ends((size_t)wg);
  for (size_t b = 0; b < blocks.size(); b++) {
    BlockDesc &d = blocks[b];
    d.cmin = d.cwidth = 0;
    if (d.kind_g & KIND_LONG) continue;
    unsigned *mw = meta.data() + b * (size_t)wg;
    std::fill(ends.begin(), ends.end(), 0);
    const bool holes = (d.kind_g & KIND_HOLES) != 0;
    if (holes) {
      d.aux = (int)rowmap.size();
      rowmap.push_back(0);                                    // [number of non-empty rows, their local rows ...]
    }
    for (int r = 0; r < d.n_rows; r++) {
      const int row = d.row_start + r;
      if (rp[row + 1] == rp[row]) continue;
      const int e = rp[row + 1] - 1 - d.nnz_start;            // the row's last nonzero, block-relative
      mw[e / ipt] |= 1u << (e % ipt);
      ends[e / ipt]++;
      if (holes) {
        rowmap.push_back(r);
        rowmap[(size_t)d.aux]++;
      }
    }
    int ord = 0;
    for (int t = 0; t < wg; t++) {
      mw[t] |= (unsigned)ord << 16;
      ord += ends[t];
    }
  }
}

// Padded SCAN plan: the nonzero-mapped blocks to the front of the list (stable: row order kept), the long-row pieces and
// zero-fill pieces behind them; returns the number of regular blocks.  (The kernel tells the two apart by block index.)
inline int regular_blocks_first(std::vector<BlockDesc> &blocks) {
  auto mid = std::stable_partition(blocks.begin(), blocks.end(), [](const BlockDesc &d) { return !(d.kind_g & KIND_LONG); });
  return (int)(mid - blocks.begin());
}

// ... and its streams: regular block b owns slots [b * cap, (b + 1) * cap) -- its nonzeros in order (columns from `sci`, the
// window-slot stream, when there is one), then padding: source -1 (value +0.0) and the block's first PLAIN column (a gather
// of its own, never a window slot: a padded product is 0 * x[col] in a slot no run covers).  pci gets two spare entries.
inline void build_padded_streams(const std::vector<BlockDesc> &blocks, int n_regular, int cap, const int *ci, const int *sci,
                                 std::vector<int> &pci, std::vector<int> &src) {
  pci.assign((size_t)n_regular * cap + 2, 0);
  src.assign((size_t)n_regular * cap, -1);
  for (int b = 0; b < n_regular; b++) {
    const BlockDesc &d = blocks[(size_t)b];
    const size_t at = (size_t)b * cap;
    for (int i = 0; i < d.nnz_count; i++) {
      pci[at + i] = (sci ? sci : ci)[(size_t)d.nnz_start + i];
      src[at + i] = d.nnz_start + i;
    }
    const int pad_col = d.nnz_count > 0 ? ci[(size_t)d.nnz_start] : 0;
    for (int i = d.nnz_count; i < cap; i++) pci[at + i] = pad_col;
  }
}

// 16-byte window loads per thread for a wanted window of `want_w` entries: 0 (none), 2, 4 or 8.  The window SHARES the
// product area's LDS (scan_kernel.hpp: dead once every thread holds its x values), so it is at most ipt * wg entries wide
// (xp <= ipt / 2) and costs no LDS -- with 16 KB of its own a 2 048-entry window took a 256 x 8 block from 8 to 4
// workgroups per CU, which is what made windows lose on short rows (webbase2: 15.9 us with its own LDS, 15.2 shared, 16.2
// without a window; profiles/r05_merge_forms.txt).
inline int scan_window_xp(int want_w, int wg, int ipt) {
  if (want_w <= 0) return 0;
  int xp = 2;
  while (xp < 8 && 2 * xp * wg < want_w) xp *= 2;
  while (xp >= 2 && (2 * xp > ipt + 1 || 2 * xp * wg > 65536)) xp /= 2;
  return xp < 2 ? 0 : xp;
}

// Window: per block the contiguous column range of at most W entries that covers most of its nonzeros, staged in LDS; a
// nonzero inside it streams an LDS slot instead of a column (sci, the plan's own column stream: a copy of ci + one spare
// element on entry).  Returns the number of nonzeros served from windows.
inline long build_scan_window(const int *ci, int64_t nnz, std::vector<BlockDesc> &blocks, int W, std::vector<int> &sci) {
  const size_t nb = blocks.size();
  std::vector<long> covered(nb, 0);
  auto work = [&](size_t b0, size_t b1) {
    std::vector<int> cols;
    for (size_t b = b0; b < b1; b++) {
      BlockDesc &d = blocks[b];
      if ((d.kind_g & KIND_LONG) || d.nnz_count == 0) continue;
      cols.assign(ci + d.nnz_start, ci + d.nnz_start + d.nnz_count);
      std::sort(cols.begin(), cols.end());
      // densest range [s, s + W) with s even: two pointers over the sorted columns
      size_t best_i = 0, best_n = 0, j = 0;
      for (size_t i = 0; i < cols.size(); i++) {
        const int s0 = cols[i] & ~1;
        while (j < cols.size() && cols[j] < s0 + W) j++;
        if (j - i > best_n) { best_n = j - i; best_i = i; }
      }
      if (best_n * 4 < cols.size()) continue;                 // a window that serves under a quarter is not worth its loads
      const int s0 = cols[best_i] & ~1, last_col = cols[best_i + best_n - 1];
      d.cmin = s0;
      d.cwidth = ((last_col - s0 + 2) & ~1);                  // even, <= W
      for (int k = d.nnz_start; k < d.nnz_start + d.nnz_count; k++)
        if (ci[k] >= s0 && ci[k] < s0 + d.cwidth) sci[(size_t)k] = SCAN_LDS_BIT | (ci[k] - s0);
      covered[b] = (long)best_n;
    }
  };
  for_block_ranges(nb, nnz, work);
  long in_window = 0;
  for (long c : covered) in_window += c;
  return in_window;
}

// --------------------------------------------------------------------------------------------------------- SLICE
// Variant SLICE (slice_kernel.hpp): the rows of a matrix whose rows are mostly very short (webbase-1M: 65-79 % of the
// rows hold ONE nonzero, 92 % at most four -- and those 92 % hold only 40 % of the nonzeros) are split by length.
//   * Rows of at most K nonzeros (empty rows included) are ROW-MAPPED: the matrix is cut into windows of `rows_per_block`
//     consecutive rows, one workgroup each; inside a window the short rows are sorted by length (longest first, ties in
//     row order) and stored as jagged planes -- plane j holds the j-th nonzero of every row that has one, which after the
//     sort is a PREFIX of the sorted order, so a plane is cnt[j] consecutive (value, column) elements, a wave's loads of
//     it are consecutive addresses, and nothing is padded.  A thread owns a sorted position, adds its row's <= K
//     products in stored order in a register, and the sums return to row order through LDS by a 16-bit slot map
//     (slot[row] = its sorted position) so that y leaves in coalesced stores.
//   * Longer rows go to the nonzero-mapped blocks of the SAME launch: a compacted copy of them (values and columns in row
//     order) is planned exactly like a SCAN matrix -- blocks of <= cap - 1 nonzeros, row-end words, x window -- except that
//     every block carries a row map (KIND_HOLES | KIND_NOFILL: the rows in between are the slices') and pieces of rows
//     longer than cap / 2 name their row directly.
// The plan owns copies of the value and column streams in this order (12 B per nonzero of capacity: HBM holds 288 GB);
// `slice_src` / `long_src` say which element of the caller's arrays each of them is.
// Reference: the same idea in the dataflow design -- ParallelCsrReadControl.java:119-145,262-276 packs several short rows
// into one cycle under a lane mask, SpmvKernel.java:68-78 skips runs of empty rows.
struct SlicePlan {
  int k = 0, rows_per_block = 0;
  std::vector<SliceDesc> slices;
  std::vector<uint16_t> slot;            // [n_rows]
  std::vector<int> slice_src;            // [nonzeros of short rows] index into the caller's arrays, plane order
  // the long rows as a sub-matrix
  std::vector<int> long_rows;            // their row numbers
  std::vector<int> long_rp;              // [long_rows + 1]
  std::vector<int> long_src;             // [nonzeros of long rows] index into the caller's arrays, row order
  std::vector<BlockDesc> blocks;         // SCAN blocks over the sub-matrix (row_start rewritten to real rows)
  std::vector<SplitRow> splits;
  std::vector<unsigned> meta;
  std::vector<int> rowmap;
  int n_long_pieces_rows = 0, n_partial_slots = 0;
};

inline void build_slice_plan(const int *rp, int n_rows, int k, int rows_per_block, int wg, int ipt, SlicePlan &sp) {
  sp = SlicePlan{};
  sp.k = k;
  sp.rows_per_block = rows_per_block;
  sp.slot.assign((size_t)n_rows, SLICE_NOT_MINE);
  sp.long_rp.push_back(0);
  std::vector<int> order;
  int64_t slice_nnz = 0;
  for (int r = 0; r < n_rows; r++) {
    const int len = rp[r + 1] - rp[r];
    if (len <= k) slice_nnz += len;
  }
  sp.slice_src.reserve((size_t)slice_nnz);
  for (int w0 = 0; w0 < n_rows; w0 += rows_per_block) {
    const int w1 = std::min(n_rows, w0 + rows_per_block);
    order.clear();
    for (int r = w0; r < w1; r++) {
      const int len = rp[r + 1] - rp[r];
      if (len <= k) {
        order.push_back(r);
      } else {
        sp.long_rows.push_back(r);
        for (int e = rp[r]; e < rp[r + 1]; e++) sp.long_src.push_back(e);
        sp.long_rp.push_back((int)sp.long_src.size());
      }
    }
    if (order.empty()) continue;                              // a window of long rows only: no slice block
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rp[a + 1] - rp[a] > rp[b + 1] - rp[b]; });
    SliceDesc d{};
    d.row_start = w0;
    d.n_rows = w1 - w0;
    d.nnz_start = (int)sp.slice_src.size();
    d.n_short = (int)order.size();
    for (size_t s = 0; s < order.size(); s++) sp.slot[(size_t)order[s]] = (uint16_t)s;
    for (int j = 0; j < SLICE_KMAX; j++) {
      int cnt = 0;
      if (j < k)
        for (int r : order) {
          if (rp[r + 1] - rp[r] <= j) break;                  // sorted: everyone behind is shorter still
          sp.slice_src.push_back(rp[r] + j);
          cnt++;
        }
      d.cnt[j] = (uint16_t)cnt;
    }
    sp.slices.push_back(d);
  }
  // the long rows: a SCAN plan over the compacted sub-matrix
  const int n_long = (int)sp.long_rows.size();
  if (n_long == 0) return;
  const int cap = wg * ipt;
  build_merge_blocks(sp.long_rp.data(), n_long, cap - 1, 1 << 30, (long)cap * LONG_PIECE_FACTOR, wg, sp.blocks, nullptr, sp.splits,
                     sp.n_long_pieces_rows, sp.n_partial_slots, false);
  build_scan_meta(sp.long_rp.data(), sp.blocks, wg, ipt, sp.meta, sp.rowmap);
  for (BlockDesc &d : sp.blocks) {
    if (d.kind_g & KIND_LONG) {                               // a piece of one row (a sub-matrix has no empty rows): its real row
      d.row_start = sp.long_rows[(size_t)d.row_start];
      continue;
    }
    const int first = sp.long_rows[(size_t)d.row_start];
    d.kind_g |= KIND_HOLES | KIND_NOFILL;
    d.aux = (int)sp.rowmap.size();
    sp.rowmap.push_back(d.n_rows);
    for (int r = 0; r < d.n_rows; r++) sp.rowmap.push_back(sp.long_rows[(size_t)(d.row_start + r)] - first);
    d.row_start = first;
  }
  for (SplitRow &s : sp.splits) s.row = sp.long_rows[(size_t)s.row];
  sp.rowmap.push_back(0);                                     // never empty: the kernel forms rowmap + aux
}

}  // namespace plan
}  // namespace caskhip
