// The host planners of the launch plans (cask_amd/csrc/plan_host.hpp, trsv_lanes_plan.hpp) on the CPU, under
// AddressSanitizer / UBSan (`make asan`; tests/test_host_cpp.py::test_planners_under_sanitizers).  The planners write
// what the kernels trust blindly -- nonzero ranges, 12-bit slot records, LDS slots, lane words, absolute LDS byte
// addresses, 16-byte units of a byte image -- and on the GPU box they only ever run behind cask_hip_csr_create, where a
// read one past the end that hits mapped memory goes unnoticed.  Here every plan is built for every
// matrix given (the reference's fixtures as .mtx, the small synthetic BASELINE families as raw CSR dumps) and the
// invariants the kernels rely on are checked:
//   merge / scan blocks   every row in exactly one block or piece, nonzero ranges a partition of [0, nnz), caps held
//   chunk tiles           slot < tile, chunk start + offset = the nonzero's column
//   12-bit records        what the kernel's thread t unpacks = the slot of the element it consumes (foreign ones replaced)
//   scan words / rowmap   one row-end bit per non-empty row, ordinals running, row map in range
//   scan window           a window slot decodes to the nonzero's column
//   slice plan            short rows: each in one position of one block, planes a prefix of the sorted order, sources a
//                         permutation; long rows: the sub-matrix plan covers them, row maps name real rows
//   vector long pieces    exactly the rows beyond the bound, pieces tile the row
//   lane-group runs       chunk images whole, lane words and LDS addresses in range, every entry placed once
// usage: test_planners <dir of .mtx fixtures> [<dir of *.csr dumps>]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <vector>

#include <dirent.h>

#include "cask/IO.hpp"
#include "plan_host.hpp"
#include "trsv_lanes_plan.hpp"

using namespace caskhip;

static long g_checks = 0, g_failures = 0, g_lane_runs = 0, g_lane_chunks = 0;
static std::string g_ctx;
#define CHECK(cond)                                                                                  \
  do {                                                                                               \
    g_checks++;                                                                                      \
    if (!(cond)) {                                                                                   \
      if (g_failures++ < 20) std::printf("FAIL %s:%d [%s] %s\n", __FILE__, __LINE__, g_ctx.c_str(), #cond); \
    }                                                                                                \
  } while (0)

struct Csr {
  std::string name;
  int n = 0, m = 0;
  std::vector<int> rp, ci;
  std::vector<double> va;
  int64_t nnz() const { return (int64_t)ci.size(); }
};

static std::vector<std::string> list_dir(const std::string &dir, const std::string &suffix) {
  std::vector<std::string> out;
  if (DIR *d = opendir(dir.c_str())) {
    while (dirent *e = readdir(d)) {
      const std::string f = e->d_name;
      if (f.size() > suffix.size() && f.compare(f.size() - suffix.size(), suffix.size(), suffix) == 0) out.push_back(dir + "/" + f);
    }
    closedir(d);
  }
  std::sort(out.begin(), out.end());
  return out;
}

static bool load_dump(const std::string &path, Csr &a) {      // int32 n, m, nnz; rp[n+1]; ci[nnz]; f64 va[nnz]
  std::ifstream f(path, std::ios::binary);
  int hdr[3];
  if (!f.read(reinterpret_cast<char *>(hdr), sizeof(hdr))) return false;
  a.n = hdr[0]; a.m = hdr[1];
  a.rp.resize((size_t)a.n + 1); a.ci.resize((size_t)hdr[2]); a.va.resize((size_t)hdr[2]);
  f.read(reinterpret_cast<char *>(a.rp.data()), (std::streamsize)(a.rp.size() * 4));
  f.read(reinterpret_cast<char *>(a.ci.data()), (std::streamsize)(a.ci.size() * 4));
  f.read(reinterpret_cast<char *>(a.va.data()), (std::streamsize)(a.va.size() * 8));
  return (bool)f;
}

// ------------------------------------------------------------------------------------------------- merge / scan
static void check_blocks(const Csr &a, const std::vector<BlockDesc> &blocks, const std::vector<BlockDesc> *longs,
                         const std::vector<SplitRow> &splits, int cap, bool rows_are_items, int n_slots) {
  std::vector<int> row_owner((size_t)a.n, 0);
  std::vector<char> nz((size_t)a.nnz(), 0);
  std::map<int, int> piece_sum;
  auto visit = [&](const BlockDesc &d) {
    CHECK(d.nnz_start >= 0 && d.nnz_count >= 0 && (int64_t)d.nnz_start + d.nnz_count <= a.nnz());
    CHECK(d.row_start >= 0 && d.row_start + d.n_rows <= a.n);
    if ((int64_t)d.nnz_start + d.nnz_count > a.nnz() || d.row_start + d.n_rows > a.n) return;
    for (int k = d.nnz_start; k < d.nnz_start + d.nnz_count; k++) nz[(size_t)k]++;
    if (d.kind_g & KIND_LONG) {
      if (d.nnz_count == 0) {                                 // zero-fill piece: a run of empty rows
        for (int r = 0; r < d.n_rows; r++) { CHECK(a.rp[d.row_start + r] == a.rp[d.row_start + r + 1]); row_owner[(size_t)d.row_start + r]++; }
      } else {
        CHECK(d.n_rows == 1);
        CHECK(d.nnz_start >= a.rp[d.row_start] && d.nnz_start + d.nnz_count <= a.rp[d.row_start + 1]);
        piece_sum[d.row_start] += d.nnz_count;
        if (d.kind_g & KIND_PARTIAL) CHECK(d.aux >= 0 && d.aux < n_slots);
      }
    } else {
      CHECK(d.nnz_start == a.rp[d.row_start] && d.nnz_start + d.nnz_count == a.rp[d.row_start + d.n_rows]);
      CHECK(d.nnz_count + (rows_are_items ? d.n_rows : 0) <= cap);
      for (int r = 0; r < d.n_rows; r++) row_owner[(size_t)d.row_start + r]++;
      if (rows_are_items) {
        const int g = d.kind_g & 0xff;
        CHECK(g >= 1 && g <= 64 && (g & (g - 1)) == 0);
      }
    }
  };
  for (const BlockDesc &d : blocks) visit(d);
  if (longs) for (const BlockDesc &d : *longs) visit(d);
  for (auto &kv : piece_sum) {
    CHECK(kv.second == a.rp[kv.first + 1] - a.rp[kv.first]);
    row_owner[(size_t)kv.first]++;
  }
  for (int r = 0; r < a.n; r++) CHECK(row_owner[(size_t)r] == 1);
  for (int64_t k = 0; k < a.nnz(); k++) CHECK(nz[(size_t)k] == 1);
  std::set<int> slots;
  for (const SplitRow &s : splits) {
    CHECK(s.row >= 0 && s.row < a.n && s.n_slots > 1 && s.first_slot >= 0 && s.first_slot + s.n_slots <= n_slots);
    for (int k = 0; k < s.n_slots; k++) CHECK(slots.insert(s.first_slot + k).second);
  }
}

static void merge_plans(const Csr &a) {
  for (int wg : {64, 256, 512})
    for (int ipt : {2, 8, 16}) {
      const int cap = wg * ipt;
      std::vector<BlockDesc> blocks;
      std::vector<SplitRow> splits;
      int n_long = 0, n_slots = 0;
      plan::build_merge_blocks(a.rp.data(), a.n, cap, 2 * wg - 1, (long)cap * LONG_PIECE_FACTOR, wg, blocks, nullptr, splits, n_long, n_slots);
      check_blocks(a, blocks, nullptr, splits, cap, true, n_slots);
      for (const BlockDesc &d : blocks)
        if (!(d.kind_g & KIND_LONG)) CHECK(d.n_rows <= 2 * wg - 1);
      if (a.nnz() == 0) continue;
      for (int tile : {64, 1024, 4096}) {
        std::vector<BlockDesc> tb(blocks);
        std::vector<std::vector<int>> chunk_starts;
        std::vector<unsigned short> ci16;
        plan::build_chunk_tiles(a.ci.data(), a.nnz(), tb, tile, chunk_starts, ci16);
        CHECK(ci16.size() == (size_t)a.nnz() + 8 && chunk_starts.size() == tb.size());
        int max_used = 0;
        for (size_t b = 0; b < tb.size(); b++) {
          const BlockDesc &d = tb[b];
          if ((d.kind_g & KIND_LONG) || d.cwidth == 0) continue;
          max_used = std::max(max_used, d.cwidth);
          // a window: 64-column chunks from an even column; anything else: the 16-column lines of x the block touches
          const int width = (d.kind_g & KIND_CONTIG) ? 64 : plan::TILE_SUB;
          CHECK(d.cwidth <= tile && d.cwidth == width * (int)chunk_starts[b].size() && (d.cmin & 1) == 0);
          std::vector<char> used(chunk_starts[b].size(), 0);
          for (int k = d.nnz_start; k < d.nnz_start + d.nnz_count; k++) {
            const int slot = ci16[(size_t)k];
            CHECK(slot < d.cwidth);
            if (slot < d.cwidth) {
              CHECK(chunk_starts[b][(size_t)(slot / width)] + slot % width == a.ci[(size_t)k]);
              used[(size_t)(slot / width)] = 1;
            }
          }
          if (d.kind_g & KIND_CONTIG)
            for (size_t c = 0; c < chunk_starts[b].size(); c++) CHECK(chunk_starts[b][c] == d.cmin + 64 * (int)c);
          else
            for (size_t c = 0; c < chunk_starts[b].size(); c++) {   // lines: aligned, ascending, every one of them needed
              CHECK(chunk_starts[b][c] % plan::TILE_SUB == 0 && used[c]);
              if (c) CHECK(chunk_starts[b][c] > chunk_starts[b][c - 1]);
            }
        }
        if (max_used > 0 && wg >= plan::TILE_SUB) {           // the table as merge_load reads it: thread t, turn u -> slot u * wg + t
          int xu = 1;
          while (xu * wg < max_used) xu *= 2;
          std::vector<int> table;
          plan::build_chunk_table(tb, chunk_starts, wg, xu, table);
          const int per_block = xu * wg / plan::TILE_SUB;
          CHECK(table.size() == tb.size() * (size_t)per_block);
          for (size_t b = 0; b < tb.size(); b++) {
            const BlockDesc &d = tb[b];
            if ((d.kind_g & (KIND_LONG | KIND_CONTIG)) || d.cwidth == 0) continue;
            for (int u = 0; u < xu; u++)
              for (int t = 0; t < wg; t++) {
                const int slot = u * wg + t, got = table[b * (size_t)per_block + (size_t)(t >> 4) * xu + u] + (t & 15);
                const size_t s = (size_t)slot / plan::TILE_SUB;
                CHECK(got == (s < chunk_starts[b].size() ? chunk_starts[b][s] : chunk_starts[b][0]) + slot % plan::TILE_SUB);
              }
          }
        }
        if (ipt == 8 && tile <= 4096) {
          std::vector<unsigned short> packed;
          plan::pack_slots12(a.nnz(), tb, ci16, wg, packed);
          CHECK(packed.size() == tb.size() * (size_t)wg * 6);
          const int max_gpair = (int)((a.nnz() + 1) / 2) - 1;
          for (size_t b = 0; b < tb.size(); b++) {
            const BlockDesc &d = tb[b];
            if ((d.kind_g & KIND_LONG) || d.cwidth <= 0) continue;
            const int base = d.nnz_start & ~1, lead = d.nnz_start - base, total = d.nnz_count + lead;
            const int npairs = (total + 1) >> 1, first = base >> 1, last = std::min(first + std::max(npairs - 1, 0), max_gpair);
            for (int t = 0; t < wg; t++) {
              const unsigned short *rec = packed.data() + (b * (size_t)wg + t) * 6;
              const unsigned w0 = rec[0] | ((unsigned)rec[1] << 16), w1 = rec[2] | ((unsigned)rec[3] << 16), w2 = rec[4] | ((unsigned)rec[5] << 16);
              // the kernel's unpack (merge_kernel.hpp): eight 12-bit fields of a 96-bit little-endian stream
              const unsigned s[8] = {w0 & 0xfff, (w0 >> 12) & 0xfff, ((w0 >> 24) | (w1 << 8)) & 0xfff, (w1 >> 4) & 0xfff,
                                     (w1 >> 16) & 0xfff, ((w1 >> 28) | (w2 << 4)) & 0xfff, (w2 >> 8) & 0xfff, (w2 >> 20) & 0xfff};
              for (int u = 0; u < 4; u++) {
                const int pr = std::min(first + u * wg + t, last);
                for (int h = 0; h < 2; h++) {
                  const int e = 2 * pr + h;
                  CHECK((int)s[2 * u + h] < d.cwidth);
                  if (e >= d.nnz_start && e < d.nnz_start + d.nnz_count) CHECK(s[2 * u + h] == ci16[(size_t)e]);
                }
              }
            }
          }
        }
        // seam placement keeps the set of blocks (and their chunk lists) intact
        std::vector<BlockDesc> sb(tb);
        std::vector<std::vector<int>> cs(chunk_starts);
        plan::place_seam_blocks(a.ci.data(), a.m - a.m / 5, sb, cs, true);
        std::multiset<long> before, after;
        for (const BlockDesc &d : tb) before.insert(((long)d.row_start << 32) | (unsigned)d.nnz_start);
        for (const BlockDesc &d : sb) after.insert(((long)d.row_start << 32) | (unsigned)d.nnz_start);
        CHECK(before == after);
        for (size_t b = 0; b < sb.size(); b++)
          if (!(sb[b].kind_g & KIND_LONG) && sb[b].cwidth > 0) CHECK((int)cs[b].size() * ((sb[b].kind_g & KIND_CONTIG) ? 64 : plan::TILE_SUB) == sb[b].cwidth);
      }
      // the pipelined plan: long pieces in their own list
      std::vector<BlockDesc> wb, wl;
      std::vector<SplitRow> ws;
      plan::build_merge_blocks(a.rp.data(), a.n, 64 * ipt, 127, 32768, 64, wb, &wl, ws, n_long, n_slots);
      check_blocks(a, wb, &wl, ws, 64 * ipt, true, n_slots);
    }
}

static void check_scan_words(const int *rp, const std::vector<BlockDesc> &blocks, int wg, int ipt, const std::vector<unsigned> &meta,
                             const std::vector<int> &rowmap, bool sub) {
  CHECK(meta.size() == blocks.size() * (size_t)wg);
  for (size_t b = 0; b < blocks.size(); b++) {
    const BlockDesc &d = blocks[b];
    if (d.kind_g & KIND_LONG) continue;
    CHECK(d.nnz_count <= wg * ipt - 1);
    int ends = 0, ord = 0;
    for (int t = 0; t < wg; t++) {
      const unsigned w = meta[b * (size_t)wg + t];
      CHECK((int)(w >> 16) == ord);
      CHECK((w & 0xffffu) >> ipt == 0);
      for (int j = 0; j < ipt; j++)
        if ((w >> j) & 1u) { ends++; ord++; CHECK(t * ipt + j < d.nnz_count); }
    }
    if (d.kind_g & KIND_HOLES) {
      CHECK(d.aux >= 0 && (size_t)d.aux < rowmap.size());
      const int cnt = rowmap[(size_t)d.aux];
      CHECK(cnt == ends && (size_t)d.aux + 1 + cnt <= rowmap.size());
      int prev = -1;
      for (int r = 0; r < cnt; r++) { const int lr = rowmap[(size_t)d.aux + 1 + r]; CHECK(lr > prev); prev = lr; }
      if (!sub) { CHECK(prev < d.n_rows); }
    } else if (!sub) {
      int nonempty = 0;
      for (int r = 0; r < d.n_rows; r++) nonempty += rp[d.row_start + r + 1] > rp[d.row_start + r];
      CHECK(nonempty == ends && nonempty == d.n_rows);
    }
  }
}

static void scan_plans(const Csr &a) {
  for (int wg : {64, 256})
    for (int ipt : {2, 8, 16}) {
      const int cap = wg * ipt;
      std::vector<BlockDesc> blocks;
      std::vector<SplitRow> splits;
      int n_long = 0, n_slots = 0;
      plan::build_merge_blocks(a.rp.data(), a.n, cap - 1, 1 << 30, (long)cap * LONG_PIECE_FACTOR, wg, blocks, nullptr, splits, n_long, n_slots, false);
      check_blocks(a, blocks, nullptr, splits, cap - 1, false, n_slots);
      std::vector<unsigned> meta;
      std::vector<int> rowmap;
      plan::build_scan_meta(a.rp.data(), blocks, wg, ipt, meta, rowmap);
      check_scan_words(a.rp.data(), blocks, wg, ipt, meta, rowmap, false);
      {                                                       // the padded plan: regular blocks first, streams a function of the index
        std::vector<BlockDesc> pb(blocks);
        const int n_regular = plan::regular_blocks_first(pb);
        CHECK(n_regular >= 0 && (size_t)n_regular <= pb.size());
        for (size_t b = 0; b < pb.size(); b++) CHECK(((pb[b].kind_g & KIND_LONG) != 0) == ((int)b >= n_regular));
        for (int b = 1; b < n_regular; b++) CHECK(pb[(size_t)b].row_start > pb[(size_t)b - 1].row_start);   // row order kept
        std::vector<SplitRow> none;
        check_blocks(a, pb, nullptr, splits, cap - 1, false, n_slots);                                   // still a partition
        if (a.nnz() > 0 && (int64_t)n_regular * cap < (1LL << 28)) {
          std::vector<int> pci, src;
          plan::build_padded_streams(pb, n_regular, cap, a.ci.data(), nullptr, pci, src);
          CHECK(pci.size() == (size_t)n_regular * cap + 2 && src.size() == (size_t)n_regular * cap);
          std::vector<char> hit((size_t)a.nnz(), 0);
          for (int b = 0; b < n_regular; b++)
            for (int i = 0; i < cap; i++) {
              const size_t at = (size_t)b * cap + i;
              const int e = src[at];
              if (i < pb[(size_t)b].nnz_count) {
                CHECK(e == pb[(size_t)b].nnz_start + i && pci[at] == a.ci[(size_t)e]);
                if (e >= 0 && e < a.nnz()) hit[(size_t)e]++;
              } else {
                CHECK(e == -1 && pci[at] >= 0 && pci[at] < a.m);                                         // padding: a real column, no source
              }
            }
          for (const BlockDesc &d : pb)
            if (d.kind_g & KIND_LONG)
              for (int k = d.nnz_start; k < d.nnz_start + d.nnz_count; k++) hit[(size_t)k]++;
          for (int64_t k = 0; k < a.nnz(); k++) CHECK(hit[(size_t)k] == 1);
        }
      }
      const int xp = plan::scan_window_xp(2048, wg, ipt);
      CHECK(xp == 0 || (2 * xp <= ipt + 1 && 2 * xp * wg <= 65536));
      if (xp > 0 && a.nnz() > 0) {
        std::vector<int> sci(a.ci);
        sci.push_back(0);
        const int W = 2 * xp * wg;
        const long in_w = plan::build_scan_window(a.ci.data(), a.nnz(), blocks, W, sci);
        long seen = 0;
        for (const BlockDesc &d : blocks) {
          if (d.kind_g & KIND_LONG) continue;
          CHECK(d.cwidth >= 0 && d.cwidth <= W && (d.cmin & 1) == 0 && (d.cwidth & 1) == 0);
          for (int k = d.nnz_start; k < d.nnz_start + d.nnz_count; k++) {
            const int c = sci[(size_t)k];
            if (c & SCAN_LDS_BIT) { seen++; CHECK((c & 0xffff) < d.cwidth && d.cmin + (c & 0xffff) == a.ci[(size_t)k]); }
            else CHECK(c == a.ci[(size_t)k]);
          }
        }
        CHECK(seen == in_w);
      }
    }
}

static void slice_plans(const Csr &a) {
  for (int k : {1, 2, 3, 4, 5, 8})
    for (int wg : {64, 256}) {
      const int ipt = 8, rows_per_block = slice_rows_per_thread(slice_kernel_km(k)) * wg;
      plan::SlicePlan sp;
      plan::build_slice_plan(a.rp.data(), a.n, k, rows_per_block, wg, ipt, sp);
      std::vector<char> seen((size_t)a.nnz(), 0);
      std::vector<int> owner((size_t)a.n, 0);
      CHECK(sp.slot.size() == (size_t)a.n);
      int64_t at = 0;
      for (const SliceDesc &d : sp.slices) {
        CHECK(d.row_start % rows_per_block == 0 && d.n_rows >= 1 && d.n_rows <= rows_per_block && d.row_start + d.n_rows <= a.n);
        CHECK(d.nnz_start == at && d.n_short >= 1 && d.n_short <= d.n_rows);
        std::vector<int> pos_row((size_t)d.n_short, -1);
        for (int r = d.row_start; r < d.row_start + d.n_rows; r++) {
          const int len = a.rp[(size_t)r + 1] - a.rp[(size_t)r];
          if (len > k) { CHECK(sp.slot[(size_t)r] == SLICE_NOT_MINE); continue; }
          const int s = sp.slot[(size_t)r];
          CHECK(s < d.n_short);
          if (s < d.n_short) { CHECK(pos_row[(size_t)s] == -1); pos_row[(size_t)s] = r; }
          owner[(size_t)r]++;
        }
        int prev_len = 1 << 30;
        for (int s = 0; s < d.n_short; s++) {                 // sorted: lengths never grow along the positions
          CHECK(pos_row[(size_t)s] >= 0);
          if (pos_row[(size_t)s] < 0) continue;
          const int len = a.rp[(size_t)pos_row[(size_t)s] + 1] - a.rp[(size_t)pos_row[(size_t)s]];
          CHECK(len <= prev_len);
          prev_len = len;
        }
        for (int j = 0; j < SLICE_KMAX; j++) {
          int want = 0;
          for (int s = 0; s < d.n_short; s++)
            if (pos_row[(size_t)s] >= 0) want += a.rp[(size_t)pos_row[(size_t)s] + 1] - a.rp[(size_t)pos_row[(size_t)s]] > j;
          CHECK(d.cnt[j] == want);
          if (j >= k) CHECK(d.cnt[j] == 0);
          for (int s = 0; s < d.cnt[j]; s++, at++) {
            CHECK((size_t)at < sp.slice_src.size());
            if ((size_t)at >= sp.slice_src.size() || pos_row[(size_t)s] < 0) continue;
            const int e = sp.slice_src[(size_t)at];
            CHECK(e == a.rp[(size_t)pos_row[(size_t)s]] + j);  // plane j, position s = the j-th nonzero of that row
            if (e >= 0 && e < a.nnz()) seen[(size_t)e]++;
          }
        }
      }
      CHECK((size_t)at == sp.slice_src.size());
      // the long rows' sub-matrix
      CHECK(sp.long_rp.size() == sp.long_rows.size() + 1 && (sp.long_rp.empty() || (size_t)sp.long_rp.back() == sp.long_src.size()));
      for (size_t i = 0; i < sp.long_rows.size(); i++) {
        const int r = sp.long_rows[i];
        CHECK(a.rp[(size_t)r + 1] - a.rp[(size_t)r] > k && sp.long_rp[i + 1] - sp.long_rp[i] == a.rp[(size_t)r + 1] - a.rp[(size_t)r]);
        for (int e = sp.long_rp[i]; e < sp.long_rp[i + 1]; e++) {
          CHECK(sp.long_src[(size_t)e] == a.rp[(size_t)r] + (e - sp.long_rp[i]));
          if (sp.long_src[(size_t)e] >= 0 && sp.long_src[(size_t)e] < a.nnz()) seen[(size_t)sp.long_src[(size_t)e]]++;
        }
      }
      for (int64_t e = 0; e < a.nnz(); e++) CHECK(seen[(size_t)e] == 1);
      if (!sp.long_rows.empty()) {
        check_scan_words(sp.long_rp.data(), sp.blocks, wg, ipt, sp.meta, sp.rowmap, true);
        std::map<int, int> pieces;
        for (const BlockDesc &d : sp.blocks) {
          CHECK(d.nnz_start >= 0 && (size_t)(d.nnz_start + d.nnz_count) <= sp.long_src.size());
          if (d.kind_g & KIND_LONG) {
            CHECK(d.row_start >= 0 && d.row_start < a.n && a.rp[(size_t)d.row_start + 1] - a.rp[(size_t)d.row_start] > k);
            pieces[d.row_start] += d.nnz_count;
            continue;
          }
          CHECK((d.kind_g & KIND_HOLES) && (d.kind_g & KIND_NOFILL));
          const int cnt = sp.rowmap[(size_t)d.aux];
          CHECK(cnt == d.n_rows);
          int64_t sum = 0;
          for (int r = 0; r < cnt; r++) {
            const int row = d.row_start + sp.rowmap[(size_t)d.aux + 1 + r];
            CHECK(row >= 0 && row < a.n);
            if (row < 0 || row >= a.n) continue;
            CHECK(a.rp[(size_t)row + 1] - a.rp[(size_t)row] > k);
            owner[(size_t)row]++;
            sum += a.rp[(size_t)row + 1] - a.rp[(size_t)row];
          }
          CHECK(sum == d.nnz_count);
        }
        for (auto &kv : pieces) { CHECK(kv.second == a.rp[(size_t)kv.first + 1] - a.rp[(size_t)kv.first]); owner[(size_t)kv.first]++; }
        for (const SplitRow &s : sp.splits) CHECK(s.row >= 0 && s.row < a.n && s.first_slot + s.n_slots <= sp.n_partial_slots);
      }
      for (int r = 0; r < a.n; r++) CHECK(owner[(size_t)r] == 1);
    }
}

static void vector_pieces(const Csr &a) {
  for (int lanes : {1, 2, 16, 64}) {
    const int bound = plan::vector_long_row_len(lanes);
    std::vector<BlockDesc> longs;
    std::vector<SplitRow> splits;
    int n_slots = 0;
    plan::build_vector_long_pieces(a.rp.data(), a.n, bound, longs, splits, n_slots);
    std::map<int, int> sum;
    for (const BlockDesc &d : longs) {
      CHECK((d.kind_g & KIND_LONG) && d.n_rows == 1 && d.nnz_count > 0 && d.nnz_count <= plan::VECTOR_LONG_PIECE);
      CHECK(d.nnz_start >= a.rp[(size_t)d.row_start] && d.nnz_start + d.nnz_count <= a.rp[(size_t)d.row_start + 1]);
      sum[d.row_start] += d.nnz_count;
    }
    for (int r = 0; r < a.n; r++) {
      const int len = a.rp[(size_t)r + 1] - a.rp[(size_t)r];
      CHECK((len > bound) == (sum.count(r) == 1));
      if (sum.count(r)) CHECK(sum[r] == len);
    }
  }
}

// ---------------------------------------------------------------------------------------------- lane-group runs
// A lower-triangular factor from the matrix's strict lower part (unit diagonal positions implied): levels, position
// space, then build_lanes_run over every run of narrow levels -- what cask_hip_precond.hip does before it uploads.
static void lanes_runs(const Csr &a) {
  using namespace caskhip_lanes;
  if (a.n != a.m || a.n == 0 || a.n > 40000) return;
  const int n = a.n;
  std::vector<int> level((size_t)n, 0);
  int n_levels = 0;
  for (int r = 0; r < n; r++) {
    int l = 0;
    for (int e = a.rp[(size_t)r]; e < a.rp[(size_t)r + 1]; e++)
      if (a.ci[(size_t)e] < r) l = std::max(l, level[(size_t)a.ci[(size_t)e]] + 1);
    level[(size_t)r] = l;
    n_levels = std::max(n_levels, l + 1);
  }
  std::vector<int> lp((size_t)n_levels + 1, 0), ord((size_t)n), pos((size_t)n);
  for (int r = 0; r < n; r++) lp[(size_t)level[(size_t)r] + 1]++;
  for (int l = 0; l < n_levels; l++) lp[(size_t)l + 1] += lp[(size_t)l];
  std::vector<int> fill(lp.begin(), lp.end() - 1);
  for (int r = 0; r < n; r++) ord[(size_t)fill[(size_t)level[(size_t)r]]++] = r;
  for (int i = 0; i < n; i++) pos[(size_t)ord[(size_t)i]] = i;
  std::vector<int> peptr((size_t)n + 1, 0), ppos;
  std::vector<double> pval;
  for (int i = 0; i < n; i++) {
    const int r = ord[(size_t)i];
    for (int e = a.rp[(size_t)r]; e < a.rp[(size_t)r + 1]; e++)
      if (a.ci[(size_t)e] < r) { ppos.push_back(pos[(size_t)a.ci[(size_t)e]]); pval.push_back(a.va[(size_t)e]); }
    peptr[(size_t)i + 1] = (int)ppos.size();
  }
  for (const char *force : {(const char *)nullptr, "4", "8", "16"}) {
    if (force) setenv("CASK_HIP_TRSV_LANES_E", force, 1); else unsetenv("CASK_HIP_TRSV_LANES_E");
    std::vector<char> lanes;
    std::vector<int> hdr;
    int l = 0;
    while (l < n_levels) {                                    // runs of narrow levels (< 256 rows), as the engine cuts them
      if (lp[(size_t)l + 1] - lp[(size_t)l] >= 256) { l++; continue; }
      int l1 = l;
      while (l1 < n_levels && lp[(size_t)l1 + 1] - lp[(size_t)l1] < 256) l1++;
      const size_t c0 = hdr.size() / LN_HDR_INTS;
      const int e = build_lanes_run(l, l1, lp[(size_t)l], lp, peptr, ppos, pval, lanes, hdr);
      CHECK(lanes.size() % LN_CHUNK_BYTES == 0 && hdr.size() % LN_HDR_INTS == 0 && lanes.size() / LN_CHUNK_BYTES == hdr.size() / LN_HDR_INTS);
      if (e) {
        CHECK(e == 4 || e == 8 || e == 16);
        const size_t c1 = hdr.size() / LN_HDR_INTS;
        g_lane_runs++;
        g_lane_chunks += (long)(c1 - c0);
        const int C = ln_slabs_per_chunk(e);
        long placed = 0, wanted = (long)peptr[(size_t)lp[(size_t)l1]] - peptr[(size_t)lp[(size_t)l]];
        int next_pos = lp[(size_t)l];
        for (size_t k = c0; k < c1; k++) {
          const int *h = hdr.data() + k * LN_HDR_INTS;
          CHECK(h[8] == next_pos && h[9] >= 0 && h[9] <= LN_ROWS && h[12] == e);
          next_pos = h[8] + h[9];
          const char *img = lanes.data() + k * (size_t)LN_CHUNK_BYTES;
          for (int s = 0; s < C; s++) {
            const char *rec = img + (size_t)s * 64 * 12 * e;
            const int *tab = reinterpret_cast<const int *>(img + LN_REC_BYTES + (size_t)s * LN_TAB_BYTES);
            for (int lane = 0; lane < 64; lane++) {
              const int w = tab[lane], dst = w & 0x1ffff, row = (w >> 17) & 0x3ff, lg = (w >> 27) & 7;
              CHECK(lg <= 6 && row <= LN_ROWS + 63 && (dst == LN_DUMP || (dst % 8 == 0 && dst < 8 * LN_RING)));
              for (int t = 0; t < e; t++) {
                double v;
                int adr;
                std::memcpy(&v, rec + 1024 * (t >> 1) + 16 * lane + 8 * (t & 1), 8);
                std::memcpy(&adr, rec + 1024 * (e / 2 + (t >> 2)) + 16 * lane + 4 * (t & 3), 4);
                CHECK(adr == LN_ZERO || (adr % 8 == 0 && adr >= 0 && adr < 8 * LN_RING));
                if (adr != LN_ZERO) placed++; else CHECK(v == 0.0);
              }
            }
          }
        }
        CHECK(next_pos == lp[(size_t)l1]);
        CHECK(placed == wanted);
      }
      l = l1;
    }
  }
  unsetenv("CASK_HIP_TRSV_LANES_E");
}

int main(int argc, char **argv) {
  if (argc < 2) { std::printf("usage: test_planners <mtx dir> [<csr dump dir>]\n"); return 2; }
  std::vector<Csr> mats;
  for (const char *sub : {"matrices", "benchmark", "systems", "failing"})
    for (const std::string &f : list_dir(std::string(argv[1]) + "/" + sub, ".mtx")) {
      if (f.find("_b.mtx") != std::string::npos || f.find("_sol.mtx") != std::string::npos) continue;   // array files
      Csr a;
      try {
        const cask::CsrMatrix m = cask::io::readMatrix(f);
        a.n = m.n; a.m = m.m; a.rp = m.row_ptr; a.ci = m.col_ind; a.va = m.values;
      } catch (const std::exception &e) { std::printf("skip %s: %s\n", f.c_str(), e.what()); continue; }
      a.name = f.substr(f.find_last_of('/') + 1);
      mats.push_back(std::move(a));
    }
  if (argc > 2)
    for (const std::string &f : list_dir(argv[2], ".csr")) {
      Csr a;
      if (!load_dump(f, a)) { std::printf("FAIL cannot read %s\n", f.c_str()); g_failures++; continue; }
      a.name = f.substr(f.find_last_of('/') + 1);
      mats.push_back(std::move(a));
    }
  for (const Csr &a : mats) {
    g_ctx = a.name;
    merge_plans(a);
    scan_plans(a);
    slice_plans(a);
    vector_pieces(a);
    lanes_runs(a);
  }
  std::printf("%zu matrices, %ld lane-group runs (%ld chunks), %ld checks, %ld failures\n", mats.size(), g_lane_runs, g_lane_chunks,
              g_checks, g_failures);
  return g_failures ? 1 : (mats.size() >= 40 ? 0 : 3);
}
