// Layout of the lane-group walk's stream (k_trsv_lanes, trsv_lanes.hpp) and the host planner that writes it.  No HIP:
// the planner writes raw byte images with memcpy -- lane words, LDS byte addresses, 16-byte units -- and is compiled and
// run under AddressSanitizer / UBSan on the CPU (`make asan`, tests/cpp/test_planners.cpp).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace caskhip_lanes {

constexpr int LN_NG = 3, LN_ST = 192, LN_T = 64 + LN_NG * LN_ST;   // wave 0 walks; LN_NG groups of three waves stage, taking turns
constexpr int LN_GRID = 9;                                // workgroup 0 solves, workgroup 8 (same XCD) reads ahead
constexpr int LN_RING = 8192;                             // x values of the most recent positions (LDS)
constexpr int LN_REC_BYTES = 24576;                       // records of a chunk: 2 048 entry slots of 12 bytes
constexpr int LN_CMAX = 8;                                // slabs per chunk at E = 4 (32 / E in general)
constexpr int LN_TAB_BYTES = 256;                         // per slab a word per lane: where its result goes | its row << 20
constexpr int LN_CHUNK_BYTES = LN_REC_BYTES + LN_CMAX * LN_TAB_BYTES;   // a chunk in memory: records, then the slabs' lane words
constexpr int LN_ROWS = 64 * LN_CMAX;                     // positions a chunk can hold
constexpr int LN_UNITS = LN_CHUNK_BYTES / 16;             // 16-byte units per chunk (at E > 4 the last ones are unused)
constexpr int LN_UJ = (LN_UNITS + LN_ST - 1) / LN_ST;     // ... per stager thread
constexpr int LN_RJ = (LN_ROWS + LN_ST - 1) / LN_ST;      // right-hand sides per stager thread
constexpr int LN_HDR_INTS = 16;   // per chunk: [8] first position, [9] positions, [10] [11] the same of the chunk LN_NG behind, [12] E
constexpr int LN_BAD_LDS_BASE = -0x4C4453;                   // progress word of a launch whose dynamic LDS does not start at 0
constexpr int LN_ZERO = 8 * LN_RING;                      // LDS byte address of a constant 0.0 ...
constexpr int LN_DUMP = LN_ZERO + 8;                      // ... and of a slot nobody reads
constexpr int LN_BUF0 = LN_ZERO + 16;
constexpr int LN_OFF_B = LN_CHUNK_BYTES, LN_OFF_D = LN_OFF_B + 8 * (LN_ROWS + 64), LN_BUF_BYTES = LN_OFF_D + 8 * (LN_ROWS + 64);
constexpr size_t LN_LDS_BYTES = LN_BUF0 + 2 * (size_t)LN_BUF_BYTES;
static_assert(LN_BUF0 % 16 == 0 && LN_BUF_BYTES % 16 == 0 && LN_REC_BYTES % 16 == 0, "16-byte LDS accesses");
static_assert(LN_DUMP < (1 << 17) && LN_ROWS + 64 < (1 << 10), "lane word: 17 bits of LDS address, 10 of row, 3 of log2 G");
static_assert(LN_LDS_BYTES <= 160 * 1024, "one workgroup's LDS on gfx950");
static_assert(LN_NG == 3 && (LN_RING & (LN_RING - 1)) == 0 && LN_RING >= (LN_NG + 2) * LN_ROWS, "ring slots by position mod LN_RING; the write-back lags LN_NG chunks");

// lane word (one per slab and lane): LDS byte address of its result -- the row's ring slot in the last lane of a group,
// the dump slot elsewhere -- | row (relative to the chunk's first position) << 17 | log2(lanes per row) << 27
constexpr int ln_lane_word(int dst, int row, int lg) { return dst | (row << 17) | (lg << 27); }
constexpr int ln_slabs_per_chunk(int e) { return 32 / e; }

// ---- host: the slabs of one run of narrow levels [l0, l1) = positions [lo, ..) -----------------------------------------
// peptr / ppos / pval: the factor's off-diagonal entries in position space (row = position, source = producer position).
// Appends whole chunks to `lanes` / `hdr` and returns E (4, 8, 16); 0 (nothing appended) when the run does not qualify.
inline int build_lanes_run(int l0, int l1, int lo, const std::vector<int> &lp, const std::vector<int> &peptr,
                           const std::vector<int> &ppos, const std::vector<double> &pval, std::vector<char> &lanes,
                           std::vector<int> &hdr) {
  for (int l = l0; l < l1; l++)
    for (int i = lp[l]; i < lp[l + 1]; i++) {
      if (peptr[i + 1] - peptr[i] > 64 * 16) return 0;
      for (int e = peptr[i]; e < peptr[i + 1]; e++)
        if (ppos[e] < lo || ppos[e] < lp[l + 1] - LN_RING) return 0;          // a source outside the run or the ring
    }
  // A slab: consecutive rows of one level whose lane groups -- 2^lg lanes for a row of up to E 2^lg entries, a size per
  // ROW -- add up to at most 64 lanes (a common size per slab needed 23 % more slabs on the cant-like factors: one row of
  // 40 entries made every group of its slab 8 lanes wide).
  struct Slab { int p0, nrows; };
  auto lg_of = [](int ne, int E) {
    const int per = (ne + E - 1) / E;
    int lg = 0;
    while ((1 << lg) < per) lg++;
    return lg;
  };
  auto cut = [&](int E, std::vector<Slab> *out) {
    size_t count = 0;
    for (int l = l0; l < l1; l++)
      for (int i = lp[l]; i < lp[l + 1];) {
        int used = 0, cnt = 0;
        while (i + cnt < lp[l + 1]) {
          const int lg = lg_of(peptr[i + cnt + 1] - peptr[i + cnt], E);
          if (lg > 6 || used + (1 << lg) > 64) break;
          used += 1 << lg;
          cnt++;
        }
        if (cnt == 0) return (size_t)-1;                      // a row too long for this E
        if (out) out->push_back(Slab{i, cnt});
        count++;
        i += cnt;
      }
    return count;
  };
  // the cheapest E: a slab is a dependent step of ~260 cycles + ~10 per entry of a lane for the walker; a chunk ~1 200
  // cycles of staging
  int best_e = 0;
  double best = 0.0;
  const char *only = std::getenv("CASK_HIP_TRSV_LANES_E");   // tests: one E instead of the cheapest
  for (int E : {4, 8, 16}) {
    if (only && std::atoi(only) != E) continue;
    const size_t n = cut(E, nullptr);
    if (n == (size_t)-1) continue;
    const double walker = (double)n * (260.0 + 10.0 * E), stagers = (double)((n + 32 / E - 1) / (32 / E)) * 1200.0;
    const double cost = std::max(walker, stagers);
    if (!best_e || cost < best) { best_e = E; best = cost; }
  }
  if (!best_e) return 0;
  const int E = best_e, C = ln_slabs_per_chunk(E), lane_bytes = 12 * E;
  std::vector<Slab> slabs;
  cut(E, &slabs);
  const size_t n_chunks = (slabs.size() + C - 1) / C;
  const size_t lanes0 = lanes.size(), hdr0 = hdr.size();
  lanes.resize(lanes0 + n_chunks * (size_t)LN_CHUNK_BYTES);
  hdr.resize(hdr0 + n_chunks * (size_t)LN_HDR_INTS, 0);
  for (size_t k = 0; k < n_chunks; k++) {
    int *h = hdr.data() + hdr0 + k * LN_HDR_INTS;
    const int pk = slabs[k * C].p0;
    int rows = 0;
    for (int s = 0; s < C; s++) {
      const size_t si = k * C + s;
      const Slab sl = si < slabs.size() ? slabs[si] : Slab{pk + rows, 0};      // padding slabs: no rows
      rows = sl.p0 + sl.nrows - pk;
      char *rec = lanes.data() + lanes0 + k * (size_t)LN_CHUNK_BYTES + (size_t)s * 64 * lane_bytes;
      int *tab = reinterpret_cast<int *>(lanes.data() + lanes0 + k * (size_t)LN_CHUNK_BYTES + LN_REC_BYTES + (size_t)s * LN_TAB_BYTES);
      // lanes: the widest groups first (sizes are powers of two, so every group starts on a multiple of its size -- what
      // the DPP steps need -- and lane 0 belongs to the widest group: its word tells the walker how many steps the slab takes)
      int lane_row[64], lane_g[64], lane_lg[64];
      for (int lane = 0; lane < 64; lane++) { lane_row[lane] = -1; lane_g[lane] = 0; lane_lg[lane] = 0; }
      int cursor = 0;
      for (int want = 6; want >= 0; want--)
        for (int row = 0; row < sl.nrows; row++) {
          const int i = sl.p0 + row;
          if (lg_of(peptr[i + 1] - peptr[i], E) != want) continue;
          for (int g = 0; g < (1 << want); g++, cursor++) { lane_row[cursor] = row; lane_g[cursor] = g; lane_lg[cursor] = want; }
        }
      for (int lane = 0; lane < 64; lane++) {
        double v[16];
        int a[16];
        const int row = lane_row[lane], g = lane_g[lane], lg = lane_lg[lane];
        const bool writer = row >= 0 && g == (1 << lg) - 1;
        // (a lane without a row reads the spare right-hand side behind the slab's rows)
        tab[lane] = ln_lane_word(writer ? ((sl.p0 + row) & (LN_RING - 1)) * 8 : LN_DUMP, sl.p0 - pk + (row >= 0 ? row : sl.nrows), lg);
        for (int t = 0; t < E; t++) {
          v[t] = 0.0;
          a[t] = LN_ZERO;
          if (row >= 0) {
            const int i = sl.p0 + row, e = peptr[i] + g + (t << lg);
            if (e < peptr[i + 1]) {
              v[t] = pval[e];
              a[t] = (ppos[e] & (LN_RING - 1)) * 8;
            }
          }
        }
        for (int q = 0; q < E / 2; q++) std::memcpy(rec + 1024 * q + 16 * lane, v + 2 * q, 16);           // unit-major (see ln_walk_chunk)
        for (int q = 0; q < E / 4; q++) std::memcpy(rec + 1024 * (E / 2 + q) + 16 * lane, a + 4 * q, 16);
      }
    }
    h[8] = pk;
    h[9] = rows;
    h[12] = E;
  }
  for (size_t k = 0; k < n_chunks; k++) {                     // the span of the chunk LN_NG behind (the last ones repeat the last)
    int *h = hdr.data() + hdr0 + k * LN_HDR_INTS;
    const int *h2 = hdr.data() + hdr0 + std::min(k + LN_NG, n_chunks - 1) * LN_HDR_INTS;
    h[10] = h2[8];
    h[11] = h2[9];
  }
  return E;
}

}  // namespace caskhip_lanes
