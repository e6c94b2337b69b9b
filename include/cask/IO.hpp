// MatrixMarket input of the CASK surface: same functions, validation and error
// messages as the reference's src/runtime/IO.hpp (readHeader :60-71, readVector
// :73-116, readDokMatrix :124-148, readMatrix :151-163, readSymMatrix :165-176,
// MmReader :178-331), written from scratch.
//
// readMatrix() does not go through the hash-of-maps DokMatrix (minutes for the
// million-row inputs of BASELINE.json): it parses the file in one pass and
// builds the CSR arrays with a counting sort, keeping the reference's
// semantics -- 1-based file indices, last duplicate wins (DokMatrix::set),
// symmetric files mirrored explicitly and rejected when (i,j) != (j,i).
#ifndef CASK_IO_HPP
#define CASK_IO_HPP

#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <numeric>
#include <regex>
#include <sstream>
#include <string>
#include <vector>

#include "SparseMatrix.hpp"

namespace cask {
namespace io {

struct MmInfo {
  const std::string type, format, dataType, symmetry;
  MmInfo(std::string t, std::string f, std::string d, std::string s) : type(t), format(f), dataType(d), symmetry(s) {}
  bool isMatrix() const { return type == "matrix"; }
  bool isSymmetric() const { return symmetry == "symmetric"; }
  bool isCoordinate() const { return format == "coordinate"; }
};

inline MmInfo readHeader(std::string path) {
  std::ifstream f{path};
  if (!f) throw std::invalid_argument("File not found " + path);
  std::string first;
  std::getline(f, first);
  static const std::regex header(
      "%%MatrixMarket (matrix|array) (coordinate|array) (real|integer) (symmetric|general)");
  std::smatch m;
  if (!std::regex_match(first, m, header)) throw std::invalid_argument("Not a valid MatrixMarket file in " + path);
  return MmInfo{m[1], m[2], m[3], m[4]};
}

namespace detail {
// first line that is not a '%' comment (the size line)
inline std::string sizeLine(std::ifstream &f) {
  std::string line;
  while (std::getline(f, line))
    if (line.empty() || line[0] != '%') break;
  return line;
}

struct Coo {
  int n = 0, m = 0;
  std::vector<int> row, col;
  std::vector<double> val;
};

inline Coo readCoordinates(const std::string &path) {
  std::ifstream f{path};
  if (!f) throw std::invalid_argument("File not found " + path);
  Coo c;
  long entries = 0;
  std::stringstream ss(sizeLine(f));
  ss >> c.n >> c.m >> entries;
  c.row.reserve(entries);
  c.col.reserve(entries);
  c.val.reserve(entries);
  // one read of the remaining bytes, then strtol/strtod over the buffer
  std::string rest((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const char *p = rest.c_str();
  char *end = nullptr;
  for (long k = 0; k < entries; k++) {
    const long i = std::strtol(p, &end, 10);
    if (end == p) throw std::invalid_argument("File has less than given nonzeros!");
    p = end;
    const long j = std::strtol(p, &end, 10);
    p = end;
    const double v = std::strtod(p, &end);
    if (end == p) throw std::invalid_argument("File has less than given nonzeros!");
    p = end;
    c.row.push_back(static_cast<int>(i - 1));      // MatrixMarket is 1-based
    c.col.push_back(static_cast<int>(j - 1));
    c.val.push_back(v);
  }
  return c;
}

// COO -> CSR: stable counting sort by row, then by column inside each row; of equal (row, col)
// entries the LAST one in file order survives.
inline CsrMatrix cooToCsr(const Coo &c) {
  const size_t nz = c.val.size();
  for (size_t k = 0; k < nz; k++)
    if (c.row[k] < 0 || c.row[k] >= c.n || c.col[k] < 0 || c.col[k] >= c.m)
      throw std::invalid_argument("MatrixMarket entry outside the declared shape");
  std::vector<int> start(static_cast<size_t>(c.n) + 1, 0);
  for (size_t k = 0; k < nz; k++) start[c.row[k] + 1]++;
  std::partial_sum(start.begin(), start.end(), start.begin());
  std::vector<size_t> order(nz);
  {
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (size_t k = 0; k < nz; k++) order[fill[c.row[k]]++] = k;
  }
  CsrMatrix out;
  out.n = c.n;
  out.m = c.m;
  out.row_ptr.assign(1, 0);
  out.values.reserve(nz);
  out.col_ind.reserve(nz);
  for (int r = 0; r < c.n; r++) {
    auto b = order.begin() + start[r], e = order.begin() + start[r + 1];
    std::stable_sort(b, e, [&](size_t x, size_t y) { return c.col[x] < c.col[y]; });
    for (auto it = b; it != e; ++it) {
      const bool last_of_key = (it + 1 == e) || c.col[*(it + 1)] != c.col[*it];
      if (!last_of_key) continue;
      out.col_ind.push_back(c.col[*it]);
      out.values.push_back(c.val[*it]);
    }
    out.row_ptr.push_back(static_cast<int>(out.col_ind.size()));
  }
  out.nnzs = static_cast<int>(out.col_ind.size());
  return out;
}

inline double lookup(const CsrMatrix &a, int i, int j, bool &found) {
  auto b = a.col_ind.begin() + a.row_ptr[i], e = a.col_ind.begin() + a.row_ptr[i + 1];
  auto it = std::lower_bound(b, e, j);
  found = it != e && *it == j;
  return found ? a.values[it - a.col_ind.begin()] : 0.0;
}
}  // namespace detail

inline cask::Vector readVector(std::string path) {
  MmInfo info = readHeader(path);
  std::ifstream f{path};
  std::stringstream ss(detail::sizeLine(f));
  int n = 0, m = 0;
  ss >> n >> m;
  Vector v(n);
  if (info.format == "coordinate") {
    int entries = 0;
    ss >> entries;
    for (int k = 0; k < entries; k++) {
      int a, b;
      double val;
      f >> a >> b >> val;
      v[a - 1] = val;                // 1-based (the reference writes v[a], off by one; IO.hpp:100)
    }
    return v;
  }
  for (int i = 0; i < n; i++) f >> v[i];
  return v;
}

// Entries exactly as the file holds them (symmetric files: one triangle).
inline DokMatrix readDokMatrix(std::string path, const MmInfo &info) {
  (void)info;
  detail::Coo c = detail::readCoordinates(path);
  DokMatrix mat(c.n, c.m);
  for (size_t k = 0; k < c.val.size(); k++) mat.set(c.row[k], c.col[k], c.val[k]);
  return mat;
}

// Full CSR; a symmetric file comes back with both triangles stored.
inline cask::CsrMatrix readMatrix(std::string path) {
  MmInfo info = readHeader(path);
  if (!info.isMatrix()) throw std::invalid_argument("Error! Expecting MatrixMarket matrix in " + path);
  if (!info.isCoordinate())      // the reference asserts this (IO.hpp:132)
    throw std::invalid_argument("Error! Expecting coordinate format in " + path);
  detail::Coo c = detail::readCoordinates(path);
  if (!info.isSymmetric()) return detail::cooToCsr(c);
  CsrMatrix half = detail::cooToCsr(c);            // duplicates resolved first
  detail::Coo full;
  full.n = c.n;
  full.m = c.m;
  for (int i = 0; i < half.n; i++)
    for (int k = half.row_ptr[i]; k < half.row_ptr[i + 1]; k++) {
      const int j = half.col_ind[k];
      const double v = half.values[k];
      full.row.push_back(i); full.col.push_back(j); full.val.push_back(v);
      if (i == j) continue;
      bool found = false;
      const double other = (j < half.n) ? detail::lookup(half, j, i, found) : 0.0;
      if (found && other != v) throw std::invalid_argument("Matrix is not symmetric");
      full.row.push_back(j); full.col.push_back(i); full.val.push_back(v);
    }
  return detail::cooToCsr(full);
}

// Binary cache of a parsed matrix (the text of a 10^6-row SuiteSparse matrix takes seconds to
// parse, the binary milliseconds): header "CASKCSR2", int64 n, m, nnz, int64 size and mtime (ns) of the text file
// the cache was made from (0, 0 = stand-alone file), then row_ptr (int32), col_ind (int32) and values (fp64)
// little-endian as they sit in CsrMatrix.  No reference counterpart.
struct SourceStamp {
  long long size = 0, mtime_ns = 0;
  bool operator==(const SourceStamp &o) const { return size == o.size && mtime_ns == o.mtime_ns; }
};
inline bool statSource(const std::string &path, SourceStamp &out) {
  struct stat st;
  if (::stat(path.c_str(), &st) != 0) return false;
  out.size = static_cast<long long>(st.st_size);
  out.mtime_ns = static_cast<long long>(st.st_mtim.tv_sec) * 1000000000LL + st.st_mtim.tv_nsec;
  return true;
}

inline void writeCsrBinary(const std::string &path, const cask::CsrMatrix &a, const SourceStamp &src = SourceStamp()) {
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::invalid_argument("Cannot write " + path);
  const char magic[8] = {'C', 'A', 'S', 'K', 'C', 'S', 'R', '2'};
  const long long dims[5] = {a.n, a.m, static_cast<long long>(a.values.size()), src.size, src.mtime_ns};
  f.write(magic, 8);
  f.write(reinterpret_cast<const char *>(dims), sizeof(dims));
  f.write(reinterpret_cast<const char *>(a.row_ptr.data()), static_cast<std::streamsize>(a.row_ptr.size() * sizeof(int)));
  f.write(reinterpret_cast<const char *>(a.col_ind.data()), static_cast<std::streamsize>(a.col_ind.size() * sizeof(int)));
  f.write(reinterpret_cast<const char *>(a.values.data()), static_cast<std::streamsize>(a.values.size() * sizeof(double)));
  if (!f) throw std::invalid_argument("Error writing " + path);
}

// `src_out` (optional) receives the stamp of the text file the cache was made from.  The arrays are validated
// (monotone row_ptr, columns inside [0, m)): a cache file may have been written by anybody.
inline cask::CsrMatrix readCsrBinary(const std::string &path, SourceStamp *src_out = nullptr) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::invalid_argument("File not found: " + path);
  char magic[8];
  long long dims[5];
  f.read(magic, 8);
  f.read(reinterpret_cast<char *>(dims), sizeof(dims));
  if (!f || std::string(magic, 8) != "CASKCSR2" || dims[0] < 0 || dims[1] < 0 || dims[2] < 0 || dims[0] > 2147483646LL ||
      dims[1] > 2147483647LL || dims[2] > 2147483647LL)
    throw std::invalid_argument("Not a CASK binary CSR file: " + path);
  if (src_out) {
    src_out->size = dims[3];
    src_out->mtime_ns = dims[4];
  }
  cask::CsrMatrix a;
  a.n = static_cast<int>(dims[0]);
  a.m = static_cast<int>(dims[1]);
  a.nnzs = static_cast<int>(dims[2]);
  a.row_ptr.resize(static_cast<size_t>(a.n) + 1);
  a.col_ind.resize(static_cast<size_t>(a.nnzs));
  a.values.resize(static_cast<size_t>(a.nnzs));
  f.read(reinterpret_cast<char *>(a.row_ptr.data()), static_cast<std::streamsize>(a.row_ptr.size() * sizeof(int)));
  f.read(reinterpret_cast<char *>(a.col_ind.data()), static_cast<std::streamsize>(a.col_ind.size() * sizeof(int)));
  f.read(reinterpret_cast<char *>(a.values.data()), static_cast<std::streamsize>(a.values.size() * sizeof(double)));
  bool ok = static_cast<bool>(f) && a.row_ptr.front() == 0 && a.row_ptr.back() == a.nnzs;
  for (int r = 0; ok && r < a.n; r++) ok = a.row_ptr[r + 1] >= a.row_ptr[r];
  for (int k = 0; ok && k < a.nnzs; k++) ok = a.col_ind[k] >= 0 && a.col_ind[k] < a.m;
  if (!ok) throw std::invalid_argument("Truncated or inconsistent binary CSR file: " + path);
  return a;
}

// readMatrix with the cache beside the text file (<path>.csrbin): written on the first read, used afterwards as
// long as the text file still has the size and modification time recorded in the cache.
inline cask::CsrMatrix readMatrixCached(const std::string &path) {
  const std::string cache = path + ".csrbin";
  SourceStamp now;
  const bool have_stamp = statSource(path, now);
  {
    std::ifstream probe(cache, std::ios::binary);
    if (probe && have_stamp) {
      try {
        SourceStamp made_from;
        cask::CsrMatrix a = readCsrBinary(cache, &made_from);
        if (made_from == now) return a;
        // the text changed since the cache was written: re-parse below and overwrite the cache
      } catch (const std::invalid_argument &) {
        // stale or foreign file: fall through to the text
      }
    }
  }
  cask::CsrMatrix a = readMatrix(path);
  try {
    if (have_stamp) writeCsrBinary(cache, a, now);
  } catch (const std::invalid_argument &) {
    // read-only directory: the cache is optional
  }
  return a;
}

// Stored triangle only; general files are rejected.
inline cask::SymCsrMatrix readSymMatrix(std::string path) {
  MmInfo info = readHeader(path);
  if (!info.isMatrix()) throw std::invalid_argument("Error! Expecting MatrixMarket matrix in " + path);
  if (!info.isSymmetric())
    throw std::invalid_argument("Error! Matrix found in " + path +
                                " is not symmetric. To read unsymmetric matrix use cask::io::readSymMatrix()");
  return cask::SymCsrMatrix(readDokMatrix(path, info));
}

// Coordinate-list reader used by the reference's integration client (test/test_spmv.cpp:21-22).
template <typename value_type>
class MmReader {
  bool sparse = false, symmetric = false, matrix = false;
  int ncols = 0, nrows = 0, nnzs = -1;
  std::ifstream *f;
  std::string path;

  using CooMatrix = cask::sparse::SparkCooMatrix<value_type>;
  using CoordType = typename CooMatrix::CoordType;

  void parseHeaderLine(const std::string &line) {
    if (line.find("coordinate") != std::string::npos) sparse = true;
    else if (line.find("array") != std::string::npos) sparse = false;
    else throw std::invalid_argument("Cannot parse header, requires either 'coordinate' or 'matrix' type");
    if (line.size() > 1 && line[0] == '%' && line[1] == '%') {
      if (line.find("matrix") == std::string::npos) throw std::invalid_argument("Unsupported file type: " + line);
      symmetric = line.find("symmetric") != std::string::npos;
    }
  }

  void parseHeader() {
    std::string line;
    if (!std::getline(*f, line)) throw std::invalid_argument("File " + path + " is empty");
    parseHeaderLine(line);
    while (std::getline(*f, line) && !line.empty() && line[0] == '%') {}
    std::stringstream ss(line);
    ss >> nrows >> ncols;
    if (sparse) ss >> nnzs;
    matrix = ncols > 1;
  }

 public:
  MmReader(std::string p) : f(new std::ifstream{p}), path(p) {
    if (!f->is_open()) {
      delete f;
      throw std::invalid_argument("Could not open file path " + path);
    }
  }
  MmReader(const MmReader &) = delete;
  MmReader &operator=(const MmReader &) = delete;
  virtual ~MmReader() { delete f; }

  std::vector<double> readVector() {
    parseHeader();
    if (matrix) throw std::invalid_argument("Object has > 1 columns ==> Use readMatrix");
    if (sparse) throw std::invalid_argument("Sparse vectors not supported");
    std::cout << "Reading vector" << std::endl;
    std::vector<double> out;
    double v;
    while (*f >> v) out.push_back(v);
    return out;
  }

  // Sorted (row, col) triplets, 0-based, symmetric entries mirrored, duplicates kept.
  CooMatrix mmreadMatrix(std::string) {
    parseHeader();
    if (!matrix) throw std::invalid_argument("Matrix has only one column ==> Use readVector");
    std::vector<CoordType> triplets;
    std::string line;
    for (int k = 0; k < nnzs; k++) {
      if (!std::getline(*f, line)) throw std::invalid_argument("File has less than given nonzeros!");
      std::stringstream ss{line};
      int i, j;
      value_type v;
      ss >> i >> j >> v;
      triplets.emplace_back(i - 1, j - 1, v);
      if (symmetric && i != j) triplets.emplace_back(j - 1, i - 1, v);
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    std::stable_sort(triplets.begin(), triplets.end(), [](const CoordType &a, const CoordType &b) {
      return std::get<0>(a) != std::get<0>(b) ? std::get<0>(a) < std::get<0>(b) : std::get<1>(a) < std::get<1>(b);
    });
    std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
    std::cout << "Sorting took: " << dt.count() << std::endl;
    CooMatrix out(nrows, ncols);
    out.data = std::move(triplets);
    return out;
  }
};

}  // namespace io
}  // namespace cask

#endif  // CASK_IO_HPP
