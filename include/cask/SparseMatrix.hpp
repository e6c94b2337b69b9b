// Client-visible containers of the CASK surface: Vector, DokMatrix, CsrMatrix,
// SymCsrMatrix.  Same names, members and semantics as the reference's
// src/runtime/SparseMatrix.hpp (Vector :37-99, DokMatrix :117-267, CsrMatrix
// :272-485, SymCsrMatrix :487-517) so its clients compile unchanged; written
// from scratch, without the Eigen dependency (the reference only takes a
// typedef from <Eigen/Sparse>, :21).
//
// These are host containers.  The dot() members are the reference's host-side
// helpers for small matrices; the SpMV engine (cask::spmv::Spmv) never calls
// them -- it runs on the GPU or fails.
#ifndef CASK_SPARSEMATRIX_HPP
#define CASK_SPARSEMATRIX_HPP

#include <algorithm>
#include <cassert>
#include <cmath>
#include <fstream>
#include <initializer_list>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace cask {

namespace sparse {
// Coordinate list as produced by io::MmReader (reference: SparseMatrix.hpp:24-34).
template <typename value_type>
class SparkCooMatrix {
 public:
  using CoordType = std::tuple<int, int, value_type>;
  int n, m;
  std::vector<CoordType> data;
  SparkCooMatrix(int rows, int cols) : n(rows), m(cols) {}
};
}  // namespace sparse

class Vector {
 public:
  std::vector<double> data;

  Vector(int n) : data(static_cast<size_t>(n), 0.0) {}
  Vector(std::initializer_list<double> l) : data(l) {}
  Vector(const std::vector<double> &v) : data(v) {}

  int size() const { return static_cast<int>(data.size()); }
  const double &operator[](int i) const { return data[i]; }
  double &operator[](int i) { return data[i]; }
  bool operator==(const Vector &o) const { return data == o.data; }

  Vector operator-(const Vector &o) const {
    if (o.data.size() != data.size())
      throw std::invalid_argument("Attempt to subtract vectors of different lengths: " +
                                  std::to_string(o.size()) + " != " + std::to_string(size()));
    Vector r(size());
    std::transform(data.begin(), data.end(), o.data.begin(), r.data.begin(),
                   [](double a, double b) { return a - b; });
    return r;
  }

  void print(std::string label = "") const {
    std::cout << label;
    for (double d : data) std::cout << d << " ";
    std::cout << std::endl;
  }

  // Euclidean norm.  (The reference forgets to zero its accumulator, :81-86; the
  // intended value -- what its own test expects, test/SparseMatrix.cpp:200-203 -- is this.)
  double norm() const {
    double s = 0.0;
    for (double d : data) s += d * d;
    return std::sqrt(s);
  }
  double distance(const Vector &other) const { return (*this - other).norm(); }

  void writeToFile(std::string path) {
    std::ofstream f{path};
    if (!f) throw std::invalid_argument("Could not open file for writing");
    for (double d : data) f << d << std::endl;
  }
};

// Dictionary of keys: row -> (ordered column -> value).
class DokMatrix {
 public:
  int n, m;
  int nnzs;
  std::unordered_map<int, std::map<int, double>> dok;

  DokMatrix() : n(0), m(0), nnzs(0) {}
  DokMatrix(int rows, int cols) : n(rows), m(cols), nnzs(0) {}
  DokMatrix(int rows, int cols, int nonzeros) : n(rows), m(cols), nnzs(nonzeros) {}

  // dense row-major pattern, square (n = floor(sqrt(count))); zeros are not stored
  DokMatrix(const std::initializer_list<double> &pattern)
      : DokMatrix(static_cast<int>(std::floor(std::sqrt(static_cast<double>(pattern.size())))), pattern) {}

  // dense row-major pattern with `rows` rows
  DokMatrix(int rows, const std::initializer_list<double> &pattern) : n(rows), m(0), nnzs(0) {
    m = rows ? static_cast<int>(pattern.size()) / rows : 0;
    int k = 0;
    for (double v : pattern) {
      if (v != 0) {
        dok[k / m][k % m] = v;
        nnzs++;
      }
      k++;
    }
  }

  double at(int i, int j) const {
    assert(i < n && j < m);
    auto r = dok.find(i);
    if (r == dok.end()) return 0;
    auto c = r->second.find(j);
    return c == r->second.end() ? 0 : c->second;
  }

  void set(int i, int j, double val) {     // last write wins; nnzs counts calls like the reference (:213-217)
    assert(i < n && j < m);
    dok[i][j] = val;
    nnzs++;
  }

  bool isNnz(int i, int j) const { return at(i, j) != 0; }

  // Mirror every off-diagonal entry.  A stored (j,i) that disagrees with (i,j) is an error.
  DokMatrix explicitSymmetric() {
    DokMatrix out(n, m);
    int count = 0;
    for (const auto &row : dok)
      for (const auto &e : row.second) {
        const int i = row.first, j = e.first;
        out.dok[i][j] = e.second;
        count++;
        if (i == j) continue;
        auto tr = dok.find(j);
        if (tr != dok.end()) {
          auto te = tr->second.find(i);
          if (te != tr->second.end()) {
            if (te->second != e.second) throw std::invalid_argument("Matrix is not symmetric");
            std::cout << "Warning! Matrix already contains transpose entry for " << i << " " << j << std::endl;
          }
        }
        out.dok[j][i] = e.second;
        count++;
      }
    out.nnzs = count;
    return out;
  }

  bool operator==(const DokMatrix &o) const { return n == o.n && m == o.m && nnzs == o.nnzs && dok == o.dok; }

  void pretty_print() const {
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < m; j++) std::cout << at(i, j) << " ";
      std::cout << "\n";
    }
  }

  DokMatrix getLowerTriangular() const { return triangle(true); }
  DokMatrix getUpperTriangular() const { return triangle(false); }

  // y = A b, per row in ascending column order
  Vector dot(const Vector &b) const {
    Vector y(b.size());
    for (const auto &row : dok)
      for (const auto &e : row.second) y[row.first] += b[e.first] * e.second;
    return y;
  }

 private:
  DokMatrix triangle(bool lower) const {
    DokMatrix t(n, m);
    for (const auto &row : dok)
      for (const auto &e : row.second)
        if (lower ? e.first <= row.first : row.first <= e.first) t.set(row.first, e.first, e.second);
    return t;
  }
};

// Compressed sparse rows, 0-based, columns ascending inside a row.
class CsrMatrix {
 public:
  int n, m;
  int nnzs;
  std::vector<double> values;
  std::vector<int> col_ind;
  std::vector<int> row_ptr;

  CsrMatrix() : n(0), m(0), nnzs(0) {}
  CsrMatrix(std::initializer_list<double> dense) : CsrMatrix(DokMatrix(dense)) {}
  CsrMatrix(int rows, std::initializer_list<double> dense) : CsrMatrix(DokMatrix(rows, dense)) {}

  CsrMatrix(const DokMatrix &d) : n(d.n), m(d.m), nnzs(d.nnzs) {
    row_ptr.reserve(static_cast<size_t>(n) + 1);
    for (int i = 0; i < n; i++) {
      row_ptr.push_back(static_cast<int>(col_ind.size()));
      auto r = d.dok.find(i);
      if (r == d.dok.end()) continue;
      for (const auto &e : r->second) {
        col_ind.push_back(e.first);
        values.push_back(e.second);
      }
    }
    row_ptr.push_back(nnzs);
  }

  CsrMatrix(int rows, int cols, int nonzeros, double *vals, int *cols_idx, int *rows_ptr)
      : n(rows), m(cols), nnzs(nonzeros), values(vals, vals + nonzeros), col_ind(cols_idx, cols_idx + nonzeros),
        row_ptr(rows_ptr, rows_ptr + rows + 1) {}

  CsrMatrix(int rows, int cols, int nonzeros, const std::vector<double> &vals, const std::vector<int> &cols_idx,
            const std::vector<int> &rows_ptr)
      : n(rows), m(cols), nnzs(nonzeros), values(vals), col_ind(cols_idx), row_ptr(rows_ptr) {}

  double &get(int i, int j) {
    for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++)
      if (col_ind[k] == j) return values[k];
    throw std::invalid_argument("No nonzero at row col:" + std::to_string(i) + " " + std::to_string(j));
  }

  bool isNnz(int i, int j) {
    for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++)
      if (col_ind[k] == j) return true;
    return false;
  }

  bool isSymmetric() const { return true; }     // as in the reference (:364-366): a stated precondition, not a check

  DokMatrix toDok() const {
    DokMatrix d(n, m, nnzs);
    for (int i = 0; i < n; i++)
      for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++) d.dok[i][col_ind[k]] = values[k];
    return d;
  }

  bool operator==(const CsrMatrix &o) const {
    return n == o.n && m == o.m && nnzs == o.nnzs && values == o.values && row_ptr == o.row_ptr &&
           col_ind == o.col_ind;
  }

  std::vector<int> getRowPtrWithOneBasedIndex() const { return shifted(row_ptr); }
  std::vector<int> getColIndWithOneBasedIndex() const { return shifted(col_ind); }

  CsrMatrix getLowerTriangular() const { return CsrMatrix(toDok().getLowerTriangular()); }
  CsrMatrix getUpperTriangular() const { return CsrMatrix(toDok().getUpperTriangular()); }

  // host-side product for small matrices (the engine is cask::spmv::Spmv)
  Vector dot(const Vector &b) const {
    Vector y(b.size());
    for (int i = 0; i < n; i++)
      for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++) y[i] += b[col_ind[k]] * values[k];
    return y;
  }

  CsrMatrix sliceRows(int startRow, int nRows) const {
    const int k0 = row_ptr[startRow], k1 = row_ptr[startRow + nRows];
    std::vector<int> rp(static_cast<size_t>(nRows) + 1);
    for (int i = 0; i <= nRows; i++) rp[i] = row_ptr[startRow + i] - k0;
    return CsrMatrix(nRows, m, k1 - k0, std::vector<double>(values.begin() + k0, values.begin() + k1),
                     std::vector<int>(col_ind.begin() + k0, col_ind.begin() + k1), rp);
  }

  // Vertical stripes of `blockSize` columns with block-local column indices.  As in the reference
  // (:459-482) each stripe's row_ptr holds n cumulative row ENDS (no leading 0) and n/m/nnzs are
  // left at their defaults -- the DFE stream format depends on exactly that.
  std::vector<CsrMatrix> sliceColumns(int blockSize) const {
    const int nBlocks = m / blockSize + (m % blockSize == 0 ? 0 : 1);
    std::vector<CsrMatrix> stripes(static_cast<size_t>(nBlocks));
    for (auto &s : stripes) s.row_ptr.reserve(static_cast<size_t>(n));
    for (int i = 0; i < n; i++) {
      for (auto &s : stripes) s.row_ptr.push_back(static_cast<int>(s.col_ind.size()));
      for (int k = row_ptr[i]; k < row_ptr[i + 1]; k++) {
        CsrMatrix &s = stripes[col_ind[k] / blockSize];
        s.col_ind.push_back(col_ind[k] % blockSize);
        s.values.push_back(values[k]);
        s.row_ptr.back()++;
      }
    }
    return stripes;
  }

  void pretty_print() const {
    for (int i = 0; i < n; i++) {
      int k = row_ptr[i];
      for (int j = 0; j < n; j++) {
        if (k < row_ptr[i + 1] && col_ind[k] == j) std::cout << values[k++] << " ";
        else std::cout << "0 ";
      }
      std::cout << "\n";
    }
  }

  void print() {
    std::cout << "CSRMatrix( n= " << n << " nnzs= " << nnzs << ")" << std::endl;
    auto dump = [](const char *label, const auto &v) {
      std::cout << label;
      for (const auto &e : v) std::cout << e << " ";
      std::cout << std::endl;
    };
    dump("values = ", values);
    dump("col_ind = ", col_ind);
    dump("row_ptr = ", row_ptr);
  }

 private:
  static std::vector<int> shifted(const std::vector<int> &v) {
    std::vector<int> r(v);
    for (int &e : r) e += 1;
    return r;
  }
};

// Symmetric matrix stored as its lower triangle.
class SymCsrMatrix {
 public:
  int n, m;
  int nnzs;              // nonzeros of the FULL matrix
  CsrMatrix matrix;      // the stored triangle

  explicit SymCsrMatrix(const DokMatrix &lower) : n(lower.n), m(lower.m), nnzs(0), matrix(lower) {
    int diag = 0;
    for (int i = 0; i < lower.n; i++) diag += lower.at(i, i) != 0;
    nnzs = 2 * (lower.nnzs - diag) + diag;
  }

  void print() {}

  void pretty_print() {
    std::cout << "Stored matrix: " << std::endl;
    matrix.pretty_print();
    std::cout << "Implicit values: " << std::endl;
    matrix.toDok().explicitSymmetric().pretty_print();
  }

  Vector dot(const Vector &b) const { return matrix.toDok().explicitSymmetric().dot(b); }
};

}  // namespace cask

#endif  // CASK_SPARSEMATRIX_HPP
