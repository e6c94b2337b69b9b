// cask::converters -- between the surface's containers and the golden operand types of the reference's
// integration client (src/runtime/Converters.hpp:12-42, used by test/test_spmv.cpp:22,45-50).  The reference
// converts a sorted COO matrix to an Eigen row-major matrix to obtain its golden product; Eigen is fetched
// at its build time and is not part of this repository, so those three functions exist when <Eigen/Sparse>
// is on the include path, and an Eigen-free pair does the same job on the surface's own types:
// tripletToCsr (COO -> CsrMatrix, duplicates last-wins like DokMatrix::set) whose CsrMatrix::dot is the
// same sequential row-major product.
#ifndef CASK_CONVERTERS_HPP
#define CASK_CONVERTERS_HPP

#include <algorithm>
#include <memory>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "SparseMatrix.hpp"

#if defined(__has_include)
#if __has_include(<Eigen/Sparse>)
#include <Eigen/Sparse>
#define CASK_CONVERTERS_HAVE_EIGEN 1
#endif
#endif

namespace cask {
namespace converters {

// COO triplets (any order; the last of equal coordinates wins) -> CSR with ascending columns per row
inline CsrMatrix tripletToCsr(const cask::sparse::SparkCooMatrix<double> &mat) {
  std::vector<size_t> order(mat.data.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    if (std::get<0>(mat.data[a]) != std::get<0>(mat.data[b])) return std::get<0>(mat.data[a]) < std::get<0>(mat.data[b]);
    return std::get<1>(mat.data[a]) < std::get<1>(mat.data[b]);
  });
  std::vector<double> values;
  std::vector<int> col_ind, row_ptr(static_cast<size_t>(mat.n) + 1, 0);
  int last_r = -1, last_c = -1;
  for (size_t k : order) {
    const int r = std::get<0>(mat.data[k]), c = std::get<1>(mat.data[k]);
    if (r < 0 || r >= mat.n || c < 0 || c >= mat.m) throw std::invalid_argument("tripletToCsr: coordinate out of range");
    if (r == last_r && c == last_c) {
      values.back() = std::get<2>(mat.data[k]);           // duplicate: the later entry replaces the earlier one
      continue;
    }
    values.push_back(std::get<2>(mat.data[k]));
    col_ind.push_back(c);
    row_ptr[static_cast<size_t>(r) + 1]++;
    last_r = r;
    last_c = c;
  }
  for (int r = 0; r < mat.n; r++) row_ptr[static_cast<size_t>(r) + 1] += row_ptr[r];
  return CsrMatrix(mat.n, mat.m, static_cast<int>(values.size()), values, col_ind, row_ptr);
}

inline Vector stdvectorToVector(const std::vector<double> &v) { return Vector(v); }

#ifdef CASK_CONVERTERS_HAVE_EIGEN
using EigenSparseMatrix = std::unique_ptr<Eigen::SparseMatrix<double, Eigen::RowMajor, int32_t>>;

inline EigenSparseMatrix tripletToEigen(cask::sparse::SparkCooMatrix<double> mat) {
  EigenSparseMatrix out(new Eigen::SparseMatrix<double, Eigen::RowMajor, int32_t>(mat.n, mat.m));
  std::vector<Eigen::Triplet<double>> entries;
  entries.reserve(mat.data.size());
  for (const auto &t : mat.data) entries.emplace_back(std::get<0>(t), std::get<1>(t), std::get<2>(t));
  out->setFromTriplets(entries.begin(), entries.end());
  return out;
}

inline Eigen::VectorXd stdvectorToEigen(std::vector<double> v) {
  return Eigen::Map<const Eigen::VectorXd>(v.data(), static_cast<Eigen::Index>(v.size()));
}

inline std::vector<double> eigenVectorToStdVector(const Eigen::VectorXd &v) {
  return std::vector<double>(v.data(), v.data() + v.size());
}
#endif

}  // namespace converters
}  // namespace cask

#endif  // CASK_CONVERTERS_HPP
