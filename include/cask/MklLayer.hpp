// cask::mkl::unittrsolve -- the triangular solves the reference takes from MKL's mkl_dcsrtrsv
// (src/runtime/MklLayer.hpp:29-85: uplo = lower/upper, no transpose, diag = 'N': the diagonal stored in
// the matrix is used, despite the function's name).  Same names and argument meaning; the solve runs on
// the GPU, level-scheduled (cask_hip_trsolve).  The second overload takes the 1-BASED row_ptr / col_ind
// arrays the reference prepares for MKL (CsrMatrix::getRowPtrWithOneBasedIndex).
#ifndef CASK_MKLLAYER_HPP
#define CASK_MKLLAYER_HPP

#include <vector>

#include "SparseMatrix.hpp"

namespace cask {
namespace mkl {

std::vector<double> unittrsolve(const CsrMatrix &m, const std::vector<double> &rhs, bool lowerTriangular);

void unittrsolve(const double *values, const int *row_ptr, const int *col_ind, const std::vector<double> &rhs,
                 double *res, bool lowerTriangular);

}  // namespace mkl
}  // namespace cask

#endif  // CASK_MKLLAYER_HPP
