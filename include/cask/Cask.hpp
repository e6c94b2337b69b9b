// CaskContext -- the thin client facade of include/Cask.hpp in the reference
// (:10-39), with its two defects repaired: getSpmv dereferences the
// implementation pointer (the reference passes the pointer to a by-value
// constructor and does not compile) and getCg returns a solver.
#ifndef CASK_HPP
#define CASK_HPP

#include <stdexcept>

#include "Cg.hpp"
#include "GeneratedImplSupport.hpp"
#include "SparseMatrix.hpp"
#include "Spmv.hpp"

namespace cask {

class CaskContext {
  cask::runtime::SpmvImplementationLoader spmvManager;

  cask::spmv::Spmv forRows(int rows) {
    auto *impl = spmvManager.architectureWithParams(rows);
    if (!impl) throw std::runtime_error("No generated SpMV implementation supports " + std::to_string(rows) + " rows");
    return spmv::Spmv(*impl);
  }

 public:
  void preprocess(const SymCsrMatrix &) {}
  cask::spmv::Spmv getSpmv(SymCsrMatrix &matrix) { return forRows(matrix.n); }
  cask::spmv::Spmv getSpmv(CsrMatrix &matrix) { return forRows(matrix.n); }
  cask::solvers::Cg getCg(SymCsrMatrix &) { return cask::solvers::Cg(); }
};

}  // namespace cask

#endif  // CASK_HPP
