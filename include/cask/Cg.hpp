// cask::solvers::Cg -- conjugate gradients on the GPU behind the interface the
// reference sketches in src/runtime/Cg.hpp (:4-14, an identity stub there):
// preprocess(SymCsrMatrix&) then Vector solve(Vector& rhs).  The recurrence,
// stopping rule (r.r <= tol^2, tol = 1e-5) and iteration cap (2000) are those
// of the reference's CPU solver pcg<double, IdentityPreconditioner>
// (src/runtime/SparseLinearSolvers.hpp:162-239).
#ifndef CASK_CG_HPP
#define CASK_CG_HPP

#include <memory>

#include "SparseMatrix.hpp"
#include "cask_hip.h"

namespace cask {
namespace solvers {

class Cg {
  std::shared_ptr<cask_hip_matrix> device;
  int n = 0;

 public:
  int maxIterations = 2000;
  double tolerance = 1E-5;
  int iterations = 0;          // as pcg reports them: index of the last non-converged pass
  bool converged = false;
  double microsecondsPerIteration = 0;

  void preprocess(cask::SymCsrMatrix &a);      // expands the stored triangle, uploads once
  void preprocess(const cask::CsrMatrix &a);   // a full (already symmetric) matrix
  Vector solve(Vector &rhs);                   // initial guess 0
};

}  // namespace solvers
}  // namespace cask

#endif  // CASK_CG_HPP
