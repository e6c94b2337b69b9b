// Solver classes of the CASK surface (reference: src/runtime/SparseLinearSolvers.hpp:24-61).
// DfeCgSolver::solve and DfeBiCgSolver::solve are declared but never defined in
// the reference; here they run on the GPU.  The reference's signatures take
// Eigen types; Eigen is fetched at build time there and is not part of this
// repository, so the Eigen overloads are compiled only when <Eigen/Sparse> is
// on the include path, and the same solvers are always available on the
// surface's own CsrMatrix / Vector types.
#ifndef CASK_SPARSE_LINEAR_SOLVERS_HPP
#define CASK_SPARSE_LINEAR_SOLVERS_HPP

#include "SparseMatrix.hpp"
#include "Utils.hpp"

#if defined(__has_include)
#if __has_include(<Eigen/Sparse>)
#include <Eigen/Sparse>
#define CASK_HAVE_EIGEN 1
#endif
#endif

namespace cask {
namespace sparse_linear_solvers {

struct SolveReport {
  int iterations = 0;
  bool converged = false;
  double microsecondsPerIteration = 0;
};

class Solver {
 public:
  int maxIterations = 2000;      // pcg's constants (SparseLinearSolvers.hpp:166-167)
  double tolerance = 1E-5;
  SolveReport report;
  virtual ~Solver() {}
  virtual void analyze(const CsrMatrix &) {}
  virtual void preprocess(const CsrMatrix &) {}
  virtual Vector solve(const CsrMatrix &A, const Vector &b) = 0;
#ifdef CASK_HAVE_EIGEN
  virtual void analyze(const Eigen::SparseMatrix<double> &) {}
  virtual void preprocess(const Eigen::SparseMatrix<double> &) {}
  virtual Eigen::VectorXd solve(const Eigen::SparseMatrix<double> &A, const Eigen::VectorXd &b);
#endif
};

// Un-preconditioned CG on the GPU; A must be symmetric positive definite and fully stored.
class DfeCgSolver : public Solver {
 public:
  using Solver::solve;
  Vector solve(const CsrMatrix &A, const Vector &b) override;
};

// Classical BiCG (A and A^T products) on the GPU for nonsymmetric systems.
class DfeBiCgSolver : public Solver {
 public:
  using Solver::solve;
  Vector solve(const CsrMatrix &A, const Vector &b) override;
};

// Un-preconditioned CG is what IdentityPreconditioner yields in the reference (:64-74).
class IdentityPreconditioner {
 public:
  IdentityPreconditioner(const CsrMatrix &) {}
  virtual ~IdentityPreconditioner() {}
  virtual std::vector<double> apply(const std::vector<double> &x) { return x; }
};

}  // namespace sparse_linear_solvers
}  // namespace cask

#endif  // CASK_SPARSE_LINEAR_SOLVERS_HPP
