// Solver classes of the CASK surface (reference: src/runtime/SparseLinearSolvers.hpp:24-61).
// DfeCgSolver::solve and DfeBiCgSolver::solve are declared but never defined in
// the reference; here they run on the GPU.  The reference's signatures take
// Eigen types; Eigen is fetched at build time there and is not part of this
// repository, so the Eigen overloads are compiled only when <Eigen/Sparse> is
// on the include path, and the same solvers are always available on the
// surface's own CsrMatrix / Vector types.
#ifndef CASK_SPARSE_LINEAR_SOLVERS_HPP
#define CASK_SPARSE_LINEAR_SOLVERS_HPP

#include <memory>

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

// ILU(0) preconditioner with the reference's members and semantics (SparseLinearSolvers.hpp:77-156):
// `pc` is the factored matrix in the pattern of `a`, `l` / `u` its lower / upper triangle INCLUDING the
// diagonal, apply() solves with both of them dividing by that diagonal (mkl_dcsrtrsv, diag 'N').  The
// factorisation runs once on the host, every apply() on the GPU (level-scheduled triangular solves).
struct cask_precond_deleter { void operator()(void *p) const; };
class ILUPreconditioner {
 public:
  DokMatrix pc;
  CsrMatrix l, u;
  std::shared_ptr<void> device;                  // cask_hip_precond*

  // pre - a is a symmetric matrix (the reference only checks CsrMatrix::isSymmetric(), which is `true`)
  ILUPreconditioner(const CsrMatrix &a);
  virtual ~ILUPreconditioner() {}
  virtual std::vector<double> apply(const std::vector<double> &x);
  void pretty_print() { pc.pretty_print(); }
};

// Standard preconditioned CG with the reference's signature and constants (tol 1E-5, 2000 passes,
// SparseLinearSolvers.hpp:162-239): `a` is the stored (lower) triangle of a symmetric matrix, as
// SymCsrMatrix::matrix holds it; the preconditioner is built from that same CsrMatrix (:171); `x`
// holds the initial guess; `iterations` is written at the end of every non-converged pass.
// Returns true when r.z <= tol^2 was reached.  Runs on the GPU (cask_hip_pcg).
bool pcgIdentity(const CsrMatrix &a, double *rhs, double *x, int &iterations, bool verbose, cask::utils::Timer *t);
bool pcgIlu(const CsrMatrix &a, double *rhs, double *x, int &iterations, bool verbose, cask::utils::Timer *t);
template <typename Precon> struct PcgDispatch;
template <> struct PcgDispatch<IdentityPreconditioner> {
  static bool run(const CsrMatrix &a, double *rhs, double *x, int &it, bool v, cask::utils::Timer *t) {
    return pcgIdentity(a, rhs, x, it, v, t);
  }
};
template <> struct PcgDispatch<ILUPreconditioner> {
  static bool run(const CsrMatrix &a, double *rhs, double *x, int &it, bool v, cask::utils::Timer *t) {
    return pcgIlu(a, rhs, x, it, v, t);
  }
};
template <typename T = double, typename Precon = IdentityPreconditioner>
bool pcg(const CsrMatrix &a, double *rhs, double *x, int &iterations, bool verbose = false,
         cask::utils::Timer *t = nullptr) {
  static_assert(sizeof(T) == sizeof(double), "the GPU solvers are fp64");
  return PcgDispatch<Precon>::run(a, rhs, x, iterations, verbose, t);
}

}  // namespace sparse_linear_solvers
}  // namespace cask

#endif  // CASK_SPARSE_LINEAR_SOLVERS_HPP
