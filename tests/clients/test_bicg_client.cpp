// The protocol of the reference's test/test_bicg.cpp (:4-13) and test/test_utils.hpp (:61-163) through the C++
// surface, on the GPU, on the surface's own CsrMatrix / Vector types (the reference's client is written against
// Eigen types; DfeBiCgSolver::solve itself is declared and never defined there).  For each system: b = A x0 with
// x0_i = 0.25 i (SimpleVectorGenerator), solve, compare with x0 under almost_equal(1E-8, 1E-11).
//   test_bicg_hip <path to bfwb62.mtx>
#include <cmath>
#include <iostream>
#include <string>

#include "cask/Converters.hpp"
#include "cask/IO.hpp"
#include "cask/SparseLinearSolvers.hpp"

using cask::CsrMatrix;
using cask::Vector;

static CsrMatrix scaledIdentity(int m, double s) {              // test_utils.hpp:75-96: one(m) and one(m) * 2
  cask::sparse::SparkCooMatrix<double> coo(m, m);
  for (int i = 0; i < m; i++) coo.data.push_back(std::make_tuple(i, i, s));
  return cask::converters::tripletToCsr(coo);
}

static bool almost_equal(double got, double exp) {             // test_utils.hpp:36 (restated; see oracle/)
  if (got == exp) return true;
  const double d = std::fabs(got - exp);
  return d <= 1E-11 || d <= 1E-8 * std::fmax(std::fabs(got), std::fabs(exp));
}

static int test(const CsrMatrix &a, cask::sparse_linear_solvers::Solver &solver, const char *what) {
  Vector x0(a.n);
  for (int i = 0; i < a.n; i++) x0[i] = i * 0.25;
  Vector b = a.dot(x0);
  Vector sol = solver.solve(a, b);
  int bad = 0;
  for (int i = 0; i < a.n; i++)
    if (!almost_equal(sol[i], x0[i])) {
      if (bad < 5) std::cerr << what << ": at " << i << " got: " << sol[i] << " exp: " << x0[i] << std::endl;
      bad++;
    }
  std::cout << what << ": iterations " << solver.report.iterations << " converged " << solver.report.converged
            << " mismatches " << bad << std::endl;
  return bad != 0 || !solver.report.converged;
}

int main(int argc, char **argv) {
  int status = 0;
  cask::sparse_linear_solvers::DfeBiCgSolver solver{};
  status |= test(scaledIdentity(16, 1.0), solver, "identity 16");
  status |= test(scaledIdentity(100, 2.0), solver, "2I 100");
  status |= test(scaledIdentity(10000, 2.0), solver, "2I 10000");
  if (argc > 1) {
    cask::io::MmReader<double> m(argv[1]);
    CsrMatrix a = cask::converters::tripletToCsr(m.mmreadMatrix(argv[1]));
    solver.tolerance = 1E-17;            // the default 1E-5 on r.r leaves errors above almost_equal on this system
    solver.maxIterations = 500;
    status |= test(a, solver, "bfwb62");
    cask::sparse_linear_solvers::DfeCgSolver cg{};              // the matrix is symmetric positive definite too
    cg.tolerance = 1E-17;
    status |= test(a, cg, "bfwb62 (CG)");
  }
  std::cout << (status == 0 ? "Test passed!" : "Test failed") << std::endl;
  return status;
}
