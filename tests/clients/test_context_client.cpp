// The reference's high-level client tests (test/ClientTestSpmv.cpp:11-26, test/ClientTestCg.cpp:8-21) -- there they
// do not compile against include/Cask.hpp and end in ASSERT_TRUE(false); here the same calls run on the GPU and
// are checked: CaskContext::getSpmv -> preprocess -> spmv, CaskContext::getCg -> preprocess -> solve.
//   test_context_hip <dir with tiny*.mtx / tinysym*.mtx>
#include <cmath>
#include <iostream>
#include <string>

#include "cask/Cask.hpp"
#include "cask/IO.hpp"
#include "cask/SparseMatrix.hpp"

static int failures = 0;
#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::cerr << __FILE__ << ":" << __LINE__ << ": " #cond << std::endl;     \
      failures++;                                                              \
    }                                                                          \
  } while (0)

static bool close(const cask::Vector &got, std::initializer_list<double> exp) {
  if (got.size() != (int)exp.size()) return false;
  int i = 0;
  for (double e : exp)
    if (std::fabs(got[i++] - e) > 1e-12 * std::fmax(1.0, std::fabs(e))) return false;
  return true;
}

int main(int argc, char **argv) {
  using namespace cask;
  const std::string dir = argc > 1 ? argv[1] : "tests/golden/systems";
  {  // ClientTestSpmv.TinySymSpmv
    CaskContext cc;
    CsrMatrix a = io::readMatrix(dir + "/tinysym.mtx");
    Vector rhs = io::readVector(dir + "/tinysym_b.mtx");
    Vector v(rhs);
    auto spmv = cc.getSpmv(a);
    spmv.preprocess(a);
    CHECK(close(spmv.spmv(v), {5, 2, 3, 9}));                       // A * {1,2,3,4}
    Vector sol = io::readVector(dir + "/tinysym_sol.mtx");
    CHECK(close(spmv.spmv(sol), {1, 2, 3, 4}));                     // A * solution = right-hand side
  }
  {  // ClientCg.SimpleSystem, and the symmetric system of test/LinearSolvers.cpp:33-52
    CaskContext cc;
    SymCsrMatrix a = io::readSymMatrix(dir + "/tiny.mtx");
    Vector rhs = io::readVector(dir + "/tiny_b.mtx");
    Vector v(rhs);
    solvers::Cg cg = cc.getCg(a);
    cg.preprocess(a);
    CHECK(close(cg.solve(v), {1, 2, 3, 4}));
    CHECK(cg.converged && cg.iterations == 0);
    SymCsrMatrix s = io::readSymMatrix(dir + "/tinysym.mtx");
    Vector b = io::readVector(dir + "/tinysym_b.mtx");
    solvers::Cg cg2 = cc.getCg(s);
    cg2.preprocess(s);
    CHECK(close(cg2.solve(b), {-2, 2, 3, 3}));
    CHECK(cg2.converged);
  }
  std::cout << (failures == 0 ? "Test passed!" : "Test failed") << std::endl;
  return failures != 0;
}
