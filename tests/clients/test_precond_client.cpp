// The reference's preconditioning unit tests, through the C++ surface, on the GPU:
// test/MklLayer.cpp:10-50 (unittrsolve, exact), test/LinearSolvers.cpp:54-146 (ILUCompute2, ILUCompute,
// ILUComputeAndApply: exact / 4 ULP; CGSymWithILUPC: the iterate after 2000 stagnating passes).
// Inputs and expected values are the ones the reference's tests hold; the checks use the same
// comparison the reference uses, except CGSymWithILUPC where the device's dot products add in a different
// order than MKL's and the bar is relative 1e-9 (ASSERT_DOUBLE_EQ in the reference).
//   test_precond_hip <dir with tinysym.mtx / tinysym_b.mtx>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <string>

#include "cask/IO.hpp"
#include "cask/MklLayer.hpp"
#include "cask/SparseLinearSolvers.hpp"

static int failures = 0;
#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::cerr << __FILE__ << ":" << __LINE__ << ": " #cond << std::endl;     \
      failures++;                                                              \
    }                                                                          \
  } while (0)

static bool double_eq(double a, double b) {      // gtest's AlmostEquals: within 4 ULP
  if (a == b) return true;
  int64_t ia, ib;
  std::memcpy(&ia, &a, 8);
  std::memcpy(&ib, &b, 8);
  if ((ia < 0) != (ib < 0)) return false;
  return std::llabs(ia - ib) <= 4;
}

int main(int argc, char **argv) {
  using namespace cask;
  using cask::sparse_linear_solvers::ILUPreconditioner;
  const std::string dir = argc > 1 ? argv[1] : "tests/golden/systems";

  {  // TestMklLayer
    CsrMatrix id{DokMatrix{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    CHECK(mkl::unittrsolve(id, {1, 2, 3, 4}, true) == (std::vector<double>{1, 2, 3, 4}));
    CsrMatrix lo{DokMatrix{1, 0, 0, 0, 1, 1, 0, 0, 0, 1, 1, 0, 1, 0, 0, 1}};
    CHECK(mkl::unittrsolve(lo, {-2, 2, 3, 4}, true) == (std::vector<double>{-2, 4, -1, 6}));
    CsrMatrix up{DokMatrix{1, 0, 0, 1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 1}};
    CHECK(mkl::unittrsolve(up, {-2, 2, 3, 4}, false) == (std::vector<double>{-6, 3, -1, 4}));
    // the raw-array overload with the 1-based arrays the reference prepares for MKL
    std::vector<double> res(4);
    auto rp = lo.getRowPtrWithOneBasedIndex();
    auto ci = lo.getColIndWithOneBasedIndex();
    mkl::unittrsolve(lo.values.data(), rp.data(), ci.data(), {-2, 2, 3, 4}, res.data(), true);
    CHECK(res == (std::vector<double>{-2, 4, -1, 6}));
  }
  {  // ILUCompute2
    CsrMatrix a{DokMatrix{2, 1, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1}};
    ILUPreconditioner ilupc{a};
    DokMatrix exp{2, 1, 1, 1, 0.5, 0.5, 0, 0, 0.5, 0, 0.5, 0, 0.5, 0, 0, 0.5};
    CHECK(ilupc.pc.n == exp.n);
    CHECK(ilupc.pc.nnzs == exp.nnzs);
    CHECK(ilupc.pc == exp);
    // ILUComputeAndApply
    auto res = ilupc.apply({1, 2, 3, 4});
    const std::vector<double> want{-16.25, 7, 11, 15};
    for (size_t i = 0; i < res.size(); i++) CHECK(double_eq(res[i], want[i]));
  }
  {  // ILUCompute
    SymCsrMatrix a = io::readSymMatrix(dir + "/tinysym.mtx");
    CsrMatrix explicitA(a.matrix.toDok().explicitSymmetric());
    ILUPreconditioner explicitPc{explicitA};
    CsrMatrix csrPc{explicitPc.pc};
    CHECK(csrPc.row_ptr == (std::vector<int>{0, 2, 3, 4, 6}));
    CHECK(csrPc.col_ind == (std::vector<int>{0, 3, 1, 2, 0, 3}));
    CHECK(csrPc.values == (std::vector<double>{1, 1, 1, 1, 1, 1}));
  }
  {  // CGSymWithILUPC
    Vector rhs = io::readVector(dir + "/tinysym_b.mtx");
    SymCsrMatrix a = io::readSymMatrix(dir + "/tinysym.mtx");
    int iterations = 0;
    Vector sol(a.n);
    const bool conv = sparse_linear_solvers::pcg<double, ILUPreconditioner>(a.matrix, &rhs[0], &sol[0], iterations);
    const double want[4] = {-1.9982580059252246, 2.0000862488691915, 3.0001293733037859, 2.9987581910958183};
    CHECK(!conv);
    CHECK(iterations == 1999);
    for (int i = 0; i < 4; i++) CHECK(std::fabs(sol[i] - want[i]) <= 1e-9 * std::fabs(want[i]));
    // CGSymMatrix with the identity preconditioner (test/LinearSolvers.cpp:33-52)
    int it2 = 0;
    Vector sol2(a.n);
    CHECK((sparse_linear_solvers::pcg<double, sparse_linear_solvers::IdentityPreconditioner>(a.matrix, &rhs[0], &sol2[0],
                                                                                            it2)));
    const double want2[4] = {-2, 2, 3, 3};
    for (int i = 0; i < 4; i++) CHECK(std::fabs(sol2[i] - want2[i]) <= 1e-12);
  }
  if (failures == 0) std::cout << "Test passed!" << std::endl;
  else std::cout << "Test failed: " << failures << " checks" << std::endl;
  return failures != 0;
}
