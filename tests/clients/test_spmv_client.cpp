// Integration client with the protocol of the reference's test/test_spmv.cpp
// (:19-83): read the matrix, x_i = 0.25 i, pick the implementation from the
// generated library's loader, preprocess, spmv, compare with a golden under
// almost_equal(got, exp, 1E-8, 1E-11).  The reference takes its golden from
// Eigen; this test client takes it from the CPU oracle (oracle/cask_oracle.c),
// which test infrastructure is allowed to link.
//   test_spmv_hip <matrix.mtx> [implId]                     the reference's form: one matrix per process (ctest -R hw,
//                                                           CMakeLists.txt:135-139)
//   test_spmv_hip --all <m1.mtx> <m2.mtx> ...               the same test over a list of matrices in ONE process (the
//                                                           suite's form: one HIP initialisation instead of 43)
//   test_spmv_hip --variants v1,v2,... <matrix.mtx>         the test once per value of CASK_HIP_VARIANT, one process; a value
//                                                           written !v must be REJECTED (std::invalid_argument)
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include "cask/GeneratedImplSupport.hpp"
#include "cask/IO.hpp"
#include "cask/Spmv.hpp"

extern "C" {
void oracle_csr_spmv(int32_t n_rows, const int32_t *row_ptr, const int32_t *col_ind, const double *values,
                     const double *x, double *y);
int64_t oracle_count_mismatches(int64_t n, const double *got, const double *expected, double rel_tol, double abs_tol,
                                int64_t *first_bad);
}

static int test(std::string path, int implId) {
  std::cout << "File: " << path << std::endl;
  std::cout << "Param MatrixPath " << path << std::endl;
  auto csrMatrix = cask::io::readMatrix(path);
  const int cols = csrMatrix.m;
  cask::Vector x(cols);
  for (int i = 0; i < cols; i++) x[i] = (double)i * 0.25;

  cask::runtime::SpmvImplementationLoader implLoader;
  cask::runtime::GeneratedSpmvImplementation *deviceImpl =
      implId == -1 ? implLoader.architectureWithParams(csrMatrix.n) : implLoader.architectureWithId(implId);
  if (!deviceImpl) {
    std::cout << "No implementation for " << csrMatrix.n << " rows" << std::endl;
    return 1;
  }
  cask::spmv::Spmv a(*deviceImpl);
  a.preprocess(csrMatrix);
  cask::Vector got = a.spmv(x);

  std::vector<double> exp(csrMatrix.n);
  oracle_csr_spmv(csrMatrix.n, csrMatrix.row_ptr.data(), csrMatrix.col_ind.data(), csrMatrix.values.data(),
                  x.data.data(), exp.data());
  int64_t first = -1;
  const int64_t bad = oracle_count_mismatches(csrMatrix.n, got.data.data(), exp.data(), 1E-8, 1E-11, &first);
  if (bad == 0) {
    std::cout << "Test passed!" << std::endl;
    return 0;
  }
  std::cerr << "Results didn't match" << std::endl;
  std::cerr << "At " << first << " got: " << got[first] << " exp: " << exp[first] << std::endl;
  std::cout << "Test failed: " << bad << " mismatches " << std::endl;
  return 1;
}

static int test_all(int argc, char **argv) {
  int failed = 0;
  for (int i = 2; i < argc; i++) {
    int status = 2;
    try {
      status = test(argv[i], -1);
    } catch (std::exception &e) {
      std::cout << "Exception: " << e.what() << std::endl;
    }
    std::cout << "Matrix " << argv[i] << (status == 0 ? " ok" : " FAILED") << std::endl;
    failed += status != 0;
  }
  std::cout << (argc - 2 - failed) << " of " << (argc - 2) << " matrices passed" << std::endl;
  return failed ? 1 : 0;
}

static int test_variants(const std::string &list, const std::string &path) {
  std::stringstream ss(list);
  std::string v;
  int failed = 0, n = 0;
  while (std::getline(ss, v, ',')) {
    const bool must_reject = !v.empty() && v[0] == '!';
    if (must_reject) v = v.substr(1);
    setenv("CASK_HIP_VARIANT", v.c_str(), 1);
    int status = 2;
    bool rejected = false;
    try {
      status = test(path, -1);
    } catch (std::invalid_argument &e) {
      rejected = true;
      std::cout << "Rejected: " << e.what() << std::endl;
    } catch (std::exception &e) {
      std::cout << "Exception: " << e.what() << std::endl;
    }
    const bool ok = must_reject ? rejected : status == 0;
    std::cout << "Variant " << v << (ok ? (must_reject ? " rejected as it must be" : " ok") : " FAILED") << std::endl;
    failed += !ok;
    n++;
  }
  unsetenv("CASK_HIP_VARIANT");
  std::cout << (n - failed) << " of " << n << " variants behaved" << std::endl;
  return failed ? 1 : 0;
}

int main(int argc, char **argv) {
  std::cout << "Program arguments:" << std::endl;
  for (int i = 0; i < argc; i++) std::cout << "   " << argv[i] << std::endl;
  if (argc > 2 && std::string(argv[1]) == "--all") {
    const int status = test_all(argc, argv);
    std::cout << (status == 0 ? "All tests passed!" : "Tests failed!") << std::endl;
    return status;
  }
  if (argc == 4 && std::string(argv[1]) == "--variants") {
    const int status = test_variants(argv[2], argv[3]);
    std::cout << (status == 0 ? "All tests passed!" : "Tests failed!") << std::endl;
    return status;
  }
  if (argc > 1) {
    int status = -1;
    try {
      if (argc == 2) status = test(argv[1], -1);
      else if (argc == 3) status = test(argv[1], std::stoi(argv[2]));
    } catch (std::exception &e) {
      std::cout << "Exception: " << e.what() << std::endl;
      status = 2;
    }
    std::cout << (status == 0 ? "All tests passed!" : "Tests failed!") << std::endl;
    return status;
  }
  return 1;
}
