// CPU unit tests of the C++ host surface (include/cask/*.hpp) against the known
// answers of the reference's gtest suites: test/SparseMatrix.cpp, test/Io.cpp,
// test/TestUtils.cpp.  No gtest in the image: a few macros do.  Run by
// tests/test_host_cpp.py; argv[1] = directory with the plain .mtx fixtures.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "cask/Converters.hpp"
#include "cask/IO.hpp"
#include "cask/SparseMatrix.hpp"
#include "cask/Utils.hpp"

static int failures = 0, checks = 0;
#define CHECK(cond)                                                                   \
  do {                                                                                \
    checks++;                                                                         \
    if (!(cond)) {                                                                    \
      failures++;                                                                     \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);                     \
    }                                                                                 \
  } while (0)
#define CHECK_THROWS(expr, type)            \
  do {                                      \
    checks++;                               \
    bool caught = false;                    \
    try { expr; } catch (type &) { caught = true; } \
    if (!caught) { failures++; std::printf("FAIL %s:%d  %s should throw\n", __FILE__, __LINE__, #expr); } \
  } while (0)

using namespace cask;

static void test_sparse_matrix() {
  // test/SparseMatrix.cpp:8-27
  DokMatrix d{1, 1, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1};
  CHECK(d.nnzs == 10 && d.n == 4);
  CHECK(d.at(0, 0) == 1 && d.at(0, 3) == 1 && d.at(3, 3) == 1 && d.at(1, 2) == 0);
  // :29-38
  DokMatrix one_row{1, {4, 5, 3, 2}};
  CHECK(one_row.nnzs == 4 && one_row.n == 1 && one_row.at(0, 1) == 5 && one_row.at(0, 3) == 2);
  // :40-61 explicit symmetry
  DokMatrix lower{1, 0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1};
  CHECK(lower.nnzs == 7);
  DokMatrix sym = lower.explicitSymmetric();
  CHECK(sym.nnzs == 10 && sym.n == 4);
  CHECK(sym == d);
  // :63-73 Dok dot
  CHECK(lower.dot(Vector{1, 2, 3, 4}) == (Vector{1, 3, 4, 5}));
  // :75-87 Csr <-> Dok
  DokMatrix dokA{2, 1, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1};
  CsrMatrix a{dokA};
  CHECK(a.toDok().dok == dokA.dok && a.toDok() == dokA);
  // :90-137 triangles
  CsrMatrix m{d};
  CHECK(m.getLowerTriangular() == CsrMatrix{lower});
  CHECK(m.getUpperTriangular() == (CsrMatrix{DokMatrix{1, 1, 1, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}}));
  // :139-161 row slices
  CsrMatrix s{DokMatrix{1, 2, 5, 4, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
  CsrMatrix e1(1, {1, 2, 5, 4});
  auto u = s.sliceRows(0, 1);
  CHECK(u.values == e1.values && u.row_ptr == e1.row_ptr && u.col_ind == e1.col_ind);
  CsrMatrix e2{2, {0, 1, 0, 0, 0, 0, 1, 0}};
  auto u2 = s.sliceRows(1, 2);
  CHECK(u2.values == e2.values && u2.row_ptr == e2.row_ptr && u2.col_ind == e2.col_ind);
  CHECK(s.sliceRows(0, 4) == s);
  // :164-178 SymCsr dot
  SymCsrMatrix sm{DokMatrix{1, 0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 1, 0, 1, 1}};
  CHECK(sm.dot(Vector{1, 2, 3, 4}) == (Vector{10, 3, 8, 8}));
  // :180-191 Csr dot
  CsrMatrix cm{DokMatrix{1, 0, 0, 0, 1, 0, 1, 0, 0, 1, 1, 0, 0, 0, 1, 1}};
  CHECK(cm.dot(Vector{1, 2, 3, 4}) == (Vector{1, 4, 5, 7}));
  // :193-203 vectors
  CHECK((Vector{1, 6, 9, 4} - Vector{3, 4, 5, 7}) == (Vector{-2, 2, 4, -3}));
  CHECK((Vector{1, 2, 3, 1, 1}).norm() == 4);
  CHECK_THROWS((Vector{1, 2} - Vector{1}), std::invalid_argument);
  // column stripes keep the reference's DFE-stream shape (SparseMatrix.hpp:459-482)
  CsrMatrix wide{2, {1, 2, 3, 4, 5, 6, 7, 8}};
  auto stripes = wide.sliceColumns(3);
  CHECK(stripes.size() == 2);
  CHECK((stripes[0].values == std::vector<double>{1, 2, 3, 5, 6, 7}) && (stripes[0].row_ptr == std::vector<int>{3, 6}));
  CHECK((stripes[1].values == std::vector<double>{4, 8}) && (stripes[1].col_ind == std::vector<int>{0, 0}) &&
        (stripes[1].row_ptr == std::vector<int>{1, 2}));
}

static void test_io(const std::string &dir) {
  // test/Io.cpp:7-29 exact parsed values
  CsrMatrix a = io::readMatrix(dir + "/matrices/test_dense_4.mtx");
  CHECK(a.n == 4 && a.m == 4 && a.nnzs == 16);
  DokMatrix dk = a.toDok();
  CHECK(dk.at(0, 0) == 0.160600717781 && dk.at(0, 3) == 0.131826930446 && dk.at(1, 0) == 0.72239480913);
  CHECK(dk.at(2, 1) == 0.00982158850327 && dk.at(3, 2) == 0.971614586317 && dk.at(3, 3) == 0.997169318601);
  // :31-38 header
  io::MmInfo info = io::readHeader(dir + "/systems/tinysym.mtx");
  CHECK(info.symmetry == "symmetric" && info.format == "coordinate" && info.type == "matrix" && info.dataType == "real");
  // :41-63 symmetric reads
  SymCsrMatrix t = io::readSymMatrix(dir + "/systems/tiny.mtx");
  CHECK(t.n == 4 && t.m == 4 && t.nnzs == 4);
  CHECK(t.matrix.toDok().explicitSymmetric() == (DokMatrix{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}));
  SymCsrMatrix ts = io::readSymMatrix(dir + "/systems/tinysym.mtx");
  CHECK(ts.nnzs == 6);
  CHECK(ts.matrix.toDok().explicitSymmetric() == (DokMatrix{1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 2}));
  // the fast readMatrix path equals the Dok path on a symmetric file
  CsrMatrix fast = io::readMatrix(dir + "/matrices/bfwb62.mtx");
  io::MmInfo bi = io::readHeader(dir + "/matrices/bfwb62.mtx");
  CsrMatrix slow(io::readDokMatrix(dir + "/matrices/bfwb62.mtx", bi).explicitSymmetric());
  CHECK(fast.values == slow.values && fast.col_ind == slow.col_ind && fast.row_ptr == slow.row_ptr && fast.nnzs == 342);
  // vectors
  Vector b = io::readVector(dir + "/systems/tinysym_sol.mtx");
  CHECK(b == (Vector{-2, 2, 3, 3}));
  // MmReader: sorted triplets with the mirror entries
  io::MmReader<double> r(dir + "/systems/tinysym.mtx");
  auto coo = r.mmreadMatrix("");
  CHECK(coo.n == 4 && coo.m == 4 && coo.data.size() == 6);
  CHECK(std::get<0>(coo.data[1]) == 0 && std::get<1>(coo.data[1]) == 3 && std::get<2>(coo.data[1]) == 1);
  // test_spmv.cpp's golden operand without Eigen: the COO goes to CSR and dot() is the row-major product
  CsrMatrix fromCoo = converters::tripletToCsr(coo);
  CHECK(fromCoo == io::readMatrix(dir + "/systems/tinysym.mtx"));
  CHECK(fromCoo.dot(Vector{1, 2, 3, 4}) == (Vector{5, 2, 3, 9}));
  cask::sparse::SparkCooMatrix<double> dup(2, 2);
  dup.data.push_back(std::make_tuple(1, 0, 3.0));
  dup.data.push_back(std::make_tuple(0, 1, 1.0));
  dup.data.push_back(std::make_tuple(1, 0, 5.0));                     // later duplicate wins (DokMatrix::set)
  CsrMatrix d2 = converters::tripletToCsr(dup);
  CHECK(d2.nnzs == 2 && d2.values == (std::vector<double>{1.0, 5.0}) && d2.row_ptr == (std::vector<int>{0, 1, 2}));
  // binary cache: round trip, and the cached reader returns the same matrix on the first and the second read
  {
    const std::string tmp = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/cask_host_test";
    io::writeCsrBinary(tmp + ".csrbin", fast);
    CHECK(io::readCsrBinary(tmp + ".csrbin") == fast);
    std::ofstream(tmp + ".junk") << "not a matrix";
    CHECK_THROWS(io::readCsrBinary(tmp + ".junk"), std::invalid_argument);
    std::ifstream src(dir + "/matrices/bfwb62.mtx", std::ios::binary);
    std::ofstream dst(tmp + ".mtx", std::ios::binary);
    dst << src.rdbuf();
    dst.close();
    std::remove((tmp + ".mtx.csrbin").c_str());
    CHECK(io::readMatrixCached(tmp + ".mtx") == fast);
    CHECK(io::readMatrixCached(tmp + ".mtx") == fast);        // from the cache this time
    CHECK(io::readCsrBinary(tmp + ".mtx.csrbin") == fast);
    // a text file that changed after the cache was written is re-parsed, not served from the stale cache
    {
      std::ofstream edit(tmp + ".mtx", std::ios::binary);
      edit << "%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 3.5\n2 2 4.5\n";
    }
    CsrMatrix edited = io::readMatrixCached(tmp + ".mtx");
    CHECK(edited.n == 2 && edited.nnzs == 2 && edited.values == (std::vector<double>{3.5, 4.5}));
    CHECK(io::readCsrBinary(tmp + ".mtx.csrbin") == edited);  // and the cache was refreshed
    // a cache with a broken row_ptr or an out-of-range column is rejected
    {
      CsrMatrix bad = edited;
      bad.col_ind[1] = 7;
      io::writeCsrBinary(tmp + ".bad.csrbin", bad);
      CHECK_THROWS(io::readCsrBinary(tmp + ".bad.csrbin"), std::invalid_argument);
    }
  }
  // errors
  CHECK_THROWS(io::readHeader(dir + "/nope.mtx"), std::invalid_argument);
  CHECK_THROWS(io::readSymMatrix(dir + "/matrices/test_dense_4.mtx"), std::invalid_argument);
  CHECK_THROWS(io::readMatrix(dir + "/systems/tiny_b.mtx").n, std::exception);   // an array file is not a coordinate matrix
}

static void test_utils() {
  using namespace cask::utils;
  // test/TestUtils.cpp:7-49
  Parameter<int> p{"p", 1, 1, 1};
  CHECK(p.first().value == 1);
  std::vector<Parameter<int>> params = {{"numPipes", 1, 3, 1}, {"frequency", 100, 150, 10}};
  ChainedParameterRange<int> cpr(params);
  cpr.start();
  auto at = [&](int a, int b) { return cpr.getParam("numPipes").value == a && cpr.getParam("frequency").value == b; };
  CHECK(at(1, 100));
  cpr.next(); CHECK(at(2, 100));
  cpr.next(); CHECK(at(3, 100));
  cpr.next(); CHECK(at(1, 110));
  cpr.start(); CHECK(at(1, 100));
  for (int i = 0; i < 15; i++) cpr.next();
  CHECK(at(1, 150));
  cpr.next(); CHECK(at(2, 150));
  cpr.next(); CHECK(at(3, 150));
  CHECK(!cpr.hasNext());
  CHECK_THROWS(cpr.next(), std::invalid_argument);
  CHECK_THROWS(cpr.getParam("nope"), std::invalid_argument);
  // alignment helpers (used by the DFE stream format)
  std::vector<double> v(5, 1.0);
  align(v, 384);
  CHECK(v.size() == 48);
  std::vector<int> w(16, 1);
  align(w, (int)sizeof(int) * 16);
  CHECK(w.size() == 16);
  CHECK(align(385, 384) == 768 && align(384, 384) == 384 && ceilDivide(7, 2) == 4 && size_bytes(v) == 384);
  CHECK_THROWS(ceilDivide(-1, 2), std::invalid_argument);
  Timer t;
  t.tic("a");
  CHECK(t.toc("a").count() >= 0 && t.get("a").count() >= 0);
  CHECK_THROWS(t.toc("b"), std::invalid_argument);
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  test_sparse_matrix();
  test_io(dir);
  test_utils();
  std::printf("%d checks, %d failures\n", checks, failures);
  return failures ? 1 : 0;
}
