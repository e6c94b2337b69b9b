// Host runtime of the CASK surface on MI355X: the bodies of cask::spmv::Spmv,
// cask::solvers::Cg, the Dfe*Solver classes and the measured DSE, all of them
// thin C++ over the C ABI of include/cask_hip.h.  Replaces the reference's
// src/runtime/Spmv.cpp (host blocking + SLiC calls), Cg.cpp (empty) and Dse.cpp.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "cask/Cask.hpp"
#include "cask/Cg.hpp"
#include "cask/MklLayer.hpp"
#include "cask/Dse.hpp"
#include "cask/SparseLinearSolvers.hpp"
#include "cask/Spmv.hpp"

// Optional hook of a generated library (tools/gen_impl.py): the design point the
// DSE chose for implementation `id`.  Weak: absent in mock-style libraries.
extern "C" int cask_hip_generated_design_point(int id, cask_hip_params *out) __attribute__((weak));

namespace {

// C status -> the exception types the reference throws (Spmv.cpp:189-232)
void check(int rc, const char *what) {
  if (rc == CASK_HIP_OK) return;
  std::string msg = std::string(what) + ": " + cask_hip_last_error();
  if (rc == CASK_HIP_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// Run-time overrides for unchanged clients (the reference selects a design with test_spmv's optional implId
// argument and a target with -t {dfe,sim,dfe_mock}; a GPU build has one target and many design points):
//   CASK_HIP_VARIANT = auto | vector | merge | merge_wave | scan | slice   overrides the variant of every matrix uploaded
//   CASK_HIP_TILE    = <doubles> | -1                        overrides the x tile width (-1: no tile)
cask_hip_params with_env_overrides(const cask_hip_params *p) {
  cask_hip_params out{};
  if (p) out = *p;
  if (const char *v = std::getenv("CASK_HIP_VARIANT")) {
    const std::string s(v);
    if (s == "auto") out.variant = CASK_HIP_VARIANT_AUTO;
    else if (s == "vector") out.variant = CASK_HIP_VARIANT_VECTOR;
    else if (s == "merge") out.variant = CASK_HIP_VARIANT_MERGE;
    else if (s == "merge_wave") out.variant = CASK_HIP_VARIANT_MERGE_WAVE;
    else if (s == "scan") out.variant = CASK_HIP_VARIANT_SCAN;
    else if (s == "slice") out.variant = CASK_HIP_VARIANT_SLICE;
    else throw std::invalid_argument("CASK_HIP_VARIANT must be auto, vector, merge, merge_wave, scan or slice");
  }
  if (const char *t = std::getenv("CASK_HIP_TILE")) out.tile_width = std::atoi(t);
  return out;
}

std::shared_ptr<cask_hip_matrix> upload(const cask::CsrMatrix &a, const cask_hip_params *p_in) {
  if (static_cast<int>(a.row_ptr.size()) != a.n + 1)
    throw std::invalid_argument("CsrMatrix: row_ptr must have n+1 entries");
  const cask_hip_params prm = with_env_overrides(p_in);
  const cask_hip_params *p = &prm;
  cask_hip_matrix *h = nullptr;
  check(cask_hip_csr_create(a.n, a.m, static_cast<int64_t>(a.col_ind.size()), a.row_ptr.data(), a.col_ind.data(),
                            a.values.data(), p, &h),
        "cask_hip_csr_create");
  return std::shared_ptr<cask_hip_matrix>(h, [](cask_hip_matrix *m) { cask_hip_csr_destroy(m); });
}

int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

cask::CsrMatrix expandSymmetric(const cask::CsrMatrix &lower) {
  // mirror the stored triangle (reference: DokMatrix::explicitSymmetric, SparseMatrix.hpp:156-189)
  const int n = lower.n;
  std::vector<int> count(static_cast<size_t>(n) + 1, 0);
  // Only the stored LOWER triangle counts, as for the reference's mkl_dcsrsymv(uplo = 'l')
  // (SparseLinearSolvers.hpp:189,206), which never reads entries above the diagonal: a caller that hands over
  // an explicitly symmetric matrix must get the same operator, not one with every off-diagonal doubled.
  for (int i = 0; i < n; i++)
    for (int k = lower.row_ptr[i]; k < lower.row_ptr[i + 1]; k++) {
      if (lower.col_ind[k] > i) continue;
      count[i + 1]++;
      if (lower.col_ind[k] != i) count[lower.col_ind[k] + 1]++;
    }
  for (int i = 0; i < n; i++) count[i + 1] += count[i];
  cask::CsrMatrix full;
  full.n = n;
  full.m = lower.m;
  full.nnzs = count[n];
  full.row_ptr = count;
  full.col_ind.resize(count[n]);
  full.values.resize(count[n]);
  std::vector<int> fill(count.begin(), count.end() - 1);
  for (int i = 0; i < n; i++)          // stored entries: columns <= i, ascending
    for (int k = lower.row_ptr[i]; k < lower.row_ptr[i + 1]; k++) {
      if (lower.col_ind[k] > i) continue;
      full.col_ind[fill[i]] = lower.col_ind[k];
      full.values[fill[i]++] = lower.values[k];
    }
  for (int i = 0; i < n; i++)          // mirrored entries: columns > row, ascending because i ascends
    for (int k = lower.row_ptr[i]; k < lower.row_ptr[i + 1]; k++) {
      const int j = lower.col_ind[k];
      if (j >= i) continue;
      full.col_ind[fill[j]] = i;
      full.values[fill[j]++] = lower.values[k];
    }
  return full;
}

}  // namespace

namespace cask {
namespace spmv {

void Spmv::preprocess(const CsrMatrix &m) {
  this->mat = m;
  cask_hip_params p{};
  bool from_library = false;
  if (impl.id >= 0 && cask_hip_generated_design_point) from_library = cask_hip_generated_design_point(impl.id, &p) == 0;
  if (!from_library) {
    // architecture integers -> design point (header comment of Spmv.hpp)
    p.variant = CASK_HIP_VARIANT_AUTO;
    p.tile_width = impl.cache_size > 0 ? impl.cache_size : 0;
    p.lanes_per_row = impl.input_width > 0 ? std::min(64, pow2_floor(impl.input_width)) : 0;
  }
  device = upload(this->mat, &p);
}

Vector Spmv::spmv(const Vector &x) {
  if (!device) throw std::runtime_error("Matrix not defined - run preprocess on the matrix");
  if (x.size() != mat.m) {
    std::stringstream ss;
    ss << "Vector has " << x.size() << " entries, matrix has " << mat.m << " columns";
    throw std::invalid_argument(ss.str());
  }
  if (impl.max_rows < mat.n) {
    std::stringstream ss;
    ss << "Matrix is too large! Maximum supported rows: " << impl.max_rows << " actual rows: " << mat.n;
    throw std::invalid_argument(ss.str());
  }
  Vector y(mat.n);
  std::cout << "Running on MI355X" << std::endl;
  const auto t0 = std::chrono::high_resolution_clock::now();
  check(cask_hip_spmv(device.get(), x.data.data(), y.data.data()), "cask_hip_spmv");
  lastSeconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  const cask_hip_params p = designPoint();
  cask_hip_csr_info info;
  cask_hip_csr_get_info(device.get(), &info);
  // same keys as the reference's log (Spmv.cpp:286-301); times in seconds like there
  utils::logResult("Input width ", p.lanes_per_row);
  utils::logResult("Pipes ", info.grid);
  utils::logResult("Iterations", 1);
  utils::logResult("Took (ms)", lastSeconds);
  utils::logResult("Gflops (actual)", lastSeconds > 0 ? 2.0 * mat.nnzs / lastSeconds / 1E9 : 0.0);
  return y;
}

cask_hip_params Spmv::designPoint() const {
  if (!device) throw std::runtime_error("Matrix not defined - run preprocess on the matrix");
  cask_hip_params p{};
  cask_hip_csr_get_params(device.get(), &p);
  return p;
}

void Spmv::setDesignPoint(const cask_hip_params &p) {
  if (!device) throw std::runtime_error("Matrix not defined - run preprocess on the matrix");
  check(cask_hip_csr_set_params(device.get(), &p), "cask_hip_csr_set_params");
}

double Spmv::tune(int warmup, int iters) {
  if (!device) throw std::runtime_error("Matrix not defined - run preprocess on the matrix");
  std::vector<cask_hip_tune_point> pts(1024);
  int n = 0, best = -1;
  check(cask_hip_tune(device.get(), nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, warmup, iters,
                      pts.data(), static_cast<int>(pts.size()), &n, &best),
        "cask_hip_tune");
  return best >= 0 ? pts[best].gflops : 0.0;
}

double Spmv::measureMicroseconds(int warmup, int iters) {
  if (!device) throw std::runtime_error("Matrix not defined - run preprocess on the matrix");
  std::vector<cask_hip_tune_point> pt(1);
  int n = 0, best = -1;
  const cask_hip_params p = designPoint();
  const int32_t variant = p.variant, lanes = p.lanes_per_row, tile = p.tile_width, wg = p.wg_size, items = p.items_per_thread;
  check(cask_hip_tune(device.get(), &variant, 1, &lanes, 1, &tile, 1, &wg, 1, &items, 1, warmup, iters, pt.data(), 1,
                      &n, &best),
        "cask_hip_tune");
  return n > 0 ? pt[0].usec : 0.0;
}

double Spmv::getFrequency() {
  cask_hip_device_props props;
  if (cask_hip_device_props_get(0, &props) == CASK_HIP_OK && props.clock_mhz > 0) return props.clock_mhz * 1E6;
  return 2400.0 * 1E6;
}

double Spmv::getEstimatedClockCycles() { return measureMicroseconds() * 1E-6 * getFrequency(); }

double Spmv::getEstimatedGFlops(const model::DeviceModel &) {
  const double us = measureMicroseconds();
  return us > 0 ? 2.0 * mat.nnzs / us * 1E-3 : 0.0;
}

model::HardwareModel Spmv::getEstimatedHardwareModel(const model::DeviceModel &deviceModel) {
  if (this->mat.n == 0) throw std::runtime_error("Matrix not defined - run preprocess on the matrix");
  return getEstimatedHardwareModel(deviceModel, this->mat.n);
}

model::HardwareModel Spmv::getEstimatedHardwareModel(const model::DeviceModel &, const int) {
  model::HardwareModel hw;
  if (!device) return hw;
  cask_hip_csr_info info;
  cask_hip_csr_get_info(device.get(), &info);
  hw.ldsBytesPerWorkgroup = info.lds_bytes;
  hw.workgroups = info.grid;
  const double us = measureMicroseconds();
  hw.memoryBandwidth = us > 0 ? info.algorithmic_bytes / us * 1E-3 : 0.0;
  return hw;
}

}  // namespace spmv

namespace solvers {

void Cg::preprocess(const cask::CsrMatrix &a) {
  if (a.n != a.m) throw std::invalid_argument("Cg needs a square matrix");
  n = a.n;
  device = upload(a, nullptr);
}

void Cg::preprocess(cask::SymCsrMatrix &a) { preprocess(expandSymmetric(a.matrix)); }

Vector Cg::solve(Vector &rhs) {
  if (!device) throw std::runtime_error("Cg: run preprocess on the matrix first");
  if (rhs.size() != n) throw std::invalid_argument("Cg: right-hand side has the wrong length");
  Vector x(n);
  int32_t it = 0, conv = 0;
  check(cask_hip_cg(device.get(), rhs.data.data(), x.data.data(), maxIterations, tolerance, &it, &conv,
                    &microsecondsPerIteration),
        "cask_hip_cg");
  iterations = it;
  converged = conv != 0;
  return x;
}

}  // namespace solvers

namespace sparse_linear_solvers {

namespace {
template <typename F>
Vector run_solver(Solver &s, const CsrMatrix &A, const Vector &b, F fn, const char *name) {
  if (A.n != A.m) throw std::invalid_argument(std::string(name) + " needs a square matrix");
  if (b.size() != A.n) throw std::invalid_argument(std::string(name) + ": right-hand side has the wrong length");
  auto dev = upload(A, nullptr);
  Vector x(A.n);
  int32_t it = 0, conv = 0;
  double us = 0;
  check(fn(dev.get(), b.data.data(), x.data.data(), s.maxIterations, s.tolerance, &it, &conv, &us), name);
  s.report.iterations = it;
  s.report.converged = conv != 0;
  s.report.microsecondsPerIteration = us;
  return x;
}
}  // namespace

Vector DfeCgSolver::solve(const CsrMatrix &A, const Vector &b) { return run_solver(*this, A, b, cask_hip_cg, "cask_hip_cg"); }

Vector DfeBiCgSolver::solve(const CsrMatrix &A, const Vector &b) {
  return run_solver(*this, A, b, cask_hip_bicg, "cask_hip_bicg");
}

// ---- preconditioning ---------------------------------------------------------------------------
void cask_precond_deleter::operator()(void *p) const { cask_hip_precond_destroy(static_cast<cask_hip_precond *>(p)); }

namespace {
std::shared_ptr<void> make_precond(int kind, const CsrMatrix &a) {
  cask_hip_precond *h = nullptr;
  check(cask_hip_precond_create(kind, a.n, a.nnzs, a.row_ptr.data(), a.col_ind.data(), a.values.data(), &h),
        "cask_hip_precond_create");
  return std::shared_ptr<void>(h, cask_precond_deleter());
}
}  // namespace

ILUPreconditioner::ILUPreconditioner(const CsrMatrix &a) {
  if (!a.isSymmetric()) throw std::invalid_argument("ILUPreconditioner only supports symmetric CSR matrices");
  if (a.n != a.m) throw std::invalid_argument("ILUPreconditioner needs a square matrix");
  device = make_precond(CASK_HIP_PRECOND_ILU0, a);
  CsrMatrix factored = a;
  check(cask_hip_precond_factor_values(static_cast<cask_hip_precond *>(device.get()), factored.values.data()),
        "cask_hip_precond_factor_values");
  pc = factored.toDok();
  l = CsrMatrix{pc.getLowerTriangular()};
  u = CsrMatrix{pc.getUpperTriangular()};
}

std::vector<double> ILUPreconditioner::apply(const std::vector<double> &x) {
  if ((int)x.size() != pc.n) throw std::invalid_argument("ILUPreconditioner::apply: vector has the wrong length");
  std::vector<double> res(x.size());
  check(cask_hip_precond_apply(static_cast<cask_hip_precond *>(device.get()), x.data(), res.data()),
        "cask_hip_precond_apply");
  return res;
}

namespace {
bool run_pcg(const CsrMatrix &a, cask_hip_precond *pc, double *rhs, double *x, int &iterations, bool verbose,
             cask::utils::Timer *t) {
  if (a.n != a.m) throw std::invalid_argument("pcg needs a square matrix");
  if (t) t->tic("cg:setup");
  auto dev = upload(expandSymmetric(a), nullptr);      // mkl_dcsrsymv('l') semantics: a holds the lower triangle
  if (t) t->toc("cg:setup");
  if (t) t->tic("cg:solve");
  int32_t it = iterations, conv = 0;
  double us = 0;
  check(cask_hip_pcg(dev.get(), pc, rhs, x, 2000, 1E-5, &it, &conv, &us), "cask_hip_pcg");
  if (t) t->toc("cg:solve");
  iterations = it;
  if (verbose) std::cout << " iterations " << iterations << " converged " << conv << " us/iteration " << us << "\n";
  return conv != 0;
}
}  // namespace

bool pcgIdentity(const CsrMatrix &a, double *rhs, double *x, int &iterations, bool verbose, cask::utils::Timer *t) {
  return run_pcg(a, nullptr, rhs, x, iterations, verbose, t);
}

bool pcgIlu(const CsrMatrix &a, double *rhs, double *x, int &iterations, bool verbose, cask::utils::Timer *t) {
  if (t) t->tic("cg:setup");
  ILUPreconditioner precon{a};
  if (t) t->toc("cg:setup");
  return run_pcg(a, static_cast<cask_hip_precond *>(precon.device.get()), rhs, x, iterations, verbose, t);
}

#ifdef CASK_HAVE_EIGEN
Eigen::VectorXd Solver::solve(const Eigen::SparseMatrix<double> &A, const Eigen::VectorXd &b) {
  Eigen::SparseMatrix<double, Eigen::RowMajor, int32_t> R(A);
  R.makeCompressed();
  CsrMatrix a(static_cast<int>(R.rows()), static_cast<int>(R.cols()), static_cast<int>(R.nonZeros()),
              std::vector<double>(R.valuePtr(), R.valuePtr() + R.nonZeros()),
              std::vector<int>(R.innerIndexPtr(), R.innerIndexPtr() + R.nonZeros()),
              std::vector<int>(R.outerIndexPtr(), R.outerIndexPtr() + R.rows() + 1));
  Vector rhs(std::vector<double>(b.data(), b.data() + b.size()));
  Vector x = solve(a, rhs);
  Eigen::VectorXd out(x.size());
  for (int i = 0; i < x.size(); i++) out[i] = x[i];
  return out;
}
#endif

}  // namespace sparse_linear_solvers

namespace dse {

std::vector<DseResult> SparkDse::run(const Benchmark &benchmark, const DseParameters &params,
                                     const cask::model::DeviceModel &deviceModel) {
  std::vector<DseResult> out;
  // lists for the engine, first list fastest (Utils.hpp:173-192)
  std::vector<int32_t> lanes, tiles, wgs(params.wgSize.begin(), params.wgSize.end()),
      items(params.itemsPerThread.begin(), params.itemsPerThread.end()),
      variants(params.variants.begin(), params.variants.end());
  for (int v = params.inputWidth.start; v <= params.inputWidth.end; v += std::max(1, params.inputWidth.step)) {
    const int l = std::min(64, pow2_floor(std::max(1, v)));
    if (lanes.empty() || lanes.back() != l) lanes.push_back(l);
  }
  if (params.alsoWithoutTile) tiles.push_back(-1);
  for (int v = params.cacheSize.start; v <= params.cacheSize.end; v += std::max(1, params.cacheSize.step)) tiles.push_back(v);

  for (int i = 0; i < benchmark.get_benchmark_size(); i++) {
    const std::string path = benchmark.get_matrix_path(i);
    const std::size_t slash = path.find_last_of("/");
    const std::string basename = slash == std::string::npos ? path : path.substr(slash + 1);
    std::cout << basename << std::endl;
    auto t0 = std::chrono::high_resolution_clock::now();
    CsrMatrix matrix = cask::io::readMatrix(path);
    std::cout << "Reading took: " << std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count()
              << std::endl;

    auto arch = std::make_shared<cask::spmv::Spmv>(params.cacheSize.start, params.inputWidth.start, 1, matrix.n, 1);
    arch->preprocess(matrix);
    std::vector<cask_hip_tune_point> pts(4096);
    int n = 0, best = -1;
    // the engine handle lives inside arch; tune through a temporary upload that shares nothing with it
    auto dev = upload(matrix, nullptr);
    check(cask_hip_tune(dev.get(), variants.data(), (int)variants.size(), lanes.data(), (int)lanes.size(), tiles.data(),
                        (int)tiles.size(), wgs.data(), (int)wgs.size(), items.data(), (int)items.size(), params.warmup,
                        params.iterations, pts.data(), (int)pts.size(), &n, &best),
          "cask_hip_tune");
    std::cout << "File Variant Lanes Tile WG Items usec_cold usec_warm GFLOPs GB/s" << std::endl;
    for (int k = 0; k < n; k++) {
      if (!pts[k].valid) continue;
      const cask_hip_params &q = pts[k].params;
      std::cout << basename << " " << q.variant << " " << q.lanes_per_row << " " << q.tile_width << " " << q.wg_size
                << " " << q.items_per_thread << " " << pts[k].usec << " " << pts[k].usec_warm << " " << pts[k].gflops
                << " " << pts[k].gbytes_per_s << std::endl;
    }
    if (best < 0) continue;
    arch->setDesignPoint(pts[best].params);
    DseResult r(path, arch);
    r.bestParams = pts[best].params;
    r.measuredGflops = pts[best].gflops;
    r.measuredMicroseconds = pts[best].usec;
    r.measuredMicrosecondsWarm = pts[best].usec_warm;
    r.copiesRotated = pts[best].copies;
    r.measuredGBs = pts[best].gbytes_per_s;
    r.pointsEvaluated = n;
    cask_hip_csr_info info;
    cask_hip_csr_get_info(dev.get(), &info);
    r.grid = info.grid;
    r.ldsBytes = info.lds_bytes;
    std::cout << "Matrix: " << basename << " Best architecture: variant " << r.bestParams.variant << " " << r.measuredGflops
              << " GFLOP/s " << 100.0 * r.measuredGBs / deviceModel.hbmPeakGBs() << "% of HBM peak" << std::endl;
    // like Dse.cpp:127-135: run the winner once
    try {
      cask::Vector lhs(matrix.m);
      auto result = arch->spmv(lhs);
    } catch (std::exception &e) {
      std::cout << "Could not run design " << e.what() << std::endl;
    }
    out.push_back(r);
  }
  return out;
}

void write_dse_results(const std::vector<DseResult> &results, double took, const cask::model::DeviceModel &deviceModel,
                       const std::string &path) {
  std::ofstream f(path);
  if (!f) throw std::invalid_argument("Could not open " + path + " for writing");
  std::time_t now = std::time(nullptr);
  std::string date = std::ctime(&now);
  while (!date.empty() && (date.back() == '\n' || date.back() == '\r')) date.pop_back();
  f << "{\n  \"date\": \"" << date << "\",\n  \"took\": " << took << ",\n  \"device\": \"" << deviceModel.getId()
    << "\",\n  \"best_architectures\": [\n";
  for (size_t i = 0; i < results.size(); i++) {
    const DseResult &r = results[i];
    const cask_hip_params &p = r.bestParams;
    f << "    {\n      \"name\": \"" << r.bestArchitecture->get_name() << "\",\n"
      << "      \"measured_gflops\": " << r.measuredGflops << ",\n"
      << "      \"measured_usec\": " << r.measuredMicroseconds << ",\n"
      << "      \"measured_usec_warm\": " << r.measuredMicrosecondsWarm << ",\n"
      << "      \"matrix_copies_rotated\": " << r.copiesRotated << ",\n"
      << "      \"measured_gbs_algorithmic\": " << r.measuredGBs << ",\n"
      << "      \"pct_hbm_peak\": " << 100.0 * r.measuredGBs / deviceModel.hbmPeakGBs() << ",\n"
      << "      \"architecture_params\": {\"variant\": " << p.variant << ", \"lanes_per_row\": " << p.lanes_per_row
      << ", \"tile_width\": " << p.tile_width << ", \"wg_size\": " << p.wg_size << ", \"items_per_thread\": "
      << p.items_per_thread << ", \"xcd_remap\": " << p.xcd_remap << ", \"nontemporal\": " << p.nontemporal
      << ", \"index16\": " << p.index16 << ", \"far_columns\": " << p.far_columns << "},\n"
      << "      \"launch\": {\"grid\": " << r.grid << ", \"lds_bytes\": " << r.ldsBytes << "},\n"
      << "      \"points_evaluated\": " << r.pointsEvaluated << ",\n      \"matrices\": [";
    for (size_t k = 0; k < r.matrices.size(); k++) f << (k ? ", " : "") << "\"" << r.matrices[k] << "\"";
    f << "]\n    }" << (i + 1 < results.size() ? "," : "") << "\n";
  }
  f << "  ]\n}\n";
}

}  // namespace dse
namespace mkl {

std::vector<double> unittrsolve(const CsrMatrix &m, const std::vector<double> &rhs, bool lowerTriangular) {
  if ((int)rhs.size() != m.n) throw std::invalid_argument("unittrsolve: right-hand side has the wrong length");
  std::vector<double> res(m.n);
  check(cask_hip_trsolve(m.n, m.nnzs, m.row_ptr.data(), m.col_ind.data(), m.values.data(), lowerTriangular ? 1 : 0,
                         rhs.data(), res.data()),
        "cask_hip_trsolve");
  return res;
}

void unittrsolve(const double *values, const int *row_ptr, const int *col_ind, const std::vector<double> &rhs,
                 double *res, bool lowerTriangular) {
  const int n = (int)rhs.size();
  std::vector<int> rp(row_ptr, row_ptr + n + 1);
  for (int &v : rp) v -= 1;                                   // the reference hands MKL 1-based arrays
  std::vector<int> ci(col_ind, col_ind + rp[n]);
  for (int &v : ci) v -= 1;
  check(cask_hip_trsolve(n, rp[n], rp.data(), ci.data(), values, lowerTriangular ? 1 : 0, rhs.data(), res),
        "cask_hip_trsolve");
}

}  // namespace mkl

}  // namespace cask

// ---- the product's MatrixMarket reader for callers without a C++ toolchain (bench.py: $CASK_MATRIX_DIR files) ----
// cask::io::readMatrix / readMatrixCached (include/cask/IO.hpp; the reference's io::readMatrix, IO.hpp:151-163) behind
// three C entry points: arrays are malloc'ed here and released with cask_host_free.
#include <cstring>

#include "cask/IO.hpp"

namespace {
thread_local std::string g_host_error;
}

extern "C" {

const char *cask_host_last_error(void) { return g_host_error.c_str(); }

void cask_host_free(void *p) { std::free(p); }

int cask_host_read_matrix(const char *path, int use_cache, int32_t *n_rows, int32_t *n_cols, int64_t *nnz, int32_t **row_ptr,
                          int32_t **col_ind, double **values) {
  if (!path || !n_rows || !n_cols || !nnz || !row_ptr || !col_ind || !values) {
    g_host_error = "NULL argument";
    return 1;
  }
  try {
    const cask::CsrMatrix a = use_cache ? cask::io::readMatrixCached(path) : cask::io::readMatrix(path);
    *n_rows = a.n;
    *n_cols = a.m;
    *nnz = a.nnzs;
    *row_ptr = static_cast<int32_t *>(std::malloc(sizeof(int32_t) * ((size_t)a.n + 1)));
    *col_ind = static_cast<int32_t *>(std::malloc(sizeof(int32_t) * std::max<size_t>(1, (size_t)a.nnzs)));
    *values = static_cast<double *>(std::malloc(sizeof(double) * std::max<size_t>(1, (size_t)a.nnzs)));
    if (!*row_ptr || !*col_ind || !*values) {
      std::free(*row_ptr); std::free(*col_ind); std::free(*values);
      g_host_error = "out of memory";
      return 1;
    }
    std::memcpy(*row_ptr, a.row_ptr.data(), sizeof(int32_t) * ((size_t)a.n + 1));
    if (a.nnzs) {
      std::memcpy(*col_ind, a.col_ind.data(), sizeof(int32_t) * (size_t)a.nnzs);
      std::memcpy(*values, a.values.data(), sizeof(double) * (size_t)a.nnzs);
    }
    return 0;
  } catch (const std::exception &e) {
    g_host_error = e.what();
    return 1;
  }
}

}  // extern "C"
