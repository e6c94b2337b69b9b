// Design-space exploration of the CASK surface, measured on the GPU.
// Counterpart of the reference's src/runtime/Dse.hpp (:14-88) and Dse.cpp
// (:12-140): Benchmark (a list of matrix paths), DseParameters (the ranges),
// DseResult (winner per matrix) and SparkDse::run.  The reference scores a
// point with an FPGA cycle/resource model; here SparkDse::run asks the engine
// (cask_hip_tune) to time every point of the cross product on the device.
#ifndef CASK_DSE_HPP
#define CASK_DSE_HPP

#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "IO.hpp"
#include "Model.hpp"
#include "Spmv.hpp"
#include "Utils.hpp"

namespace cask {
namespace dse {

// The matrices a DSE run covers: an ordered list of MatrixMarket paths.
class Benchmark {
 public:
  void add_matrix_path(std::string path) { files_.push_back(std::move(path)); }
  int get_benchmark_size() const { return static_cast<int>(files_.size()); }
  bool empty() const { return files_.empty(); }

  std::string get_matrix_path(int id) const {
    if (id < 0 || id >= get_benchmark_size())
      throw std::invalid_argument("Benchmark::Index out of range " + std::to_string(id));
    return files_[static_cast<std::size_t>(id)];
  }

  void describe(std::ostream &out) const {
    out << "Benchmark(" << std::endl;
    for (const std::string &f : files_) out << "  " << f << std::endl;
    out << ")" << std::endl;
  }

 private:
  std::vector<std::string> files_;
};

inline std::ostream &operator<<(std::ostream &s, Benchmark &b) {
  b.describe(s);
  return s;
}

// Ranges to sweep.  inputWidth / cacheSize keep the reference's names and
// {start, end, step} form (src/frontend/params.json); they mean lanes per row
// (rounded down to a power of two) and x tile width in doubles.  numPipes and
// numControllers are accepted and ignored (the grid is derived; GPUs are
// processes).  wgSize / itemsPerThread / variants are the GPU-only knobs.
class DseParameters {
 public:
  bool gflopsOnly = true;
  cask::utils::Parameter<> numPipes{"numPipes", 1, 1, 1};
  cask::utils::Parameter<> inputWidth{"inputWidth", 4, 32, 4};
  cask::utils::Parameter<> cacheSize{"cacheSize", 1024, 4096, 1024};
  cask::utils::Parameter<> numControllers{"numControllers", 1, 1, 1};
  std::vector<int> wgSize{256, 512};
  std::vector<int> itemsPerThread{4, 8};
  std::vector<int> variants{CASK_HIP_VARIANT_VECTOR, CASK_HIP_VARIANT_MERGE, CASK_HIP_VARIANT_MERGE_WAVE, CASK_HIP_VARIANT_SCAN,
                            CASK_HIP_VARIANT_SLICE};
  bool alsoWithoutTile = true;       // add tile_width = -1 (x from L2) to the cacheSize range
  int warmup = 3, iterations = 20;
};

inline std::ostream &operator<<(std::ostream &s, DseParameters &d) {
  s << "DseParams(" << std::endl;
  s << "  inputWidth (lanes per row) = " << d.inputWidth << std::endl;
  s << "  cacheSize (x tile, doubles) = " << d.cacheSize << std::endl;
  s << "  wgSize = ";
  for (int v : d.wgSize) s << v << " ";
  s << std::endl << "  itemsPerThread = ";
  for (int v : d.itemsPerThread) s << v << " ";
  s << std::endl << ")" << std::endl;
  return s;
}

class DseResult {
 public:
  std::shared_ptr<cask::spmv::Spmv> bestArchitecture;
  std::vector<std::string> matrices;
  cask_hip_params bestParams{};
  double measuredGflops = 0, measuredMicroseconds = 0, measuredGBs = 0;   // cold (rotating device copies)
  double measuredMicrosecondsWarm = 0;                                    // one copy replayed (Infinity-Cache resident)
  int copiesRotated = 1;
  int pointsEvaluated = 0;
  int grid = 0, ldsBytes = 0;

  DseResult(std::string path, std::shared_ptr<cask::spmv::Spmv> arch) : bestArchitecture(arch) {
    matrices.push_back(path);
  }
  DseResult(std::shared_ptr<cask::spmv::Spmv> arch) : bestArchitecture(arch) {}
};

class SparkDse {
 public:
  SparkDse() {}
  // one DseResult (the measured best design point) per matrix of the benchmark
  std::vector<DseResult> run(const Benchmark &benchmark, const DseParameters &dseParams,
                             const cask::model::DeviceModel &deviceModel);
};

// dse_out.json, the layout of the reference's writer (src/main.cpp:81-117) with
// measured_* in place of estimated_* fields.
void write_dse_results(const std::vector<DseResult> &results, double took, const cask::model::DeviceModel &deviceModel,
                       const std::string &path = "dse_out.json");

}  // namespace dse
}  // namespace cask

#endif  // CASK_DSE_HPP
