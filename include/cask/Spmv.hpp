// cask::spmv::Spmv -- the SpMV operator of the CASK surface, MI355X backend.
//
// Same public surface as the reference's src/runtime/Spmv.hpp (:49-201):
// construct from a GeneratedSpmvImplementation (or from the five architecture
// integers, for DSE), preprocess(const CsrMatrix&), Vector spmv(const Vector&),
// public member impl, virtual get_name / getEstimatedClockCycles /
// getGFlopsCount / isValid.  What is behind it is new: preprocess() uploads the
// matrix ONCE to HBM and builds a launch plan through the C ABI
// (include/cask_hip.h); spmv() moves x up, runs one HIP kernel, moves y down.
// The reference's FPGA partition/blocking/cycle model (Spmv.cpp:25-107) has no
// counterpart: design points are measured, not modelled.
//
// Parameter mapping: impl.cache_size -> x tile width in LDS (doubles),
// impl.input_width -> lanes per row, impl.num_pipes / num_controllers ->
// nothing (the grid is derived; GPUs are processes), impl.max_rows -> no limit.
// A generated library may refine the point per implementation id through
// cask_hip_generated_design_point() (tools/gen_impl.py writes it from dse_out.json).
#ifndef CASK_SPMV_HPP
#define CASK_SPMV_HPP

#include <memory>
#include <sstream>
#include <string>

#include "GeneratedImplSupport.hpp"
#include "Model.hpp"
#include "SparseMatrix.hpp"
#include "Utils.hpp"
#include "cask_hip.h"

namespace cask {
namespace spmv {

class Spmv {
  CsrMatrix mat;
  std::shared_ptr<cask_hip_matrix> device;     // device-resident CSR + plan; shared by copies of this object
  double lastSeconds = 0;                      // wall time of the last device run

 public:
  runtime::GeneratedSpmvImplementation impl;

  // DSE-style construction from architecture integers (reference: Spmv.hpp:56-66)
  Spmv(int _cacheSize, int _inputWidth, int _numPipes, int _maxRows, int _numControllers)
      : impl(-1, cask::runtime::spmvRunMock, cask::runtime::spmvWriteMock, cask::runtime::spmvReadMock, _maxRows,
             _numPipes, _cacheSize, _inputWidth, false, _numControllers) {}

  Spmv(runtime::GeneratedSpmvImplementation _impl) : impl(_impl) {}
  virtual ~Spmv() {}

  // Upload the matrix and plan the launch.  Replaces Spmv::preprocess (Spmv.cpp:329-365).
  void preprocess(const CsrMatrix &mat);

  // y = A x on the GPU.  Throws std::invalid_argument on a size mismatch and
  // std::runtime_error when preprocess() has not run or the device fails
  // (reference error behaviour: Spmv.cpp:189-232).  There is no CPU fallback.
  Vector spmv(const Vector &v);

  // ---- design point ------------------------------------------------------------
  cask_hip_params designPoint() const;                 // resolved point of the active plan
  void setDesignPoint(const cask_hip_params &p);       // re-plan (matrix stays resident)
  // measured DSE over the engine's default ranges; returns the best GFLOP/s and leaves it active
  double tune(int warmup = 3, int iters = 20);
  // median device microseconds per SpMV (device-resident vectors)
  double measureMicroseconds(int warmup = 5, int iters = 50);

  // ---- reference API kept for clients and the DSE ------------------------------------
  virtual std::string get_name() { return std::string("Simple"); }
  bool operator==(const Spmv &other) const { return impl == other.impl; }
  virtual bool isValid() { return impl.num_pipes >= impl.num_controllers; }
  virtual double getGFlopsCount() { return 2 * this->mat.nnzs / 1E9; }
  double getFrequency();                               // shader clock in Hz
  // measured time expressed in shader cycles, so that the reference's formula
  // GFlopsCount * frequency / cycles (Spmv.hpp:80-82) stays meaningful
  virtual double getEstimatedClockCycles();
  double getEstimatedGFlops(const model::DeviceModel &deviceModel);
  model::HardwareModel getEstimatedHardwareModel(const model::DeviceModel &deviceModel);
  model::HardwareModel getEstimatedHardwareModel(const model::DeviceModel &deviceModel, const int matrixDimension);

  std::string to_string(const model::DeviceModel &deviceModel) {
    std::stringstream s;
    s << get_name() << " " << impl.cache_size << " " << impl.input_width << " " << impl.num_pipes << " "
      << impl.num_controllers << " " << getEstimatedClockCycles() << " " << getEstimatedGFlops(deviceModel);
    return s.str();
  }
};

inline std::ostream &operator<<(std::ostream &s, Spmv &a) {
  s << " Estimated clock cycles = " << a.getEstimatedClockCycles();
  return s;
}

// The reference models an FPGA variant that run-length-skips empty rows
// (Spmv.hpp:211-259).  On the GPU empty rows cost a row_ptr compare in every
// variant; the class is kept so DSE code that names it still compiles.
class SkipEmptyRowsSpmv : public Spmv {
 public:
  SkipEmptyRowsSpmv(int _cacheSize, int _inputWidth, int _numPipes, int _maxRows, int _numControllers)
      : Spmv(_cacheSize, _inputWidth, _numPipes, _maxRows, _numControllers) {}
  SkipEmptyRowsSpmv(runtime::GeneratedSpmvImplementation _impl) : Spmv(_impl) {}
  std::string get_name() override { return std::string("SkipEmpty"); }
};

}  // namespace spmv
}  // namespace cask

#endif  // CASK_SPMV_HPP
