// The generated-library seam of the CASK surface: the device function triple,
// GeneratedSpmvImplementation and SpmvImplementationLoader, with the exact
// types of the reference's src/runtime/GeneratedImplSupport.hpp (:31-124) so
// that a libSpmv_<target>.so built against either header is interchangeable:
// the library's only C++ symbol is
//   cask::runtime::SpmvImplementationLoader::SpmvImplementationLoader()
// (_ZN4cask7runtime24SpmvImplementationLoaderC1Ev), whose body registers heap
// GeneratedSpmvImplementation objects (reference generator: src/frontend/cask.py:255-283;
// ours: tools/gen_impl.py -> libSpmv_hip.so).
//
// For target "hip" the triple is either the no-op mocks (the Spmv class talks
// to the engine through include/cask_hip.h directly) or the DFE-compatible
// functions of include/cask_hip_dfe.h, which execute the reference's own LMem
// stream format on the GPU behind the unchanged SLiC signatures.
#ifndef CASK_GENERATEDIMPLSUPPORT_HPP
#define CASK_GENERATEDIMPLSUPPORT_HPP

#include <climits>
#include <cstdint>
#include <cstddef>
#include <functional>
#include <string>
#include <vector>

namespace cask {
namespace runtime {

// SLiC-shaped signatures: Spmv_<id>_dramRead / Spmv_<id> / Spmv_<id>_dramWrite
// (reference: GeneratedImplSupport.hpp:31-49, SpmvManager.java:336-432).
inline void spmvReadMock(const int64_t, const int64_t *, const int64_t *, uint8_t *, const char *) {}

inline void spmvRunMock(int64_t, int64_t, int64_t, const int64_t *, const int32_t *, const int64_t *,
                        const int32_t *, const int32_t *, const int64_t *, const int32_t *, const int32_t *,
                        const int64_t *) {}

inline void spmvWriteMock(const int64_t, const int64_t *, const int64_t *, const uint8_t *, const char *) {}

class GeneratedSpmvImplementation {
  using SpmvFunctionT = decltype(spmvRunMock);
  using SpmvDramWriteFunctionT = decltype(spmvWriteMock);
  using SpmvDramReadFunctionT = decltype(spmvReadMock);

 public:
  const int id, max_rows, num_pipes, cache_size, input_width, dram_reduction_enabled, num_controllers;
  std::function<SpmvFunctionT> Spmv;
  std::function<SpmvDramWriteFunctionT> write;
  std::function<SpmvDramReadFunctionT> read;

  GeneratedSpmvImplementation(int _id, SpmvFunctionT _run, SpmvDramWriteFunctionT _write,
                              SpmvDramReadFunctionT _read, int _max_rows, int _num_pipes, int _cache_size,
                              int _input_width, int _dram_reduction_enabled, int _num_controllers)
      : id(_id), max_rows(_max_rows), num_pipes(_num_pipes), cache_size(_cache_size), input_width(_input_width),
        dram_reduction_enabled(_dram_reduction_enabled), num_controllers(_num_controllers), Spmv(_run),
        write(_write), read(_read) {}

  // two implementations are the same architecture when every parameter but the id agrees
  bool operator==(const GeneratedSpmvImplementation &o) const {
    const int mine[] = {max_rows, num_pipes, cache_size, input_width, dram_reduction_enabled, num_controllers};
    const int theirs[] = {o.max_rows, o.num_pipes, o.cache_size, o.input_width, o.dram_reduction_enabled,
                          o.num_controllers};
    for (int k = 0; k < 6; k++)
      if (mine[k] != theirs[k]) return false;
    return true;
  }

  // "id=0 max_rows=... pipes=... cache=... width=... controllers=..." for logs
  std::string describe() const {
    return "id=" + std::to_string(id) + " max_rows=" + std::to_string(max_rows) + " pipes=" + std::to_string(num_pipes) +
           " cache=" + std::to_string(cache_size) + " width=" + std::to_string(input_width) +
           " controllers=" + std::to_string(num_controllers);
  }

  // whether a matrix with `rows` rows fits (the only capability test the reference applies, Spmv.cpp:201-207)
  bool supportsRows(int rows) const { return max_rows >= rows; }
};

class SpmvImplementationLoader {
  std::vector<GeneratedSpmvImplementation *> impls;

 public:
  SpmvImplementationLoader();   // defined in the generated library

  // The registered implementation with the smallest max_rows that still holds `maxRows` rows;
  // nullptr when none does (a generated library should always register a catch-all).
  GeneratedSpmvImplementation *architectureWithParams(int maxRows) {
    GeneratedSpmvImplementation *tightest = nullptr;
    for (std::size_t k = 0; k < impls.size(); k++) {
      GeneratedSpmvImplementation *candidate = impls[k];
      if (!candidate->supportsRows(maxRows)) continue;
      if (tightest == nullptr || candidate->max_rows < tightest->max_rows) tightest = candidate;
    }
    return tightest;
  }

  GeneratedSpmvImplementation *architectureWithId(int id) { return impls.at(static_cast<std::size_t>(id)); }

  // number of registered implementations (new; the reference exposes none)
  int size() const { return static_cast<int>(impls.size()); }
};

}  // namespace runtime
}  // namespace cask

#endif  // CASK_GENERATEDIMPLSUPPORT_HPP
