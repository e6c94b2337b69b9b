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
#include <functional>
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

  bool operator==(const GeneratedSpmvImplementation &o) const {
    return max_rows == o.max_rows && num_pipes == o.num_pipes && cache_size == o.cache_size &&
           input_width == o.input_width && dram_reduction_enabled == o.dram_reduction_enabled &&
           num_controllers == o.num_controllers;
  }
};

class SpmvImplementationLoader {
  std::vector<GeneratedSpmvImplementation *> impls;

 public:
  SpmvImplementationLoader();   // defined in the generated library

  // smallest registered max_rows that still holds `maxRows` rows; nullptr if none
  GeneratedSpmvImplementation *architectureWithParams(int maxRows) {
    GeneratedSpmvImplementation *best = nullptr;
    for (GeneratedSpmvImplementation *a : impls)
      if (a->max_rows >= maxRows && (!best || a->max_rows < best->max_rows)) best = a;
    return best;
  }

  GeneratedSpmvImplementation *architectureWithId(int id) { return impls.at(id); }
};

}  // namespace runtime
}  // namespace cask

#endif  // CASK_GENERATEDIMPLSUPPORT_HPP
