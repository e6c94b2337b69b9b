// Diagnostic builds of the merge kernel (make build/libcask_hip_diag<N>.so, -DCASK_DIAG=<N>; never shipped, never
// loaded by tests or bench unless CASK_HIP_DIAGNOSTIC_LIB names one).  Each bit removes ONE cost from the kernel so
// that an interleaved A/B against the product library prices it (tools/ab_lib.sh); results of such a build are wrong
// by design.  What they measured: profiles/r02_solver_modes.txt (bits 1, 2, 4), profiles/r04_run_records.txt
// (16, 32, 64).  In the product build every switch is false and the code it guards folds away.
#pragma once

#ifndef CASK_DIAG
#define CASK_DIAG 0
#endif

namespace caskhip {
namespace diag {
constexpr bool NO_PARTIALS = (CASK_DIAG & 1) != 0;        // solver pass: made-up scalars instead of summing the partial sums
constexpr bool NO_SECOND_WINDOW = (CASK_DIAG & 2) != 0;   // solver pass: operand not composed (no second x window)
constexpr bool NO_OWN_ROWS = (CASK_DIAG & 4) != 0;        // solver pass: no own-row updates
constexpr bool FREE_SLOTS = (CASK_DIAG & 16) != 0;        // every block reads block 0's slot records (an L2 hit)
constexpr bool FREE_VALUES = (CASK_DIAG & 32) != 0;       // every block streams block 0's values (L2 hits)
constexpr bool NO_TAIL = (CASK_DIAG & 64) != 0;           // the workgroup returns once its products are parked
}  // namespace diag
}  // namespace caskhip
