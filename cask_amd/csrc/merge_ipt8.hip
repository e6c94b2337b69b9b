// k_spmv_merge instantiations for items_per_thread = 8 (see merge_launch.hpp).
#include "merge_launch_impl.hpp"

namespace caskhip {
template void launch_merge_blocks<8>(const MergeLaunch &, const double *, double *, hipStream_t);

}
