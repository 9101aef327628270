// The split predict kernels, double generation, the fused step: one translation unit per slice of leaf_split.hpp's
// instantiations, so that they compile in parallel.
#include <hip/hip_runtime.h>

#include "leaf_split.hpp"

namespace gpso {
template int launch_leaf_tiles_bf16_v<double, true, 0>(hipStream_t, int, const void*, const double*, const double*, const float*, const double*, const double*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, const void*, const float*, int64_t, const RawLeaves&);
template int launch_leaf_tiles_bf16_v<double, true, 1>(hipStream_t, int, const void*, const double*, const double*, const float*, const double*, const double*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, const void*, const float*, int64_t, const RawLeaves&);
}  // namespace gpso
