// The split predict kernels, float generation, the fused step, Matern-1/2 and squared exponential: one translation unit per slice of leaf_split.hpp's
// instantiations, so that they compile in parallel.
#include <hip/hip_runtime.h>

#include "leaf_split.hpp"

namespace gpso {
template int launch_leaf_tiles_bf16_v<float, true, 1>(hipStream_t, int, const void*, const float*, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, const void*, const float*, int64_t, const RawLeaves&);
}  // namespace gpso
