// The split predict kernels, float generation, round 3's two-phase step (GPSO_SPLIT_KERNEL_TWO_PHASE): one translation unit per slice of leaf_split.hpp's
// instantiations, so that they compile in parallel.
#include <hip/hip_runtime.h>

#include "leaf_split.hpp"

namespace gpso {
template int launch_leaf_tiles_bf16_v<float, false, 0>(hipStream_t, int, const void*, const float*, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, const void*, const float*, int64_t, const RawLeaves&);
template int launch_leaf_tiles_bf16_v<float, false, 1>(hipStream_t, int, const void*, const float*, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, const void*, const float*, int64_t, const RawLeaves&);
}  // namespace gpso
