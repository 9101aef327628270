// The split predict kernels, float generation, the fused step, Matern-5/2 and -3/2: one translation unit per slice of leaf_split.hpp's
// instantiations, so that they compile in parallel.
#include <hip/hip_runtime.h>

#include "leaf_split.hpp"

namespace gpso {
bool leaf_step32_built() { return kLeafStep32; }
template int launch_leaf_tiles_bf16_v<float, true, 0>(hipStream_t, int, const void*, const float*, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, const void*, const float*, int64_t, const RawLeaves&);
}  // namespace gpso
