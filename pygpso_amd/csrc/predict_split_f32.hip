// The split predict kernels for float generation: one translation unit per generation type, so that the two halves
// of leaf_split.hpp's instantiations compile in parallel.
#include <hip/hip_runtime.h>

#include "leaf_split.hpp"

namespace gpso {
template int launch_leaf_tiles_bf16<float>(hipStream_t, int, const void*, const float*, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, int, const void*, const float*, int64_t);
}  // namespace gpso
