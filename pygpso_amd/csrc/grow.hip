// On-device ternary leaf-centre generator (K12 of SURVEY.md 2.4).
//
// Replays LeafNode.grow / ternary_split (gpso/param_space.py:175-200, 257-307) in float64 with the
// SAME operation order, so rows are bit-identical to the reference's Python floats:
//     widths  w_d = hi_d - lo_d ;  k = first arg-max_d w_d ;  delta = w_k / 3
//     cuts    lo_k + i * delta, i = 0..3 ;  children l, c, r take consecutive cut pairs
//     centre  (lo_d + hi_d) / 2
// Row order is the reference's: level-major, level j obtained by expanding every node of level
// j-1, in order, into (l, c, r); levels 0..depth-1 -> (3^depth - 1) / 2 rows per box.
//
// This translation unit MUST be compiled with -ffp-contract=off (see Makefile): a fused
// multiply-add in `lo + i * delta` would change the last bit and with it arg-max ties downstream.
#include "common.hpp"
#include "kernels.hpp"

namespace gpso {

constexpr int kGrowMaxD = 64;

__global__ __launch_bounds__(128) void grow_kernel(const double* __restrict__ bounds, int d,
                                                   int depth, int64_t rows,
                                                   double* __restrict__ out) {
  const int seg = blockIdx.y;
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  // level j and position p of this row: rows of level j start at (3^j - 1) / 2
  int level = 0;
  int64_t start = 0, width = 1;  // width = 3^level
  while (start + width <= row) {
    start += width;
    width *= 3;
    ++level;
  }
  int64_t p = row - start;
  double lo[kGrowMaxD], hi[kGrowMaxD];
  const double* b = bounds + (int64_t)seg * d * 2;
  for (int k = 0; k < d; ++k) {
    lo[k] = b[2 * k];
    hi[k] = b[2 * k + 1];
  }
  int64_t div = width;  // 3^level
  for (int t = 0; t < level; ++t) {
    div /= 3;
    const int child = (int)((p / div) % 3);  // 0 = l, 1 = c, 2 = r
    int kmax = 0;
    double wmax = hi[0] - lo[0];
    for (int k = 1; k < d; ++k) {
      const double w = hi[k] - lo[k];
      if (w > wmax) {  // first maximum wins, as np.argmax
        wmax = w;
        kmax = k;
      }
    }
    const double delta = wmax / 3;
    const double base = lo[kmax];
    const double c0 = base + (double)child * delta;
    const double c1 = base + (double)(child + 1) * delta;
    lo[kmax] = c0;
    hi[kmax] = c1;
  }
  double* o = out + ((int64_t)seg * rows + row) * d;
  for (int k = 0; k < d; ++k) o[k] = (lo[k] + hi[k]) / 2;
}

void launch_grow(hipStream_t st, const double* bounds_dev, int nseg, int d, int depth,
                 double* out_dev) {
  int64_t rows = 0, w = 1;
  for (int j = 0; j < depth; ++j) {
    rows += w;
    w *= 3;
  }
  if (rows == 0 || nseg == 0) return;
  const dim3 grid((unsigned)((rows + 127) / 128), (unsigned)nseg);
  hipLaunchKernelGGL(grow_kernel, grid, dim3(128), 0, st, bounds_dev, d, depth, rows, out_dev);
}

}  // namespace gpso
