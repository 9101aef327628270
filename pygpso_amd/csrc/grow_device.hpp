// Device-side ternary geometry shared by grow.hip and the one-launch small predict kernel (predict.hip): the reference's
// float64 operation order (gpso/param_space.py:175-200, 257-307).  Every translation unit that includes this MUST be
// compiled with -ffp-contract=off: a fused multiply-add in `lo + i * delta` would change the last bit.
#pragma once
#include "common.hpp"

namespace gpso {

// Per-thread box state lives in LDS as [dimension][thread] (conflict-free, no scratch): kGrowThreads threads per block
// in grow.hip; the box of thread t is (lo, hi)[k * STRIDE + t].
constexpr int kGrowThreads = 64;

// level j and position p of a row: rows of level j start at (3^j - 1) / 2
__device__ __forceinline__ void grow_locate(int64_t row, int& level, int64_t& width, int64_t& p) {
  level = 0;
  int64_t start = 0;
  width = 1;  // 3^level
  while (start + width <= row) {
    start += width;
    width *= 3;
    ++level;
  }
  p = row - start;
}

// one ternary split of the box held in (lo, hi)[k * STRIDE + t]: child 0 = l, 1 = c, 2 = r
template <int STRIDE = kGrowThreads>
__device__ __forceinline__ void grow_split(double* lo, double* hi, int t, int d, int child) {
  int kmax = 0;
  double wmax = hi[t] - lo[t];
  for (int k = 1; k < d; ++k) {
    const double w = hi[k * STRIDE + t] - lo[k * STRIDE + t];
    if (w > wmax) {  // first maximum wins, as np.argmax
      wmax = w;
      kmax = k;
    }
  }
  const double delta = wmax / 3;
  const double base = lo[kmax * STRIDE + t];
  const double c0 = base + (double)child * delta;
  const double c1 = base + (double)(child + 1) * delta;
  lo[kmax * STRIDE + t] = c0;
  hi[kmax * STRIDE + t] = c1;
}

// the dimension a split of the box would cut (first widest), and the box's centre in that dimension
template <int STRIDE = kGrowThreads>
__device__ __forceinline__ double grow_split_centre(const double* lo, const double* hi, int t, int d, int* kmax_out) {
  int kmax = 0;
  double wmax = hi[t] - lo[t];
  for (int k = 1; k < d; ++k) {
    const double w = hi[k * STRIDE + t] - lo[k * STRIDE + t];
    if (w > wmax) {
      wmax = w;
      kmax = k;
    }
  }
  *kmax_out = kmax;
  return (lo[kmax * STRIDE + t] + hi[kmax * STRIDE + t]) / 2;
}

}  // namespace gpso
