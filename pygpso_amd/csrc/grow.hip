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
#include "grow_device.hpp"
#include "kernels.hpp"

namespace gpso {

// Per-thread box state lives in LDS as [dimension][thread] (conflict-free, no scratch: a per-thread
// double lo[64], hi[64] array spills): one wave per block, 2 * d * 64 doubles of dynamic LDS (grow_device.hpp).

__global__ __launch_bounds__(kGrowThreads) void grow_kernel(const double* __restrict__ bounds, int d,
                                                            int depth, int64_t rows,
                                                            double* __restrict__ out) {
  extern __shared__ double grow_lds[];
  double* lo = grow_lds;
  double* hi = grow_lds + (size_t)d * kGrowThreads;
  const int seg = blockIdx.y, t = threadIdx.x;
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + t;
  if (row >= rows) return;
  int level;
  int64_t width, p;
  grow_locate(row, level, width, p);
  const double* b = bounds + (int64_t)seg * d * 2;
  for (int k = 0; k < d; ++k) {
    lo[k * kGrowThreads + t] = b[2 * k];
    hi[k * kGrowThreads + t] = b[2 * k + 1];
  }
  int64_t div = width;  // 3^level
  for (int s = 0; s < level; ++s) {
    div /= 3;
    grow_split(lo, hi, t, d, (int)((p / div) % 3));
  }
  double* o = out + ((int64_t)seg * rows + row) * d;
  for (int k = 0; k < d; ++k) o[k] = (lo[k * kGrowThreads + t] + hi[k * kGrowThreads + t]) / 2;
}

// De-duplicated growth for the scoring path.  A centre child's centre is its parent's centre
// (gpso/param_space.py:186-200 emits both), so only 3^(depth-1) of the (3^depth - 1) / 2 rows of a box
// are distinct points: the root and, on every level j >= 1, the 2 * 3^(j-1) l / r children -- which
// get the closed-form compact slots [3^(j-1), 3^j) of their box, slot = 3^(j-1) + 2 * (p / 3) + (p % 3 == 2).
// "Is" holds in exact arithmetic; in float64 the two centres can differ in the last bit (they never do
// for boxes cut from the unit cube, they do for about half of arbitrary boxes).  A centre child whose
// centre is NOT bit-identical to its parent's is a different input to the predict kernels and is kept:
// it is appended behind the nseg * U analytic slots through an atomic counter (*count starts at
// nseg * U and ends as the number of live rows).  key[slot] = seg * rows + reference row index: the
// arg-max runs on (ucb, key), so neither the compaction nor the order of the appended rows can change
// the winner or its index -- a dropped row has the same bits, hence the same ucb, as an EARLIER row.
// Multi-GPU: a rank generates the reference rows [row_lo, row_hi) of every box only; uniq is then the
// number of analytic slots inside that range and slot_base the analytic slot of its first row
// (grow_unique_before), so the compact list of a rank is dense.  Keys stay global reference indices.
__global__ __launch_bounds__(kGrowThreads) void grow_unique_kernel(const double* __restrict__ bounds, int d,
                                                                   int depth, int64_t rows, int64_t row_lo,
                                                                   int64_t row_hi, int64_t uniq,
                                                                   int64_t slot_base, double* __restrict__ out,
                                                                   int64_t* __restrict__ key,
                                                                   unsigned long long* __restrict__ count) {
  extern __shared__ double grow_lds[];
  double* lo = grow_lds;
  double* hi = grow_lds + (size_t)d * kGrowThreads;
  const int seg = blockIdx.y, t = threadIdx.x;
  const int64_t row = row_lo + (int64_t)blockIdx.x * blockDim.x + t;
  if (row >= row_hi) return;
  int level;
  int64_t width, p;
  grow_locate(row, level, width, p);
  const double* b = bounds + (int64_t)seg * d * 2;
  for (int k = 0; k < d; ++k) {
    lo[k * kGrowThreads + t] = b[2 * k];
    hi[k * kGrowThreads + t] = b[2 * k + 1];
  }
  int64_t div = width;
  for (int s = 0; s + 1 < level; ++s) {
    div /= 3;
    grow_split(lo, hi, t, d, (int)((p / div) % 3));
  }
  const int child = (level > 0) ? (int)(p % 3) : 0;
  int64_t slot;
  if (level == 0) {
    slot = (int64_t)seg * uniq - slot_base;
  } else if (child != 1) {
    grow_split(lo, hi, t, d, child);
    slot = (int64_t)seg * uniq + (width / 3 + 2 * (p / 3) + (child == 2 ? 1 : 0) - slot_base);
  } else {
    // centre child: compare its centre with the parent's, bit for bit, in the only dimension the split
    // touches (the others hold the same lo / hi)
    int kmax = 0;
    double wmax = hi[t] - lo[t];
    for (int k = 1; k < d; ++k) {
      const double w = hi[k * kGrowThreads + t] - lo[k * kGrowThreads + t];
      if (w > wmax) {
        wmax = w;
        kmax = k;
      }
    }
    const double parent_c = (lo[kmax * kGrowThreads + t] + hi[kmax * kGrowThreads + t]) / 2;
    grow_split(lo, hi, t, d, 1);
    const double child_c = (lo[kmax * kGrowThreads + t] + hi[kmax * kGrowThreads + t]) / 2;
    if (__builtin_bit_cast(long long, parent_c) == __builtin_bit_cast(long long, child_c)) return;
    slot = (int64_t)atomicAdd(count, 1ull);
  }
  key[slot] = (int64_t)seg * rows + row;
  double* o = out + slot * d;
  for (int k = 0; k < d; ++k) o[k] = (lo[k * kGrowThreads + t] + hi[k * kGrowThreads + t]) / 2;
}

// ---- small calls (round 4): growth AND input scaling in one launch, the boxes passed BY VALUE ---------------------
// The optimiser's exploration levels score a few hundred to a few thousand rows (gpso/optimisation.py:366-382): the
// call is launch-bound -- two host-to-device copies (boxes, row count), grow, prep, tiles, two arg-max stages and a
// copy back were ~100 us for 162 rows at N = 52.  Here the boxes travel as kernel arguments, every thread scales its
// own row (prep_leaves_kernel's arithmetic, element for element) and stores it where the tile kernel reads; rows that
// do not repeat their parent bit for bit are appended at base + atomicAdd(extra) -- `extra` is zero between calls (the
// last kernel of a call resets it).  Rows at or beyond the live count hold whatever the buffer held: the tile kernel
// computes them, nothing reads them.
template <typename TG>
__global__ __launch_bounds__(kGrowThreads) void grow_unique_prep_kernel(GrowBoxes boxes, int d, int dp, int depth,
                                                                        int64_t rows, int64_t row_lo, int64_t row_hi,
                                                                        int64_t uniq, int64_t slot_base, int64_t base,
                                                                        const double* __restrict__ ls,
                                                                        TG* __restrict__ out_s, TG* __restrict__ norm,
                                                                        int64_t* __restrict__ key,
                                                                        unsigned long long* __restrict__ extra) {
  extern __shared__ double grow_lds[];
  double* lo = grow_lds;
  double* hi = grow_lds + (size_t)d * kGrowThreads;
  const int seg = blockIdx.y, t = threadIdx.x;
  const int64_t row = row_lo + (int64_t)blockIdx.x * blockDim.x + t;
  if (row >= row_hi) return;
  int level;
  int64_t width, p;
  grow_locate(row, level, width, p);
  const double* b = boxes.b + (int64_t)seg * d * 2;
  for (int k = 0; k < d; ++k) {
    lo[k * kGrowThreads + t] = b[2 * k];
    hi[k * kGrowThreads + t] = b[2 * k + 1];
  }
  int64_t div = width;
  for (int s = 0; s + 1 < level; ++s) {
    div /= 3;
    grow_split(lo, hi, t, d, (int)((p / div) % 3));
  }
  const int child = (level > 0) ? (int)(p % 3) : 0;
  int64_t slot;
  if (level == 0) {
    slot = (int64_t)seg * uniq - slot_base;
  } else if (child != 1) {
    grow_split(lo, hi, t, d, child);
    slot = (int64_t)seg * uniq + (width / 3 + 2 * (p / 3) + (child == 2 ? 1 : 0) - slot_base);
  } else {
    int kmax = 0;
    double wmax = hi[t] - lo[t];
    for (int k = 1; k < d; ++k) {
      const double w = hi[k * kGrowThreads + t] - lo[k * kGrowThreads + t];
      if (w > wmax) {
        wmax = w;
        kmax = k;
      }
    }
    const double parent_c = (lo[kmax * kGrowThreads + t] + hi[kmax * kGrowThreads + t]) / 2;
    grow_split(lo, hi, t, d, 1);
    const double child_c = (lo[kmax * kGrowThreads + t] + hi[kmax * kGrowThreads + t]) / 2;
    if (__builtin_bit_cast(long long, parent_c) == __builtin_bit_cast(long long, child_c)) return;
    slot = base + (int64_t)atomicAdd(extra, 1ull);
  }
  key[slot] = (int64_t)seg * rows + row;
  TG acc = 0;
  for (int k = 0; k < dp; ++k) {
    TG v = 0;
    if (k < d) v = (TG)(((lo[k * kGrowThreads + t] + hi[k * kGrowThreads + t]) / 2) / ls[k]);
    out_s[slot * dp + k] = v;
    acc += v * v;
  }
  norm[slot] = acc;
}

static int64_t grow_rows_of(int depth) {
  int64_t rows = 0, w = 1;
  for (int j = 0; j < depth; ++j) {
    rows += w;
    w *= 3;
  }
  return rows;
}

void launch_grow(hipStream_t st, const double* bounds_dev, int nseg, int d, int depth,
                 double* out_dev) {
  const int64_t rows = grow_rows_of(depth);
  if (rows == 0 || nseg == 0) return;
  const dim3 grid((unsigned)((rows + kGrowThreads - 1) / kGrowThreads), (unsigned)nseg);
  hipLaunchKernelGGL(grow_kernel, grid, dim3(kGrowThreads), (size_t)2 * d * kGrowThreads * 8, st, bounds_dev, d,
                     depth, rows, out_dev);
}

int64_t grow_unique_rows(int depth) {
  int64_t u = (depth >= 1) ? 1 : 0;
  for (int j = 1; j < depth; ++j) u *= 3;
  return u;
}

// number of analytic slots (root + l / r children) among the reference rows [0, row)
int64_t grow_unique_before(int64_t row) {
  if (row <= 0) return 0;
  int64_t start = 0, width = 1;  // width = 3^level of the level that holds `row`
  while (start + width <= row) {
    start += width;
    width *= 3;
  }
  const int64_t p = row - start;  // position inside that level (level >= 1 here)
  return width / 3 + 2 * (p / 3) + (p % 3 >= 1 ? 1 : 0);
}

void launch_grow_unique(hipStream_t st, const double* bounds_dev, int nseg, int d, int depth, int64_t row_lo,
                        int64_t row_hi, double* out_dev, int64_t* key_dev, int64_t* count_dev) {
  const int64_t rows = grow_rows_of(depth);
  if (row_hi <= row_lo || nseg == 0) return;
  const int64_t slot_base = grow_unique_before(row_lo);
  const int64_t uniq = grow_unique_before(row_hi) - slot_base;
  const dim3 grid((unsigned)((row_hi - row_lo + kGrowThreads - 1) / kGrowThreads), (unsigned)nseg);
  hipLaunchKernelGGL(grow_unique_kernel, grid, dim3(kGrowThreads), (size_t)2 * d * kGrowThreads * 8, st,
                     bounds_dev, d, depth, rows, row_lo, row_hi, uniq, slot_base, out_dev, key_dev,
                     reinterpret_cast<unsigned long long*>(count_dev));
}

template <typename TG>
void launch_grow_unique_prep(hipStream_t st, const GrowBoxes& boxes, int nseg, int d, int dp, int depth, int64_t row_lo,
                             int64_t row_hi, const double* ls_dev, TG* leaves_s, TG* lnorm, int64_t* key_dev,
                             unsigned long long* extra_dev) {
  const int64_t rows = grow_rows_of(depth);
  if (row_hi <= row_lo || nseg == 0) return;
  const int64_t slot_base = grow_unique_before(row_lo);
  const int64_t uniq = grow_unique_before(row_hi) - slot_base;
  const dim3 grid((unsigned)((row_hi - row_lo + kGrowThreads - 1) / kGrowThreads), (unsigned)nseg);
  hipLaunchKernelGGL((grow_unique_prep_kernel<TG>), grid, dim3(kGrowThreads), (size_t)2 * d * kGrowThreads * 8, st, boxes,
                     d, dp, depth, rows, row_lo, row_hi, uniq, slot_base, (int64_t)nseg * uniq, ls_dev, leaves_s, lnorm,
                     key_dev, extra_dev);
}
template void launch_grow_unique_prep<float>(hipStream_t, const GrowBoxes&, int, int, int, int, int64_t, int64_t, const double*, float*, float*, int64_t*, unsigned long long*);
template void launch_grow_unique_prep<double>(hipStream_t, const GrowBoxes&, int, int, int, int, int64_t, int64_t, const double*, double*, double*, int64_t*, unsigned long long*);

}  // namespace gpso
