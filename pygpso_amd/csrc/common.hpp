// Shared device-side definitions for libgpso_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpso {

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kPadN = 128;     // training size is padded to a multiple of this
constexpr int kFitBlock = 64;  // panel / tile edge of the factorisation kernels

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---- MFMA traits ----------------------------------------------------------------------------
// Both instructions compute a 16x16 tile D = A[16x4] * B[4x16] + C per wave with
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D:       4 values per lane, column j = l & 15, row:
//       f32 (v_mfma_f32_16x16x4_f32):  row = 4 * (l >> 4) + r
//       f64 (v_mfma_f64_16x16x4_f64):  row = (l >> 4) + 4 * r
template <typename T>
struct Mfma;

template <>
struct Mfma<float> {
  using vec4 = f32x4;
  static __device__ __forceinline__ vec4 mma(float a, float b, vec4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // row of accumulator register r on lane l
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
  // which A-operand row must carry "logical row m" so that register r of lane l ends up holding
  // logical row 4 * (l >> 4) + r  (identity for the f32 layout)
  static __device__ __forceinline__ int arow_for_k4(int m) { return m; }
};

template <>
struct Mfma<double> {
  using vec4 = f64x4;
  static __device__ __forceinline__ vec4 mma(double a, double b, vec4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
  // f64 C/D puts A-row m into register (m >> 2) of lane group (m & 3); feeding logical row
  // 4 * (m & 3) + (m >> 2) there makes register r of lane l hold logical row 4 * (l >> 4) + r
  static __device__ __forceinline__ int arow_for_k4(int m) { return 4 * (m & 3) + (m >> 2); }
};

// ---- hyper-parameters as the kernels see them ---------------------------------------------------
struct KernParams {
  int kernel;       // GPSO_MATERN52 ...
  double variance;  // sigma^2
  double noise;     // sigma_n^2
  double mean_c;    // constant mean
};

// k(r^2) for the four stationary kernels, GPflow semantics:
// Matern family clamps r^2 at 1e-36 before the square root; SE uses r^2 directly.
__device__ __forceinline__ double kern_from_r2(int kernel, double r2, double variance) {
  if (kernel == 3) return variance * exp(-0.5 * r2);
  const double r = sqrt(fmax(r2, 1e-36));
  if (kernel == 0) {
    const double s5 = 2.23606797749978969641;
    return variance * (1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * exp(-s5 * r);
  }
  if (kernel == 1) {
    const double s3 = 1.73205080756887729353;
    return variance * (1.0 + s3 * r) * exp(-s3 * r);
  }
  return variance * exp(-r);
}

__device__ __forceinline__ float kern_from_r2(int kernel, float r2, float variance) {
  if (kernel == 3) return variance * __expf(-0.5f * r2);
  const float r = __builtin_sqrtf(fmaxf(r2, 1e-36f));
  if (kernel == 0) {
    const float s5 = 2.2360679775f;
    return variance * (1.0f + s5 * r + (5.0f / 3.0f) * (r * r)) * __expf(-s5 * r);
  }
  if (kernel == 1) {
    const float s3 = 1.7320508076f;
    return variance * (1.0f + s3 * r) * __expf(-s3 * r);
  }
  return variance * __expf(-r);
}

// ---- hot-loop form: k as a function of u = C2 * r^2 with the scale C2 folded into the norms ----
// (Matern-5/2: C2 = 5 so t = sqrt(u) = sqrt5 r and 1 + sqrt5 r + 5/3 r^2 = 1 + t + t^2/3;
//  Matern-3/2: C2 = 3; Matern-1/2 and SE: C2 = 1).  KERNEL is a compile-time constant.
template <int KERNEL>
struct KernScale {
  static constexpr double C2 = (KERNEL == 0) ? 5.0 : (KERNEL == 1) ? 3.0 : 1.0;
};

// float: raw v_sqrt_f32 / v_exp_f32 (1 ulp each), no range fix-ups -- the hot loop's VALU budget
template <int KERNEL>
__device__ __forceinline__ float kern_from_scaled(float u, float variance) {
  constexpr float kNegLog2e = -1.4426950408889634f;
  if (KERNEL == 3) return variance * __builtin_amdgcn_exp2f(u * (0.5f * kNegLog2e));
  const float t = __builtin_amdgcn_sqrtf(fmaxf(u, (float)(KernScale<KERNEL>::C2 * 1e-36)));
  const float e = __builtin_amdgcn_exp2f(t * kNegLog2e);
  if (KERNEL == 0) return (variance * fmaf(t, fmaf(t, 1.0f / 3.0f, 1.0f), 1.0f)) * e;
  if (KERNEL == 1) return (variance * (1.0f + t)) * e;
  return variance * e;
}
// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, I1)
template <int I, int I1, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < I1) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, I1>(f);
  }
}
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }

// exp(x) for x <= ~0 in full double precision without the library's special-case handling:
// x = k ln2 + r, |r| <= ln2 / 2, degree-13 Taylor in r (remainder < 2e-17), v_ldexp_f64.
__device__ __forceinline__ double exp_lean(double x) {
  const double kf = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(-kf, 6.93147180369123816490e-01, x);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;  // 1/13!
  p = fma(p, r, 2.08767569878681e-09);    // 1/12!
  p = fma(p, r, 2.505210838544172e-08);   // 1/11!
  p = fma(p, r, 2.755731922398589e-07);   // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);  // 1/9!
  p = fma(p, r, 2.48015873015873e-05);    // 1/8!
  p = fma(p, r, 1.984126984126984e-04);   // 1/7!
  p = fma(p, r, 1.388888888888889e-03);   // 1/6!
  p = fma(p, r, 8.333333333333333e-03);   // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);  // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);  // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)fmax(kf, -1100.0));
}

// sqrt(x), x > 0: v_rsq_f64 seed, two Newton steps, one Heron correction (as in the Cholesky)
__device__ __forceinline__ double sqrt_lean(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x, r * r, 1.5);
  r = r * fma(-0.5 * x, r * r, 1.5);
  double s = x * r;
  return fma(0.5 * r, fma(-s, s, x), s);
}

// double (the parity path): full double precision, lean exp / sqrt (|rel err| ~ 1e-16)
template <int KERNEL>
__device__ __forceinline__ double kern_from_scaled(double u, double variance) {
  if (KERNEL == 3) return variance * exp_lean(-0.5 * u);
  const double t = sqrt_lean(fmax(u, KernScale<KERNEL>::C2 * 1e-36));
  const double e = exp_lean(-t);
  if (KERNEL == 0) return (variance * fma(t, fma(t, 1.0 / 3.0, 1.0), 1.0)) * e;
  if (KERNEL == 1) return variance * (1.0 + t) * e;
  return variance * e;
}

// The same map in three stages -- t = sqrt part, e = exponential, k = polynomial * e -- so that a hot loop
// can place them apart (the wave issues in order: a dependent transcendental right behind its producer
// stalls everything behind it, MFMAs included).  kern_stage3(t, kern_stage2(t), v) with t = kern_stage1(u)
// is kern_from_scaled(u, v) operation for operation.
template <int KERNEL>
__device__ __forceinline__ float kern_stage1(float u) {
  if (KERNEL == 3) return u;
  return __builtin_amdgcn_sqrtf(fmaxf(u, (float)(KernScale<KERNEL>::C2 * 1e-36)));
}
template <int KERNEL>
__device__ __forceinline__ float kern_stage2(float t) {
  constexpr float kNegLog2e = -1.4426950408889634f;
  if (KERNEL == 3) return __builtin_amdgcn_exp2f(t * (0.5f * kNegLog2e));
  return __builtin_amdgcn_exp2f(t * kNegLog2e);
}
template <int KERNEL>
__device__ __forceinline__ float kern_stage3(float t, float e, float variance) {
  if (KERNEL == 0) return (variance * fmaf(t, fmaf(t, 1.0f / 3.0f, 1.0f), 1.0f)) * e;
  if (KERNEL == 1) return (variance * (1.0f + t)) * e;
  return variance * e;
}
template <int KERNEL>
__device__ __forceinline__ double kern_stage1(double u) {
  if (KERNEL == 3) return u;
  return sqrt_lean(fmax(u, KernScale<KERNEL>::C2 * 1e-36));
}
template <int KERNEL>
__device__ __forceinline__ double kern_stage2(double t) {
  if (KERNEL == 3) return exp_lean(-0.5 * t);
  return exp_lean(-t);
}
template <int KERNEL>
__device__ __forceinline__ double kern_stage3(double t, double e, double variance) {
  if (KERNEL == 0) return (variance * fma(t, fma(t, 1.0 / 3.0, 1.0), 1.0)) * e;
  if (KERNEL == 1) return variance * (1.0 + t) * e;
  return variance * e;
}

// d k / d(r^2) * variance-scaled, used by the gradient reductions
__device__ __forceinline__ double dkern_dr2(int kernel, double r2, double variance) {
  if (kernel == 3) return -0.5 * variance * exp(-0.5 * r2);
  const double r = sqrt(fmax(r2, 1e-36));
  if (kernel == 0) {
    const double s5 = 2.23606797749978969641;
    return -variance * (5.0 / 6.0) * (1.0 + s5 * r) * exp(-s5 * r);
  }
  if (kernel == 1) {
    const double s3 = 1.73205080756887729353;
    return -variance * 1.5 * exp(-s3 * r);
  }
  return -variance * 0.5 * exp(-r) / r;
}

// k(r^2) with a run-time kernel id: float as kern_from_r2; double with the lean exp / sqrt (the
// library calls were most of the Gram kernel's time in double)
__device__ __forceinline__ float kern_from_r2_lean(int kernel, float r2, float variance) {
  return kern_from_r2(kernel, r2, variance);
}
__device__ __forceinline__ double kern_from_r2_lean(int kernel, double r2, double variance) {
  if (kernel == 3) return variance * exp_lean(-0.5 * r2);
  const double r = sqrt_lean(fmax(r2, 1e-36));
  if (kernel == 0) {
    const double s5 = 2.23606797749978969641;
    return variance * (1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * exp_lean(-s5 * r);
  }
  if (kernel == 1) {
    const double s3 = 1.73205080756887729353;
    return variance * (1.0 + s3 * r) * exp_lean(-s3 * r);
  }
  // (the negation is kept out of the optimiser's reach: with a run-time kernel id hipcc 7.2 -O3 folded
  // it into the merged branches wrongly here -- exp(+r), and 1e220 on the diagonal; tools/micro/gram_check.hip)
  double mr = -r;
  asm volatile("" : "+v"(mr));
  return variance * exp_lean(mr);
}

// k and dk/d(r^2) together with the lean exp / sqrt (full double precision, no special-case handling:
// every argument of exp is <= 0 and r^2 is clamped away from 0) -- the gradient kernel evaluates both
// per matrix entry and the library calls were most of its time
__device__ __forceinline__ void kern_and_dkern_lean(int kernel, double r2_k, double r2_dk, double variance,
                                                    double& k, double& dk) {
  if (kernel == 3) {
    k = variance * exp_lean(-0.5 * r2_k);
    dk = -0.5 * variance * exp_lean(-0.5 * r2_dk);
    return;
  }
  const double rk = sqrt_lean(fmax(r2_k, 1e-36)), rd = sqrt_lean(fmax(r2_dk, 1e-36));
  if (kernel == 0) {
    const double s5 = 2.23606797749978969641;
    k = variance * (1.0 + s5 * rk + (5.0 / 3.0) * (rk * rk)) * exp_lean(-s5 * rk);
    dk = -variance * (5.0 / 6.0) * (1.0 + s5 * rd) * exp_lean(-s5 * rd);
  } else if (kernel == 1) {
    const double s3 = 1.73205080756887729353;
    k = variance * (1.0 + s3 * rk) * exp_lean(-s3 * rk);
    dk = -variance * 1.5 * exp_lean(-s3 * rd);
  } else {
    k = variance * exp_lean(-rk);
    dk = -variance * 0.5 * exp_lean(-rd) / rd;
  }
}

// k and dk/d(r^2) at ONE squared distance (one sqrt, one exp): for callers whose r^2 is free of
// cancellation (direct differences), where evaluating k at the GEMM-form r^2 instead would change it by
// ~1e-15 relative at most
__device__ __forceinline__ void kern_and_dkern_same(int kernel, double r2, double variance, double& k, double& dk) {
  if (kernel == 3) {
    const double e = variance * exp_lean(-0.5 * r2);
    k = e;
    dk = -0.5 * e;
    return;
  }
  const double r = sqrt_lean(fmax(r2, 1e-36));
  if (kernel == 0) {
    const double s5 = 2.23606797749978969641;
    const double e = variance * exp_lean(-s5 * r);
    k = (1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * e;
    dk = -(5.0 / 6.0) * (1.0 + s5 * r) * e;
  } else if (kernel == 1) {
    const double s3 = 1.73205080756887729353;
    const double e = variance * exp_lean(-s3 * r);
    k = (1.0 + s3 * r) * e;
    dk = -1.5 * e;
  } else {
    double mr = -r;
    asm volatile("" : "+v"(mr));  // (see kern_from_r2_lean: keeps the negation out of a mis-folding merge)
    const double e = variance * exp_lean(mr);
    k = e;
    dk = -0.5 * e / r;
  }
}

// ---- split-bf16 helpers (predict.hip: leaf_tiles_bf16_kernel; fit.hip: syrk_bf16_kernel) ----------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (a, b) -> packed bf16 pair (round to nearest even); a, b are replaced by the remainders
__device__ __forceinline__ unsigned bf16_split_pair(float& a, float& b) {
  const f32x2 v = {a, b};
  const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  a -= __builtin_bit_cast(float, u << 16);
  b -= __builtin_bit_cast(float, u & 0xffff0000u);
  return u;
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// (a, b) -> packed fp16 pair (round to nearest even); a, b are replaced by the remainders
__device__ __forceinline__ unsigned f16_split_pair(float& a, float& b) {
  const f32x2 v = {a, b};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  a -= (float)h[0];
  b -= (float)h[1];
  return __builtin_bit_cast(unsigned, h);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

}  // namespace gpso
