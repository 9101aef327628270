// GP posterior fit on gfx950 (K1-K6 of SURVEY.md 2.4), all matrices resident in HBM, row-major,
// leading dimension npad (multiple of 128):
//
//   scale_x      X/lengthscale (+ squared norms, + MFMA-fragment packing for the predict kernel)
//   gram         K = k(X,X) + noise I      x.x^T on MFMA 16x16x4, Matern/SE map fused as epilogue
//   potrf        blocked right-looking Cholesky, 64-wide panels:
//                  potrf_diag (one wave, LDS) -> also inverts the diagonal block
//                  panel  L21 = A21 inv(L11)^T,  trailing A22 -= L21 L21^T   (MFMA tile GEMM)
//   trtri        L^-1 by level doubling: off-diagonal block = -inv(B) * C * inv(A), batched per level
//   pack_linv    re-tile L^-1 into the fragment-major layout leaf_tiles_kernel streams
//   solve_alpha  a = L^-1 (y-c), alpha = L^-T a, NLML (wavefront reductions, double accumulators)
//   gradient     Kinv = L^-T L^-1, then sum W o dK/dtheta reductions (analytic d NLML / d theta)
//
// Replaces what gpflow GPR.training_loss + its autodiff do per L-BFGS-B evaluation
// (gpso/gp_surrogate.py:500-503).
#include <algorithm>
#include <climits>
#include <functional>
#include <utility>
#include <vector>
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"
#include <type_traits>

namespace gpso {

// =============================================================================================
// scale + pack inputs
// =============================================================================================
// xs / xnorm feed the Gram and gradient kernels (always double, see gram_kernel); xs_p is the
// MFMA-fragment packing the predict kernels stream, in THEIR generation type T.
template <typename T>
__global__ __launch_bounds__(256) void scale_x_kernel(const double* __restrict__ x64, int64_t n,
                                                      int64_t npad, int d, int dp,
                                                      const double* __restrict__ ls,
                                                      T* __restrict__ xs, T* __restrict__ xnorm,
                                                      T* __restrict__ xs_p /* nullable: the fragment packing as well */) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  // packed element ((kt * dp4 + c) * 64 + lane) holds row kt * 16 + arow_for_k4(lane & 15), column 4 c + (lane >> 4)
  // (pack_xs_kernel); arow_for_k4 is its own inverse (identity, or the transpose of a 4 x 4 index)
  const int dp4 = dp / 4;
  const int64_t pbase = (i >> 4) * dp4 * 64 + Mfma<T>::arow_for_k4((int)(i & 15));
  T acc = 0;
  for (int k = 0; k < dp; ++k) {
    T v = 0;
    if (i < n && k < d) v = (T)(x64[i * d + k] / ls[k]);
    xs[i * dp + k] = v;
    if (xs_p != nullptr) xs_p[pbase + (k >> 2) * 64 + 16 * (k & 3)] = v;
    acc += v * v;
  }
  xnorm[i] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_xs_kernel(const T* __restrict__ xs, int64_t npad, int dp,
                                                      T* __restrict__ xs_p) {
  // one thread per packed element: ((kt * dp4 + c) * 64 + lane)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int dp4 = dp / 4;
  if (idx >= npad * dp) return;
  const int lane = (int)(idx & 63);
  const int64_t q = idx >> 6;
  const int c = (int)(q % dp4);
  const int64_t kt = q / dp4;
  const int row = Mfma<T>::arow_for_k4(lane & 15);
  xs_p[idx] = xs[(kt * 16 + row) * dp + 4 * c + (lane >> 4)];
}

template <typename T>
void launch_scale_x(hipStream_t st, const double* x64, int64_t n, int64_t npad, int d, int dp,
                    const double* ls, T* xs, T* xnorm, T* xs_p) {
  hipLaunchKernelGGL((scale_x_kernel<T>), dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st,
                     x64, n, npad, d, dp, ls, xs, xnorm, xs_p);  // (one launch: the packing rides along)
}
// float copies of the (double) scaled inputs for float generation of the cross-Gram tile: exactly what
// scale_x_kernel<float> computes from the raw inputs -- (float)(x / l) is the rounding of the double
// quotient xs64 holds, the norm is accumulated in float in the same order -- without needing the raw
// inputs (a rank that RECEIVED the posterior only has the scaled ones)
__global__ __launch_bounds__(256) void gen_inputs_f32_kernel(const double* __restrict__ xs64, int64_t npad,
                                                             int dp, float* __restrict__ xs,
                                                             float* __restrict__ xnorm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  float acc = 0;
  for (int k = 0; k < dp; ++k) {
    const float v = (float)xs64[i * dp + k];
    xs[i * dp + k] = v;
    acc += v * v;
  }
  xnorm[i] = acc;
}

void launch_gen_inputs_f32(hipStream_t st, const double* xs64, int64_t npad, int dp, float* xs, float* xnorm,
                           float* xs_p) {
  hipLaunchKernelGGL(gen_inputs_f32_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, st, xs64, npad,
                     dp, xs, xnorm);
  hipLaunchKernelGGL((pack_xs_kernel<float>), dim3((unsigned)((npad * dp + 255) / 256)), dim3(256), 0, st, xs,
                     npad, dp, xs_p);
}
template void launch_scale_x<float>(hipStream_t, const double*, int64_t, int64_t, int, int, const double*, float*, float*, float*);
template void launch_scale_x<double>(hipStream_t, const double*, int64_t, int64_t, int, int, const double*, double*, double*, double*);

// =============================================================================================
// Gram matrix, lower 64x64 tiles only: x.x^T on the f64 MFMA, kernel map as epilogue
// =============================================================================================
// One workgroup per tile (ti >= tj; 1-D grid: a 2-D grid whose upper half exits at once leaves the 8
// XCDs unevenly loaded), one wave per 16-row strip.  r^2 = |x|^2 + |x'|^2 - 2 x.x' is ALWAYS formed in
// double from double inputs, whatever the matrix type T: in float that cancellation leaves ~1e-5 of
// absolute error in r^2 at the reference's lengthscales, i.e. a relative perturbation of K that the
// conditioning of K + sigma_n^2 I (1e6..1e7 at GPflow's noise floor) turns into a wrong posterior
// (measured: NLML off by 1.5e-3, negative variances -- profiles/r02a_precision_before.jsonl).  The
// contraction is D / 4 MFMAs per 16x16 tile and the kernel is bound by its N^2 / 2 * s bytes of stores.
// The four column tiles of a strip are INTERLEAVED (tile x, lane index i <-> column 4 i + x) so that a
// lane owns 4 consecutive columns: 16-byte (float) / 32-byte (double) stores, 256 / 512 contiguous
// bytes per row and instruction.
template <typename T>
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ xs,
                                                   const double* __restrict__ xnorm, int64_t n,
                                                   int64_t npad, int dp, int kernel, double variance,
                                                   double noise, T* __restrict__ K, int* __restrict__ info) {
  using vec4 = typename Mfma<T>::vec4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t blk = blockIdx.x;
  // (the factorisation's verdict starts at "no failing pivot": set here, not by a host-to-device copy of its own)
  if (info != nullptr && blk == 0 && threadIdx.x == 0) *info = 0x7fffffff;
  int ti = (int)((__builtin_sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= blk) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > blk) --ti;
  const int tj = (int)(blk - (int64_t)ti * (ti + 1) / 2);
  const int64_t i0 = (int64_t)ti * 64 + wave * 16;
  const int64_t j0 = (int64_t)tj * 64;
  // the two 64-row panels of scaled inputs go through LDS (coalesced copy, odd row stride): fetched
  // straight from global as MFMA fragments they are 8-byte accesses 8 dp bytes apart, five dependent
  // loads per k-step -- at D = 40 that, not the stores, was the kernel's time
  extern __shared__ __align__(16) double gram_lds[];
  const int st = dp + 1;
  double* pi = gram_lds;
  double* pj = (ti == tj) ? pi : gram_lds + 64 * st;
  for (int e = threadIdx.x; e < 64 * dp; e += 256) {
    const int r = e / dp, k = e - r * dp;
    pi[r * st + k] = xs[(int64_t)ti * 64 * dp + e];
    if (ti != tj) pj[r * st + k] = xs[(int64_t)tj * 64 * dp + e];
  }
  __syncthreads();
  f64x4 s[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) s[x] = f64x4{0, 0, 0, 0};
  const double* pa = pi + (wave * 16 + (lane & 15)) * st + (lane >> 4);
  const double* pb = pj + (4 * (lane & 15)) * st + (lane >> 4);
  for (int c = 0; c < dp / 4; ++c) {
    const double a = pa[4 * c];
#pragma unroll
    for (int x = 0; x < 4; ++x) s[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[x * st + 4 * c], s[x], 0, 0, 0);
  }
  const int64_t jc = j0 + 4 * (lane & 15);  // first of this lane's 4 columns
  double nbj[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) nbj[x] = xnorm[jc + x];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = i0 + (lane >> 4) + 4 * r;  // f64 accumulator layout
    const double nai = xnorm[i];
    vec4 v;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int64_t j = jc + x;
      if (i < n && j < n) {
        const double r2 = fma(-2.0, s[x][r], nai + nbj[x]);
        T kv = kern_from_r2_lean(kernel, (T)r2, (T)variance);
        if (i == j) kv += (T)noise;
        v[x] = kv;
      } else {
        v[x] = (i == j) ? T(1) : T(0);
      }
    }
    *reinterpret_cast<vec4*>(K + i * npad + jc) = v;
  }
}

template <typename T>
void launch_gram(hipStream_t st, const double* xs, const double* xnorm, int64_t n, int64_t npad, int dp,
                 const KernParams& kp, T* K, int* info) {
  const int64_t nt = npad / 64;
  hipLaunchKernelGGL((gram_kernel<T>), dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256),
                     (size_t)2 * 64 * (dp + 1) * sizeof(double), st, xs, xnorm, n, npad, dp, kp.kernel, kp.variance,
                     kp.noise, K, info);
}
template void launch_gram<float>(hipStream_t, const double*, const double*, int64_t, int64_t, int, const KernParams&, float*, int*);
template void launch_gram<double>(hipStream_t, const double*, const double*, int64_t, int64_t, int, const KernParams&, double*, int*);

// =============================================================================================
// 64x64 diagonal block: Cholesky + triangular inverse in LDS, 4 waves, blocked by 16 columns
// =============================================================================================
// The block is latency-bound (a chain of 64 pivots), so the work between two pivots is kept short:
//   factor   per 16-column panel: left-looking columns whose dot products only span the panel
//            (<= 15 terms, wave 0, lane = row), then ONE batched rank-16 update of the trailing
//            columns by all 256 threads (the panel row of each lane lives in registers);
//   invert   the four 16x16 diagonal blocks in parallel (one per wave), then the off-diagonal
//            blocks by distance: X[ib][jb] = -Xd[ib] * sum_kb L[ib][kb] X[kb][jb], one thread per
//            element of a 16x16 block.
// Accumulation is in double whatever T is; sqrt / divide are v_rsq_f64 + Newton (the library
// sqrt/div sequences would sit on the pivot chain).
#ifndef GPSO_STAMP
#define GPSO_STAMP(i)  // tools/micro/diag_phases.hip defines this to record s_memtime stamps
#endif
constexpr int kDS = kFitBlock + 1;  // LDS row stride (conflict-free row-per-lane access)
constexpr int kPB = 16;             // panel width inside the block

// 1/sqrt(x): hardware v_rsq_f64 seed (~2^-23 relative) + NEWTON Newton steps: two give full double
// precision; one (~2e-14) is far below float rounding and is what float contexts use
template <int NEWTON>
__device__ __forceinline__ double rsqrt_newton(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = -0.5 * x;
#pragma unroll
  for (int s = 0; s < NEWTON; ++s) r = r * fma(hx, r * r, 1.5);
  return r;
}

// one wave: D (16x16) = sum_{k<K} A(i,k) B(k,j) with both operands in LDS (generic strides), on the
// f64 MFMA (16x16x4).  Element r of lane l of the result is D[(l >> 4) + 4 r][l & 15].  K is a
// compile-time constant so that every fragment is in flight before the first MFMA issues (with a
// run-time trip count each k-step paid a full LDS round trip).
template <int K>
__device__ __forceinline__ f64x4 mma16_lds(const double* Ab, int sai, int sak, const double* Bb,
                                           int sbk, int sbj, int lane) {
  double a[K / 4], b[K / 4];
#pragma unroll
  for (int s = 0; s < K / 4; ++s) {
    a[s] = Ab[(lane & 15) * sai + (4 * s + (lane >> 4)) * sak];
    b[s] = Bb[(4 * s + (lane >> 4)) * sbk + (lane & 15) * sbj];
  }
  f64x4 acc{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
  return acc;
}

// the same, accumulated onto acc
template <int K>
__device__ __forceinline__ f64x4 mma16_lds_acc(f64x4 acc, const double* Ab, int sai, int sak, const double* Bb,
                                               int sbk, int sbj, int lane) {
  double a[K / 4], b[K / 4];
#pragma unroll
  for (int s = 0; s < K / 4; ++s) {
    a[s] = Ab[(lane & 15) * sai + (4 * s + (lane >> 4)) * sak];
    b[s] = Bb[(4 * s + (lane >> 4)) * sbk + (lane & 15) * sbj];
  }
#pragma unroll
  for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
  return acc;
}

// value of lane (16 * (lane / 16) + N) for every lane: 64-bit DPP row_newbcast, one v_mov_b64_dpp
template <int N>
__device__ __forceinline__ double row_bcast_f64(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + N, 0xf, 0xf, false);
}


__device__ __forceinline__ double readlane_f64(double x, int src_lane /* wave-uniform */) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// one thread: row r, columns cb .. cb+15 of the lower-triangular f64 LDS tile -> global T tile; 16-byte stores, vec4 groups entirely above the
// diagonal are skipped (nothing reads them: Lf's upper part is never used, linv is pre-zeroed)
template <typename T, int V0 = 0, int V1 = 4>
__device__ __forceinline__ void lower_cols_to_global(const double* S, T* __restrict__ dst, int64_t ld,
                                                     int r, int cb) {
  using vec4 = typename Mfma<T>::vec4;
  T* o = dst + (int64_t)r * ld + cb;
#pragma unroll
  for (int v = V0; v < V1; ++v) {
    if (cb + 4 * v <= r) {
      vec4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cb + 4 * v + e;
        x[e] = (c <= r) ? (T)S[r * kDS + c] : (T)0;
      }
      *reinterpret_cast<vec4*>(o + 4 * v) = x;
    }
  }
}

// all 256 threads: lower part of a 64x64 f64 LDS tile (stride kDS) -> global T tile (the destination's
// upper part must be pre-zeroed or unused); a thread owns 16 (or 32) consecutive columns of one row (16-byte stores)
// (NT = 256: all threads, 16 columns each; NT = 128: t in [0, 128), 32 columns each)
// ZERO_ABOVE: the groups above the diagonal are written as zeros as well -- the destination then needs no zero fill
// (role D's diagonal tile of L^-1 is read WHOLE by the products of the later steps)
template <typename T, int NT = 256, bool ZERO_ABOVE = false>
__device__ __forceinline__ void lower_tile_to_global(const double* S, T* __restrict__ dst, int64_t ld, int t) {
  using vec4 = typename Mfma<T>::vec4;
  constexpr int kPerRow = NT / kFitBlock, kCols = kFitBlock / kPerRow;
  const int r = t / kPerRow, cb = kCols * (t % kPerRow);
  T* o = dst + (int64_t)r * ld + cb;
#pragma unroll
  for (int v = 0; v < kCols / 4; ++v) {
    if (ZERO_ABOVE || cb + 4 * v <= r) {  // groups entirely above the diagonal: destination is pre-zeroed / never read
      vec4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cb + 4 * v + e;
        x[e] = (c <= r) ? (T)S[r * kDS + c] : (T)0;
      }
      *reinterpret_cast<vec4*>(o + 4 * v) = x;
    }
  }
}

// 1 / x in full double precision: v_rcp_f64 seed + two Newton steps (the library division is ~30
// instructions)
__device__ __forceinline__ double rcp_newton(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// One wave, four columns of the inverse of the 16x16 lower-triangular block at Ls[b0.., b0..]:
// the 16-lane row g = lane / 16 of the wave computes column c = cbase + g, lane % 16 = row i.
// Forward substitution right-looking; the finished entry X[k][c] reaches the rows below through a
// row-local 64-bit DPP broadcast, so a step is mul -> v_mov_b64_dpp -> fma on all 64 lanes
// (~55 instructions for the four columns).
__device__ __forceinline__ void diag_inv16(const double* Ls, double* Xs, int b0, int cbase, int lane) {
  const int i = lane & 15, c = cbase + (lane >> 4);
  const double* Lrow = Ls + (b0 + i) * kDS + b0;
  double ld[kPB];
#pragma unroll
  for (int k = 0; k < kPB; ++k) {
    const double v = Lrow[k];
    ld[k] = (k < i) ? v : 0.0;  // entries on / above the diagonal are unspecified in Ls: select, not multiply
  }
  const double invd = rcp_newton(Lrow[i]);
  double r = (i == c) ? 1.0 : 0.0;  // residual e_c - sum_{k<i} L[i][k] X[k][c]
  static_for<0, kPB - 1>([&](auto k_) {
    constexpr int k = decltype(k_)::value;
    const double xk = row_bcast_f64<k>(r * invd);  // X[k][c]
    r = fma(-ld[k], xk, r);
  });
  Xs[(b0 + i) * kDS + b0 + c] = r * invd;
}

// all 256 threads; Ls holds the symmetric block (lower part used); on return the lower part of Ls
// is L (entries above the diagonal are unspecified: nothing reads them and the stores mask them).
// A non-positive pivot turns its column into NaN (v_rsq of a non-positive number), which spreads
// only to later columns: the first non-finite / non-positive diagonal entry is the failing pivot,
// reported with atomicMin(info, global index) for rows < n.
// While wave 0 factors panel p, wave 1 inverts the 16x16 diagonal block of panel p-1 into Xs (off
// the pivot chain); the last diagonal block is left to trinv64_lds.
// Wave 2 stores the 16 finished columns of panel p-1 to Lout (global) at the same time; the last 16
// columns are left to the caller.
// F32CHAIN (float fits of the general path): the panel lives in float registers, the pivot's 1/sqrt is
// one v_rsq_f32 (1 ulp, no Newton step) and the multipliers come back from a float copy of the finished
// columns (Fs[jj][row], 4 KB borrowed from the inverse's scratch, free until trinv64_lds).  The factor
// is stored as float anyway and every trailing update of a float fit already rounds the block to
// float; what changes is the rounding of the <= 63 updates inside the block (float instead of double).
__device__ __forceinline__ float readlane_t(float x, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
}
__device__ __forceinline__ double readlane_t(double x, int src_lane) { return readlane_f64(x, src_lane); }
template <int NEWTON>
__device__ __forceinline__ float rsqrt_t(float x) { return __builtin_amdgcn_rsqf(x); }
template <int NEWTON>
__device__ __forceinline__ double rsqrt_t(double x) { return rsqrt_newton<NEWTON>(x); }



// One wave, panel columns c0 .. c0+15 of the 64x64 block in Ls; lane = row, the 16 panel entries of the
// row in registers (type F, as pairs): the 16-pivot chain of a panel runs without branches and without
// selects.  Right-looking: column jj is final at step jj.  On the pivot chain: readlane -> rsqrt -> scale
// -> update of column jj+1 (multiplier by v_readlane).  Off the chain: the finished column goes to its
// final place in Ls and to a column-major copy Cs[jj][row] (type F, 4 / 8 KB borrowed from the inverse's
// scratch, free until trinv64_lds), from which the multipliers of the columns from jj+3 on come back as
// wave-uniform reads, two per instruction.  What shapes the loop (tools/micro/lat_probe.hip): a wave
// alone on its SIMD issues one instruction per ~5.6 clocks WHATEVER its kind (s_waitcnt included), and
// an LDS write -> read round trip is ~70 clocks.  So
//  * the reads of column jj are issued right behind its store and consumed ONE STEP LATER (software
//    pipeline: step jj applies column jj-1 to the columns >= jj+2; column jj+2, which the next step's
//    chain needs, takes its multiplier by v_readlane as well), into double-buffered registers;
//  * the delayed update walks the pairs from the last one issued down, so one counted s_waitcnt covers
//    all of them;
//  * float panels update two columns per v_pk_fma_f32.
// Every entry receives its column updates in the order 0, 1, 2, ...: same bits as the plain loop.
// (Measured and not kept, tools/micro/diag_phases: the chain freed of v_readlane by repeating the
// operations of lanes jj+1 / jj+2 on wave-uniform copies fetched a step ahead, with and without the
// delayed updates pinned under the rsqrt, LDS multipliers two steps late: no faster.)
template <typename F, int NEWTON>
__device__ __forceinline__ void chol_panel16(double* Ls, F* Cs, int c0, int lane) {
  typedef F F2 __attribute__((ext_vector_type(2)));
  F2 lp[kPB / 2], m2[2][kPB / 2];
#pragma unroll
  for (int k = 0; k < kPB / 2; ++k) {
    lp[k].x = (F)Ls[lane * kDS + c0 + 2 * k];
    lp[k].y = (F)Ls[lane * kDS + c0 + 2 * k + 1];
  }
  F lprev = 0;
  static_for<0, kPB>([&](auto jj_) {
    constexpr int jj = decltype(jj_)::value;
    constexpr int b = jj & 1;  // m2[b]: filled with column jj's multipliers here, consumed by step jj+1
    const F cur = (jj & 1) ? lp[jj >> 1].y : lp[jj >> 1].x;
    const F l = cur * rsqrt_t<NEWTON>(readlane_t(cur, c0 + jj));
    Cs[jj * kFitBlock + lane] = l;
    Ls[lane * kDS + c0 + jj] = (double)l;
    // multiplier pairs of column jj for the columns >= jj+3 (the first pair may start one column early)
#pragma unroll
    for (int P = (jj + 3) >> 1; P < kPB / 2; ++P)
      m2[b][P] = *reinterpret_cast<const F2*>(Cs + jj * kFitBlock + c0 + 2 * P);
    if constexpr (jj + 1 < kPB) {
      const F mu = readlane_t(l, c0 + jj + 1);
      if constexpr ((jj + 1) & 1) lp[(jj + 1) >> 1].y = fma_t(-l, mu, lp[(jj + 1) >> 1].y);
      else lp[(jj + 1) >> 1].x = fma_t(-l, mu, lp[(jj + 1) >> 1].x);
    }
    if constexpr (jj >= 1 && jj + 2 < kPB) {
      constexpr int k0 = jj + 2;  // first column of the delayed update (column jj-1 applied)
      const F2 nl = {-lprev, -lprev};
      static_for<0, kPB / 2 - ((k0 + 1) >> 1)>([&](auto q_) {
        constexpr int P = kPB / 2 - 1 - decltype(q_)::value;
        // (the one packed FMA left in the library -- the toolchain has shown a wait-state hazard in front of a
        // chain of packed FMAs fed straight from LDS reads, profiles/r02h_packed_mean_bug.txt, cured by ~128 clocks
        // of distance: here the multiplier pairs were read a whole step, >= 150 clocks, before this use, and the
        // panel has been bit-identical over 27 000 probe runs; two scalar FMAs instead cost 7 % of a panel)
        lp[P] = __builtin_elementwise_fma(nl, m2[b ^ 1][P], lp[P]);
      });
      if constexpr (k0 & 1) lp[k0 >> 1].y = fma_t(-lprev, m2[b ^ 1][k0 >> 1].y, lp[k0 >> 1].y);
    }
    if constexpr (jj + 2 < kPB) {
      const F mu2 = readlane_t(l, c0 + jj + 2);
      if constexpr ((jj + 2) & 1) lp[(jj + 2) >> 1].y = fma_t(-l, mu2, lp[(jj + 2) >> 1].y);
      else lp[(jj + 2) >> 1].x = fma_t(-l, mu2, lp[(jj + 2) >> 1].x);
    }
    lprev = l;
  });
}

// (scratch: 8 KB that nothing else uses until trinv64_lds -- its Ts)
// EARLY (the step kernel's diagonal role): the waves that idle while wave 0 factors panels 2 and 3 form the products
// of the block inverse that only need finished panels -- panel 2: T10 = L10 X00 (wave 3); panel 3: X10 = -X11 T10, then
// the k <= 31 tiles of T = L21 [X00 0; X10 X11] (wave 3), its other two tiles (wave 2, behind its stores) and
// T32 = L32 X22 (wave 1, behind the diagonal inverse it has just formed) -- so that trinv64_lds<.., EARLY> is left with
// three stages instead of five on the pivot chain.  The products wait in the parts of Xs ABOVE its diagonal blocks,
// which nothing reads (every store of the inverse masks them): block (0,1) T10, block (2,3) T32, rows 0-31 x columns
// 32-63 T.  Same operations on the same operands as trinv64_lds's own stages: same bits.
__device__ __forceinline__ double* early_t10(double* Xs) { return Xs + kPB; }
__device__ __forceinline__ double* early_t32(double* Xs) { return Xs + 2 * kPB * kDS + 3 * kPB; }
__device__ __forceinline__ double* early_t(double* Xs) { return Xs + 2 * kPB; }
template <int NEWTON, typename T, bool F32CHAIN = false, bool EARLY = false>
__device__ __forceinline__ void chol64_lds(double* Ls, double* Xs, T* __restrict__ Lout, int64_t ld,
                                           int64_t k0, int64_t n, int* info, double* scratch) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // trailing update on the f64 MFMA: tile t of the lower triangle (row-major) right of panel p,
  // A[ib][mb] -= L[ib][panel] L[mb][panel]^T
  auto update_tile = [&](int p, int t) {
    int ib = 0, mb = t;
    while (mb > ib) {
      mb -= ib + 1;
      ++ib;
    }
    const int c = p * kPB, r0 = (p + 1 + ib) * kPB, m0 = (p + 1 + mb) * kPB;
    const f64x4 d = mma16_lds<kPB>(Ls + r0 * kDS + c, kDS, 1, Ls + m0 * kDS + c, 1, kDS, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ls[(r0 + (lane >> 4) + 4 * r) * kDS + m0 + (lane & 15)] -= d[r];
  };
  for (int c0 = 0; c0 < kFitBlock; c0 += kPB) {
    GPSO_STAMP(2 * (c0 / kPB));
    // EARLY: the two tiles of panel 0's update that panel 1 does not read -- (3,2) and (3,3) -- are left to waves 2 and
    // 3 beside panel 1 (one round of four tiles behind panel 0 instead of two); they are in place before the update of
    // panel 1 touches the same tiles, so every entry still receives its updates in the order 0, 1, 2
    if (EARLY && c0 == kPB && wave >= 2) update_tile(0, wave + 2);
    if (wave == 0) {
      if constexpr (F32CHAIN) chol_panel16<float, NEWTON>(Ls, reinterpret_cast<float*>(scratch), c0, lane);
      else chol_panel16<double, NEWTON>(Ls, scratch, c0, lane);
    } else if (wave == 1 && c0 > 0) {
      for (int cb = 0; cb < kPB; cb += 4) diag_inv16(Ls, Xs, c0 - kPB, cb, lane);
      if (EARLY && c0 == 3 * kPB) {  // X22 is this wave's own work of a moment ago; L32 is final since panel 2
        const f64x4 t = mma16_lds<kPB>(Ls + 3 * kPB * kDS + 2 * kPB, kDS, 1, Xs + 2 * kPB * kDS + 2 * kPB, kDS, 1, lane);
        double* t32 = early_t32(Xs);
#pragma unroll
        for (int r = 0; r < 4; ++r) t32[((lane >> 4) + 4 * r) * kDS + (lane & 15)] = t[r];
      }
    } else if (wave == 2 && c0 > 0) {
      lower_cols_to_global<T>(Ls, Lout, ld, lane, c0 - kPB);
      if (EARLY && c0 == 3 * kPB) {  // T[a][1] = L21[a][1] X11: X11 was finished during panel 2
        for (int a = 0; a < 2; ++a) {
          const f64x4 t = mma16_lds<16>(Ls + (32 + kPB * a) * kDS + kPB, kDS, 1, Xs + kPB * kDS + kPB, kDS, 1, lane);
          double* tt = early_t(Xs);
#pragma unroll
          for (int r = 0; r < 4; ++r) tt[(kPB * a + (lane >> 4) + 4 * r) * kDS + kPB + (lane & 15)] = t[r];
        }
      }
    } else if (EARLY && wave == 3 && c0 == 2 * kPB) {  // T10 = L10 X00 (X00: finished during panel 1)
      const f64x4 t = mma16_lds<kPB>(Ls + kPB * kDS, kDS, 1, Xs, kDS, 1, lane);
      double* t10 = early_t10(Xs);
#pragma unroll
      for (int r = 0; r < 4; ++r) t10[((lane >> 4) + 4 * r) * kDS + (lane & 15)] = t[r];
    } else if (EARLY && wave == 3 && c0 == 3 * kPB) {
      {  // X10 = -X11 T10
        const f64x4 x = mma16_lds<kPB>(Xs + kPB * kDS + kPB, kDS, 1, early_t10(Xs), kDS, 1, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(kPB + (lane >> 4) + 4 * r) * kDS + (lane & 15)] = -x[r];
      }
      for (int a = 0; a < 2; ++a) {  // T[a][0] = L21[a][0:2] [X00; X10]
        const f64x4 t = mma16_lds<32>(Ls + (32 + kPB * a) * kDS, kDS, 1, Xs, kDS, 1, lane);
        double* tt = early_t(Xs);
#pragma unroll
        for (int r = 0; r < 4; ++r) tt[(kPB * a + (lane >> 4) + 4 * r) * kDS + (lane & 15)] = t[r];
      }
    }
    __syncthreads();
    GPSO_STAMP(2 * (c0 / kPB) + 1);
    // trailing update: one tile per wave and step
    {
      const int p = c0 / kPB, nt = 3 - p;  // trailing tiles per side
      const int ntile = (EARLY && p == 0) ? 4 : nt * (nt + 1) / 2;
      for (int t = wave; t < ntile; t += 4) update_tile(p, t);
    }
    __syncthreads();
  }
  if (wave == 0) {
    const double dg = Ls[lane * kDS + lane];
    const bool ok = (dg > 0.0) && (dg < 1.0e300);  // false for NaN / inf as well
    const unsigned long long badmask = __ballot(!ok && (k0 + lane < n));
    if (badmask != 0 && lane == 0) atomicMin(info, (int)(k0 + __builtin_ctzll(badmask)));
  }
}

// all 256 threads; lower part of Ls = L  ->  lower part of Xs = L^-1 (entries above the diagonal
// BLOCKS are never written: stores mask them).  Recursive in two levels so that every stage keeps
// the MFMA busy on all waves and only five barriers sit on the chain:
//   16x16 diagonal inverses (diag_inv16; FIRST3_DONE: blocks 0..2 were already inverted beside the
//   factorisation and only block 3 is left, four columns per wave) -> blocks (1,0), (3,2):
//   X = -Xd (L Xd) -> the 32x32 block [2:4][0:2]: X21 = -X22 (L21 X11), one 16x16 tile per wave in
//   both products.
constexpr int kTsLd = 33;                  // row stride of the 32x32 scratch
constexpr int kTsDoubles = 32 * kTsLd;
// Lout != nullptr: waves 2 and 3, idle in the level-16 stages, store the last 16 columns of the L
// tile there (chol64_lds stored the others beside the factorisation).
template <bool FIRST3_DONE, typename T, bool EARLY = false>
__device__ __forceinline__ void trinv64_lds(const double* Ls, double* Xs,
                                            double* Ts /* [32][33] scratch */, T* Lout, int64_t ld) {
  static_assert(!EARLY || FIRST3_DONE, "the early products are formed beside the factorisation (chol64_lds<.., EARLY>)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  GPSO_STAMP(8);
  if (FIRST3_DONE) {
    diag_inv16(Ls, Xs, 3 * kPB, 4 * wave, lane);
  } else {
    for (int cb = 0; cb < kPB; cb += 4) diag_inv16(Ls, Xs, wave * kPB, cb, lane);
  }
  __syncthreads();
  GPSO_STAMP(9);
  if constexpr (EARLY) {
    // X10, T32 and T are there already: X32 = -X33 T32, then X21 = -X22' T
    if (wave == 1) {
      const f64x4 x = mma16_lds<kPB>(Xs + 3 * kPB * kDS + 3 * kPB, kDS, 1, early_t32(Xs), kDS, 1, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) Xs[(3 * kPB + (lane >> 4) + 4 * r) * kDS + 2 * kPB + (lane & 15)] = -x[r];
    } else if (Lout != nullptr && lane < kPB) {
      if (wave == 2) lower_cols_to_global<T, 0, 2>(Ls, Lout, ld, 3 * kPB + lane, 3 * kPB);
      if (wave == 3) lower_cols_to_global<T, 2, 4>(Ls, Lout, ld, 3 * kPB + lane, 3 * kPB);
    }
    __syncthreads();
    GPSO_STAMP(12);
    {
      const int a = wave >> 1, b = wave & 1;
      const double* Ab = Xs + (32 + kPB * a) * kDS + 32;
      const double* tt = early_t(Xs);
      const f64x4 x = (a == 1) ? mma16_lds<32>(Ab, kDS, 1, tt + kPB * b, kDS, 1, lane)
                               : mma16_lds<16>(Ab, kDS, 1, tt + kPB * b, kDS, 1, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Xs[(32 + kPB * a + (lane >> 4) + 4 * r) * kDS + kPB * b + (lane & 15)] = -x[r];
    }
    __syncthreads();
    GPSO_STAMP(13);
    return;
  }
  // level 16: blocks (1,0) and (3,2)
  if (wave < 2) {
    const int jb = 2 * wave, ib = jb + 1;
    const f64x4 t = mma16_lds<kPB>(Ls + ib * kPB * kDS + jb * kPB, kDS, 1,
                                   Xs + jb * kPB * kDS + jb * kPB, kDS, 1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[(wave * kPB + (lane >> 4) + 4 * r) * kTsLd + (lane & 15)] = t[r];
  } else if (Lout != nullptr && lane < kPB) {
    if (wave == 2) lower_cols_to_global<T, 0, 2>(Ls, Lout, ld, 3 * kPB + lane, 3 * kPB);
    if (wave == 3) lower_cols_to_global<T, 2, 4>(Ls, Lout, ld, 3 * kPB + lane, 3 * kPB);
  }
  __syncthreads();
  GPSO_STAMP(10);
  if (wave < 2) {
    const int jb = 2 * wave, ib = jb + 1;
    const f64x4 x = mma16_lds<kPB>(Xs + ib * kPB * kDS + ib * kPB, kDS, 1, Ts + wave * kPB * kTsLd,
                                   kTsLd, 1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      Xs[(ib * kPB + (lane >> 4) + 4 * r) * kDS + jb * kPB + (lane & 15)] = -x[r];
  }
  __syncthreads();
  GPSO_STAMP(11);
  // level 32: tile (a, b) of the 32x32 block per wave
  {
    const int a = wave >> 1, b = wave & 1;
    // T = L21 X11; X11 is lower triangular: column block b only needs k >= 16 b
    const double* Ab = Ls + (32 + kPB * a) * kDS + kPB * b;
    const double* Bb = Xs + (kPB * b) * kDS + kPB * b;
    const f64x4 t = (b == 0) ? mma16_lds<32>(Ab, kDS, 1, Bb, kDS, 1, lane)
                             : mma16_lds<16>(Ab, kDS, 1, Bb, kDS, 1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ts[(kPB * a + (lane >> 4) + 4 * r) * kTsLd + kPB * b + (lane & 15)] = t[r];
  }
  __syncthreads();
  GPSO_STAMP(12);
  {
    const int a = wave >> 1, b = wave & 1;
    // X21 = -X22 T; X22 is lower triangular: row block a only needs k < 16 (a + 1)
    const double* Ab = Xs + (32 + kPB * a) * kDS + 32;
    const f64x4 x = (a == 1) ? mma16_lds<32>(Ab, kDS, 1, Ts + kPB * b, kTsLd, 1, lane)
                             : mma16_lds<16>(Ab, kDS, 1, Ts + kPB * b, kTsLd, 1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      Xs[(32 + kPB * a + (lane >> 4) + 4 * r) * kDS + kPB * b + (lane & 15)] = -x[r];
  }
  __syncthreads();
  GPSO_STAMP(13);
}

// inverse of every 64x64 diagonal block of an already-factorised L (gpso_set_posterior path)
template <typename T>
__global__ __launch_bounds__(256) void trinv_diag_kernel(const T* __restrict__ L,
                                                         T* __restrict__ linv, int64_t ld) {
  __shared__ double Ls[kFitBlock * kDS];
  __shared__ double Xs[kFitBlock * kDS];
  __shared__ double Ts[kTsDoubles];
  const int tid = threadIdx.x;
  const int64_t k0 = (int64_t)blockIdx.x * kFitBlock;
  const T* A = L + k0 * ld + k0;
  for (int e = tid; e < kFitBlock * kFitBlock; e += 256) {
    const int r = e >> 6, c = e & 63;
    Ls[r * kDS + c] = (c <= r) ? (double)A[(int64_t)r * ld + c] : 0.0;
  }
  __syncthreads();
  trinv64_lds<false, T>(Ls, Xs, Ts, nullptr, 0);
  lower_tile_to_global<T>(Xs, linv + k0 * ld + k0, ld, tid);
}

// =============================================================================================
// batched tile GEMM descriptor:  C = alpha * opA * opB + beta * C
// =============================================================================================
// (layout of the split-bf16 planes a GEMM may emit beside C: see SyrkBf16Desc below)
__host__ __device__ __forceinline__ int64_t syrk_plane_offset(int64_t row, int k, int nkb);

struct GemmDesc {
  const void* A;
  int64_t sai, sak;  // opA(i,k) = A[i*sai + k*sak]
  const void* B;
  int64_t sbk, sbj;  // opB(k,j) = B[k*sbk + j*sbj]
  void* C;
  int64_t ldc;
  int64_t batchA, batchB, batchC;  // element strides between batch entries (blockIdx.z)
  int m, n, k;                     // sizes of a full batch entry, multiples of 64
  int m_last;                      // rows of the LAST batch entry (multiple of 64, <= m)
  int nbatch;
  double alpha, beta;
  int lower_only;  // only tiles with tj <= ti (needs m == n)
  int kmode;       // 0: all k | 1: k >= TS tj | 2: k >= TS ti | 3: k < TS (ti + 1) | 4: k < TS (tj + 1)
  int ts_hint;     // 0: tile size by the launch's own tile count | 64 / 128: as the launch this one is a PART of chose (same bits)
  // float only, nullable: also write C (rows / columns relative to C) as three bf16 planes in the layout
  // syrk_bf16_kernel reads (the TRSM of the two-level Cholesky hands its panel to the SYRK this way)
  unsigned short* split;
  int64_t split_stride;  // bf16 elements between planes
  int split_nkb;         // 32-column blocks per row of the plane layout
  int64_t split_row0, split_col0;  // position of C's (0, 0) in the planes
  int split_np;          // 3: bf16 pieces | 2: fp16 pieces of value * split_scale (a power of two)
  float split_scale;
};

// =============================================================================================
// tile GEMM through LDS-DMA, 128x128 or 64x64 tiles
// =============================================================================================
// (A register-staged predecessor spent most of a k-step waiting: a lone 128x128x256 tile took 38 us
// against 14 us of MFMA time.)  The A and B panels of a k-step go global -> LDS with
// global_load_lds into fragment-major double buffers -- every 16-byte DMA lands exactly where the
// lane that will feed it to the MFMA reads it, so fragment reads are conflict-free and linear in the
// lane id -- with ONE barrier per k-step and the next step's DMA in flight under the MFMAs.
//   k-block = 16 consecutive k.  A lane's fragment is the 4 values k = 4 (lane / 16) + e, e < 4, of
//   row (lane % 16) of a 16-row block; MFMA step e takes element e on both operands (the k order
//   inside a block is a permutation shared by A and B).
//   k-contiguous operand (KC): one 16-byte DMA per lane brings its own fragment (float; two for
//   double): LDS [tile][slab][lane] x 16 B.
//   row-contiguous operand (RC: consecutive rows adjacent in memory, k strided): a lane brings 4
//   (double: 2) consecutive rows at one k; the LDS image is [k][strip rows] and, the tiles of a strip
//   being interleaved, one 16-byte LDS read yields the same k for all four tiles.
//   Per operand and k-block: float 8 DMA instructions / 8 KB, double 16 / 16 KB.  A k-step is two
//   k-blocks for float and one for double: 16 DMA instructions and 16 KB per operand either way,
//   64 KB of LDS for the two buffers of both operands (2 workgroups per CU).
// C is read once, up front, into the accumulators (scaled by beta / alpha) so that the epilogue is
// stores only.
__device__ __forceinline__ void gemm_glds16(const void* gsrc_lane, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, int TS>
struct G128 {
  static constexpr int kSlabs = sizeof(T) / 4;              // 16-byte DMAs per KC fragment
  // k-blocks per step / ring depth: a step must be short enough for 4 buffers to fit 64 KB (the DMA then
  // runs 3 steps ahead); only the 128-tile double kernel (32 KB per 16-k step) stays at 2 buffers
  static constexpr int kBlocksPerStep = (sizeof(T) == 4 && TS == 64) ? 2 : 1;
  static constexpr int kStepK = 16 * kBlocksPerStep;
  static constexpr int kWE = TS / 2;                         // rows of a wave strip (2 x 2 waves)
  static constexpr int kWT = kWE / 16;                       // MFMA tiles per strip
  static constexpr int kStripBytes = kWE * 16 * (int)sizeof(T);  // one strip, one k-block
  static constexpr int kBlockBytes = 2 * kStripBytes;        // one operand, one k-block
  static constexpr int kOperandBytes = kBlockBytes * kBlocksPerStep;
  static constexpr int kBufBytes = 2 * kOperandBytes;        // A + B of one step
  static constexpr int kInstrPerStrip = kWT * kSlabs;        // DMAs per strip and k-block (KC and RC alike)
  static constexpr int kInstrPerBlock = 2 * kInstrPerStrip;
  static constexpr int kInstrPerStep = kInstrPerBlock * kBlocksPerStep;  // per operand: 16 | 8
  static constexpr int kNbuf = (sizeof(T) == 8 && TS == 128) ? 2 : 4;  // LDS ring depth
};

// Rows (A) / columns (B) of a wave strip are dealt to its kWT MFMA tiles INTERLEAVED: tile x, lane
// index i  <->  strip element kWT * i + x.  A lane then owns kWT consecutive columns of C (16-byte
// accesses to C instead of 4-byte ones) and a row-contiguous operand can be read kWT tiles at a time.
//
// DMA number `ins` of a k-block: element offsets (row, k) this lane fetches 16 bytes from, and the LDS
// byte offset (inside the operand's k-block image) of the wave-wide destination.
//   KC: [tile x][slab][lane] x 16 B -- the lane's own fragment (k = 4 (lane/16) + e).
//   RC: [k (16)][strip rows] elements; a lane brings 16 bytes = consecutive rows at one k, lanes ordered
//       (k, row group) so the image is linear in the lane id.
template <typename T, int TS, bool KC>
__device__ __forceinline__ void g128_dma_coords(int ins, int lane, int& row, int& k, int& lds_off) {
  using G = G128<T, TS>;
  const int s = ins / G::kInstrPerStrip, n = ins % G::kInstrPerStrip;
  if (KC) {
    const int x = n / G::kSlabs, slab = n % G::kSlabs;
    row = s * G::kWE + G::kWT * (lane & 15) + x;
    k = 4 * (lane >> 4) + slab * (4 / G::kSlabs);
  } else {
    constexpr int RPL = 16 / (int)sizeof(T);  // rows per lane
    constexpr int LPK = G::kWE / RPL;         // lanes per k
    constexpr int KPI = 64 / LPK;             // k values per DMA instruction
    row = s * G::kWE + RPL * (lane % LPK);
    k = n * KPI + lane / LPK;
  }
  lds_off = s * G::kStripBytes + n * 1024;
}

// fragments f[x][e] (tile x of strip s, MFMA step e: k = 4 (lane/16) + e) of this lane
template <typename T, int TS, bool KC>
__device__ __forceinline__ void g128_frags(const unsigned char* blk, int s, int lane,
                                           typename Mfma<T>::vec4 (&f)[G128<T, TS>::kWT]) {
  using G = G128<T, TS>;
  using vec4 = typename Mfma<T>::vec4;
  const unsigned char* st = blk + s * G::kStripBytes;
  if constexpr (KC) {
#pragma unroll
    for (int x = 0; x < G::kWT; ++x) {
      if constexpr (sizeof(T) == 4) {
        f[x] = *reinterpret_cast<const vec4*>(st + x * 1024 + lane * 16);
      } else {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 lo = *reinterpret_cast<const f64x2*>(st + x * 2048 + lane * 16);
        const f64x2 hi = *reinterpret_cast<const f64x2*>(st + x * 2048 + 1024 + lane * 16);
        f[x] = vec4{lo[0], lo[1], hi[0], hi[1]};
      }
    }
  } else {
    typedef T vecW __attribute__((ext_vector_type(G::kWT)));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const vecW v = *reinterpret_cast<const vecW*>(
          st + ((4 * (lane >> 4) + e) * G::kWE + G::kWT * (lane & 15)) * (int)sizeof(T));
#pragma unroll
      for (int x = 0; x < G::kWT; ++x) f[x][e] = v[x];
    }
  }
}

// TS = 128 (wave tile 64 x 64) or 64 (32 x 32): the smaller tile when 128-tiles would leave most of the
// chip idle or make a few long-K tiles the critical path.  kmode / lower_only act at TS granularity.
// NBUF LDS buffers form a ring: the DMA of step t + NBUF - 1 is issued while step t computes (a 64-tile
// step is ~0.45 us of MFMA work, less than a DMA round trip, so it runs 3 steps ahead).
template <typename T, int TS, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmDesc g) {
  using M = Mfma<T>;
  using vec4 = typename M::vec4;
  using G = G128<T, TS>;
  constexpr int WT = G::kWT, WE = G::kWE;  // wave tile: WT x WT MFMA tiles, WE rows / columns
  constexpr int NBUF = G::kNbuf;
  typedef T vecW __attribute__((ext_vector_type(WT)));
  extern __shared__ __align__(32) unsigned char lds[];  // [NBUF buffers][A | B][k-block] images
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Tile of this workgroup.  The grid is 1-D over the ACTIVE tiles only and ordered so that (a) the
  // hardware's round-robin of consecutive workgroups over the 8 XCDs deals every XCD the same amount
  // of work -- tiles of equal k-length are neighbours -- and (b) the longest tiles start first:
  //   lower_only      compact triangular enumeration, row-major (k-length depends on the row only)
  //   kmode 1, 4      column-major (k-length depends on tj; 4: last column first)
  //   kmode 3         row-major, last row first (k-length grows with ti)
  //   otherwise       row-major
  // (A 2-D grid whose upper-triangle workgroups exit at once ran the K^-1 product at 78 TFLOP/s
  // against 141 for a full square: XCD 0 got up to 35 % more work than XCD 7.)
  const int bz = blockIdx.z;
  const int nti = g.m / TS, ntj = g.n / TS;
  int ti, tj;
  {
    const int id = (int)blockIdx.x;
    if (g.lower_only) {
      ti = (int)((__builtin_sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
      while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
      while (ti * (ti + 1) / 2 > id) --ti;
      tj = id - ti * (ti + 1) / 2;
    } else if (g.kmode == 1 || g.kmode == 4) {
      tj = id / nti;
      ti = id % nti;
      if (g.kmode == 4) tj = ntj - 1 - tj;
    } else {
      ti = id / ntj;
      tj = id % ntj;
      if (g.kmode == 3) ti = nti - 1 - ti;
    }
  }
  const int m_here = (bz == g.nbatch - 1) ? g.m_last : g.m;
  if (ti * TS >= m_here) return;
  const T* A = static_cast<const T*>(g.A) + (int64_t)bz * g.batchA;
  const T* B = static_cast<const T*>(g.B) + (int64_t)bz * g.batchB;
  T* C = static_cast<T*>(g.C) + (int64_t)bz * g.batchC;

  int k_lo = 0, k_hi = g.k;
  if (g.kmode == 1) k_lo = TS * tj;
  if (g.kmode == 2) k_lo = TS * ti;
  if (g.kmode == 3) k_hi = min(g.k, TS * (ti + 1));
  if (g.kmode == 4) k_hi = min(g.k, TS * (tj + 1));

  // this wave's DMA instructions: numbers wave, wave + 4, ... of the kInstrPerStep per operand
  constexpr int NI = G::kInstrPerStep / 4;  // per wave, operand and step
  const T* a_src[NI];
  const T* b_src[NI];
  int a_dst[NI], b_dst[NI];
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    const int n = wave + 4 * s;
    const int kb = n / G::kInstrPerBlock, ins = n % G::kInstrPerBlock;
    int row, k, off;
    g128_dma_coords<T, TS, A_KC>(ins, lane, row, k, off);
    a_src[s] = A + (int64_t)(ti * TS + row) * g.sai + (int64_t)(k_lo + kb * 16 + k) * g.sak;
    a_dst[s] = kb * G::kBlockBytes + off;
    g128_dma_coords<T, TS, B_KC>(ins, lane, row, k, off);
    b_src[s] = B + (int64_t)(tj * TS + row) * g.sbj + (int64_t)(k_lo + kb * 16 + k) * g.sbk;
    b_dst[s] = G::kOperandBytes + kb * G::kBlockBytes + off;
  }
  const int64_t a_adv = (int64_t)G::kStepK * g.sak, b_adv = (int64_t)G::kStepK * g.sbk;
  auto issue = [&](int buf) {
#pragma unroll
    for (int s = 0; s < NI; ++s) {
      gemm_glds16(a_src[s], lds + buf * G::kBufBytes + a_dst[s]);
      gemm_glds16(b_src[s], lds + buf * G::kBufBytes + b_dst[s]);
      a_src[s] += a_adv;
      b_src[s] += b_adv;
    }
  };

  const int wr = wave >> 1, wc = wave & 1;
  vec4 acc[WT][WT];
#pragma unroll
  for (int d = 0; d < NBUF - 1; ++d)
    if (k_lo + d * G::kStepK < k_hi) issue(d);
  // C element of (tile a, tile b, register r): row c_row(a, r), columns c_col .. c_col + WT - 1 (b)
  T* c_base = C + (int64_t)(ti * TS + wr * WE) * g.ldc + tj * TS + wc * WE + WT * (lane & 15);
  auto c_ptr = [&](int a, int r) { return c_base + (int64_t)(WT * M::crow(lane, r) + a) * g.ldc; };
  if (g.beta != 0.0) {  // all loads of the C tile in flight together, beside the first DMAs
    const T scale = (T)(g.beta / g.alpha);
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const vecW v = *reinterpret_cast<const vecW*>(c_ptr(a, r));
#pragma unroll
        for (int b = 0; b < WT; ++b) acc[a][b][r] = v[b];
      }
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
      for (int b = 0; b < WT; ++b) acc[a][b] *= scale;
  } else {
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
      for (int b = 0; b < WT; ++b) acc[a][b] = vec4{0, 0, 0, 0};
  }
  // s_waitcnt vmcnt(N): everything but the N most recent memory instructions of this wave has landed.
  // With NBUF - 2 newer steps in flight (2 NI DMAs each) that is exactly the current step's data; in
  // the tail (fewer newer steps) wait for everything.
  constexpr int kAhead = (NBUF - 2) * 2 * NI;
  constexpr int kWaitAhead = 0x0f70 | (kAhead & 0xf) | ((kAhead >> 4) << 14);
  int buf = 0;
  for (int k0 = k_lo; k0 < k_hi; k0 += G::kStepK) {
    if (k0 + (NBUF - 2) * G::kStepK < k_hi) __builtin_amdgcn_s_waitcnt(kWaitAhead);
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    // raw s_barrier: __syncthreads() carries a release fence that the compiler lowers to vmcnt(0),
    // which would drain the DMAs running ahead.  After it: everyone's DMA of this step has landed
    // and everyone is done with step - 1.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (k0 + (NBUF - 1) * G::kStepK < k_hi) issue((buf + NBUF - 1) % NBUF);  // into the buffer of step - 1
    const unsigned char* cur = lds + buf * G::kBufBytes;
#pragma unroll
    for (int kb = 0; kb < G::kBlocksPerStep; ++kb) {
      vec4 a4[WT], b4[WT];
      g128_frags<T, TS, A_KC>(cur + kb * G::kBlockBytes, wr, lane, a4);
      g128_frags<T, TS, B_KC>(cur + G::kOperandBytes + kb * G::kBlockBytes, wc, lane, b4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < WT; ++a)
#pragma unroll
          for (int b = 0; b < WT; ++b) acc[a][b] = M::mma(a4[a][e], b4[b][e], acc[a][b]);
    }
    buf = (buf + 1 == NBUF) ? 0 : buf + 1;
  }
  const T alpha = (T)g.alpha;
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vecW v;
#pragma unroll
      for (int b = 0; b < WT; ++b) v[b] = alpha * acc[a][b][r];
      *reinterpret_cast<vecW*>(c_ptr(a, r)) = v;
      if constexpr (sizeof(T) == 4) {
        if (g.split != nullptr) {  // the same WT values as bf16 pieces (x = h0 + h1 + h2), pairs of columns
          const int64_t row = (int64_t)ti * TS + wr * WE + WT * M::crow(lane, r) + a;
          const int col = tj * TS + wc * WE + WT * (lane & 15);
#pragma unroll
          for (int b = 0; b < WT; b += 2) {
            float x0 = v[b], x1 = v[b + 1];
            unsigned short* dst = g.split + syrk_plane_offset(g.split_row0 + row, (int)g.split_col0 + col + b, g.split_nkb);
            if (g.split_np == 2) {
              x0 *= g.split_scale;
              x1 *= g.split_scale;
#pragma unroll
              for (int p = 0; p < 2; ++p) *reinterpret_cast<unsigned*>(dst + p * g.split_stride) = f16_split_pair(x0, x1);
            } else {
#pragma unroll
              for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(dst + p * g.split_stride) = bf16_split_pair(x0, x1);
            }
          }
        }
      }
    }
}

template <typename T, int TS, bool A_KC, bool B_KC>
static void launch_gemm128(hipStream_t st, const GemmDesc& g) {
  constexpr int kLds = G128<T, TS>::kNbuf * G128<T, TS>::kBufBytes;
  if (ensure_dyn_lds(reinterpret_cast<const void*>(&gemm128_kernel<T, TS, A_KC, B_KC>), kLds)) return;
  const int64_t nti = g.m / TS, ntj = g.n / TS;
  const int64_t ntiles = g.lower_only ? nti * (nti + 1) / 2 : nti * ntj;  // lower_only: m == n
  const dim3 grid((unsigned)ntiles, 1, (unsigned)g.nbatch);
  hipLaunchKernelGGL((gemm128_kernel<T, TS, A_KC, B_KC>), grid, dim3(256), kLds, st, g);
}

template <typename T, int TS>
static void launch_gemm_dma(hipStream_t st, const GemmDesc& g, bool a_kc, bool b_kc) {
  if (a_kc && b_kc) launch_gemm128<T, TS, true, true>(st, g);
  else if (a_kc) launch_gemm128<T, TS, true, false>(st, g);
  else if (b_kc) launch_gemm128<T, TS, false, true>(st, g);
  else launch_gemm128<T, TS, false, false>(st, g);
}

// the tile size launch_gemm picks for a launch of these extents (parts of a split launch pass it on as ts_hint)
static int gemm_tile_choice(int m, int n, int m_last, int k, int nbatch, bool lower_only) {
  const bool div128 = (m % 128 == 0) && (n % 128 == 0) && (m_last % 128 == 0) && (k % 128 == 0);
  const int64_t tiles128 = (int64_t)(m / 128) * (n / 128) * nbatch / (lower_only ? 2 : 1);
  return (div128 && tiles128 >= 512) ? 128 : 64;
}

template <typename T>
static void launch_gemm(hipStream_t st, const GemmDesc& g) {
  if (g.m <= 0 || g.n <= 0 || g.nbatch <= 0) return;
  const bool a_kc = (g.sak == 1), a_rc = (g.sai == 1), b_kc = (g.sbk == 1), b_rc = (g.sbj == 1);
  const bool dma_ok = (a_kc || a_rc) && (b_kc || b_rc) && g.alpha != 0.0 && g.k % 64 == 0 &&
                      (!g.lower_only || g.m == g.n);
  // 128 x 128 tiles when every extent allows it and they still give every CU about two tiles
  // (lower_only launches compute only half of the grid); 64 x 64 tiles otherwise
  const bool div128 = (g.m % 128 == 0) && (g.n % 128 == 0) && (g.m_last % 128 == 0) && (g.k % 128 == 0);
  const int64_t tiles128 = (int64_t)(g.m / 128) * (g.n / 128) * g.nbatch / (g.lower_only ? 2 : 1);
  // every product of the fit has unit-stride operands, k a multiple of 64 and alpha != 0 (checked
  // once here: anything else is a programming error, reported through the launch-error state)
  if (!dma_ok) {
    note_launch_error("launch_gemm: operand strides / sizes the LDS-DMA tile kernel cannot take");
    return;
  }
  if (g.ts_hint == 128 ? div128 : (g.ts_hint == 0 && div128 && tiles128 >= 512)) launch_gemm_dma<T, 128>(st, g, a_kc, b_kc);
  else launch_gemm_dma<T, 64>(st, g, a_kc, b_kc);
}

// =============================================================================================
// rank-W update of the trailing matrix on the bf16 matrix cores (float fits, two-level path)
// =============================================================================================
// C (lower 128x128 tiles of an m x m block) -= A A^T, A = the panel L21 (m x W) just produced by the
// TRSM GEMM -- whose epilogue also wrote it as THREE bf16 planes (x = h0 + h1 + h2, each piece the bf16
// rounding of the remainder: 3 x 8 mantissa bits hold a float exactly).  The product is recovered from six
// v_mfma_f32_16x16x32_bf16 per tile pair (h0h0 + h0h1 + h1h0 + h1h1 + h0h2 + h2h0, small terms first, f32
// accumulation: the dropped h1h2 / h2h1 / h2h2 are <= 2^-24 relative) -- the bf16 pipe runs at 16x the
// f32 MFMA rate, so the six cost 3/8 of the eight f32 MFMAs they replace.  Both operands are rows of
// the same k-contiguous planes, so no value is split inside this kernel: fragments (8 consecutive k of one
// row = 16 bytes per lane) go global -> LDS by global_load_lds into a 3-deep ring of 48 KB steps (K = 32
// per step: 2 operands x 3 pieces x 8 row tiles x 1 KB), one workgroup of four waves per CU, each wave a
// 64 x 64 corner (4 x 4 MFMA tiles, strip rows dealt to the tiles interleaved so that a lane owns four
// consecutive columns of C: 16-byte accesses, as in gemm128_kernel).
// Plane layout (private to the TRSM epilogue that writes it and this kernel): the planes are stored as the
// 1 KB fragments this kernel's MFMAs consume, so every LDS-DMA instruction reads 1 KB of contiguous memory:
// fragment (T, kb) of a plane = rows of MFMA tile T x the 32 k of block kb, lane l = (i, kg) holding
// k = 32 kb + 8 kg .. + 7 of the tile's row i; tile T = 4 (row / 64) + row % 4 with i = (row % 64) / 4
// (the four tiles of a 64-row strip are dealt its rows interleaved, so that a lane owns four consecutive
// columns of C).  Element (row, k) of plane p: p * plane_stride + syrk_plane_offset(row, k, nkb).
__host__ __device__ __forceinline__ int64_t syrk_plane_offset(int64_t row, int k, int nkb) {
  const int64_t T = 4 * (row >> 6) + (row & 3);
  const int i = (int)((row & 63) >> 2);
  return ((T * nkb + (k >> 5)) * 64 + (((k & 31) >> 3) * 16 + i)) * 8 + (k & 7);
}
// A float matrix as three bf16 planes in that layout (nkb = 32-column blocks per row of the layout).
struct Bf16Planes {
  unsigned short* p;  // plane 0 (nullptr: absent)
  int64_t stride;     // bf16 elements between planes
  int nkb;
  float scale = 1.0f;  // fp16 pieces (GemmBf16Desc::np == 2): the planes hold value * scale, a power of two
};
// super-tile: tile rows x tile columns that one XCD works on at a time (square for the triangular enumeration);
// filled by launch_gemm_bf16.  mg_*: q = n / d as __umulhi(n, mg) (exact for n < 2^20, 1 < d <= 4096).
struct BfTiling {
  int nti, ntj, sr, sc, nsi, nsj, per_super, nsuper, total_slots;
  unsigned mg_per_super, mg_nbatch, mg_nsj, mg_sc;
};
// C (m x n) = alpha * Aop Bop^T + beta * C with Aop = rows a_row0.. / columns a_col0.. of the matrix in planes A
// (m x k) and Bop likewise (n x k): both operands k-contiguous rows, so a transposed factor is simply the
// planes of the transposed matrix.  Row offsets are multiples of 64, column offsets of 32; m, n multiples of
// 128, k of 64 and every tile's k-range a multiple of 64, at least 128.  The result goes to C (float, nullable) and / or to planes:
// `out` at (o_row0, o_col0), `out_t` -- the TRANSPOSE of the result -- at (ot_row0, ot_col0).  Batch entry z
// shifts every row and column offset by z * batch_shift and C along its diagonal.
struct GemmBf16Desc {
  Bf16Planes A, B;
  int64_t a_row0, a_col0, b_row0, b_col0;
  float* C;
  int64_t ldc;
  int m, n, k;
  float alpha;
  int beta;        // 0 | 1
  int lower_only;  // only tiles tj <= ti (m == n)
  int kmode;       // 0: all k | 2: k >= 128 ti | 3: k < 128 (ti + 1)
  int nbatch;
  int64_t batch_shift;
  Bf16Planes out, out_t;
  int64_t o_row0, o_col0, ot_row0, ot_col0;
  BfTiling tl;
  // pieces per value: 3 = bf16 (six MFMAs per product, h0h0 + h0h1 + h1h0 + h1h1 + h0h2 + h2h0), 2 = fp16 pieces of the
  // scaled values (THREE MFMAs per product, h1h0 + h0h1 + h0h0; the dropped h1h1 is 2^-22 relative): round 5.  alpha
  // is the caller's; the kernel divides the scales of A and B out (exact: powers of two)
  int np;
};
constexpr int kSyrkBf16MinRows = 3072;  // below: too few 128-tiles for one workgroup per CU to pay (measured)
constexpr int kSyrkNbuf = 3;
constexpr int syrk_step_bytes(int np) { return 2 * np * 8 * 1024; }  // [operand][piece][row tile] x 1 KB
constexpr int syrk_lds_bytes(int np) { return kSyrkNbuf * syrk_step_bytes(np) + 4 * 1024; }  // + a spare KB per wave (C prefetch target)

#ifndef GPSO_GSTAMP
#define GPSO_GSTAMP(tile, i)  // tools/micro/syrk_bench.hip defines this to record s_memtime stamps per tile
#endif
static unsigned bf16_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }
static BfTiling bf16_tiling(const GemmBf16Desc& g) {
  BfTiling tl;
  tl.nti = g.m / 128;
  tl.ntj = g.n / 128;
  tl.sr = std::min(4, tl.nti);
  tl.sc = std::min(g.lower_only ? 4 : 8, tl.ntj);
  tl.nsi = (tl.nti + tl.sr - 1) / tl.sr;
  tl.nsj = (tl.ntj + tl.sc - 1) / tl.sc;
  tl.per_super = tl.sr * tl.sc;
  tl.nsuper = g.lower_only ? tl.nsi * (tl.nsi + 1) / 2 : tl.nsi * tl.nsj;
  tl.total_slots = (tl.nsuper * g.nbatch + 7) / 8 * 8 * tl.per_super;
  tl.mg_per_super = bf16_magic(tl.per_super);
  tl.mg_nbatch = bf16_magic(g.nbatch);
  tl.mg_nsj = bf16_magic(tl.nsj);
  tl.mg_sc = bf16_magic(tl.sc);
  return tl;
}
struct BfTile {
  int ti, tj, bz, k_lo, nsteps;
};

// s_waitcnt vmcnt(N) lgkmcnt(0), N a multiple of 12 (DMA groups of one wave), 63 = no vector-memory wait
template <int N>
__device__ __forceinline__ void bf16_wait() {
  __builtin_amdgcn_s_waitcnt(0x0070 | (N & 0xf) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

// The kernel is PERSISTENT: one workgroup of four waves per CU walks slots blockIdx.x, + gridDim.x, ... of a 1-D
// sequence of tile slots, and its k-loop is ONE software pipeline across all of its tiles (s_memtime stamps of the
// one-tile-per-workgroup form: 14k of a tile's ~68k clocks were decode, address set-up, the first DMA round trip
// and the C loads, with nothing else resident on the CU to hide them).
//  * slots: super-tiles of 4 x 8 tiles (4 x 4 in the triangular case); super-tile s runs on XCD s % 8 (the
//    hardware deals consecutive workgroups to the XCDs round-robin and gridDim.x is a multiple of 8), so the 32
//    workgroups of an XCD read 12-16 operand tiles between them at a time and most LDS-DMA traffic is served by
//    that XCD's L2.  The decode has no loops, tables or divisions.
//  * pipeline: item e = (tile, k-step).  While the 96 MFMAs of item e run on the fragments in one register set,
//    the wave reads the fragments of item e + 1 from LDS into the other set and issues the DMAs of item e + 3
//    into the buffer item e was read from -- one DMA or LDS read behind each MFMA (sched_barrier pins that: the
//    12 DMA instructions of a step occupy the CU's address path for ~770 clocks, and issued in one burst they
//    stalled the wave for ~600 of the ~1540 clocks of its MFMAs).  Items e + 1 .. e + 3 may belong to the NEXT
//    tile: its first three steps are in flight while this tile's accumulators are stored.
template <int NP>
__global__ __launch_bounds__(256, 1) void gemm_bf16_kernel(GemmBf16Desc g) {
  typedef float vecW __attribute__((ext_vector_type(4)));
  constexpr int kSyrkStepBytes = syrk_step_bytes(NP);
  constexpr int ND = 4 * NP;  // DMA instructions per wave and step (12 | 8)
  constexpr int NR = 8 * NP;  // fragment reads per wave and step (24 | 16)
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const BfTiling& tl = g.tl;
  auto fdiv = [](unsigned n, int d, unsigned mg) { return d == 1 ? n : __umulhi(n, mg); };
  auto decode = [&](int id, BfTile& t) -> bool {
    const unsigned x = (unsigned)id & 7u, j = (unsigned)id >> 3;
    const unsigned js = fdiv(j, tl.per_super, tl.mg_per_super), tt = j - js * tl.per_super;
    const unsigned sg = js * 8 + x;
    if (sg >= (unsigned)(tl.nsuper * g.nbatch)) return false;
    const unsigned sl = fdiv(sg, g.nbatch, tl.mg_nbatch);
    t.bz = (int)(sg - sl * g.nbatch);
    int I, J;
    if (g.lower_only) {  // compact triangular enumeration of the super-tiles, row-major
      I = (int)((__builtin_sqrtf(8.0f * (float)sl + 1.0f) - 1.0f) * 0.5f);
      while ((I + 1) * (I + 2) / 2 <= (int)sl) ++I;
      while (I * (I + 1) / 2 > (int)sl) --I;
      J = (int)sl - I * (I + 1) / 2;
    } else {
      I = (int)fdiv(sl, tl.nsj, tl.mg_nsj);
      J = (int)sl - I * tl.nsj;
      if (g.kmode == 3) I = tl.nsi - 1 - I;  // the longest k-ranges first (kmode 2: small ti, kmode 3: large ti)
    }
    const unsigned tr = fdiv(tt, tl.sc, tl.mg_sc);
    t.ti = I * tl.sr + (int)tr;
    t.tj = J * tl.sc + (int)(tt - tr * tl.sc);
    if (t.ti >= tl.nti || t.tj >= tl.ntj || (g.lower_only && t.tj > t.ti)) return false;
    int k_lo = 0, k_hi = g.k;
    if (g.kmode == 2) k_lo = 128 * t.ti;
    if (g.kmode == 3) k_hi = min(g.k, 128 * (t.ti + 1));
    t.k_lo = k_lo;
    t.nsteps = (k_hi - k_lo) / 32;
    return true;
  };
  int slot = (int)blockIdx.x - (int)gridDim.x;
  auto next_tile = [&](BfTile& t) -> bool {
    for (slot += (int)gridDim.x; slot < tl.total_slots; slot += (int)gridDim.x)
      if (decode(slot, t)) return true;
    return false;
  };

  // this wave's DMA pieces: numbers wave, wave + 4, ... of the 48 per step; piece q = (operand, plane, tile),
  // tile = 4 (64-row strip of the 128 rows) + x: 1 KB of contiguous memory each.  Source of piece s = the
  // cursor's per-operand base (a scalar register pair that advances 1 KB per step) + voff[s], a per-lane
  // 32-bit offset that never changes (plane, tile and lane; three planes of a 16384^2 matrix span 1.5 GB).
  unsigned voff[ND];
  int dst[ND];
#pragma unroll
  for (int s = 0; s < ND; ++s) {
    const int q = wave + 4 * s, op = q / (8 * NP), p = (q % (8 * NP)) / 8, tile = q % 8;
    const Bf16Planes& P = (op == 0) ? g.A : g.B;
    voff[s] = (unsigned)(((int64_t)p * P.stride + (int64_t)tile * P.nkb * 512) * 2) + (unsigned)lane * 16u;
    dst[s] = q * 1024;
  }
  uint64_t src[2];  // the DMA cursor: next 32-k block, per operand
  auto set_src = [&](const BfTile& t) {
    const int64_t zs = (int64_t)t.bz * g.batch_shift;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const Bf16Planes& P = (op == 0) ? g.A : g.B;
      const int64_t row0 = (op == 0 ? g.a_row0 + (int64_t)t.ti * 128 : g.b_row0 + (int64_t)t.tj * 128) + zs;
      const int64_t col0 = (op == 0 ? g.a_col0 : g.b_col0) + zs + t.k_lo;
      src[op] = reinterpret_cast<uint64_t>(P.p) + (uint64_t)(((4 * (row0 >> 6)) * P.nkb + (col0 >> 5)) * 1024);
    }
  };
  // The DMA is issued as two instructions in two different MFMA gaps -- M0 (the LDS destination), then the load
  // with the scalar base + per-lane offset form -- because each gap hides ~8 clocks of other instructions and the
  // compiler's form of this (per-lane 64-bit addresses: s_mov, 64-bit add, M0, load in one gap) made a group of
  // six MFMAs take ~133 clocks instead of 96.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma_m0 = [&](int s, int buf) {
    const unsigned a = lds0 + (unsigned)(buf * kSyrkStepBytes + dst[s]);
    asm volatile("s_mov_b32 m0, %0" ::"s"(a) : "memory");
  };
  auto dma_go = [&](int s) {
    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff[s]), "s"(src[s / (2 * NP)]) : "memory");
  };

  u32x4 F[2][2][4][NP];  // fragments: [set][operand][tile][piece]
  auto frag_ptr = [&](int buf, int r) {  // read r of a step: operand r / (4 NP), tile (r % (4 NP)) / NP, piece r % NP
    const int op = r / (4 * NP), x = (r % (4 * NP)) / NP, p = r % NP;
    return reinterpret_cast<const u32x4*>(lds + buf * kSyrkStepBytes + ((op * NP + p) * 8 + (op == 0 ? wr : wc) * 4 + x) * 1024 + lane * 16);
  };
  auto wait_groups = [&](int newer) {  // every DMA group of this wave but the newest `newer`, and every LDS read
    if (newer >= 2) bf16_wait<2 * ND>();
    else if (newer == 1) bf16_wait<ND>();
    else bf16_wait<0>();
  };

  BfTile cur, nxt;
  if (!next_tile(cur)) return;
  bool have_next = false, switched = false;
  int rem;        // DMA groups the cursor has left in its tile
  int ahead = 2;  // DMA groups issued beyond the current item
  int buf = 0;
  set_src(cur);
  rem = cur.nsteps - kSyrkNbuf;
#pragma unroll
  for (int d = 0; d < kSyrkNbuf; ++d)
#pragma unroll
    for (int s = 0; s < ND; ++s) {
      dma_m0(s, d);
      asm volatile("s_nop 0");
      dma_go(s);
      if (s == ND - 1) src[0] += 1024, src[1] += 1024;
    }
  wait_groups(2);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int r = 0; r < NR; ++r) F[0][r / (4 * NP)][(r % (4 * NP)) / NP][r % NP] = *frag_ptr(0, r);
  const float alpha_eff = g.alpha / (g.A.scale * g.B.scale);  // (scales are powers of two: exact)

  f32x4 acc[4][4];
  float* c_base = nullptr;
  int64_t c_row0 = 0;  // element offset of this wave's 64 x 64 corner of C
  bool with_c = false, first_tile = true;
  auto c_ptr = [&](int a, int r) { return c_base + (int64_t)(4 * (4 * (lane >> 4) + r) + a) * g.ldc; };

  // kSteady: neither among a tile's first two nor its last three steps -- reads and DMAs unconditional, no branches
  auto step_body = [&](auto setc, auto steadyc, int st) {
    constexpr int S = decltype(setc)::value;
    constexpr bool kSteady = decltype(steadyc)::value;
    if (!kSteady && rem == 0 && have_next && !switched) {  // the cursor moves on to the next tile
      set_src(nxt);
      rem = nxt.nsteps;
      switched = true;
    }
    const bool do_dma = kSteady || rem > 0;
    const bool do_rd = kSteady || st + 1 < cur.nsteps || have_next;
    // item e + 1 has landed (the reads below need it); the fragments of this item, read during the last one, are
    // in registers on every wave once all have passed the barrier, so its buffer may be overwritten.  (Step 0 of
    // a later tile: that wait was made before the stores of the previous tile's epilogue, see below.)
    if (kSteady) bf16_wait<ND>();
    else if (st == 0 && !first_tile) bf16_wait<63>();
    else wait_groups(ahead - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!kSteady && with_c && st == cur.nsteps - 1) {
      // touch this wave's 64 rows x 2 lines of C (results discarded): the epilogue's loads then hit the L2 instead
      // of paying an HBM round trip with nothing to hide it.  Issued before this step's DMAs: the counted waits
      // look at the newest instructions only.
      // (as LDS-DMA into a spare kilobyte of LDS per wave: no register receives data the compiler knows nothing of)
      const float* row = g.C + c_row0 + (int64_t)lane * g.ldc;
      const unsigned spare = lds0 + (unsigned)(kSyrkNbuf * kSyrkStepBytes + wave * 1024);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\tglobal_load_lds_dword %1, off offset:128" ::"s"(spare), "v"(row) : "memory");
    }
    const int buf_rd = (buf + 1 == kSyrkNbuf) ? 0 : buf + 1;
    auto read_frag = [&](int r) {
      if (do_rd) F[1 - S][r / (4 * NP)][(r % (4 * NP)) / NP][r % NP] = *frag_ptr(buf_rd, r);
    };
#define GPSO_SY(PA, PB)                                                                                                                \
  c = NP == 2 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, F[S][0][a][PA]), __builtin_bit_cast(f16x8, F[S][1][b][PB]), c, 0, 0, 0) \
              : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[S][0][a][PA]), __builtin_bit_cast(bf16x8, F[S][1][b][PB]), c, 0, 0, 0)
#define GPSO_SLOT __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {  // one MFMA, then at most one other instruction: nothing else ever queues up
      const int a = ch / 4, b = ch % 4;
      f32x4 c = acc[a][b];
      if constexpr (NP == 3) {
        GPSO_SY(NP - 1, 0);
        GPSO_SLOT;
        if (ch < 12 && do_dma) dma_m0(ch, buf);
        GPSO_SLOT;
        GPSO_SY(0, NP - 1);
        GPSO_SLOT;
        if (ch < 12 && do_dma) dma_go(ch);
        GPSO_SLOT;
        GPSO_SY(1, 1);
        GPSO_SLOT;
        read_frag(ch);  // reads 0..15, one per group
        GPSO_SLOT;
        GPSO_SY(1, 0);
        GPSO_SLOT;
        if (ch < 8) read_frag(16 + ch);  // reads 16..23
        GPSO_SLOT;
        GPSO_SY(0, 1);
        GPSO_SLOT;
        GPSO_SY(0, 0);
        GPSO_SLOT;
      } else {  // fp16 pieces: three products, small terms first; 8 DMAs and 16 reads per step in the 48 gaps
        GPSO_SY(1, 0);
        GPSO_SLOT;
        if (ch < ND && do_dma) dma_m0(ch, buf);
        GPSO_SLOT;
        GPSO_SY(0, 1);
        GPSO_SLOT;
        if (ch < ND && do_dma) dma_go(ch);
        GPSO_SLOT;
        GPSO_SY(0, 0);
        GPSO_SLOT;
        read_frag(ch);  // reads 0..15
        GPSO_SLOT;
      }
      acc[a][b] = c;
    }
#undef GPSO_SLOT
#undef GPSO_SY
    if (do_dma) --rem, src[0] += 1024, src[1] += 1024;
    else --ahead;
    buf = buf_rd;
  };
  const std::integral_constant<int, 0> set0;
  const std::integral_constant<int, 1> set1;

  int tile_no = 0;
  for (;; ++tile_no) {
    GPSO_GSTAMP(tile_no, 0);
    have_next = next_tile(nxt);
    switched = false;
    const int64_t zs = (int64_t)cur.bz * g.batch_shift;
    const int row_w = cur.ti * 128 + wr * 64, col_w = cur.tj * 128 + wc * 64 + 4 * (lane & 15);  // this lane's rows / columns in C
    c_row0 = zs * g.ldc + zs + (int64_t)row_w * g.ldc + cur.tj * 128 + wc * 64;
    c_base = (g.C != nullptr) ? g.C + zs * g.ldc + zs + (int64_t)row_w * g.ldc + col_w : nullptr;
    with_c = g.beta != 0 && c_base != nullptr;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    // (every k-range is a multiple of 64, so a tile starts on register set 0 and its last step leaves the next
    // tile's first fragments there: at the tile boundary only that set is live)
    int st = 0;
    step_body(set0, std::false_type{}, st);
    step_body(set1, std::false_type{}, st + 1);
    for (st = 2; st + 4 < cur.nsteps; st += 2) {
      step_body(set0, std::true_type{}, st);
      step_body(set1, std::true_type{}, st + 1);
    }
    for (; st < cur.nsteps; st += 2) {
      step_body(set0, std::false_type{}, st);
      step_body(set1, std::false_type{}, st + 1);
    }
    GPSO_GSTAMP(tile_no, 1);
    // Step 0 of the next tile reads item e + 1 = its step 1: wait for that group HERE, before this tile's stores
    // enter the queue (only DMAs are newer, so the count is exact whatever order loads and stores retire in).
    if (have_next) wait_groups(ahead - 1);
    vecW cin[4][4];  // (the register set whose fragments this tile's last step consumed is free now)
    if (with_c) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) cin[a][r] = *reinterpret_cast<const vecW*>(c_ptr(a, r));
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        vecW v;
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = alpha_eff * acc[a][b][r] + (with_c ? cin[a][r][b] : 0.0f);
        if (c_base != nullptr) *reinterpret_cast<vecW*>(c_ptr(a, r)) = v;
        const int64_t row = row_w + 4 * (4 * (lane >> 4) + r) + a;  // (row, col_w .. col_w + 3) of the result
        if (g.out.p != nullptr) {
#pragma unroll
          for (int b = 0; b < 4; b += 2) {
            float x0 = v[b] * g.out.scale, x1 = v[b + 1] * g.out.scale;
            unsigned short* d = g.out.p + syrk_plane_offset(g.o_row0 + zs + row, (int)(g.o_col0 + zs) + col_w + b, g.out.nkb);
#pragma unroll
            for (int p = 0; p < NP; ++p)
              *reinterpret_cast<unsigned*>(d + p * g.out.stride) = NP == 2 ? f16_split_pair(x0, x1) : bf16_split_pair(x0, x1);
          }
        }
        if (g.out_t.p != nullptr) {  // element (row, col) of the result is element (col, row) of the transpose
#pragma unroll
          for (int b = 0; b < 4; b += 2) {
            float x0 = v[b] * g.out_t.scale, x1 = v[b + 1] * g.out_t.scale;
            unsigned short* d0 = g.out_t.p + syrk_plane_offset(g.ot_row0 + zs + col_w + b, (int)(g.ot_col0 + zs + row), g.out_t.nkb);
            unsigned short* d1 = g.out_t.p + syrk_plane_offset(g.ot_row0 + zs + col_w + b + 1, (int)(g.ot_col0 + zs + row), g.out_t.nkb);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const unsigned u = NP == 2 ? f16_split_pair(x0, x1) : bf16_split_pair(x0, x1);
              d0[p * g.out_t.stride] = (unsigned short)(u & 0xffffu);
              d1[p * g.out_t.stride] = (unsigned short)(u >> 16);
            }
          }
        }
      }
    GPSO_GSTAMP(tile_no, 2);
    if (!have_next) break;
    cur = nxt;
    first_tile = false;
  }
}

static int bf16_gemm_grid(int total_slots) {
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n / 8 * 8 > 0 ? n / 8 * 8 : 8;  // a multiple of the 8 XCDs
  }
  return std::min(total_slots, cus[dev]);
}
// reserve_cus: compute units left to another stream (a workgroup of this kernel fills a CU's LDS, so that many
// CUs stay free for whatever else is in flight)
static void launch_gemm_bf16(hipStream_t st, GemmBf16Desc g, int reserve_cus = 0) {
  if (g.np != 2) g.np = 3;
  const int lds = syrk_lds_bytes(g.np);
  const void* fn = g.np == 2 ? reinterpret_cast<const void*>(&gemm_bf16_kernel<2>) : reinterpret_cast<const void*>(&gemm_bf16_kernel<3>);
  if (ensure_dyn_lds(fn, lds)) return;
  g.tl = bf16_tiling(g);
  int grid = bf16_gemm_grid(g.tl.total_slots);
  if (reserve_cus > 0) grid = std::max(8, std::min(grid, (bf16_gemm_grid(INT_MAX) - reserve_cus) / 8 * 8));
  if (g.np == 2) hipLaunchKernelGGL(gemm_bf16_kernel<2>, dim3((unsigned)grid), dim3(256), lds, st, g);
  else hipLaunchKernelGGL(gemm_bf16_kernel<3>, dim3((unsigned)grid), dim3(256), lds, st, g);
}
bool fit_plane_scales(double variance, double noise, FitPlanes& pl) {
  // scaled bound 2^13; the scale itself must leave a typical entry's SECOND piece (2^-11 of it) a normal fp16 number
  auto scale_for = [](double bound) { return std::ldexp(1.0, 13 - (int)std::ceil(std::log2(std::max(bound, 1e-300)))); };
  if (!(variance > 0.0) || !(noise > 0.0)) return false;
  const double sl = scale_for(std::sqrt(variance + noise)), sx = scale_for(1.0 / std::sqrt(noise)),
               sw = scale_for(std::sqrt((variance + noise) / noise));
  // typical entries: |L| ~ sqrt(variance), |L^-1| and the W blocks ~ 1 / sqrt(variance) and up
  const double lo = std::ldexp(1.0, -2), hi = std::ldexp(1.0, 40);
  if (sl * std::sqrt(variance) < lo || sx / std::sqrt(variance) < lo || sw < lo * 0.25 || sl > hi || sx > hi || sw > hi) return false;
  pl.sL = (float)sl;
  pl.sX = (float)sx;
  pl.sW = (float)sw;
  pl.np = 2;
  return true;
}

// a float block (rows x cols at src, leading dimension ld) -> planes at (row0, col0) and / or its transpose at
// (trow0, tcol0): the diagonal blocks of L^-1, which the step kernels produce in float
__global__ __launch_bounds__(256) void block_to_planes_kernel(const float* __restrict__ src, int64_t ld, int rows,
                                                              int cols, int64_t batch_src, int64_t batch_shift,
                                                              Bf16Planes out, int64_t row0, int64_t col0,
                                                              Bf16Planes out_t, int64_t trow0, int64_t tcol0, int np) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index inside a block
  const int per_row = cols / 2;
  if (idx >= (int64_t)rows * per_row) return;
  const int64_t zs = (int64_t)blockIdx.y * batch_shift;
  const int64_t r = idx / per_row;
  const int c = 2 * (int)(idx % per_row);
  const float* s = src + (int64_t)blockIdx.y * batch_src + r * ld + c;
  float a = s[0], b = s[1];
  if (np == 2) {  // (both plane sets of a call share their scale)
    const float sc = out.p != nullptr ? out.scale : out_t.scale;
    a *= sc;
    b *= sc;
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    if (p >= np) break;
    const unsigned u = np == 2 ? f16_split_pair(a, b) : bf16_split_pair(a, b);
    if (out.p != nullptr)
      *reinterpret_cast<unsigned*>(out.p + p * out.stride + syrk_plane_offset(row0 + zs + r, (int)(col0 + zs) + c, out.nkb)) = u;
    if (out_t.p != nullptr) {
      out_t.p[p * out_t.stride + syrk_plane_offset(trow0 + zs + c, (int)(tcol0 + zs + r), out_t.nkb)] = (unsigned short)(u & 0xffffu);
      out_t.p[p * out_t.stride + syrk_plane_offset(trow0 + zs + c + 1, (int)(tcol0 + zs + r), out_t.nkb)] = (unsigned short)(u >> 16);
    }
  }
}

// float matrix (rows x cols, leading dimension ld) -> planes of its own (tools/micro/syrk_bench.hip)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int64_t ld, int64_t rows,
                                                           int64_t cols, unsigned short* __restrict__ planes,
                                                           int64_t plane_stride) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index
  const int64_t per_row = cols / 2;
  if (idx >= rows * per_row) return;
  const int64_t r = idx / per_row, c = 2 * (idx % per_row);
  float a = src[r * ld + c], b = src[r * ld + c + 1];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const unsigned u = bf16_split_pair(a, b);
    *reinterpret_cast<unsigned*>(planes + p * plane_stride + syrk_plane_offset(r, (int)c, (int)(cols / 32))) = u;
  }
}

// =============================================================================================
// look-ahead Cholesky step: ONE launch per 64-column step, diagonal chain and bulk update overlap
// =============================================================================================
// The 64x64 diagonal factorisation is a chain of pivots that one workgroup runs for ~20 us while
// the rest of the chip idles, and at N = 2048 that chain was half of the fit.  Here the launch that
// follows diagonal block k does both
//   role D  (workgroup 0):  diagonal block k+1: its own panel tile L10 = K[k+1,k] X_k^T, the
//                           rank-64 update of ITS tile only, S = K[k+1,k+1] - L10 L10^T, then
//                           chol(S) -> Lf[k+1,k+1] and its inverse -> linv[k+1,k+1];
//   role PU (all others):   tile (i,j), k+1 <= j <= jmax, i >= j:
//                           Li = K[i,k] X_k^T, Lj = K[j,k] X_k^T (recomputed per tile: 64^3 each),
//                           K[i,j] -= Li Lj^T; the j == k+1 tiles also store Li -> Lf[i,k]
//                           (tile (k+1,k+1) only stores: its update is role D's).
// Both roles only READ column k of K and WRITE disjoint tiles, so no ordering inside the launch is
// needed; L goes to its own matrix Lf (out of place) because other tiles still read K[i,k] while
// Li is produced.  K ends up holding Schur-complement debris.
constexpr int kTL = kFitBlock + 4;  // LDS row stride of a T tile: vec4 fragment reads conflict-free

// wave tile product on MFMA: acc[tj] (16x16, rows 16w.. of A) = A_rows[16 x 64] * B[64 x 64]^T,
// both operands row-major [row][k] in LDS with stride kTL.  k runs in the permuted order
// 16 kk + 4 (lane >> 4) + e on both sides.
// B_LOWER: B is lower triangular (B[row][k] = 0 for k > row), so column tile tj only needs the
// k-blocks kk <= tj: 40 instead of 64 MFMAs.
template <typename T, bool B_LOWER>
__device__ __forceinline__ void mma_abt(const T* A_rows, const T* B, int lane,
                                        typename Mfma<T>::vec4 (&acc)[4]) {
  using M = Mfma<T>;
  using vec4 = typename M::vec4;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) acc[tj] = vec4{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int ko = 16 * kk + 4 * (lane >> 4);
    const vec4 a4 = *reinterpret_cast<const vec4*>(A_rows + (lane & 15) * kTL + ko);
    vec4 b4[4];
#pragma unroll
    for (int tj = B_LOWER ? kk : 0; tj < 4; ++tj)
      b4[tj] = *reinterpret_cast<const vec4*>(B + (16 * tj + (lane & 15)) * kTL + ko);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int tj = B_LOWER ? kk : 0; tj < 4; ++tj) acc[tj] = M::mma(a4[e], b4[tj][e], acc[tj]);
  }
}

// wave tile product with both operands stored [k][row] (row-major tiles of X read "down the
// columns"): acc[tj][a][b] = sum_m A[m][a0 + a] * B[m][16 tj + b].  Fragments are 4-byte column
// reads (stride kTL = 68 words: the four k groups of a wave land 16 banks apart, conflict-free).
template <typename T>
__device__ __forceinline__ void mma_atb(const T* A, int a0, const T* B, int lane,
                                        typename Mfma<T>::vec4 (&acc)[4]) {
  using M = Mfma<T>;
  using vec4 = typename M::vec4;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) acc[tj] = vec4{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    T a[4], b[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = 16 * kk + 4 * (lane >> 4) + e;
      a[e] = A[m * kTL + a0 + (lane & 15)];
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) b[tj][e] = B[m * kTL + 16 * tj + (lane & 15)];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) acc[tj] = M::mma(a[e], b[tj][e], acc[tj]);
  }
}

// all 256 threads: 64x64 tile at src (row stride ld) -> LDS tile (stride kTL), 16-byte loads
template <typename T>
__device__ __forceinline__ void tile_to_lds(const T* __restrict__ src, int64_t ld, T* dst, int tid) {
  using vec4 = typename Mfma<T>::vec4;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int idx = tid + 256 * v, r = idx >> 4, c = 4 * (idx & 15);
    *reinterpret_cast<vec4*>(dst + r * kTL + c) = *reinterpret_cast<const vec4*>(src + (int64_t)r * ld + c);
  }
}

// all 256 threads: 64x64 global tile -> LDS tile TRANSPOSED (dst[c][r] = src[r][c])
template <typename T>
__device__ __forceinline__ void tile_to_lds_t(const T* __restrict__ src, int64_t ld, T* dst, int tid) {
  using vec4 = typename Mfma<T>::vec4;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int idx = tid + 256 * v, r = idx >> 4, c = 4 * (idx & 15);
    const vec4 x = *reinterpret_cast<const vec4*>(src + (int64_t)r * ld + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[(c + e) * kTL + r] = x[e];
  }
}

// all 256 threads: LDS tile holding the TRANSPOSE -> 64x64 global tile (dst[m][c] = src[c][m]), 16-byte stores
template <typename T>
__device__ __forceinline__ void lds_t_to_tile(const T* src, T* __restrict__ dst, int64_t ld, int tid) {
  using vec4 = typename Mfma<T>::vec4;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int idx = tid + 256 * v, m = idx >> 4, c = 4 * (idx & 15);
    vec4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = src[(c + e) * kTL + m];
    *reinterpret_cast<vec4*>(dst + (int64_t)m * ld + c) = x;
  }
}

// K^-1 = X^T X (X = L^-1) one row block of X at a time: tile (i, j), j <= i, gains
// sum_{r in [max(r0, i), r1)} X[r,i]^T X[r,j].  The contribution of r = i is the tile's first, so it
// overwrites; later ones accumulate.  All 256 threads; uses two T tiles of LDS.
template <typename T>
__device__ __forceinline__ void kinv_accum_tile(const T* __restrict__ linv, T* __restrict__ kinv,
                                                int64_t ld, int i, int j, int r0, int r1,
                                                unsigned char* lds, int tid, int lane, int wave) {
  using M = Mfma<T>;
  using vec4 = typename M::vec4;
  T* TA = reinterpret_cast<T*>(lds);
  T* TB = TA + kFitBlock * kTL;
  const int64_t T64 = kFitBlock;
  const int rb = max(r0, i);
  if (rb >= r1) return;
  vec4 sum[4];
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) sum[tj] = vec4{0, 0, 0, 0};
  for (int r = rb; r < r1; ++r) {
    tile_to_lds<T>(linv + (r * T64) * ld + i * T64, ld, TA, tid);
    if (i != j) tile_to_lds<T>(linv + (r * T64) * ld + j * T64, ld, TB, tid);
    __syncthreads();
    vec4 acc[4];
    mma_atb<T>(TA, wave * 16, (i != j) ? TB : TA, lane, acc);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) sum[tj] += acc[tj];
    __syncthreads();
  }
  T* C = kinv + (i * T64) * ld + j * T64;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T* c = C + (int64_t)(wave * 16 + M::crow(lane, r)) * ld + 16 * tj + (lane & 15);
      *c = (rb == i) ? sum[tj][r] : *c + sum[tj][r];
    }
}

// row blocks [r0, ntile) of X into K^-1 (the ones no step launch could take: the last two)
template <typename T>
__global__ __launch_bounds__(256) void kinv_rows_kernel(const T* __restrict__ linv, T* __restrict__ kinv,
                                                        int64_t ld, int r0, int ntile) {
  extern __shared__ __align__(32) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = (int)blockIdx.x, i = (int)blockIdx.y;
  if (j > i) return;
  kinv_accum_tile<T>(linv, kinv, ld, i, j, r0, ntile, lds, tid, lane, wave);
}

// LDS carving of the step kernel.  role PU: three T tiles.  role D: Ls (f64) | { two T tiles, later
// overlaid by Xs (f64) } | Ts.  float: 73 KB (2 workgroups per CU), double: 107 KB.
template <typename T>
struct StepLds {
  static constexpr int kTileBytes = kFitBlock * kTL * (int)sizeof(T);
  static constexpr int kF64Bytes = kFitBlock * kDS * 8;
  static constexpr int kOver = (2 * kTileBytes > kF64Bytes) ? 2 * kTileBytes : kF64Bytes;
  static constexpr int kBytesD = kF64Bytes + kOver + kTsDoubles * 8;
  static constexpr int kBytes = (3 * kTileBytes > kBytesD) ? 3 * kTileBytes : kBytesD;
};

template <typename T>
__global__ __launch_bounds__(256) void potrf_step_kernel(T* __restrict__ K, T* __restrict__ Lf,
                                                         T* __restrict__ linv, int64_t ld, int k,
                                                         int jmax, int ntile, int64_t n,
                                                         double* __restrict__ diag64,
                                                         int* __restrict__ info, T* __restrict__ W,
                                                         T* __restrict__ kinv, int64_t row_base) {
  using M = Mfma<T>;
  using vec4 = typename M::vec4;
  using Lay = StepLds<T>;
  constexpr int kF64Bytes = Lay::kF64Bytes, kOver = Lay::kOver;
  extern __shared__ __align__(32) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t T64 = kFitBlock;

  // 1-D grid over the ACTIVE workgroups only: [0] role D | role PU tiles | role PB tiles | role KI tiles.
  // (A 3-D grid whose idle workgroups exit at once is harmless on an empty chip, but every workgroup of
  // this launch reserves ~75 KB of LDS: beside a bulk GEMM on another stream -- the two-level path --
  // the idle ones queue for the few free slots and the step takes twice as long.)
  const int nT = (k < 0) ? 0 : ntile - (k + 1);           // trailing tile rows / columns
  const int nPU = nT * (nT + 1) / 2;                      // tiles (i >= j >= k + 1)
  const int nPB = (k < 0) ? 0 : nT * (k + 1);             // tiles (i > k, jp <= k)
  int role = 0, bx = 0, by = 0;                           // decoded into the old (x, y) coordinates
  {
    int id = (int)blockIdx.x;
    if (id == 0) {
      role = 0;
    } else if ((id -= 1) < nPU) {
      // triangle, row-major over r = i - (k + 1) >= c = j - (k + 1): old by = c, old bx = r - c + 1
      int r = (int)((__builtin_sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
      while ((r + 1) * (r + 2) / 2 <= id) ++r;
      while (r * (r + 1) / 2 > id) --r;
      const int c = id - r * (r + 1) / 2;
      role = 3;
      by = c;
      bx = r - c + 1;
    } else if ((id -= nPU) < nPB) {
      role = 1;
      by = id / nT;       // jp
      bx = id % nT + 1;   // i - k
    } else {
      id -= nPB;          // role KI: triangle over i <= k - 1, j <= i
      int r = (int)((__builtin_sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
      while ((r + 1) * (r + 2) / 2 <= id) ++r;
      while (r * (r + 1) / 2 > id) --r;
      role = 2;
      bx = r + 1;
      by = id - r * (r + 1) / 2;
    }
  }
  if (role == 2) {
    // ---------------- role KI: row block k - 1 of X = L^-1 (complete since the previous launch) into
    // K^-1 = X^T X, tile (i, j), j <= i <= k - 1
    const int i = bx - 1, j = by;
    if (kinv == nullptr || k < 1 || i > k - 1 || j > i) return;
    kinv_accum_tile<T>(linv, kinv, ld, i, j, k - 1, k, lds, tid, lane, wave);
    return;
  }
  if (role == 1) {
    // ---------------- role PB: tile (i, jp) of the inverse's right-hand side, jp <= k < i ----------
    // L X = I solved by the same elimination: B starts as the identity, step k finishes row block k,
    // X[k,jp] = X_kk B[k,jp], and updates the rows below, B[i,jp] -= L[i,k] X[k,jp].  B is kept
    // TRANSPOSED in the scratch matrix, W[jp,i] = B[i,jp]^T, so that every product is the A B^T form
    // of mma_abt:  Xkj^T = W[jp,k] X_kk^T,  W[jp,i] -= Xkj^T Li^T.  (jp == k: B[k,k] = I, Xkj = X_kk,
    // and the first contribution overwrites W.)  The i == k+1 tiles also store X[k,jp] -> linv.
    const int jp = by, i = k + bx;
    if (k < 0 || jp > k || i >= ntile) return;
    T* TI = reinterpret_cast<T*>(lds);
    T* TW = TI + kFitBlock * kTL;
    T* TX = TW + kFitBlock * kTL;
    const T* Xkk = linv + (k * T64) * ld + k * T64;
    // this lane's entries of the tile it will update, in flight from the start
    T* Cw = W + (jp * T64) * ld + i * T64 + (int64_t)(wave * 16) * ld + (lane & 15);
    T c_old[4][4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_old[tj][r] = (jp < k) ? Cw[(int64_t)M::crow(lane, r) * ld + 16 * tj] : (T)0;
    tile_to_lds<T>(K + (i * T64) * ld + k * T64, ld, TI, tid);
    tile_to_lds<T>(Xkk, ld, TX, tid);
    if (jp < k) tile_to_lds<T>(W + (jp * T64) * ld + k * T64, ld, TW, tid);
    else tile_to_lds_t<T>(Xkk, ld, TW, tid);
    __syncthreads();
    vec4 acc[4];
    mma_abt<T, true>(TI + wave * 16 * kTL, TX, lane, acc);  // Li
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        TI[(wave * 16 + M::crow(lane, r)) * kTL + 16 * tj + (lane & 15)] = acc[tj][r];
    if (jp < k) {
      mma_abt<T, true>(TW + wave * 16 * kTL, TX, lane, acc);  // Xkj^T rows of this wave
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          TW[(wave * 16 + M::crow(lane, r)) * kTL + 16 * tj + (lane & 15)] = acc[tj][r];
    }
    __syncthreads();
    if (i == k + 1 && jp < k) lds_t_to_tile<T>(TW, linv + (k * T64) * ld + jp * T64, ld, tid);
    mma_abt<T, false>(TW + wave * 16 * kTL, TI, lane, acc);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(int64_t)M::crow(lane, r) * ld + 16 * tj] = c_old[tj][r] - acc[tj][r];
    return;
  }
  if (role == 0) {
    // ---------------- role D: diagonal block kd = k + 1 ----------------------------------------
    const int kd = k + 1;
    if (kd >= ntile) return;
    double* Ls = reinterpret_cast<double*>(lds);
    T* TA = reinterpret_cast<T*>(lds + kF64Bytes);
    T* TX = TA + kFitBlock * kTL;
    double* Xs = reinterpret_cast<double*>(lds + kF64Bytes);  // overlays TA / TX once they are dead
    double* Ts = reinterpret_cast<double*>(lds + kF64Bytes + kOver);
    const T* Akk = K + (kd * T64) * ld + kd * T64;
    GPSO_STAMP(14);
    if (k >= 0) {
      // S = A[kd,kd] - L10 L10^T needs the lower 16x16 tiles only: three per wave (the last wave:
      // one).  This lane's entries of A[kd,kd] for those tiles are fetched first and stay in flight
      // while L10 is formed.
      constexpr int kSti[4][3] = {{0, 3, 3}, {1, 1, 3}, {2, 2, 2}, {3, -1, -1}};
      constexpr int kStj[4][3] = {{0, 0, 1}, {0, 1, 2}, {0, 1, 2}, {3, -1, -1}};
      int sti[3], stj[3];
      T akk[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        sti[t] = stj[t] = -1;
#pragma unroll
        for (int w = 0; w < 4; ++w)
          if (wave == w) {
            sti[t] = kSti[w][t];
            stj[t] = kStj[w][t];
          }
        if (sti[t] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            akk[t][r] = Akk[(int64_t)(16 * sti[t] + M::crow(lane, r)) * ld + 16 * stj[t] + (lane & 15)];
        }
      }
      tile_to_lds<T>(K + (kd * T64) * ld + k * T64, ld, TA, tid);
      tile_to_lds<T>(linv + (k * T64) * ld + k * T64, ld, TX, tid);
      __syncthreads();
      GPSO_STAMP(17);
      vec4 acc[4];
      mma_abt<T, true>(TA + wave * 16 * kTL, TX, lane, acc);  // L10 rows of this wave
      // (Lf[kd,k] is stored by the role-PU workgroup of tile (kd,kd), which forms the same L10)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)  // rows of this wave only: no other wave reads them yet
          TA[(wave * 16 + M::crow(lane, r)) * kTL + 16 * tj + (lane & 15)] = acc[tj][r];
      __syncthreads();
      GPSO_STAMP(18);
      // all fragments of the (up to) three tiles are fetched before the first MFMA issues
      vec4 fa[3][4], fb[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (sti[t] >= 0) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int ko = 16 * kk + 4 * (lane >> 4);
            fa[t][kk] = *reinterpret_cast<const vec4*>(TA + (16 * sti[t] + (lane & 15)) * kTL + ko);
            fb[t][kk] = *reinterpret_cast<const vec4*>(TA + (16 * stj[t] + (lane & 15)) * kTL + ko);
          }
        }
      // the (up to) three tiles of a wave accumulate interleaved: three independent MFMA chains of 16
      // instead of one of 48 (a dependent MFMA waits for the previous one's passes); the last wave's
      // missing tiles cost nothing (wave-uniform skip of whole k-sweeps)
      vec4 a3[3] = {vec4{0, 0, 0, 0}, vec4{0, 0, 0, 0}, vec4{0, 0, 0, 0}};
      if (sti[1] >= 0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 3; ++t) a3[t] = M::mma(fa[t][kk][e], fb[t][kk][e], a3[t]);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int e = 0; e < 4; ++e) a3[0] = M::mma(fa[0][kk][e], fb[0][kk][e], a3[0]);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int ti = sti[t], tj = stj[t];
        if (ti < 0) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti + M::crow(lane, r), col = 16 * tj + (lane & 15);
          const T upd = akk[t][r] - a3[t][r];
          Ls[row * kDS + col] = (col <= row) ? (double)upd : 0.0;
        }
      }
    } else {
      for (int e = tid; e < kFitBlock * kFitBlock; e += 256) {
        const int r = e >> 6, c = e & 63;
        Ls[r * kDS + c] = (c <= r) ? (double)Akk[(int64_t)r * ld + c] : 0.0;
      }
    }
    __syncthreads();
    const int64_t k0 = kd * T64;
    GPSO_STAMP(15);
    // (row_base: global row of this sub-matrix's first row -- pivot indices, padding test and the
    // diagonal go by global row; every tile address above is relative to the sub-matrix)
    chol64_lds<(sizeof(T) == 4) ? 1 : 2, T, sizeof(T) == 4, true>(Ls, Xs, Lf + k0 * ld + k0, ld, row_base + k0, n, info, Ts);
    // the unrounded diagonal: its logarithms are summed by nlml_block (alpha_sum_kernel's last workgroup), off this chain
    if (tid < kFitBlock) diag64[row_base + k0 + tid] = Ls[tid * kDS + tid];
    GPSO_STAMP(7);
    trinv64_lds<true, T, true>(Ls, Xs, Ts, Lf + k0 * ld + k0, ld);
    lower_tile_to_global<T, 256, true>(Xs, linv + k0 * ld + k0, ld, tid);
    GPSO_STAMP(16);
    return;
  }
  // ---------------- role PU: tile (i, j) of the trailing update -----------------------------------
  if (k < 0) return;
  const int j = k + 1 + by;
  const int i = j + bx - 1;
  if (j > jmax || i >= ntile) return;
  const bool diag_next = (i == k + 1);  // tile (k+1,k+1): role D updates it; only L10 is stored here
  T* TI = reinterpret_cast<T*>(lds);
  T* TJ = TI + kFitBlock * kTL;
  T* TX = TJ + kFitBlock * kTL;
  // this lane's entries of the tile it will update, in flight from the start
  T* Cu = K + (i * T64) * ld + j * T64 + (int64_t)(wave * 16) * ld + (lane & 15);
  T c_old[4][4];
  if (!(diag_next && j == k + 1)) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_old[tj][r] = Cu[(int64_t)M::crow(lane, r) * ld + 16 * tj];
  }
  tile_to_lds<T>(K + (i * T64) * ld + k * T64, ld, TI, tid);
  if (i != j) tile_to_lds<T>(K + (j * T64) * ld + k * T64, ld, TJ, tid);
  tile_to_lds<T>(linv + (k * T64) * ld + k * T64, ld, TX, tid);
  __syncthreads();
  vec4 acc[4];
  mma_abt<T, true>(TI + wave * 16 * kTL, TX, lane, acc);
  {
    T* Lo = Lf + (i * T64) * ld + k * T64;
    const bool keep = (j == k + 1);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + M::crow(lane, r), col = 16 * tj + (lane & 15);
        TI[row * kTL + col] = acc[tj][r];
        if (keep) Lo[(int64_t)row * ld + col] = acc[tj][r];
      }
  }
  if (diag_next && j == k + 1) return;
  if (i != j) {
    mma_abt<T, true>(TJ + wave * 16 * kTL, TX, lane, acc);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        TJ[(wave * 16 + M::crow(lane, r)) * kTL + 16 * tj + (lane & 15)] = acc[tj][r];
  }
  __syncthreads();
  mma_abt<T, false>(TI + wave * 16 * kTL, (i != j) ? TJ : TI, lane, acc);
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) Cu[(int64_t)M::crow(lane, r) * ld + 16 * tj] = c_old[tj][r] - acc[tj][r];
}

// last row block of the inverse (no rows below it, so no step launch finished it):
// X[k,jp] = X_kk B[k,jp] for jp < k = ntile - 1, one workgroup per tile
template <typename T>
__global__ __launch_bounds__(256) void inv_lastrow_kernel(T* __restrict__ linv, const T* __restrict__ W,
                                                          int64_t ld, int k) {
  using M = Mfma<T>;
  using vec4 = typename M::vec4;
  extern __shared__ __align__(32) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jp = (int)blockIdx.x;
  const int64_t T64 = kFitBlock;
  T* TW = reinterpret_cast<T*>(lds);
  T* TX = TW + kFitBlock * kTL;
  tile_to_lds<T>(W + (jp * T64) * ld + k * T64, ld, TW, tid);
  tile_to_lds<T>(linv + (k * T64) * ld + k * T64, ld, TX, tid);
  __syncthreads();
  vec4 acc[4];
  mma_abt<T, true>(TW + wave * 16 * kTL, TX, lane, acc);
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      TW[(wave * 16 + M::crow(lane, r)) * kTL + 16 * tj + (lane & 15)] = acc[tj][r];
  __syncthreads();
  lds_t_to_tile<T>(TW, linv + (k * T64) * ld + jp * T64, ld, tid);
}

// =============================================================================================
// blocked Cholesky
// =============================================================================================
// Right-looking in 64-wide steps, one potrf_step_kernel launch per step (see above).
//   npad <= kSingleLevelMax (or the GPSO_OPT_FIT_SINGLE_LEVEL_MAX option): every step updates the
//       whole trailing matrix (rank 64); at these sizes
//       the matrix lives in the L2 / Infinity Cache and a step's bulk work is shorter than the
//       diagonal chain it hides behind.  The off-diagonal blocks of L^-1 ride along in the same
//       launches (role PB + one last-row launch), so no separate triangular inverse follows; when
//       the gradient is wanted K^-1 = L^-T L^-1 rides along as well (role KI + one tail launch).
//   larger: two-level by fit_outer_panel(npad)-wide diagonal blocks, see launch_potrf.
// measured crossovers (posterior fit, ms, single | two-level): float 3584: 1.38 | 1.44, 4096: 1.86 | 1.73;
// double 2560: 1.24 | 1.36, 3072: 1.85 | 1.68
template <typename T>
constexpr int64_t kSingleLevelMax = (sizeof(T) == 4) ? 3584 : 2560;

// Single-level factorisation (+ inverse, + optionally K^-1) of the ntile x ntile tile sub-matrix whose
// first row / column is global row row_base; all pointers address that sub-matrix, ld is the leading
// dimension of the full matrices.
template <typename T>
static void potrf_block(hipStream_t st, T* K, T* Lf, T* linv, T* work, T* kinv, int64_t ld, int ntile,
                        int64_t row_base, int64_t n, double* diag64, int* info) {
  // the kernels need more than 64 KB of LDS: opt in, per device (the attribute belongs to the
  // device's function object)
  if (ensure_dyn_lds(reinterpret_cast<const void*>(&potrf_step_kernel<T>), StepLds<T>::kBytes) ||
      ensure_dyn_lds(reinterpret_cast<const void*>(&inv_lastrow_kernel<T>), 2 * StepLds<T>::kTileBytes) ||
      ensure_dyn_lds(reinterpret_cast<const void*>(&kinv_rows_kernel<T>), 2 * StepLds<T>::kTileBytes))
    return;
  const int lds_bytes = StepLds<T>::kBytes;
  auto step = [&](int k) {
    // one workgroup per ACTIVE role instance (decoded in the kernel): D, the trailing tiles (PU), the tiles
    // of the inverse's right-hand side (PB), and -- when K^-1 is wanted -- the tiles of its finished part (KI)
    const int nT = (k < 0) ? 0 : ntile - (k + 1);
    const int nPU = nT * (nT + 1) / 2, nPB = (k < 0) ? 0 : nT * (k + 1);
    const int nKI = (kinv != nullptr && k >= 1) ? k * (k + 1) / 2 : 0;
    hipLaunchKernelGGL((potrf_step_kernel<T>), dim3((unsigned)(1 + nPU + nPB + nKI)), dim3(256), lds_bytes, st, K, Lf,
                       linv, ld, k, ntile - 1, ntile, n, diag64, info, work, kinv, row_base);
  };
  step(-1);  // diagonal block 0
  for (int k = 0; k < ntile - 1; ++k) step(k);
  if (ntile > 1)
    hipLaunchKernelGGL((inv_lastrow_kernel<T>), dim3((unsigned)(ntile - 1)), dim3(256),
                       2 * StepLds<T>::kTileBytes, st, linv, work, ld, ntile - 1);
  if (kinv != nullptr)
    hipLaunchKernelGGL((kinv_rows_kernel<T>), dim3((unsigned)ntile, (unsigned)ntile), dim3(256),
                       2 * StepLds<T>::kTileBytes, st, linv, kinv, ld, std::max(0, ntile - 2), ntile);
}

template <typename T>
bool potrf_is_single_level(int64_t npad, int64_t single_max) {
  return npad <= (single_max >= 0 ? single_max : kSingleLevelMax<T>);
}
template bool potrf_is_single_level<float>(int64_t, int64_t);
template bool potrf_is_single_level<double>(int64_t, int64_t);

template <typename T>
int launch_potrf(hipStream_t st, T* K, T* Lf, T* linv, T* work, T* kinv, int64_t n, int64_t npad,
                 double* diag64, int* info, int64_t single_max, const FitPlanes* planes) {
  const bool single = potrf_is_single_level<T>(npad, single_max);
  if (single) {
    // only the 64-row blocks that hold training rows: a block of padding alone is an identity block of the factor and
    // of its inverse, nothing downstream reads it (every consumer of L, L^-1, K^-1 and the diagonal stops at row n), and
    // its step would be a whole link of the launch chain (15 us) -- float-predict contexts pad N to 256
    potrf_block<T>(st, K, Lf, linv, work, kinv, npad, (int)((n + kFitBlock - 1) / kFitBlock), 0, n, diag64, info);
    return 1 | (kinv != nullptr ? 2 : 0);
  }
  // two-level, by kOuterPanel = fit_outer_panel(npad)-wide diagonal blocks:
  const int64_t kOuterPanel = fit_outer_panel(npad);
  //   1. the diagonal block is factored AND inverted by the single-level routine (a latency chain of
  //      kOuterPanel / 64 steps with hardly any bulk work);
  //   2. rows below:  L21 = A21 X11^T  -- a plain GEMM, X11 the block's lower-triangular inverse;
  //   3. A22 -= L21 L21^T  (lower tiles).
  // The level-doubling inverse then starts at level kOuterPanel (launch_trtri's first_level).
  // Look-ahead (float fits with planes and a side stream): the diagonal block is a ~15 us x wp / 64 chain of
  // launches that leaves the chip idle, so block p + 1 is factored on the side stream as soon as the update of
  // panel p has produced its block column, while the rest of that update (a persistent kernel that leaves
  // kLookaheadCus compute units to the chain) runs here.  (tools/micro/lookahead_probe.hip: beside a GEMM on all
  // CUs the chain takes 416 us instead of 263, with 32 CUs left to it 307.)
  constexpr int kLookaheadCus = 48;  // measured: 8-16 leave the chain's ~100-workgroup launches crawling; 32-64 within 2 %
  const bool can_look = sizeof(T) == 4 && planes != nullptr && planes->L != nullptr && planes->side != nullptr;
  // Round 6, double fits (the fit of "float64" and "mixed" contexts).  Measured at C4 (profiles/r06_fit_c4_f64_timeline_before.txt):
  // 9.75 ms = 2.07 ms of diagonal-block chains with the chip idle + 3.5 ms of TRSM / SYRK products + 3.65 ms of level-doubling
  // inverse behind them + 0.5 ms.  (a) look-ahead as in the float fit: the next diagonal block is factored on `side` as soon as
  // its block column of the update is done, beside the rest of the update.  (b) the inverse does not wait for the factorisation:
  // a block's inverse needs only the panels inside it, so the products of each pair are issued on `inv` the moment both halves
  // are final -- the FIRST product of a pair, W = L[B,A] Linv[A,A], already when the A half is (the last level's starts at half
  // time and runs beside the second half's chains, which leave most of the chip idle).  Same products on the same tiles in the
  // same k order as the sequential schedule: the same bits (ts_hint keeps the tile size a split launch would otherwise lose).
  const int ov = (sizeof(T) == 8 && planes != nullptr) ? planes->overlap : 0;  // bit 0: look-ahead, bit 1: overlapped inverse
  const bool look64 = (ov & 1) && planes->side != nullptr && planes->ev_col != nullptr && planes->ev_chain != nullptr;
  const int64_t npanels = npad / kOuterPanel;
  const bool dag64 = (ov & 2) && planes->inv != nullptr && planes->ev_panel != nullptr && planes->ev_inv != nullptr &&
                     npad % kOuterPanel == 0 && npanels >= 2 && (npanels & (npanels - 1)) == 0;
  // (bits 8..: products of pairs wider than 2^(bits) panels wait for the end of the factorisation -- experiments)
  const int64_t defer_from = (ov >> 8) > 0 ? (kOuterPanel << ((ov >> 8) - 1)) : npad;
  std::vector<std::pair<int64_t, int64_t>> deferred_w;  // (lo, s) of first products held back
  // the products of pair [lo, lo + 2 s) of the inverse's level s (launch_trtri's two GEMMs, one pair each), on `inv`
  auto pair_w = [&](int64_t lo, int64_t s) {  // W[B,A] = L[B,A] Linv[A,A]
    GemmDesc a{};
    const int64_t o = lo * npad + lo;
    a.A = Lf + o + s * npad; a.sai = npad; a.sak = 1;
    a.B = linv + o; a.sbk = npad; a.sbj = 1;
    a.C = work + o + s * npad; a.ldc = npad;
    a.m = a.n = a.k = a.m_last = (int)s; a.nbatch = 1;
    a.alpha = 1.0; a.beta = 0.0; a.kmode = 1;
    a.ts_hint = gemm_tile_choice((int)s, (int)s, (int)s, (int)s, (int)(npad / (2 * s)), false);
    launch_gemm<T>(planes->inv, a);
  };
  auto pair_x = [&](int64_t lo, int64_t s) {  // Linv[B,A] = -Linv[B,B] W[B,A]
    GemmDesc b{};
    const int64_t o = lo * npad + lo;
    b.A = linv + o + s * npad + s; b.sai = npad; b.sak = 1;
    b.B = work + o + s * npad; b.sbk = npad; b.sbj = 1;
    b.C = linv + o + s * npad; b.ldc = npad;
    b.m = b.n = b.k = b.m_last = (int)s; b.nbatch = 1;
    b.alpha = -1.0; b.beta = 0.0; b.kmode = 3;
    b.ts_hint = gemm_tile_choice((int)s, (int)s, (int)s, (int)s, (int)(npad / (2 * s)), false);
    launch_gemm<T>(planes->inv, b);
  };
  // block [lo, lo + size) has its inverse complete (in the order of `inv`): an A half starts its parent's first product, a B
  // half finishes the parent
  std::function<void(int64_t, int64_t)> complete = [&](int64_t lo, int64_t size) {
    if (size >= npad) return;
    const int64_t parent = lo / (2 * size) * (2 * size);
    if (lo == parent) {
      if (size >= defer_from) deferred_w.push_back({parent, size});
      else pair_w(parent, size);
    } else {
      for (size_t i = 0; i < deferred_w.size(); ++i)
        if (deferred_w[i].first == parent && deferred_w[i].second == size) {
          pair_w(parent, size);
          deferred_w.erase(deferred_w.begin() + (long)i);
          break;
        }
      pair_x(parent, size);
      complete(parent, 2 * size);
    }
  };
  bool chain_on_side = false;  // the diagonal block of this iteration was launched on the side stream
  for (int64_t c0 = 0; c0 < npad; c0 += kOuterPanel) {
    const int64_t wp = std::min<int64_t>(kOuterPanel, npad - c0);
    const int64_t off = c0 * npad + c0;
    if (chain_on_side) {
      (void)hipStreamWaitEvent(st, planes->ev_chain, 0);
      chain_on_side = false;
    } else {
      potrf_block<T>(st, K + off, Lf + off, linv + off, work + off, nullptr, npad, (int)(wp / kFitBlock), c0, n,
                     diag64, info);
    }
    const int64_t r1 = c0 + wp;
    const int m2 = (int)(npad - r1);
    if (m2 <= 0) {
      if (dag64) {  // the last panel is final: the remaining products of the inverse, then the caller's stream joins
        (void)hipEventRecord(planes->ev_panel, st);
        (void)hipStreamWaitEvent(planes->inv, planes->ev_panel, 0);
        complete(c0, wp);
        (void)hipEventRecord(planes->ev_inv, planes->inv);
        (void)hipStreamWaitEvent(st, planes->ev_inv, 0);
      }
      break;
    }
    GemmDesc t{};  // L21 = A21 X11^T:  opB(k, j) = X11[j][k], zero for k > j
    t.A = K + r1 * npad + c0; t.sai = npad; t.sak = 1;
    t.B = linv + off; t.sbk = 1; t.sbj = npad;
    t.C = Lf + r1 * npad + c0; t.ldc = npad;
    t.m = m2; t.n = (int)wp; t.k = (int)wp; t.m_last = m2; t.nbatch = 1;
    t.alpha = 1.0; t.beta = 0.0; t.kmode = 4;
    // float fits with bf16 planes: the TRSM also emits L21 into the planes of L (the level-doubling inverse
    // and the rank-wp update read it from there), and the update runs on the bf16 matrix cores
    // (gemm_bf16_kernel) while the trailing matrix is large enough to fill the chip with 128-tiles
    const bool have_planes = sizeof(T) == 4 && planes != nullptr && planes->L != nullptr;
    if (have_planes) {
      t.split = planes->L;
      t.split_stride = planes->stride;
      t.split_nkb = planes->nkb;
      t.split_row0 = r1;
      t.split_col0 = c0;
      t.split_np = planes->np;
      t.split_scale = planes->sL;
    }
    launch_gemm<T>(st, t);
    if (have_planes && m2 >= kSyrkBf16MinRows) {
      if constexpr (sizeof(T) == 4) {
        GemmBf16Desc sy{};
        sy.A = sy.B = Bf16Planes{planes->L, planes->stride, planes->nkb, planes->sL};
        sy.np = planes->np;
        sy.a_row0 = sy.b_row0 = r1;
        sy.a_col0 = sy.b_col0 = c0;
        sy.C = reinterpret_cast<float*>(K + r1 * npad + r1);
        sy.ldc = npad;
        sy.m = sy.n = m2;
        sy.k = (int)wp;
        sy.alpha = -1.0f;
        sy.beta = 1;
        sy.lower_only = 1;
        sy.nbatch = 1;
        const int64_t wn = std::min<int64_t>(kOuterPanel, npad - r1);  // the next panel
        if (can_look && m2 - wn >= kSyrkBf16MinRows) {
          GemmBf16Desc col = sy;  // its block column first: rows r1.., columns r1 .. r1 + wn
          col.n = (int)wn;
          col.lower_only = 0;
          launch_gemm_bf16(st, col);
          (void)hipEventRecord(planes->ev_col, st);
          GemmBf16Desc rest = sy;  // then everything to the right of it (enqueued before the chain's 17 launches:
          rest.a_row0 = rest.b_row0 = r1 + wn;  // the host needs ~100 us for those)
          rest.C = reinterpret_cast<float*>(K + (r1 + wn) * npad + (r1 + wn));
          rest.m = rest.n = (int)(m2 - wn);
          launch_gemm_bf16(st, rest, kLookaheadCus);
          (void)hipStreamWaitEvent(planes->side, planes->ev_col, 0);
          const int64_t off1 = r1 * npad + r1;
          potrf_block<T>(planes->side, K + off1, Lf + off1, linv + off1, work + off1, nullptr, npad,
                         (int)(wn / kFitBlock), r1, n, diag64, info);
          (void)hipEventRecord(planes->ev_chain, planes->side);
          chain_on_side = true;
        } else {
          launch_gemm_bf16(st, sy);
        }
      }
      continue;
    }
    GemmDesc u{};  // A22 -= L21 L21^T
    u.A = Lf + r1 * npad + c0; u.sai = npad; u.sak = 1;
    u.B = u.A; u.sbk = 1; u.sbj = npad;
    u.C = K + r1 * npad + r1; u.ldc = npad;
    u.m = m2; u.n = m2; u.k = (int)wp; u.m_last = m2; u.nbatch = 1;
    u.alpha = -1.0; u.beta = 1.0; u.lower_only = 1;
    if (dag64) {  // panel c0 is final (its block column of L and its diagonal block's inverse): its part of the inverse
      (void)hipEventRecord(planes->ev_panel, st);
      (void)hipStreamWaitEvent(planes->inv, planes->ev_panel, 0);
      complete(c0, wp);
    }
    const int64_t wn = std::min<int64_t>(kOuterPanel, npad - r1);  // the next panel
    if (look64 && m2 > wn) {
      const int ts = gemm_tile_choice(m2, m2, m2, (int)wp, 1, true);
      GemmDesc col = u;  // the next panel's block column first: rows r1 .., columns r1 .. r1 + wn (its upper tiles ride along)
      col.n = (int)wn;
      col.lower_only = 0;
      col.ts_hint = ts;
      launch_gemm<T>(st, col);
      (void)hipEventRecord(planes->ev_col, st);
      GemmDesc rest = u;  // then everything to the right of it, while the next diagonal block is factored on the side stream
      rest.A = rest.B = Lf + (r1 + wn) * npad + c0;
      rest.C = K + (r1 + wn) * npad + (r1 + wn);
      rest.m = rest.n = rest.m_last = (int)(m2 - wn);
      rest.ts_hint = ts;
      launch_gemm<T>(st, rest);
      (void)hipStreamWaitEvent(planes->side, planes->ev_col, 0);
      const int64_t off1 = r1 * npad + r1;
      potrf_block<T>(planes->side, K + off1, Lf + off1, linv + off1, work + off1, nullptr, npad, (int)(wn / kFitBlock), r1, n,
                     diag64, info);
      (void)hipEventRecord(planes->ev_chain, planes->side);
      chain_on_side = true;
      continue;
    }
    launch_gemm<T>(st, u);
  }
  return dag64 ? 1 : 0;
}
template int launch_potrf<float>(hipStream_t, float*, float*, float*, float*, float*, int64_t, int64_t, double*, int*, int64_t, const FitPlanes*);
template int launch_potrf<double>(hipStream_t, double*, double*, double*, double*, double*, int64_t, int64_t, double*, int*, int64_t, const FitPlanes*);

// =============================================================================================
// triangular inverse by level doubling
// =============================================================================================
// level with half-size s: pairs p = 0.. ; A = [2ps, 2ps+s), B = [2ps+s, min(2ps+2s, npad));
//   W[B,A]    = L[B,A] * Linv[A,A]         (Linv[A,A] lower  -> k >= 64 tj)
//   Linv[B,A] = -Linv[B,B] * W[B,A]        (Linv[B,B] lower  -> k <  64 (ti+1))
template <typename T>
void launch_trtri(hipStream_t st, const T* L, T* linv, T* work, int64_t npad, int64_t first_level) {
  for (int64_t s = first_level; s < npad; s *= 2) {
    const int64_t span = 2 * s;
    const int nfull = (int)(npad / span);          // pairs with a full-size B
    const int64_t rem = npad - (int64_t)nfull * span;  // leftover columns
    int nbatch = nfull;
    int m_last = (int)s;
    if (rem > s) {  // one more pair with a partial B of rem - s rows
      nbatch = nfull + 1;
      m_last = (int)(rem - s);
    }
    if (nbatch == 0) continue;
    const int64_t bstride = span * npad + span;
    GemmDesc a{};
    a.A = L + s * npad; a.sai = npad; a.sak = 1;  // L[B rows, A cols]
    a.B = linv; a.sbk = npad; a.sbj = 1;          // Linv[A, A]
    a.C = work + s * npad; a.ldc = npad;
    a.batchA = a.batchB = a.batchC = bstride;
    a.m = (int)s; a.n = (int)s; a.k = (int)s; a.m_last = m_last; a.nbatch = nbatch;
    a.alpha = 1.0; a.beta = 0.0; a.kmode = 1;
    launch_gemm<T>(st, a);
    GemmDesc b{};
    b.A = linv + s * npad + s; b.sai = npad; b.sak = 1;  // Linv[B, B]
    b.B = work + s * npad; b.sbk = npad; b.sbj = 1;      // W[B, A]
    b.C = linv + s * npad; b.ldc = npad;
    b.batchA = b.batchB = b.batchC = bstride;
    b.m = (int)s; b.n = (int)s; b.k = (int)s; b.m_last = m_last; b.nbatch = nbatch;
    b.alpha = -1.0; b.beta = 0.0; b.kmode = 3;
    launch_gemm<T>(st, b);
  }
}
// The same level doubling on the bf16 matrix cores (float fits; N_pad / first_level a power of two).  All
// operands are read as k-contiguous rows of plane sets: L (written by the TRSM epilogues), X = L^-1, XT = its
// transpose, WT = scratch.  Level with half-size s, pair [A | B]:
//   WT[A,B]   = XT[A,A] L[B,A]^T            (= (L[B,A] Linv[A,A])^T;  XT[A,A][j][r] = 0 for r < j  -> k >= 128 ti)
//   Linv[B,A] = -X[B,B] (WT[A,B])^T         (X[B,B][i][k] = 0 for k > i                         -> k < 128 (ti + 1))
// the second product writes the float result into linv and its pieces into X and -- transposed -- into XT, which
// is what the next level reads.  The diagonal blocks of L^-1 (float, from the step kernels) enter X / XT first.
void launch_trtri_bf16(hipStream_t st, float* linv, const FitPlanes& pl, int64_t npad, int64_t first_level,
                       bool keep_xt) {
  const Bf16Planes L{pl.L, pl.stride, pl.nkb, pl.sL}, X{pl.X, pl.stride, pl.nkb, pl.sX}, XT{pl.XT, pl.stride, pl.nkb, pl.sX},
      WT{pl.WT, pl.stride, pl.nkb, pl.sW}, none{nullptr, 0, 0, 1.0f};
  {
    const int fl = (int)first_level, nb = (int)(npad / first_level);
    const int64_t pairs = (int64_t)fl * fl / 2;
    hipLaunchKernelGGL(block_to_planes_kernel, dim3((unsigned)((pairs + 255) / 256), (unsigned)nb), dim3(256), 0, st, linv,
                       npad, fl, fl, (int64_t)fl * npad + fl, (int64_t)fl, X, (int64_t)0, (int64_t)0, XT, (int64_t)0,
                       (int64_t)0, pl.np);
  }
  for (int64_t s = first_level; s < npad; s *= 2) {
    const int nb = (int)(npad / (2 * s));
    GemmBf16Desc a{};  // WT[A,B]
    a.np = pl.np;
    a.A = XT; a.a_row0 = 0; a.a_col0 = 0;
    a.B = L; a.b_row0 = s; a.b_col0 = 0;
    a.C = nullptr; a.ldc = npad;
    a.m = a.n = a.k = (int)s;
    a.alpha = 1.0f; a.beta = 0; a.kmode = 2;
    a.nbatch = nb; a.batch_shift = 2 * s;
    a.out = WT; a.o_row0 = 0; a.o_col0 = s;
    a.out_t = none;
    launch_gemm_bf16(st, a);
    GemmBf16Desc b{};  // Linv[B,A]
    b.np = pl.np;
    b.A = X; b.a_row0 = s; b.a_col0 = s;
    b.B = WT; b.b_row0 = 0; b.b_col0 = s;
    b.C = linv + s * npad; b.ldc = npad;
    b.m = b.n = b.k = (int)s;
    b.alpha = -1.0f; b.beta = 0; b.kmode = 3;
    b.nbatch = nb; b.batch_shift = 2 * s;
    b.out = X; b.o_row0 = s; b.o_col0 = 0;
    b.out_t = XT; b.ot_row0 = 0; b.ot_col0 = s;
    if (2 * s >= npad) {  // last level: nobody reads X afterwards; XT only feeds K^-1 = XT XT^T (gradient)
      b.out = none;
      if (!keep_xt) b.out_t = none;
    }
    launch_gemm_bf16(st, b);
  }
}
bool trtri_bf16_applies(int64_t npad, int64_t first_level) {
  const int64_t q = npad / first_level;
  return npad % first_level == 0 && q >= 2 && (q & (q - 1)) == 0;
}

template void launch_trtri<float>(hipStream_t, const float*, float*, float*, int64_t, int64_t);
template void launch_trtri<double>(hipStream_t, const double*, double*, double*, int64_t, int64_t);

// =============================================================================================
// pack L^-1 into MFMA fragment-major tiles (layout documented in predict.hip)
// =============================================================================================
// Only the lower 16x16 tiles are stored, row-major over the triangle: tile (rt, kt <= rt) at
// (rt (rt + 1) / 2 + kt) * 256 elements.  (Round 1 also stored the all-zero upper tiles, which doubled
// this buffer and the multi-GPU broadcast of it.)  TF = fit type, TP = predict type.
template <typename TF, typename TP>
__global__ __launch_bounds__(256) void pack_linv_kernel(const TF* __restrict__ linv, int64_t n,
                                                        int64_t npad, TP* __restrict__ linv_p, int64_t tile0) {
  using vec4 = typename Mfma<TP>::vec4;
  const int64_t idx = tile0 * 64 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (tile, lane)
  const int64_t npad16 = npad / 16;
  if (idx >= npad16 * (npad16 + 1) / 2 * 64) return;
  const int lane = (int)(idx & 63);
  const int64_t tile = idx >> 6;
  int64_t rt = (int64_t)((__builtin_sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
  while ((rt + 1) * (rt + 2) / 2 <= tile) ++rt;
  while (rt * (rt + 1) / 2 > tile) --rt;
  const int64_t kt = tile - rt * (rt + 1) / 2;
  const int64_t row = rt * 16 + (lane & 15);
  const int64_t col = kt * 16 + 4 * (lane >> 4);
  vec4 v{0, 0, 0, 0};
  if (row < n) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col + r <= row && col + r < n) v[r] = (TP)linv[row * npad + col + r];
  }
  reinterpret_cast<vec4*>(linv_p)[idx] = v;
}

template <typename TF, typename TP>
void launch_pack_linv(hipStream_t st, const TF* linv, int64_t n, int64_t npad, TP* linv_p, int64_t rt0) {
  const int64_t tile0 = rt0 * (rt0 + 1) / 2;  // tiles are stored row-major over the triangle: tile row rt0 starts here
  const int64_t total = ((npad / 16) * (npad / 16 + 1) / 2 - tile0) * 64;
  if (total <= 0) return;
  hipLaunchKernelGGL((pack_linv_kernel<TF, TP>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     linv, n, npad, linv_p, tile0);
}
template void launch_pack_linv<float, float>(hipStream_t, const float*, int64_t, int64_t, float*, int64_t);
template void launch_pack_linv<double, double>(hipStream_t, const double*, int64_t, int64_t, double*, int64_t);
template void launch_pack_linv<double, float>(hipStream_t, const double*, int64_t, int64_t, float*, int64_t);

// =============================================================================================
// single-RHS solves through L^-1 and the NLML
// =============================================================================================
// white[i] = sum_{k<=i} Linv[i][k] (y[k] - c): one wave per row, wavefront reduction
// amax_rows != nullptr: the pass also leaves max_k |L^-1[i][k]| of every row there (alpha_sum_kernel folds them) -- the
// fp16 split of the predict path scales L^-1 by the maximum (predict.hip: pack_linv_f16_kernel), and this kernel reads
// every entry anyway.  (One atomicMax per wave on a single address instead: 2 048 of them serialise in the L2, 20 us.)
template <typename T>
__global__ __launch_bounds__(256) void white_kernel(const T* __restrict__ linv,
                                                    const double* __restrict__ y64, int64_t n,
                                                    int64_t npad, double mean_c,
                                                    T* __restrict__ white, float* __restrict__ amax_rows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= npad) return;
  double acc = 0.0;
  float m = 0.0f;
  if (i < n) {
    // (the loads of four steps are issued before the first is used: one row per wave, nothing else hides the latency)
    const T* row = linv + i * npad;
    int64_t k = lane;
    for (; k + 192 <= i; k += 256) {
      T l[4];
      double r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        l[u] = row[k + 64 * u];
        r[u] = y64[k + 64 * u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc += (double)l[u] * (r[u] - mean_c);
        m = fmaxf(m, fabsf((float)l[u]));
      }
    }
    for (; k <= i; k += 64) {
      const T l = row[k];
      acc += (double)l * (y64[k] - mean_c);
      m = fmaxf(m, fabsf((float)l));
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) white[i] = (T)acc;
  if (amax_rows != nullptr) {
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) amax_rows[i] = m;
  }
}

// alpha[j] = sum_{i>=j} Linv[i][j] white[i].  Stage 1: block (column block cb, row chunk rc of 64
// rows) -> part[rc][j] (double); stage 2 sums the chunks in order (deterministic).  (Chunks of 256 rows
// left a thread 64 dependent loads: 22 us at N = 2048 for 8 MB.)
// The same pass also sums Linv[i][j]^2 over i: the squared column norms of L^-1 are the diagonal of
// (K + noise I)^-1, which the precision self-test needs (the loads are shared, the extra FMA is free).
template <typename T>
__global__ __launch_bounds__(256) void alpha_part_kernel(const T* __restrict__ linv,
                                                         const T* __restrict__ white, int64_t n,
                                                         int64_t npad, double* __restrict__ part,
                                                         double* __restrict__ part_sq) {
  __shared__ double sh[2][4][64];
  const int g = threadIdx.x >> 6, c = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.y * kAlphaChunk;
  double acc = 0.0, sq = 0.0;
  if (r0 + kAlphaChunk > (int64_t)blockIdx.x * 64 && j < n) {  // chunk reaches below this column block
    const int64_t hi = min(n, r0 + kAlphaChunk);
#pragma unroll 4
    for (int64_t i = r0 + g; i < hi; i += 4)
      if (i >= j) {
        const double l = (double)linv[i * npad + j];
        acc = fma(l, (double)white[i], acc);
        sq = fma(l, l, sq);
      }
  }
  sh[0][g][c] = acc;
  sh[1][g][c] = sq;
  __syncthreads();
  if (g == 0) {
    part[(int64_t)blockIdx.y * npad + j] = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
    part_sq[(int64_t)blockIdx.y * npad + j] = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
  }
}

template <typename T>
__device__ __forceinline__ void nlml_block(const T* __restrict__ white, int64_t n, const double* __restrict__ diag64,
                                           double* __restrict__ out);
// (the chunks are summed in order -- deterministic -- but their loads are issued eight at a time: 8 workgroups with
// 64 dependent loads per thread took 11.8 us at N = 2048)
// nlml_out != nullptr: one workgroup more than the columns need, which sums the NLML (white and the diagonal of L are
// complete when this kernel starts) -- a launch of its own cost 5 us behind this one
// alpha_p: the predict-type copy of alpha (float, or double when alpha_p_f64) written in the same pass
template <typename T>
__global__ __launch_bounds__(256) void alpha_sum_kernel(const double* __restrict__ part,
                                                        const double* __restrict__ part_sq, int nchunk,
                                                        int64_t npad, T* __restrict__ alpha,
                                                        double* __restrict__ kinv_diag, void* __restrict__ alpha_p,
                                                        int alpha_p_f64, const float* __restrict__ amax_rows,
                                                        float* __restrict__ amax_out, const T* __restrict__ white,
                                                        int64_t n, const double* __restrict__ diag64,
                                                        double* __restrict__ nlml_out) {
  if (nlml_out != nullptr && blockIdx.x == gridDim.x - 1) {  // (workgroup-uniform)
    nlml_block<T>(white, n, diag64, nlml_out);
    return;
  }
  if (amax_rows != nullptr && blockIdx.x == 0) {  // max |L^-1| from white_kernel's row maxima (N_pad >= 256 here)
    __shared__ float shm[4];
    float m = 0.0f;
    for (int64_t r = threadIdx.x; r < npad; r += 256) m = fmaxf(m, amax_rows[r]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax_out[0] = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
  }
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npad) return;
  double acc = 0.0, sq = 0.0;
  int c = 0;
  for (; c + 8 <= nchunk; c += 8) {
    double a[8], q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = part[(int64_t)(c + u) * npad + j];
      q[u] = part_sq[(int64_t)(c + u) * npad + j];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc += a[u];
      sq += q[u];
    }
  }
  for (; c < nchunk; ++c) {
    acc += part[(int64_t)c * npad + j];
    sq += part_sq[(int64_t)c * npad + j];
  }
  const T al = (T)acc;
  alpha[j] = al;
  kinv_diag[j] = sq;
  if (alpha_p != nullptr) {
    if (alpha_p_f64) static_cast<double*>(alpha_p)[j] = (double)al;
    else static_cast<float*>(alpha_p)[j] = (float)al;
  }
}

template <typename T>
__device__ __forceinline__ void nlml_block(const T* __restrict__ white, int64_t n, const double* __restrict__ diag64,
                                           double* __restrict__ out) {
  __shared__ double sh[8];
  double acc = 0.0, lg = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double a = (double)white[i];
    acc += a * a;
    lg += log(diag64[i]);
  }
  acc = wave_sum(acc);
  lg = wave_sum(lg);
  if ((threadIdx.x & 63) == 0) {
    sh[threadIdx.x >> 6] = acc;
    sh[4 + (threadIdx.x >> 6)] = lg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double quad = sh[0] + sh[1] + sh[2] + sh[3];
    const double ld = sh[4] + sh[5] + sh[6] + sh[7];
    out[0] = 0.5 * quad + ld + 0.5 * (double)n * 1.83787706640934548356;  // log(2 pi)
  }
}

template <typename T>
void launch_solve_alpha(hipStream_t st, const T* linv, const double* y64, int64_t n, int64_t npad,
                        double mean_c, const double* diag64, T* white, T* alpha,
                        double* alpha_part, double* kinv_diag, double* nlml_out, void* alpha_p, int alpha_p_f64,
                        float* amax_rows, float* linv_absmax) {
  if (linv_absmax == nullptr) amax_rows = nullptr;
  hipLaunchKernelGGL((white_kernel<T>), dim3((unsigned)(npad / 4)), dim3(256), 0, st, linv, y64, n,
                     npad, mean_c, white, amax_rows);
  const int nchunk = (int)((npad + kAlphaChunk - 1) / kAlphaChunk);
  double* part_sq = alpha_part + (size_t)nchunk * npad;
  hipLaunchKernelGGL((alpha_part_kernel<T>), dim3((unsigned)(npad / 64), (unsigned)nchunk), dim3(256),
                     0, st, linv, white, n, npad, alpha_part, part_sq);
  hipLaunchKernelGGL((alpha_sum_kernel<T>), dim3((unsigned)((npad + 255) / 256) + (nlml_out ? 1u : 0u)), dim3(256), 0, st,
                     alpha_part, part_sq, nchunk, npad, alpha, kinv_diag, alpha_p, alpha_p_f64, amax_rows, linv_absmax,
                     white, n, diag64, nlml_out);
}
template void launch_solve_alpha<float>(hipStream_t, const float*, const double*, int64_t, int64_t, double, const double*, float*, float*, double*, double*, double*, void*, int, float*, float*);
template void launch_solve_alpha<double>(hipStream_t, const double*, const double*, int64_t, int64_t, double, const double*, double*, double*, double*, double*, double*, void*, int, float*, float*);

// =============================================================================================
// analytic gradient of the NLML  
// =============================================================================================
// one block per lower 64x64 tile (ti >= tj); partial[(blk) * (n_ls + 2) + h], h: ls..., variance, noise
template <typename T>
__global__ __launch_bounds__(256) void grad_tile_kernel(const T* __restrict__ kinv,
                                                        const T* __restrict__ alpha,
                                                        const double* __restrict__ xs,
                                                        const double* __restrict__ xnorm, int64_t n,
                                                        int64_t npad, int dp, int n_ls,
                                                        const double* __restrict__ ls, int kernel,
                                                        double variance,
                                                        double* __restrict__ partial) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // rows of the two 64-point blocks at an ODD stride (in doubles): a wave reads xj[jj][k] for 64 consecutive jj -- at
  // stride D_pad (a multiple of 4) the 64 lanes fell on 8 (D_pad = 12) or 4 (D_pad = 40) distinct bank pairs, an 8- to
  // 16-way conflict on every operand of the distance loop (C5: 5.4 ms of a 32 ms NLML + gradient evaluation)
  const int ds = dp | 1;
  double* xi = reinterpret_cast<double*>(lds_raw);  // [64][ds]
  double* xj = xi + 64 * ds;                        // [64][ds]
  __shared__ double red[4];
  // 1-D grid over the lower tiles only (a 2-D grid with the upper half exiting at once leaves the 8
  // XCDs unevenly loaded): blk -> (ti, tj <= ti), row-major
  const int64_t blk = blockIdx.x;
  int ti = (int)((__builtin_sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= blk) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > blk) --ti;
  const int tj = (int)(blk - (int64_t)ti * (ti + 1) / 2);
  const int H = n_ls + 2;
  for (int e = threadIdx.x; e < 64 * dp; e += 256) {
    const int r = e / dp, c = e - r * dp;
    xi[r * ds + c] = xs[(int64_t)ti * 64 * dp + e];
    xj[r * ds + c] = xs[(int64_t)tj * 64 * dp + e];
  }
  __syncthreads();
  double g_var = 0.0, g_noise = 0.0, g_iso = 0.0;
  double base[16];  // w * W * dk/dr2 per entry, for the ARD passes
  // A thread's 16 entries share their column (jj = lane) and take the rows wave, wave + 4, ...: the distance loop runs
  // over the dimensions ONCE for all of them -- one read of xj[jj][k] and sixteen wave-uniform (broadcast) reads of
  // xi[.][k] per dimension instead of thirty-two reads; every entry's sums run over k in the same order as before.
  double s_all[16], r2d_all[16];  // x.x' and the squared distance from direct differences (>= 0, free of cancellation)
#pragma unroll
  for (int p = 0; p < 16; ++p) s_all[p] = r2d_all[p] = 0.0;
  {
    const int jj = threadIdx.x & 63, i0 = threadIdx.x >> 6;
    for (int k = 0; k < dp; ++k) {
      const double b = xj[jj * ds + k];
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const double a = xi[(i0 + 4 * p) * ds + k];
        s_all[p] += a * b;
        const double df = a - b;
        r2d_all[p] = fma(df, df, r2d_all[p]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int idx = threadIdx.x + 256 * p;
    const int ii = idx >> 6, jj = idx & 63;
    const int64_t i = (int64_t)ti * 64 + ii, j = (int64_t)tj * 64 + jj;
    base[p] = 0.0;
    if (i >= n || j >= n || j > i) continue;
    const double w = (i == j) ? 1.0 : 2.0;
    const double s = s_all[p], r2d = r2d_all[p];
    const double r2 = -2.0 * s + (xnorm[i] + xnorm[j]);  // GEMM form (in double, as gram_kernel): the K the loss saw
    const double ai = (double)alpha[i], aj = (double)alpha[j];
    const double W = 0.5 * ((double)kinv[i * npad + j] - ai * aj);
    // the derivative is taken at the direct-difference distance: the Matern-1/2 factor 1/r would
    // otherwise amplify a GEMM-form r^2 that cancelled to ~0 (or to the 1e-36 clamp) between
    // near-coincident points in float32, while the per-dimension terms below use direct differences
    double kv, dk;
    kern_and_dkern_lean(kernel, r2, r2d, variance, kv, dk);
    g_var += w * W * kv / variance;
    if (i == j) g_noise += W;
    base[p] = w * W * dk;
    g_iso += base[p] * (-2.0 * r2d);
  }
  auto block_sum = [&](double v) -> double {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  const double sv = block_sum(g_var);
  const double sn = block_sum(g_noise);
  if (n_ls == 1) {
    const double sl = block_sum(g_iso) / ls[0];
    if (threadIdx.x == 0) partial[blk * H + 0] = sl;
  } else {
    for (int d = 0; d < n_ls; ++d) {
      double acc = 0.0;
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int idx = threadIdx.x + 256 * p;
        const int ii = idx >> 6, jj = idx & 63;
        const double df = xi[ii * ds + d] - xj[jj * ds + d];
        acc += base[p] * (-2.0 * df * df);
      }
      const double sd = block_sum(acc) / ls[d];
      if (threadIdx.x == 0) partial[blk * H + d] = sd;
    }
  }
  if (threadIdx.x == 0) {
    partial[blk * H + n_ls] = sv;
    partial[blk * H + n_ls + 1] = sn;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void grad_final_kernel(const double* __restrict__ partial,
                                                         int64_t nblk, int n_ls,
                                                         const T* __restrict__ alpha, int64_t n,
                                                         double* __restrict__ grad_out) {
  __shared__ double red[4];
  const int H = n_ls + 2;
  for (int h = 0; h <= H; ++h) {
    double acc = 0.0;
    if (h < H) {
      for (int64_t b = threadIdx.x; b < nblk; b += blockDim.x) acc += partial[b * H + h];
    } else {
      for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc -= (double)alpha[i];
    }
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) grad_out[h] = red[0] + red[1] + red[2] + red[3];
  }
}

template <typename T>
void launch_gradient(hipStream_t st, const T* linv, const T* alpha, const double* xs, const double* xnorm,
                     int64_t n, int64_t npad, int d, int dp, int n_ls, const double* ls,
                     const KernParams& kp, T* kinv, bool kinv_ready, double* partial, double* grad_out,
                     const FitPlanes* xt_planes) {
  (void)d;
  if (!kinv_ready && sizeof(T) == 4 && xt_planes != nullptr && xt_planes->XT != nullptr) {
    // float fit whose inverse was built on the bf16 matrix cores: the planes of L^-T are resident, and
    // Kinv = L^-T (L^-T)^T is a product of k-contiguous rows of them (lower tiles, k >= 128 ti)
    if constexpr (sizeof(T) == 4) {
      GemmBf16Desc g{};
      g.A = g.B = Bf16Planes{xt_planes->XT, xt_planes->stride, xt_planes->nkb, xt_planes->sX};
      g.np = xt_planes->np;
      g.C = reinterpret_cast<float*>(kinv);
      g.ldc = npad;
      g.m = g.n = g.k = (int)npad;
      g.alpha = 1.0f;
      g.beta = 0;
      g.lower_only = 1;
      g.kmode = 2;
      g.nbatch = 1;
      launch_gemm_bf16(st, g);
    }
  } else if (!kinv_ready) {
    // Kinv = Linv^T Linv, lower tiles:  opA(i,k) = Linv[k][i],  opB(k,j) = Linv[k][j],  k >= 64 ti
    GemmDesc g{};
    g.A = linv; g.sai = 1; g.sak = npad;
    g.B = linv; g.sbk = npad; g.sbj = 1;
    g.C = kinv; g.ldc = npad;
    g.m = (int)npad; g.n = (int)npad; g.k = (int)npad; g.m_last = (int)npad; g.nbatch = 1;
    g.alpha = 1.0; g.beta = 0.0; g.lower_only = 1; g.kmode = 2;
    launch_gemm<T>(st, g);
  }
  const int nt = (int)(npad / 64);
  const size_t lds = (size_t)2 * 64 * (dp | 1) * sizeof(double);
  const int64_t nblk = (int64_t)nt * (nt + 1) / 2;
  hipLaunchKernelGGL((grad_tile_kernel<T>), dim3((unsigned)nblk), dim3(256), lds, st,
                     kinv, alpha, xs, xnorm, n, npad, dp, n_ls, ls, kp.kernel, kp.variance, partial);
  hipLaunchKernelGGL((grad_final_kernel<T>), dim3(1), dim3(256), 0, st, partial, nblk, n_ls, alpha, n,
                     grad_out);
}
template void launch_gradient<float>(hipStream_t, const float*, const float*, const double*, const double*, int64_t, int64_t, int, int, int, const double*, const KernParams&, float*, bool, double*, double*, const FitPlanes*);
template void launch_gradient<double>(hipStream_t, const double*, const double*, const double*, const double*, int64_t, int64_t, int, int, int, const double*, const KernParams&, double*, bool, double*, double*, const FitPlanes*);

// =============================================================================================
// fused fit for N <= 128: ONE launch, one workgroup, everything in LDS
// =============================================================================================
// The reference's real problem sizes are N = 5 .. ~100 (14 fits with N = 5 .. 52 in the
// toy run, each 10 - 45 L-BFGS-B loss evaluations, gpso/gp_surrogate.py:496-503).  There the general
// path is pure launch latency: 17 launches, 0.10 ms of device time, per evaluation.  This kernel does
// the whole evaluation -- scale X, Gram, Cholesky, L^-1, alpha, NLML, K^-1 and the analytic gradient,
// the tile packing the predict kernels read -- in a single workgroup with the matrices in LDS as
// 64x64 double blocks (A, B, C, D4 below), whatever the context's matrix type T (the outputs are
// rounded to T / TP on the way out).  Block algebra for 64 < N <= 128 (N <= 64: only the first line):
//     K00 = L00 L00^T ;  X00 = L00^-1
//     L10 = K10 X00^T ;  S = K11 - L10 L10^T = L11 L11^T ;  X11 = L11^-1 ;  X10 = -X11 (L10 X00)
// The 64-pivot chains (chol64_lds / trinv64_lds, shared with the general path) are what is left on
// the critical path: ~15 us each.
#ifndef GPSO_SSTAMP
#define GPSO_SSTAMP(i)  // tools/micro/small_phases.hip defines this to record s_memtime stamps
#endif
constexpr int kSmallN = 128;
constexpr int kBlk = kFitBlock * kDS;  // doubles per 64x64 LDS block
constexpr int kSmallLdsDoubles = 4 * kBlk + kTsDoubles + 6 * kSmallN + 4 * (kGradMaxLs + 2) + 16;
constexpr int kSmallLdsBytes = kSmallLdsDoubles * 8;

// 64x64 Gram block (bi, bj) from the scaled inputs in LDS (xs[row][dp]) into blk (stride kDS): the
// same MFMA sequence and epilogue as gram_kernel<double>, so the entries are bit-identical to it
__device__ __forceinline__ void small_gram_block(const double* xs, const double* nrm, int dp, int bi, int bj,
                                                 int n, int kernel, double variance, double noise,
                                                 double* blk, int wave, int lane) {
  // (a diagonal block is only read on and below its diagonal: column tiles t <= wave)
  const int tmax = (bi == bj) ? wave : 3;
  f64x4 s[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) s[t] = f64x4{0, 0, 0, 0};
  const int i0 = 64 * bi + 16 * wave, j0 = 64 * bj;
  for (int c = 0; c < dp / 4; ++c) {
    const double a = xs[(i0 + (lane & 15)) * dp + 4 * c + (lane >> 4)];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t > tmax) continue;
      const double b = xs[(j0 + 16 * t + (lane & 15)) * dp + 4 * c + (lane >> 4)];
      s[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, s[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (t > tmax) continue;
      const int ii = 16 * wave + (lane >> 4) + 4 * r, jj = 16 * t + (lane & 15);
      const int i = 64 * bi + ii, j = 64 * bj + jj;
      double v;
      if (i < n && j < n) {
        const double r2 = fma(-2.0, s[t][r], nrm[i] + nrm[j]);
        v = kern_from_r2_lean(kernel, r2, variance);
        if (i == j) v += noise;
      } else {
        v = (i == j) ? 1.0 : 0.0;
      }
      blk[ii * kDS + jj] = v;
    }
}

// this wave's 4 tiles (row strip `wave`, column tiles 0..3) of a 64x64x64 product of LDS blocks:
// out[t] = sum_k A[16 wave + i][k] * Bop(k, 16 t + j);  Bop(k, j) = Bb[k * sbk + j * sbj]
__device__ __forceinline__ void small_mm_strip(const double* Ab, const double* Bb, int sbk, int sbj, int wave,
                                               int lane, f64x4 (&out)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
    out[t] = mma16_lds<64>(Ab + 16 * wave * kDS, kDS, 1, Bb + 16 * t * sbj, sbk, sbj, lane);
}
__device__ __forceinline__ void small_store_strip(double* blk, int wave, int lane, const f64x4 (&v)[4], double sign) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) blk[(16 * wave + (lane >> 4) + 4 * r) * kDS + 16 * t + (lane & 15)] = sign * v[t][r];
}

template <typename T, typename TP>
__global__ __launch_bounds__(256) void small_fit_kernel(SmallFitArgs g) {
  extern __shared__ __align__(32) unsigned char lds_raw[];
  double* A = reinterpret_cast<double*>(lds_raw);
  double* B = A + kBlk;
  double* C = B + kBlk;
  double* D4 = C + kBlk;
  double* Ts = D4 + kBlk;
  double* nrm = Ts + kTsDoubles;       // [128] squared norms of the scaled inputs
  double* resid = nrm + kSmallN;       // [128] y - c
  double* wht = resid + kSmallN;       // [128] a = L^-1 (y - c)
  double* alp = wht + kSmallN;         // [128] alpha
  double* dg = alp + kSmallN;          // [128] diagonal of L
  double* kd = dg + kSmallN;           // [128] diag(K_y^-1)
  double* gacc = kd + kSmallN;         // [4][kGradMaxLs + 2] per-wave gradient partials
  double* red = gacc + 4 * (kGradMaxLs + 2);  // [16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = g.n, dp = g.dp, nb = (n > 64) ? 2 : 1, rows = 64 * nb;
  int* info = reinterpret_cast<int*>(g.scal + 1);
  T* Lf = static_cast<T*>(g.Lf);
  T* linv = static_cast<T*>(g.linv);
  constexpr int64_t ld = kSmallN;

  GPSO_SSTAMP(0);
  // ---- 0. scaled inputs: LDS (in B, free until X00 is formed) + the global copies the predict path reads
  double* xs = B;
  if (tid == 0) *info = INT_MAX;
  if (tid < 8 + 48) {  // the hyper-parameter block (layout: api.hip set_theta)
    double h = 0.0;
    if (tid == 0) h = (double)n;
    else if (tid == 1) h = (double)g.d;
    else if (tid == 2) h = (double)g.kernel;
    else if (tid == 3) h = (double)g.n_ls;
    else if (tid == 4) h = g.variance;
    else if (tid == 5) h = g.noise;
    else if (tid == 6) h = g.mean_c;
    else if (tid >= 8) h = g.ls[tid - 8];
    g.hyper[tid] = h;
  }
  for (int e = tid; e < kSmallN * dp; e += 256) {
    const int i = e / dp, k = e % dp;
    double v = 0.0;
    if (i < n && k < g.d) v = g.x64[(int64_t)i * g.d + k] / g.ls[k];
    if (i < rows) xs[e] = v;
    g.xs64[e] = v;
  }
  __syncthreads();
  if (tid < kSmallN) {
    double acc = 0.0;
    if (tid < rows)
      for (int k = 0; k < dp; ++k) acc += xs[tid * dp + k] * xs[tid * dp + k];
    nrm[tid] = acc;
    g.xnorm64[tid] = acc;
    resid[tid] = (tid < n) ? g.y64[tid] - g.mean_c : 0.0;
  }
  {  // MFMA A-fragment packing (pack_xs_kernel<double>)
    const int dp4 = dp / 4;
    for (int idx = tid; idx < kSmallN * dp; idx += 256) {
      const int l = idx & 63, q = idx >> 6, c = q % dp4, kt = q / dp4;
      const int row = kt * 16 + Mfma<double>::arow_for_k4(l & 15);
      g.xs_p64[idx] = (row < rows) ? xs[row * dp + 4 * c + (l >> 4)] : 0.0;
    }
  }
  __syncthreads();
  GPSO_SSTAMP(1);
  // ---- 1. Gram blocks: K00 -> A, K10 -> C, K11 -> D4
  small_gram_block(xs, nrm, dp, 0, 0, n, g.kernel, g.variance, g.noise, A, wave, lane);
  if (nb == 2) {
    small_gram_block(xs, nrm, dp, 1, 0, n, g.kernel, g.variance, g.noise, C, wave, lane);
    small_gram_block(xs, nrm, dp, 1, 1, n, g.kernel, g.variance, g.noise, D4, wave, lane);
  }
  __syncthreads();
  for (int e = tid; e < kBlk; e += 256) B[e] = 0.0;  // X00 is written on and below the diagonal only
  __syncthreads();
  // ---- 2. K00 = L00 L00^T, X00 = L00^-1 (B)
  GPSO_SSTAMP(2);
  chol64_lds<2, T>(A, B, Lf, ld, 0, n, info, Ts);
  if (tid < kFitBlock) dg[tid] = A[tid * kDS + tid];
  GPSO_SSTAMP(3);
  trinv64_lds<true, T>(A, B, Ts, Lf, ld);
  GPSO_SSTAMP(4);
  if (nb == 2) {
    // ---- 3. L10 = K10 X00^T -> C (and Lf[1][0])
    f64x4 v[4];
    small_mm_strip(C, B, 1, kDS, wave, lane, v);  // Bop(k, j) = X00[j][k]
    __syncthreads();
    small_store_strip(C, wave, lane, v, 1.0);
    for (int e = tid; e < kBlk; e += 256) A[e] = 0.0;  // L00 is dead: A becomes X11
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Lf[(int64_t)(64 + 16 * wave + (lane >> 4) + 4 * r) * ld + 16 * t + (lane & 15)] = (T)v[t][r];
    __syncthreads();
    // ---- 4. S = K11 - L10 L10^T (lower 16x16 tiles) in place in D4
    for (int t = wave; t < 10; t += 4) {
      int ti = 0, tj = t;
      while (tj > ti) {
        tj -= ti + 1;
        ++ti;
      }
      const f64x4 p = mma16_lds<64>(C + 16 * ti * kDS, kDS, 1, C + 16 * tj * kDS, 1, kDS, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) D4[(16 * ti + (lane >> 4) + 4 * r) * kDS + 16 * tj + (lane & 15)] -= p[r];
    }
    __syncthreads();
    // ---- 5. S = L11 L11^T, X11 = L11^-1 (A)
    GPSO_SSTAMP(5);
    chol64_lds<2, T>(D4, A, Lf + 64 * ld + 64, ld, 64, n, info, Ts);
    if (tid < kFitBlock) dg[64 + tid] = D4[tid * kDS + tid];
    GPSO_SSTAMP(6);
    trinv64_lds<true, T>(D4, A, Ts, Lf + 64 * ld + 64, ld);
    GPSO_SSTAMP(7);
    // ---- 6. X10 = -X11 (L10 X00) -> C
    small_mm_strip(C, B, kDS, 1, wave, lane, v);  // W = L10 X00
    __syncthreads();
    small_store_strip(D4, wave, lane, v, 1.0);    // L11 is dead: D4 holds W
    __syncthreads();
    small_mm_strip(A, D4, kDS, 1, wave, lane, v);  // X11 W
    __syncthreads();
    small_store_strip(C, wave, lane, v, -1.0);
  } else if (tid < kFitBlock) {
    dg[64 + tid] = 1.0;
  }
  __syncthreads();
  GPSO_SSTAMP(8);
  // X = L^-1 lives in LDS as 64x64 blocks (zeros above the diagonal are real zeros): X00 = B, and for
  // nb == 2 X10 = C, X11 = A.  Block (rb, cb) of X, or nullptr for the zero block (0, 1):
  auto Xblk = [&](int rb, int cb) -> const double* { return rb == 0 ? (cb == 0 ? B : nullptr) : (cb == 0 ? C : A); };
  // ---- 7. L^-1 to global: the lower triangle of rows < 64 nb (all any later reader looks at: the bf16
  //         packing and the debug getters read on / below the diagonal, rows < n) + the predict
  //         kernels' tile packing (every tile row of the 128-padded problem: rows >= n are zero)
  lower_tile_to_global<T>(B, linv, ld, tid);
  if (nb == 2) {
    lower_tile_to_global<T>(A, linv + 64 * ld + 64, ld, tid);
    for (int e = tid; e < 64 * 16; e += 256) {  // X10: the full block, 16-byte stores
      using vec4 = typename Mfma<T>::vec4;
      const int r = e >> 4, c = 4 * (e & 15);
      vec4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (T)C[r * kDS + c + q];
      *reinterpret_cast<vec4*>(linv + (int64_t)(64 + r) * ld + c) = v;
    }
  }
  {
    using vecP = typename Mfma<TP>::vec4;
    vecP* out = static_cast<vecP*>(g.linv_p);
    const int nrt = (n + 15) / 16;                       // tile rows that hold data
    const int nzt = max(nrt, min(g.zero_tile_rows, 8));  // ... and those a previous fit may have left non-zero
    for (int idx = tid; idx < nzt * (nzt + 1) / 2 * 64; idx += 256) {
      const int l = idx & 63, tile = idx >> 6;
      int rt = 0, kt = tile;
      while (kt > rt) {
        kt -= rt + 1;
        ++rt;
      }
      const int row = rt * 16 + (l & 15), col = kt * 16 + 4 * (l >> 4);
      vecP v{0, 0, 0, 0};
      const double* blk = (row < rows) ? Xblk(row >> 6, col >> 6) : nullptr;
      if (row < n && blk != nullptr) {
        const double* src = blk + (row & 63) * kDS + (col & 63);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r <= row && col + r < n) v[r] = (TP)src[r];
      }
      out[idx] = v;
    }
  }
  GPSO_SSTAMP(9);
  // ---- 8. a = X (y - c), alpha = X^T a, diag(K_y^-1) = column norms of X, NLML -- on the MFMA: a vector
  //         is fed as a B operand whose 16 columns are all that vector (column stride 0), the result is read
  //         from column 0; X^T is X read with swapped strides.  (Rows >= n of X are unit rows and the
  //         residual is zero there, so the padding drops out by itself.)
  for (int b = 0; b < nb; ++b) {  // a[64 b + 16 wave + .]
    f64x4 acc{0, 0, 0, 0};
    for (int cb = 0; cb <= b; ++cb)
      acc = mma16_lds_acc<64>(acc, Xblk(b, cb) + 16 * wave * kDS, kDS, 1, resid + 64 * cb, 1, 0, lane);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) wht[64 * b + 16 * wave + (lane >> 4) + 4 * r] = acc[r];
    }
  }
  if (nb == 1 && tid < 64) wht[64 + tid] = 0.0;
  __syncthreads();
  for (int b = 0; b < nb; ++b) {  // alpha[64 b + 16 wave + .] = sum over row blocks rb >= b of X[rb][b]^T a[rb]
    f64x4 acc{0, 0, 0, 0}, sq{0, 0, 0, 0};
    for (int rb = b; rb < nb; ++rb) {
      const double* xb = Xblk(rb, b) + 16 * wave;  // column strip of the block
      acc = mma16_lds_acc<64>(acc, xb, 1, kDS, wht + 64 * rb, 1, 0, lane);
      sq = mma16_lds_acc<64>(sq, xb, 1, kDS, xb, kDS, 1, lane);  // 16x16 diagonal tile of X^T X
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (lane >> 4) + 4 * r;
      if ((lane & 15) == 0) alp[64 * b + 16 * wave + i] = acc[r];
      if ((lane & 15) == i) kd[64 * b + 16 * wave + i] = sq[r];
    }
  }
  if (nb == 1 && tid < 64) {
    alp[64 + tid] = 0.0;
    kd[64 + tid] = 1.0;
  }
  __syncthreads();
  if (tid < kSmallN) {
    const bool live = tid < n;  // (padding rows: unit rows of X -- zero them like the general path does)
    static_cast<T*>(g.white)[tid] = (T)(live ? wht[tid] : 0.0);
    static_cast<T*>(g.alpha_f)[tid] = (T)(live ? alp[tid] : 0.0);
    static_cast<TP*>(g.alpha_p)[tid] = (TP)(live ? alp[tid] : 0.0);
    g.kinv_diag[tid] = live ? kd[tid] : 0.0;
    g.diag64[tid] = dg[tid];
  }
  {
    double q = 0.0, lg = 0.0, sa = 0.0;
    if (tid < n) {
      q = wht[tid] * wht[tid];
      lg = log(dg[tid]);
      sa = alp[tid];
    }
    q = wave_sum(q);
    lg = wave_sum(lg);
    sa = wave_sum(sa);
    if (lane == 0) {
      red[wave] = q;
      red[4 + wave] = lg;
      red[8 + wave] = sa;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const double quad = (red[0] + red[1]) + (red[2] + red[3]);
    const double ldet = (red[4] + red[5]) + (red[6] + red[7]);
    const double nl = 0.5 * quad + ldet + 0.5 * (double)n * 1.83787706640934548356;  // log(2 pi)
    const double gc = -((red[8] + red[9]) + (red[10] + red[11]));
    g.scal[0] = nl;
    if (g.want_grad) g.scal[8 + g.n_ls + 2] = gc;
    if (g.scal_host != nullptr) {
      g.scal_host[0] = nl;
      *reinterpret_cast<int*>(g.scal_host + 1) = atomicMin(info, INT_MAX);  // (the factorisation's verdict, as the device holds it)
      if (g.want_grad) g.scal_host[8 + g.n_ls + 2] = gc;
      if (!g.want_grad && g.done_token != 0.0) {  // (the last host store of an evaluation without gradient: this thread's)
        __threadfence_system();
        *reinterpret_cast<volatile double*>(g.scal_host + 7) = g.done_token;
      }
    }
  }
  GPSO_SSTAMP(10);
  if (!g.want_grad) return;
  // ---- 9. K^-1 = X^T X tile by tile on the MFMA, consumed at once by the gradient reductions
  //         (W = (K^-1 - alpha alpha^T) / 2; sums of W o dK/dtheta over the lower triangle)
  double* xg = D4;  // scaled inputs again (D4 is dead)
  for (int e = tid; e < rows * dp; e += 256) {
    const int i = e / dp, k = e % dp;
    xg[e] = (i < n && k < g.d) ? g.x64[(int64_t)i * g.d + k] / g.ls[k] : 0.0;
  }
  const int H = g.n_ls + 2;
  for (int h = tid; h < 4 * (kGradMaxLs + 2); h += 256) gacc[h] = 0.0;
  __syncthreads();
  const int nt16 = (n + 15) / 16;
  T* kinv = static_cast<T*>(g.kinv);
  double g_var = 0.0, g_noise = 0.0, g_iso = 0.0;  // per lane, over all tiles of this wave
  for (int t = wave; t < nt16 * (nt16 + 1) / 2; t += 4) {
    int ti = 0, tj = t;
    while (tj > ti) {
      tj -= ti + 1;
      ++ti;
    }
    // K^-1[ti][tj] = sum_{k >= 16 ti} X[k][16 ti + i] X[k][16 tj + j]: the part of the sum inside the row
    // block of tile ti (k-steps of 16 from the tile's own rows on), then the whole row block below it
    f64x4 acc{0, 0, 0, 0};
    {
      const int bi = ti >> 2, bj = tj >> 2;  // column blocks of the two tiles (bj <= bi)
      const double* xi = Xblk(bi, bi) + 16 * (ti & 3);
      const double* xj = Xblk(bi, bj) + 16 * (tj & 3);
      for (int kq = (ti & 3); kq < 4; ++kq)
        acc = mma16_lds_acc<16>(acc, xi + 16 * kq * kDS, 1, kDS, xj + 16 * kq * kDS, kDS, 1, lane);
      if (nb == 2 && bi == 0)
        acc = mma16_lds_acc<64>(acc, C + 16 * (ti & 3), 1, kDS, C + 16 * (tj & 3), kDS, 1, lane);
    }
    // squared distances of this lane's four entries (rows i_r, column j) from direct differences: the
    // column point is read once per dimension and shared by the four rows
    double base[4], r2d[4] = {0.0, 0.0, 0.0, 0.0};
    const int jcol = 16 * tj + (lane & 15), irow0 = 16 * ti + (lane >> 4);
    for (int c = 0; c < dp; c += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double xj = xg[jcol * dp + c + q];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double df = xg[(irow0 + 4 * r) * dp + c + q] - xj;
          r2d[r] = fma(df, df, r2d[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = irow0 + 4 * r, j = jcol;
      base[r] = 0.0;
      if (kinv != nullptr && i < kSmallN) kinv[(int64_t)i * ld + j] = (T)acc[r];
      if (i >= n || j >= n || j > i) continue;
      const double w = (i == j) ? 1.0 : 2.0;
      const double Wij = 0.5 * (acc[r] - alp[i] * alp[j]);
      double kv, dk;
      kern_and_dkern_same(g.kernel, r2d[r], g.variance, kv, dk);
      g_var += w * Wij * kv / g.variance;
      if (i == j) g_noise += Wij;
      base[r] = w * Wij * dk;
      g_iso += base[r] * (-2.0 * r2d[r]);
    }
    if (g.n_ls > 1) {
      for (int dd = 0; dd < g.n_ls; ++dd) {
        double a2 = 0.0;
        const double xj = xg[jcol * dp + dd];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double df = xg[(irow0 + 4 * r) * dp + dd] - xj;
          a2 += base[r] * (-2.0 * df * df);
        }
        a2 = wave_sum(a2);
        if (lane == 0) gacc[wave * (kGradMaxLs + 2) + dd] += a2;
      }
    }
  }
  // the sums every kernel needs: one reduction per wave, after its last tile
  g_var = wave_sum(g_var);
  g_noise = wave_sum(g_noise);
  g_iso = wave_sum(g_iso);
  if (lane == 0) {
    gacc[wave * (kGradMaxLs + 2) + g.n_ls] = g_var;
    gacc[wave * (kGradMaxLs + 2) + g.n_ls + 1] = g_noise;
    if (g.n_ls == 1) gacc[wave * (kGradMaxLs + 2)] = g_iso;
  }
  __syncthreads();
  if (tid < H) {
    double v = (gacc[tid] + gacc[(kGradMaxLs + 2) + tid]) + (gacc[2 * (kGradMaxLs + 2) + tid] + gacc[3 * (kGradMaxLs + 2) + tid]);
    if (tid < g.n_ls) v /= g.ls[g.n_ls == 1 ? 0 : tid];
    g.scal[8 + tid] = v;
    if (g.scal_host != nullptr) g.scal_host[8 + tid] = v;
  }
  if (g.scal_host != nullptr && g.done_token != 0.0) {  // every writer releases at system scope, then the token
    __threadfence_system();
    __syncthreads();
    if (tid == 0) *reinterpret_cast<volatile double*>(g.scal_host + 7) = g.done_token;
  }
  GPSO_SSTAMP(11);
}

bool small_fit_eligible(int64_t n, int dp) { return n <= kSmallN && (n <= 64 || dp <= 32); }

template <typename T, typename TP>
int launch_small_fit(hipStream_t st, const SmallFitArgs& args) {
  const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&small_fit_kernel<T, TP>), kSmallLdsBytes);
  if (rc) return rc;
  hipLaunchKernelGGL((small_fit_kernel<T, TP>), dim3(1), dim3(256), kSmallLdsBytes, st, args);
  return 0;
}
template int launch_small_fit<double, double>(hipStream_t, const SmallFitArgs&);
template int launch_small_fit<double, float>(hipStream_t, const SmallFitArgs&);
template int launch_small_fit<float, float>(hipStream_t, const SmallFitArgs&);

// =============================================================================================
// precision self-test (float-predict contexts): the predict path at the training inputs against the
// closed form the fit implies
// =============================================================================================
// With K_y = K + noise I = L L^T, alpha = K_y^-1 (y - c) and k_i = K_y e_i - noise e_i:
//     mean(x_i) = k_i . alpha + c                       = y_i - noise * alpha_i
//     var_y(x_i) = sigma^2 - |L^-1 k_i|^2 + noise       = 2 noise - noise^2 * (K_y^-1)_ii
// (K_y^-1)_ii = squared column norm of L^-1 (alpha_sum_kernel).  The predict kernels recompute k_i
// from X, so the differences measure the whole chain: the rounding of the apply (the training inputs
// are where var is smallest, i.e. where the cancellation sigma^2 - |A|^2 is worst) and, for a float
// factorisation, its BACKWARD error (at x_i the solve weights K_y^-1 k_i are nearly a unit vector; at a
// general leaf they amplify a factorisation error by up to |K_y^-1| sigma^2, which is why max_i
// (K_y^-1)_ii is reported as well: the host multiplies the measured errors by it for float factors).
// out: [0] max |d mean|, [1] max |d var|, [2] max |y - c|, [3] min predicted var, [4] max |alpha|,
// [5] max_i (K_y^-1)_ii.
template <typename T>
__global__ __launch_bounds__(256) void selftest_kernel(const double* __restrict__ mean,
                                                       const double* __restrict__ var,
                                                       const double* __restrict__ y64,
                                                       const T* __restrict__ alpha,
                                                       const double* __restrict__ kinv_diag, int64_t n,
                                                       double noise, double mean_c,
                                                       double* __restrict__ out) {
  __shared__ double sh[6][4];
  double v[6] = {0.0, 0.0, 0.0, 1.0e300, 0.0, 0.0};
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double a = (double)alpha[i];
    const double m_ref = y64[i] - noise * a;
    const double v_ref = 2.0 * noise - noise * noise * kinv_diag[i];
    const double dm = fabs(mean[i] - m_ref), dv = fabs(var[i] - v_ref);
    v[0] = (dm > v[0] || dm != dm) ? dm : v[0];  // NaN sticks
    v[1] = (dv > v[1] || dv != dv) ? dv : v[1];
    v[2] = fmax(v[2], fabs(y64[i] - mean_c));
    v[3] = fmin(v[3], var[i]);
    v[4] = fmax(v[4], fabs(a));
    v[5] = fmax(v[5], kinv_diag[i]);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    for (int off = 32; off > 0; off >>= 1) {
      const double o = __shfl_xor(v[q], off);
      if (q == 3) v[q] = fmin(v[q], o);
      else v[q] = (o > v[q] || o != o) ? o : v[q];
    }
    if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 0; q < 6; ++q) {
      double r = sh[q][0];
      for (int w = 1; w < 4; ++w) {
        const double o = sh[q][w];
        if (q == 3) r = fmin(r, o);
        else r = (o > r || o != o) ? o : r;
      }
      out[q] = r;
    }
  }
}

template <typename T>
void launch_selftest(hipStream_t st, const double* mean, const double* var, const double* y64,
                     const T* alpha, const double* kinv_diag, int64_t n, double noise, double mean_c,
                     double* out) {
  hipLaunchKernelGGL((selftest_kernel<T>), dim3(1), dim3(256), 0, st, mean, var, y64, alpha, kinv_diag, n,
                     noise, mean_c, out);
}
template void launch_selftest<float>(hipStream_t, const double*, const double*, const double*, const float*, const double*, int64_t, double, double, double*);
template void launch_selftest<double>(hipStream_t, const double*, const double*, const double*, const double*, const double*, int64_t, double, double, double*);

// =============================================================================================
// conversions / interop
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void convert_in_kernel(const double* __restrict__ src,
                                                         T* __restrict__ dst, int64_t rows,
                                                         int64_t cols, int64_t ld_dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  dst[(idx / cols) * ld_dst + idx % cols] = (T)src[idx];
}
template <typename T>
void launch_convert_in(hipStream_t st, const double* src, T* dst, int64_t rows, int64_t cols,
                       int64_t ld_dst) {
  const int64_t total = rows * cols;
  if (total == 0) return;
  hipLaunchKernelGGL((convert_in_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, src, dst, rows, cols, ld_dst);
}
template void launch_convert_in<float>(hipStream_t, const double*, float*, int64_t, int64_t, int64_t);
template void launch_convert_in<double>(hipStream_t, const double*, double*, int64_t, int64_t, int64_t);

template <typename T>
__global__ __launch_bounds__(256) void convert_out_kernel(const T* __restrict__ src, int64_t ld_src,
                                                          double* __restrict__ dst, int64_t rows,
                                                          int64_t cols, int lower_only) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int64_t i = idx / cols, j = idx % cols;
  double v = (double)src[i * ld_src + j];
  if (lower_only == 1 && j > i) v = 0.0;
  if (lower_only == 2 && j > i) v = (double)src[j * ld_src + i];  // mirror a symmetric lower matrix
  dst[idx] = v;
}
template <typename T>
void launch_convert_out(hipStream_t st, const T* src, int64_t ld_src, double* dst, int64_t rows,
                        int64_t cols, int lower_only) {
  const int64_t total = rows * cols;
  if (total == 0) return;
  hipLaunchKernelGGL((convert_out_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, src, ld_src, dst, rows, cols, lower_only);
}
template void launch_convert_out<float>(hipStream_t, const float*, int64_t, double*, int64_t, int64_t, int);
template void launch_convert_out<double>(hipStream_t, const double*, int64_t, double*, int64_t, int64_t, int);

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_vec_kernel(const TS* __restrict__ src, TD* __restrict__ dst,
                                                          int64_t len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) dst[i] = (TD)src[i];
}
template <typename TS, typename TD>
void launch_convert_vec(hipStream_t st, const TS* src, TD* dst, int64_t len) {
  if (len <= 0) return;
  hipLaunchKernelGGL((convert_vec_kernel<TS, TD>), dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, src,
                     dst, len);
}
template void launch_convert_vec<double, float>(hipStream_t, const double*, float*, int64_t);
template void launch_convert_vec<float, float>(hipStream_t, const float*, float*, int64_t);
template void launch_convert_vec<double, double>(hipStream_t, const double*, double*, int64_t);

template <typename T>
__global__ __launch_bounds__(256) void install_chol_kernel(const double* __restrict__ L64, int64_t n,
                                                           int64_t npad, T* __restrict__ K) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npad * npad) return;
  const int64_t i = idx / npad, j = idx % npad;
  T v = 0;
  if (i < n && j <= i) v = (T)L64[i * n + j];
  if (i >= n && i == j) v = 1;
  K[idx] = v;
}
template <typename T>
void launch_install_chol(hipStream_t st, const double* L64, int64_t n, int64_t npad, T* K, T* linv) {
  const int64_t total = npad * npad;
  hipLaunchKernelGGL((install_chol_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, L64, n, npad, K);
  hipLaunchKernelGGL((trinv_diag_kernel<T>), dim3((unsigned)(npad / kFitBlock)), dim3(256), 0, st, K,
                     linv, npad);
}
template void launch_install_chol<float>(hipStream_t, const double*, int64_t, int64_t, float*, float*);
template void launch_install_chol<double>(hipStream_t, const double*, int64_t, int64_t, double*, double*);

}  // namespace gpso
