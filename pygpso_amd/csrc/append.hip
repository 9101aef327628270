// Rank-k append at FIXED hyper-parameters (gpso/gp_surrogate.py:496-498 sets `model.data = (x, y)`
// and the reference then re-optimises and refactorises from scratch, O(N^3), although N grew by 1-7 points).
//
// With the posterior of the first n points resident (L11, X11 = L11^-1, a1 = X11 (y1 - c), alpha1, diag(K_y^-1)) and k new
// points (k <= 64, n + k <= N_pad), the extended factor is
//
//     L = | L11  0  |     L21 = K21 L11^-T = (X11 K12)^T        S = K22 + noise I - L21 L21^T = L22 L22^T
//         | L21 L22 |     X   = | X11             0     |      a2 = L22^-1 (y2 - c - L21 a1)
//                               | -L22^-1 L21 X11 L22^-1 |     alpha = X^T a,  nlml += 1/2 |a2|^2 + sum log diag L22 + k/2 log 2 pi
//
// i.e. TWO passes over the resident L^-1 (row pass: B = X11 K12; column pass: W = B^T X11) instead of a factorisation:
// N^2 s bytes and 2 N^2 k flops -- HBM-bound, against N^3 / 3 flops.  All arithmetic is double whatever the matrix type
// TF (the loads are TF, the k-wide accumulators double): the passes are bandwidth-bound, and S = K22 - L21 L21^T is a
// cancellation the float fit also pays for.  Kernels (one stream, in order; round 6: the new points are read from pinned host
// memory by the cross kernel itself -- no copies in front of the first launch -- and the repack of the predict-ready copies is one
// launch instead of two):
//
//   append_cross_kernel      K12 (n x k) and K22 + noise I (k x k) from the scaled inputs; the new rows' scaled inputs,
//                            norms and MFMA fragments land where the fit leaves them (same formulas as scale_x_kernel /
//                            gram_kernel: GEMM-form r^2 in double)
//   append_pass_kernel<.., false>  row pass, tile by tile: B[r][:] = sum_{c <= r} X11[r][c] K12[c][:]  (partials per tile chunk)
//   append_gram_part_kernel  B from its chunks; partial [B | a1]^T [B | a1] per 64-row block  (L21 L21^T and L21 a1 in one go)
//   append_chol_kernel       ONE workgroup: S, its Cholesky (first failing pivot -> info = n + p), L22^-1, a2, nlml
//   append_pass_kernel<.., true>   column pass: W[:, c] = sum_{r >= c} B[r][:] X11[r][c]
//   append_finish_kernel     R = -L22^-1 W: the new rows of L^-1 and L, alpha1 += R^T a2, diag(K_y^-1) += sum R^2,
//                            max |new entries| for the fp16 split's scale; one extra workgroup writes the k x k corner
//
// A failed pivot leaves the resident posterior of the n points untouched (append_finish_kernel exits on info).
#include <climits>

#include "common.hpp"
#include "kernels.hpp"

namespace gpso {

// ---- 1. cross-Gram -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void append_cross_kernel(AppendArgs a) {
  __shared__ double xn[kAppendMax * 48];  // scaled new rows [k][dp]
  __shared__ double nn[kAppendMax];       // their squared norms
  const int tid = threadIdx.x;
  const int k = a.k, dp = a.dp, d = a.d, KP = a.kp;
  const int64_t n = a.n;
  // (the new points come straight from the caller's staging buffer in pinned host memory: k d + k doubles per workgroup over
  // the fabric instead of two copy operations in front of the first launch; workgroup 0 files them in x64 / y64)
  for (int e = tid; e < k * dp; e += 256) {
    const int t = e / dp, kk = e - t * dp;
    xn[e] = (kk < d) ? a.xnew[t * d + kk] / a.ls[kk] : 0.0;
  }
  __syncthreads();
  if (tid < k) {
    double acc = 0.0;
    for (int kk = 0; kk < dp; ++kk) acc += xn[tid * dp + kk] * xn[tid * dp + kk];
    nn[tid] = acc;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    // the new rows of the scaled inputs, where scale_x_kernel<double> puts them (plain, norms, MFMA A fragments)
    const int dp4 = dp / 4;
    for (int e = tid; e < k * dp; e += 256) {
      const int t = e / dp, kk = e - t * dp;
      const int64_t i = n + t;
      a.xs64[i * dp + kk] = xn[e];
      const int64_t pbase = (i >> 4) * dp4 * 64 + Mfma<double>::arow_for_k4((int)(i & 15));
      a.xs_p64[pbase + (kk >> 2) * 64 + 16 * (kk & 3)] = xn[e];
    }
    if (tid < k) a.xnorm64[n + tid] = nn[tid];
    for (int e = tid; e < k * d; e += 256) a.x64[n * d + e] = a.xnew[e];
    if (tid < k) a.y64[n + tid] = a.ynew[tid];
    if (tid == 0) {
      *a.info = INT_MAX;
      if (a.f16_scal != nullptr) a.f16_scal[3] = a.f16_scal[1];  // the scale the resident pieces were packed with
    }
  }
  // 64 rows per workgroup, staged in LDS (odd stride); thread = (row, j mod 4): the k kernel values of a row are dealt to
  // four threads.  (First version: one thread per row reading its scaled inputs from global memory k times: 13 - 22 us.)
  __shared__ double xr[64 * 49];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int st = dp + 1;
  for (int e = tid; e < 64 * dp; e += 256) {
    const int rr = e / dp, kk = e - rr * dp;
    const int64_t i = r0 + rr;
    xr[rr * st + kk] = (i < n) ? a.xs64[i * dp + kk] : (i < n + k ? xn[(i - n) * dp + kk] : 0.0);
  }
  __syncthreads();
  const int rr = tid & 63, jg = tid >> 6;
  const int64_t i = r0 + rr;
  if (i >= n + k) return;
  const double ni = (i < n) ? a.xnorm64[i] : nn[i - n];
  double* out = a.Kc + i * KP;
  for (int j = jg; j < KP; j += 4) {
    double kv = 0.0;
    if (j < k) {
      double s = 0.0;
      for (int kk = 0; kk < dp; ++kk) s = fma(xr[rr * st + kk], xn[j * dp + kk], s);
      kv = kern_from_r2_lean(a.kernel, fma(-2.0, s, ni + nn[j]), a.variance);
      if (i - n == j) kv += a.noise;
    }
    out[j] = kv;
  }
}

// ---- 3. B from its chunks; partial Gram of [B | a1] per 64-row block -----------------------------------------------------
// sums the chunks of row block `tb` in order (deterministic), writes B, forms the block's partial [B | a1]^T [B | a1].
// tile: 64 (KP + 2) doubles of LDS.
template <typename TF, int KP>
__device__ __forceinline__ void append_gram_part(const double* __restrict__ bpart, int nq, int64_t npad, int tb,
                                                 double* __restrict__ Bm, const TF* __restrict__ white, int64_t n,
                                                 double* __restrict__ part, double* tile) {
  constexpr int W = KP + 1, LD = KP + 2;
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)tb * 64;
  for (int e = tid; e < 64 * KP; e += 256) {
    const int rr = e / KP, j = e - rr * KP;
    const int64_t r = r0 + rr;
    double v = 0.0;
    if (r < n) {
      for (int q = 0; q < nq; ++q) v += bpart[((int64_t)q * npad + r) * KP + j];
      Bm[r * KP + j] = v;
    }
    tile[rr * LD + j] = v;
  }
  if (tid < 64) tile[tid * LD + KP] = (r0 + tid < n) ? (double)white[r0 + tid] : 0.0;
  __syncthreads();
  double* out = part + (int64_t)tb * W * W;
  for (int e = tid; e < W * W; e += 256) {
    const int i = e / W, j = e - i * W;
    if (i > j) continue;  // (upper triangle incl. the a1 column: what append_chol reads)
    double s = 0.0;
#pragma unroll 8
    for (int rr = 0; rr < 64; ++rr) s = fma(tile[rr * LD + i], tile[rr * LD + j], s);
    out[e] = s;
  }
}

// ---- 4. the k x k corner: Schur complement, Cholesky, inverse, a2, NLML (ONE workgroup) --------------------------------------
// global sm (doubles): [0, 4096) L22 (ld 64) | [4096, 8192) L22^-1 (ld 64) | [8192, 8256) a2
// LDS (doubles, carved from `lds`): S [KP x KP] | X, lower triangle packed by rows: (i, j <= i) at i (i + 1) / 2 + j | v | a2 | psum
template <int KP>
constexpr int append_chol_lds_doubles() { return KP * KP + KP * (KP + 1) / 2 + 2 * KP + 256 + 2; }
template <int KP>
__device__ __forceinline__ void append_chol(const AppendArgs& a, int nchunk, double* lds) {
  constexpr int W = KP + 1;
  constexpr int LD = KP;
  double* S = lds;
  double* X = S + KP * KP;
  double* v = X + KP * (KP + 1) / 2;
  double* a2 = v + KP;
  double* psum = a2 + KP;
  int* bad = reinterpret_cast<int*>(psum + 256);
  const int tid = threadIdx.x, k = a.k;
  const int64_t n = a.n;
  if (tid == 0) *bad = INT_MAX;
  // S = K22 + noise I - sum over the chunks of B^T B; v = B^T a1.  PARTS threads share an entry (chunks p, p + PARTS, ..),
  // their sums are added in order: deterministic
  constexpr int EP = 1 << (32 - __builtin_clz((unsigned)(W * W - 1)));  // entries, rounded up to a power of two
  constexpr int PARTS = EP >= 256 ? 1 : 256 / EP;
  for (int e0 = 0; e0 < W * W; e0 += 256 / PARTS) {
    const int e = e0 + tid % (256 / PARTS), p = tid / (256 / PARTS);
    const int i = e / W, j = e - i * W;
    const bool live = e < W * W && i <= j && i < k;
    double g = 0.0;
    if (live) {
      int c = p;
      for (; c + 3 * PARTS < nchunk; c += 4 * PARTS) {
        const double g0 = a.part[(int64_t)c * W * W + e], g1 = a.part[(int64_t)(c + PARTS) * W * W + e];
        const double g2 = a.part[(int64_t)(c + 2 * PARTS) * W * W + e], g3 = a.part[(int64_t)(c + 3 * PARTS) * W * W + e];
        g += (g0 + g1) + (g2 + g3);
      }
      for (; c < nchunk; c += PARTS) g += a.part[(int64_t)c * W * W + e];
    }
    if (PARTS > 1) {
      __syncthreads();
      psum[tid] = g;
      __syncthreads();
      g = 0.0;
      if (p == 0)
        for (int pp = 0; pp < PARTS; ++pp) g += psum[pp * (256 / PARTS) + tid];
    }
    if (!live || p != 0) continue;
    if (j == KP) {
      v[i] = g;
    } else if (j < k) {
      const double s = a.Kc[(n + j) * KP + i] - g;  // (row n + j of the cross block holds K22[j][:])
      S[j * LD + i] = s;
      S[i * LD + j] = s;
    }
  }
  __syncthreads();
  // right-looking Cholesky in LDS, lower triangle
  for (int p = 0; p < k; ++p) {
    const double dpp = S[p * LD + p];
    if (!(dpp > 0.0) || !(dpp < 1.0e300)) {
      if (tid == 0) *bad = p;
      break;  // (block-uniform: every thread reads the same LDS word)
    }
    const double lpp = sqrt(dpp);
    __syncthreads();
    for (int i = p + tid; i < k; i += 256) S[i * LD + p] = (i == p) ? lpp : S[i * LD + p] / lpp;
    __syncthreads();
    const int m = k - p - 1;
    for (int e = tid; e < m * m; e += 256) {
      const int i = p + 1 + e / m, j = p + 1 + e % m;
      if (j <= i) S[i * LD + j] = fma(-S[i * LD + p], S[j * LD + p], S[i * LD + j]);
    }
    __syncthreads();
  }
  __syncthreads();
  if (*bad != INT_MAX) {
    if (tid == 0) {
      *a.info = (int)(n + *bad);
      a.host_out[2] = (double)(n + *bad);
      a.host_out[1] = 2.0;
    }
    return;
  }
  // X = L22^-1 by forward substitution, one column per thread
  if (tid < k) {
    const int j = tid;
    for (int i = 0; i < k; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      if (i < j) continue;
      for (int q = j; q < i; ++q) s = fma(-S[i * LD + q], X[q * (q + 1) / 2 + j], s);
      X[i * (i + 1) / 2 + j] = s / S[i * LD + i];
    }
  }
  __syncthreads();
  if (tid < k) {
    double s = 0.0;
    for (int j = 0; j <= tid; ++j) s = fma(X[tid * (tid + 1) / 2 + j], a.y64[n + j] - a.mean_c - v[j], s);
    a2[tid] = s;
  }
  __syncthreads();
  for (int e = tid; e < k * k; e += 256) {
    const int i = e / k, j = e - i * k;
    a.sm[i * 64 + j] = (j <= i) ? S[i * LD + j] : 0.0;
    a.sm[4096 + i * 64 + j] = (j <= i) ? X[i * (i + 1) / 2 + j] : 0.0;
  }
  if (tid < k) {
    a.sm[8192 + tid] = a2[tid];
    a.diag64[n + tid] = S[tid * LD + tid];
  }
  if (tid == 0) {
    double quad = 0.0, ld = 0.0;
    for (int j = 0; j < k; ++j) {
      quad = fma(a2[j], a2[j], quad);
      ld += log(S[j * LD + j]);
    }
    const double f = a.nlml[0] + 0.5 * quad + ld + 0.5 * (double)k * 1.83787706640934548356;  // log(2 pi)
    a.nlml[0] = f;
    a.host_out[0] = f;
    a.host_out[1] = 1.0;
  }
}

// ---- 5b. the new rows: R = -L22^-1 W per column, alpha, diag(K_y^-1), the k x k corner ------------------------------------
template <typename TP>
__device__ __forceinline__ void store_alpha_p(void* alpha_p, int64_t i, double v) {
  static_cast<TP*>(alpha_p)[i] = (TP)v;
}
// L22^-1 (packed) and a2 from the corner's global record into LDS
__device__ __forceinline__ void append_load_corner(const AppendArgs& a, double* Xs, double* a2s) {
  const int tid = threadIdx.x, k = a.k;
  for (int e = tid; e < k * k; e += 256) {
    const int i = e / k, j = e - i * k;
    if (j <= i) Xs[i * (i + 1) / 2 + j] = a.sm[4096 + i * 64 + j];
  }
  if (tid < k) a2s[tid] = a.sm[8192 + tid];
  __syncthreads();
}
// the k x k corner: rows n .. n + k - 1, columns n .. (end of the 64-column block that holds n + k - 1); one workgroup
template <typename TF, typename TP>
__device__ __forceinline__ void append_finish_corner(const AppendArgs& a, const double* Xs, const double* a2s, TF* __restrict__ linv,
                                                     TF* __restrict__ Lf, TF* __restrict__ white, TF* __restrict__ alpha_f) {
  const int tid = threadIdx.x, lane = tid & 63, k = a.k;
  const int64_t n = a.n, npad = a.npad;
  const int64_t cend = min(npad, (n + k + 63) / 64 * 64);
  const int wcols = (int)(cend - n);
  float m = 0.0f;
  for (int e = tid; e < k * wcols; e += 256) {
    const int t = e / wcols, j = e - t * wcols;
    double x = 0.0, l = 0.0;
    if (j <= t) {
      x = Xs[t * (t + 1) / 2 + j];
      l = a.sm[t * 64 + j];
    }
    linv[(n + t) * npad + n + j] = (TF)x;
    Lf[(n + t) * npad + n + j] = (TF)l;
    m = fmaxf(m, fabsf((float)x));
  }
  if (tid < k) {
    double al = 0.0, sq = 0.0;
    for (int j = tid; j < k; ++j) {
      const double x = Xs[j * (j + 1) / 2 + tid];
      al = fma(x, a2s[j], al);
      sq = fma(x, x, sq);
    }
    const TF alf = (TF)al;
    white[n + tid] = (TF)a2s[tid];
    alpha_f[n + tid] = alf;
    store_alpha_p<TP>(a.alpha_p, n + tid, (double)alf);
    a.kinv_diag[n + tid] = sq;
  }
  if (tid == 0) a.hyper[0] = (double)(n + k);
  if (a.f16_scal != nullptr) {
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned*>(a.f16_scal), __builtin_bit_cast(unsigned, m));
  }
}
// one thread per column c: R = -L22^-1 W at that column, alpha, diag(K_y^-1); the wave's largest |new entry| for the fp16 scale
template <typename TF, typename TP, int KP>
__device__ __forceinline__ void append_finish_columns(const AppendArgs& a, const double* __restrict__ wpart, int ct, int64_t c,
                                                      const double* Xs, const double* a2s, TF* __restrict__ linv, TF* __restrict__ Lf,
                                                      TF* __restrict__ alpha_f) {
  const int k = a.k;
  const int64_t n = a.n, npad = a.npad;
  float m = 0.0f;
  if (c < n) {
    // W[:, c]: the chunks of the column pass that hold tiles of this column block, in order
    const int ntile = (int)((n + 63) / 64);
    const int q0 = (int)(c / 64) / ct, q1 = (ntile - 1) / ct;
    double w[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) w[j] = 0.0;
    for (int q = q0; q <= q1; ++q) {
      const double* src = wpart + ((int64_t)q * npad + c) * KP;
#pragma unroll
      for (int j = 0; j < KP; ++j) w[j] += src[j];
    }
    // R = -L22^-1 W: the new rows of L^-1 at this column; L's new rows are B^T
    double al = 0.0, sq = 0.0;
    for (int t = 0; t < k; ++t) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < KP; ++j)
        if (j <= t) s = fma(Xs[t * (t + 1) / 2 + j], w[j], s);
      const TF rv = (TF)(-s);
      linv[(n + t) * npad + c] = rv;
      Lf[(n + t) * npad + c] = (TF)a.Bm[c * KP + t];
      al = fma((double)rv, a2s[t], al);
      sq = fma((double)rv, (double)rv, sq);
      m = fmaxf(m, fabsf((float)rv));
    }
    const TF alf = (TF)((double)alpha_f[c] + al);
    alpha_f[c] = alf;
    store_alpha_p<TP>(a.alpha_p, c, (double)alf);
    a.kinv_diag[c] += sq;
  }
  if (a.f16_scal != nullptr) {
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned*>(a.f16_scal), __builtin_bit_cast(unsigned, m));
  }
}

// ---- 2. / 5a. the two passes over L^-1, tile by tile ----------------------------------------------------------------------------
// One workgroup = one 64-row block of L^-1 (row pass: B = X11 K12) or one 64-column block (column pass: W = X11^T B) times
// a chunk of `ct` of its 64x64 tiles; grid (tiles per side, nq <= 8).  A tile
// arrives with 16-byte loads, kPassDepth tiles ahead of its use, is masked to the lower triangle / the first
// n rows and staged in LDS beside the 64 operand rows (K12 / B) of its contraction index; the product runs on the f64
// matrix instruction (16x16x4: wave g owns 16 rows / columns of the block, the k <= 64 right-hand sides are its 16-wide
// column blocks), accumulators live across the chunk's tiles.  Partial sums per chunk go to part[q][index][KP]; the
// consumer adds the chunks in order (deterministic).
// (First versions, profiles/r05_append_experiments.txt: one wave per row re-reading K12 from the L2 for every row -- 8.6 TB
// of L2 traffic at N = 16 384 -- and one workgroup per column strip: 0.71 + 1.98 ms at C5, 14 + 183 us at C3; then vector
// FMAs with scalar operand loads in the inner loop, latency-chained: 0.61 + 0.39 ms, 34 + 36 us.)
template <int KP>
struct PassShape {
  static constexpr int NB = KP <= 16 ? 1 : KP / 16;  // 16-wide blocks of right-hand sides (KP = 8: half a block is padding)
  static constexpr int UW = 16 * NB;
};
template <typename TF, int KP>
constexpr int append_pass_lds_bytes() { return 64 * 65 * (int)sizeof(TF) + 64 * PassShape<KP>::UW * 8; }
#ifndef GPSO_PASS_DEPTH
#define GPSO_PASS_DEPTH 2
#endif
constexpr int kPassDepth = GPSO_PASS_DEPTH;  // tiles in flight ahead of the one in use

template <typename TF, int KP, bool COLS>
__global__ __launch_bounds__(256) void append_pass_kernel(const TF* __restrict__ linv, int64_t n, int64_t npad,
                                                          const double* __restrict__ U, double* __restrict__ part, int ct) {
  constexpr int EV = 16 / (int)sizeof(TF);  // elements per 16-byte load
  constexpr int VPR = 64 / EV;              // loads per tile row
  constexpr int RPP = 256 / VPR;            // tile rows per pass of the workgroup
  constexpr int NP = 64 / RPP;              // passes
  constexpr int NB = PassShape<KP>::NB, UW = PassShape<KP>::UW;
  constexpr int NU = 64 * KP / 256;         // operand doubles per thread and tile
  constexpr int LD = 65;
  typedef TF vecT __attribute__((ext_vector_type(EV)));
  extern __shared__ __align__(16) unsigned char pass_lds[];
  TF* T = reinterpret_cast<TF*>(pass_lds);                                      // the tile, masked, [64][65]
  double* Us = reinterpret_cast<double*>(pass_lds + 64 * LD * sizeof(TF));      // the operand rows, [64][UW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tb = blockIdx.x, q = blockIdx.y;
  const int ntile = (int)((n + 63) / 64);
  int lo, hi;  // range of the other tile index
  if (!COLS) {
    lo = q * ct;
    hi = min((q + 1) * ct, tb + 1);
  } else {
    lo = max(tb, q * ct);
    hi = min((q + 1) * ct, ntile);
  }
  if (lo >= hi) return;  // (the consumer knows which chunks exist)
  if (KP < UW)
    for (int e = tid; e < 64 * (UW - KP); e += 256) Us[(e / (UW - KP)) * UW + KP + e % (UW - KP)] = 0.0;  // padding columns
  const int vrow = tid / VPR, vcol = (tid % VPR) * EV;
  vecT v[kPassDepth][NP];
  double uu[kPassDepth][NU];
  auto fetch = [&](int o, vecT* vv, double* uv) {
    const int64_t R0 = (COLS ? o : tb) * 64, C0 = (COLS ? tb : o) * 64;
#pragma unroll
    for (int p = 0; p < NP; ++p) vv[p] = *reinterpret_cast<const vecT*>(linv + (R0 + p * RPP + vrow) * npad + C0 + vcol);
    // the operand rows of the contraction index: K12 rows of the tile's columns (row pass) / B rows of its rows (column pass)
    const int64_t X0 = COLS ? R0 : C0;
#pragma unroll
    for (int w = 0; w < NU; ++w) {
      const int idx = w * 256 + tid;  // (x, j) = (idx / KP, idx % KP)
      uv[w] = (X0 + idx / KP < n) ? U[X0 * KP + idx] : 0.0;
    }
  };
  f64x4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = f64x4{0, 0, 0, 0};
  auto use = [&](int o, const vecT* vv, const double* uv) {
    const int64_t R0 = (COLS ? o : tb) * 64, C0 = (COLS ? tb : o) * 64;
    __syncthreads();  // (the previous tile has been read)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int64_t gr = R0 + p * RPP + vrow;
#pragma unroll
      for (int x = 0; x < EV; ++x) {
        const int64_t gc = C0 + vcol + x;
        T[(p * RPP + vrow) * LD + vcol + x] = (gr < n && gc <= gr) ? vv[p][x] : (TF)0;
      }
    }
#pragma unroll
    for (int w = 0; w < NU; ++w) {
      const int idx = w * 256 + tid;
      Us[(idx / KP) * UW + idx % KP] = uv[w];
    }
  };
  auto multiply = [&]() {
    __syncthreads();
    // 16 k-steps of the f64 matrix instruction: D[i][j] += A[i][kk] B[kk][j], A = the tile (row pass: i = row, kk = column;
    // column pass: i = column, kk = row) for this wave's 16 rows / columns, B = the operand rows
    const int i = lane & 15, kk = lane >> 4;
#pragma unroll 4
    for (int s4 = 0; s4 < 16; ++s4) {
      const double av = COLS ? (double)T[(4 * s4 + kk) * LD + 16 * g + i] : (double)T[(16 * g + i) * LD + 4 * s4 + kk];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Us[(4 * s4 + kk) * UW + 16 * b + i], acc[b], 0, 0, 0);
    }
  };
  // software pipeline, kPassDepth tiles in flight: slot s holds tile lo + (multiple of kPassDepth) + s (static register slots)
#pragma unroll
  for (int s_ = 0; s_ < kPassDepth; ++s_)
    if (lo + s_ < hi) fetch(lo + s_, v[s_], uu[s_]);
  for (int o = lo; o < hi; o += kPassDepth) {
#pragma unroll
    for (int s_ = 0; s_ < kPassDepth; ++s_) {
      if (o + s_ < hi) {
        use(o + s_, v[s_], uu[s_]);
        if (o + s_ + kPassDepth < hi) fetch(o + s_ + kPassDepth, v[s_], uu[s_]);
        multiply();
      }
    }
  }
  // accumulator register r of lane l: index 16 g + (l >> 4) + 4 r of the block, right-hand side 16 b + (l & 15)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = 16 * b + (lane & 15);
    if (j < KP) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        part[((int64_t)q * npad + (int64_t)tb * 64 + 16 * g + (lane >> 4) + 4 * r) * KP + j] = acc[b][r];
    }
  }
}

// ---- the consumers as kernels of their own ------------------------------------------------------------------------------------
// (Round 6 measured the alternative the round-5 verdict asked for -- each pass carrying its consumer in its epilogue behind
// last-arriver tickets, agent-scope fences around one atomic counter per block: 8 launches -> 5, same bits -- and it is SLOWER
// on this chip: the fused row pass took 31 us at C3 against 7.5 + 7.9 + 11.2 us for pass + Gram + corner as three launches, and
// at C5 the streaming itself fell from 147 to 244 us per pass with 1 150 workgroups each writing back / invalidating its XCD's L2
// at its end.  A launch boundary is the cheaper grid-wide synchronisation here: profiles/r06_append_experiments.txt.)
template <typename TF, int KP>
__global__ __launch_bounds__(256) void append_gram_part_kernel(const double* __restrict__ bpart, int ct, int64_t npad,
                                                               double* __restrict__ Bm, const TF* __restrict__ white,
                                                               int64_t n, double* __restrict__ part) {
  __shared__ double tile[64 * (KP + 2)];
  append_gram_part<TF, KP>(bpart, (int)blockIdx.x / ct + 1, npad, (int)blockIdx.x, Bm, white, n, part, tile);
}
template <int KP>
__global__ __launch_bounds__(256) void append_chol_kernel(AppendArgs a, int nchunk) {
  __shared__ double lds[append_chol_lds_doubles<KP>()];
  append_chol<KP>(a, nchunk, lds);
}
// grid: column blocks of 256 (four 64-column blocks, one wave each) + one workgroup for the k x k corner
template <typename TF, typename TP, int KP>
__global__ __launch_bounds__(256) void append_finish_kernel(AppendArgs a, const double* __restrict__ wpart, int ct,
                                                            TF* __restrict__ linv, TF* __restrict__ Lf,
                                                            TF* __restrict__ white, TF* __restrict__ alpha_f) {
  if (*a.info != INT_MAX) return;  // S was not positive definite: the resident posterior stays as it is
  __shared__ double Xs[KP * (KP + 1) / 2];  // L22^-1, lower triangle packed by rows
  __shared__ double a2s[KP];
  append_load_corner(a, Xs, a2s);
  const int nblk = (int)((a.n + 255) / 256);
  if ((int)blockIdx.x == nblk) append_finish_corner<TF, TP>(a, Xs, a2s, linv, Lf, white, alpha_f);
  else append_finish_columns<TF, TP, KP>(a, wpart, ct, (int64_t)blockIdx.x * 256 + threadIdx.x, Xs, a2s, linv, Lf, alpha_f);
}

// ---- launcher ---------------------------------------------------------------------------------------------------------
int append_kp(int k) { return k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64; }
constexpr int kAppendChunks = 8;  // chunks of tiles per block row / column of a pass, at most
static size_t append_off_bm(int64_t npad, int kp) { return (size_t)(npad + kAppendMax) * kp; }
static size_t append_off_pass(int64_t npad, int kp) { return append_off_bm(npad, kp) + (size_t)npad * kp; }
static size_t append_off_part(int64_t npad, int kp) { return append_off_pass(npad, kp) + (size_t)kAppendChunks * npad * kp; }
static size_t append_off_sm(int64_t npad, int kp) { return append_off_part(npad, kp) + (size_t)((npad + 63) / 64) * (kp + 1) * (kp + 1); }
static size_t append_off_info(int64_t npad, int kp) { return append_off_sm(npad, kp) + 8192 + 64; }
size_t append_scratch_doubles(int64_t npad, int kp) { return append_off_info(npad, kp) + 8; }

template <typename TF, typename TP, int KP>
static void launch_append_kp(hipStream_t st, AppendArgs a, double* pass, TF* linv, TF* Lf, TF* white, TF* alpha_f) {
  const int64_t n = a.n;
  const int ntile = (int)((n + 63) / 64);
  const int ct = (ntile + kAppendChunks - 1) / kAppendChunks, nq = (ntile + ct - 1) / ct;
  constexpr int lds = append_pass_lds_bytes<TF, KP>();
  if (ensure_dyn_lds(reinterpret_cast<const void*>(&append_pass_kernel<TF, KP, false>), lds) ||
      ensure_dyn_lds(reinterpret_cast<const void*>(&append_pass_kernel<TF, KP, true>), lds))
    return;  // (recorded with note_launch_error)
  hipLaunchKernelGGL(append_cross_kernel, dim3((unsigned)((n + a.k + 63) / 64)), dim3(256), 0, st, a);
  hipLaunchKernelGGL((append_pass_kernel<TF, KP, false>), dim3((unsigned)ntile, (unsigned)nq), dim3(256), lds, st, linv, n, a.npad,
                     a.Kc, pass, ct);
  hipLaunchKernelGGL((append_gram_part_kernel<TF, KP>), dim3((unsigned)ntile), dim3(256), 0, st, pass, ct, a.npad, a.Bm, white, n, a.part);
  hipLaunchKernelGGL((append_chol_kernel<KP>), dim3(1), dim3(256), 0, st, a, ntile);
  hipLaunchKernelGGL((append_pass_kernel<TF, KP, true>), dim3((unsigned)ntile, (unsigned)nq), dim3(256), lds, st, linv, n, a.npad,
                     a.Bm, pass, ct);
  hipLaunchKernelGGL((append_finish_kernel<TF, TP, KP>), dim3((unsigned)((n + 255) / 256) + 1), dim3(256), 0, st, a, pass, ct, linv,
                     Lf, white, alpha_f);
}

template <typename TF, typename TP>
void launch_append(hipStream_t st, AppendArgs a, void* scratch, TF* linv, TF* Lf, TF* white, TF* alpha_f) {
  const int kp = append_kp(a.k);
  a.kp = kp;
  double* q = static_cast<double*>(scratch);
  a.Kc = q;
  a.Bm = q + append_off_bm(a.npad, kp);
  double* pass = q + append_off_pass(a.npad, kp);
  a.part = q + append_off_part(a.npad, kp);
  a.sm = q + append_off_sm(a.npad, kp);
  a.info = reinterpret_cast<int*>(q + append_off_info(a.npad, kp));
  switch (kp) {
    case 8: launch_append_kp<TF, TP, 8>(st, a, pass, linv, Lf, white, alpha_f); break;
    case 16: launch_append_kp<TF, TP, 16>(st, a, pass, linv, Lf, white, alpha_f); break;
    case 32: launch_append_kp<TF, TP, 32>(st, a, pass, linv, Lf, white, alpha_f); break;
    default: launch_append_kp<TF, TP, 64>(st, a, pass, linv, Lf, white, alpha_f); break;
  }
}
template void launch_append<float, float>(hipStream_t, AppendArgs, void*, float*, float*, float*, float*);
template void launch_append<double, float>(hipStream_t, AppendArgs, void*, double*, double*, double*, double*);
template void launch_append<double, double>(hipStream_t, AppendArgs, void*, double*, double*, double*, double*);

}  // namespace gpso
