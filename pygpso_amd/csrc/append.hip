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
// cancellation the float fit also pays for.  Kernels (one stream, in order):
//
//   append_cross_kernel      K12 (n x k) and K22 + noise I (k x k) from the scaled inputs; the new rows' scaled inputs,
//                            norms and MFMA fragments land where the fit leaves them (same formulas as scale_x_kernel /
//                            gram_kernel: GEMM-form r^2 in double)
//   append_rows_kernel       B[r][:] = sum_{c <= r} X11[r][c] K12[c][:]          one wave per row, coalesced row reads
//   append_gram_part_kernel  partial [B | a1]^T [B | a1] per 64-row chunk          (L21 L21^T and L21 a1 in one go)
//   append_chol_kernel       ONE workgroup: S, its Cholesky (first failing pivot -> info = n + p), L22^-1, a2, nlml
//   append_cols_kernel       W[:, c] = sum_{r >= c} B[r][:] X11[r][c] per 64-column strip (B blocks through LDS), then
//                            R = -L22^-1 W: the new rows of L^-1 and L, alpha1 += R^T a2, diag(K_y^-1) += sum R^2,
//                            max |new entries| for the fp16 split's scale; one extra workgroup writes the k x k corner
//
// A failed pivot leaves the resident posterior of the n points untouched (append_cols_kernel exits on info).
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
  for (int e = tid; e < k * dp; e += 256) {
    const int t = e / dp, kk = e - t * dp;
    xn[e] = (kk < d) ? a.x64[(n + t) * d + kk] / a.ls[kk] : 0.0;
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
    if (tid == 0) {
      *a.info = INT_MAX;
      if (a.f16_scal != nullptr) a.f16_scal[3] = a.f16_scal[1];  // the scale the resident pieces were packed with
    }
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  if (i >= n + k) return;
  double* out = a.Kc + i * KP;
  if (i < n) {
    const double* xi = a.xs64 + i * dp;
    const double ni = a.xnorm64[i];
    for (int j = 0; j < k; ++j) {
      double s = 0.0;
      for (int kk = 0; kk < dp; ++kk) s = fma(xi[kk], xn[j * dp + kk], s);
      out[j] = kern_from_r2_lean(a.kernel, fma(-2.0, s, ni + nn[j]), a.variance);
    }
  } else {
    const int t = (int)(i - n);
    for (int j = 0; j < k; ++j) {
      double s = 0.0;
      for (int kk = 0; kk < dp; ++kk) s = fma(xn[t * dp + kk], xn[j * dp + kk], s);
      double kv = kern_from_r2_lean(a.kernel, fma(-2.0, s, nn[t] + nn[j]), a.variance);
      if (t == j) kv += a.noise;
      out[j] = kv;
    }
  }
  for (int j = k; j < KP; ++j) out[j] = 0.0;
}

// ---- 2. row pass: B = X11 K12 ---------------------------------------------------------------------------------------
template <typename TF, int KP>
__global__ __launch_bounds__(256) void append_rows_kernel(const TF* __restrict__ linv, int64_t n, int64_t npad,
                                                          const double* __restrict__ Kc, double* __restrict__ Bm) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  double acc[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) acc[j] = 0.0;
  const TF* row = linv + r * npad;
  int64_t c = lane;
  if constexpr (KP <= 16) {
    // (two steps in flight: one row per wave, nothing else hides the latency)
    for (; c + 64 <= r; c += 128) {
      const double l0 = (double)row[c], l1 = (double)row[c + 64];
      double k0[KP], k1[KP];
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        k0[j] = Kc[c * KP + j];
        k1[j] = Kc[(c + 64) * KP + j];
      }
#pragma unroll
      for (int j = 0; j < KP; ++j) acc[j] = fma(l1, k1[j], fma(l0, k0[j], acc[j]));
    }
  }
  for (; c <= r; c += 64) {
    const double l = (double)row[c];
#pragma unroll
    for (int j = 0; j < KP; ++j) acc[j] = fma(l, Kc[c * KP + j], acc[j]);
  }
  double mine = 0.0;
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    const double s = wave_sum(acc[j]);
    if (lane == j) mine = s;
  }
  if (lane < KP) Bm[r * KP + lane] = mine;
}

// ---- 3. partial Gram of [B | a1] per 64-row chunk -------------------------------------------------------------------
template <typename TF, int KP>
__global__ __launch_bounds__(256) void append_gram_part_kernel(const double* __restrict__ Bm, const TF* __restrict__ white,
                                                               int64_t n, double* __restrict__ part) {
  constexpr int W = KP + 1, LD = KP + 2;
  __shared__ double tile[64 * LD];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int e = tid; e < 64 * W; e += 256) {
    const int rr = e / W, j = e - rr * W;
    const int64_t r = r0 + rr;
    double v = 0.0;
    if (r < n) v = (j < KP) ? Bm[r * KP + j] : (double)white[r];
    tile[rr * LD + j] = v;
  }
  __syncthreads();
  double* out = part + (int64_t)blockIdx.x * W * W;
  for (int e = tid; e < W * W; e += 256) {
    const int i = e / W, j = e - i * W;
    if (i > j) continue;  // (upper triangle incl. the a1 column: what append_chol_kernel reads)
    double s = 0.0;
#pragma unroll 8
    for (int rr = 0; rr < 64; ++rr) s = fma(tile[rr * LD + i], tile[rr * LD + j], s);
    out[e] = s;
  }
}

// ---- 4. the k x k corner: Schur complement, Cholesky, inverse, a2, NLML ---------------------------------------------
// sm (doubles): [0, 4096) L22 (ld 64) | [4096, 8192) L22^-1 (ld 64) | [8192, 8256) a2
template <int KP>
__global__ __launch_bounds__(256) void append_chol_kernel(AppendArgs a, int nchunk) {
  constexpr int W = KP + 1;
  constexpr int LD = 64;
  __shared__ double S[64 * LD];
  __shared__ double X[64 * 65 / 2];  // L22^-1, lower triangle packed by rows: (i, j <= i) at i (i + 1) / 2 + j
  __shared__ double v[64], a2[64];
  __shared__ int bad;
  const int tid = threadIdx.x, k = a.k;
  const int64_t n = a.n;
  if (tid == 0) bad = INT_MAX;
  // S = K22 + noise I - sum over the chunks of B^T B (fixed order: deterministic); v = B^T a1
  for (int e = tid; e < W * W; e += 256) {
    const int i = e / W, j = e - i * W;
    if (i > j || i >= k) continue;
    double g = 0.0;
    for (int c = 0; c < nchunk; ++c) g += a.part[(int64_t)c * W * W + e];
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
      if (tid == 0) bad = p;
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
  if (bad != INT_MAX) {
    if (tid == 0) {
      *a.info = (int)(n + bad);
      a.host_out[2] = (double)(n + bad);
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

// ---- 5. column pass and the new rows -----------------------------------------------------------------------------
template <typename TP>
__device__ __forceinline__ void store_alpha_p(void* alpha_p, int64_t i, double v) {
  static_cast<TP*>(alpha_p)[i] = (TP)v;
}

template <typename TF, typename TP, int KP>
__global__ __launch_bounds__(256) void append_cols_kernel(AppendArgs a, TF* __restrict__ linv, TF* __restrict__ Lf,
                                                          TF* __restrict__ white, TF* __restrict__ alpha_f) {
  if (*a.info != INT_MAX) return;  // S was not positive definite: the resident posterior stays as it is
  constexpr int LDB = KP;  // (rows of the block are read wave-uniformly: LDS broadcasts, no padding needed)
  constexpr int JW = KP < 16 ? KP : 16;
  constexpr int kBufDoubles = 64 * KP > 3 * JW * 64 ? 64 * KP : 3 * JW * 64;
  __shared__ double buf[kBufDoubles];   // a 64-row block of B; afterwards the cross-wave reduction, JW accumulators at a time
  __shared__ double Xs[64 * 65 / 2];    // L22^-1, lower triangle packed by rows
  __shared__ double a2s[64];
  double* Bs = buf;
  double (*red)[JW][64] = reinterpret_cast<double (*)[JW][64]>(buf);
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = a.k;
  const int64_t n = a.n, npad = a.npad;
  const int nstrip = (int)((n + 63) / 64);
  for (int e = tid; e < k * k; e += 256) {
    const int i = e / k, j = e - i * k;
    if (j <= i) Xs[i * (i + 1) / 2 + j] = a.sm[4096 + i * 64 + j];
  }
  if (tid < k) a2s[tid] = a.sm[8192 + tid];
  __syncthreads();

  if ((int)blockIdx.x == nstrip) {
    // the k x k corner: rows n .. n + k - 1, columns n .. (end of the 64-column block that holds n + k - 1)
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
    return;
  }

  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  double w[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) w[j] = 0.0;
  const int nrb = nstrip;
  for (int rb = (int)blockIdx.x; rb < nrb; ++rb) {
    const int64_t r0 = (int64_t)rb * 64;
    __syncthreads();
    for (int e = tid; e < 64 * KP; e += 256) {
      const int rr = e / KP, j = e - rr * KP;
      Bs[rr * LDB + j] = (r0 + rr < n) ? a.Bm[(r0 + rr) * KP + j] : 0.0;
    }
    // this wave's 16 rows of the block: the loads first, then the updates
    double l[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int64_t r = r0 + g + 4 * u;
      l[u] = (r < n && r >= c) ? (double)linv[r * npad + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double* brow = Bs + (g + 4 * u) * LDB;
#pragma unroll
      for (int j = 0; j < KP; ++j) w[j] = fma(l[u], brow[j], w[j]);
    }
  }
  // sum the four waves' partial columns (fixed order), 16 accumulators at a time; wave 0 keeps the totals
#pragma unroll
  for (int j0 = 0; j0 < KP; j0 += JW) {
    __syncthreads();
    if (g > 0) {
#pragma unroll
      for (int j = 0; j < JW; ++j) red[g - 1][j][lane] = w[j0 + j];
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int j = 0; j < JW; ++j) w[j0 + j] = (w[j0 + j] + red[0][j][lane]) + (red[1][j][lane] + red[2][j][lane]);
    }
  }
  if (g != 0 || c >= n) return;
  // R = -L22^-1 W: the new rows of L^-1 at this column; L's new rows are B^T
  double al = 0.0, sq = 0.0;
  float m = 0.0f;
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
  if (a.f16_scal != nullptr) {
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned*>(a.f16_scal), __builtin_bit_cast(unsigned, m));
  }
}

// ---- launcher ---------------------------------------------------------------------------------------------------------
int append_kp(int k) { return k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64; }
size_t append_scratch_doubles(int64_t npad, int kp) {
  const size_t chunks = (size_t)(npad + 63) / 64;
  return (size_t)(npad + kAppendMax) * kp + (size_t)npad * kp + chunks * (size_t)(kp + 1) * (kp + 1) + 8192 + 64 + 8;
}

template <typename TF, typename TP, int KP>
static void launch_append_kp(hipStream_t st, AppendArgs a, TF* linv, TF* Lf, TF* white, TF* alpha_f) {
  const int64_t n = a.n;
  const int nchunk = (int)((n + 63) / 64);
  hipLaunchKernelGGL(append_cross_kernel, dim3((unsigned)((n + a.k + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL((append_rows_kernel<TF, KP>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, linv, n, a.npad, a.Kc, a.Bm);
  hipLaunchKernelGGL((append_gram_part_kernel<TF, KP>), dim3((unsigned)nchunk), dim3(256), 0, st, a.Bm, white, n, a.part);
  hipLaunchKernelGGL((append_chol_kernel<KP>), dim3(1), dim3(256), 0, st, a, nchunk);
  hipLaunchKernelGGL((append_cols_kernel<TF, TP, KP>), dim3((unsigned)nchunk + 1), dim3(256), 0, st, a, linv, Lf, white, alpha_f);
}

template <typename TF, typename TP>
void launch_append(hipStream_t st, AppendArgs a, void* scratch, TF* linv, TF* Lf, TF* white, TF* alpha_f) {
  const int kp = append_kp(a.k);
  a.kp = kp;
  double* q = static_cast<double*>(scratch);
  a.Kc = q;
  q += (size_t)(a.npad + kAppendMax) * kp;
  a.Bm = q;
  q += (size_t)a.npad * kp;
  a.part = q;
  q += (size_t)((a.npad + 63) / 64) * (kp + 1) * (kp + 1);
  a.sm = q;
  q += 8192 + 64;
  a.info = reinterpret_cast<int*>(q);
  switch (kp) {
    case 8: launch_append_kp<TF, TP, 8>(st, a, linv, Lf, white, alpha_f); break;
    case 16: launch_append_kp<TF, TP, 16>(st, a, linv, Lf, white, alpha_f); break;
    case 32: launch_append_kp<TF, TP, 32>(st, a, linv, Lf, white, alpha_f); break;
    default: launch_append_kp<TF, TP, 64>(st, a, linv, Lf, white, alpha_f); break;
  }
}
const int* append_info_ptr(void* scratch, int64_t npad, int k) {
  const int kp = append_kp(k);
  const size_t off = (size_t)(npad + kAppendMax) * kp + (size_t)npad * kp + (size_t)((npad + 63) / 64) * (kp + 1) * (kp + 1) + 8192 + 64;
  return reinterpret_cast<const int*>(static_cast<double*>(scratch) + off);
}
template void launch_append<float, float>(hipStream_t, AppendArgs, void*, float*, float*, float*, float*);
template void launch_append<double, float>(hipStream_t, AppendArgs, void*, double*, double*, double*, double*);
template void launch_append<double, double>(hipStream_t, AppendArgs, void*, double*, double*, double*, double*);

}  // namespace gpso
