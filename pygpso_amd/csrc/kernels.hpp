// Host-callable launchers of the gfx950 kernels (definitions in predict.hip / fit.hip / grow.hip).
#pragma once
#include "common.hpp"

namespace gpso {

// tile shape of the dominant predict kernel: BM rows of L^-1 x (4 waves * CT * 16) leaves per workgroup
// (chosen per launch: f32 uses 256 x 128 when N_pad is a multiple of 256, else 128 x 256; f64 64 x 128;
//  always 32 (f32) / 8 (f64) accumulator tiles per wave)
constexpr int kLeafPad = 256;  // leaf batches are padded to a multiple of this
// rows of L^-1 per workgroup for a given padded N and D / 4, and the resulting number of row blocks
template <typename T>
int leaf_tiles_bm(int64_t npad, int dp4);
template <typename T>
inline int leaf_tiles_nbi(int64_t npad, int dp4) { return (int)(npad / leaf_tiles_bm<T>(npad, dp4)); }

// ---- predict.hip ----------------------------------------------------------------------------------
template <typename T, typename TIN>
void launch_prep_leaves(hipStream_t st, const TIN* xs, int64_t m, int64_t mpad, int d, int dp,
                        const double* ls, T* out, T* norm);
template <typename T>
void launch_leaf_tiles(hipStream_t st, const T* linv_p, const T* xs_p, const T* xnorm,
                       const T* alpha, const T* leaves_s, const T* lnorm, T* part_var, T* part_mean,
                       int64_t npad, int dp4, int64_t mpad, const KernParams& kp);
// split-bf16 apply (float contexts): nsplit = 2 -> bf16x3, 3 -> bf16x6; needs npad % 256 == 0.
// linv_b = nsplit * npad * npad bf16, produced by launch_pack_linv_bf16 from the f32 L^-1
void launch_pack_linv_bf16(hipStream_t st, int nsplit, const float* linv, int64_t n, int64_t npad,
                           void* linv_b);
void launch_leaf_tiles_bf16(hipStream_t st, int nsplit, const void* linv_b, const float* xs_p,
                            const float* xnorm, const float* alpha, const float* leaves_s,
                            const float* lnorm, float* part_var, float* part_mean, int64_t npad,
                            int dp4, int64_t mpad, const KernParams& kp);
template <typename T>
void launch_leaf_finalize(hipStream_t st, const T* part_var, const T* part_mean, int nbi,
                          int64_t mpad, int64_t m, const KernParams& kp, double varsigma,
                          double* mean, double* var, double* ucb);
void launch_seg_argmax(hipStream_t st, const double* mean, const double* var, const double* ucb,
                       const int64_t* seg_off_dev, int nseg, int nblk, void* partial_dev,
                       double* out_vals_dev /* [nseg*4]: mean, var, ucb, bit-cast int64 index */);
constexpr int kArgmaxBlocks = 64;      // stage-1 blocks per segment
constexpr size_t kArgmaxPartialBytes = 16;  // sizeof(Best)

// ---- fit.hip --------------------------------------------------------------------------------------
// scaled inputs: xs[npad*dp] = X/ls (zero padded), xnorm[npad], xs_p = MFMA-fragment packing
template <typename T>
void launch_scale_x(hipStream_t st, const double* x64, int64_t n, int64_t npad, int d, int dp,
                    const double* ls, T* xs, T* xnorm, T* xs_p);
// K = k(X, X) + noise * I on rows < n; identity on the padding
template <typename T>
void launch_gram(hipStream_t st, const T* xs, const T* xnorm, int64_t n, int64_t npad, int dp,
                 const KernParams& kp, T* K);
// blocked right-looking Cholesky, K (destroyed) -> Lf (lower); also writes the inverted 64x64 diagonal
// blocks into linv, the unrounded diagonal of L to diag64[npad], and the first failing pivot (or
// INT_MAX) to info
// Returns bit flags.  Bit 0: linv already holds the COMPLETE inverse (small sizes: it is built beside
// the factorisation, using work as scratch) and launch_trtri must be skipped.  Bit 1: kinv (nullable:
// only wanted with the gradient) already holds K^-1 = L^-T L^-1 (lower tiles).
template <typename T>
int launch_potrf(hipStream_t st, T* K, T* Lf, T* linv, T* work, T* kinv, int64_t n, int64_t npad,
                 double* diag64, int* info, int64_t single_level_max /* < 0: default */);
// L^-1 by level-doubling from level first_level (64 or the factorisation's outer panel width): needs the
// inverses of the first_level-wide diagonal blocks already in linv; work = npad x npad scratch
constexpr int kFitOuterPanel = 512;
template <typename T>
void launch_trtri(hipStream_t st, const T* L, T* linv, T* work, int64_t npad, int64_t first_level);
// zero rows/cols >= n and re-tile L^-1 into the MFMA fragment-major layout the predict kernel reads
template <typename T>
void launch_pack_linv(hipStream_t st, const T* linv, int64_t n, int64_t npad, T* linv_p);
// a = L^-1 (y - c), alpha = L^-T a, nlml = 1/2 a.a + sum log diag64 + n/2 log 2pi  (double accumulators)
template <typename T>
void launch_solve_alpha(hipStream_t st, const T* linv, const double* y64, int64_t n, int64_t npad,
                        double mean_c, const double* diag64, T* white, T* alpha,
                        double* alpha_part /* [ceil(npad/64) * npad] scratch */, double* nlml_out);
// Kinv = L^-T L^-1 (lower tiles; skipped when kinv_ready), then the gradient reductions of SURVEY.md A.3;
// grad_out[n_ls + 3] = d nlml / d (ls..., variance, noise, c)
template <typename T>
void launch_gradient(hipStream_t st, const T* linv, const T* alpha, const T* xs, const T* xnorm,
                     int64_t n, int64_t npad, int d, int dp, int n_ls, const double* ls,
                     const KernParams& kp, T* kinv, bool kinv_ready, double* partial, double* grad_out);
constexpr int kGradMaxLs = 64;

// float64 host -> T device conversions and getters
template <typename T>
void launch_convert_in(hipStream_t st, const double* src, T* dst, int64_t rows, int64_t cols,
                       int64_t ld_dst);
template <typename T>
void launch_convert_out(hipStream_t st, const T* src, int64_t ld_src, double* dst, int64_t rows,
                        int64_t cols, int lower_only);
// install L (row-major lower, n x n float64 on device) into the padded T buffer + invert diagonal blocks
template <typename T>
void launch_install_chol(hipStream_t st, const double* L64, int64_t n, int64_t npad, T* K, T* linv);

// ---- grow.hip -------------------------------------------------------------------------------------
// centres of the ternary subtree (levels 0..depth-1) under each box; out[nseg*rows*d] float64
void launch_grow(hipStream_t st, const double* bounds_dev, int nseg, int d, int depth,
                 double* out_dev);

}  // namespace gpso
