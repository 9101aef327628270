// Host-callable launchers of the gfx950 kernels (definitions in predict.hip / fit.hip / grow.hip).
//
// Type roles:  TF = fit type (Gram, Cholesky, L^-1, alpha: float | double)
//              T / TP = predict ("apply") type (packed L^-1, alpha, accumulators: float | double)
//              TG = generation type of the cross-Gram tile's x.x* contraction and r^2 (float | double;
//                   double by default also in float contexts, see predict.hip)
#pragma once
#include "common.hpp"

namespace gpso {

// ---- runtime helpers (api.hip) ------------------------------------------------------------------
// Opt a kernel into `bytes` of dynamic LDS (> 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize)
// on the CURRENT device; remembered per (function, device), thread-safe.  Returns 0 or a negative
// GPSO_E_* code (also recorded with note_launch_error).
int ensure_dyn_lds(const void* fn, int bytes);
// Remember the first launcher-side error of the call in flight (thread-local); the C-ABI entry
// points turn it into GPSO_E_HIP with this message.
void note_launch_error(const char* msg);

// tile shape of the dominant predict kernel: BM rows of L^-1 x (4 waves * CT * 16) leaves per workgroup
constexpr int kLeafPad = 256;  // leaf batches are padded to a multiple of this
// rows of L^-1 per workgroup for a given padded N and D / 4, and the resulting number of row blocks
template <typename T>
int leaf_tiles_bm(int64_t npad, int dp4);
template <typename T>
inline int leaf_tiles_nbi(int64_t npad, int dp4) { return (int)(npad / leaf_tiles_bm<T>(npad, dp4)); }

// ---- predict.hip ----------------------------------------------------------------------------------
// m_live (nullable, device): number of live leaves; rows at or beyond it are padding
template <typename TG, typename TIN>
void launch_prep_leaves(hipStream_t st, const TIN* xs, int64_t m, int64_t mpad, int d, int dp,
                        const double* ls, const int64_t* m_live, TG* out, TG* norm);
// partial sums per row block of L^-1: part_var / part_mean [nbi][mpad] double
template <typename T, typename TG>
int launch_leaf_tiles(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm,
                      const T* alpha, const TG* leaves_s, const TG* lnorm, double* part_var,
                      double* part_mean, int64_t npad, int dp4, int64_t mpad, const KernParams& kp,
                      const int64_t* m_live);
// split-bf16 apply (float contexts): nsplit = 2 -> bf16x3, 3 -> bf16x6; needs npad % 256 == 0.
// linv_b = nsplit * npad * npad bf16, produced by launch_pack_linv_bf16 from the fit-type L^-1
// rt0 > 0: only the 16-row tiles from rt0 on are (re)packed -- the rows a gpso_append wrote
template <typename TF>
void launch_pack_linv_bf16(hipStream_t st, int nsplit, const TF* linv, int64_t n, int64_t npad,
                           void* linv_b, int64_t rt0 = 0);
// dynamic LDS of the split-bf16 kernel: A pieces (2 buffers x nsplit x 16 KiB) + 3 X buffers + the 8 waves' leaf
// fragments, for a generation type of tg_bytes
// c16: the contraction runs on the fp16 pipe (float generation only): the X fragments of a k-step and the leaf fragments
// of a wave are fp16 piece pairs of 32-dimension chunks instead of float groups of four dimensions
bool leaf_step32_built();  // predict_split_f32.hip: was the library built with -DGPSO_STEP32=1 (leaf_split.hpp)
extern int g_leaf_last_splits;  // predict.hip: what the launcher of the split kernels chose last (gpso_last_count(ctx, 3))
extern int g_leaf_row_loop;  // predict.hip: GPSO_OPT_ROW_LOOP (a workgroup of the split predict kernels loops over row blocks)
inline int leaf_c16_chunks(int dp4) { return (dp4 + 8) / 8; }  // D_pad inputs + the norm slot, 32 slots per chunk
inline size_t leaf_bf16_lds_bytes(int nsplit, int dp4, int tg_bytes, bool c16 = false) {
  const size_t xfrag = c16 ? (size_t)leaf_c16_chunks(dp4) * 4096 : (size_t)2 * dp4 * 64 * tg_bytes;
  return (size_t)2 * nsplit * 16 * 64 * 16 + (size_t)3 * (xfrag + 64 * tg_bytes + 256) + (size_t)8 * xfrag;
}
// fp16 piece pairs of the float scaled inputs in the fragment order of the fp16 contraction (predict.hip); scal: 4 device
// floats ([0] max |x / l|, [1] := 2^sx, [2] := 2^-2sx); xs_h16: npad / 16 * leaf_c16_chunks * 2 KB
void launch_gen_inputs_f16(hipStream_t st, const float* xs32, const float* xnorm32, int64_t npad, int dp, float* scal,
                           void* xs_h16);
// f16_inv_scale_a != nullptr: the fp16 split (two pieces, three products; predict.hip) -- linv_b and the scale (device,
// 2 floats: max |L^-1|, 2^-sa) come from launch_pack_linv_f16
// FUSED: the fused step (round 4) or round 3's two-phase step -- same bits; one explicit instantiation per slice
// (TG, FUSED, KS) in predict_split_*.hip
// the caller's unscaled float leaves for the fp16-contraction kernels, whose prologue can scale them itself (round 5: the
// prep launch and the dependency gap behind it are 12 us of a 740 us step at C3); x == nullptr: leaves_s / lnorm are read
struct RawLeaves {
  const float* x = nullptr;
  const double* ls = nullptr;  // lengthscale per input dimension (device)
  int64_t m = 0;
  int d = 0;
  bool step32 = false;  // (rides along: GPSO_SPLIT_KERNEL_FUSED32 -- the fused step on the 32x32x16 instruction, where built)
};
template <typename TG, bool FUSED, int KS /* kernel families 0-1 | 2-3 */>
int launch_leaf_tiles_bf16_v(hipStream_t st, int nsplit, const void* linv_b, const TG* xs_p,
                             const TG* xnorm, const float* alpha, const TG* leaves_s,
                             const TG* lnorm, double* part_var, double* part_mean, int64_t npad,
                             int dp4, int64_t mpad, const KernParams& kp, const int64_t* m_live,
                             const float* f16_inv_scale_a, const void* xs_h16, const float* c16_scale, int64_t n_rows,
                             const RawLeaves& rawl);
// variant: GPSO_OPT_SPLIT_KERNEL (0 the fused step, 1 the two-phase step); xs_h16 and c16_scale both set: the contraction
// on the fp16 pipe (fp16 split, float generation); n_rows: N (0: unknown) -- the fused step stops at the k-steps that
// hold padding points only
template <typename TG>
inline int launch_leaf_tiles_bf16(hipStream_t st, int nsplit, const void* linv_b, const TG* xs_p,
                                  const TG* xnorm, const float* alpha, const TG* leaves_s,
                                  const TG* lnorm, double* part_var, double* part_mean, int64_t npad,
                                  int dp4, int64_t mpad, const KernParams& kp, const int64_t* m_live,
                                  const float* f16_inv_scale_a = nullptr, int variant = 0, const void* xs_h16 = nullptr,
                                  const float* c16_scale = nullptr, int64_t n_rows = 0, const RawLeaves& rawl = RawLeaves{}) {
#define GPSO_V(FUSED, KS) launch_leaf_tiles_bf16_v<TG, FUSED, KS>(st, nsplit, linv_b, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, f16_inv_scale_a, xs_h16, c16_scale, n_rows, rawl)
  const bool low = kp.kernel == 0 || kp.kernel == 1;
  const int rc = variant != 1 ? (low ? GPSO_V(true, 0) : GPSO_V(true, 1)) : (low ? GPSO_V(false, 0) : GPSO_V(false, 1));
#undef GPSO_V
  return rc;
}
// scal: 2 device floats -- [0] max |L^-1|, [1] := 2^-sa.  have_max false: the maximum is computed here first (memset +
// absmax_kernel); true: the fit left it in scal[0] (launch_solve_alpha: its own pass over L^-1).
// rt0 > 0 (after a gpso_append; scal[0] already holds the new maximum, scal[3] the 2^-sa the resident pieces were packed
// with): only the 16-row tiles from rt0 on are repacked -- unless the maximum crossed a power of two, which the kernel
// sees on the device and then repacks everything
template <typename TF>
void launch_pack_linv_f16(hipStream_t st, const TF* linv, int64_t n, int64_t npad, float* scal, void* linv_b,
                          bool have_max = false, int64_t rt0 = 0);
// what finalising a leaf needs (leaf_finalize_kernel, or fused into the arg-max's first stage: launch_seg_argmax)
struct LeafFinalize {
  const double* part_var = nullptr;
  const double* part_mean = nullptr;
  int nbi = 0;
  int64_t mpad = 0;
  double variance = 0, noise = 0, mean_c = 0, varsigma = 0;
  double* mean = nullptr;
  double* var = nullptr;
  double* ucb = nullptr;
  // nullable: the leaves' squared norms (generation type: float | double).  A leaf with a NaN coordinate has a NaN norm
  // and gets NaN mean / var / ucb -- what GPflow returns for it (and np.argmax then picks it); the kernel map's clamp
  // max(r^2, 1e-36) would otherwise turn its r^2 into 0 against every training point (finite garbage, negative variance)
  const void* lnorm = nullptr;
  int lnorm_f64 = 0;
};
void launch_leaf_finalize(hipStream_t st, const double* part_var, const double* part_mean, int nbi,
                          int64_t mpad, int64_t m, const KernParams& kp, double varsigma,
                          double* mean, double* var, double* ucb, const void* lnorm = nullptr, int lnorm_f64 = 0);
void launch_seg_argmax(hipStream_t st, const double* mean, const double* var, const double* ucb,
                       const int64_t* seg_off_dev, int nseg, int nblk, void* partial_dev,
                       double* out_vals_dev /* [nseg*4 + 2]: per segment mean, var, ucb, bit-cast int64 index; spare; 0.0 (status slot) */,
                       const LeafFinalize* fin = nullptr /* the leaves are still partial sums: finalise them in stage 1 */,
                       double* host_vals = nullptr /* pinned host copy of out_vals, written by the kernel itself */,
                       double done_token = 0.0 /* nseg == 1 only: host_vals[nseg * 4 + 2] := token behind the records */);
// arg-max over a de-duplicated, keyed leaf list (grow.hip: launch_grow_unique); out_vals_dev[nseg*4 + 2]:
// per segment mean, var, ucb, bit-cast reference row index; then the bit-cast live row count; then 0.0 (status slot)
void launch_keyed_argmax(hipStream_t st, const double* mean, const double* var, const double* ucb,
                         const int64_t* key_dev, int64_t rows, int64_t uniq, int nseg, const int64_t* live_dev,
                         int nblk, void* partial_dev, int64_t* pos_dev, double* out_vals_dev,
                         const LeafFinalize* fin = nullptr, double* host_vals = nullptr, double done_token = 0.0);
// out_dev[c] = number of live leaves inside chunk c of a batch whose total live count is *live_dev
void launch_chunk_live(hipStream_t st, const int64_t* live_dev, int64_t chunk, int nchunk, int64_t* out_dev);
// *acc += sum_i mix(words[i], i, salt)  (64-bit wrap-around sum of a per-word mix: order-independent, so the parallel
// reduction is deterministic) -- the posterior fingerprint of gpso_posterior_hash
void launch_hash_words(hipStream_t st, const void* words, size_t nwords, uint64_t salt, unsigned long long* acc);
constexpr int kArgmaxBlocks = 256;     // stage-1 blocks per segment, at most (argmax_blocks picks: one pass of 256 leaves per block)
inline int argmax_blocks(int64_t m, int nseg) {
  const int64_t per_seg = (m / (nseg > 0 ? nseg : 1) + 255) / 256;
  return (int)(per_seg < 1 ? 1 : per_seg > kArgmaxBlocks ? kArgmaxBlocks : per_seg);
}
constexpr size_t kArgmaxPartialBytes = 16;  // sizeof(Best)

// ---- fit.hip --------------------------------------------------------------------------------------
// scaled inputs: xs[npad*dp] = X/ls (zero padded), xnorm[npad], xs_p (nullable) = MFMA-fragment packing
template <typename T>
void launch_scale_x(hipStream_t st, const double* x64, int64_t n, int64_t npad, int d, int dp,
                    const double* ls, T* xs, T* xnorm, T* xs_p);
// float generation inputs (xs, xnorm, fragment packing) derived from the double scaled inputs
void launch_gen_inputs_f32(hipStream_t st, const double* xs64, int64_t npad, int dp, float* xs, float* xnorm,
                           float* xs_p);
// K = k(X, X) + noise * I on rows < n (lower 64x64 tiles only); identity on the padding.  r^2 is
// always formed in double from the double scaled inputs.
template <typename T>
void launch_gram(hipStream_t st, const double* xs, const double* xnorm, int64_t n, int64_t npad, int dp,
                 const KernParams& kp, T* K, int* info = nullptr /* device: set to INT_MAX ("no failing pivot") */);
// blocked right-looking Cholesky, K (destroyed) -> Lf (lower); also writes the inverted 64x64 diagonal
// blocks into linv, the unrounded diagonal of L to diag64[npad], and the first failing pivot (or
// INT_MAX) to info
// Returns bit flags.  Bit 0: linv already holds the COMPLETE inverse (small sizes: it is built beside
// the factorisation, using work as scratch) and launch_trtri must be skipped.  Bit 1: kinv (nullable:
// only wanted with the gradient) already holds K^-1 = L^-T L^-1 (lower tiles).
// Float fits above the single-level limit can run their large products on the bf16 matrix cores (3-way split,
// six bf16 MFMAs per product: float-class accuracy).  The operands then live as bf16 planes of N_pad x N_pad
// matrices (fit_plane_set_bytes(npad) each): L, X = L^-1, XT = X^T, WT = scratch (fit.hip: gemm_bf16_kernel).
struct FitPlanes {
  unsigned short *L, *X, *XT, *WT;
  int64_t stride;  // 16-bit elements between the planes of a set
  int nkb;         // npad / 32
  // Round 5: the planes hold either three bf16 pieces per value (np = 3: six MFMAs per product) or TWO fp16 pieces of the
  // value times a power of two (np = 2: three MFMAs per product -- the split that carries the predict path): fp16 has 5
  // exponent bits, so every plane set is scaled into its range by a bound known on the host -- |L| <= sqrt(s2 + noise),
  // |L^-1| <= 1 / sqrt(noise), |L[B,A] L^-1[A,A]| <= sqrt((s2 + noise) / noise) -- with 2^13 as the scaled bound (a factor
  // of 8 below fp16's largest number); fit_plane_scales() fills them
  int np = 3;
  float sL = 1.0f, sX = 1.0f, sW = 1.0f;
  // look-ahead (all three set, or none): the next diagonal block is factored on `side` while the rest of the
  // current rank-W update runs on the caller's stream; ev_col / ev_chain order the two streams
  hipStream_t side = nullptr;
  hipEvent_t ev_col = nullptr, ev_chain = nullptr;
  // Round 6 (double fits; L == nullptr): `side` also carries the look-ahead of the float64 two-level fit, and `inv` (nullable)
  // the level-doubling inverse, issued pair by pair as soon as the panels it reads are final -- the first product of the last
  // level starts when HALF of the factorisation is done (launch_potrf); ev_panel / ev_inv order the streams
  hipStream_t inv = nullptr;
  hipEvent_t ev_panel = nullptr, ev_inv = nullptr;
  int overlap = 0;  // double fits: bit 0 look-ahead on `side`, bit 1 the inverse on `inv` (GPSO_OPT_FIT_OVERLAP)
};
inline size_t fit_plane_set_bytes(int64_t npad) { return (size_t)3 * (size_t)npad * (size_t)npad * 2; }
// power-of-two scales of the fp16 planes for hyper-parameters (variance, noise); false: a scale leaves the range in which
// the second piece of a typical entry is still a normal fp16 number (noise / variance below ~1e-9): keep bf16 pieces
bool fit_plane_scales(double variance, double noise, FitPlanes& pl);
// planes (float fits, nullable): with them the TRSM GEMMs also emit L into planes->L and the rank-W trailing
// updates run on the bf16 matrix cores
// does launch_potrf take the single-level path at this size?  (It then writes every entry of L^-1 that anything reads --
// the lower 64-tiles, diagonal tiles whole -- and linv needs no zero fill; the two-level path's GEMMs read whole panels.)
template <typename T>
bool potrf_is_single_level(int64_t npad, int64_t single_level_max /* < 0: default */);
template <typename T>
int launch_potrf(hipStream_t st, T* K, T* Lf, T* linv, T* work, T* kinv, int64_t n, int64_t npad,
                 double* diag64, int* info, int64_t single_level_max /* < 0: default */, const FitPlanes* planes);
// the level-doubling inverse on the bf16 matrix cores (after a launch_potrf with the same planes)
void launch_trtri_bf16(hipStream_t st, float* linv, const FitPlanes& planes, int64_t npad, int64_t first_level,
                       bool keep_xt /* also emit the planes of L^-T at the last level (K^-1 for the gradient) */);
bool trtri_bf16_applies(int64_t npad, int64_t first_level);
// L^-1 by level-doubling from level first_level (64 or the factorisation's outer panel width): needs the
// inverses of the first_level-wide diagonal blocks already in linv; work = npad x npad scratch
// width of the diagonal blocks of the two-level factorisation (and first level of the level-doubling
// inverse).  Measured (posterior fit, float, ms, 512 | 1024): N 4096 1.78 | 1.66, 8192 6.16 | 5.91,
// 16384 29.95 | 30.18 -- the wider panel halves the passes over the trailing matrix, which pays while
// the SYRKs are short.
inline int fit_outer_panel(int64_t npad) { return (npad >= 4096) ? 1024 : 512; }
template <typename T>
void launch_trtri(hipStream_t st, const T* L, T* linv, T* work, int64_t npad, int64_t first_level);
// zero rows/cols >= n and re-tile the lower 16x16 tiles of L^-1 into the MFMA fragment-major layout
// the predict kernel reads: npad16 (npad16 + 1) / 2 tiles of 256 elements
template <typename TF, typename TP>
void launch_pack_linv(hipStream_t st, const TF* linv, int64_t n, int64_t npad, TP* linv_p, int64_t rt0 = 0 /* first 16-row tile row to pack */);
inline size_t packed_linv_elems(int64_t npad) { return (size_t)(npad / 16) * (size_t)(npad / 16 + 1) / 2 * 256; }
// a = L^-1 (y - c), alpha = L^-T a, nlml = 1/2 a.a + sum log diag64 + n/2 log 2pi  (double accumulators);
// kinv_diag[npad] = squared column norms of L^-1 = diag((K + noise I)^-1)
constexpr int kAlphaChunk = 64;
inline size_t alpha_part_doubles(int64_t npad) { return (size_t)2 * ((npad + kAlphaChunk - 1) / kAlphaChunk) * npad; }
template <typename T>
void launch_solve_alpha(hipStream_t st, const T* linv, const double* y64, int64_t n, int64_t npad,
                        double mean_c, const double* diag64, T* white, T* alpha,
                        double* alpha_part /* alpha_part_doubles(npad) scratch */, double* kinv_diag,
                        double* nlml_out, void* alpha_p = nullptr /* predict-type copy of alpha */, int alpha_p_f64 = 0,
                        float* amax_rows = nullptr /* [npad] scratch */, float* linv_absmax = nullptr /* receives max |L^-1| */);
// Kinv = L^-T L^-1 (lower tiles; skipped when kinv_ready), then the gradient reductions of SURVEY.md A.3;
// grad_out[n_ls + 3] = d nlml / d (ls..., variance, noise, c)
template <typename T>
void launch_gradient(hipStream_t st, const T* linv, const T* alpha, const double* xs, const double* xnorm,
                     int64_t n, int64_t npad, int d, int dp, int n_ls, const double* ls,
                     const KernParams& kp, T* kinv, bool kinv_ready, double* partial, double* grad_out,
                     const FitPlanes* xt_planes = nullptr /* float: planes of L^-T left by launch_trtri_bf16(keep_xt) */);
constexpr int kGradMaxLs = 64;
// fused single-launch fit for N <= 128 (one workgroup, matrices in LDS): everything launch_scale_x<double>
// .. launch_pack_linv / launch_convert_vec produce, in one kernel (definition + field docs: fit.hip)
struct SmallFitArgs {
  const double* x64;
  const double* y64;
  double ls[48];     // lengthscale per input dimension (isotropic: repeated), by value: the kernel also
                     // writes the hyper-parameter block the predict path reads, so the evaluation needs no
                     // host-to-device copy of its own
  double* hyper;     // [8 + 48] hyper-parameter block (device, output)
  int n, d, dp, kernel, n_ls, want_grad;
  int zero_tile_rows;  // 16-row tile rows of linv_p an earlier fit may have left non-zero (8 = unknown)
  double variance, noise, mean_c;
  // outputs
  double* xs64;    // [128 * dp] scaled inputs
  double* xnorm64; // [128]
  double* xs_p64;  // [128 * dp] MFMA A fragments (double layout)
  void* Lf;        // [128 * 128] T
  void* linv;      // [128 * 128] T
  void* kinv;      // [128 * 128] T, nullable
  void* white;     // [128] T
  void* alpha_f;   // [128] T
  void* alpha_p;   // [128] TP
  void* linv_p;    // packed lower tiles, TP
  double* diag64;  // [128]
  double* kinv_diag;  // [128]
  double* scal;    // [0] nlml, [1] info (int), [8 ..] gradient (ls..., variance, noise, c)
  double* scal_host;  // nullable: pinned host memory (device-visible) that receives the same scalars straight from the
                      // kernel -- the evaluation then needs no device-to-host copy operation behind it either
  double done_token;  // != 0: scal_host[7] := token behind the scalars (system-scope release; the host spins on it)
};

bool small_fit_eligible(int64_t n, int dp);
template <typename T, typename TP>
int launch_small_fit(hipStream_t st, const SmallFitArgs& args);

// precision self-test: predictions at the training inputs vs the closed form the fit implies;
// out[6] = max |d mean|, max |d var|, max |y - c|, min predicted var, max |alpha|, max_i (K_y^-1)_ii
template <typename T>
void launch_selftest(hipStream_t st, const double* mean, const double* var, const double* y64,
                     const T* alpha, const double* kinv_diag, int64_t n, double noise, double mean_c,
                     double* out);

// conversions and getters
template <typename TS, typename TD>
void launch_convert_vec(hipStream_t st, const TS* src, TD* dst, int64_t len);
template <typename T>
void launch_convert_in(hipStream_t st, const double* src, T* dst, int64_t rows, int64_t cols,
                       int64_t ld_dst);
template <typename T>
void launch_convert_out(hipStream_t st, const T* src, int64_t ld_src, double* dst, int64_t rows,
                        int64_t cols, int lower_only);
// install L (row-major lower, n x n float64 on device) into the padded T buffer + invert diagonal blocks
template <typename T>
void launch_install_chol(hipStream_t st, const double* L64, int64_t n, int64_t npad, T* K, T* linv);

// ---- append.hip: rank-k append at fixed hyper-parameters ----------------------------------------------------------------
// The posterior of the first n points is resident; k <= kAppendMax new points (already copied behind the old ones in
// x64 / y64) extend L, L^-1, a, alpha, diag(K_y^-1), the NLML and the scaled inputs in place: two passes over L^-1
// (append.hip).  Everything below lives on the device; scratch = append_scratch_doubles(npad, append_kp(k)) doubles.
constexpr int kAppendMax = 64;
struct AppendArgs {
  double* x64;         // [npad * d] raw inputs: rows n .. n + k - 1 are filed by the cross kernel
  double* y64;         // [npad]
  const double *xnew, *ynew;  // the k new points [k * d], [k] in pinned HOST memory (read by the cross kernel itself: no copies)

  const double* ls;    // lengthscale per input dimension (device)
  int64_t n, npad;
  int k, kp, d, dp, kernel;
  double variance, noise, mean_c;
  double *xs64, *xnorm64, *xs_p64;  // scaled inputs (rows n .. n + k - 1 are written)
  double *Kc, *Bm, *part, *sm;      // scratch (carved by launch_append)
  int* info;                        // scratch: INT_MAX, or the first failing pivot (n + p)
  double* diag64;                   // [npad] diagonal of L
  double* nlml;                     // device scalar: the resident NLML, updated in place
  double* kinv_diag;                // [npad] squared column norms of L^-1
  void* alpha_p;                    // [npad] predict-type copy of alpha
  double* hyper;                    // hyper block: slot 0 (n) is updated
  float* f16_scal;                  // nullable: scale slot of the fp16 split ([0] max |L^-1| grows by atomicMax, [3] := [1])
  double* host_out;                 // pinned host: [0] the new NLML, [1] 1.0 = extended / 2.0 = not positive definite, [2] pivot
};
int append_kp(int k);
size_t append_scratch_doubles(int64_t npad, int kp);
template <typename TF, typename TP>
void launch_append(hipStream_t st, AppendArgs a, void* scratch, TF* linv, TF* Lf, TF* white, TF* alpha_f);

// ---- grow.hip -------------------------------------------------------------------------------------
// centres of the ternary subtree (levels 0..depth-1) under each box; out[nseg*rows*d] float64
void launch_grow(hipStream_t st, const double* bounds_dev, int nseg, int d, int depth,
                 double* out_dev);
// the same centres WITHOUT the rows that repeat an earlier row bit for bit (a centre child repeats its
// parent): grow_unique_rows(depth) = 3^(depth-1) analytic slots per box, then the centre children whose
// centre differs from the parent's in the last bit; key_dev[slot] = seg * rows + reference row index;
// *count_dev must hold nseg * grow_unique_rows(depth) on entry and holds the live row count on exit
// Only the reference rows [row_lo, row_hi) of every box (a rank's share of a sharded call; the whole list
// is [0, rows)): analytic slots per box = grow_unique_before(row_hi) - grow_unique_before(row_lo).
int64_t grow_unique_rows(int depth);
int64_t grow_unique_before(int64_t row);
void launch_grow_unique(hipStream_t st, const double* bounds_dev, int nseg, int d, int depth, int64_t row_lo,
                        int64_t row_hi, double* out_dev, int64_t* key_dev, int64_t* count_dev);
// Small calls (round 4, launch-bound: the optimiser's exploration levels): growth and input scaling in ONE launch with
// the boxes passed by value -- rows land scaled (x / l in TG, norms) where the tile kernel reads them, keys as above; a
// row that does not repeat its parent goes to nseg * uniq + atomicAdd(extra); *extra_dev is zero between calls
constexpr int kGrowBoxDoubles = 96;  // nseg * d * 2 doubles of boxes fit the kernel-argument struct
struct GrowBoxes {
  double b[kGrowBoxDoubles];
};
template <typename TG>
void launch_grow_unique_prep(hipStream_t st, const GrowBoxes& boxes, int nseg, int d, int dp, int depth, int64_t row_lo,
                             int64_t row_hi, const double* ls_dev, TG* leaves_s, TG* lnorm, int64_t* key_dev,
                             unsigned long long* extra_dev);
// ... and the whole arg-max of a small batch in ONE launch of one workgroup: finalises every live leaf (LeafFinalize),
// reduces per segment with np.argmax's rule, writes the records (device and pinned host) and resets *extra_dev
struct SmallBest {
  LeafFinalize fin;
  const int64_t* key = nullptr;       // keyed (grown) batches: reference keys, rows per box, analytic slots per box
  int64_t rows = 0, uniq = 0, base = 0;
  unsigned long long* extra = nullptr;
  const int64_t* seg_off = nullptr;   // plain batches: segment offsets (device), total leaves m
  int64_t m = 0;
  int nseg = 0;
  double* out_vals = nullptr;         // [nseg * 4 + 2]
  double* host_vals = nullptr;        // nullable pinned host copy
  double done_token = 0.0;            // != 0: written to host_vals[nseg * 4 + 2] behind the records (system-scope release):
                                      // the host spins on it instead of polling an event (round 5)
};
// ... and, where ONE row block covers L^-1 (N_pad = 128 or 256) and the native tile kernel runs, the whole call in ONE
// launch (predict.hip: leaf_tiles_v2_one_kernel): rows made in the prologue, leaves finalised and reduced in the epilogue,
// the last workgroup to arrive writes the records
struct OneLaunch {
  int mode = 0;  // 1: rows grown from `boxes` (all reference rows of every box), 2: raw rows at `raw`
  GrowBoxes boxes;
  int nseg = 0, d = 0, depth = 0, raw_f64 = 0;
  int64_t rows = 0, uniq = 0;  // grown: reference rows and analytic slots per box
  int64_t total = 0;           // live rows: nseg * uniq (grown) or m (raw)
  const void* raw = nullptr;
  const int64_t* seg_off = nullptr;  // raw: segment offsets (device)
  const double* ls = nullptr;        // lengthscale per dimension (device)
  void* leaves_s = nullptr;          // [mpad * dp] TG: the scaled rows (written by the prologue, read by the tile code)
  void* lnorm = nullptr;             // [mpad] TG
  int64_t* key = nullptr;            // [mpad]: reference keys of grown rows
  LeafFinalize fin;
  void* partial = nullptr;           // [grid * nseg] winners per workgroup (ucb, id)
  int64_t* ppos = nullptr;           // [grid * nseg] their rows
  unsigned* ticket = nullptr;        // arrival counter, zero between calls
  unsigned* fallback = nullptr;      // raised when a centre child does not repeat its parent; zero between calls
  double* out_vals = nullptr;        // [nseg * 4 + 2]: records, live rows (bit-cast), status (1.0 = fallback wanted)
  double* host_vals = nullptr;       // nullable pinned host copy
  double done_token = 0.0;           // as SmallBest::done_token
};
template <typename T, typename TG>
int launch_leaf_tiles_one(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm, const T* alpha,
                          double* part_var, double* part_mean, int64_t npad, int dp4, int64_t mpad, const KernParams& kp,
                          const OneLaunch& one);
constexpr int64_t kSmallBestMaxRows = 16384;
void launch_small_best(hipStream_t st, const SmallBest& a, bool keyed);
// winners of several ranks -> the global winner per segment, np.argmax order on (ucb, global index):
// gathered[world][stride], a rank's payload = nseg x (mean, var, ucb, bit-cast index) [, spare, status: stride ==
// kGroupPayload(nseg)]; base[world][nseg] (nullable) is added to a rank's indices first; out[nseg][4] [, bit-cast rank
// of the worst status, worst status]
void launch_reduce_winners(hipStream_t st, const double* gathered, const int64_t* base, int world, int nseg,
                           int stride, double* out, double* host_out = nullptr /* pinned host copy of out, written by the kernel */,
                           double done_token = 0.0 /* nseg <= 64 (one workgroup): host_out[nseg * 4 + 2] := token behind the records */);
// doubles of one rank's group payload: the winners, the live row count of a growth call (bit-cast), the status
inline int group_payload_doubles(int nseg) { return nseg * 4 + 2; }

}  // namespace gpso
