/*
 * gpso_hip.h -- C-ABI of libgpso_hip.so: the MI355X (gfx950) GP-surrogate + leaf-UCB engine.
 *
 * The reference (jajcayn/pygpso v0.6.1) is pure Python and has NO native interface: its numerics
 * are GPflow/TensorFlow calls.  Every entry point below therefore cites the reference call site
 * it replaces (paths relative to the reference repo) instead of an existing FFI symbol.  The
 * Python binding a maintainer adds is the ctypes stub in INTEGRATION.md (shipped as
 * pygpso_amd/_lib.py).
 *
 * Conventions
 *   - plain pointers and sizes only; all host matrices are C-contiguous row-major float64;
 *   - every call returns GPSO_OK (0) or a negative GPSO_E_* code; the message is available from
 *     gpso_last_error(ctx) (ctx == NULL: the message of the last failed gpso_create on this thread);
 *   - the library owns its device memory; the caller owns every buffer it passes in;
 *   - calls are synchronous at the boundary: host outputs are valid on return;
 *   - a context is bound to one device and one HIP stream, is NOT thread-safe and NOT fork-safe
 *     (the reference drives this path from a single Python thread, gpso/optimisation.py:594-622);
 *   - "mem" arguments say where a caller buffer lives: GPSO_MEM_HOST or GPSO_MEM_DEVICE
 *     (a device pointer valid on the context's device, e.g. torch.Tensor.data_ptr()).
 */
#ifndef GPSO_HIP_H
#define GPSO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpso_ctx gpso_ctx;

/* status codes */
#define GPSO_OK 0
#define GPSO_E_ARG (-1)   /* bad argument / shape                                             */
#define GPSO_E_HIP (-2)   /* a HIP runtime call failed                                        */
#define GPSO_E_NOTPD (-3) /* K + noise*I not positive definite (message names the pivot);     */
                          /* the reference lets TF's InvalidArgumentError escape here         */
#define GPSO_E_OOM (-4)   /* device allocation failed                                         */
#define GPSO_E_STATE (-5) /* call order: no training data / no posterior resident yet         */
#define GPSO_E_RCCL (-6)  /* an RCCL call of the multi-GPU group failed (or librccl could not be loaded) */
#define GPSO_E_PRECISION (-7) /* float predict arithmetic cannot meet the stated tolerance on this */
                          /* posterior (measured by the self-test, see gpso_precision_info): open a   */
                          /* GPSO_MIXED or GPSO_F64 context instead                                   */

/* arithmetic type of a context (what the kernels compute in).  Also the type tag of leaf buffers
 * (xs_dtype: GPSO_F64 | GPSO_F32 only). */
#define GPSO_F64 0   /* fit and predict in float64 (the reference's dtype: gpflow.default_float)      */
#define GPSO_F32 1   /* fit (Gram, Cholesky, L^-1, alpha) and predict apply in float32                */
#define GPSO_MIXED 2 /* fit in float64, predict apply in float32 (or split bf16): the hyper-parameter */
                     /* path is bit-identical to GPSO_F64, only predictions carry float rounding      */

/* stationary kernels, [gpflow.kernels]: the objects passed as gp_kernel at gpso/gp_surrogate.py:393-434 */
#define GPSO_MATERN52 0 /* default, gpso/gp_surrogate.py:424 */
#define GPSO_MATERN32 1
#define GPSO_MATERN12 2
#define GPSO_SQEXP 3 /* SquaredExponential / RBF */

#define GPSO_MEM_HOST 0
#define GPSO_MEM_DEVICE 1

/* options (gpso_set_option) */
#define GPSO_OPT_PREDICT_MATH 1 /* arithmetic of the L^-1 apply in predict/best_ucb, float-predict contexts: */
#define GPSO_MATH_NATIVE 0      /*   f32 MFMA (what GPSO_F64 contexts always use, in double)             */
#define GPSO_MATH_AUTO 1        /*   default: a ladder walked per posterior -- F16X3 where the posterior */
                                /*   allows it (padded N a multiple of 256, leaf fragments fit LDS) and   */
                                /*   its self-test passes with it, else BF16X6, else the f32 MFMA kernel  */
#define GPSO_MATH_BF16X3 3      /*   2-way bf16 split, 3 bf16 MFMAs per product: |d var| ~ 2e-5 sigma^2  */
#define GPSO_MATH_BF16X6 6      /*   3-way bf16 split, 6 bf16 MFMAs per f32 product, f32 accumulation:   */
                                /*   f32-class accuracy (measured beside native f32 in bench.py)         */
#define GPSO_MATH_F16X3 13      /*   2-way fp16 split (2 x 11 bits), 3 fp16 MFMAs per product, operands   */
                                /*   scaled by powers of two into fp16's range, f32 accumulation: each    */
                                /*   operand carried to 2^-22, the dropped low x low product 2^-22        */
#define GPSO_OPT_FIT_SINGLE_LEVEL_MAX 2 /* tuning / test hook: largest padded N whose Cholesky runs single-level */
                                        /* with L^-1 built beside it (default 3584, double 2560); 0 = always two-level +     */
                                        /* level-doubling triangular inverse.  Results agree to rounding.       */
#define GPSO_OPT_GENERATION 3   /* float-predict contexts: arithmetic of the cross-Gram x.x* contraction / r^2 */
#define GPSO_GEN_F64 0          /*   double: r^2 free of float cancellation error                              */
#define GPSO_GEN_F32 1          /*   float: GPflow's GEMM form evaluated in float (|d r^2| ~ 1e-5 at l ~ 0.1) */
#define GPSO_GEN_AUTO 2         /*   (default) float when the precision self-test of the posterior at hand     */
                                /*   passes with it by a margin of 8x, double otherwise; double when no self-test can run and    */
                                /*   for Matern-1/2 (its sqrt at r -> 0 is blind to the test)                  */
#define GPSO_OPT_PRECISION_CHECK 4 /* 1: the first predict after every fit runs the self-test and returns       */
                                   /* GPSO_E_PRECISION when it fails (default in GPSO_F32 / GPSO_MIXED); 0: off  */
                                   /* (default in GPSO_F64)                                                     */
#define GPSO_OPT_FIT_FUSED_SMALL 5 /* 1 (default): N <= 128 is fitted by the single-launch, single-workgroup     */
                                   /* kernel; 0: the general multi-launch path (test hook: results agree to     */
                                   /* rounding)                                                                 */
#define GPSO_OPT_FIT_BF16_SYRK 6   /* GPSO_F32 contexts, N above the single-level limit: the large products of the fit     */
                                   /* (rank-W trailing updates, level-doubling inverse, K^-1) run on the 16-bit matrix     */
                                   /* cores as split products with f32 accumulation.  2 (default): TWO fp16 pieces of the  */
                                   /* power-of-two scaled operands, three MFMAs per product (the dropped low x low term is */
                                   /* 2^-22 relative) -- where the hyper-parameters keep every operand in fp16's range      */
                                   /* (noise / variance above ~1e-9), else as 1; 1: THREE bf16 pieces, six MFMAs per       */
                                   /* product (round 3-4; f32-class entry by entry); 0: f32 MFMA                            */
#define GPSO_OPT_TIMING 7          /* 1 (default): every fit / predict-type entry point records the event pairs           */
                                   /* gpso_last_ms reads (two to four HIP calls + an elapsed-time query: 10-25 us of a     */
                                   /* call); 0: none, gpso_last_ms returns 0 -- for callers in a loop of small evaluations */
                                   /* (the drop-in surrogate switches it off); k > 1: every k-th call is timed, the others */
                                   /* leave gpso_last_ms at the last sample (bench.py: sampled kernel times inside its     */
                                   /* timed region)                                                                        */
#define GPSO_OPT_SPLIT_KERNEL 8    /* which step the split predict kernels (F16X3, BF16X6, BF16X3) run: 0 (default) the  */
#define GPSO_SPLIT_KERNEL_AUTO 0   /*   FUSED step of round 4 (every wave applies step q with the generation of step q+1  */
#define GPSO_SPLIT_KERNEL_TWO_PHASE 1 /* dealt into its MFMA shadows), 1 round 3's two-phase step.  Both give the SAME    */
                                   /*   BITS (tests/test_gpu_parity.py); the option exists for that comparison            */
#define GPSO_SPLIT_KERNEL_FUSED16 2 /* the fused step by name (= AUTO)                                                     */
#define GPSO_SPLIT_KERNEL_FUSED32 3 /* the fused step on the 32x32x16 matrix instruction: fewer issue slots, 4.9 % fewer  */
                                   /*   clocks, an 11 % lower clock -- 7 % slower; only in builds with -DGPSO_STEP32=1,    */
                                   /*   GPSO_E_ARG otherwise.  Same tolerances, not the same bits.                         */
#define GPSO_OPT_SMALL_CALLS 9     /* best-UCB calls on small batches (<= 16384 rows) are launch-bound.  1 (default): a   */
                                   /* short sequence -- THREE launches (growth + input scaling with the boxes by value,    */
                                   /* tiles, one-workgroup finalize + arg-max writing pinned host memory), or ONE where   */
                                   /* that measures faster (native tile kernel, N_pad = 256: rows made in the kernel's     */
                                   /* prologue, leaves finalised and reduced in its epilogue, the last workgroup to arrive */
                                   /* writes the records); no copy operation either way.  2: three launches only; 3: one   */
                                   /* launch wherever it applies (N_pad = 128 too); 0: always the general sequence (2      */
                                   /* copies in, 5 launches, a copy back).  Same bits in all (tests/test_gpu_parity.py)    */
#define GPSO_OPT_CONTRACTION 10    /* the x.x* contraction of the fp16-split kernel (GPSO_MATH_F16X3) under float         */
#define GPSO_CONTRACTION_AUTO 0    /*   generation: 0 (default) on the fp16 pipe -- the scaled inputs split into fp16      */
#define GPSO_CONTRACTION_F32 1     /*   piece pairs like L^-1, three products, the training input's norm riding in a spare */
#define GPSO_CONTRACTION_F16 2     /*   slot; 1: the f32 matrix instruction (rounds 1-3); 2: same as 0 today.  Different  */
                                   /*   roundings of r^2, both inside the float class: the self-test rules on whichever    */
                                   /*   runs                                                                                */
#define GPSO_OPT_FUSED_PREP 11     /* 1 (default): float leaves (GPSO_F32) of a one-chunk batch are scaled by the lengthscales in  */
                                   /* the fp16-contraction kernel's own prologue -- no prep launch in front of it; 0: the separate */
                                   /* prep kernel.  Same bits (tests/test_gpu_parity.py); the option exists for that comparison     */
#define GPSO_OPT_FIT_OVERLAP 12    /* float64 fits above the single-level limit (N_pad > 2560; the fit of GPSO_F64 and GPSO_MIXED      */
                                   /* contexts).  Bit 1 (value 2, default): the level-doubling inverse runs on a second stream, pair   */
                                   /* by pair, as soon as the panels a product reads are final -- beside the rest of the factorisation */
                                   /* instead of behind it.  Bit 0: the next diagonal block is factored on a side stream beside the    */
                                   /* trailing update (as the float fit does; measured not to pay in double: the chain's launches need */
                                   /* whole free CUs and starve behind the update's tiles).  0: round 5's sequential schedule.  The    */
                                   /* same products in the same order on the same tiles: the same bits whatever the value.             */
#define GPSO_OPT_ROW_LOOP 13       /* 1 (default): a workgroup of the split predict kernels keeps its 256 leaves and loops over row     */
                                   /* blocks of L^-1, the row blocks of a leaf tile shared by as many workgroups as gives the shortest  */
                                   /* modelled makespan (C3: one -- 256 workgroups of 288 k-steps instead of 2 048 of 8 .. 64; ragged   */
                                   /* batches and batches whose live count only the device knows: one row block per workgroup);        */
                                   /* 0: one row block per workgroup always (rounds 1-5); v >= 2: exactly min(v, row blocks)            */
                                   /* workgroups per leaf tile (tests).  Process-wide.  Same bits whatever the value.                   */
/* floating-point options (gpso_set_option_f64): tolerances of the self-test */
#define GPSO_OPTF_TOL_VAR 100  /* max |d var| at the training inputs, relative to the kernel variance (default 1e-4; GPSO_F32: 1e-3) */
#define GPSO_OPTF_TOL_MEAN 101 /* max |d mean| at the training inputs, relative to max |y - c|   (default 1e-4; GPSO_F32: 1e-3) */

/* which device-resident matrix / vector the debug getters copy out (as float64) */
#define GPSO_MAT_CHOL 0  /* L, lower Cholesky factor of K + noise*I (upper triangle returned as 0) */
#define GPSO_MAT_LINV 1  /* L^-1 (lower)                                                        */
#define GPSO_MAT_KINV 2  /* (K + noise*I)^-1, valid after a gpso_fit_eval with grad != NULL      */
#define GPSO_MAT_GRAM 3  /* K + noise*I as assembled (only valid before factorisation: debug)    */
#define GPSO_VEC_ALPHA 0 /* alpha = (K + noise*I)^-1 (y - c)                                     */
#define GPSO_VEC_WHITE 1 /* a = L^-1 (y - c)                                                     */

/* ---- lifetime ---------------------------------------------------------------------------- */

/* Replaces: constructing gpflow.models.GPR (gpso/gp_surrogate.py:488-495) -- the model object
 * that owns data, hyper-parameters and cached factorisation.  dtype: GPSO_F64 | GPSO_F32. */
int gpso_create(gpso_ctx** out, int device, int dtype);
void gpso_destroy(gpso_ctx* ctx);
const char* gpso_last_error(const gpso_ctx* ctx);

/* Run all work on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead
 * of the context's own stream; NULL restores the private stream. */
int gpso_set_stream(gpso_ctx* ctx, void* hip_stream);
int gpso_synchronize(gpso_ctx* ctx);
/* No counterpart in the reference (GPflow computes in float64 throughout): selects how the
 * predict path multiplies by L^-1.  The split modes work on row blocks of 256: GPSO_F32 / GPSO_MIXED
 * contexts pad N > 128 to a multiple of 256 for them (GPSO_F64 contexts and N <= 128 pad to 128
 * and run the native kernel), and keep an extra nsplit * N_pad^2 16-bit copy of L^-1. */
int gpso_set_option(gpso_ctx* ctx, int option, int value);
int gpso_set_option_f64(gpso_ctx* ctx, int option, double value);
/* Order the context's stream behind everything already queued on producer_stream (a hipStream_t, e.g.
 * torch.cuda.current_stream().cuda_stream; NULL = the legacy default stream): call it before handing
 * the library device buffers that another stream is still writing (GPSO_MEM_DEVICE leaves / outputs).
 * The library itself is synchronous at the boundary, so no ordering is needed after a call returns. */
int gpso_wait_stream(gpso_ctx* ctx, void* producer_stream);

/* ---- fit (GPSurrogate.gp_update -> GPRSurrogate._gp_train) --------------------------------- */

/* Replaces: GPSurrogate.current_training_data hand-off + `model.data = (x, y)`
 * (gpso/gp_surrogate.py:232-244, :498).  X[N*D], y[N] float64 host. */
int gpso_set_data(gpso_ctx* ctx, const double* X, const double* y, int64_t n, int d);

/* Replaces: ONE evaluation of gpflow GPR.training_loss (+ its reverse-mode gradient) inside
 * optimiser.minimize (gpso/gp_surrogate.py:500-503).  Hyper-parameters are the CONSTRAINED values:
 * lengthscales[n_ls] (n_ls = 1 isotropic, or = D for ARD), kernel variance, noise variance,
 * constant mean.  Outputs: *nlml = 1/2 |L^-1(y-c)|^2 + sum log L_ii + N/2 log 2pi;
 * grad (nullable) = d nlml / d (lengthscales..., variance, noise, mean_c), n_ls + 3 values.
 * Leaves L, L^-1, alpha resident (the posterior is predict-ready afterwards). */
int gpso_fit_eval(gpso_ctx* ctx, int kernel, const double* lengthscales, int n_ls, double variance,
                  double noise, double mean_c, double* nlml, double* grad);

/* The same evaluation in the OPTIMISER's variables -- what one call of the closure SciPy's L-BFGS-B drives costs in
 * the reference: GPflow's parameter transforms, training_loss and its reverse-mode gradient
 * (gpso/gp_surrogate.py:500-503; transforms: SURVEY.md Appendix A.1).  u = the unconstrained vector in tf.Module's
 * order: lengthscales[n_ls], kernel variance, likelihood variance [, constant mean when train_mean != 0; otherwise the
 * mean is mean_c_fixed].  lengthscale = softplus(u), variance = softplus(u), noise = 1e-6 + softplus(u), mean = u; the
 * transforms and the chain rule d/du = d/dtheta * sigmoid(u) run on the host side of the library with numpy's own
 * formulas (bit-identical to evaluating them in Python and calling gpso_fit_eval).  Outputs: *nlml; grad_u (nullable)
 * [n_ls + 2 + (train_mean != 0)]; theta_out (nullable) [n_ls + 3]: the constrained values used (lengthscales...,
 * variance, noise, mean). */
int gpso_fit_eval_u(gpso_ctx* ctx, int kernel, const double* u, int n_ls, int train_mean, double mean_c_fixed,
                    double* nlml, double* grad_u, double* theta_out);

/* Replaces: `model.data = (x, y)` at gpso/gp_surrogate.py:496-498 for an update that KEEPS the hyper-parameters (the
 * reference always re-optimises right after, :500-503, and so refactorises from scratch although N grew by 1-7 points
 * per iteration, gpso/optimisation.py:324-329).  With the posterior of the N points resident at theta
 * (after gpso_fit_eval), the k new points Xnew[k*D], ynew[k] (host float64) extend L, L^-1, alpha, the NLML and the
 * predict-ready copies of L^-1 IN PLACE: L21 = K21 L11^-T from one pass over the resident L^-1, L22 = chol(K22 + noise I -
 * L21 L21^T) in one workgroup, the new rows of L^-1 = -L22^-1 L21 L11^-1 from a second pass -- O(N^2 k) and N^2 s bytes
 * instead of O(N^3); arithmetic in double whatever the context's matrix type.  *nlml (nullable) receives the NLML of
 * the N + k points at theta.
 * Returns GPSO_OK (extended in place); 1 when the posterior of the N + k points was instead REFITTED from scratch at
 * theta -- k > 64 (k > 32 below a padded size of 4096, where the refit is as fast), N + k above the padded size
 * (gpso_padded_n: every buffer's layout changes), or N + k <= 128 (the one-launch fit is the shorter exact update
 * there) -- gpso_last_error then says which; GPSO_E_NOTPD with the failing
 * pivot (N + p) when the appended block is not positive definite: the posterior of the N points then stays resident
 * unchanged; GPSO_E_STATE without a posterior fitted on this context.  The gradient-side K^-1 (GPSO_MAT_KINV) is not
 * extended.  The precision self-test runs again before the next prediction of a float-predict context. */
int gpso_append(gpso_ctx* ctx, const double* Xnew, const double* ynew, int64_t k, double* nlml);

/* Interop / debug: install a posterior computed elsewhere (host float64: X[N*D], L[N*N] row-major
 * lower, alpha[N]) -- the device still derives L^-1 and its tile packing itself.  Mirrors loading
 * saved GPflow parameters into a placeholder model (gpso/gp_surrogate.py:463-473). */
int gpso_set_posterior(gpso_ctx* ctx, const double* X, const double* L, const double* alpha,
                       int64_t n, int d, int kernel, const double* lengthscales, int n_ls,
                       double variance, double noise, double mean_c);

/* ---- predict (gpflow_model.predict_y users) ----------------------------------------------- */

/* Replaces: gpflow_model.predict_y(coords) at gpso/gp_surrogate.py:298 (gp_predict) and
 * gpso/plotting.py:351-353.  Xs[M*D] in xs_dtype (GPSO_F64|GPSO_F32) living in xs_mem;
 * mean[M], var[M] float64 living in out_mem; var INCLUDES the noise variance. */
int gpso_predict(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m, double* mean,
                 double* var, int out_mem);

/* Replaces: GPSurrogate.gp_eval_best_ucb (gpso/gp_surrogate.py:313-328): predict_y, then
 * ucb = mean + varsigma * VAR, then first arg-max -- per segment.  seg_off[nseg+1] (host) delimits
 * independent leaf batches (NULL with nseg = 1 means one segment [0, M)); outputs (host, nseg each):
 * idx = winner's row index relative to its segment start, and its mean / var / ucb. */
int gpso_best_ucb(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m,
                  const int64_t* seg_off, int nseg, double varsigma, int64_t* idx, double* mean,
                  double* var, double* ucb);

/* The same calls as a non-blocking pair (round 5).  _begin checks and enqueues the call on the context's stream and
 * returns a ticket (0 or 1) without waiting, or a negative status; gpso_best_ucb_end(ticket) waits for THAT call and fills
 * the outputs (nseg each, nullable).  Two calls may be in flight on a context: the device work of the second queues up
 * behind the first while the host still reads the first's result, so the GPU does not idle over the host's round trip.
 * The reference's loop (gpso/optimisation.py:366-382) is synchronous -- the blocking calls above are _begin + _end --;
 * callers with several independent batches (conditional grids, one batch per tree level of several trees) overlap them.
 * Between a _begin and its _end the leaves buffer (GPSO_MEM_DEVICE) must stay untouched; host leaves are copied before
 * _begin returns.  nseg <= 1024.  Calls that replace or change the posterior (gpso_set_data, gpso_fit_eval, gpso_append,
 * gpso_set_posterior, gpso_alloc_posterior) return GPSO_E_STATE while a ticket is open.  gpso_last_ms is not updated by
 * asynchronous calls. */
int gpso_best_ucb_begin(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off,
                        int nseg, double varsigma);
int gpso_best_ucb_grow_begin(gpso_ctx* ctx, const double* bounds, int nseg, int depth, double varsigma);
int gpso_best_ucb_end(gpso_ctx* ctx, int ticket, int64_t* idx, double* mean, double* var, double* ucb);

/* ---- ternary geometry on device (LeafNode.grow, gpso/param_space.py:175-200,257-307) -------- */

/* Centres of levels 0..depth-1 of the ternary subtree under each of nseg boxes, generated on the
 * device with the reference's float64 recurrence (bit-for-bit), level-major; rows per box =
 * (3^depth - 1) / 2.  bounds[nseg*D*2] = (lo, hi) per dimension, host float64.
 * gpso_grow copies the centres out (host float64 [nseg*rows*D]);
 * gpso_best_ucb_grow scores them without leaving the device: one call = one exploration level
 * of GPSOptimiser._tree_explore (gpso/optimisation.py:366-403).  Rows that repeat an earlier row bit for
 * bit are scored once; idx is the REFERENCE row index (first maximum over the full duplicated list), so
 * the result equals gpso_best_ucb on the rows gpso_grow returns. */
int64_t gpso_grow_rows(int depth);
int gpso_grow(gpso_ctx* ctx, const double* bounds, int nseg, int d, int depth, double* out_coords);
int gpso_best_ucb_grow(gpso_ctx* ctx, const double* bounds, int nseg, int depth, double varsigma,
                       int64_t* idx, double* mean, double* var, double* ucb);

/* ---- multi-GPU group (no counterpart in the reference: it has no distributed path) --------------- */

/* One context per GPU -- in one process per GPU (torchrun / MPI style) or on several threads of one
 * process -- joined into a group through RCCL.  The predict path shards (leaves are independent given
 * the posterior): the GP is fitted on ONE rank, its predict-ready state is broadcast over xGMI, every
 * rank scores a contiguous share of the leaves with no data-path collective, and the per-segment
 * winners (4 doubles each) are all-gathered and folded on the device with np.argmax's order on
 * (ucb, global index) -- so every rank returns the same result, bit-identical to the single-GPU call.
 * librccl is loaded (dlopen) by the first of these calls; failures return GPSO_E_RCCL.
 * Every rank of a group must make the same sequence of group calls with the same arguments (as for any
 * collective); the arithmetic options (dtype, predict math, generation) must match across the group. */
#define GPSO_UNIQUE_ID_BYTES 128
/* rank 0 creates the group id (ncclGetUniqueId) and hands the 128 bytes to every rank by any means
 * (the launcher's rendezvous, MPI, a file); ctx-less: the message of a failure is gpso_last_error(NULL) */
int gpso_comm_unique_id(void* out_id /* [GPSO_UNIQUE_ID_BYTES] */);
/* collective over the group: ncclCommInitRank on the context's device */
int gpso_comm_init(gpso_ctx* ctx, int rank, int world, const void* unique_id);
int gpso_comm_destroy(gpso_ctx* ctx);
/* returns 1 when the context belongs to a group, 0 otherwise; rank / world are filled either way */
int gpso_comm_info(const gpso_ctx* ctx, int* rank, int* world);
/* the contiguous share [lo, hi) of m leaves (or reference rows) that `rank` of `world` handles */
void gpso_shard_range(int64_t m, int rank, int world, int64_t* lo, int64_t* hi);
/* collective: the posterior resident on `root` (after gpso_fit_eval / gpso_set_posterior) becomes
 * resident on every rank -- one RCCL broadcast per buffer of gpso_posterior_buffers, device to device */
int gpso_broadcast_posterior(gpso_ctx* ctx, int root);
/* After gpso_append on the root: bring the peers up to date moving ONLY what the appends wrote -- the new rows of the scaled
 * inputs and of each predict-ready copy of L^-1 (whole 16-row tile rows), alpha, the hyper block: ~1.1 MB instead of the 1.07
 * GB of gpso_broadcast_posterior for 7 new points at N = 16 384 (the reference re-creates its model on every update,
 * gpso/gp_surrogate.py:496-498; there is no hand-off to replace).  Collective, same header / verdict protocol: the root
 * offers rows when its posterior is the one of the last hand-off extended in place (same predict math, same fp16 scale);
 * every rank says whether it holds that base; if all do, the ranges travel (one ncclBroadcast each, grouped) and the call
 * returns GPSO_OK -- otherwise EVERY rank takes the whole range as gpso_broadcast_posterior does and the call returns 1.
 * gpso_last_count(ctx, 0) = the bytes that travelled. */
int gpso_broadcast_posterior_rows(gpso_ctx* ctx, int root);
/* The same decision without a communicator (hand-offs by plain device copies: several contexts of one process, tests): the
 * ranges [offsets[i], offsets[i] + nbytes[i]) of the posterior arena -- offsets as gpso_posterior_span reports them, valid on
 * every context of the same shape and type (gpso_posterior_span_at) -- that a peer holding the posterior of the last
 * hand-off lacks.  Returns their number (<= 9; cap >= 10): 0 = the peer is up to date; when rows do not apply, ONE range: the
 * whole span.  After copying them the receiver calls gpso_adopt_posterior, the sender gpso_posterior_mark_synced (which
 * gpso_broadcast_posterior / _rows do themselves on every rank, and a receiver's gpso_adopt_posterior does not: a receiver
 * that becomes a sender marks explicitly). */
int gpso_posterior_dirty_ranges(gpso_ctx* ctx, int64_t* offsets, int64_t* nbytes, int cap);
int gpso_posterior_mark_synced(gpso_ctx* ctx);

/* collective gpso_best_ucb: xs holds THIS rank's rows gpso_shard_range(m_global, rank, world) of the
 * batch (m_local of them); seg_off[nseg+1] (host, NULL = one segment) is in GLOBAL rows; idx is relative
 * to the global segment start.  Outputs are identical on every rank. */
int gpso_best_ucb_sharded(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m_local,
                          int64_t m_global, const int64_t* seg_off, int nseg, double varsigma,
                          int64_t* idx, double* mean, double* var, double* ucb);
/* collective gpso_best_ucb_grow: every rank generates and scores the reference rows
 * gpso_shard_range(gpso_grow_rows(depth), rank, world) of every box on its own device (O(D) bytes in) */
int gpso_best_ucb_grow_sharded(gpso_ctx* ctx, const double* bounds, int nseg, int depth, double varsigma,
                               int64_t* idx, double* mean, double* var, double* ucb);

/* Collective safety: every rank takes part in every collective of a group call whatever happened in its own half
 * of it (bad arguments, no posterior, GPSO_E_PRECISION from the self-test, a failed allocation): a failing rank
 * sends an empty payload carrying its status, the fold takes the worst status over the ranks, and EVERY rank
 * returns that code (its own message on the rank that failed, "rank r failed ..." on the others).  In particular a
 * posterior that fails the precision self-test on the fitting rank makes gpso_broadcast_posterior return
 * GPSO_E_PRECISION on every rank.  What cannot be agreed on is a failure of HIP or RCCL itself in the middle of
 * a call; gpso_comm_abort (ncclCommAbort), called from another thread for the context whose call is blocked, is
 * the way out of that one -- the blocked call then returns GPSO_E_RCCL / GPSO_E_HIP and the context must leave the
 * group (gpso_comm_destroy) before it joins another. */
int gpso_comm_abort(gpso_ctx* ctx);

/* The two halves of the sharded calls on their own, for an arbitrary (rank, world) and WITHOUT a communicator: what
 * a group of `world` ranks computes can be replayed on one device (tests, debugging; also a transport other than
 * RCCL): gpso_shard_winners / gpso_shard_winners_grow run rank `rank`'s local half of gpso_best_ucb_sharded /
 * gpso_best_ucb_grow_sharded and copy out its payload -- gpso_group_payload_doubles(nseg) doubles: nseg x (mean, var,
 * ucb, bit-cast int64 index), a spare slot, the status of the half (also the return value); gpso_fold_winners takes
 * the payloads of all ranks concatenated in rank order (m_global >= 0 with seg_off: payloads of gpso_shard_winners,
 * whose indices are relative to the local piece of each segment; m_global < 0: of gpso_shard_winners_grow) and
 * returns what the group call returns, the worst status included.  These run the very kernels and index arithmetic
 * of the group calls; only ncclAllGather is replaced by the caller's concatenation. */
int gpso_group_payload_doubles(int nseg);
int gpso_shard_winners(gpso_ctx* ctx, int rank, int world, const void* xs, int xs_dtype, int xs_mem,
                       int64_t m_local, int64_t m_global, const int64_t* seg_off, int nseg, double varsigma,
                       double* payload);
int gpso_shard_winners_grow(gpso_ctx* ctx, int rank, int world, const double* bounds, int nseg, int depth,
                            double varsigma, double* payload);
int gpso_fold_winners(gpso_ctx* ctx, const double* gathered, int world, int64_t m_global, const int64_t* seg_off,
                      int nseg, int64_t* idx, double* mean, double* var, double* ucb);

/* ---- introspection ------------------------------------------------------------------------- */

/* padded problem size the device works with (a multiple of 128; of 256 in float-predict contexts above
 * N = 128), 0 before gpso_set_data */
int64_t gpso_padded_n(const gpso_ctx* ctx);
/* N and D of the problem the context currently holds (0, 0 before any data / posterior arrived) */
int gpso_problem_shape(const gpso_ctx* ctx, int64_t* n, int* d);
int gpso_get_matrix(gpso_ctx* ctx, int which, double* out /* [N*N] host */);
int gpso_get_vector(gpso_ctx* ctx, int which, double* out /* [N] host */);

/* Device pointers + byte sizes of everything a peer GPU needs to predict: hyper-parameter block, packed
 * lower tiles of L^-1, scaled X (double: plain, as MFMA fragments, norms), alpha [, the bf16 pieces of
 * L^-1 when a split-bf16 predict math is on]: fills up to cap (>= 7) entries, returns the count (or a
 * negative status).  On a context that holds a posterior the call first lets GPSO_GEN_AUTO rule (it may
 * run the self-test) and records the choice in the hyper block, so that the receiver generates alike. */
int gpso_posterior_buffers(gpso_ctx* ctx, void** ptrs, int64_t* nbytes, int cap);
/* Round 4: all of those buffers are slices of ONE device allocation per context (the "posterior arena"), laid out as
 * [packed L^-1 | hyper | X / l | its fragments | norms | alpha | scale slot + 16-bit planes of L^-1] -- a function of
 * (padded N, padded D, dtype) only -- so that what the resident posterior USES is one contiguous range whichever predict
 * math it runs (f32 MFMA kernel: [packed L^-1 .. alpha]; split math: [hyper .. the planes in use], the packed L^-1 then
 * stays home).  gpso_posterior_span returns that range: device pointer, its offset inside the arena (the same on every
 * context of the same shape / dtype) and its length; gpso_posterior_span_at maps an (offset, length) announced by a
 * peer onto this context's arena after gpso_alloc_posterior.  gpso_broadcast_posterior moves exactly this range with a
 * single ncclBroadcast.  Replaces (as does gpso_posterior_buffers) nothing in the reference: it has no second device
 * (gpso/gp_surrogate.py:313-328 runs where the GPflow model lives). */
int gpso_posterior_span(gpso_ctx* ctx, void** ptr /* nullable */, int64_t* offset /* nullable */, int64_t* nbytes /* nullable */);
int gpso_posterior_span_at(gpso_ctx* ctx, int64_t offset, int64_t nbytes, void** ptr);
/* 64-bit fingerprint of the predict-ready posterior resident on the context (a position-salted wrap-around sum over the
 * words of the buffers in use; deterministic).  The fit is bit-deterministic, so contexts that ran the same
 * gpso_set_data + gpso_fit_eval hold the same value: a group may REPLICATE the fit on every rank and compare
 * fingerprints instead of broadcasting the factor (SURVEY 8e "measure both"; pygpso_amd.distributed: posterior=
 * "replicate").  Stands where gpso/gp_surrogate.py:500-503 leaves its one GPflow model. */
int gpso_posterior_hash(gpso_ctx* ctx, uint64_t* out);
/* Allocate the same buffers for n, d on a receiving rank so they can be broadcast into. */
int gpso_alloc_posterior(gpso_ctx* ctx, int64_t n, int d);
/* After the buffers were filled by a broadcast: mark the posterior resident (reads the
 * hyper-parameter block back from the device). */
int gpso_adopt_posterior(gpso_ctx* ctx);

/* Precision self-test of the resident posterior (runs it if it has not run since the last fit; needs
 * the training targets, i.e. a posterior produced by gpso_fit_eval on this context): the predict path
 * is evaluated at the training inputs, where mean and variance have the closed form
 * mean_i = y_i - noise alpha_i, var_i = 2 noise - noise^2 (K_y^-1)_ii.  out[12]:
 *   [0] max |d mean|   [1] max |d var|   [2] max |y - c|   [3] min predicted var (incl. noise)
 *   [4] max |alpha|    [5] kernel variance   [6] tolerance on [0] (absolute)   [7] tolerance on [1]
 *   [8] amplification a: 1 for a float64 factor (GPSO_MIXED / GPSO_F64); for a float factor (GPSO_F32)
 *       max(1, sqrt(sigma^2 max_i (K_y^-1)_ii)) -- the training inputs see the factor's backward error
 *       unamplified, a general leaf amplified by the solve weights (calibrated heuristic)
 *   [9] max_i (K_y^-1)_ii
 *   [10] generation arithmetic in use on this posterior: 0 double, 1 float (GPSO_GEN_AUTO's choice)
 *   [11] predict math in use on this posterior (GPSO_MATH_NATIVE / BF16X3 / BF16X6 / F16X3)
 * Returns GPSO_OK, GPSO_E_PRECISION when sqrt(a) [0] > [6] or a [1] > [7] (or a value is not finite), or
 * GPSO_E_STATE when no fitted posterior with targets is resident. */
int gpso_precision_info(gpso_ctx* ctx, double* out);

/* last-call device timings in milliseconds measured with HIP events on the context's stream:
 * what = 0: dominant predict kernel (leaf tiles) of the last predict/best_ucb call,
 *        1: whole last predict/best_ucb call, 2: whole last gpso_fit_eval / gpso_append call,
 *        3: the collective of the last gpso_best_ucb*_sharded call (ncclAllGather of the winners + the fold kernel). */
double gpso_last_ms(gpso_ctx* ctx, int what);

/* leaf counts of the last predict / best_ucb / best_ucb_grow call: what = 0: rows the predict kernels
 * actually scored, 1: rows of the reference's list.  They differ for gpso_best_ucb_grow, which drops the
 * rows of LeafNode.grow that repeat an earlier row bit for bit (a centre child's centre is its parent's,
 * gpso/param_space.py:186-200: (3^depth - 1) / 2 rows per box, 3^(depth-1) distinct).
 * what = 2: the arithmetic the LARGE PRODUCTS of the last gpso_fit_eval ran on (GPSO_FITMATH_*: what a roofline figure of
 * the fit is priced against -- the f32 / f64 matrix instruction, or the two-level float fit's 16-bit pieces: six bf16 or
 * three fp16 MFMAs per f32 product).
 * what = 3: how many workgroups shared a leaf tile's row blocks in the last launch of a split predict kernel (GPSO_OPT_ROW_LOOP;
 * process-wide, for tests and tools). */
#define GPSO_FITMATH_NONE 0
#define GPSO_FITMATH_SMALL 1   /* N <= 128: one launch, vector arithmetic in double */
#define GPSO_FITMATH_F32 2     /* v_mfma_f32_16x16x4_f32 (float fits up to N_pad = 3584, or GPSO_OPT_FIT_BF16_SYRK = 0) */
#define GPSO_FITMATH_F64 3     /* v_mfma_f64_16x16x4_f64 (float64 / mixed contexts) */
#define GPSO_FITMATH_BF16X6 4  /* two-level float fit: three bf16 pieces per operand, six MFMAs per product */
#define GPSO_FITMATH_F16X3 5   /* two-level float fit: two scaled fp16 pieces per operand, three MFMAs per product */
int64_t gpso_last_count(gpso_ctx* ctx, int what);

const char* gpso_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GPSO_HIP_H */
