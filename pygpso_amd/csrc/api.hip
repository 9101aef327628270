// C-ABI of libgpso_hip.so (see include/gpso_hip.h): context, device memory, call sequencing.
// No torch, no BLAS/solver libraries: every kernel launched here is hand-written (predict.hip,
// fit.hip, grow.hip).
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gpso_hip.h"
#include "common.hpp"
#include "kernels.hpp"

using namespace gpso;

namespace {

thread_local std::string g_create_error;

constexpr int kMaxD = 48;                      // padded input dimension limit (LDS budgets)
constexpr int64_t kLeafChunk = (int64_t)1 << 20;  // leaves processed per pass of the tile kernel
constexpr int kHyperHeader = 8;                // doubles in front of the lengthscales

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct Engine {
  virtual ~Engine() {}
  virtual int set_data(const double* X, const double* y, int64_t n, int d) = 0;
  virtual int fit_eval(int kernel, const double* ls, int n_ls, double variance, double noise,
                       double mean_c, double* nlml, double* grad) = 0;
  virtual int set_posterior(const double* X, const double* L, const double* alpha, int64_t n, int d,
                            int kernel, const double* ls, int n_ls, double variance, double noise,
                            double mean_c) = 0;
  virtual int predict(const void* xs, int xs_dtype, int xs_mem, int64_t m, double* mean,
                      double* var, int out_mem) = 0;
  virtual int best_ucb(const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off,
                       int nseg, double varsigma, int64_t* idx, double* mean, double* var,
                       double* ucb) = 0;
  virtual int grow(const double* bounds, int nseg, int d, int depth, double* out) = 0;
  virtual int best_ucb_grow(const double* bounds, int nseg, int depth, double varsigma,
                            int64_t* idx, double* mean, double* var, double* ucb) = 0;
  virtual int get_matrix(int which, double* out) = 0;
  virtual int get_vector(int which, double* out) = 0;
  virtual int posterior_buffers(void** ptrs, int64_t* nbytes, int cap) = 0;
  virtual int alloc_posterior(int64_t n, int d) = 0;
  virtual int adopt_posterior() = 0;
  virtual int64_t padded_n() const = 0;
  virtual int set_option(int option, int value) = 0;
};

}  // namespace

struct gpso_ctx {
  int device = 0;
  int dtype = GPSO_F64;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  std::vector<hipEvent_t> tile_ev;  // start/stop pairs around the leaf-tile kernel, one per chunk
  int tile_pairs = 0;               // pairs recorded by the call in flight
  double last_ms[3] = {0, 0, 0};
  std::string err;
  Engine* eng = nullptr;
  hipEvent_t ev_wait = nullptr;  // completion marker of the call in flight
  double* pinned = nullptr;      // pinned host scratch for the small result read-backs
  size_t pinned_doubles = 0;

  // Wait for everything queued on s.  hipStreamSynchronize parks the thread on an interrupt and costs
  // tens of microseconds to wake up -- as much as the device work of a small fit.  The calling thread
  // is blocked in this library anyway, so poll an event for up to kSpinMs and only then block.
  hipError_t wait(hipStream_t s) {
    constexpr double kSpinMs = 20.0;
    hipError_t e = hipEventRecord(ev_wait, s);
    if (e != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0;; ++it) {
      e = hipEventQuery(ev_wait);
      if (e != hipErrorNotReady) return e;
      if ((it & 63) == 63 &&
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > kSpinMs)
        return hipStreamSynchronize(s);
    }
  }
  double* pinned_scratch(size_t doubles) {
    if (doubles > pinned_doubles) {
      if (pinned) (void)hipHostFree(pinned);
      pinned = nullptr;
      pinned_doubles = 0;
      const size_t want = std::max<size_t>(doubles, 256);
      if (hipHostMalloc(reinterpret_cast<void**>(&pinned), want * 8, hipHostMallocDefault) != hipSuccess) return nullptr;
      pinned_doubles = want;
    }
    return pinned;
  }

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

#define HIPCHECK(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return ctx->fail(e_ == hipErrorOutOfMemory ? GPSO_E_OOM : GPSO_E_HIP, "%s failed: %s", \
                       #call, hipGetErrorString(e_));                                      \
  } while (0)

namespace {

template <typename T>
struct EngineT : Engine {
  gpso_ctx* ctx;
  explicit EngineT(gpso_ctx* c) : ctx(c) {}

  // problem
  int64_t n = 0, npad = 0;
  int d = 0, dp = 0;
  bool have_data = false, have_post = false, have_kinv = false, chol_valid = false;
  KernParams kp{0, 1.0, 1e-3, 0.0};
  int n_ls = 1;
  std::vector<double> ls_host;

  // device buffers
  DevBuf x64, y64, hyper, xs, xnorm, xs_p, K, Lf, linv, work, kinvb, linv_p, white, alpha, logdet, scal, gpart;
  // split-bf16 copy of L^-1 (float contexts with GPSO_OPT_PREDICT_MATH != native)
  DevBuf linv_b;
  int math = GPSO_MATH_NATIVE;
  int64_t single_level_max = -1;  // < 0: library default
  bool linv_b_valid = false;
  std::vector<int64_t> segoff_cache;  // what the device copy of seg_off currently holds
  // predict workspace
  DevBuf leaves_raw, leaves_s, lnorm, pvar, pmean, omean, ovar, oucb, segoff, best, oidx, ovals;

  ~EngineT() override {
    for (DevBuf* b : {&x64, &y64, &hyper, &xs, &xnorm, &xs_p, &K, &Lf, &linv, &work, &kinvb, &linv_p, &white,
                      &alpha, &logdet, &scal, &gpart, &leaves_raw, &leaves_s, &lnorm, &pvar, &pmean,
                      &omean, &ovar, &oucb, &segoff, &best, &oidx, &ovals, &linv_b})
      if (b->p) (void)hipFree(b->p);
  }

  int64_t padded_n() const override { return npad; }

  int nsplit() const { return math == GPSO_MATH_BF16X6 ? 3 : 2; }
  bool bf16_usable() const { return sizeof(T) == 4 && math != GPSO_MATH_NATIVE && npad > 0 && npad % 256 == 0; }

  // (re)build the bf16 pieces of L^-1 from the f32 L^-1 resident in `linv`
  int pack_bf16() {
    linv_b_valid = false;
    if (!bf16_usable()) return GPSO_OK;
    int rc = ensure(linv_b, (size_t)nsplit() * npad * npad * 2);
    if (rc) return rc;
    if constexpr (sizeof(T) == 4) launch_pack_linv_bf16(st(), nsplit(), as<float>(linv), n, npad, linv_b.p);
    HIPCHECK(hipGetLastError());
    linv_b_valid = true;
    return GPSO_OK;
  }

  int set_option(int option, int value) override {
    if (option == GPSO_OPT_FIT_SINGLE_LEVEL_MAX) {
      if (value < 0) return ctx->fail(GPSO_E_ARG, "single-level limit %d must be >= 0", value);
      single_level_max = value;
      return GPSO_OK;
    }
    if (option != GPSO_OPT_PREDICT_MATH) return ctx->fail(GPSO_E_ARG, "unknown option %d", option);
    if (value != GPSO_MATH_NATIVE && value != GPSO_MATH_BF16X3 && value != GPSO_MATH_BF16X6)
      return ctx->fail(GPSO_E_ARG, "unknown predict math %d", value);
    if (value != GPSO_MATH_NATIVE && sizeof(T) != 4)
      return ctx->fail(GPSO_E_ARG, "split-bf16 predict math needs a GPSO_F32 context");
    if (value == math) return GPSO_OK;
    math = value;
    linv_b_valid = false;
    if (chol_valid) return pack_bf16();  // L^-1 is resident: make the new mode usable right away
    return GPSO_OK;
  }
  hipStream_t st() const { return ctx->stream; }
  double* ls_dev() const { return static_cast<double*>(hyper.p) + kHyperHeader; }
  template <typename U>
  U* as(const DevBuf& b) const { return static_cast<U*>(b.p); }

  int ensure(DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return GPSO_OK;
    if (b.p) {
      HIPCHECK(hipStreamSynchronize(st()));
      HIPCHECK(hipFree(b.p));
      b.p = nullptr;
      b.bytes = 0;
    }
    HIPCHECK(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return GPSO_OK;
  }

  int shape(int64_t n_, int d_) {
    if (n_ < 1) return ctx->fail(GPSO_E_ARG, "need at least one training point (n=%lld)", (long long)n_);
    if (d_ < 1 || d_ > kMaxD) return ctx->fail(GPSO_E_ARG, "input dimension %d outside [1, %d]", d_, kMaxD);
    if (n_ > 65536) return ctx->fail(GPSO_E_ARG, "n=%lld above the supported 65536", (long long)n_);
    n = n_;
    d = d_;
    npad = (n + kPadN - 1) / kPadN * kPadN;
    dp = (d + 3) / 4 * 4;
    int rc;
    const size_t s = sizeof(T);
    if ((rc = ensure(x64, (size_t)n * d * 8))) return rc;
    if ((rc = ensure(y64, (size_t)n * 8))) return rc;
    if ((rc = ensure(hyper, (size_t)(kHyperHeader + kMaxD) * 8))) return rc;
    if ((rc = ensure(xs, (size_t)npad * dp * s))) return rc;
    if ((rc = ensure(xnorm, (size_t)npad * s))) return rc;
    if ((rc = ensure(xs_p, (size_t)npad * dp * s))) return rc;
    if ((rc = ensure(linv_p, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(alpha, (size_t)npad * s))) return rc;
    return GPSO_OK;
  }

  int ensure_fit_buffers() {
    int rc;
    const size_t s = sizeof(T);
    if ((rc = ensure(K, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(Lf, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(linv, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(work, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(white, (size_t)npad * s))) return rc;
    if ((rc = ensure(logdet, (size_t)npad * 8))) return rc;
    if ((rc = ensure(scal, (size_t)(8 + kGradMaxLs + 3) * 8))) return rc;
    const size_t nt = (size_t)(npad / 64);
    if ((rc = ensure(gpart, nt * nt * (size_t)(kGradMaxLs + 2) * 8))) return rc;
    return GPSO_OK;
  }

  int set_theta(int kernel, const double* ls, int n_ls_, double variance, double noise, double mean_c) {
    if (kernel < 0 || kernel > 3) return ctx->fail(GPSO_E_ARG, "unknown kernel id %d", kernel);
    if (!(n_ls_ == 1 || n_ls_ == d)) return ctx->fail(GPSO_E_ARG, "n_ls=%d must be 1 or D=%d", n_ls_, d);
    if (n_ls_ > kGradMaxLs) return ctx->fail(GPSO_E_ARG, "too many lengthscales");
    for (int k = 0; k < n_ls_; ++k)
      if (!(ls[k] > 0.0)) return ctx->fail(GPSO_E_ARG, "lengthscale[%d]=%g must be positive", k, ls[k]);
    if (!(variance > 0.0)) return ctx->fail(GPSO_E_ARG, "kernel variance %g must be positive", variance);
    kp.kernel = kernel;
    kp.variance = variance;
    kp.noise = noise;
    kp.mean_c = mean_c;
    n_ls = n_ls_;
    ls_host.assign(ls, ls + n_ls_);
    // staged in pinned memory (upper half of the scratch; the read-backs use the lower half): the copy
    // is then a plain stream operation and needs no host synchronisation here
    static_assert(128 + kHyperHeader + kMaxD <= 256, "pinned scratch layout");
    double* scratch = ctx->pinned_scratch(256);
    if (!scratch) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    double* h = scratch + 128;
    constexpr size_t kHyperBytes = (size_t)(kHyperHeader + kMaxD) * 8;
    std::memset(h, 0, kHyperBytes);
    h[0] = (double)n; h[1] = (double)d; h[2] = (double)kernel; h[3] = (double)n_ls_;
    h[4] = variance; h[5] = noise; h[6] = mean_c;
    for (int k = 0; k < kMaxD; ++k) h[kHyperHeader + k] = ls[n_ls_ == 1 ? 0 : std::min(k, n_ls_ - 1)];
    HIPCHECK(hipMemcpyAsync(hyper.p, h, kHyperBytes, hipMemcpyHostToDevice, st()));
    return GPSO_OK;
  }

  // ------------------------------------------------------------------------------------------
  int set_data(const double* X, const double* y, int64_t n_, int d_) override {
    if (!X || !y) return ctx->fail(GPSO_E_ARG, "X / y must not be NULL");
    int rc = shape(n_, d_);
    if (rc) return rc;
    HIPCHECK(hipMemcpyAsync(x64.p, X, (size_t)n * d * 8, hipMemcpyHostToDevice, st()));
    HIPCHECK(hipMemcpyAsync(y64.p, y, (size_t)n * 8, hipMemcpyHostToDevice, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    have_data = true;
    have_post = have_kinv = chol_valid = false;
    return GPSO_OK;
  }

  int fit_eval(int kernel, const double* ls, int n_ls_, double variance, double noise,
               double mean_c, double* nlml, double* grad) override {
    if (!have_data) return ctx->fail(GPSO_E_STATE, "gpso_fit_eval before gpso_set_data");
    if (!ls) return ctx->fail(GPSO_E_ARG, "lengthscales must not be NULL");
    int rc = ensure_fit_buffers();
    if (rc) return rc;
    if ((rc = set_theta(kernel, ls, n_ls_, variance, noise, mean_c))) return rc;
    have_post = have_kinv = chol_valid = false;
    hipStream_t s = st();
    HIPCHECK(hipEventRecord(ctx->ev[4], s));
    launch_scale_x<T>(s, as<double>(x64), n, npad, d, dp, ls_dev(), as<T>(xs), as<T>(xnorm), as<T>(xs_p));
    launch_gram<T>(s, as<T>(xs), as<T>(xnorm), n, npad, dp, kp, as<T>(K));
    HIPCHECK(hipMemsetAsync(linv.p, 0, (size_t)npad * npad * sizeof(T), s));
    const int imax = INT_MAX;
    int* info_dev = reinterpret_cast<int*>(as<double>(scal) + 1);
    HIPCHECK(hipMemcpyAsync(info_dev, &imax, sizeof(int), hipMemcpyHostToDevice, s));
    if (grad && (rc = ensure(kinvb, (size_t)npad * npad * sizeof(T)))) return rc;
    const int done = launch_potrf<T>(s, as<T>(K), as<T>(Lf), as<T>(linv), as<T>(work),
                                     grad ? as<T>(kinvb) : nullptr, n, npad, as<double>(logdet), info_dev,
                                     single_level_max);
    if (!(done & 1)) launch_trtri<T>(s, as<T>(Lf), as<T>(linv), as<T>(work), npad, kFitOuterPanel);
    launch_solve_alpha<T>(s, as<T>(linv), as<double>(y64), n, npad, mean_c, as<double>(logdet),
                          as<T>(white), as<T>(alpha), as<double>(gpart),
                          as<double>(scal));
    if (grad)
      launch_gradient<T>(s, as<T>(linv), as<T>(alpha), as<T>(xs), as<T>(xnorm), n, npad, d, dp, n_ls,
                         ls_dev(), kp, as<T>(kinvb), (done & 2) != 0, as<double>(gpart), as<double>(scal) + 8);
    launch_pack_linv<T>(s, as<T>(linv), n, npad, as<T>(linv_p));
    if ((rc = pack_bf16())) return rc;
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(ctx->ev[5], s));
    constexpr size_t kHostDoubles = 8 + kGradMaxLs + 3;
    double* host = ctx->pinned_scratch(kHostDoubles);
    if (!host) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    HIPCHECK(hipMemcpyAsync(host, scal.p, kHostDoubles * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(ctx->wait(s));
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->last_ms[2] = ms;
    int info;
    std::memcpy(&info, &host[1], sizeof(int));
    if (info != INT_MAX)
      return ctx->fail(GPSO_E_NOTPD, "K + noise*I is not positive definite: Cholesky failed at pivot %d", info);
    if (nlml) *nlml = host[0];
    if (grad) {
      // device order: ls..., variance, noise, then -sum(alpha)
      for (int h = 0; h < n_ls + 3; ++h) grad[h] = host[8 + h];
      have_kinv = true;
    }
    have_post = chol_valid = true;
    return GPSO_OK;
  }

  int set_posterior(const double* X, const double* L, const double* alpha64, int64_t n_, int d_,
                    int kernel, const double* ls, int n_ls_, double variance, double noise,
                    double mean_c) override {
    if (!X || !L || !alpha64 || !ls) return ctx->fail(GPSO_E_ARG, "NULL argument");
    int rc = shape(n_, d_);
    if (rc) return rc;
    if ((rc = ensure_fit_buffers())) return rc;
    if ((rc = set_theta(kernel, ls, n_ls_, variance, noise, mean_c))) return rc;
    hipStream_t s = st();
    HIPCHECK(hipMemcpyAsync(x64.p, X, (size_t)n * d * 8, hipMemcpyHostToDevice, s));
    have_data = false;  // y unknown: a later fit needs gpso_set_data
    double* tmp = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)n * n * 8 + (size_t)n * 8));
    hipError_t e1 = hipMemcpyAsync(tmp, L, (size_t)n * n * 8, hipMemcpyHostToDevice, s);
    hipError_t e2 = hipMemcpyAsync(tmp + (size_t)n * n, alpha64, (size_t)n * 8, hipMemcpyHostToDevice, s);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      (void)hipFree(tmp);
      return ctx->fail(GPSO_E_HIP, "upload of L / alpha failed");
    }
    launch_scale_x<T>(s, as<double>(x64), n, npad, d, dp, ls_dev(), as<T>(xs), as<T>(xnorm), as<T>(xs_p));
    (void)hipMemsetAsync(linv.p, 0, (size_t)npad * npad * sizeof(T), s);
    launch_install_chol<T>(s, tmp, n, npad, as<T>(Lf), as<T>(linv));
    launch_trtri<T>(s, as<T>(Lf), as<T>(linv), as<T>(work), npad, kFitBlock);
    (void)hipMemsetAsync(alpha.p, 0, (size_t)npad * sizeof(T), s);
    launch_convert_in<T>(s, tmp + (size_t)n * n, as<T>(alpha), 1, n, npad);
    launch_pack_linv<T>(s, as<T>(linv), n, npad, as<T>(linv_p));
    if ((rc = pack_bf16())) {
      (void)hipFree(tmp);
      return rc;
    }
    hipError_t e3 = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e3 != hipSuccess) return ctx->fail(GPSO_E_HIP, "set_posterior: %s", hipGetErrorString(e3));
    HIPCHECK(hipGetLastError());
    have_post = chol_valid = true;
    have_kinv = false;
    return GPSO_OK;
  }

  // ------------------------------------------------------------------------------------------
  // leaves already on the device as raw coordinates (dtype xs_dtype) -> mean/var(/ucb) device arrays
  int score_device_leaves(const void* xs_dev, int xs_dtype, int64_t m, double varsigma, bool want_ucb,
                          double* mean_dev, double* var_dev, double* ucb_dev) {
    // row blocks of L^-1 = partial sums per leaf: the split-bf16 kernel always works on 256-row blocks,
    // the native kernels on the shape leaf_tiles_bm picks
    const bool use_bf16 = bf16_usable() && linv_b_valid;
    const int nbi = use_bf16 ? (int)(npad / 256) : leaf_tiles_nbi<T>(npad, dp / 4);
    const int64_t chunk = std::min<int64_t>(m, kLeafChunk);
    const int64_t cpad = (chunk + kLeafPad - 1) / kLeafPad * kLeafPad;
    int rc;
    if ((rc = ensure(leaves_s, (size_t)cpad * dp * sizeof(T)))) return rc;
    if ((rc = ensure(lnorm, (size_t)cpad * sizeof(T)))) return rc;
    if ((rc = ensure(pvar, (size_t)nbi * cpad * sizeof(T)))) return rc;
    if ((rc = ensure(pmean, (size_t)nbi * cpad * sizeof(T)))) return rc;
    hipStream_t s = st();
    ctx->tile_pairs = 0;
    const size_t in_elem = (xs_dtype == GPSO_F64) ? 8 : 4;
    for (int64_t off = 0; off < m; off += chunk) {
      const int64_t mc = std::min<int64_t>(chunk, m - off);
      const int64_t mp = (mc + kLeafPad - 1) / kLeafPad * kLeafPad;
      const char* src = static_cast<const char*>(xs_dev) + (size_t)off * d * in_elem;
      if (xs_dtype == GPSO_F64)
        launch_prep_leaves<T, double>(s, reinterpret_cast<const double*>(src), mc, mp, d, dp, ls_dev(), as<T>(leaves_s), as<T>(lnorm));
      else
        launch_prep_leaves<T, float>(s, reinterpret_cast<const float*>(src), mc, mp, d, dp, ls_dev(), as<T>(leaves_s), as<T>(lnorm));
      while ((int)ctx->tile_ev.size() < 2 * (ctx->tile_pairs + 1)) {
        hipEvent_t e;
        HIPCHECK(hipEventCreate(&e));
        ctx->tile_ev.push_back(e);
      }
      HIPCHECK(hipEventRecord(ctx->tile_ev[2 * ctx->tile_pairs], s));
      if (use_bf16) {
        if constexpr (sizeof(T) == 4)
          launch_leaf_tiles_bf16(s, nsplit(), linv_b.p, as<float>(xs_p), as<float>(xnorm), as<float>(alpha),
                                 as<float>(leaves_s), as<float>(lnorm), as<float>(pvar), as<float>(pmean),
                                 npad, dp / 4, mp, kp);
      } else {
        launch_leaf_tiles<T>(s, as<T>(linv_p), as<T>(xs_p), as<T>(xnorm), as<T>(alpha), as<T>(leaves_s),
                             as<T>(lnorm), as<T>(pvar), as<T>(pmean), npad, dp / 4, mp, kp);
      }
      HIPCHECK(hipEventRecord(ctx->tile_ev[2 * ctx->tile_pairs + 1], s));
      ++ctx->tile_pairs;
      launch_leaf_finalize<T>(s, as<T>(pvar), as<T>(pmean), nbi, mp, mc, kp, varsigma, mean_dev + off,
                              var_dev + off, want_ucb ? ucb_dev + off : nullptr);
    }
    HIPCHECK(hipGetLastError());
    return GPSO_OK;  // (no host wait here: the kernel time is read after the call's final sync)
  }

  // after the stream has been synchronised: total leaf-tile kernel time of the call
  void collect_tile_ms() {
    float total = 0;
    for (int i = 0; i < ctx->tile_pairs; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ctx->tile_ev[2 * i], ctx->tile_ev[2 * i + 1]) == hipSuccess) total += ms;
    }
    ctx->last_ms[0] = total;
    ctx->tile_pairs = 0;
  }

  int stage_leaves(const void* xs, int xs_dtype, int xs_mem, int64_t m, const void** dev_ptr) {
    if (xs_mem == GPSO_MEM_DEVICE) {
      *dev_ptr = xs;
      return GPSO_OK;
    }
    const size_t bytes = (size_t)m * d * ((xs_dtype == GPSO_F64) ? 8 : 4);
    int rc = ensure(leaves_raw, bytes);
    if (rc) return rc;
    HIPCHECK(hipMemcpyAsync(leaves_raw.p, xs, bytes, hipMemcpyHostToDevice, st()));
    *dev_ptr = leaves_raw.p;
    return GPSO_OK;
  }

  int check_predict_args(const void* xs, int xs_dtype, int xs_mem, int64_t m) {
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident: call gpso_fit_eval / gpso_set_posterior first");
    if (m < 0) return ctx->fail(GPSO_E_ARG, "negative leaf count");
    if (m > 0 && !xs) return ctx->fail(GPSO_E_ARG, "xs must not be NULL");
    if (xs_dtype != GPSO_F64 && xs_dtype != GPSO_F32) return ctx->fail(GPSO_E_ARG, "bad xs_dtype %d", xs_dtype);
    if (xs_mem != GPSO_MEM_HOST && xs_mem != GPSO_MEM_DEVICE) return ctx->fail(GPSO_E_ARG, "bad xs_mem %d", xs_mem);
    return GPSO_OK;
  }

  int predict(const void* xs, int xs_dtype, int xs_mem, int64_t m, double* mean, double* var,
              int out_mem) override {
    int rc = check_predict_args(xs, xs_dtype, xs_mem, m);
    if (rc) return rc;
    if (m == 0) return GPSO_OK;
    if (!mean || !var) return ctx->fail(GPSO_E_ARG, "mean / var must not be NULL");
    hipStream_t s = st();
    HIPCHECK(hipEventRecord(ctx->ev[2], s));
    const void* dev = nullptr;
    if ((rc = stage_leaves(xs, xs_dtype, xs_mem, m, &dev))) return rc;
    double *md = mean, *vd = var;
    if (out_mem == GPSO_MEM_HOST) {
      if ((rc = ensure(omean, (size_t)m * 8))) return rc;
      if ((rc = ensure(ovar, (size_t)m * 8))) return rc;
      md = as<double>(omean);
      vd = as<double>(ovar);
    }
    if ((rc = score_device_leaves(dev, xs_dtype, m, 0.0, false, md, vd, nullptr))) return rc;
    if (out_mem == GPSO_MEM_HOST) {
      HIPCHECK(hipMemcpyAsync(mean, md, (size_t)m * 8, hipMemcpyDeviceToHost, s));
      HIPCHECK(hipMemcpyAsync(var, vd, (size_t)m * 8, hipMemcpyDeviceToHost, s));
    }
    HIPCHECK(hipEventRecord(ctx->ev[3], s));
    HIPCHECK(ctx->wait(s));
    collect_tile_ms();
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ms[1] = ms;
    return GPSO_OK;
  }

  int best_ucb_device(const void* dev, int xs_dtype, int64_t m, const int64_t* seg_off, int nseg,
                      double varsigma, int64_t* idx, double* mean, double* var, double* ucb) {
    int rc;
    hipStream_t s = st();
    if ((rc = ensure(omean, (size_t)m * 8))) return rc;
    if ((rc = ensure(ovar, (size_t)m * 8))) return rc;
    if ((rc = ensure(oucb, (size_t)m * 8))) return rc;
    if ((rc = ensure(segoff, (size_t)(nseg + 1) * 8))) return rc;
    if ((rc = ensure(best, (size_t)nseg * kArgmaxBlocks * kArgmaxPartialBytes))) return rc;
    if ((rc = ensure(ovals, (size_t)nseg * 4 * 8))) return rc;
    std::vector<int64_t> so(nseg + 1);
    if (seg_off) {
      for (int i = 0; i <= nseg; ++i) so[i] = seg_off[i];
      if (so[0] != 0 || so[nseg] != m) return ctx->fail(GPSO_E_ARG, "seg_off must start at 0 and end at M");
      for (int i = 0; i < nseg; ++i)
        if (so[i + 1] < so[i]) return ctx->fail(GPSO_E_ARG, "seg_off must be non-decreasing");
    } else {
      if (nseg != 1) return ctx->fail(GPSO_E_ARG, "seg_off == NULL requires nseg == 1");
      so[0] = 0;
      so[1] = m;
    }
    if (so != segoff_cache) {  // the segmentation rarely changes between calls: upload only then
      HIPCHECK(hipMemcpyAsync(segoff.p, so.data(), (size_t)(nseg + 1) * 8, hipMemcpyHostToDevice, s));
      HIPCHECK(hipStreamSynchronize(s));
      segoff_cache = so;
    }
    if (m > 0)
      if ((rc = score_device_leaves(dev, xs_dtype, m, varsigma, true, as<double>(omean), as<double>(ovar), as<double>(oucb)))) return rc;
    launch_seg_argmax(s, as<double>(omean), as<double>(ovar), as<double>(oucb), as<int64_t>(segoff),
                      nseg, kArgmaxBlocks, best.p, as<double>(ovals));
    double* vals = ctx->pinned_scratch((size_t)nseg * 4);
    if (!vals) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    HIPCHECK(hipMemcpyAsync(vals, ovals.p, (size_t)nseg * 32, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipEventRecord(ctx->ev[3], s));
    HIPCHECK(ctx->wait(s));
    HIPCHECK(hipGetLastError());
    collect_tile_ms();
    for (int i = 0; i < nseg; ++i) {
      if (idx) std::memcpy(&idx[i], &vals[4 * i + 3], 8);
      if (mean) mean[i] = vals[4 * i];
      if (var) var[i] = vals[4 * i + 1];
      if (ucb) ucb[i] = vals[4 * i + 2];
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ms[1] = ms;
    return GPSO_OK;
  }

  int best_ucb(const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off, int nseg,
               double varsigma, int64_t* idx, double* mean, double* var, double* ucb) override {
    int rc = check_predict_args(xs, xs_dtype, xs_mem, m);
    if (rc) return rc;
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    const void* dev = nullptr;
    if (m > 0 && (rc = stage_leaves(xs, xs_dtype, xs_mem, m, &dev))) return rc;
    return best_ucb_device(dev, xs_dtype, m, seg_off, nseg, varsigma, idx, mean, var, ucb);
  }

  // ------------------------------------------------------------------------------------------
  int grow_to_device(const double* bounds, int nseg, int d_, int depth, int64_t* rows_out) {
    if (!bounds) return ctx->fail(GPSO_E_ARG, "bounds must not be NULL");
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    if (d_ < 1 || d_ > kMaxD) return ctx->fail(GPSO_E_ARG, "input dimension %d outside [1, %d]", d_, kMaxD);
    if (depth < 0 || depth > 16) return ctx->fail(GPSO_E_ARG, "depth %d outside [0, 16]", depth);
    const int64_t rows = gpso_grow_rows(depth);
    *rows_out = rows;
    int rc;
    const size_t bb = (size_t)nseg * d_ * 2 * 8;
    const size_t ob = (size_t)nseg * rows * d_ * 8;
    // bounds live behind the generated rows in the same buffer
    if ((rc = ensure(leaves_raw, ob + bb))) return rc;
    double* bdev = reinterpret_cast<double*>(static_cast<char*>(leaves_raw.p) + ob);
    HIPCHECK(hipMemcpyAsync(bdev, bounds, bb, hipMemcpyHostToDevice, st()));
    launch_grow(st(), bdev, nseg, d_, depth, as<double>(leaves_raw));
    HIPCHECK(hipGetLastError());
    return GPSO_OK;
  }

  int grow(const double* bounds, int nseg, int d_, int depth, double* out) override {
    if (!out) return ctx->fail(GPSO_E_ARG, "out must not be NULL");
    int64_t rows = 0;
    int rc = grow_to_device(bounds, nseg, d_, depth, &rows);
    if (rc) return rc;
    if (rows > 0)
      HIPCHECK(hipMemcpyAsync(out, leaves_raw.p, (size_t)nseg * rows * d_ * 8, hipMemcpyDeviceToHost, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    return GPSO_OK;
  }

  int best_ucb_grow(const double* bounds, int nseg, int depth, double varsigma, int64_t* idx,
                    double* mean, double* var, double* ucb) override {
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident: call gpso_fit_eval / gpso_set_posterior first");
    HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    int64_t rows = 0;
    int rc = grow_to_device(bounds, nseg, d, depth, &rows);
    if (rc) return rc;
    std::vector<int64_t> so(nseg + 1);
    for (int i = 0; i <= nseg; ++i) so[i] = (int64_t)i * rows;
    return best_ucb_device(leaves_raw.p, GPSO_F64, (int64_t)nseg * rows, so.data(), nseg, varsigma,
                           idx, mean, var, ucb);
  }

  // ------------------------------------------------------------------------------------------
  int get_matrix(int which, double* out) override {
    if (!out) return ctx->fail(GPSO_E_ARG, "out must not be NULL");
    const T* src = nullptr;
    int lower = 1;
    switch (which) {
      case GPSO_MAT_CHOL:
        if (!chol_valid) return ctx->fail(GPSO_E_STATE, "no factor resident");
        src = as<T>(Lf);
        break;
      case GPSO_MAT_LINV:
        if (!chol_valid) return ctx->fail(GPSO_E_STATE, "no factor resident");
        src = as<T>(linv);
        break;
      case GPSO_MAT_KINV:
        if (!have_kinv) return ctx->fail(GPSO_E_STATE, "Kinv only exists after gpso_fit_eval with grad");
        src = as<T>(kinvb);
        lower = 2;
        break;
      default:
        return ctx->fail(GPSO_E_ARG, "unknown matrix id %d", which);
    }
    double* tmp = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)n * n * 8));
    launch_convert_out<T>(st(), src, npad, tmp, n, n, lower);
    hipError_t e = hipMemcpyAsync(out, tmp, (size_t)n * n * 8, hipMemcpyDeviceToHost, st());
    if (e == hipSuccess) e = hipStreamSynchronize(st());
    (void)hipFree(tmp);
    if (e != hipSuccess) return ctx->fail(GPSO_E_HIP, "get_matrix: %s", hipGetErrorString(e));
    return GPSO_OK;
  }

  int get_vector(int which, double* out) override {
    if (!out) return ctx->fail(GPSO_E_ARG, "out must not be NULL");
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident");
    const T* src = (which == GPSO_VEC_ALPHA) ? as<T>(alpha) : (which == GPSO_VEC_WHITE) ? as<T>(white) : nullptr;
    if (!src) return ctx->fail(GPSO_E_ARG, "unknown vector id %d", which);
    double* tmp = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)n * 8));
    launch_convert_out<T>(st(), src, npad, tmp, 1, n, 0);
    hipError_t e = hipMemcpyAsync(out, tmp, (size_t)n * 8, hipMemcpyDeviceToHost, st());
    if (e == hipSuccess) e = hipStreamSynchronize(st());
    (void)hipFree(tmp);
    if (e != hipSuccess) return ctx->fail(GPSO_E_HIP, "get_vector: %s", hipGetErrorString(e));
    return GPSO_OK;
  }

  int posterior_buffers(void** ptrs, int64_t* nbytes, int cap) override {
    if (npad == 0) return ctx->fail(GPSO_E_STATE, "no problem shape yet");
    if (cap < 5) return ctx->fail(GPSO_E_ARG, "need room for 5 buffers");
    const size_t s = sizeof(T);
    ptrs[0] = hyper.p;  nbytes[0] = (int64_t)(kHyperHeader + kMaxD) * 8;
    ptrs[1] = linv_p.p; nbytes[1] = (int64_t)(npad * npad * s);
    ptrs[2] = xs_p.p;   nbytes[2] = (int64_t)(npad * dp * s);
    ptrs[3] = xnorm.p;  nbytes[3] = (int64_t)(npad * s);
    ptrs[4] = alpha.p;  nbytes[4] = (int64_t)(npad * s);
    if (bf16_usable()) {
      if (cap < 6) return ctx->fail(GPSO_E_ARG, "need room for 6 buffers");
      int rc = ensure(linv_b, (size_t)nsplit() * npad * npad * 2);
      if (rc) return rc;
      ptrs[5] = linv_b.p; nbytes[5] = (int64_t)nsplit() * npad * npad * 2;
      return 6;
    }
    return 5;
  }

  int alloc_posterior(int64_t n_, int d_) override {
    int rc = shape(n_, d_);
    if (rc) return rc;
    have_data = have_post = have_kinv = chol_valid = false;
    return GPSO_OK;
  }

  int adopt_posterior() override {
    if (npad == 0) return ctx->fail(GPSO_E_STATE, "gpso_alloc_posterior first");
    double h[kHyperHeader + kMaxD];
    HIPCHECK(hipMemcpyAsync(h, hyper.p, sizeof(h), hipMemcpyDeviceToHost, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    if ((int64_t)h[0] != n || (int)h[1] != d)
      return ctx->fail(GPSO_E_ARG, "received posterior is for n=%lld d=%d, buffers were sized for n=%lld d=%d",
                       (long long)h[0], (int)h[1], (long long)n, d);
    kp.kernel = (int)h[2];
    n_ls = (int)h[3];
    kp.variance = h[4];
    kp.noise = h[5];
    kp.mean_c = h[6];
    ls_host.assign(h + kHyperHeader, h + kHyperHeader + n_ls);
    have_post = true;
    chol_valid = have_kinv = false;
    linv_b_valid = bf16_usable();  // the bf16 pieces travel with the posterior when the mode is on
    return GPSO_OK;
  }
};

}  // namespace

// ==============================================================================================
extern "C" {

int gpso_create(gpso_ctx** out, int device, int dtype) {
  if (!out) {
    g_create_error = "out must not be NULL";
    return GPSO_E_ARG;
  }
  *out = nullptr;
  if (dtype != GPSO_F64 && dtype != GPSO_F32) {
    g_create_error = "dtype must be GPSO_F64 or GPSO_F32";
    return GPSO_E_ARG;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
    return GPSO_E_HIP;
  }
  if (device < 0 || device >= count) {
    g_create_error = "device index out of range";
    return GPSO_E_ARG;
  }
  if ((e = hipSetDevice(device)) != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return GPSO_E_HIP;
  }
  gpso_ctx* ctx = new gpso_ctx();
  ctx->device = device;
  ctx->dtype = dtype;
  if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
    delete ctx;
    return GPSO_E_HIP;
  }
  ctx->stream = ctx->own_stream;
  if ((e = hipEventCreateWithFlags(&ctx->ev_wait, hipEventDisableTiming)) != hipSuccess) {
    g_create_error = std::string("hipEventCreate: ") + hipGetErrorString(e);
    gpso_destroy(ctx);
    return GPSO_E_HIP;
  }
  for (auto& ev : ctx->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) {
      g_create_error = std::string("hipEventCreate: ") + hipGetErrorString(e);
      gpso_destroy(ctx);
      return GPSO_E_HIP;
    }
  if (dtype == GPSO_F64)
    ctx->eng = new EngineT<double>(ctx);
  else
    ctx->eng = new EngineT<float>(ctx);
  *out = ctx;
  return GPSO_OK;
}

void gpso_destroy(gpso_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  delete ctx->eng;
  for (auto& ev : ctx->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->tile_ev) (void)hipEventDestroy(ev);
  if (ctx->ev_wait) (void)hipEventDestroy(ctx->ev_wait);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* gpso_last_error(const gpso_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

#define ENTER()                                         \
  if (!ctx) return GPSO_E_ARG;                          \
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GPSO_E_HIP, "hipSetDevice failed");

int gpso_set_stream(gpso_ctx* ctx, void* hip_stream) {
  ENTER();
  (void)hipStreamSynchronize(ctx->stream);
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return GPSO_OK;
}

int gpso_synchronize(gpso_ctx* ctx) {
  ENTER();
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return GPSO_OK;
}

int gpso_set_option(gpso_ctx* ctx, int option, int value) {
  ENTER();
  return ctx->eng->set_option(option, value);
}

int gpso_set_data(gpso_ctx* ctx, const double* X, const double* y, int64_t n, int d) {
  ENTER();
  return ctx->eng->set_data(X, y, n, d);
}

int gpso_fit_eval(gpso_ctx* ctx, int kernel, const double* lengthscales, int n_ls, double variance,
                  double noise, double mean_c, double* nlml, double* grad) {
  ENTER();
  return ctx->eng->fit_eval(kernel, lengthscales, n_ls, variance, noise, mean_c, nlml, grad);
}

int gpso_set_posterior(gpso_ctx* ctx, const double* X, const double* L, const double* alpha,
                       int64_t n, int d, int kernel, const double* lengthscales, int n_ls,
                       double variance, double noise, double mean_c) {
  ENTER();
  return ctx->eng->set_posterior(X, L, alpha, n, d, kernel, lengthscales, n_ls, variance, noise, mean_c);
}

int gpso_predict(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m, double* mean,
                 double* var, int out_mem) {
  ENTER();
  if (out_mem != GPSO_MEM_HOST && out_mem != GPSO_MEM_DEVICE) return ctx->fail(GPSO_E_ARG, "bad out_mem %d", out_mem);
  return ctx->eng->predict(xs, xs_dtype, xs_mem, m, mean, var, out_mem);
}

int gpso_best_ucb(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m,
                  const int64_t* seg_off, int nseg, double varsigma, int64_t* idx, double* mean,
                  double* var, double* ucb) {
  ENTER();
  return ctx->eng->best_ucb(xs, xs_dtype, xs_mem, m, seg_off, nseg, varsigma, idx, mean, var, ucb);
}

int64_t gpso_grow_rows(int depth) {
  int64_t rows = 0, w = 1;
  for (int j = 0; j < depth; ++j) {
    rows += w;
    w *= 3;
  }
  return rows;
}

int gpso_grow(gpso_ctx* ctx, const double* bounds, int nseg, int d, int depth, double* out_coords) {
  ENTER();
  return ctx->eng->grow(bounds, nseg, d, depth, out_coords);
}

int gpso_best_ucb_grow(gpso_ctx* ctx, const double* bounds, int nseg, int depth, double varsigma,
                       int64_t* idx, double* mean, double* var, double* ucb) {
  ENTER();
  return ctx->eng->best_ucb_grow(bounds, nseg, depth, varsigma, idx, mean, var, ucb);
}

int64_t gpso_padded_n(const gpso_ctx* ctx) {
  return ctx ? ctx->eng->padded_n() : 0;
}

int gpso_get_matrix(gpso_ctx* ctx, int which, double* out) {
  ENTER();
  return ctx->eng->get_matrix(which, out);
}

int gpso_get_vector(gpso_ctx* ctx, int which, double* out) {
  ENTER();
  return ctx->eng->get_vector(which, out);
}

int gpso_posterior_buffers(gpso_ctx* ctx, void** ptrs, int64_t* nbytes, int cap) {
  ENTER();
  if (!ptrs || !nbytes) return ctx->fail(GPSO_E_ARG, "NULL argument");
  return ctx->eng->posterior_buffers(ptrs, nbytes, cap);
}

int gpso_alloc_posterior(gpso_ctx* ctx, int64_t n, int d) {
  ENTER();
  return ctx->eng->alloc_posterior(n, d);
}

int gpso_adopt_posterior(gpso_ctx* ctx) {
  ENTER();
  return ctx->eng->adopt_posterior();
}

double gpso_last_ms(gpso_ctx* ctx, int what) {
  if (!ctx || what < 0 || what > 2) return -1.0;
  return ctx->last_ms[what];
}

const char* gpso_version(void) { return "gpso-hip 0.1.0 (gfx950)"; }

}  // extern "C"
