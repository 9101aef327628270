// C-ABI of libgpso_hip.so (see include/gpso_hip.h): context, device memory, call sequencing.
// No torch, no BLAS/solver libraries: every kernel launched here is hand-written (predict.hip,
// fit.hip, grow.hip).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gpso_hip.h"
#include "common.hpp"
#include "kernels.hpp"
#include "rccl_api.hpp"

using namespace gpso;

namespace gpso {

namespace {
thread_local std::string g_launch_error;
std::mutex g_lds_mutex;
std::set<std::pair<const void*, int>> g_lds_done;  // (kernel, device) pairs already opted in
}  // namespace

void note_launch_error(const char* msg) {
  if (g_launch_error.empty()) g_launch_error = msg;
}

int ensure_dyn_lds(const void* fn, int bytes) {
  if (bytes <= 64 * 1024) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    note_launch_error("hipGetDevice failed");
    return GPSO_E_HIP;
  }
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  if (g_lds_done.count({fn, dev})) return 0;
  // the attribute belongs to the DEVICE's function object: every device a context lives on opts in
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    note_launch_error((std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e)).c_str());
    return GPSO_E_HIP;
  }
  g_lds_done.insert({fn, dev});
  return 0;
}

}  // namespace gpso

namespace {

thread_local std::string g_create_error;

constexpr int kMaxD = 48;                      // padded input dimension limit (LDS budgets)
constexpr int64_t kLeafChunk = (int64_t)1 << 20;  // leaves processed per pass of the tile kernel
constexpr int kHyperHeader = 8;                // doubles in front of the lengthscales

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool view = false;  // a slice of the posterior arena (EngineT::carve_posterior): not freed, not re-allocated on its own
};

struct Engine {
  virtual ~Engine() {}
  virtual int set_data(const double* X, const double* y, int64_t n, int d) = 0;
  virtual int fit_eval(int kernel, const double* ls, int n_ls, double variance, double noise,
                       double mean_c, double* nlml, double* grad) = 0;
  virtual int set_posterior(const double* X, const double* L, const double* alpha, int64_t n, int d,
                            int kernel, const double* ls, int n_ls, double variance, double noise,
                            double mean_c) = 0;
  virtual int append(const double* Xnew, const double* ynew, int64_t k, double* nlml) = 0;
  virtual int predict(const void* xs, int xs_dtype, int xs_mem, int64_t m, double* mean,
                      double* var, int out_mem) = 0;
  virtual int best_ucb(const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off,
                       int nseg, double varsigma, int64_t* idx, double* mean, double* var,
                       double* ucb) = 0;
  virtual int best_ucb_begin(const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off, int nseg,
                             double varsigma, const double* bounds, int depth) = 0;
  virtual int best_ucb_end(int ticket, int64_t* idx, double* mean, double* var, double* ucb) = 0;
  virtual int grow(const double* bounds, int nseg, int d, int depth, double* out) = 0;
  virtual int best_ucb_grow(const double* bounds, int nseg, int depth, double varsigma,
                            int64_t* idx, double* mean, double* var, double* ucb) = 0;
  virtual int get_matrix(int which, double* out) = 0;
  virtual int get_vector(int which, double* out) = 0;
  virtual int posterior_buffers(void** ptrs, int64_t* nbytes, int cap) = 0;
  virtual int posterior_span(void** ptr, int64_t* offset, int64_t* nbytes) = 0;
  virtual int posterior_span_at(int64_t offset, int64_t nbytes, void** ptr) = 0;
  virtual int posterior_hash(uint64_t* out) = 0;
  virtual int alloc_posterior(int64_t n, int d) = 0;
  virtual int adopt_posterior() = 0;
  virtual int64_t padded_n() const = 0;
  virtual void problem_shape(int64_t* n_out, int* d_out) const = 0;
  virtual int set_option(int option, int value) = 0;
  virtual int set_option_f64(int option, double value) = 0;
  virtual int precision_info(double* out) = 0;
  virtual int broadcast_posterior(int root) = 0;
  virtual int broadcast_posterior_rows(int root) = 0;
  virtual int posterior_dirty_ranges(int64_t* offsets, int64_t* nbytes, int cap) = 0;
  virtual int posterior_mark_synced() = 0;
  virtual int best_ucb_sharded(const void* xs, int xs_dtype, int xs_mem, int64_t m_local, int64_t m_global,
                               const int64_t* seg_off, int nseg, double varsigma, int64_t* idx, double* mean,
                               double* var, double* ucb) = 0;
  virtual int best_ucb_grow_sharded(const double* bounds, int nseg, int depth, double varsigma, int64_t* idx,
                                    double* mean, double* var, double* ucb) = 0;
  // the two halves of the sharded calls for an arbitrary (rank, world), without a communicator
  virtual int shard_winners(int rank, int world, const void* xs, int xs_dtype, int xs_mem, int64_t m_local,
                            int64_t m_global, const int64_t* seg_off, int nseg, double varsigma, double* payload) = 0;
  virtual int shard_winners_grow(int rank, int world, const double* bounds, int nseg, int depth, double varsigma,
                                 double* payload) = 0;
  virtual int fold_winners(const double* gathered, int world, int64_t m_global, const int64_t* seg_off, int nseg,
                           int64_t* idx, double* mean, double* var, double* ucb) = 0;
};

// contiguous share [lo, hi) of m items for `rank` of `world`: global order is preserved across ranks
inline void shard_range(int64_t m, int rank, int world, int64_t* lo, int64_t* hi) {
  const int64_t base = m / world, extra = m % world;
  *lo = rank * base + std::min<int64_t>(rank, extra);
  *hi = *lo + base + (rank < extra ? 1 : 0);
}

}  // namespace

// current-device guard: HIP calls act on the calling thread's current device
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
    if (prev == dev) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct gpso_ctx {
  int device = 0;
  int dtype = GPSO_F64;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};  // [0..1] spare, [2..3] predict-type call, [4..5] fit, [6..7] the collective of a sharded call
  std::vector<hipEvent_t> tile_ev;  // start/stop pairs around the leaf-tile kernel, one per chunk
  int tile_pairs = 0;               // pairs recorded by the call in flight
  double last_ms[4] = {0, 0, 0, 0};
  bool timing = true;  // GPSO_OPT_TIMING: does the call in flight record the event pairs gpso_last_ms reads (two to four HIP calls)?
  int timing_every = 1;  // ... 0: never, 1: every fit / predict-type call, k: every k-th one (the others leave gpso_last_ms alone)
  long timed_calls = 0;
  void tick_timing() { timing = timing_every > 0 && (timed_calls++ % timing_every) == 0; }
  // multi-GPU group (gpso_comm_init): one RCCL communicator per context, collectives on ctx->stream
  ncclComm_t comm = nullptr;
  // gpso_comm_abort (callable from another thread): the communicator is gone -- not to be destroyed again, and not to be
  // handed to a collective again either (need_comm refuses until gpso_comm_destroy + gpso_comm_init)
  std::atomic<bool> comm_aborted{false};
  int rank = 0, world = 1;
  int64_t last_count[3] = {0, 0, 0};  // leaves scored / leaves asked for by the last predict-type call; [2] GPSO_FITMATH_* of the last fit
  std::string err;
  Engine* eng = nullptr;
  hipEvent_t ev_wait = nullptr;  // completion marker of the call in flight
  hipStream_t side_stream = nullptr;  // look-ahead of the two-level fits (kernels.hpp: FitPlanes)
  hipEvent_t ev_col = nullptr, ev_chain = nullptr;
  hipStream_t inv_stream = nullptr;   // the overlapped inverse of the two-level double fit
  hipEvent_t ev_panel = nullptr, ev_inv = nullptr;
  double* pinned = nullptr;      // pinned host scratch for the small result read-backs
  size_t pinned_doubles = 0;
  double* stage = nullptr;       // pinned staging of small host inputs (training data, bounds)
  size_t stage_doubles = 0;
  // gpso_best_ucb_begin / _end: two calls may be in flight; each has a pinned result slot of its own and a completion event
  static constexpr int kSlots = 2;
  static constexpr size_t kSlotDoubles = 4 * 1024 + 4;  // nseg <= 1024 per asynchronous call (+ the completion token)
  double slot_token[kSlots] = {0, 0};
  double* slot_host = nullptr;   // [kSlots][kSlotDoubles], pinned
  hipEvent_t slot_ev[kSlots] = {};
  int slot_nseg[kSlots] = {0, 0};  // > 0: the slot holds a call that has not been ended
  int slot_mode[kSlots] = {0, 0};
  int slot_next = 0;
  int slots_busy() const { return (slot_nseg[0] > 0) + (slot_nseg[1] > 0); }

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
      // (fine-grained -- hipHostMallocCoherent --: kernels write their records and the completion token here WHILE they run,
      // and the host reads them behind a system-scope release; coarse-grained host memory is only coherent at kernel boundaries)
      if (hipHostMalloc(reinterpret_cast<void**>(&pinned), want * 8, hipHostMallocCoherent) != hipSuccess) return nullptr;
      std::memset(pinned, 0, want * 8);  // (no stale completion token)
      pinned_doubles = want;
    }
    return pinned;
  }
  // Completion tokens (round 5): where the last kernel of a call is a single workgroup it writes a sequence number behind
  // its records in pinned host memory and the host spins on THAT word -- no event to record, no driver call per poll.  A
  // token that does not arrive within kSpinMs falls back to a blocking wait (where a failed launch surfaces as before).
  // (process-wide: a pinned buffer handed on by a destroyed context -- ContextPool -- never holds a number a later call waits for)
  static std::atomic<uint64_t>& seq_counter() {
    static std::atomic<uint64_t> c{0};
    return c;
  }
  double next_token() { return (double)(seq_counter().fetch_add(1) + 1); }  // (exact in a double for 2^53 calls)
  hipError_t wait_token(const double* word, double token, hipStream_t s) {
    constexpr double kSpinMs = 20.0;
    const volatile double* w = word;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0;; ++it) {
      if (*w == token) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return hipSuccess;
      }
      if ((it & 1023) == 1023 &&
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > kSpinMs)
        return hipStreamSynchronize(s);
    }
  }

  // staging buffer for host -> device copies that must not force a stream synchronisation.  It is
  // reused by the next call, so wait for the copies of the previous one first (they are long done in
  // practice: every entry point ends with a wait on the stream).
  double* pinned_stage(size_t doubles) {
    if (stream) (void)hipStreamSynchronize(stream);
    if (doubles > stage_doubles) {
      if (stage) (void)hipHostFree(stage);
      stage = nullptr;
      stage_doubles = 0;
      const size_t want = std::max<size_t>(doubles, 4096);
      if (hipHostMalloc(reinterpret_cast<void**>(&stage), want * 8, hipHostMallocDefault) != hipSuccess) return nullptr;
      stage_doubles = want;
    }
    return stage;
  }

  int fail(int code, const char* fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

// Host-side resources of a context -- its stream, its events, its pinned buffers -- cost ~2 ms to create and ~1 ms to
// destroy (tools/context_cost.py), a sixth of a whole 50-evaluation GPSO run whose surrogate lives for 32 ms.  gpso_destroy
// hands them to a small per-device pool and gpso_create takes them from there (at most kKeep sets per device are kept; they
// are released at process exit by the runtime).  Device memory is not pooled.
struct ContextPool {
  struct Set {
    hipStream_t stream = nullptr;
    hipEvent_t ev_wait = nullptr, ev[8] = {};
    double *pinned = nullptr, *stage = nullptr;
    size_t pinned_doubles = 0, stage_doubles = 0;
  };
  static constexpr int kKeep = 4, kDevices = 64;
  std::mutex mu;
  std::vector<Set> free_sets[kDevices];
  static ContextPool& get() {
    static ContextPool* p = new ContextPool();  // (never destroyed: no HIP calls from a static destructor)
    return *p;
  }
  bool take(int device, Set& out) {
    if (device < 0 || device >= kDevices) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (free_sets[device].empty()) return false;
    out = free_sets[device].back();
    free_sets[device].pop_back();
    return true;
  }
  bool give(int device, const Set& s) {
    if (device < 0 || device >= kDevices) return false;
    std::lock_guard<std::mutex> lock(mu);
    if ((int)free_sets[device].size() >= kKeep) return false;
    free_sets[device].push_back(s);
    return true;
  }
};

#define HIPCHECK(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return ctx->fail(e_ == hipErrorOutOfMemory ? GPSO_E_OOM : GPSO_E_HIP, "%s failed: %s", \
                       #call, hipGetErrorString(e_));                                      \
  } while (0)

#define RCCLCHECK(call)                                                                              \
  do {                                                                                               \
    ncclResult_t r_ = (call);                                                                        \
    if (r_ != ncclSuccess)                                                                           \
      return ctx->fail(GPSO_E_RCCL, "%s failed: %s", #call, RcclApi::get().GetErrorString(r_));      \
  } while (0)

namespace {

// TF: fit type (Gram, Cholesky, L^-1, alpha);  TP: predict / apply type.
//   GPSO_F64   -> EngineT<double, double>
//   GPSO_F32   -> EngineT<float, float>
//   GPSO_MIXED -> EngineT<double, float>
template <typename TF, typename TP>
struct EngineT : Engine {
  gpso_ctx* ctx;
  explicit EngineT(gpso_ctx* c) : ctx(c) {
    check = sizeof(TP) == 4;  // float predict arithmetic: self-test on by default
    // default tolerances: a double factor with a float apply measures 1e-6 .. 3e-5 (bf16x3) -> 1e-4; an
    // all-float engine is a 1e-4-class instrument already at noise 1e-3 -> 1e-3
    tol_var = tol_mean = (sizeof(TF) == 4) ? 1.0e-3 : 1.0e-4;
  }
  static constexpr bool kFloatPredict = sizeof(TP) == 4;

  // problem
  int64_t n = 0, npad = 0;
  int d = 0, dp = 0;
  bool have_data = false, have_post = false, have_kinv = false, chol_valid = false;
  KernParams kp{0, 1.0, 1e-3, 0.0};
  int n_ls = 1;
  std::vector<double> ls_host;

  // device buffers.  Fit side: xs64 / xnorm64 (scaled inputs, always double), K, Lf, linv, work, kinvb,
  // white, alpha_f in TF.  Predict side: linv_p, alpha in TP; xs_p64 (+ xnorm64) or xs_p32 / xnorm32
  // in the generation type.
  DevBuf xs_h16, c16_scal;  // fp16 piece pairs of xs32 in the fp16 contraction's fragment order, and their scale (4 floats)
  DevBuf x64, y64, hyper, xs64, xnorm64, xs_p64, xs32, xnorm32, xs_p32, K, Lf, linv, work, kinvb, linv_p,
      white, alpha_f, alpha, logdet, scal, gpart, apart, kinv_diag, getter_tmp;
  // split-bf16 copy of L^-1 (float predict with GPSO_OPT_PREDICT_MATH != native)
  DevBuf linv_b;
  DevBuf pl_L, pl_X, pl_XT, pl_WT;  // bf16 plane sets of the two-level float fit (kernels.hpp: FitPlanes)
  DevBuf app;                       // scratch of gpso_append (kernels.hpp: append_scratch_doubles)
  std::vector<double> x_host, y_host;  // host mirror of the training data (gpso_append's refit path needs all of it)
  int fit_overlap = 2;              // GPSO_OPT_FIT_OVERLAP (bit 0 look-ahead, bit 1 overlapped inverse: the default): two-level double fits overlap chains, updates and the inverse (0: sequential, round 5)
  int fit_planes_mode = 2;          // GPSO_OPT_FIT_BF16_SYRK: 0 f32 MFMA | 1 bf16 pieces (6 MFMAs per product) | 2 fp16 pieces (3) where representable
  // predict math: the OPTION (math_auto: GPSO_MATH_AUTO) and what the resident posterior uses (math, and
  // math_native_fallback when the self-test preferred the f32 MFMA kernel for it)
  // GPSO_MATH_AUTO walks the ladder fp16 split (3 MFMAs per product) -> bf16x6 -> f32 MFMA kernel, one rung down each
  // time the self-test of the posterior at hand fails with the rung it is on (selftest_with_fallback)
  static constexpr int kAutoFirst = kFloatPredict ? GPSO_MATH_F16X3 : GPSO_MATH_NATIVE;
  int math = kAutoFirst;
  bool math_auto = kFloatPredict, math_native_fallback = false;
  // generation of the cross-Gram tile in float-predict contexts: the OPTION (gen_mode) and what the
  // resident posterior actually uses (gen_eff32).  GPSO_GEN_AUTO starts every posterior in float -- the
  // fast form -- and lets the precision self-test decide: if the float form misses the tolerances the
  // posterior moves to double generation and is tested again (decide_generation).
  int gen_mode = GPSO_GEN_AUTO;
  bool gen_eff32 = true;
  bool gen_decided = false;     // AUTO: has the self-test ruled on this posterior?
  bool gen32_inputs_ok = false; // xs32 / xnorm32 / xs_p32 match the resident posterior
  int split_variant = GPSO_SPLIT_KERNEL_AUTO;  // GPSO_OPT_SPLIT_KERNEL
  int contraction = GPSO_CONTRACTION_AUTO;     // GPSO_OPT_CONTRACTION
  // the x.x* contraction of the fp16-split kernel runs on the fp16 pipe under float generation (D_pad + 1 slots in at most
  // two chunks of 32: every D this library accepts).  Measured in one process on one posterior (tools/c16_check.py,
  // profiles/r04_c16_check.jsonl), fp16 pipe against the f32 instruction: D = 6 +4 %, 12 +8 %, 20 +17 %, 33 +28 %, 40 +30 %
  bool c16_fallback = false;  // this posterior's float generation keeps the f32 contraction (decide_generation)
  bool c16_in_use(bool gen64) const {
    if (!kFloatPredict || gen64 || c16_fallback || !f16_split() || contraction == GPSO_CONTRACTION_F32 || leaf_c16_chunks(dp / 4) > 2) return false;
    return leaf_bf16_lds_bytes(2, dp / 4, 4, true) <= 160 * 1024;
  }
  bool small_calls = true, one_launch = true, one_launch_everywhere = false;  // GPSO_OPT_SMALL_CALLS
  bool fuse_prep = true;  // GPSO_OPT_FUSED_PREP: float leaves are scaled in the fp16-contraction kernel's prologue (same bits)
  bool one_refused = false;                    // the one-launch kernel asked for the general sequence (this call only)
  int64_t single_level_max = -1;  // < 0: library default
  bool fused_small = true;        // GPSO_OPT_FIT_FUSED_SMALL
  int small_tile_rows = 8;        // tile rows of the 128-padded linv_p that may be non-zero (8: all / unknown)
  bool linv_b_valid = false;
  bool linv_b_pending = false;  // the split pieces of the resident L^-1 are still to be packed (a fit that returned a gradient)
  std::vector<int64_t> segoff_cache;  // what the device copy of seg_off currently holds
  double* host_direct = nullptr;      // pinned host memory the arg-max of the call in flight writes its records to
  double* result_slot = nullptr;      // asynchronous call being enqueued: its own pinned slot instead of the shared scratch
  double* result_host(size_t doubles) { return result_slot != nullptr ? result_slot : ctx->pinned_scratch(doubles + 1); }  // (+ the token's slot)
  double call_token = 0.0;  // the completion token the LAST kernel of the call in flight writes (0: none -- wait for the event)
  // a fresh token for the kernel about to be launched as the call's last one, or 0 where the call is timed (events needed)
  // or that kernel is not a single workgroup
  double arm_token(bool single_workgroup, int nseg) {
    call_token = (single_workgroup && !ctx->timing && host_direct != nullptr) ? ctx->next_token() : 0.0;
    // (whatever an earlier call left in the token's word -- a record of a call with more segments -- cannot pass for it)
    if (call_token != 0.0) host_direct[4 * nseg + 2] = 0.0;
    return call_token;
  }
  DevBuf extra_cnt;                   // small growth calls: rows appended behind the analytic slots (zero between calls)
  // predict workspace
  DevBuf leaves_raw, leaves_s, lnorm, pvar, pmean, omean, ovar, oucb, segoff, best, oidx, ovals, grow_key,
      live_cnt, best_pos, gath, wbase, ovals2, bhdr;
  // precision self-test
  bool check = false, st_done = false, st_have = false;
  double tol_var = 1.0e-4, tol_mean = 1.0e-4;
  double st_vals[6] = {0, 0, 0, 0, 0, 0};
  DevBuf st_mean, st_var, st_out;

  ~EngineT() override {
    for (DevBuf* b : {&xs_h16, &c16_scal, &x64, &y64, &hyper, &xs64, &xnorm64, &xs_p64, &xs32, &xnorm32, &xs_p32, &K, &Lf, &linv,
                      &work, &kinvb, &linv_p, &white, &alpha_f, &alpha, &logdet, &scal, &gpart, &apart,
                      &kinv_diag, &getter_tmp, &leaves_raw, &leaves_s, &lnorm, &pvar, &pmean, &omean, &ovar,
                      &oucb, &segoff, &best, &oidx, &ovals, &linv_b, &st_mean, &st_var, &st_out, &grow_key, &live_cnt,
                      &best_pos, &gath, &wbase, &ovals2, &bhdr, &pl_L, &pl_X, &pl_XT, &pl_WT, &app, &amax_rows, &arena, &hash_out, &extra_cnt, &one_ctl, &one_partial, &one_ppos})
      if (b->p && !b->view) (void)hipFree(b->p);
  }

  int64_t padded_n() const override { return npad; }
  void problem_shape(int64_t* n_out, int* d_out) const override {
    *n_out = n;
    *d_out = d;
  }

  // generation type actually used by the resident posterior
  bool gen_double() const { return !kFloatPredict || !gen_eff32; }
  // the split-bf16 kernel keeps its leaf fragments in LDS: with double generation they do not fit the
  // 160 KB for D > 24 (bf16x6) / D > 36 (bf16x3) -- those calls run the native f32 kernel instead (accuracy
  // decides the generation type, the kernel follows)
  bool bf16_fits(bool gen64) const {
    return c16_in_use(gen64) || leaf_bf16_lds_bytes(nsplit(), dp / 4, gen64 ? 8 : 4) <= 160 * 1024;
  }
  int nsplit() const { return math == GPSO_MATH_BF16X6 ? 3 : 2; }
  bool f16_split() const { return math == GPSO_MATH_F16X3; }
  // bytes of the split copy of L^-1: the planes and, behind them, one 256-byte slot for the power-of-two scale of the
  // fp16 split (2 floats, written on the device at packing time) -- it travels with the planes in a hand-off.  Under
  // GPSO_MATH_AUTO room for three planes whatever the stage, so that both sides of a hand-off list the same sizes.
  // Round 4: the 256-byte scale slot sits IN FRONT of the planes, so that the part of the buffer a posterior uses
  // (slot + nsplit planes) is a prefix of it -- the posterior arena sends exactly that (posterior_span).
  size_t split_planes_alloc() const { return (size_t)(math_auto ? 3 : nsplit()) * npad * npad * 2; }
  size_t split_bytes() const { return split_planes_alloc() + 256; }
  size_t split_bytes_in_use() const { return (size_t)nsplit() * npad * npad * 2 + 256; }
  float* f16_scale() const { return reinterpret_cast<float*>(linv_b.p); }
  void* split_planes() const { return static_cast<char*>(linv_b.p) + 256; }
  // max |L^-1| of the fp16 split comes out of the fit's own pass over L^-1 (white_kernel leaves row maxima,
  // alpha_sum_kernel folds them into the scale slot): no extra pass, no memset.  f16_handed_at: where this fit's solve
  // was told to leave it (cleared by every packing and at the start of every fit).
  DevBuf amax_rows;
  const void* f16_handed_at = nullptr;
  // -> where the fit's pass leaves the maximum, nullptr when the fp16 split does not apply to this posterior
  float* f16_max_for_fit() {
    f16_handed_at = nullptr;
    if constexpr (!kFloatPredict) return nullptr;
    if (!f16_split() || !bf16_usable()) return nullptr;
    if (ensure(linv_b, split_bytes()) != GPSO_OK || ensure(amax_rows, (size_t)npad * 4) != GPSO_OK) return nullptr;
    f16_handed_at = f16_scale();
    return f16_scale();
  }
  bool bf16_usable() const { return kFloatPredict && math != GPSO_MATH_NATIVE && !math_native_fallback && npad > 0 && npad % 256 == 0; }

  // An evaluation WITH gradient is a step of the hyper-parameter search: the next call is another evaluation, not a
  // prediction, and its 16-bit pieces of L^-1 (10 us at C3, 0.5 ms at C5) would be packed for nothing.  They are packed
  // when something first asks for them: every predict-type call, the self-test and the hand-off paths come through
  // decide_generation(), which calls this.
  int ensure_split_pieces() {
    if (!linv_b_pending) return GPSO_OK;
    linv_b_pending = false;
    return chol_valid ? pack_bf16() : GPSO_OK;
  }
  // (re)build the bf16 pieces of L^-1 from the fit-type L^-1 resident in `linv`
  int pack_bf16() {
    linv_b_valid = false;
    linv_b_pending = false;
    if (!bf16_usable()) return GPSO_OK;
    int rc = ensure(linv_b, split_bytes());
    if (rc) return rc;
    if (f16_split()) {
      const bool have_max = f16_handed_at != nullptr && f16_handed_at == f16_scale();
      launch_pack_linv_f16<TF>(st(), as<TF>(linv), n, npad, f16_scale(), split_planes(), have_max);
      f16_handed_at = nullptr;
    } else {
      launch_pack_linv_bf16<TF>(st(), nsplit(), as<TF>(linv), n, npad, split_planes());
    }
    HIPCHECK(hipGetLastError());
    linv_b_valid = true;
    return GPSO_OK;
  }

  int set_option(int option, int value) override {
    switch (option) {
      case GPSO_OPT_FIT_SINGLE_LEVEL_MAX:
        if (value < 0) return ctx->fail(GPSO_E_ARG, "single-level limit %d must be >= 0", value);
        single_level_max = value;
        return GPSO_OK;
      case GPSO_OPT_FIT_BF16_SYRK:
        if (value < 0 || value > 2) return ctx->fail(GPSO_E_ARG, "fit plane mode must be 0 (f32 MFMA), 1 (bf16 pieces) or 2 (fp16 pieces)");
        fit_planes_mode = value;
        return GPSO_OK;
      case GPSO_OPT_ROW_LOOP:
        if (value < 0) return ctx->fail(GPSO_E_ARG, "row loop: 0, 1 or a split count >= 2");
        gpso::g_leaf_row_loop = (int)value;  // (process-wide: a property of the kernels' launch, not of a posterior)
        return GPSO_OK;
      case GPSO_OPT_FIT_OVERLAP:
        if (value < 0) return ctx->fail(GPSO_E_ARG, "fit overlap must be >= 0");
        fit_overlap = value;
        return GPSO_OK;
      case GPSO_OPT_TIMING:
        if (value < 0 || value > 1000000) return ctx->fail(GPSO_E_ARG, "timing must be 0 (off), 1 (every call) or k (every k-th call)");
        ctx->timing_every = value;
        ctx->timed_calls = 0;  // (the next fit / predict-type call is a sampled one)
        ctx->timing = value != 0;
        if (!ctx->timing) ctx->last_ms[0] = ctx->last_ms[1] = ctx->last_ms[2] = ctx->last_ms[3] = 0.0;
        return GPSO_OK;
      case GPSO_OPT_FIT_FUSED_SMALL:
        if (value != 0 && value != 1) return ctx->fail(GPSO_E_ARG, "fused small fit must be 0 or 1");
        fused_small = value != 0;
        return GPSO_OK;
      case GPSO_OPT_SMALL_CALLS:
        if (value < 0 || value > 3) return ctx->fail(GPSO_E_ARG, "small calls must be 0 .. 3");
        small_calls = value != 0;
        one_launch = value == 1 || value == 3;
        one_launch_everywhere = value == 3;
        return GPSO_OK;
      case GPSO_OPT_FUSED_PREP:
        if (value != 0 && value != 1) return ctx->fail(GPSO_E_ARG, "fused prep must be 0 or 1");
        fuse_prep = value != 0;
        return GPSO_OK;
      case GPSO_OPT_CONTRACTION:
        if (value != GPSO_CONTRACTION_AUTO && value != GPSO_CONTRACTION_F32 && value != GPSO_CONTRACTION_F16) return ctx->fail(GPSO_E_ARG, "unknown contraction %d", value);
        if (value != contraction) {
          contraction = value;
          gen_decided = false;  // (other bits: the self-test rules again)
          st_done = false;
        }
        return GPSO_OK;
      case GPSO_OPT_SPLIT_KERNEL:
        if (value < GPSO_SPLIT_KERNEL_AUTO || value > GPSO_SPLIT_KERNEL_FUSED32) return ctx->fail(GPSO_E_ARG, "unknown split kernel %d", value);
        if (value == GPSO_SPLIT_KERNEL_FUSED32 && !gpso::leaf_step32_built())
          return ctx->fail(GPSO_E_ARG, "the 32x32x16 step is not in this build (measured slower: leaf_split.hpp, GPSO_STEP32)");
        split_variant = value;
        return GPSO_OK;
      case GPSO_OPT_PRECISION_CHECK:
        if (value != 0 && value != 1) return ctx->fail(GPSO_E_ARG, "precision check must be 0 or 1");
        check = value != 0;
        return GPSO_OK;
      case GPSO_OPT_GENERATION:
        if (value != GPSO_GEN_F64 && value != GPSO_GEN_F32 && value != GPSO_GEN_AUTO)
          return ctx->fail(GPSO_E_ARG, "unknown generation mode %d", value);
        if (!kFloatPredict) {
          if (value == GPSO_GEN_F32) return ctx->fail(GPSO_E_ARG, "GPSO_GEN_F32 needs a float-predict context");
          return GPSO_OK;
        }
        if (value != gen_mode) {
          if (have_post && !have_data && value != GPSO_GEN_F64)
            return ctx->fail(GPSO_E_STATE, "set GPSO_OPT_GENERATION before installing a posterior");
          gen_mode = value;
          reset_generation();
        }
        return GPSO_OK;
      case GPSO_OPT_PREDICT_MATH:
        break;
      default:
        return ctx->fail(GPSO_E_ARG, "unknown option %d", option);
    }
    if (value != GPSO_MATH_NATIVE && value != GPSO_MATH_BF16X3 && value != GPSO_MATH_BF16X6 && value != GPSO_MATH_F16X3 &&
        value != GPSO_MATH_AUTO)
      return ctx->fail(GPSO_E_ARG, "unknown predict math %d", value);
    if (value != GPSO_MATH_NATIVE && value != GPSO_MATH_AUTO && !kFloatPredict)
      return ctx->fail(GPSO_E_ARG, "split (bf16 / fp16) predict math needs a GPSO_F32 or GPSO_MIXED context");
    const bool want_auto = value == GPSO_MATH_AUTO;
    if (want_auto) value = kAutoFirst;
    if (value == math && want_auto == math_auto && !math_native_fallback) return GPSO_OK;
    math = value;
    math_auto = want_auto;
    math_native_fallback = false;
    linv_b_valid = false;
    st_done = false;
    reset_generation();
    if (chol_valid) {  // L^-1 is resident: make the new mode usable right away
      int rc = pack_bf16();
      if (rc) return rc;
    }
    return GPSO_OK;
  }

  int set_option_f64(int option, double value) override {
    if (!(value > 0.0)) return ctx->fail(GPSO_E_ARG, "tolerance %g must be positive", value);
    if (option == GPSO_OPTF_TOL_VAR) tol_var = value;
    else if (option == GPSO_OPTF_TOL_MEAN) tol_mean = value;
    else return ctx->fail(GPSO_E_ARG, "unknown floating-point option %d", option);
    return GPSO_OK;
  }

  hipStream_t st() const { return ctx->stream; }
  // calls that replace or change the posterior are refused while an asynchronous best-UCB call is open: its kernels
  // may still be reading what the call would overwrite
  int refuse_if_async(const char* what) {
    if (ctx->slots_busy() == 0) return GPSO_OK;
    return ctx->fail(GPSO_E_STATE, "%s while %d asynchronous best-UCB call(s) are open: gpso_best_ucb_end them first", what, ctx->slots_busy());
  }
  double* ls_dev() const { return static_cast<double*>(hyper.p) + kHyperHeader; }
  template <typename U>
  U* as(const DevBuf& b) const { return static_cast<U*>(b.p); }

  int ensure(DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return GPSO_OK;
    if (b.view) return ctx->fail(GPSO_E_STATE, "internal: a slice of the posterior arena (%zu bytes) was asked to hold %zu", b.bytes, bytes);
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

  // like ensure(), but the old contents survive a re-allocation
  int ensure_keep(DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return GPSO_OK;
    void* np_ = nullptr;
    HIPCHECK(hipMalloc(&np_, bytes));
    if (b.p) {
      HIPCHECK(hipMemcpyAsync(np_, b.p, b.bytes, hipMemcpyDeviceToDevice, st()));
      HIPCHECK(hipStreamSynchronize(st()));
      HIPCHECK(hipFree(b.p));
    }
    b.p = np_;
    b.bytes = bytes;
    return GPSO_OK;
  }

  // launcher-side errors (hipFuncSetAttribute, unsupported GEMM shapes) + the HIP launch status
  int launch_status() {
    HIPCHECK(hipGetLastError());
    if (!gpso::g_launch_error.empty()) {
      const std::string msg = gpso::g_launch_error;
      gpso::g_launch_error.clear();
      return ctx->fail(GPSO_E_HIP, "%s", msg.c_str());
    }
    return GPSO_OK;
  }

  int shape(int64_t n_, int d_) {
    if (n_ < 1) return ctx->fail(GPSO_E_ARG, "need at least one training point (n=%lld)", (long long)n_);
    if (d_ < 1 || d_ > kMaxD) return ctx->fail(GPSO_E_ARG, "input dimension %d outside [1, %d]", d_, kMaxD);
    if (n_ > 65536) return ctx->fail(GPSO_E_ARG, "n=%lld above the supported 65536", (long long)n_);
    n = n_;
    d = d_;
    // Float-predict contexts pad N > 128 to a multiple of 256, the row block of the split predict kernels: with a pad of
    // 128 every N in (256 k + 128, 256 k + 256] ran them and every N in (256 k, 256 k + 128] fell to the f32 MFMA kernel
    // (3.2x the time per flop) -- half of all N.  The padding rows are identity rows of the factor: the blocked
    // factorisation does the same operations on the real rows, up to two 64-column steps more on the padding.
    const int64_t pad = (kFloatPredict && n > kPadN) ? 2 * kPadN : kPadN;
    npad = (n + pad - 1) / pad * pad;
    dp = (d + 3) / 4 * 4;
    int rc;
    // (room for the padded size: gpso_append adds rows in place)
    if ((rc = ensure(x64, (size_t)npad * d * 8))) return rc;
    if ((rc = ensure(y64, (size_t)npad * 8))) return rc;
    if ((rc = carve_posterior())) return rc;
    if (kFloatPredict) {
      if ((rc = ensure(xs32, (size_t)npad * dp * 4))) return rc;
      if ((rc = ensure(xnorm32, (size_t)npad * 4))) return rc;
      if ((rc = ensure(xs_p32, (size_t)npad * dp * 4))) return rc;
      if ((rc = ensure(xs_h16, (size_t)(npad / 16) * leaf_c16_chunks(dp / 4) * 2048))) return rc;
      if ((rc = ensure(c16_scal, 16))) return rc;
    }
    return GPSO_OK;
  }

  // ---- the posterior arena (round 4) ----------------------------------------------------------------------------
  // Everything a peer needs to predict lives in ONE allocation, laid out as
  //     [ packed L^-1 | hyper block | X / l | its MFMA fragments | norms | alpha | split pieces of L^-1 (scale slot, planes) ]
  // (each slice 256-byte aligned), so that what a posterior actually uses is one contiguous range whichever predict
  // math it runs -- native: [packed L^-1 .. alpha], split: [hyper .. the planes in use] -- and gpso_broadcast_posterior
  // moves it with ONE ncclBroadcast.  The layout is a function of (N_pad, D_pad, predict type) only: every rank of a
  // group carves the same offsets.  The slices are DevBuf views of the arena.
  DevBuf arena;
  int64_t arena_npad = -1;
  int arena_dp = -1;
  static size_t up256(size_t b) { return (b + 255) / 256 * 256; }
  bool split_slot_exists() const { return kFloatPredict && npad % 256 == 0; }
  int carve_posterior() {
    if (arena.p != nullptr && arena_npad == npad && arena_dp == dp) return GPSO_OK;
    const size_t b_linv = up256(packed_linv_elems(npad) * sizeof(TP));
    const size_t b_hyper = up256((size_t)(kHyperHeader + kMaxD) * 8);
    const size_t b_xs = up256((size_t)npad * dp * 8);
    const size_t b_norm = up256((size_t)npad * 8);
    const size_t b_alpha = up256((size_t)npad * sizeof(TP));
    // room for three planes whatever the option: GPSO_OPT_PREDICT_MATH may change after the shape is known
    const size_t b_split = split_slot_exists() ? up256((size_t)3 * npad * npad * 2 + 256) : 0;
    const size_t total = b_linv + b_hyper + 2 * b_xs + b_norm + b_alpha + b_split;
    if (arena.bytes < total) {
      if (arena.p) {
        HIPCHECK(hipStreamSynchronize(st()));
        HIPCHECK(hipFree(arena.p));
        arena.p = nullptr;
        arena.bytes = 0;
        arena_npad = -1;
        for (DevBuf* b : {&linv_p, &hyper, &xs64, &xs_p64, &xnorm64, &alpha, &linv_b}) *b = DevBuf{};
      }
      HIPCHECK(hipMalloc(&arena.p, total));
      arena.bytes = total;
    }
    char* q = static_cast<char*>(arena.p);
    auto slice = [&](DevBuf& b, size_t bytes) {
      b.p = bytes ? q : nullptr;
      b.bytes = bytes;
      b.view = true;
      q += bytes;
    };
    slice(linv_p, b_linv);
    slice(hyper, b_hyper);
    slice(xs64, b_xs);
    slice(xs_p64, b_xs);
    slice(xnorm64, b_norm);
    slice(alpha, b_alpha);
    slice(linv_b, b_split);
    arena_npad = npad;
    arena_dp = dp;
    small_tile_rows = 8;  // the slices moved: whatever linv_p's bytes held belongs to another layout
    linv_b_valid = false;
    linv_p_valid = false;
    return GPSO_OK;
  }
  // the contiguous range of the arena the resident posterior uses: offset into the arena and length
  void posterior_range(size_t* off, size_t* bytes) const {
    const char* base = static_cast<const char*>(arena.p);
    if (math_in_use() != GPSO_MATH_NATIVE) {
      *off = (size_t)(static_cast<const char*>(hyper.p) - base);
      *bytes = (size_t)(static_cast<const char*>(linv_b.p) - static_cast<const char*>(hyper.p)) + split_bytes_in_use();
    } else {
      *off = 0;
      *bytes = (size_t)(static_cast<const char*>(alpha.p) - base) + alpha.bytes;
    }
  }
  bool linv_p_valid = false;  // the packed f32 / f64 L^-1 of the resident posterior is here (a receiver of a split posterior: no)

  // ---- what the peers of a group hold of THIS posterior (round 6: gpso_broadcast_posterior_rows) --------------------------------
  // A hand-off (gpso_broadcast_posterior, or a span copy followed by gpso_posterior_mark_synced / gpso_adopt_posterior)
  // records the rows the other side then holds; gpso_append extends the posterior in place and leaves the record alone, so
  // the NEXT hand-off may move only what the appends wrote -- the new rows of the scaled inputs and of each predict-ready
  // copy of L^-1, alpha, the hyper block -- instead of the whole range (C5: ~1.1 MB instead of 1.07 GB for 7 points).
  // Anything that makes another posterior (gpso_set_data, gpso_fit_eval, gpso_set_posterior, a new shape) clears it.
  int64_t sync_n = -1;      // rows the peers hold; -1: unknown
  int sync_math = -1;       // predict math in use then (the ladder may move: other pieces)
  float sync_scale = 0.0f;  // fp16 split: the scale the peers' planes were packed with (an append that crosses a power of two repacks all rows)
  void forget_peers() { sync_n = -1; }
  // the scale slot's [1] (2^-sa) of the fp16 split as the device holds it now; 0 where it does not apply
  int current_split_scale(float* out) {
    *out = 0.0f;
    if (!(kFloatPredict && f16_split() && bf16_usable() && linv_b_valid)) return GPSO_OK;
    float* h = reinterpret_cast<float*>(ctx->pinned_scratch(256) + 124);  // (a slot of the scratch nothing else reads)
    HIPCHECK(hipMemcpyAsync(h, f16_scale() + 1, 4, hipMemcpyDeviceToHost, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    *out = *h;
    return GPSO_OK;
  }
  // ranges of the arena (offset, bytes) that differ between the posterior of the first n0 rows and this one of n rows at the
  // same hyper-parameters, predict math and fp16 scale -- the same list on every rank (a function of n0, n, the layout)
  void rows_ranges(int64_t n0, std::vector<std::pair<int64_t, int64_t>>& out) const {
    out.clear();
    const char* base = static_cast<const char*>(arena.p);
    auto rel = [&](const DevBuf& b, size_t off, size_t bytes) {
      if (bytes) out.push_back({(int64_t)(static_cast<const char*>(b.p) - base + off), (int64_t)bytes});
    };
    const int64_t rt0 = n0 / 16, rt1 = (n - 1) / 16, npad16 = npad / 16;
    rel(hyper, 0, (size_t)(kHyperHeader + kMaxD) * 8);
    rel(xs64, (size_t)n0 * dp * 8, (size_t)(n - n0) * dp * 8);
    rel(xs_p64, (size_t)rt0 * dp * 16 * 8, (size_t)(rt1 + 1 - rt0) * dp * 16 * 8);  // MFMA fragments: 16-row groups
    rel(xnorm64, (size_t)n0 * 8, (size_t)(n - n0) * 8);
    rel(alpha, 0, (size_t)npad * sizeof(TP));  // alpha_1 += R^T a_2: every entry moved
    if (math_in_use() != GPSO_MATH_NATIVE) {
      rel(linv_b, 0, 256);  // the scale slot
      for (int s_ = 0; s_ < nsplit(); ++s_)  // plane s: tile row rt at ((s npad16 + rt) npad 32) bytes behind the slot
        rel(linv_b, 256 + ((size_t)s_ * npad16 + rt0) * npad * 32, (size_t)(rt1 + 1 - rt0) * npad * 32);
    } else {
      const size_t t0 = (size_t)rt0 * (rt0 + 1) / 2, t1 = (size_t)(rt1 + 1) * (rt1 + 2) / 2;  // tiles row-major over the triangle
      rel(linv_p, t0 * 256 * sizeof(TP), (t1 - t0) * 256 * sizeof(TP));
    }
  }
  // can the peers be brought up to date by rows?  (root side; scale = current_split_scale)
  bool rows_apply(float scale) const {
    return have_post && sync_n >= 0 && sync_n <= n && sync_math == math_in_use() && scale == sync_scale &&
           (sync_n == n || sync_n / 16 <= (n - 1) / 16);
  }

  int ensure_fit_buffers() {
    int rc;
    const size_t s = sizeof(TF);
    if ((rc = ensure(K, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(Lf, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(linv, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(work, (size_t)npad * npad * s))) return rc;
    if ((rc = ensure(white, (size_t)npad * s))) return rc;
    if ((rc = ensure(alpha_f, (size_t)npad * s))) return rc;
    if ((rc = ensure(logdet, (size_t)npad * 8))) return rc;
    if ((rc = ensure(kinv_diag, (size_t)npad * 8))) return rc;
    if ((rc = ensure(apart, alpha_part_doubles(npad) * 8))) return rc;
    if ((rc = ensure(scal, (size_t)(8 + kGradMaxLs + 3) * 8))) return rc;
    const size_t nt = (size_t)(npad / 64);
    if ((rc = ensure(gpart, nt * nt * (size_t)(kGradMaxLs + 2) * 8))) return rc;
    return GPSO_OK;
  }

  // upload = false: the caller's kernel writes the device copy of the block itself (fused small fit)
  int set_theta(int kernel, const double* ls, int n_ls_, double variance, double noise, double mean_c,
                bool upload = true) {
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
    if (!upload) return GPSO_OK;
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

  // X / lengthscale for the fit (double) and, packed as MFMA fragments, for the predict generation
  int scale_inputs() {
    launch_scale_x<double>(st(), as<double>(x64), n, npad, d, dp, ls_dev(), as<double>(xs64), as<double>(xnorm64),
                           as<double>(xs_p64));
    return GPSO_OK;
  }
  // a new posterior (or new options): generation starts over -- float unless double was asked for -- and the
  // self-test has to rule again
  void reset_generation() {
    f16_handed_at = nullptr;  // (a fit that failed between its solve and its packing leaves nothing behind)
    math_native_fallback = false;
    if (math_auto) math = kAutoFirst;  // the ladder starts over with every posterior
    gen_eff32 = kFloatPredict && gen_mode != GPSO_GEN_F64;
    gen_decided = false;
    gen32_inputs_ok = false;
    c16_fallback = false;
    st_done = false;
  }
  // float copies of the scaled inputs, made when a float-generation predict first needs them, from the
  // double ones (resident after a fit AND after a hand-off): two short launches, not two per loss evaluation
  int ensure_generation_inputs() {
    if (gen_double() || gen32_inputs_ok) return GPSO_OK;
    launch_gen_inputs_f32(st(), as<double>(xs64), npad, dp, as<float>(xs32), as<float>(xnorm32), as<float>(xs_p32));
    launch_gen_inputs_f16(st(), as<float>(xs32), as<float>(xnorm32), npad, dp, as<float>(c16_scal), xs_h16.p);
    gen32_inputs_ok = true;
    return GPSO_OK;
  }

  // ------------------------------------------------------------------------------------------
  int set_data(const double* X, const double* y, int64_t n_, int d_) override {
    if (!X || !y) return ctx->fail(GPSO_E_ARG, "X / y must not be NULL");
    int rc = refuse_if_async("gpso_set_data");
    if (rc) return rc;
    rc = shape(n_, d_);
    if (rc) return rc;
    // small problems (the optimiser loop: N <= a few hundred) are staged through pinned memory so that
    // the call needs no stream synchronisation; large ones copy straight from the caller's pages
    const size_t doubles = (size_t)n * d + (size_t)n;
    double* stage = doubles <= (1u << 17) ? ctx->pinned_stage(doubles) : nullptr;
    if (stage) {
      std::memcpy(stage, X, (size_t)n * d * 8);
      std::memcpy(stage + (size_t)n * d, y, (size_t)n * 8);
      HIPCHECK(hipMemcpyAsync(x64.p, stage, (size_t)n * d * 8, hipMemcpyHostToDevice, st()));
      HIPCHECK(hipMemcpyAsync(y64.p, stage + (size_t)n * d, (size_t)n * 8, hipMemcpyHostToDevice, st()));
    } else {
      HIPCHECK(hipMemcpyAsync(x64.p, X, (size_t)n * d * 8, hipMemcpyHostToDevice, st()));
      HIPCHECK(hipMemcpyAsync(y64.p, y, (size_t)n * 8, hipMemcpyHostToDevice, st()));
      HIPCHECK(hipStreamSynchronize(st()));
    }
    if (X != x_host.data()) x_host.assign(X, X + (size_t)n * d);
    if (y != y_host.data()) y_host.assign(y, y + (size_t)n);
    have_data = true;
    have_post = have_kinv = chol_valid = linv_p_valid = false;
    st_done = st_have = false;
    forget_peers();
    return GPSO_OK;
  }

  int fit_eval(int kernel, const double* ls, int n_ls_, double variance, double noise,
               double mean_c, double* nlml, double* grad) override {
    ctx->tick_timing();
    if (!have_data) return ctx->fail(GPSO_E_STATE, "gpso_fit_eval before gpso_set_data");
    if (!ls) return ctx->fail(GPSO_E_ARG, "lengthscales must not be NULL");
    int rc = refuse_if_async("gpso_fit_eval");
    if (rc) return rc;
    rc = ensure_fit_buffers();
    if (rc) return rc;
    const bool small = fused_small && small_fit_eligible(n, dp);
    if ((rc = set_theta(kernel, ls, n_ls_, variance, noise, mean_c, !small))) return rc;
    have_post = have_kinv = chol_valid = false;
    st_done = st_have = false;
    forget_peers();
    reset_generation();
    hipStream_t s = st();
    double fit_token = 0.0;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[4], s));
    if (grad && (rc = ensure(kinvb, (size_t)npad * npad * sizeof(TF)))) return rc;
    ctx->last_count[2] = small ? GPSO_FITMATH_SMALL : (sizeof(TF) == 8 ? GPSO_FITMATH_F64 : GPSO_FITMATH_F32);
    if (small) {
      // N <= 128: the whole evaluation in ONE launch (fit.hip: small_fit_kernel)
      SmallFitArgs a{};
      a.x64 = as<double>(x64); a.y64 = as<double>(y64); a.hyper = as<double>(hyper);
      for (int k = 0; k < kMaxD; ++k) a.ls[k] = ls[n_ls_ == 1 ? 0 : std::min(k, n_ls_ - 1)];
      a.n = (int)n; a.d = d; a.dp = dp; a.kernel = kernel; a.n_ls = n_ls; a.want_grad = grad ? 1 : 0;
      a.zero_tile_rows = small_tile_rows;  // (tile rows of linv_p beyond this fit's that may hold old data)
      small_tile_rows = (int)((n + 15) / 16);
      a.variance = variance; a.noise = noise; a.mean_c = mean_c;
      a.xs64 = as<double>(xs64); a.xnorm64 = as<double>(xnorm64); a.xs_p64 = as<double>(xs_p64);
      a.Lf = Lf.p; a.linv = linv.p; a.kinv = grad ? kinvb.p : nullptr;
      a.white = white.p; a.alpha_f = alpha_f.p; a.alpha_p = alpha.p; a.linv_p = linv_p.p;
      a.diag64 = as<double>(logdet); a.kinv_diag = as<double>(kinv_diag); a.scal = as<double>(scal);
      // the loss, the factorisation's verdict and the gradient land in pinned host memory straight from the kernel
      a.scal_host = ctx->pinned_scratch(8 + kGradMaxLs + 3);
      if (!a.scal_host) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
      fit_token = ctx->timing ? 0.0 : ctx->next_token();  // (slot 7 of the scalars: unused by the layout)
      a.scal_host[7] = 0.0;
      a.done_token = fit_token;
      if ((rc = launch_small_fit<TF, TP>(s, a))) return launch_status();
    } else {
      if ((rc = scale_inputs())) return rc;
      int* info_dev = reinterpret_cast<int*>(as<double>(scal) + 1);
      launch_gram<TF>(s, as<double>(xs64), as<double>(xnorm64), n, npad, dp, kp, as<TF>(K), info_dev);
      // (the single-level factorisation writes all of L^-1 that is ever read: no 4 N_pad^2-byte zero fill -- 10 us at C3)
      if (!potrf_is_single_level<TF>(npad, single_level_max))
        HIPCHECK(hipMemsetAsync(linv.p, 0, (size_t)npad * npad * sizeof(TF), s));
      FitPlanes planes{};
      const FitPlanes* pl = nullptr;
      if constexpr (sizeof(TF) == 4) {
        if (fit_planes_mode != 0 && npad > (single_level_max >= 0 ? single_level_max : 3584)) {
          for (DevBuf* b : {&pl_L, &pl_X, &pl_XT, &pl_WT})
            if ((rc = ensure(*b, fit_plane_set_bytes(npad)))) return rc;
          planes = FitPlanes{static_cast<unsigned short*>(pl_L.p), static_cast<unsigned short*>(pl_X.p),
                             static_cast<unsigned short*>(pl_XT.p), static_cast<unsigned short*>(pl_WT.p),
                             (int64_t)npad * npad, (int)(npad / 32)};
          if (ctx->side_stream == nullptr) {
            HIPCHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            HIPCHECK(hipEventCreateWithFlags(&ctx->ev_col, hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming));
          }
          planes.side = ctx->side_stream;
          planes.ev_col = ctx->ev_col;
          planes.ev_chain = ctx->ev_chain;
          // fp16 pieces (three MFMAs per product) when the hyper-parameters leave every plane set inside fp16's range
          // after its power-of-two scaling; bf16 pieces (six) otherwise -- the fallback rung
          if (fit_planes_mode == 2) (void)fit_plane_scales(variance, noise, planes);
          pl = &planes;
          ctx->last_count[2] = planes.np == 2 ? GPSO_FITMATH_F16X3 : GPSO_FITMATH_BF16X6;
        }
      } else {
        // double fits above the single-level limit (round 6): the diagonal blocks' chains are looked ahead on a side stream
        // and the level-doubling inverse runs on a third one, pair by pair, as soon as the panels it reads are final
        // (fit.hip: launch_potrf) -- GPSO_OPT_FIT_OVERLAP = 0 keeps round 5's sequential schedule (same bits either way)
        if (fit_overlap && !potrf_is_single_level<TF>(npad, single_level_max)) {
          if (ctx->side_stream == nullptr) {
            int lo_p = 0, hi_p = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);  // (hi_p = the numerically smallest = highest priority)
            HIPCHECK(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, hi_p));
            HIPCHECK(hipEventCreateWithFlags(&ctx->ev_col, hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming));
          }
          if (ctx->inv_stream == nullptr) {
            HIPCHECK(hipStreamCreateWithFlags(&ctx->inv_stream, hipStreamNonBlocking));
            HIPCHECK(hipEventCreateWithFlags(&ctx->ev_panel, hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&ctx->ev_inv, hipEventDisableTiming));
          }
          planes.side = ctx->side_stream;
          planes.ev_col = ctx->ev_col;
          planes.ev_chain = ctx->ev_chain;
          planes.inv = ctx->inv_stream;
          planes.ev_panel = ctx->ev_panel;
          planes.ev_inv = ctx->ev_inv;
          planes.overlap = fit_overlap;
          pl = &planes;
        }
      }
      const int done = launch_potrf<TF>(s, as<TF>(K), as<TF>(Lf), as<TF>(linv), as<TF>(work),
                                        grad ? as<TF>(kinvb) : nullptr, n, npad, as<double>(logdet), info_dev,
                                        single_level_max, pl);
      bool inv_done = (done & 1) != 0, xt_ready = false;
      // (the single-level path leaves the padding blocks of L^-1 unwritten; the K^-1 = L^-T L^-1 GEMM of launch_gradient
      // would read them -- that path always builds K^-1 beside the factorisation: hold it to that)
      if (grad && potrf_is_single_level<TF>(npad, single_level_max) && (done & 2) == 0)
        return ctx->fail(GPSO_E_STATE, "internal: the single-level factorisation did not leave K^-1 for the gradient");
      if constexpr (sizeof(TF) == 4) {
        if (!inv_done && pl != nullptr && trtri_bf16_applies(npad, fit_outer_panel(npad))) {
          launch_trtri_bf16(s, as<float>(linv), planes, npad, fit_outer_panel(npad), grad);
          inv_done = true;
          xt_ready = true;
        }
      }
      if (!inv_done) launch_trtri<TF>(s, as<TF>(Lf), as<TF>(linv), as<TF>(work), npad, fit_outer_panel(npad));
      float* linv_max_out = f16_max_for_fit();  // (may allocate amax_rows: before the argument list below)
      launch_solve_alpha<TF>(s, as<TF>(linv), as<double>(y64), n, npad, mean_c, as<double>(logdet),
                             as<TF>(white), as<TF>(alpha_f), as<double>(apart), as<double>(kinv_diag),
                             as<double>(scal), alpha.p, sizeof(TP) == 8, as<float>(amax_rows), linv_max_out);
      if (grad)
        launch_gradient<TF>(s, as<TF>(linv), as<TF>(alpha_f), as<double>(xs64), as<double>(xnorm64), n, npad, d, dp,
                            n_ls, ls_dev(), kp, as<TF>(kinvb), (done & 2) != 0, as<double>(gpart),
                            as<double>(scal) + 8, xt_ready ? &planes : nullptr);
      // the packed f32 / f64 copy of L^-1 feeds the NATIVE tile kernel only: a posterior that is going to predict with
      // split math (the default of float-predict contexts) packs it when -- if ever -- something asks for it
      // (ensure_linv_p: the self-test walking down GPSO_MATH_AUTO's ladder, a hand-off of all buffers)
      linv_p_lazy = bf16_usable();
      if (!linv_p_lazy) launch_pack_linv<TF, TP>(s, as<TF>(linv), n, npad, as<TP>(linv_p));  // (alpha's predict-type copy: alpha_sum_kernel)
      small_tile_rows = 8;
    }
    if (grad && !small) {  // (see ensure_split_pieces)
      linv_b_valid = false;
      linv_b_pending = bf16_usable();
    } else if ((rc = pack_bf16())) {
      return rc;
    }
    if ((rc = launch_status())) return rc;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[5], s));
    constexpr size_t kHostDoubles = 8 + kGradMaxLs + 3;
    double* host = ctx->pinned_scratch(kHostDoubles);
    if (!host) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    if (!small) HIPCHECK(hipMemcpyAsync(host, scal.p, kHostDoubles * 8, hipMemcpyDeviceToHost, s));
    if (small && fit_token != 0.0 && !ctx->timing) HIPCHECK(ctx->wait_token(&host[7], fit_token, s));
    else HIPCHECK(ctx->wait(s));
    float ms = 0;
    if (ctx->timing && hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->last_ms[2] = ms;
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
    have_post = chol_valid = st_have = true;
    linv_p_valid = small || !linv_p_lazy;
    if (small) linv_p_lazy = false;
    return GPSO_OK;
  }
  // the packed native-type L^-1, made on demand from the resident factor's inverse
  bool linv_p_lazy = false;
  int ensure_linv_p() {
    if (linv_p_valid || !linv_p_lazy || !chol_valid) return GPSO_OK;
    launch_pack_linv<TF, TP>(st(), as<TF>(linv), n, npad, as<TP>(linv_p));
    small_tile_rows = 8;
    linv_p_valid = true;
    linv_p_lazy = false;
    return launch_status();
  }

  // ---- rank-k append at the resident hyper-parameters (append.hip) ------------------------------------------------------
  // Returns GPSO_OK when the resident factor was extended in place, 1 when the posterior of the n + k points was refitted
  // from scratch at the same hyper-parameters instead (gpso_last_error says why), or a negative status.
  int append(const double* Xn, const double* yn, int64_t k, double* nlml) override {
    ctx->tick_timing();
    if (!Xn || !yn) return ctx->fail(GPSO_E_ARG, "Xnew / ynew must not be NULL");
    if (k < 1) return ctx->fail(GPSO_E_ARG, "need at least one new point (k=%lld)", (long long)k);
    if (int rca = refuse_if_async("gpso_append")) return rca;
    if (!have_data || !have_post || !chol_valid || (int64_t)y_host.size() != n)
      return ctx->fail(GPSO_E_STATE, "gpso_append needs a posterior fitted on this context (gpso_set_data + gpso_fit_eval)");
    const int64_t n_new = n + k;
    if (n_new > 65536) return ctx->fail(GPSO_E_ARG, "n=%lld above the supported 65536", (long long)n_new);
    const char* refit = nullptr;
    if (k > kAppendMax) refit = "more than 64 new points in one call";
    else if (n_new > npad) refit = "the padded size grows: every buffer changes its layout";
    else if (fused_small && small_fit_eligible(n_new, dp)) refit = "N <= 128: the one-launch fit is the shorter exact update";
    // (measured, float, N = 2048: append of 64 points 0.55 ms, of 7 points 0.06 ms, posterior fit 0.52 ms -- the k x k corner
    // is factorised by one workgroup; at N = 8192 the same 64 points take 1.1 ms against 4.2: profiles/r05_append_bench.jsonl)
    else if (k > 32 && npad < 4096) refit = "more than 32 new points beside fewer than 4096: the refit is as fast";
    if (refit != nullptr) {
      std::vector<double> X(x_host), y(y_host), ls(ls_host);
      X.insert(X.end(), Xn, Xn + (size_t)k * d);
      y.insert(y.end(), yn, yn + (size_t)k);
      const KernParams th = kp;
      const int nls = n_ls, d_ = d;
      const int64_t n_old = n;
      int rc = set_data(X.data(), y.data(), n_new, d_);
      if (rc == GPSO_OK) rc = fit_eval(th.kernel, ls.data(), nls, th.variance, th.noise, th.mean_c, nlml, nullptr);
      if (rc < 0) {
        // the header's promise holds on this path too: a failed append leaves the posterior of the first n points resident
        // (set_data dropped it: put the old points back and refit at the resident hyper-parameters -- the fit that
        // succeeded before the call)
        const std::string why = ctx->err;
        X.resize((size_t)n_old * d_);
        y.resize((size_t)n_old);
        int rc2 = set_data(X.data(), y.data(), n_old, d_);
        if (rc2 == GPSO_OK) rc2 = fit_eval(th.kernel, ls.data(), nls, th.variance, th.noise, th.mean_c, nullptr, nullptr);
        if (rc2 < 0) return ctx->fail(rc, "%s (and the posterior of the first %lld points could not be restored: status %d -- "
                                          "gpso_set_data + gpso_fit_eval start over)", why.c_str(), (long long)n_old, rc2);
        return ctx->fail(rc, "%s (the posterior of the first %lld points is resident again)", why.c_str(), (long long)n_old);
      }
      ctx->fail(1, "gpso_append: posterior of the %lld points refitted from scratch at the resident hyper-parameters (%s)",
                (long long)n_new, refit);
      return 1;
    }
    hipStream_t s = st();
    int rc;
    const int kpad = append_kp((int)k);
    if ((rc = ensure(app, append_scratch_doubles(npad, kpad) * 8))) return rc;
    double* stage = ctx->pinned_stage((size_t)k * d + (size_t)k);  // (waits for the stream)
    double* host = ctx->pinned_scratch(8);
    if (!stage || !host) return ctx->fail(GPSO_E_OOM, "pinned host staging");
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[4], s));
    // (no copies: append_cross_kernel reads the new points from this pinned buffer and files them in x64 / y64 itself)
    std::memcpy(stage, Xn, (size_t)k * d * 8);
    std::memcpy(stage + (size_t)k * d, yn, (size_t)k * 8);
    AppendArgs a{};
    a.x64 = as<double>(x64); a.y64 = as<double>(y64); a.ls = ls_dev();
    a.xnew = stage; a.ynew = stage + (size_t)k * d;
    a.n = n; a.npad = npad; a.k = (int)k; a.d = d; a.dp = dp; a.kernel = kp.kernel;
    a.variance = kp.variance; a.noise = kp.noise; a.mean_c = kp.mean_c;
    a.xs64 = as<double>(xs64); a.xnorm64 = as<double>(xnorm64); a.xs_p64 = as<double>(xs_p64);
    a.diag64 = as<double>(logdet); a.nlml = as<double>(scal); a.kinv_diag = as<double>(kinv_diag);
    a.alpha_p = alpha.p; a.hyper = as<double>(hyper); a.host_out = host;
    // the fp16 split's scale follows max |L^-1|: the slot holds it when the pieces are built, or when the fit's solve
    // handed it over for a packing still to come
    const bool f16_max_live = kFloatPredict && f16_split() && bf16_usable() &&
                              (linv_b_valid || (linv_b_pending && f16_handed_at != nullptr && f16_handed_at == f16_scale()));
    a.f16_scal = f16_max_live ? f16_scale() : nullptr;
    host[0] = host[1] = 0.0;
    launch_append<TF, TP>(s, a, app.p, as<TF>(linv), as<TF>(Lf), as<TF>(white), as<TF>(alpha_f));
    // the 16-row tiles that hold new rows, in whichever predict-ready copies of L^-1 are built (the others are made on
    // demand from the extended matrix)
    const int64_t rt0 = n / 16;
    auto repack = [&](int64_t rows) {
      if (linv_b_valid) {
        if (f16_split()) launch_pack_linv_f16<TF>(s, as<TF>(linv), rows, npad, f16_scale(), split_planes(), true, rt0);
        else launch_pack_linv_bf16<TF>(s, nsplit(), as<TF>(linv), rows, npad, split_planes(), rt0);
      }
      if (linv_p_valid) launch_pack_linv<TF, TP>(s, as<TF>(linv), rows, npad, as<TP>(linv_p), rt0);
    };
    repack(n_new);
    if ((rc = launch_status())) return rc;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[5], s));
    HIPCHECK(ctx->wait(s));
    float ms = 0;
    if (ctx->timing && hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->last_ms[2] = ms;
    if (host[1] != 1.0) {
      // K22 + noise I - L21 L21^T is not positive definite: the posterior of the n points is still what the device holds
      // (append_cols_kernel did not run); the tiles just repacked and the scaled-input rows go back to their padding state
      const int pivot = (int)host[2];
      repack(n);
      (void)scale_inputs();
      gen32_inputs_ok = false;
      HIPCHECK(hipStreamSynchronize(s));
      return ctx->fail(GPSO_E_NOTPD, "K + noise*I is not positive definite: the appended block's Cholesky failed at pivot %d "
                       "(the posterior of the first %lld points is unchanged)", pivot, (long long)n);
    }
    x_host.insert(x_host.end(), Xn, Xn + (size_t)k * d);
    y_host.insert(y_host.end(), yn, yn + (size_t)k);
    n = n_new;
    if (nlml) *nlml = host[0];
    have_kinv = false;
    st_done = false;           // the self-test looks at the extended posterior before the next prediction
    gen32_inputs_ok = false;   // float / fp16 copies of the scaled inputs: derived again from the extended double ones
    small_tile_rows = 8;
    return GPSO_OK;
  }


  int set_posterior(const double* X, const double* L, const double* alpha64, int64_t n_, int d_,
                    int kernel, const double* ls, int n_ls_, double variance, double noise,
                    double mean_c) override {
    if (!X || !L || !alpha64 || !ls) return ctx->fail(GPSO_E_ARG, "NULL argument");
    int rc = refuse_if_async("gpso_set_posterior");
    if (rc) return rc;
    rc = shape(n_, d_);
    if (rc) return rc;
    if ((rc = ensure_fit_buffers())) return rc;
    if ((rc = set_theta(kernel, ls, n_ls_, variance, noise, mean_c))) return rc;
    if ((rc = ensure(getter_tmp, (size_t)n * n * 8 + (size_t)n * 8))) return rc;
    hipStream_t s = st();
    double* tmp = as<double>(getter_tmp);
    HIPCHECK(hipMemcpyAsync(x64.p, X, (size_t)n * d * 8, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(tmp, L, (size_t)n * n * 8, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(tmp + (size_t)n * n, alpha64, (size_t)n * 8, hipMemcpyHostToDevice, s));
    have_data = false;  // y unknown: a later fit needs gpso_set_data
    have_post = have_kinv = chol_valid = false;
    st_done = st_have = false;
    forget_peers();
    reset_generation();
    if ((rc = scale_inputs())) return rc;
    HIPCHECK(hipMemsetAsync(linv.p, 0, (size_t)npad * npad * sizeof(TF), s));
    launch_install_chol<TF>(s, tmp, n, npad, as<TF>(Lf), as<TF>(linv));
    launch_trtri<TF>(s, as<TF>(Lf), as<TF>(linv), as<TF>(work), npad, kFitBlock);
    HIPCHECK(hipMemsetAsync(alpha_f.p, 0, (size_t)npad * sizeof(TF), s));
    launch_convert_in<TF>(s, tmp + (size_t)n * n, as<TF>(alpha_f), 1, n, npad);
    launch_pack_linv<TF, TP>(s, as<TF>(linv), n, npad, as<TP>(linv_p));
    launch_convert_vec<TF, TP>(s, as<TF>(alpha_f), as<TP>(alpha), npad);
    small_tile_rows = 8;
    if ((rc = pack_bf16())) return rc;
    HIPCHECK(hipStreamSynchronize(s));  // the caller's host buffers are free again on return
    if ((rc = launch_status())) return rc;
    have_post = chol_valid = linv_p_valid = true;
    return GPSO_OK;
  }

  // ------------------------------------------------------------------------------------------
  // leaves already on the device as raw coordinates (dtype xs_dtype) -> mean/var(/ucb) device arrays.
  // TG = generation type (double unless the context is float-predict with GPSO_GEN_F32).
  // defer (nullable): the caller's arg-max can finalise the leaves itself (launch_seg_argmax / launch_keyed_argmax with
  // a LeafFinalize) -- when the batch is ONE chunk the finalize launch is skipped and *defer says what to finalise;
  // otherwise defer->part_var stays NULL and the leaves are finalised here as before
  template <typename TG>
  int score_leaves_t(const void* xs_dev, int xs_dtype, int64_t m, double varsigma, bool want_ucb,
                     double* mean_dev, double* var_dev, double* ucb_dev, const int64_t* m_live, LeafFinalize* defer,
                     bool prepared = false /* leaves_s / lnorm already hold the scaled rows (one chunk): no prep launch */) {
    // row blocks of L^-1 = partial sums per leaf: the split-bf16 kernel always works on 256-row blocks,
    // the native kernels on the shape leaf_tiles_bm picks
    const bool use_bf16 = bf16_usable() && linv_b_valid && bf16_fits(sizeof(TG) == 8);
    if (!use_bf16) {
      int rcp = ensure_linv_p();
      if (rcp) return rcp;
    }
    if (!use_bf16 && !linv_p_valid)
      return ctx->fail(GPSO_E_STATE, "this posterior was received with the split pieces of L^-1 only (the sender predicts with "
                                     "split math): the f32 MFMA kernel has nothing to read -- keep the sender's predict math, or "
                                     "hand the posterior over with gpso_posterior_buffers (all buffers)");
    const int nbi = use_bf16 ? (int)(npad / 256) : leaf_tiles_nbi<TP>(npad, dp / 4);
    const int64_t chunk = std::min<int64_t>(m, kLeafChunk);
    const int64_t cpad = (chunk + kLeafPad - 1) / kLeafPad * kLeafPad;
    int rc;
    if ((rc = ensure(leaves_s, (size_t)cpad * dp * sizeof(TG)))) return rc;
    if ((rc = ensure(lnorm, (size_t)cpad * sizeof(TG)))) return rc;
    if ((rc = ensure(pvar, (size_t)nbi * cpad * 8))) return rc;
    if ((rc = ensure(pmean, (size_t)nbi * cpad * 8))) return rc;
    hipStream_t s = st();
    ctx->tile_pairs = 0;
    const size_t in_elem = (xs_dtype == GPSO_F64) ? 8 : 4;
    const TG* xsp = nullptr;
    const TG* xnr = nullptr;
    if constexpr (sizeof(TG) == 8) {
      xsp = as<double>(xs_p64);
      xnr = as<double>(xnorm64);
    } else {
      xsp = as<float>(xs_p32);
      xnr = as<float>(xnorm32);
    }
    // m_live counts live rows of the WHOLE batch; a batch processed in several chunks needs the count
    // per chunk (slots 1.. of live_cnt, filled on the device)
    const int nchunk = (int)((m + chunk - 1) / chunk);
    if (m_live != nullptr && nchunk > 1) {
      if ((rc = ensure_keep(live_cnt, (size_t)(1 + nchunk) * 8))) return rc;
      m_live = as<int64_t>(live_cnt);
      launch_chunk_live(s, m_live, chunk, nchunk, as<int64_t>(live_cnt) + 1);
    }
    for (int64_t off = 0; off < m; off += chunk) {
      const int64_t mc = std::min<int64_t>(chunk, m - off);
      const int64_t mp = (mc + kLeafPad - 1) / kLeafPad * kLeafPad;
      const int64_t* m_live_c = (m_live != nullptr && nchunk > 1) ? as<int64_t>(live_cnt) + 1 + off / chunk : m_live;
      const char* src = static_cast<const char*>(xs_dev) + (size_t)off * d * in_elem;
      // Round 5: the fp16-contraction kernels scale float leaves in their own prologue (one launch and one dependency gap
      // less per call: 12 us of a 740 us step at C3).  Only where nothing else reads the scaled copies: one chunk, all rows
      // live, the caller's leaves in float.  A NaN coordinate reaches the partial sums by itself on this path (no clamp in
      // float generation), so the finalize stage needs no norms.
      RawLeaves rawl{};
      rawl.step32 = split_variant == GPSO_SPLIT_KERNEL_FUSED32;
      if constexpr (kFloatPredict && sizeof(TG) == 4) {
        if (fuse_prep && use_bf16 && !prepared && xs_dtype == GPSO_F32 && nchunk == 1 && m_live_c == nullptr && c16_in_use(false)) {
          rawl.x = reinterpret_cast<const float*>(src);
          rawl.ls = ls_dev();
          rawl.m = mc;
          rawl.d = d;
        }
      }
      const bool fused_prep = rawl.x != nullptr;
      if (fused_prep) {
        // (nothing to launch)
      } else if (prepared) {
        if (nchunk != 1) return ctx->fail(GPSO_E_STATE, "internal: prepared leaves in more than one chunk");
      } else if (xs_dtype == GPSO_F64) {
        launch_prep_leaves<TG, double>(s, reinterpret_cast<const double*>(src), mc, mp, d, dp, ls_dev(), m_live_c, as<TG>(leaves_s), as<TG>(lnorm));
      } else {
        launch_prep_leaves<TG, float>(s, reinterpret_cast<const float*>(src), mc, mp, d, dp, ls_dev(), m_live_c, as<TG>(leaves_s), as<TG>(lnorm));
      }
      while ((int)ctx->tile_ev.size() < 2 * (ctx->tile_pairs + 1)) {
        hipEvent_t e;
        HIPCHECK(hipEventCreate(&e));
        ctx->tile_ev.push_back(e);
      }
      if (ctx->timing) HIPCHECK(hipEventRecord(ctx->tile_ev[2 * ctx->tile_pairs], s));
      if (use_bf16) {
        if constexpr (kFloatPredict)
          rc = launch_leaf_tiles_bf16<TG>(s, nsplit(), split_planes(), xsp, xnr, as<float>(alpha), as<TG>(leaves_s),
                                          as<TG>(lnorm), as<double>(pvar), as<double>(pmean), npad, dp / 4, mp,
                                          kp, m_live_c, f16_split() ? f16_scale() : nullptr, split_variant,
                                          c16_in_use(sizeof(TG) == 8) ? xs_h16.p : nullptr, as<float>(c16_scal), n, rawl);
      } else {
        rc = launch_leaf_tiles<TP, TG>(s, as<TP>(linv_p), xsp, xnr, as<TP>(alpha), as<TG>(leaves_s),
                                       as<TG>(lnorm), as<double>(pvar), as<double>(pmean), npad, dp / 4, mp, kp,
                                       m_live_c);
      }
      if (rc) return launch_status();
      if (ctx->timing) HIPCHECK(hipEventRecord(ctx->tile_ev[2 * ctx->tile_pairs + 1], s));
      ++ctx->tile_pairs;
      if (defer != nullptr && nchunk == 1 && want_ucb) {
        *defer = LeafFinalize{as<double>(pvar), as<double>(pmean), nbi, mp, kp.variance, kp.noise, kp.mean_c, varsigma,
                              mean_dev, var_dev, ucb_dev, fused_prep ? nullptr : lnorm.p, sizeof(TG) == 8};
        continue;
      }
      launch_leaf_finalize(s, as<double>(pvar), as<double>(pmean), nbi, mp, mc, kp, varsigma, mean_dev + off,
                           var_dev + off, want_ucb ? ucb_dev + off : nullptr, fused_prep ? nullptr : lnorm.p, sizeof(TG) == 8);
    }
    return launch_status();  // (no host wait here: the kernel time is read after the call's final sync)
  }

  int score_device_leaves(const void* xs_dev, int xs_dtype, int64_t m, double varsigma, bool want_ucb,
                          double* mean_dev, double* var_dev, double* ucb_dev, const int64_t* m_live = nullptr,
                          LeafFinalize* defer = nullptr, bool prepared = false) {
    if (defer != nullptr) *defer = LeafFinalize{};
    if constexpr (kFloatPredict) {
      if (!gen_double()) {
        int rc = ensure_generation_inputs();
        if (rc) return rc;
        return score_leaves_t<float>(xs_dev, xs_dtype, m, varsigma, want_ucb, mean_dev, var_dev, ucb_dev, m_live, defer, prepared);
      }
    }
    return score_leaves_t<double>(xs_dev, xs_dtype, m, varsigma, want_ucb, mean_dev, var_dev, ucb_dev, m_live, defer, prepared);
  }

  // after the stream has been synchronised: total leaf-tile kernel time of the call
  void collect_tile_ms() {
    if (!ctx->timing) {
      ctx->tile_pairs = 0;
      return;
    }
    float total = 0;
    for (int i = 0; i < ctx->tile_pairs; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ctx->tile_ev[2 * i], ctx->tile_ev[2 * i + 1]) == hipSuccess) total += ms;
    }
    ctx->last_ms[0] = total;
    ctx->tile_pairs = 0;
  }

  // ---- precision self-test (see fit.hip: selftest_kernel) ---------------------------------------
  double tol_var_abs() const { return tol_var * kp.variance; }
  double tol_mean_abs() const { return tol_mean * std::max(st_vals[2], 1.0e-300); }
  // A float FACTOR (GPSO_F32) carries a backward error E that the training inputs see unamplified (there
  // the solve weights w = K_y^-1 k_i are ~ a unit vector) but a general leaf sees as w^T E w, i.e. times
  // |w|_1^2 <= sigma^2 |K_y^-1| =: kappa.  The bound itself is far too pessimistic (measured |w|^2 at the
  // leaves: 440 where kappa ~ 2.6e4); sqrt(kappa), with kappa estimated from max_i (K_y^-1)_ii, is a
  // CALIBRATED HEURISTIC: it covers every float32 case of tools/fuzz_gpu.py and tools/precision_probe.py
  // (profiles/r02*_precision*.jsonl).  GPSO_MIXED (double factor) needs no such factor and is exact.
  double amplification() const {
    return sizeof(TF) == 4 ? std::max(1.0, std::sqrt(kp.variance * st_vals[5])) : 1.0;
  }
  // margin > 1: the readings must sit that far INSIDE the tolerances
  bool st_pass(double margin = 1.0) const {
    const bool finite = std::isfinite(st_vals[0]) && std::isfinite(st_vals[1]) && std::isfinite(st_vals[3]);
    const double amp = amplification();  // var: w^T E w ~ |w|^2; mean: w^T E alpha ~ |w|
    return finite && margin * st_vals[0] * std::sqrt(amp) <= tol_mean_abs() && margin * st_vals[1] * amp <= tol_var_abs();
  }
  // GPSO_GEN_AUTO keeps float generation only when it is comfortably inside the tolerances: double
  // generation is typically 1e-6-class, and a posterior that float generation brings within a factor of
  // a few of the gate is better served by it
  static constexpr double kAutoMargin = 8.0;
  int run_selftest() {
    if (st_done) return GPSO_OK;
    if (!st_have || !have_data) return ctx->fail(GPSO_E_STATE, "self-test needs a posterior fitted on this context (gpso_fit_eval)");
    int rc;
    if ((rc = ensure(st_mean, (size_t)n * 8))) return rc;
    if ((rc = ensure(st_var, (size_t)n * 8))) return rc;
    if ((rc = ensure(st_out, 8 * 8))) return rc;  // 6 used
    if ((rc = score_device_leaves(x64.p, GPSO_F64, n, 0.0, false, as<double>(st_mean), as<double>(st_var), nullptr))) return rc;
    launch_selftest<TF>(st(), as<double>(st_mean), as<double>(st_var), as<double>(y64), as<TF>(alpha_f),
                        as<double>(kinv_diag), n, kp.noise, kp.mean_c, as<double>(st_out));
    double* host = ctx->pinned_scratch(8);
    if (!host) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    HIPCHECK(hipMemcpyAsync(host, st_out.p, 6 * 8, hipMemcpyDeviceToHost, st()));
    HIPCHECK(ctx->wait(st()));
    if ((rc = launch_status())) return rc;
    for (int i = 0; i < 6; ++i) st_vals[i] = host[i];
    st_done = true;
    return GPSO_OK;
  }
  // GPSO_GEN_AUTO: let the self-test choose the generation arithmetic of this posterior.  Without a
  // self-test (switched off, or a posterior installed from outside: no targets) the choice is double.
  int decide_generation() {
    if (int rcp = ensure_split_pieces()) return rcp;
    if (!kFloatPredict || gen_decided) return GPSO_OK;
    if (gen_mode != GPSO_GEN_AUTO) {
      gen_decided = true;
      return GPSO_OK;
    }
    // Matern-1/2 is exp(-sqrt(r^2)): near r = 0 the square root turns the 1e-6-class error of a float r^2
    // into 1e-3 of k -- for leaves CLOSE to a training input, while AT the training inputs (where the
    // self-test looks) the float form cancels exactly.  The test cannot see it, so this kernel generates
    // in double.
    const bool can_test = check && st_have && have_data && kp.kernel != GPSO_MATERN12;
    if (!can_test) {
      gen_eff32 = false;
      st_done = false;
      gen_decided = true;
      return GPSO_OK;
    }
    int rc = run_selftest();
    if (rc) return rc;
    if (!st_pass(kAutoMargin) && gen_eff32) {
      // not comfortably inside with float generation: measure double generation as well
      double r32[6];
      std::copy(st_vals, st_vals + 6, r32);
      const bool pass32 = st_pass();
      const bool was_c16 = c16_in_use(false);
      if (pass32 && bf16_usable() && linv_b_valid && !bf16_fits(true)) {
        // the split-bf16 kernel the caller opted into cannot hold double fragments at this D: double
        // generation would also mean the (slower) native kernel.  Inside the tolerances: stay.
        gen_decided = true;
        return GPSO_OK;
      }
      gen_eff32 = false;
      st_done = false;
      if ((rc = run_selftest())) return rc;
      // the float form of r^2 is what fails at small lengthscales / noise -- then double is clearly
      // better and stays.  Where the readings are the same within 1.5x the error is not the generation's
      // (a float factor's, the apply's): float generation is as good and is kept.
      const bool finite64 = std::isfinite(st_vals[0]) && std::isfinite(st_vals[1]);
      if (pass32 && finite64 && r32[0] <= 1.5 * st_vals[0] && r32[1] <= 1.5 * st_vals[1]) {
        gen_eff32 = true;
        std::copy(r32, r32 + 6, st_vals);
      } else if (was_c16 && finite64) {
        // float generation with the contraction on the fp16 pipe is not as good as double here: before paying for double
        // generation (the f64 matrix instruction: 1.3x the kernel time at C3's shape), a look at float generation with
        // the f32 contraction of rounds 1-3 -- its r^2 AT a training input is closer to zero (same floats in the norms
        // and the products), which is where this test looks
        double r64[6];
        std::copy(st_vals, st_vals + 6, r64);
        c16_fallback = true;
        gen_eff32 = true;
        st_done = false;
        if ((rc = run_selftest())) return rc;
        // (kept on the terms it is kept on when it runs first: comfortably inside the tolerances, or as good as double)
        if (!(st_pass(kAutoMargin) || (st_pass() && st_vals[0] <= 1.5 * r64[0] && st_vals[1] <= 1.5 * r64[1]))) {
          c16_fallback = false;
          gen_eff32 = false;
          std::copy(r64, r64 + 6, st_vals);
        }
      }
    }
    gen_decided = true;
    return GPSO_OK;
  }
  // the self-test, and under GPSO_MATH_AUTO a second look with the f32 MFMA kernel where the split-bf16 apply
  // misses the tolerances on this posterior
  int selftest_with_fallback() {
    int rc = run_selftest();
    if (rc) return rc;
    if (!st_pass() && math_auto && bf16_usable() && linv_b_valid) {
      if (math == GPSO_MATH_F16X3 && chol_valid) {  // next rung: six bf16 products (L^-1 is resident: repack)
        math = GPSO_MATH_BF16X6;
        if ((rc = pack_bf16())) return rc;
        st_done = false;
        if ((rc = run_selftest()) || st_pass()) return rc;
      }
      math_native_fallback = true;
      st_done = false;
      rc = run_selftest();
    }
    return rc;
  }
  int math_in_use() const {
    return (bf16_usable() && linv_b_valid && bf16_fits(gen_double())) ? math : GPSO_MATH_NATIVE;
  }
  // called at the top of every predict-type entry point
  int precision_gate() {
    int rc = decide_generation();
    if (rc) return rc;
    if (!check || !st_have || !have_data) return GPSO_OK;  // posteriors installed from outside carry no targets
    if ((rc = selftest_with_fallback())) return rc;
    if (!st_pass())
      return ctx->fail(GPSO_E_PRECISION,
                       "float predict arithmetic fails the self-test on this posterior: at the training inputs "
                       "max |d mean| = %.3g (tolerance %.3g), max |d var| = %.3g (tolerance %.3g = %.1e sigma^2), "
                       "amplification applied %.3g (float factor: sigma^2 max (K_y^-1)_ii), min predicted var %.3g "
                       "at noise %.3g, max |alpha| %.3g; use a GPSO_MIXED or GPSO_F64 context",
                       st_vals[0], tol_mean_abs(), st_vals[1], tol_var_abs(), tol_var, amplification(), st_vals[3],
                       kp.noise, st_vals[4]);
    return GPSO_OK;
  }
  int precision_info(double* out) override {
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident");
    int rc = decide_generation();
    if (rc) return rc;
    if ((rc = selftest_with_fallback())) return rc;
    for (int i = 0; i < 5; ++i) out[i] = st_vals[i];
    out[5] = kp.variance;
    out[6] = tol_mean_abs();
    out[7] = tol_var_abs();
    out[8] = amplification();
    out[9] = st_vals[5];
    out[10] = gen_double() ? 0.0 : 1.0;
    out[11] = (double)math_in_use();
    if (!st_pass()) return ctx->fail(GPSO_E_PRECISION, "self-test: max |d mean| %.3g (tol %.3g), max |d var| %.3g (tol %.3g), amplification %.3g",
                                     st_vals[0], out[6], st_vals[1], out[7], out[8]);
    return GPSO_OK;
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
    return precision_gate();
  }

  int predict(const void* xs, int xs_dtype, int xs_mem, int64_t m, double* mean, double* var,
              int out_mem) override {
    ctx->tick_timing();
    int rc = check_predict_args(xs, xs_dtype, xs_mem, m);
    if (rc) return rc;
    if (m == 0) return GPSO_OK;
    if (!mean || !var) return ctx->fail(GPSO_E_ARG, "mean / var must not be NULL");
    hipStream_t s = st();
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], s));
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
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[3], s));
    HIPCHECK(ctx->wait(s));
    collect_tile_ms();
    ctx->last_count[0] = ctx->last_count[1] = m;
    float ms = 0;
    if (ctx->timing && hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ms[1] = ms;
    return GPSO_OK;
  }

  // ---- best-UCB sequencing: enqueue the local work (winners stay on the device, no host wait) ->
  //      [multi-GPU: all-gather + fold] -> one read-back ------------------------------------------------
  // ---- one-launch small calls (predict.hip: leaf_tiles_v2_one_kernel) -------------------------------------------------
  // Applies when the native tile kernel runs this posterior with ONE row block (N_pad = 128 or 256) and the batch is
  // small.  Returns 0 when the call was enqueued (records on their way to ovals and pinned host memory), 2 when it does
  // not apply -- the caller goes on with the next-best sequence --, or a negative status.
  DevBuf one_ctl, one_partial, one_ppos;
  bool one_pending = false;  // the call in flight is a one-launch call: finish_best looks at its fallback verdict
  bool coll_timed = false;   // the call in flight recorded the event pair around its collective (all-gather + fold)
  template <typename TG>
  int try_one_launch_t(OneLaunch& one, int64_t total, int nseg, double varsigma) {
    const int64_t cpad = (total + kLeafPad - 1) / kLeafPad * kLeafPad;
    int rc;
    if ((rc = ensure(leaves_s, (size_t)cpad * dp * sizeof(TG)))) return rc;
    if ((rc = ensure(lnorm, (size_t)cpad * sizeof(TG)))) return rc;
    if ((rc = ensure(pvar, (size_t)cpad * 8))) return rc;
    if ((rc = ensure(pmean, (size_t)cpad * 8))) return rc;
    if ((rc = ensure(grow_key, (size_t)cpad * 8))) return rc;
    const size_t wgs = (size_t)(cpad / 64);  // (at most: 64 leaves per workgroup is the smallest tile)
    if ((rc = ensure(one_partial, wgs * nseg * kArgmaxPartialBytes))) return rc;
    if ((rc = ensure(one_ppos, wgs * nseg * 8))) return rc;
    if (one_ctl.p == nullptr) {
      if ((rc = ensure(one_ctl, 64))) return rc;
      HIPCHECK(hipMemsetAsync(one_ctl.p, 0, 64, st()));
    }
    one.total = total;
    one.nseg = nseg;
    one.d = d;
    one.ls = ls_dev();
    one.leaves_s = leaves_s.p;
    one.lnorm = lnorm.p;
    one.key = as<int64_t>(grow_key);
    one.fin = LeafFinalize{as<double>(pvar), as<double>(pmean), 1, cpad, kp.variance, kp.noise, kp.mean_c, varsigma,
                           as<double>(omean), as<double>(ovar), as<double>(oucb), lnorm.p, sizeof(TG) == 8};
    one.partial = one_partial.p;
    one.ppos = as<int64_t>(one_ppos);
    one.ticket = static_cast<unsigned*>(one_ctl.p);
    one.fallback = static_cast<unsigned*>(one_ctl.p) + 1;
    one.out_vals = as<double>(ovals);
    one.host_vals = host_direct = result_host((size_t)nseg * 4 + 2);
    one.done_token = arm_token(true, nseg);  // (the last-arriving workgroup's thread 0 writes every host record, then the token)
    const TG* xsp;
    const TG* xnr;
    if constexpr (sizeof(TG) == 8) {
      xsp = as<double>(xs_p64);
      xnr = as<double>(xnorm64);
    } else {
      xsp = as<float>(xs_p32);
      xnr = as<float>(xnorm32);
    }
    ctx->tile_pairs = 0;
    while ((int)ctx->tile_ev.size() < 2) {
      hipEvent_t e;
      HIPCHECK(hipEventCreate(&e));
      ctx->tile_ev.push_back(e);
    }
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->tile_ev[0], st()));
    rc = launch_leaf_tiles_one<TP, TG>(st(), as<TP>(linv_p), xsp, xnr, as<TP>(alpha), as<double>(pvar), as<double>(pmean),
                                      npad, dp / 4, cpad, kp, one);
    if (rc == 2) {
      host_direct = nullptr;
      return 2;
    }
    if (rc) return launch_status();
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->tile_ev[1], st()));
    ctx->tile_pairs = ctx->timing ? 1 : 0;
    one_pending = true;
    return launch_status();
  }
  int try_one_launch(OneLaunch& one, int64_t total, int nseg, double varsigma) {
    if (!small_calls || !one_launch || one_refused || total <= 0 || total > kSmallBestMaxRows || nseg > 16) return 2;
    if (npad != 128 && npad != 256) return 2;
    // measured (tools/micro/one_dbg.py, float64, same box; wall us per best_ucb_grow call, one | three launches):
    // N = 52 / 162 rows 56.3 | 54.1, N = 100 / 1458 rows 61.0 | 59.8, N = 128 / 13122 rows 91.5 | 88.6, N = 256 / 4374 rows
    // 78.4 | 85.7 -- the two agent-scope fences of the last-arriver fold cost what the two launch boundaries they replace
    // cost; the single launch only pays at N_pad = 256
    if (npad == 128 && !one_launch_everywhere) return 2;
    if (math_in_use() != GPSO_MATH_NATIVE) return 2;
    if (ensure_linv_p() != GPSO_OK || !linv_p_valid) return 2;
    if (leaf_tiles_nbi<TP>(npad, dp / 4) != 1) return 2;
    if constexpr (kFloatPredict) {
      if (!gen_double()) {
        int rc = ensure_generation_inputs();
        if (rc) return rc;
        return try_one_launch_t<float>(one, total, nseg, varsigma);
      }
    }
    return try_one_launch_t<double>(one, total, nseg, varsigma);
  }

  // per segment (mean, var, ucb, bit-cast index relative to the segment start) -> ovals; seg_off: host,
  // nseg + 1 entries over [0, m]
  int enqueue_best_leaves(const void* dev, int xs_dtype, int64_t m, const int64_t* seg_off, int nseg,
                          double varsigma) {
    int rc;
    one_pending = false;  // (a call whose result was never read -- gpso_shard_winners -- leaves nothing behind)
    host_direct = nullptr;
    call_token = 0.0;
    hipStream_t s = st();
    if ((rc = ensure(omean, (size_t)m * 8))) return rc;
    if ((rc = ensure(ovar, (size_t)m * 8))) return rc;
    if ((rc = ensure(oucb, (size_t)m * 8))) return rc;
    if ((rc = ensure(segoff, (size_t)(nseg + 1) * 8))) return rc;
    if ((rc = ensure(best, (size_t)nseg * kArgmaxBlocks * kArgmaxPartialBytes))) return rc;
    if ((rc = ensure(ovals, (size_t)group_payload_doubles(nseg) * 8))) return rc;
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
    if (m > 0) {  // one launch for the whole call where it applies
      OneLaunch one{};
      one.mode = 2;
      one.raw = dev;
      one.raw_f64 = xs_dtype == GPSO_F64 ? 1 : 0;
      one.seg_off = as<int64_t>(segoff);
      if ((rc = try_one_launch(one, m, nseg, varsigma)) != 2) {
        ctx->last_count[0] = ctx->last_count[1] = m;
        return rc;
      }
    }
    LeafFinalize fin{};
    if (m > 0)
      if ((rc = score_device_leaves(dev, xs_dtype, m, varsigma, true, as<double>(omean), as<double>(ovar), as<double>(oucb), nullptr, &fin))) return rc;
    // (round 4: the leaves of a one-chunk batch are finalised by the arg-max's first stage, and its second stage writes
    // the winners' records straight into the pinned host memory finish_best reads: two launches and a copy less per call)
    host_direct = result_host((size_t)nseg * 4 + 2);
    if (small_calls && fin.part_var != nullptr && m <= kSmallBestMaxRows && nseg <= 64) {
      // small batch: finalize and both arg-max stages in one launch of one workgroup
      SmallBest sb{};
      sb.fin = fin;
      sb.seg_off = as<int64_t>(segoff);
      sb.m = m;
      sb.nseg = nseg;
      sb.out_vals = as<double>(ovals);
      sb.host_vals = host_direct;
      sb.done_token = arm_token(true, nseg);
      launch_small_best(s, sb, false);
    } else {
      launch_seg_argmax(s, as<double>(omean), as<double>(ovar), as<double>(oucb), as<int64_t>(segoff),
                        nseg, argmax_blocks(m, nseg), best.p, as<double>(ovals), fin.part_var ? &fin : nullptr, host_direct,
                        arm_token(nseg == 1, nseg));
    }
    ctx->last_count[0] = ctx->last_count[1] = m;
    return launch_status();
  }

  // the reference rows [row_lo, row_hi) of the sub-tree under every box, de-duplicated (grow.hip), scored;
  // per segment (mean, var, ucb, bit-cast REFERENCE row index) -> ovals, live row count -> ovals[nseg*4]
  int enqueue_best_grow(const double* bounds, int nseg, int depth, int64_t row_lo, int64_t row_hi,
                        double varsigma) {
    if (!bounds) return ctx->fail(GPSO_E_ARG, "bounds must not be NULL");
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    if (depth < 0 || depth > 16) return ctx->fail(GPSO_E_ARG, "depth %d outside [0, 16]", depth);
    int rc;
    one_pending = false;
    host_direct = nullptr;
    call_token = 0.0;
    hipStream_t s = st();
    const int64_t rows = gpso_grow_rows(depth);
    const int64_t uniq = grow_unique_before(row_hi) - grow_unique_before(row_lo);
    const int64_t cap = (int64_t)nseg * (row_hi - row_lo);  // worst case: no centre child repeats its parent
    const size_t bb = (size_t)nseg * d * 2 * 8;
    const size_t ob = (size_t)cap * d * 8;
    if ((rc = ensure(leaves_raw, ob + bb))) return rc;
    if ((rc = ensure(grow_key, (size_t)std::max<int64_t>(cap, 1) * 8))) return rc;
    if ((rc = ensure_keep(live_cnt, 64))) return rc;
    if ((rc = ensure(omean, (size_t)cap * 8))) return rc;
    if ((rc = ensure(ovar, (size_t)cap * 8))) return rc;
    if ((rc = ensure(oucb, (size_t)cap * 8))) return rc;
    if ((rc = ensure(best, (size_t)nseg * kArgmaxBlocks * kArgmaxPartialBytes))) return rc;
    if ((rc = ensure(best_pos, (size_t)nseg * kArgmaxBlocks * 8))) return rc;
    if ((rc = ensure(ovals, (size_t)group_payload_doubles(nseg) * 8))) return rc;
    // Small calls (the optimiser's exploration levels: a few hundred to a few thousand rows) are launch-bound: boxes by
    // value, growth + input scaling in one launch, tiles, one-workgroup finalize + arg-max writing pinned host memory --
    // three launches and no copy operation instead of two copies in, five launches and a copy back
    if (small_calls && cap > 0 && cap <= kSmallBestMaxRows && nseg <= 64 && (size_t)nseg * d * 2 <= (size_t)kGrowBoxDoubles) {
      if (row_lo == 0 && row_hi == rows && !one_refused) {  // one launch for the whole call where it applies
        OneLaunch one{};
        one.mode = 1;
        std::memcpy(one.boxes.b, bounds, bb);
        one.depth = depth;
        one.rows = rows;
        one.uniq = uniq;
        if ((rc = try_one_launch(one, (int64_t)nseg * uniq, nseg, varsigma)) != 2) {
          ctx->last_count[1] = cap;
          return rc;
        }
      }
      const int64_t cpad = (cap + kLeafPad - 1) / kLeafPad * kLeafPad;
      const size_t tg = gen_double() ? 8 : 4;
      if ((rc = ensure(leaves_s, (size_t)cpad * dp * tg))) return rc;
      if ((rc = ensure(lnorm, (size_t)cpad * tg))) return rc;
      if (extra_cnt.p == nullptr) {
        if ((rc = ensure(extra_cnt, 64))) return rc;
        HIPCHECK(hipMemsetAsync(extra_cnt.p, 0, 64, s));
      }
      GrowBoxes gb;
      std::memcpy(gb.b, bounds, bb);
      unsigned long long* extra = static_cast<unsigned long long*>(extra_cnt.p);
      if constexpr (kFloatPredict) {
        if (!gen_double()) {
          if ((rc = ensure_generation_inputs())) return rc;
          launch_grow_unique_prep<float>(s, gb, nseg, d, dp, depth, row_lo, row_hi, ls_dev(), as<float>(leaves_s), as<float>(lnorm), as<int64_t>(grow_key), extra);
        }
      }
      if (gen_double())
        launch_grow_unique_prep<double>(s, gb, nseg, d, dp, depth, row_lo, row_hi, ls_dev(), as<double>(leaves_s), as<double>(lnorm), as<int64_t>(grow_key), extra);
      LeafFinalize fin{};
      if ((rc = score_device_leaves(nullptr, GPSO_F64, cap, varsigma, true, as<double>(omean), as<double>(ovar), as<double>(oucb), nullptr, &fin, true)))
        return rc;
      if (fin.part_var == nullptr) return ctx->fail(GPSO_E_STATE, "internal: small growth call was not deferred");
      host_direct = result_host((size_t)nseg * 4 + 2);
      SmallBest sb{};
      sb.fin = fin;
      sb.key = as<int64_t>(grow_key);
      sb.rows = rows;
      sb.uniq = uniq;
      sb.base = (int64_t)nseg * uniq;
      sb.extra = extra;
      sb.nseg = nseg;
      sb.out_vals = as<double>(ovals);
      sb.host_vals = host_direct;
      sb.done_token = arm_token(true, nseg);
      launch_small_best(s, sb, true);
      ctx->last_count[1] = cap;
      return launch_status();
    }
    double* bdev = reinterpret_cast<double*>(static_cast<char*>(leaves_raw.p) + ob);
    double* stage = ctx->pinned_stage((size_t)nseg * d * 2 + 1);
    if (!stage) return ctx->fail(GPSO_E_OOM, "pinned host staging");
    std::memcpy(stage, bounds, bb);
    const int64_t live0 = (int64_t)nseg * uniq;
    std::memcpy(stage + (size_t)nseg * d * 2, &live0, 8);
    HIPCHECK(hipMemcpyAsync(bdev, stage, bb, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(live_cnt.p, stage + (size_t)nseg * d * 2, 8, hipMemcpyHostToDevice, s));
    launch_grow_unique(s, bdev, nseg, d, depth, row_lo, row_hi, as<double>(leaves_raw), as<int64_t>(grow_key),
                       as<int64_t>(live_cnt));
    LeafFinalize fin{};
    if (cap > 0)
      if ((rc = score_device_leaves(leaves_raw.p, GPSO_F64, cap, varsigma, true, as<double>(omean), as<double>(ovar),
                                    as<double>(oucb), as<int64_t>(live_cnt), &fin)))
        return rc;
    host_direct = result_host((size_t)nseg * 4 + 2);
    launch_keyed_argmax(s, as<double>(omean), as<double>(ovar), as<double>(oucb), as<int64_t>(grow_key), rows, uniq,
                        nseg, as<int64_t>(live_cnt), argmax_blocks(cap, nseg), best.p, as<int64_t>(best_pos), as<double>(ovals),
                        fin.part_var ? &fin : nullptr, host_direct, arm_token(nseg == 1, nseg));
    ctx->last_count[1] = cap;
    return launch_status();
  }

  // ---- group calls ---------------------------------------------------------------------------------------
  // Every rank takes part in every collective of a call, whatever happened in its local half: a rank whose half
  // fails (arguments, state, precision gate, allocation) still sends a payload -- no winners and its status in the
  // last slot (kernels.hpp: group_payload_doubles) -- the fold takes the worst status over the ranks, and EVERY rank
  // returns that code.  No rank is left inside a collective by a peer that returned early, and a posterior that
  // fails the precision self-test on the fitting rank makes the whole group return GPSO_E_PRECISION together.
  int ensure_group_buffers(int nseg, int world) {
    int rc;
    const size_t pd = (size_t)group_payload_doubles(nseg);
    if ((rc = ensure(ovals, pd * 8))) return rc;
    if ((rc = ensure(gath, (size_t)world * pd * 8))) return rc;
    if ((rc = ensure(ovals2, pd * 8))) return rc;
    return GPSO_OK;
  }
  // base (host, [world][nseg]): what is added to a rank's winner index to make it relative to the GLOBAL segment
  // start -- the offset of the rank's piece of the segment inside the segment
  static void segment_bases(int64_t m_global, const std::vector<int64_t>& so, int nseg, int world,
                            std::vector<int64_t>& base) {
    base.assign((size_t)world * nseg, 0);
    for (int r = 0; r < world; ++r) {
      int64_t rlo, rhi;
      shard_range(m_global, r, world, &rlo, &rhi);
      for (int i = 0; i < nseg; ++i) base[(size_t)r * nseg + i] = std::min(std::max(so[i], rlo), rhi) - so[i];
    }
  }
  int global_segments(int64_t m_global, const int64_t* seg_off, int nseg, std::vector<int64_t>& so) {
    so.resize(nseg + 1);
    if (seg_off) {
      for (int i = 0; i <= nseg; ++i) so[i] = seg_off[i];
      if (so[0] != 0 || so[nseg] != m_global) return ctx->fail(GPSO_E_ARG, "seg_off must start at 0 and end at the global M");
      for (int i = 0; i < nseg; ++i)
        if (so[i + 1] < so[i]) return ctx->fail(GPSO_E_ARG, "seg_off must be non-decreasing");
    } else {
      if (nseg != 1) return ctx->fail(GPSO_E_ARG, "seg_off == NULL requires nseg == 1");
      so[0] = 0;
      so[1] = m_global;
    }
    return GPSO_OK;
  }
  std::vector<int64_t> wbase_cache;  // what the device copy of the bases holds (they change with the batch shape only)
  int upload_base(const std::vector<int64_t>& base) {
    if (wbase.p != nullptr && base == wbase_cache) return GPSO_OK;
    wbase_cache.clear();
    int rc = ensure(wbase, base.size() * 8);
    if (rc) return rc;
    double* stage = ctx->pinned_stage(base.size());
    if (!stage) return ctx->fail(GPSO_E_OOM, "pinned host staging");
    std::memcpy(stage, base.data(), base.size() * 8);
    HIPCHECK(hipMemcpyAsync(wbase.p, stage, base.size() * 8, hipMemcpyHostToDevice, st()));
    wbase_cache = base;
    return GPSO_OK;
  }
  // local half of gpso_best_ucb_sharded for (rank, world): rows shard_range(m_global, rank, world) of the batch ->
  // this rank's per-segment winners in ovals (indices relative to the LOCAL piece of each segment), status slot 0
  int sharded_local(int rank, int world, const void* xs, int xs_dtype, int xs_mem, int64_t m_local, int64_t m_global,
                    const int64_t* seg_off, int nseg, double varsigma) {
    int rc = check_predict_args(xs, xs_dtype, xs_mem, m_local);
    if (rc) return rc;
    int64_t lo, hi;
    shard_range(m_global, rank, world, &lo, &hi);
    if (m_local != hi - lo)
      return ctx->fail(GPSO_E_ARG, "rank %d of %d must pass rows [%lld, %lld) of the %lld leaves (gpso_shard_range), got %lld rows",
                       rank, world, (long long)lo, (long long)hi, (long long)m_global, (long long)m_local);
    std::vector<int64_t> so, sl(nseg + 1);
    if ((rc = global_segments(m_global, seg_off, nseg, so))) return rc;
    // local segmentation: the part of every global segment inside [lo, hi)
    for (int i = 0; i <= nseg; ++i) sl[i] = std::min(std::max(so[i], lo), hi) - lo;
    const void* dev = nullptr;
    if (m_local > 0 && (rc = stage_leaves(xs, xs_dtype, xs_mem, m_local, &dev))) return rc;
    one_refused = true;  // (the status slot of a group payload belongs to the group's verdict: the general sequences)
    rc = enqueue_best_leaves(dev, xs_dtype, m_local, sl.data(), nseg, varsigma);
    one_refused = false;
    return rc;
  }
  int sharded_local_grow(int rank, int world, const double* bounds, int nseg, int depth, double varsigma) {
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident: call gpso_broadcast_posterior first");
    int rc = precision_gate();
    if (rc) return rc;
    int64_t lo, hi;
    shard_range(gpso_grow_rows(depth), rank, world, &lo, &hi);
    one_refused = true;  // (a group payload has no way to ask for a re-run: the sequences with the append counter)
    rc = enqueue_best_grow(bounds, nseg, depth, lo, hi, varsigma);
    one_refused = false;
    return rc;
  }
  // after the local half: on failure replace whatever it left in ovals by "no winner" rows and the status
  // (on success the arg-max kernels have written the status slot themselves: no host step on the fast path)
  int publish_local(int nseg, int local_rc) {
    if (local_rc == GPSO_OK) return GPSO_OK;
    const std::string keep = ctx->err;  // the local half's message
    hipStream_t s = st();
    const size_t pd = (size_t)group_payload_doubles(nseg);
    double* stage = ctx->pinned_stage(pd);  // (waits for the stream)
    if (!stage) return ctx->fail(GPSO_E_OOM, "pinned host staging");
    const int64_t none = -1;
    for (int i = 0; i < nseg; ++i) {
      stage[4 * i] = stage[4 * i + 1] = stage[4 * i + 2] = std::nan("");
      std::memcpy(&stage[4 * i + 3], &none, 8);
    }
    stage[4 * nseg] = 0.0;
    stage[4 * nseg + 1] = (double)local_rc;
    HIPCHECK(hipMemcpyAsync(ovals.p, stage, pd * 8, hipMemcpyHostToDevice, s));
    ctx->err = keep;
    return GPSO_OK;
  }
  // multi-GPU: all-gather every rank's payload and fold the winners with the arg-max rule on (ucb, global index)
  // and the statuses with min; base = wbase (uploaded before the local half) or none.  Result -> ovals2.
  int exchange_winners(int nseg, bool with_base) {
    RcclApi& R = RcclApi::get();
    hipStream_t s = st();
    const int pd = group_payload_doubles(nseg);
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[6], s));
    RCCLCHECK(R.AllGather(ovals.p, gath.p, (size_t)pd, ncclDouble, ctx->comm, s));
    // (the fold writes the group's records straight into the pinned memory finish_best reads: no copy operation behind it)
    host_direct = result_host((size_t)nseg * 4 + 2);
    launch_reduce_winners(s, as<double>(gath), with_base ? as<int64_t>(wbase) : nullptr, ctx->world, nseg, pd,
                          as<double>(ovals2), host_direct, arm_token(nseg <= 64, nseg));
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[7], s));
    coll_timed = ctx->timing;
    return launch_status();
  }

  // one read-back of nseg x (mean, var, ucb, index) [+ the live row count] [+ the group's verdict], then the host
  // wait.  mode 0: winners only; 1: + live count -> last_count[0]; 2: a folded group payload -- *verdict_out: the worst
  // status of the group (*who: the rank it came from); the outputs are only written when that is GPSO_OK
  int finish_best(const DevBuf& src, int nseg, int mode, int64_t* idx, double* mean, double* var, double* ucb,
                  int* verdict_out = nullptr, int64_t* who = nullptr) {
    hipStream_t s = st();
    const bool one = one_pending;
    one_pending = false;
    const size_t doubles = (size_t)nseg * 4 + ((mode == 2 || one) ? 2 : mode);
    double* vals = ctx->pinned_scratch(doubles);
    if (!vals) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    // (mode 0 / 1: the arg-max's second stage has written the records into this very memory: no copy operation)
    const bool direct = host_direct != nullptr && host_direct == vals && src.p == (mode == 2 ? ovals2.p : ovals.p);
    host_direct = nullptr;
    if (!direct) HIPCHECK(hipMemcpyAsync(vals, src.p, doubles * 8, hipMemcpyDeviceToHost, s));
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[3], s));
    const double token = call_token;
    call_token = 0.0;
    if (direct && token != 0.0 && !ctx->timing) HIPCHECK(ctx->wait_token(&vals[4 * nseg + 2], token, s));
    else HIPCHECK(ctx->wait(s));
    int rc;
    if ((rc = launch_status())) return rc;
    collect_tile_ms();
    float ms = 0;
    if (ctx->timing && hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ms[1] = ms;
    if (coll_timed && hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]) == hipSuccess) ctx->last_ms[3] = ms;
    coll_timed = false;
    // (a centre child does not repeat its parent: the caller re-runs.  Only where a re-run exists -- a folded group
    // payload keeps the GROUP's verdict in this slot, and its half was enqueued with the one-launch kernel refused)
    if (one && mode != 2 && vals[4 * nseg + 1] != 0.0) return 1;
    if (mode == 2) {
      const int verdict = (int)vals[4 * nseg + 1];
      if (verdict_out) *verdict_out = verdict;
      if (who) std::memcpy(who, &vals[4 * nseg], 8);
      if (verdict != GPSO_OK) return GPSO_OK;  // (outputs untouched; the caller turns the verdict into its return value)
    }
    for (int i = 0; i < nseg; ++i) {
      if (idx) std::memcpy(&idx[i], &vals[4 * i + 3], 8);
      if (mean) mean[i] = vals[4 * i];
      if (var) var[i] = vals[4 * i + 1];
      if (ucb) ucb[i] = vals[4 * i + 2];
    }
    if (mode == 1) std::memcpy(&ctx->last_count[0], &vals[4 * nseg], 8);
    return GPSO_OK;
  }
  // the group's verdict as this rank's return value: its own failure keeps its own message
  int group_verdict(int verdict, int64_t who, int local_rc, const std::string& local_msg, const char* call) {
    if (verdict == GPSO_OK) return GPSO_OK;
    if (local_rc != GPSO_OK) {
      ctx->err = local_msg;
      return local_rc;
    }
    return ctx->fail(verdict, "%s: rank %lld of the group failed its half of the call with status %d (its message is "
                     "on that rank); every rank returns together", call, (long long)who, verdict);
  }

  int best_ucb(const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off, int nseg,
               double varsigma, int64_t* idx, double* mean, double* var, double* ucb) override {
    ctx->tick_timing();
    int rc = check_predict_args(xs, xs_dtype, xs_mem, m);
    if (rc) return rc;
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    const void* dev = nullptr;
    if (m > 0 && (rc = stage_leaves(xs, xs_dtype, xs_mem, m, &dev))) return rc;
    if ((rc = enqueue_best_leaves(dev, xs_dtype, m, seg_off, nseg, varsigma))) return rc;
    return finish_best(ovals, nseg, 0, idx, mean, var, ucb);
  }

  // ---- the non-blocking pair (round 5): gpso_best_ucb_begin / gpso_best_ucb_grow_begin enqueue a call and return a
  // ticket; gpso_best_ucb_end waits for THAT call and reads its records.  Two calls may be in flight: the device work of
  // the second is queued while the first still runs, so the GPU never idles over the host's round trip (pinned-memory
  // read, ctypes, Python: 42 us of a 0.77 ms step at C3, profiles/r04_step_timeline.txt).  Every call has its own pinned
  // result slot -- the arg-max kernels write their records there -- and a completion event; the device-side scratch is
  // shared and ordered by the stream.  No timing events, no one-launch kernel (it may ask for a re-run).
  int best_ucb_begin(const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off, int nseg, double varsigma,
                     const double* bounds, int depth) override {
    const bool grown = bounds != nullptr;
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident: call gpso_fit_eval / gpso_set_posterior first");
    if (nseg < 1 || (size_t)nseg * 4 + 3 > gpso_ctx::kSlotDoubles) return ctx->fail(GPSO_E_ARG, "nseg must be in [1, 1024] for an asynchronous call");
    int rc = grown ? precision_gate() : check_predict_args(xs, xs_dtype, xs_mem, m);
    if (rc) return rc;
    // any free slot (the round-robin one first): calls may be ended out of order, so after begin A, begin B, end B the free slot
    // is the one slot_next does NOT name -- refuse only when both hold a call
    int k = ctx->slot_next;
    if (ctx->slot_nseg[k] > 0) k = (k + 1) % gpso_ctx::kSlots;
    if (ctx->slot_nseg[k] > 0)
      return ctx->fail(GPSO_E_STATE, "two asynchronous best-UCB calls are already in flight: end one (gpso_best_ucb_end) first");
    if (ctx->slot_host == nullptr) {
      if (hipHostMalloc(reinterpret_cast<void**>(&ctx->slot_host), gpso_ctx::kSlots * gpso_ctx::kSlotDoubles * 8, hipHostMallocCoherent) != hipSuccess)
        return ctx->fail(GPSO_E_OOM, "pinned result slots");
      std::memset(ctx->slot_host, 0, gpso_ctx::kSlots * gpso_ctx::kSlotDoubles * 8);
      for (auto& ev : ctx->slot_ev) HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    {
      // the enqueue runs untimed, into this call's own result slot, with the one-launch kernel refused: whichever way the
      // region is left (an early return included), the context's synchronous calls find their state as it was
      struct AsyncRegion {
        EngineT* e;
        bool timing_was;
        AsyncRegion(EngineT* e_, double* slot) : e(e_), timing_was(e_->ctx->timing) {
          e->ctx->timing = false;
          e->result_slot = slot;
          e->one_refused = true;
        }
        ~AsyncRegion() {
          e->one_refused = false;
          e->result_slot = nullptr;
          e->host_direct = nullptr;
          e->call_token = 0.0;
          e->ctx->timing = timing_was;
        }
      } region(this, ctx->slot_host + (size_t)k * gpso_ctx::kSlotDoubles);
      if (grown) {
        rc = enqueue_best_grow(bounds, nseg, depth, 0, gpso_grow_rows(depth), varsigma);
      } else {
        const void* dev = nullptr;
        if (m > 0) rc = stage_leaves(xs, xs_dtype, xs_mem, m, &dev);
        if (rc == GPSO_OK) rc = enqueue_best_leaves(dev, xs_dtype, m, seg_off, nseg, varsigma);
      }
      ctx->slot_token[k] = call_token;
    }
    if (rc) return rc;
    if (xs_mem == GPSO_MEM_HOST && !grown && m > 0) HIPCHECK(hipStreamSynchronize(st()));  // (the caller's host leaves are free again on return)
    HIPCHECK(hipEventRecord(ctx->slot_ev[k], st()));
    ctx->slot_nseg[k] = nseg;
    ctx->slot_mode[k] = grown ? 1 : 0;
    ctx->slot_next = (k + 1) % gpso_ctx::kSlots;
    return k;
  }
  int best_ucb_end(int ticket, int64_t* idx, double* mean, double* var, double* ucb) override {
    if (ticket < 0 || ticket >= gpso_ctx::kSlots || ctx->slot_nseg[ticket] <= 0)
      return ctx->fail(GPSO_E_ARG, "ticket %d names no asynchronous call in flight", ticket);
    const int nseg = ctx->slot_nseg[ticket];
    ctx->slot_nseg[ticket] = 0;
    const double* vals = ctx->slot_host + (size_t)ticket * gpso_ctx::kSlotDoubles;
    // the call's completion token where its last kernel writes one, else its own event: spin (the thread is blocked here
    // anyway), then block
    const auto t0 = std::chrono::steady_clock::now();
    if (ctx->slot_token[ticket] != 0.0) {
      HIPCHECK(ctx->wait_token(&vals[4 * nseg + 2], ctx->slot_token[ticket], st()));
    } else
    for (int it = 0;; ++it) {
      const hipError_t e = hipEventQuery(ctx->slot_ev[ticket]);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) return ctx->fail(GPSO_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
      if ((it & 63) == 63 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
        HIPCHECK(hipEventSynchronize(ctx->slot_ev[ticket]));
        break;
      }
    }
    int rc;
    if ((rc = launch_status())) return rc;
    for (int i = 0; i < nseg; ++i) {
      if (idx) std::memcpy(&idx[i], &vals[4 * i + 3], 8);
      if (mean) mean[i] = vals[4 * i];
      if (var) var[i] = vals[4 * i + 1];
      if (ucb) ucb[i] = vals[4 * i + 2];
    }
    if (ctx->slot_mode[ticket] == 1) std::memcpy(&ctx->last_count[0], &vals[4 * nseg], 8);
    return GPSO_OK;
  }

  int need_comm() {
    if (ctx->comm == nullptr) return ctx->fail(GPSO_E_STATE, "this context is not part of a group: call gpso_comm_init first");
    if (ctx->comm_aborted.load())
      return ctx->fail(GPSO_E_STATE, "the communicator of this context was aborted (gpso_comm_abort): gpso_comm_destroy + "
                                     "gpso_comm_init again before the next group call");
    return GPSO_OK;
  }

  // This rank's contiguous share of a batch of m_global leaves (rows shard_range(m_global, rank, world) of
  // it) -> the GLOBAL per-segment winners, identical on every rank and identical to gpso_best_ucb on the
  // whole batch.  seg_off (host, nseg + 1 entries over [0, m_global], NULL: one segment) is global.
  int best_ucb_sharded(const void* xs, int xs_dtype, int xs_mem, int64_t m_local, int64_t m_global,
                       const int64_t* seg_off, int nseg, double varsigma, int64_t* idx, double* mean,
                       double* var, double* ucb) override {
    ctx->tick_timing();
    int rc = need_comm();
    if (rc) return rc;
    // (arguments every rank passes alike fail alike: no collective has started yet)
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    if (m_global < 0) return ctx->fail(GPSO_E_ARG, "negative global leaf count");
    std::vector<int64_t> so, base;
    if ((rc = global_segments(m_global, seg_off, nseg, so))) return rc;
    if ((rc = ensure_group_buffers(nseg, ctx->world))) return rc;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    segment_bases(m_global, so, nseg, ctx->world, base);
    int local = upload_base(base);
    if (local == GPSO_OK)
      local = sharded_local(ctx->rank, ctx->world, xs, xs_dtype, xs_mem, m_local, m_global, seg_off, nseg, varsigma);
    const std::string local_msg = ctx->err;
    if ((rc = publish_local(nseg, local))) return rc;
    if ((rc = exchange_winners(nseg, true))) return rc;
    int64_t who = -1;
    int verdict = GPSO_OK;
    if ((rc = finish_best(ovals2, nseg, 2, idx, mean, var, ucb, &verdict, &who))) return rc;
    return group_verdict(verdict, who, local, local_msg, "gpso_best_ucb_sharded");
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
    double* stage = ctx->pinned_stage((size_t)nseg * d_ * 2);
    if (!stage) return ctx->fail(GPSO_E_OOM, "pinned host staging");
    std::memcpy(stage, bounds, bb);
    HIPCHECK(hipMemcpyAsync(bdev, stage, bb, hipMemcpyHostToDevice, st()));
    launch_grow(st(), bdev, nseg, d_, depth, as<double>(leaves_raw));
    return launch_status();
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

  // One exploration level: the sub-tree centres of every box are generated on the device WITHOUT the rows
  // that repeat an earlier row bit for bit (a centre child repeats its parent: 1/3 of the reference's
  // list, gpso/param_space.py:186-200), scored, and the winner is reported by its REFERENCE row index --
  // the result is what gp_eval_best_ucb(leaf.grow(depth)) returns (grow.hip: grow_unique_kernel).
  int best_ucb_grow(const double* bounds, int nseg, int depth, double varsigma, int64_t* idx,
                    double* mean, double* var, double* ucb) override {
    ctx->tick_timing();
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident: call gpso_fit_eval / gpso_set_posterior first");
    int rc = precision_gate();
    if (rc) return rc;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    if ((rc = enqueue_best_grow(bounds, nseg, depth, 0, gpso_grow_rows(depth), varsigma))) return rc;
    rc = finish_best(ovals, nseg, 1, idx, mean, var, ucb);
    if (rc == 1) {  // the one-launch kernel met a near-duplicate it has no slot for: the sequence with the append counter
      one_refused = true;
      rc = enqueue_best_grow(bounds, nseg, depth, 0, gpso_grow_rows(depth), varsigma);
      one_refused = false;
      if (rc) return rc;
      rc = finish_best(ovals, nseg, 1, idx, mean, var, ucb);
    }
    return rc;
  }

  // The same on a group: every rank generates and scores the reference rows shard_range(rows, rank, world)
  // of every box (no leaf crosses a link: O(D) bytes in); the winners are all-gathered and folded on the
  // (ucb, reference row index) order.  Identical result on every rank, identical to gpso_best_ucb_grow.
  int best_ucb_grow_sharded(const double* bounds, int nseg, int depth, double varsigma, int64_t* idx,
                            double* mean, double* var, double* ucb) override {
    ctx->tick_timing();
    int rc = need_comm();
    if (rc) return rc;
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    if (depth < 0 || depth > 16) return ctx->fail(GPSO_E_ARG, "depth %d outside [0, 16]", depth);
    if ((rc = ensure_group_buffers(nseg, ctx->world))) return rc;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    const int local = sharded_local_grow(ctx->rank, ctx->world, bounds, nseg, depth, varsigma);
    const std::string local_msg = ctx->err;
    if ((rc = publish_local(nseg, local))) return rc;
    if ((rc = exchange_winners(nseg, false))) return rc;
    int64_t who = -1;
    int verdict = GPSO_OK;
    if ((rc = finish_best(ovals2, nseg, 2, idx, mean, var, ucb, &verdict, &who))) return rc;
    ctx->last_count[0] = -1;  // (the local live count stays on the device: no second read-back)
    return group_verdict(verdict, who, local, local_msg, "gpso_best_ucb_grow_sharded");
  }

  // ---- the two halves on their own, for an arbitrary (rank, world) and without a communicator: what a group of
  // `world` ranks computes can be replayed on ONE device, the all-gather replaced by the caller's concatenation ----
  int shard_winners(int rank, int world, const void* xs, int xs_dtype, int xs_mem, int64_t m_local, int64_t m_global,
                    const int64_t* seg_off, int nseg, double varsigma, double* payload) override {
    ctx->tick_timing();
    if (!payload) return ctx->fail(GPSO_E_ARG, "payload must not be NULL");
    if (world < 1 || rank < 0 || rank >= world) return ctx->fail(GPSO_E_ARG, "rank %d / world %d", rank, world);
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    int rc = ensure_group_buffers(nseg, world);
    if (rc) return rc;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    const int local = sharded_local(rank, world, xs, xs_dtype, xs_mem, m_local, m_global, seg_off, nseg, varsigma);
    return payload_out(nseg, local, payload);
  }
  int shard_winners_grow(int rank, int world, const double* bounds, int nseg, int depth, double varsigma,
                         double* payload) override {
    ctx->tick_timing();
    if (!payload) return ctx->fail(GPSO_E_ARG, "payload must not be NULL");
    if (world < 1 || rank < 0 || rank >= world) return ctx->fail(GPSO_E_ARG, "rank %d / world %d", rank, world);
    if (nseg < 1) return ctx->fail(GPSO_E_ARG, "nseg must be >= 1");
    int rc = ensure_group_buffers(nseg, world);
    if (rc) return rc;
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], st()));
    const int local = sharded_local_grow(rank, world, bounds, nseg, depth, varsigma);
    return payload_out(nseg, local, payload);
  }
  int payload_out(int nseg, int local, double* payload) {
    one_pending = false;
    host_direct = nullptr;
    call_token = 0.0;
    const std::string local_msg = ctx->err;
    int rc = publish_local(nseg, local);
    if (rc) return rc;
    const size_t pd = (size_t)group_payload_doubles(nseg);
    HIPCHECK(hipMemcpyAsync(payload, ovals.p, pd * 8, hipMemcpyDeviceToHost, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    collect_tile_ms();
    ctx->err = local_msg;
    return local;  // (the payload carries the same status in its last slot)
  }
  // gathered [world][group_payload_doubles(nseg)] (host, rank order) -> the group's result.  m_global >= 0: payloads
  // of gpso_shard_winners (indices relative to local segment pieces: the bases are added); < 0: of
  // gpso_shard_winners_grow (global reference row indices)
  int fold_winners(const double* gathered, int world, int64_t m_global, const int64_t* seg_off, int nseg, int64_t* idx,
                   double* mean, double* var, double* ucb) override {
    ctx->tick_timing();
    if (!gathered) return ctx->fail(GPSO_E_ARG, "gathered must not be NULL");
    if (world < 1 || nseg < 1) return ctx->fail(GPSO_E_ARG, "world %d / nseg %d", world, nseg);
    int rc = ensure_group_buffers(nseg, world);
    if (rc) return rc;
    hipStream_t s = st();
    if (ctx->timing) HIPCHECK(hipEventRecord(ctx->ev[2], s));
    const bool with_base = m_global >= 0;
    if (with_base) {
      std::vector<int64_t> so, base;
      if ((rc = global_segments(m_global, seg_off, nseg, so))) return rc;
      segment_bases(m_global, so, nseg, world, base);
      if ((rc = upload_base(base))) return rc;
    }
    const int pd = group_payload_doubles(nseg);
    HIPCHECK(hipMemcpyAsync(gath.p, gathered, (size_t)world * pd * 8, hipMemcpyHostToDevice, s));
    host_direct = result_host((size_t)nseg * 4 + 2);
    launch_reduce_winners(s, as<double>(gath), with_base ? as<int64_t>(wbase) : nullptr, world, nseg, pd,
                          as<double>(ovals2), host_direct, arm_token(nseg <= 64, nseg));
    if ((rc = launch_status())) return rc;
    int64_t who = -1;
    int verdict = GPSO_OK;
    if ((rc = finish_best(ovals2, nseg, 2, idx, mean, var, ucb, &verdict, &who))) return rc;
    if (verdict != GPSO_OK)
      return ctx->fail(verdict, "gpso_fold_winners: the payload of rank %lld carries status %d", (long long)who, verdict);
    return GPSO_OK;
  }

  // Make the posterior resident on `root` resident on every rank of the group: a 48-byte header (shape,
  // arithmetic options and the root's status), checked on every rank and agreed with an all-reduce(min) so that
  // either all ranks go on or all return the same code; the receivers' allocation is agreed the same way; then
  // ONE RCCL broadcast of the contiguous range of the posterior arena this posterior uses (posterior_span), straight
  // out of / into the library's device memory, on the context's stream.  A posterior that fails the precision self-test on the root is NOT sent:
  // every rank returns GPSO_E_PRECISION (the root is the only rank that can test it -- it holds the targets).
  int agree_min(int64_t* hd_slot, int64_t* host_slot, int64_t mine, int64_t* agreed) {
    RcclApi& R = RcclApi::get();
    hipStream_t s = st();
    *host_slot = mine;
    HIPCHECK(hipMemcpyAsync(hd_slot, host_slot, 8, hipMemcpyHostToDevice, s));
    RCCLCHECK(R.AllReduce(hd_slot, hd_slot, 1, ncclInt64, ncclMin, ctx->comm, s));
    HIPCHECK(hipMemcpyAsync(host_slot, hd_slot, 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    *agreed = *host_slot;
    return GPSO_OK;
  }
  int broadcast_posterior(int root) override {
    int rc = need_comm();
    if (rc) return rc;
    RcclApi& R = RcclApi::get();
    if (root < 0 || root >= ctx->world) return ctx->fail(GPSO_E_ARG, "root %d outside the group of %d", root, ctx->world);
    hipStream_t s = st();
    const bool is_root = ctx->rank == root;
    if ((rc = ensure(bhdr, 128))) return rc;
    int64_t* hd = as<int64_t>(bhdr);
    int64_t* host = reinterpret_cast<int64_t*>(ctx->pinned_scratch(8));
    if (!host) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    int64_t mine = GPSO_OK;  // this rank's status of the call so far
    std::string why;
    if (is_root) {
      // still take part in the collectives below when something is wrong: every rank must leave together
      if (!have_post) {
        mine = ctx->fail(GPSO_E_STATE, "gpso_broadcast_posterior: the root has no posterior resident");
      } else if ((rc = decide_generation()) != GPSO_OK) {
        mine = rc;
      } else if (check && st_have && have_data) {
        if ((rc = selftest_with_fallback()) != GPSO_OK) mine = rc;  // (settles GPSO_MATH_AUTO)
        else if (!st_pass()) mine = precision_gate();                // GPSO_E_PRECISION with the measured errors
      }
      why = ctx->err;
    }
    // the predict math the posterior travels with (under GPSO_MATH_AUTO the root's self-test has chosen)
    // (bit 12: GPSO_MATH_AUTO -- its split buffer has room for three planes whatever the rung, so the option itself must agree)
    const int64_t my_opts = (int64_t)(math_native_fallback ? GPSO_MATH_NATIVE : math) | ((int64_t)math_auto << 12) | ((int64_t)ctx->dtype << 16);
    // the ONE range of the root's posterior arena that travels (offset and length are the same on every rank: the
    // arena's layout depends on the shape and the types only)
    void* span_ptr = nullptr;
    int64_t span_off = 0, span_bytes = 0;
    if (is_root && mine == GPSO_OK && (rc = posterior_span(&span_ptr, &span_off, &span_bytes)) != GPSO_OK) {
      mine = rc;
      why = ctx->err;
    }
    if (is_root) {
      host[0] = n; host[1] = d; host[2] = my_opts; host[3] = mine; host[4] = host[5] = 0; host[6] = span_off; host[7] = span_bytes;
      HIPCHECK(hipMemcpyAsync(hd, host, 64, hipMemcpyHostToDevice, s));
    }
    RCCLCHECK(R.Broadcast(hd, hd, 64, ncclChar, root, ctx->comm, s));
    HIPCHECK(hipMemcpyAsync(host, hd, 64, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    const int64_t rn = host[0], rd = host[1], ropts = host[2], root_status = host[3];
    span_off = host[6];
    span_bytes = host[7];
    if (!is_root && root_status == GPSO_OK) {
      const int64_t rmath = ropts & 0xfff, rauto = (ropts >> 12) & 1, rdtype = ropts >> 16;
      if (rdtype == (int64_t)ctx->dtype && math_auto && rauto &&
          (rmath == GPSO_MATH_F16X3 || rmath == GPSO_MATH_BF16X6 || rmath == GPSO_MATH_NATIVE)) {
        // the root's choice (its self-test ruled): the rung of the ladder, or the f32 MFMA kernel
        math_native_fallback = rmath == GPSO_MATH_NATIVE && kFloatPredict;
        if (rmath != GPSO_MATH_NATIVE) math = (int)rmath;
      } else if (rdtype != (int64_t)ctx->dtype || rmath != math || rauto != (int64_t)math_auto) {
        mine = ctx->fail(GPSO_E_ARG, "gpso_broadcast_posterior: dtype / predict math options differ from the root's");
        why = ctx->err;
      }
    }
    // agree: min over the ranks (slot 4 of the header block); the root's status counts on every rank
    int64_t agreed = 0;
    if ((rc = agree_min(hd + 4, host + 4, std::min<int64_t>(mine, root_status), &agreed))) return rc;
    if (agreed != GPSO_OK) {
      if (mine != GPSO_OK) {
        ctx->err = why;
        return (int)mine;
      }
      return ctx->fail((int)agreed, "gpso_broadcast_posterior: %s (status %d); every rank returns together",
                       root_status != GPSO_OK ? "the root cannot send its posterior" : "another rank of the group cannot take the root's posterior",
                       (int)agreed);
    }
    // receivers allocate; agreed again, so that an allocation failure leaves no rank inside the broadcasts
    mine = is_root ? GPSO_OK : alloc_posterior(rn, (int)rd);
    if (mine == GPSO_OK && !is_root) mine = posterior_span_at(span_off, span_bytes, &span_ptr);
    why = ctx->err;
    if ((rc = agree_min(hd + 5, host + 5, mine, &agreed))) return rc;
    if (agreed != GPSO_OK) {
      if (mine != GPSO_OK) {
        ctx->err = why;
        return (int)mine;
      }
      return ctx->fail((int)agreed, "gpso_broadcast_posterior: a rank of the group could not allocate the posterior (status %d)", (int)agreed);
    }
    RCCLCHECK(R.Broadcast(span_ptr, span_ptr, (size_t)span_bytes, ncclChar, root, ctx->comm, s));
    ctx->last_count[0] = ctx->last_count[1] = span_bytes;  // (gpso_last_count after a broadcast: the bytes that travelled)
    if (!is_root) {
      if ((rc = adopt_posterior())) return rc;  // (synchronises the stream)
    } else {
      HIPCHECK(ctx->wait(s));
    }
    return posterior_mark_synced();  // every rank: the group holds these n rows in this math at this scale
  }

  // ---- the peers of a group brought up to date after gpso_append: only what the appends wrote travels -----------------------
  int posterior_mark_synced() override {
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident");
    float scale = 0.0f;
    int rc = current_split_scale(&scale);
    if (rc) return rc;
    sync_n = n;
    sync_math = math_in_use();
    sync_scale = scale;
    return GPSO_OK;
  }
  // the ranges (offsets into the posterior arena: the same on every context of this shape and type) that a peer holding the
  // posterior as of the last hand-off lacks; returns their number, 0 when the peer is up to date, and -- when rows do not
  // apply (another posterior since, another predict math, the fp16 scale crossed a power of two) -- ONE range: the whole span
  int posterior_dirty_ranges(int64_t* offsets, int64_t* nbytes, int cap) override {
    if (!offsets || !nbytes || cap < 10) return ctx->fail(GPSO_E_ARG, "need room for 10 ranges");
    void* span_ptr = nullptr;
    int64_t span_off = 0, span_bytes = 0;
    int rc = posterior_span(&span_ptr, &span_off, &span_bytes);  // (settles the arithmetic choices, builds pending pieces)
    if (rc) return rc;
    float scale = 0.0f;
    if ((rc = current_split_scale(&scale))) return rc;
    if (!rows_apply(scale)) {
      offsets[0] = span_off;
      nbytes[0] = span_bytes;
      return 1;
    }
    if (sync_n == n) return 0;
    std::vector<std::pair<int64_t, int64_t>> rr;
    rows_ranges(sync_n, rr);
    for (size_t i = 0; i < rr.size(); ++i) {
      offsets[i] = rr[i].first;
      nbytes[i] = rr[i].second;
    }
    return (int)rr.size();
  }
  // Collective.  Returns GPSO_OK when only rows travelled, 1 when the whole range did (any rank that cannot take rows -- it
  // never received this posterior, holds another shape / math / scale -- makes every rank take the whole), < 0 on error;
  // gpso_last_count(ctx, 0) = the bytes that travelled.  Same header / verdict protocol as gpso_broadcast_posterior.
  int broadcast_posterior_rows(int root) override {
    int rc = need_comm();
    if (rc) return rc;
    RcclApi& R = RcclApi::get();
    if (root < 0 || root >= ctx->world) return ctx->fail(GPSO_E_ARG, "root %d outside the group of %d", root, ctx->world);
    hipStream_t s = st();
    const bool is_root = ctx->rank == root;
    if ((rc = ensure(bhdr, 256))) return rc;
    int64_t* hd = as<int64_t>(bhdr);
    int64_t* host = reinterpret_cast<int64_t*>(ctx->pinned_scratch(32));
    if (!host) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    // the root says what it can offer: rows since n_base in (math, scale), or nothing but the whole range (n_base = -1; also
    // when anything is wrong on the root: the full call then reports it on every rank)
    int64_t n_base = -1, n_now = 0, mth = 0, scale_bits = 0;
    if (is_root) {
      void* span_ptr = nullptr;
      int64_t span_off = 0, span_bytes = 0;
      float scale = 0.0f;
      if (have_post && posterior_span(&span_ptr, &span_off, &span_bytes) == GPSO_OK && current_split_scale(&scale) == GPSO_OK &&
          rows_apply(scale) && !(check && st_have && have_data && !st_pass())) {
        n_base = sync_n;
        n_now = n;
        mth = math_in_use();
        scale_bits = (int64_t)__builtin_bit_cast(unsigned, scale);
      }
      host[0] = n_base; host[1] = n_now; host[2] = mth; host[3] = scale_bits; host[4] = npad; host[5] = dp; host[6] = host[7] = 0;
      HIPCHECK(hipMemcpyAsync(hd, host, 64, hipMemcpyHostToDevice, s));
    }
    RCCLCHECK(R.Broadcast(hd, hd, 64, ncclChar, root, ctx->comm, s));
    HIPCHECK(hipMemcpyAsync(host, hd, 64, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    n_base = host[0]; n_now = host[1]; mth = host[2]; scale_bits = host[3];
    const int64_t r_npad = host[4], r_dp = host[5];
    // can THIS rank take rows?  (the peers' record is what the last hand-off left on them)
    int64_t mine = 1;
    if (n_base < 0) mine = 0;
    else if (!is_root)
      mine = (have_post && arena.p != nullptr && sync_n == n_base && n == n_base && npad == r_npad && dp == r_dp && math_in_use() == (int)mth &&
              (int64_t)__builtin_bit_cast(unsigned, sync_scale) == scale_bits && n_now <= npad) ? 1 : 0;
    int64_t agreed = 0;
    if ((rc = agree_min(hd + 8, host + 8, mine, &agreed))) return rc;
    if (agreed != 1) {
      rc = broadcast_posterior(root);
      return rc == GPSO_OK ? 1 : rc;
    }
    int64_t moved = 0;
    if (n_now > n_base) {
      const int64_t n_keep = n;
      n = n_now;  // (rows_ranges describes the posterior of n_now rows)
      std::vector<std::pair<int64_t, int64_t>> rr;
      rows_ranges(n_base, rr);
      n = n_keep;
      char* base = static_cast<char*>(arena.p);
      RCCLCHECK(R.GroupStart());
      for (const auto& r : rr) {
        RCCLCHECK(R.Broadcast(base + r.first, base + r.first, (size_t)r.second, ncclChar, root, ctx->comm, s));
        moved += r.second;
      }
      RCCLCHECK(R.GroupEnd());
    }
    ctx->last_count[0] = ctx->last_count[1] = moved;
    if (!is_root) {
      n = n_now;
      if ((rc = adopt_posterior())) return rc;  // (reads the hyper block that just arrived; synchronises the stream)
    } else {
      HIPCHECK(ctx->wait(s));
    }
    return posterior_mark_synced();
  }

  // ------------------------------------------------------------------------------------------
  int get_matrix(int which, double* out) override {
    if (!out) return ctx->fail(GPSO_E_ARG, "out must not be NULL");
    const TF* src = nullptr;
    int lower = 1;
    switch (which) {
      case GPSO_MAT_CHOL:
        if (!chol_valid) return ctx->fail(GPSO_E_STATE, "no factor resident");
        src = as<TF>(Lf);
        break;
      case GPSO_MAT_LINV:
        if (!chol_valid) return ctx->fail(GPSO_E_STATE, "no factor resident");
        src = as<TF>(linv);
        break;
      case GPSO_MAT_KINV:
        if (!have_kinv) return ctx->fail(GPSO_E_STATE, "Kinv only exists after gpso_fit_eval with grad");
        src = as<TF>(kinvb);
        lower = 2;
        break;
      default:
        return ctx->fail(GPSO_E_ARG, "unknown matrix id %d", which);
    }
    int rc = ensure(getter_tmp, (size_t)n * n * 8);
    if (rc) return rc;
    launch_convert_out<TF>(st(), src, npad, as<double>(getter_tmp), n, n, lower);
    HIPCHECK(hipMemcpyAsync(out, getter_tmp.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    return GPSO_OK;
  }

  int get_vector(int which, double* out) override {
    if (!out) return ctx->fail(GPSO_E_ARG, "out must not be NULL");
    if (!have_post || !chol_valid) return ctx->fail(GPSO_E_STATE, "no fitted posterior resident");
    const TF* src = (which == GPSO_VEC_ALPHA) ? as<TF>(alpha_f) : (which == GPSO_VEC_WHITE) ? as<TF>(white) : nullptr;
    if (!src) return ctx->fail(GPSO_E_ARG, "unknown vector id %d", which);
    int rc = ensure(getter_tmp, (size_t)n * 8);
    if (rc) return rc;
    launch_convert_out<TF>(st(), src, npad, as<double>(getter_tmp), 1, n, 0);
    HIPCHECK(hipMemcpyAsync(out, getter_tmp.p, (size_t)n * 8, hipMemcpyDeviceToHost, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    return GPSO_OK;
  }

  // hyper | packed L^-1 | scaled inputs (double: plain, MFMA fragments, norms) | alpha [| bf16 pieces].
  // The generation inputs always travel in double; a receiver derives the float copies itself when the
  // sender's choice (slot 7 of the hyper block, written here: 1 = float generation, + 2 = GPSO_MATH_AUTO
  // settled on the f32 MFMA kernel for this posterior) is float generation.
  // settle the posterior's arithmetic choices (generation, GPSO_MATH_AUTO's rung) and write them into slot 7 of the
  // hyper block, which travels: 1 = float generation, 2 = GPSO_MATH_AUTO settled on the f32 MFMA kernel, 4 = the split
  // pieces are built, 8 = the packed L^-1 travels too, 16 = float generation keeps the f32 contraction, 256 x the predict math
  // (span_only: the flag describes the contiguous range of posterior_span -- the packed L^-1 is part of it only when the
  // posterior runs the f32 / f64 MFMA kernel; otherwise every buffer travels)
  int settle_and_flag(bool span_only) {
    if (!have_post) return GPSO_OK;
    int rc = decide_generation();
    if (rc) return rc;
    if (check && st_have && have_data && (rc = selftest_with_fallback())) return rc;  // settles GPSO_MATH_AUTO
    const bool with_linv_p = !span_only || math_in_use() == GPSO_MATH_NATIVE;
    if (with_linv_p && (rc = ensure_linv_p())) return rc;
    double* flag = ctx->pinned_scratch(256) + 120;  // (a slot neither the read-backs nor set_theta use)
    *flag = (gen_double() ? 0.0 : 1.0) + (math_native_fallback ? 2.0 : 0.0) + ((bf16_usable() && linv_b_valid) ? 4.0 : 0.0) +
            ((with_linv_p && linv_p_valid) ? 8.0 : 0.0) + (c16_fallback ? 16.0 : 0.0) + 256.0 * math;  // (which split the pieces are: a receiver under GPSO_MATH_AUTO follows)
    HIPCHECK(hipMemcpyAsync(as<double>(hyper) + 7, flag, 8, hipMemcpyHostToDevice, st()));
    HIPCHECK(hipStreamSynchronize(st()));  // callers copy these buffers on streams of their own
    return GPSO_OK;
  }
  int posterior_buffers(void** ptrs, int64_t* nbytes, int cap) override {
    if (npad == 0) return ctx->fail(GPSO_E_STATE, "no problem shape yet");
    if (cap < 7) return ctx->fail(GPSO_E_ARG, "need room for 7 buffers");
    int rc = settle_and_flag(false);
    if (rc) return rc;
    const size_t s = sizeof(TP);
    int k = 0;
    ptrs[k] = hyper.p;   nbytes[k++] = (int64_t)(kHyperHeader + kMaxD) * 8;
    ptrs[k] = linv_p.p;  nbytes[k++] = (int64_t)(packed_linv_elems(npad) * s);
    ptrs[k] = xs64.p;    nbytes[k++] = (int64_t)(npad * dp * 8);
    ptrs[k] = xs_p64.p;  nbytes[k++] = (int64_t)(npad * dp * 8);
    ptrs[k] = xnorm64.p; nbytes[k++] = (int64_t)(npad * 8);
    ptrs[k] = alpha.p;   nbytes[k++] = (int64_t)(npad * s);
    // the 16-bit pieces of L^-1 travel when the mode is on -- a sender whose pieces were never built (a posterior it
    // adopted itself, math switched afterwards) still sends the buffer, so that both sides of a hand-off list the
    // same buffers, and says so in the hyper block (bit 4 of slot 7 clear): the receivers then run the f32 kernel
    if (bf16_usable()) {
      ptrs[k] = linv_b.p; nbytes[k++] = (int64_t)split_bytes();
    }
    return k;
  }
  // The ONE contiguous range of the posterior arena the resident posterior uses (what gpso_broadcast_posterior sends):
  // *offset = its distance from the arena's start -- the same on every context of the same shape and type.
  int posterior_span(void** ptr, int64_t* offset, int64_t* nbytes) override {
    if (npad == 0 || arena.p == nullptr) return ctx->fail(GPSO_E_STATE, "no problem shape yet");
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident");
    int rc = settle_and_flag(true);
    if (rc) return rc;
    if (math_in_use() == GPSO_MATH_NATIVE && !linv_p_valid)
      return ctx->fail(GPSO_E_STATE, "this context received its posterior without the packed L^-1 and cannot pass it on for the f32 kernel");
    size_t off, bytes;
    posterior_range(&off, &bytes);
    if (ptr) *ptr = static_cast<char*>(arena.p) + off;
    if (offset) *offset = (int64_t)off;
    if (nbytes) *nbytes = (int64_t)bytes;
    return GPSO_OK;
  }
  // where a span (offset, nbytes) announced by a peer lands in THIS context's arena (after gpso_alloc_posterior)
  int posterior_span_at(int64_t offset, int64_t nbytes, void** ptr) override {
    if (arena.p == nullptr) return ctx->fail(GPSO_E_STATE, "gpso_alloc_posterior first");
    if (offset < 0 || nbytes <= 0 || offset % 256 != 0 || (size_t)offset + (size_t)nbytes > arena.bytes)
      return ctx->fail(GPSO_E_ARG, "posterior span [%lld, +%lld) does not fit this context's arena of %zu bytes (different "
                       "dtype / shape on the sender?)", (long long)offset, (long long)nbytes, arena.bytes);
    *ptr = static_cast<char*>(arena.p) + offset;
    return GPSO_OK;
  }
  // 64-bit fingerprint of the predict-ready posterior (the buffers gpso_posterior_buffers lists, the parts in use):
  // two contexts that ran the same deterministic fit hold the same value -- what a group that REPLICATES the fit on
  // every rank compares instead of broadcasting ("measure both")
  DevBuf hash_out;
  int posterior_hash(uint64_t* out) override {
    if (!have_post) return ctx->fail(GPSO_E_STATE, "no posterior resident");
    int rc = settle_and_flag(true);
    if (rc) return rc;
    if ((rc = ensure(hash_out, 8))) return rc;
    hipStream_t s = st();
    HIPCHECK(hipMemsetAsync(hash_out.p, 0, 8, s));
    const size_t sp = sizeof(TP);
    uint64_t salt = 1;
    auto add = [&](const void* p_, size_t bytes) { launch_hash_words(s, p_, bytes / 8, salt++, static_cast<unsigned long long*>(hash_out.p)); };
    add(hyper.p, (size_t)(kHyperHeader + kMaxD) * 8);
    add(xs64.p, (size_t)npad * dp * 8);
    add(xs_p64.p, (size_t)npad * dp * 8);
    add(xnorm64.p, (size_t)npad * 8);
    add(alpha.p, (size_t)npad * sp / 8 * 8);
    if (math_in_use() != GPSO_MATH_NATIVE) {
      if (f16_split()) add(f16_scale(), 8);
      add(split_planes(), (size_t)nsplit() * npad * npad * 2);
    } else {
      add(linv_p.p, packed_linv_elems(npad) * sp);
    }
    uint64_t* host = reinterpret_cast<uint64_t*>(ctx->pinned_scratch(8));
    if (!host) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    HIPCHECK(hipMemcpyAsync(host, hash_out.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(ctx->wait(s));
    if ((rc = launch_status())) return rc;
    *out = host[0];
    return GPSO_OK;
  }

  int alloc_posterior(int64_t n_, int d_) override {
    int rc = refuse_if_async("gpso_alloc_posterior");
    if (rc) return rc;
    rc = shape(n_, d_);
    if (rc) return rc;
    have_data = have_post = have_kinv = chol_valid = linv_p_valid = false;
    st_done = st_have = false;
    forget_peers();
    return GPSO_OK;
  }

  int adopt_posterior() override {
    if (npad == 0) return ctx->fail(GPSO_E_STATE, "gpso_alloc_posterior first");
    double* h = ctx->pinned_scratch(kHyperHeader + kMaxD);
    if (!h) return ctx->fail(GPSO_E_OOM, "pinned host scratch");
    HIPCHECK(hipMemcpyAsync(h, hyper.p, (size_t)(kHyperHeader + kMaxD) * 8, hipMemcpyDeviceToHost, st()));
    HIPCHECK(hipStreamSynchronize(st()));
    // (a posterior EXTENDED by gpso_append on the sender arrives with more rows than the buffers were announced for: the
    // layout is a function of the padded size, which an in-place append cannot change)
    const int64_t hn = (int64_t)h[0];
    if ((int)h[1] != d || hn < n || hn > npad)
      return ctx->fail(GPSO_E_ARG, "received posterior is for n=%lld d=%d, buffers were sized for n=%lld d=%d",
                       (long long)h[0], (int)h[1], (long long)n, d);
    n = hn;
    kp.kernel = (int)h[2];
    n_ls = (int)h[3];
    kp.variance = h[4];
    kp.noise = h[5];
    kp.mean_c = h[6];
    ls_host.assign(h + kHyperHeader, h + kHyperHeader + n_ls);
    have_post = true;
    chol_valid = have_kinv = false;
    small_tile_rows = 8;           // linv_p came from elsewhere
    st_done = st_have = false;     // the fitting rank ran the self-test; no targets here
    // generation arithmetic and predict math: the sender's choice (its self-test ruled), unless this context insists
    const int sender = (int)h[7];
    math_native_fallback = math_auto && (sender & 2) != 0;
    const int sender_math = sender >> 8;
    if (math_auto && kFloatPredict && (sender_math == GPSO_MATH_F16X3 || sender_math == GPSO_MATH_BF16X6)) math = sender_math;
    // the split pieces the sender actually built travel with it (and are the split this context runs)
    linv_b_valid = bf16_usable() && (sender & 4) != 0 && sender_math == math;
    linv_b_pending = false;
    linv_p_valid = (sender & 8) != 0;
    gen_eff32 = kFloatPredict && (gen_mode == GPSO_GEN_F32 || (gen_mode == GPSO_GEN_AUTO && (sender & 1) != 0));
    gen_decided = true;
    gen32_inputs_ok = false;
    c16_fallback = (sender & 16) != 0;  // (the sender's float generation kept the f32 contraction: same arithmetic here)
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
  if (dtype != GPSO_F64 && dtype != GPSO_F32 && dtype != GPSO_MIXED) {
    g_create_error = "dtype must be GPSO_F64, GPSO_F32 or GPSO_MIXED";
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
  DeviceGuard guard(device);  // the caller's current device is restored on return
  if (!guard.ok) {
    g_create_error = "hipSetDevice failed";
    return GPSO_E_HIP;
  }
  gpso_ctx* ctx = new gpso_ctx();
  ctx->device = device;
  ctx->dtype = dtype;
  ContextPool::Set pooled;
  if (ContextPool::get().take(device, pooled)) {  // a destroyed context's stream, events and pinned buffers
    ctx->own_stream = pooled.stream;
    ctx->ev_wait = pooled.ev_wait;
    for (int i = 0; i < 8; ++i) ctx->ev[i] = pooled.ev[i];
    ctx->pinned = pooled.pinned;
    ctx->pinned_doubles = pooled.pinned_doubles;
    ctx->stage = pooled.stage;
    ctx->stage_doubles = pooled.stage_doubles;
    if (ctx->pinned) std::memset(ctx->pinned, 0, ctx->pinned_doubles * 8);
    ctx->stream = ctx->own_stream;
  } else {
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
  }
  if (dtype == GPSO_F64)
    ctx->eng = new EngineT<double, double>(ctx);
  else if (dtype == GPSO_F32)
    ctx->eng = new EngineT<float, float>(ctx);
  else
    ctx->eng = new EngineT<double, float>(ctx);
  *out = ctx;
  return GPSO_OK;
}

void gpso_destroy(gpso_ctx* ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm && !ctx->comm_aborted) (void)RcclApi::get().CommDestroy(ctx->comm);
  delete ctx->eng;
  for (auto& ev : ctx->tile_ev) (void)hipEventDestroy(ev);
  if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
  if (ctx->inv_stream) (void)hipStreamSynchronize(ctx->inv_stream);
  for (hipEvent_t e : {ctx->ev_col, ctx->ev_chain, ctx->ev_panel, ctx->ev_inv})
    if (e) (void)hipEventDestroy(e);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->inv_stream) (void)hipStreamDestroy(ctx->inv_stream);
  if (ctx->slot_host) (void)hipHostFree(ctx->slot_host);
  for (auto& ev : ctx->slot_ev)
    if (ev) (void)hipEventDestroy(ev);
  // stream, events and pinned buffers go to the pool when they are complete and the pool has room (ContextPool)
  bool complete = ctx->own_stream != nullptr && ctx->ev_wait != nullptr;
  for (auto& ev : ctx->ev) complete = complete && ev != nullptr;
  ContextPool::Set give;
  if (complete) {
    (void)hipStreamSynchronize(ctx->own_stream);
    give.stream = ctx->own_stream;
    give.ev_wait = ctx->ev_wait;
    for (int i = 0; i < 8; ++i) give.ev[i] = ctx->ev[i];
    give.pinned = ctx->pinned;
    give.pinned_doubles = ctx->pinned_doubles;
    give.stage = ctx->stage;
    give.stage_doubles = ctx->stage_doubles;
  }
  if (!complete || !ContextPool::get().give(ctx->device, give)) {
    for (auto& ev : ctx->ev)
      if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_wait) (void)hipEventDestroy(ctx->ev_wait);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  }
  delete ctx;
}

const char* gpso_last_error(const gpso_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

// every entry point runs on the context's device and puts the caller's current device back on return
#define ENTER()                                                              \
  if (!ctx) return GPSO_E_ARG;                                               \
  DeviceGuard guard_(ctx->device);                                           \
  if (!guard_.ok) return ctx->fail(GPSO_E_HIP, "hipSetDevice(%d) failed", ctx->device); \
  gpso::g_launch_error.clear();

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

int gpso_set_option_f64(gpso_ctx* ctx, int option, double value) {
  ENTER();
  return ctx->eng->set_option_f64(option, value);
}

int gpso_wait_stream(gpso_ctx* ctx, void* producer_stream) {
  ENTER();
  hipStream_t prod = static_cast<hipStream_t>(producer_stream);
  if (prod == ctx->stream) return GPSO_OK;
  if (hipStreamQuery(prod) == hipSuccess) return GPSO_OK;  // nothing pending there: nothing to wait for (one call, no event)
  HIPCHECK(hipEventRecord(ctx->ev_wait, prod));
  HIPCHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_wait, 0));
  return GPSO_OK;
}

int gpso_precision_info(gpso_ctx* ctx, double* out) {
  ENTER();
  if (!out) return ctx->fail(GPSO_E_ARG, "out must not be NULL");
  return ctx->eng->precision_info(out);
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

// GPflow-2's parameter transforms, bit for bit what numpy computes for them on the host
// side of the reference's optimiser loop: softplus(u) = logaddexp(0, u) in numpy's own case split (libm log1p / exp),
// sigmoid(u) = (1 + tanh(u / 2)) / 2.
static double gpso_softplus(double u) {
  if (u == 0.0) return 0.693147180559945309417232121458176568;  // log 2
  if (u < 0.0) return 0.0 + std::log1p(std::exp(u));
  if (u > 0.0) return u + std::log1p(std::exp(-u));
  return u;  // NaN
}
static double gpso_sigmoid(double u) { return 0.5 * (1.0 + std::tanh(0.5 * u)); }

int gpso_fit_eval_u(gpso_ctx* ctx, int kernel, const double* u, int n_ls, int train_mean, double mean_c_fixed,
                    double* nlml, double* grad_u, double* theta_out) {
  ENTER();
  if (!u) return ctx->fail(GPSO_E_ARG, "u must not be NULL");
  if (n_ls < 1 || n_ls > kGradMaxLs) return ctx->fail(GPSO_E_ARG, "n_ls=%d outside [1, %d]", n_ls, kGradMaxLs);
  double ls[kGradMaxLs], g[kGradMaxLs + 3];
  for (int k = 0; k < n_ls; ++k) ls[k] = gpso_softplus(u[k]);
  const double variance = gpso_softplus(u[n_ls]);
  const double noise = 1.0e-6 + gpso_softplus(u[n_ls + 1]);  // gpflow.likelihoods.Gaussian DEFAULT_VARIANCE_LOWER_BOUND
  const double mean_c = train_mean ? u[n_ls + 2] : mean_c_fixed;
  if (theta_out) {
    for (int k = 0; k < n_ls; ++k) theta_out[k] = ls[k];
    theta_out[n_ls] = variance;
    theta_out[n_ls + 1] = noise;
    theta_out[n_ls + 2] = mean_c;
  }
  const int rc = ctx->eng->fit_eval(kernel, ls, n_ls, variance, noise, mean_c, nlml, grad_u ? g : nullptr);
  if (rc != GPSO_OK || !grad_u) return rc;
  // chain rule: d/du = d/dtheta * sigmoid(u) for the softplus-transformed parameters, identity for the mean
  for (int k = 0; k < n_ls + 2; ++k) grad_u[k] = g[k] * gpso_sigmoid(u[k]);
  if (train_mean) grad_u[n_ls + 2] = g[n_ls + 2];
  return GPSO_OK;
}

int gpso_append(gpso_ctx* ctx, const double* Xnew, const double* ynew, int64_t k, double* nlml) {
  ENTER();
  return ctx->eng->append(Xnew, ynew, k, nlml);
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

int gpso_best_ucb_begin(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m, const int64_t* seg_off, int nseg,
                        double varsigma) {
  ENTER();
  return ctx->eng->best_ucb_begin(xs, xs_dtype, xs_mem, m, seg_off, nseg, varsigma, nullptr, 0);
}

int gpso_best_ucb_grow_begin(gpso_ctx* ctx, const double* bounds, int nseg, int depth, double varsigma) {
  ENTER();
  if (!bounds) return ctx->fail(GPSO_E_ARG, "bounds must not be NULL");
  return ctx->eng->best_ucb_begin(nullptr, GPSO_F64, GPSO_MEM_DEVICE, 0, nullptr, nseg, varsigma, bounds, depth);
}

int gpso_best_ucb_end(gpso_ctx* ctx, int ticket, int64_t* idx, double* mean, double* var, double* ucb) {
  ENTER();
  return ctx->eng->best_ucb_end(ticket, idx, mean, var, ucb);
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

int gpso_problem_shape(const gpso_ctx* ctx, int64_t* n, int* d) {
  if (!ctx) return GPSO_E_ARG;
  int64_t nn = 0;
  int dd = 0;
  ctx->eng->problem_shape(&nn, &dd);
  if (n) *n = nn;
  if (d) *d = dd;
  return GPSO_OK;
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

int gpso_posterior_span(gpso_ctx* ctx, void** ptr, int64_t* offset, int64_t* nbytes) {
  ENTER();
  return ctx->eng->posterior_span(ptr, offset, nbytes);
}

int gpso_posterior_span_at(gpso_ctx* ctx, int64_t offset, int64_t nbytes, void** ptr) {
  ENTER();
  if (!ptr) return ctx->fail(GPSO_E_ARG, "ptr must not be NULL");
  return ctx->eng->posterior_span_at(offset, nbytes, ptr);
}

int gpso_posterior_hash(gpso_ctx* ctx, uint64_t* out) {
  ENTER();
  if (!out) return ctx->fail(GPSO_E_ARG, "out must not be NULL");
  return ctx->eng->posterior_hash(out);
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
  if (!ctx || what < 0 || what > 3) return -1.0;
  return ctx->last_ms[what];
}

// ---- multi-GPU group --------------------------------------------------------------------------
int gpso_comm_unique_id(void* out) {
  if (!out) return GPSO_E_ARG;
  RcclApi& R = RcclApi::get();
  if (!R.ok) {
    g_create_error = R.load_error;
    return GPSO_E_RCCL;
  }
  ncclUniqueId id;
  const ncclResult_t r = R.GetUniqueId(&id);
  if (r != ncclSuccess) {
    g_create_error = std::string("ncclGetUniqueId: ") + R.GetErrorString(r);
    return GPSO_E_RCCL;
  }
  static_assert(sizeof(id) == GPSO_UNIQUE_ID_BYTES, "unique id size");
  std::memcpy(out, &id, sizeof(id));
  return GPSO_OK;
}

int gpso_comm_init(gpso_ctx* ctx, int rank, int world, const void* unique_id) {
  ENTER();
  if (!unique_id) return ctx->fail(GPSO_E_ARG, "unique_id must not be NULL");
  if (world < 1 || rank < 0 || rank >= world) return ctx->fail(GPSO_E_ARG, "rank %d / world %d", rank, world);
  if (ctx->comm) return ctx->fail(GPSO_E_STATE, "the context already belongs to a group%s (gpso_comm_destroy first)",
                                  ctx->comm_aborted.load() ? " whose communicator was aborted" : "");
  RcclApi& R = RcclApi::get();
  if (!R.ok) return ctx->fail(GPSO_E_RCCL, "%s", R.load_error.c_str());
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  RCCLCHECK(R.CommInitRank(&ctx->comm, world, id, rank));
  ctx->rank = rank;
  ctx->world = world;
  return GPSO_OK;
}

int gpso_comm_destroy(gpso_ctx* ctx) {
  ENTER();
  if (ctx->comm && !ctx->comm_aborted) {
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    RCCLCHECK(RcclApi::get().CommDestroy(ctx->comm));
  }
  ctx->comm = nullptr;
  ctx->comm_aborted = false;
  ctx->rank = 0;
  ctx->world = 1;
  return GPSO_OK;
}

int gpso_comm_info(const gpso_ctx* ctx, int* rank, int* world) {
  if (!ctx) return GPSO_E_ARG;
  if (rank) *rank = ctx->rank;
  if (world) *world = ctx->world;
  return ctx->comm ? 1 : 0;
}

void gpso_shard_range(int64_t m, int rank, int world, int64_t* lo, int64_t* hi) {
  int64_t a = 0, b = 0;
  if (world >= 1 && rank >= 0 && rank < world && m >= 0) shard_range(m, rank, world, &a, &b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

int gpso_broadcast_posterior(gpso_ctx* ctx, int root) {
  ENTER();
  return ctx->eng->broadcast_posterior(root);
}

int gpso_broadcast_posterior_rows(gpso_ctx* ctx, int root) {
  ENTER();
  return ctx->eng->broadcast_posterior_rows(root);
}

int gpso_posterior_dirty_ranges(gpso_ctx* ctx, int64_t* offsets, int64_t* nbytes, int cap) {
  ENTER();
  return ctx->eng->posterior_dirty_ranges(offsets, nbytes, cap);
}

int gpso_posterior_mark_synced(gpso_ctx* ctx) {
  ENTER();
  return ctx->eng->posterior_mark_synced();
}

int gpso_best_ucb_sharded(gpso_ctx* ctx, const void* xs, int xs_dtype, int xs_mem, int64_t m_local,
                          int64_t m_global, const int64_t* seg_off, int nseg, double varsigma, int64_t* idx,
                          double* mean, double* var, double* ucb) {
  ENTER();
  return ctx->eng->best_ucb_sharded(xs, xs_dtype, xs_mem, m_local, m_global, seg_off, nseg, varsigma, idx, mean,
                                    var, ucb);
}

int gpso_best_ucb_grow_sharded(gpso_ctx* ctx, const double* bounds, int nseg, int depth, double varsigma,
                               int64_t* idx, double* mean, double* var, double* ucb) {
  ENTER();
  return ctx->eng->best_ucb_grow_sharded(bounds, nseg, depth, varsigma, idx, mean, var, ucb);
}

int gpso_shard_winners(gpso_ctx* ctx, int rank, int world, const void* xs, int xs_dtype, int xs_mem,
                       int64_t m_local, int64_t m_global, const int64_t* seg_off, int nseg, double varsigma,
                       double* payload) {
  ENTER();
  return ctx->eng->shard_winners(rank, world, xs, xs_dtype, xs_mem, m_local, m_global, seg_off, nseg, varsigma, payload);
}

int gpso_shard_winners_grow(gpso_ctx* ctx, int rank, int world, const double* bounds, int nseg, int depth,
                            double varsigma, double* payload) {
  ENTER();
  return ctx->eng->shard_winners_grow(rank, world, bounds, nseg, depth, varsigma, payload);
}

int gpso_fold_winners(gpso_ctx* ctx, const double* gathered, int world, int64_t m_global, const int64_t* seg_off,
                      int nseg, int64_t* idx, double* mean, double* var, double* ucb) {
  ENTER();
  return ctx->eng->fold_winners(gathered, world, m_global, seg_off, nseg, idx, mean, var, ucb);
}

int gpso_group_payload_doubles(int nseg) { return nseg < 0 ? 0 : gpso::group_payload_doubles(nseg); }

// Callable from ANOTHER thread than the one blocked inside a group call of this context: it only touches the
// communicator (ncclCommAbort makes the collective kernels in flight exit, so the blocked call returns an error).
int gpso_comm_abort(gpso_ctx* ctx) {
  if (!ctx) return GPSO_E_ARG;
  ncclComm_t c = ctx->comm;
  if (c == nullptr) return GPSO_OK;
  RcclApi& R = RcclApi::get();
  if (!R.ok || R.CommAbort == nullptr) return GPSO_E_RCCL;
  if (ctx->comm_aborted.exchange(true)) return GPSO_OK;  // (already aborted: the handle is gone)
  return R.CommAbort(c) == ncclSuccess ? GPSO_OK : GPSO_E_RCCL;
}

int64_t gpso_last_count(gpso_ctx* ctx, int what) {
  if (!ctx || what < 0 || what > 3) return -1;
  if (what == 3) return gpso::g_leaf_last_splits;
  return ctx->last_count[what];
}

const char* gpso_version(void) { return "gpso-hip 0.5.0 (gfx950)"; }

}  // extern "C"
