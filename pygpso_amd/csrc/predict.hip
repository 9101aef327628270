// Leaf-UCB predict path (K7-K11 of SURVEY.md 2.4) for gfx950.
//
//   prep_leaves    raw leaf coordinates -> x*/lengthscale (T) + squared norms
//   leaf_tiles     the dominant kernel: for a tile of leaves and a block of BM rows of L^-1,
//                  generate the cross-Gram tile K_{n*} on the fly (MFMA for the x.x* contraction,
//                  VALU/transcendental epilogue for the Matern/SE map), feed it -- still in
//                  registers, in accumulator layout -- as the B operand of the MFMA that applies
//                  L^-1, and reduce |L^-1 k*|^2 (and k*.alpha on the diagonal block) per leaf.
//                  Nothing of size N x M ever touches HBM.
//   leaf_finalize  sum the per-row-block partials, form mean / var(+noise) / ucb = mean + vs*var
//   seg_argmax_*   first-max arg-max of ucb per segment (np.argmax tie rule)
//
// Replaces gpflow_model.predict_y + the numpy UCB/argmax of gpso/gp_surrogate.py:313-328.
//
// Compiled with -ffp-contract=off (Makefile): every fused multiply-add below is written out
// (fma_t), so all unrolled instances of the per-leaf arithmetic are the same instruction sequence
// and a leaf's result cannot depend on which column slot of a tile it lands in.
#include <cstdlib>

#include "common.hpp"
#include "grow_device.hpp"
#include "kernels.hpp"
#include "leaf_split.hpp"  // (LDS-DMA helpers, the fp16 pair split; the split kernels themselves: predict_split_*.hip)

namespace gpso {
int g_leaf_last_splits = 0;  // workgroups per leaf tile of the last split-kernel launch (gpso_last_count(ctx, 3))
int g_leaf_row_loop = 1;  // (leaf_split.hpp: a workgroup of the split predict kernels keeps its leaf tile and loops over row blocks; GPSO_OPT_ROW_LOOP)
int leaf_cu_count() {
  static int n = 0;  // (one device type per process: MI355X, 256)
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}


// Completion token (round 5): the LAST kernel of a call, where it is a single workgroup, writes a sequence number behind its
// records in pinned host memory; the host spins on that word instead of recording and polling an event (a few microseconds
// of every call; what a 50 us call of the optimiser's regime notices).  The caller has made every record store of the
// workgroup happen-before this one (one writer thread, or a system-scope fence per writer + a workgroup barrier).
__device__ __forceinline__ void publish_done(double* host_vals, int slot, double token) {
  if (host_vals == nullptr || token == 0.0) return;
  __threadfence_system();
  *reinterpret_cast<volatile double*>(host_vals + slot) = token;
}

// np.argmax semantics: first maximum wins; NaN counts as the maximum (first NaN wins)
struct Best {
  double u;
  int64_t i;
};
__device__ __forceinline__ bool better(const Best& a, const Best& b) {
  if (a.i < 0) return false;
  if (b.i < 0) return true;
  const bool an = a.u != a.u, bn = b.u != b.u;
  if (an != bn) return an;
  if (!an && a.u != b.u) return a.u > b.u;
  return a.i < b.i;
}

__device__ __forceinline__ bool leaf_norm_is_nan(const void* lnorm, int f64, int64_t j) {
  if (lnorm == nullptr) return false;
  if (f64) {
    const double q = static_cast<const double*>(lnorm)[j];
    return q != q;
  }
  const float q = static_cast<const float*>(lnorm)[j];
  return q != q;
}
// one leaf: sum the row blocks' partials, form mean / var (+ noise) / ucb, store them; returns ucb
__device__ __forceinline__ double finalize_leaf(const LeafFinalize& f, int64_t j) {
  double v = 0, mu = 0;
  for (int b = 0; b < f.nbi; ++b) {
    v += f.part_var[(int64_t)b * f.mpad + j];
    mu += f.part_mean[(int64_t)b * f.mpad + j];
  }
  if (leaf_norm_is_nan(f.lnorm, f.lnorm_f64, j)) v = mu = __builtin_nan("");  // (a NaN coordinate: NaN out, as the reference)
  // [gpflow base_conditional] fvar = k** - sum A^2 ; predict_y adds the noise variance
  const double vy = __dadd_rn(__dsub_rn(f.variance, v), f.noise);
  const double my = __dadd_rn(mu, f.mean_c);
  f.mean[j] = my;
  f.var[j] = vy;
  // gpso/gp_surrogate.py:326  ucb = mean + varsigma * var  (two roundings, as numpy does)
  double prod = f.varsigma * vy;
  asm volatile("" : "+v"(prod));  // keep the product rounded on its own: no fma contraction
  const double u = my + prod;
  f.ucb[j] = u;
  return u;
}


// ---------------------------------------------------------------------------------------------
// raw leaf coordinates -> x* / lengthscale in the generation type TG, + squared norms
template <typename TG, typename TIN>
__global__ __launch_bounds__(256) void prep_leaves_kernel(const TIN* __restrict__ xs, int64_t m,
                                                          int64_t mpad, int d, int dp,
                                                          const double* __restrict__ ls /*[dp]*/,
                                                          const int64_t* __restrict__ m_live,
                                                          TG* __restrict__ out, TG* __restrict__ norm) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= mpad) return;
  const bool live = j < m && (m_live == nullptr || j < *m_live);
  TG acc = 0;
  for (int k = 0; k < dp; ++k) {
    TG v = 0;
    if (live && k < d) v = (TG)((double)xs[j * d + k] / ls[k]);
    out[j * dp + k] = v;
    acc += v * v;
  }
  norm[j] = acc;
}


// LDS images written by global_load_lds: A fragments are "slabs" of 64 lanes x 16 B (a vec4 of
// double = two slabs); X fragments (one TG per lane) are the linear memory image, 64 x sizeof(TG)
// bytes moved as 256-byte pieces.
template <typename T>
struct V2Lds;
template <>
struct V2Lds<float> {
  static constexpr int SLABS = 1;
  static __device__ __forceinline__ f32x4 frag(const unsigned char* base, int lane) {
    return *reinterpret_cast<const f32x4*>(base + lane * 16);
  }
};
template <>
struct V2Lds<double> {
  static constexpr int SLABS = 2;
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ f64x4 frag(const unsigned char* base, int lane) {
    const f64x2 lo = *reinterpret_cast<const f64x2*>(base + lane * 16);
    const f64x2 hi = *reinterpret_cast<const f64x2*>(base + 1024 + lane * 16);
    return f64x4{lo[0], lo[1], hi[0], hi[1]};
  }
};
// one X fragment (64 x TG, contiguous in memory) -> LDS, 256 bytes per DMA instruction
template <typename TG>
__device__ __forceinline__ void glds_xfrag(const TG* src_frag, unsigned char* dst, int lane) {
#pragma unroll
  for (int w = 0; w < (int)sizeof(TG) / 4; ++w)
    glds4(reinterpret_cast<const unsigned*>(src_frag) + w * 64 + lane, dst + w * 256);
}

// address of tile (rt, kt <= rt) of the triangle-packed L^-1, in vec4 units
__device__ __forceinline__ size_t linv_tile(int rt, int kt) {
  return ((size_t)rt * (size_t)(rt + 1) / 2 + (size_t)kt) * 64;
}

#ifndef GPSO_PSTAMP
#define GPSO_PSTAMP(kt, i)  // tools/micro/leaf_phases.hip defines this to record s_memtime stamps
#endif
template <typename T, typename TG, int RT, int CT, int KERNEL, bool DIAG>
__device__ __forceinline__ void leaf_v2_step(
    int kt, int kt_diag0, bool gen, bool gen_diag, int lane, int dp4,
    const unsigned char* panel_b /* [RT][SLABS] KiB */, const unsigned char* xs_b /* [dp4] X fragments */,
    const TG* xb, const typename Mfma<TG>::vec4& na, const typename Mfma<T>::vec4* __restrict__ al4,
    const TG (&nb)[CT], T variance, typename Mfma<T>::vec4 (&acc)[RT][CT], T (&macc)[CT],
    typename Mfma<T>::vec4 (&p_cur)[CT]) {
  using M = Mfma<T>;
  using MG = Mfma<TG>;
  using vec4 = typename M::vec4;
  using vecG = typename MG::vec4;
  using L = V2Lds<T>;
  constexpr int E = CT * 4;
  constexpr int FB = L::SLABS * 1024;  // bytes of one A fragment in LDS
  constexpr int XB = 64 * (int)sizeof(TG);
  constexpr TG C2 = (TG)KernScale<KERNEL>::C2;
  // ---- generation MFMAs for k-tile kt + 1 (short dependent chains; issued ahead of the apply) --
  vecG s[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) s[t] = vecG{0, 0, 0, 0};
  if (gen) {
    for (int c = 0; c < dp4; ++c) {
      const TG xa = reinterpret_cast<const TG*>(xs_b + c * XB)[lane];
#pragma unroll
      for (int t = 0; t < CT; ++t) s[t] = MG::mma(xa, xb[(t * dp4 + c) * 64 + lane], s[t]);
    }
  }
  GPSO_PSTAMP(kt, 2);
  vec4 p_nxt[CT];
  // A operands from LDS, two row tiles ahead of their use: the reads are pinned in front of the
  // previous tile's MFMAs (sched_barrier), otherwise the scheduler sinks them to just before their
  // use and every row tile pays a full LDS round trip
  vec4 a[3];
  a[0] = L::frag(panel_b, lane);
  if (RT > 1) a[1] = L::frag(panel_b + FB, lane);
  // ---- apply k-tile kt, with the map of k-tile kt + 1 sliced between the MFMAs -----------------
  // The map runs in three stages per entry (sqrt | exp | polynomial, common.hpp), the 3 E stage-ops dealt
  // over the row tiles 1 .. RT-1 in stage-major order: an entry's stages are then E ops -- at least a row
  // tile's MFMAs -- apart, and no transcendental waits on the instruction right in front of it (the wave
  // issues in order: such a wait would hold back the MFMAs behind it as well).  Row tile 0 carries none:
  // the generation MFMAs that produce s are still in flight there.
  T mt[CT][4], me[CT][4];
  constexpr int OPS = 3 * E;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (rt + 2 < RT) a[(rt + 2) % 3] = L::frag(panel_b + (rt + 2) * FB, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DIAG && kt > kt_diag0 + rt)) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = M::mma(a[rt % 3][r], p_cur[t][r], acc[rt][t]);
    }
    constexpr int RS = (RT > 1) ? RT - 1 : 1;  // row tiles that carry stage-ops
    const int slot = (RT > 1) ? rt - 1 : 0;
    if (slot >= 0) {
#pragma unroll
      for (int o = slot * OPS / RS; o < (slot + 1) * OPS / RS; ++o) {
        const int stage = o / E, e = o % E, t = e >> 2, r = e & 3;
        if (stage == 0) {
          // u = C2 * r^2, GPflow's GEMM form r^2 = -2 x.x* + (|x|^2 + |x*|^2) combined in TG (norms
          // pre-scaled by C2), rounded to T for the map
          mt[t][r] = kern_stage1<KERNEL>((T)fma_t((TG)(TG(-2) * C2), s[t][r], na[r] + nb[t]));
        } else if (stage == 1) {
          me[t][r] = kern_stage2<KERNEL>(mt[t][r]);
        } else {
          p_nxt[t][r] = kern_stage3<KERNEL>(mt[t][r], me[t][r], variance);
        }
      }
    }
  }
  GPSO_PSTAMP(kt, 3);
  if (gen && gen_diag) {  // k-tile kt + 1 lies in the diagonal block: its share of k*.alpha
    const vec4 a4 = al4[(kt + 1) * 4 + (lane >> 4)];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        macc[t] = fma_t(p_nxt[t][r], a4[r], macc[t]);
        if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(macc[t]));  // no packed accumulation (see leaf_bf16_gen)
      }
  }
#pragma unroll
  for (int t = 0; t < CT; ++t) p_cur[t] = p_nxt[t];
}

template <typename T, typename TG, int CT>
inline size_t leaf_v2_lds_bytes(int rt, int dp4) {
  return (size_t)2 * rt * (sizeof(T) * 4 / 16) * 1024 + (size_t)2 * dp4 * 64 * sizeof(TG) +
         (size_t)4 * CT * dp4 * 64 * sizeof(TG);
}

// =============================================================================================
// leaf_tiles v2: the BM x 16 panel of L^-1 and the X fragments of the next k-tiles are brought
// into LDS by the whole workgroup with direct-to-LDS loads (global_load_lds: no VGPRs, one KiB per
// wave instruction, fragment-major source == linear LDS image), double buffered, one barrier per
// k-tile; every wave reads its A operands from LDS just in time; the cross-Gram tile of k-tile
// kt + 1 is generated WHILE the MFMAs of k-tile kt run (generation MFMAs first, the Matern / SE map
// sliced into the issue shadow of the apply MFMAs).
// m_live (nullable): device count of live leaves; workgroups whose first leaf is at or beyond it
// exit at once (on-device growth sizes its launch for the worst case, see grow.hip).
// =============================================================================================
// ---- ONE-LAUNCH small calls (round 4) ---------------------------------------------------------------------------------
// When a single row block covers all of L^-1 (N_pad = BM: the optimiser's regime, N <= 128 -- and N <= 256), a workgroup's
// leaves are finished inside the workgroup.  leaf_tiles_v2_one_kernel then is the whole best-UCB call: its prologue makes
// the workgroup's rows (replays the ternary geometry for its compact slots -- the inverse of grow_unique_kernel's
// row -> slot map -- or takes raw rows, and scales them: prep_leaves_kernel's arithmetic), the tile code runs unchanged,
// the epilogue finalises the leaves (finalize_leaf), reduces them per segment with np.argmax's rule, and the LAST
// workgroup to arrive (one ticket counter, agent-scope fences around it) folds the workgroups' winners and writes the
// records to device and pinned host memory.  A centre child that does NOT repeat its parent bit for bit (never for boxes
// cut from the unit cube) cannot get a closed-form slot: the kernel raises `fallback` and the host runs the general
// sequence.  Same arithmetic per leaf as the multi-launch sequences: the SAME BITS (tests/test_gpu_parity.py).
template <typename TG, int LW>
__device__ __forceinline__ void one_launch_prologue(const OneLaunch& o, int dp, unsigned char* lds, int tid) {
  const int64_t slot = (int64_t)blockIdx.x * LW + tid;
  TG* out_s = static_cast<TG*>(o.leaves_s);
  TG* norm = static_cast<TG*>(o.lnorm);
  if (tid >= LW) return;
  const int d = o.d;
  if (slot >= o.total) {  // padding rows of the last workgroup: clean zeros
    for (int k = 0; k < dp; ++k) out_s[slot * dp + k] = 0;
    norm[slot] = 0;
    return;
  }
  TG acc = 0;
  if (o.mode == 2) {  // raw rows
    for (int k = 0; k < dp; ++k) {
      TG v = 0;
      if (k < d) {
        const double x = o.raw_f64 ? static_cast<const double*>(o.raw)[slot * d + k] : (double)static_cast<const float*>(o.raw)[slot * d + k];
        v = (TG)(x / o.ls[k]);
      }
      out_s[slot * dp + k] = v;
      acc += v * v;
    }
    norm[slot] = acc;
    return;
  }
  // grown rows: compact slot -> (box, level, position), the inverse of slot = 3^(j-1) + 2 (p / 3) + (p % 3 == 2)
  double* lo = reinterpret_cast<double*>(lds);
  double* hi = lo + (size_t)d * LW;
  const int seg = (int)(slot / o.uniq);
  const int64_t sl = slot % o.uniq;
  int level = 0;
  int64_t width = 1, p = 0;  // 3^level, position inside the level
  if (sl > 0) {
    int64_t w3 = 1;  // 3^(level - 1)
    level = 1;
    while (sl >= 3 * w3) {
      w3 *= 3;
      ++level;
    }
    const int64_t within = sl - w3;
    p = 3 * (within / 2) + ((within & 1) ? 2 : 0);
    width = 3 * w3;
  }
  const double* b = o.boxes.b + (int64_t)seg * d * 2;
  for (int k = 0; k < d; ++k) {
    lo[k * LW + tid] = b[2 * k];
    hi[k * LW + tid] = b[2 * k + 1];
  }
  int64_t div = width;
  for (int st = 0; st < level; ++st) {
    div /= 3;
    grow_split<LW>(lo, hi, tid, d, (int)((p / div) % 3));
  }
  const int64_t row = (width - 1) / 2 + p;  // reference row index inside the box
  o.key[slot] = (int64_t)seg * o.rows + row;
  for (int k = 0; k < dp; ++k) {
    TG v = 0;
    if (k < d) v = (TG)(((lo[k * LW + tid] + hi[k * LW + tid]) / 2) / o.ls[k]);
    out_s[slot * dp + k] = v;
    acc += v * v;
  }
  norm[slot] = acc;
  // the chain of centre children below this node (they have no slot of their own): each must repeat its parent's centre
  for (int lev = level + 1; lev < o.depth; ++lev) {
    int kmax;
    const double parent_c = grow_split_centre<LW>(lo, hi, tid, d, &kmax);
    grow_split<LW>(lo, hi, tid, d, 1);
    const double child_c = (lo[kmax * LW + tid] + hi[kmax * LW + tid]) / 2;
    if (__builtin_bit_cast(long long, parent_c) != __builtin_bit_cast(long long, child_c)) atomicOr(o.fallback, 1u);
  }
}

template <int LW>
__device__ __forceinline__ void one_launch_epilogue(const OneLaunch& o, int tid, unsigned char* lds) {
  // (scratch in the kernel's DYNAMIC LDS -- free by now: static LDS on top of it would break the 160 KB opt-in)
  Best* sh = reinterpret_cast<Best*>(lds);
  int64_t* shp = reinterpret_cast<int64_t*>(lds + 64);
  unsigned& last_flag = *reinterpret_cast<unsigned*>(lds + 96);
  const int64_t j = (int64_t)blockIdx.x * LW + tid;
  const bool mine_row = tid < LW && j < o.total;
  double u = 0;
  int64_t id = -1;
  if (mine_row) {
    u = finalize_leaf(o.fin, j);
    id = (o.mode == 1) ? o.key[j] : j;
  }
  for (int seg = 0; seg < o.nseg; ++seg) {
    Best mine{0.0, -1};
    int64_t mypos = -1;
    if (mine_row) {
      const bool in = (o.mode == 1) ? (id / o.rows == seg) : (j >= o.seg_off[seg] && j < o.seg_off[seg + 1]);
      if (in) {
        mine = Best{u, id};
        mypos = j;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      Best c;
      c.u = __shfl_xor(mine.u, off);
      c.i = __shfl_xor(mine.i, off);
      const int64_t cp = __shfl_xor(mypos, off);
      if (better(c, mine)) {
        mine = c;
        mypos = cp;
      }
    }
    if ((tid & 63) == 0) {
      sh[tid >> 6] = mine;
      shp[tid >> 6] = mypos;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (better(sh[w], mine)) {
          mine = sh[w];
          mypos = shp[w];
        }
      static_cast<Best*>(o.partial)[(int64_t)blockIdx.x * o.nseg + seg] = mine;
      o.ppos[(int64_t)blockIdx.x * o.nseg + seg] = mypos;
    }
    __syncthreads();
  }
  // last arriver folds: this workgroup's stores (leaf values, partials) are released before the ticket, the folding
  // workgroup acquires behind it
  if (tid == 0) {
    __threadfence();
    const unsigned t = atomicAdd(o.ticket, 1u);
    last_flag = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last_flag) return;
  __threadfence();  // EVERY thread of the folding workgroup acquires before it reads the other workgroups' partials and leaves
  const int nwg = (int)gridDim.x;
  for (int seg = 0; seg < o.nseg; ++seg) {
    Best mine{0.0, -1};
    int64_t mypos = -1;
    for (int b = tid; b < nwg; b += 256) {
      const Best c = static_cast<const Best*>(o.partial)[(int64_t)b * o.nseg + seg];
      if (better(c, mine)) {
        mine = c;
        mypos = o.ppos[(int64_t)b * o.nseg + seg];
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      Best c;
      c.u = __shfl_xor(mine.u, off);
      c.i = __shfl_xor(mine.i, off);
      const int64_t cp = __shfl_xor(mypos, off);
      if (better(c, mine)) {
        mine = c;
        mypos = cp;
      }
    }
    if ((tid & 63) == 0) {
      sh[tid >> 6] = mine;
      shp[tid >> 6] = mypos;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (better(sh[w], mine)) {
          mine = sh[w];
          mypos = shp[w];
        }
      double rec[4];
      if (mine.i < 0) {
        rec[0] = rec[1] = rec[2] = __builtin_nan("");
        rec[3] = __builtin_bit_cast(double, (int64_t)-1);
      } else {
        rec[0] = o.fin.mean[mypos];
        rec[1] = o.fin.var[mypos];
        rec[2] = o.fin.ucb[mypos];
        rec[3] = __builtin_bit_cast(double, (o.mode == 1) ? (int64_t)(mine.i - (int64_t)seg * o.rows) : (int64_t)(mine.i - o.seg_off[seg]));
      }
      for (int k = 0; k < 4; ++k) {
        o.out_vals[seg * 4 + k] = rec[k];
        if (o.host_vals != nullptr) o.host_vals[seg * 4 + k] = rec[k];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const unsigned fb = atomicExch(o.fallback, 0u);  // (read and reset for the next call)
    const double live = __builtin_bit_cast(double, (int64_t)o.total);
    const double status = fb ? 1.0 : 0.0;  // 1: a centre child does not repeat its parent -- the host runs the general sequence
    o.out_vals[o.nseg * 4] = live;
    o.out_vals[o.nseg * 4 + 1] = status;
    if (o.host_vals != nullptr) {
      o.host_vals[o.nseg * 4] = live;
      o.host_vals[o.nseg * 4 + 1] = status;
    }
    *o.ticket = 0u;
    publish_done(o.host_vals, o.nseg * 4 + 2, o.done_token);  // (thread 0 wrote every host record itself)
  }
}

template <typename T, typename TG, int BM, int CT, int KERNEL, bool ONE>
__device__ __forceinline__ void leaf_tiles_v2_body(
    const T* __restrict__ linv_p, const TG* __restrict__ xs_p, const TG* __restrict__ xnorm,
    const T* __restrict__ alpha, const TG* __restrict__ leaves_s, const TG* __restrict__ lnorm,
    double* __restrict__ part_var, double* __restrict__ part_mean, int dp4, int64_t mpad, int nbi,
    T variance, const int64_t* __restrict__ m_live, const OneLaunch& one) {
  using M = Mfma<T>;
  using vec4 = typename M::vec4;
  using vecG = typename Mfma<TG>::vec4;
  using L = V2Lds<T>;
  constexpr int RT = BM / 16;
  constexpr int FB = L::SLABS * 1024;        // bytes of one A fragment
  constexpr int XB = 64 * (int)sizeof(TG);   // bytes of one X fragment
  constexpr TG C2 = (TG)KernScale<KERNEL>::C2;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  if (m_live != nullptr && (int64_t)blockIdx.x * (4 * CT * 16) >= *m_live) return;  // workgroup-uniform
  if constexpr (ONE) {
    one_launch_prologue<TG, 4 * CT * 16>(one, dp4 * 4, lds_raw, threadIdx.x);
    __syncthreads();  // (the rows this workgroup reads below are its own: visible behind the barrier)
  }
  unsigned char* panel = lds_raw;                              // [2][RT] fragments of FB bytes
  unsigned char* xsl = panel + (size_t)2 * RT * FB;            // [2][dp4] fragments of XB bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: loops over it stay uniform
  TG* xb = reinterpret_cast<TG*>(xsl + (size_t)2 * dp4 * XB) + (size_t)wave * CT * dp4 * 64;

  const int bi = nbi - 1 - (int)blockIdx.y;  // heaviest row blocks are dispatched first
  const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * (CT * 16);
  const int dp = dp4 * 4;
  const int kt_diag0 = bi * RT, kt_end = kt_diag0 + RT;
  const vec4* linv4 = reinterpret_cast<const vec4*>(linv_p);
  const vecG* xn4 = reinterpret_cast<const vecG*>(xnorm);
  const vec4* al4 = reinterpret_cast<const vec4*>(alpha);

  // LDS-DMA duty: ONE LDS window (one M0 value) per wave between two workgroup barriers, as in
  // leaf_tiles_bf16_kernel: wave w moves the RT / 4 consecutive fragments w RT / 4 .. of the panel (4 or 8 KB),
  // addressed as M0 + the instruction's immediate offset, which the hardware adds to the global address as well
  // (the per-fragment global base is pre-biased).  The X fragments of a k-tile (dp4 x 64 values: at most three
  // per thread) travel through registers instead: loaded at the top of the step, written to LDS in front of
  // the barrier that ends it.
  constexpr int NFW = RT / 4, WBYTES = NFW * FB;
  static_assert(WBYTES <= 8192, "a wave's fragments must fit the 13-bit signed immediate around the window centre");
  const int lane_b = lane * (int)sizeof(vec4);
  const unsigned char* pgb[NFW];
#pragma unroll
  for (int j = 0; j < NFW; ++j) {
    const size_t r = (size_t)(kt_diag0 + wave * NFW + j);
    pgb[j] = reinterpret_cast<const unsigned char*>(linv4 + r * (r + 1) / 2 * 64) - (j * FB - WBYTES / 2);
  }
  auto uniform = [](const unsigned char* p) {  // keeps a wave-uniform address in scalar registers (no per-lane hoisting)
    const unsigned long long g = (unsigned long long)p;
    unsigned lo = (unsigned)g, hi = (unsigned)(g >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
  };
  auto issue_panel = [&](int kt, int buf) {
    unsigned char* centre = panel + (size_t)buf * RT * FB + (size_t)wave * WBYTES + WBYTES / 2;
    static_for<0, NFW>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      // tiles above the diagonal do not exist in the packed triangle; their slot is never read
      if (kt <= kt_diag0 + wave * NFW + j) {
        const unsigned char* src = uniform(pgb[j] + (size_t)kt * 64 * sizeof(vec4)) + lane_b;
        static_for<0, L::SLABS>([&](auto h_) {
          constexpr int h = decltype(h_)::value;
          // slab h: bytes 16 h .. of every lane's vec4 -> LDS slab h of the fragment (the immediate also moves the
          // global address: the slab's +16 there is what is left after the -1024 h that undoes it)
          glds16_off<j * FB + h * 1024 - WBYTES / 2>(src + 16 * h - h * 1024, centre);
        });
      }
    });
  };
  constexpr int XL = 3;  // X values per thread and k-tile (dp4 <= 12)
  TG xr[XL];
  auto load_xs = [&](int kt) {
#pragma unroll
    for (int u = 0; u < XL; ++u)
      if (tid + 256 * u < dp4 * 64) xr[u] = xs_p[(size_t)kt * dp4 * 64 + tid + 256 * u];
  };
  auto store_xs = [&](int buf) {
    TG* xd = reinterpret_cast<TG*>(xsl + (size_t)buf * dp4 * XB);
#pragma unroll
    for (int u = 0; u < XL; ++u)
      if (tid + 256 * u < dp4 * 64) xd[tid + 256 * u] = xr[u];
  };

  // prologue: panel(0) -> P[0], xs(0) -> X[1], xs(1) -> X[0]; this wave's leaf fragments -> xb
  issue_panel(0, 0);
  load_xs(0);
  store_xs(1);
  if (kt_end > 1) {
    load_xs(1);
    store_xs(0);
  }
  for (int t = 0; t < CT; ++t)
    for (int c = 0; c < dp4; ++c)
      xb[(t * dp4 + c) * 64 + lane] =
          leaves_s[(col0 + t * 16 + (lane & 15)) * dp + 4 * c + (lane >> 4)];
  TG nb[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) nb[t] = lnorm[col0 + t * 16 + (lane & 15)] * C2;
  vec4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[rt][t] = vec4{0, 0, 0, 0};
  T macc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) macc[t] = 0;
  vecG na = xn4[lane >> 4] * C2;  // norms of k-tile 0
  __syncthreads();                // (hipcc drains the LDS-DMA queue before the barrier)

  // G(0): not overlapped with anything
  vec4 p_cur[CT];
  {
    vecG s[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) s[t] = vecG{0, 0, 0, 0};
    const unsigned char* x1 = xsl + (size_t)1 * dp4 * XB;
    for (int c = 0; c < dp4; ++c) {
      const TG xa = reinterpret_cast<const TG*>(x1 + c * XB)[lane];
#pragma unroll
      for (int t = 0; t < CT; ++t) s[t] = Mfma<TG>::mma(xa, xb[(t * dp4 + c) * 64 + lane], s[t]);
    }
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        p_cur[t][r] = kern_from_scaled<KERNEL>((T)fma_t((TG)(TG(-2) * C2), s[t][r], na[r] + nb[t]), variance);
    if (kt_diag0 == 0) {
      const vec4 a4 = al4[lane >> 4];
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          macc[t] = fma_t(p_cur[t][r], a4[r], macc[t]);
          if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(macc[t]));  // no packed accumulation (see leaf_bf16_gen)
        }
    }
  }
  na = (kt_end > 1) ? xn4[4 + (lane >> 4)] * C2 : na;  // norms of k-tile 1
  __syncthreads();  // X[1] may be overwritten from here on

  // off-diagonal k-tiles: straight-line steps; then the RT k-tiles of the diagonal block
#define GPSO_V2_STEP(DIAGF, GEN_DIAG)                                                              \
  {                                                                                                \
    const int b = kt & 1;                                                                          \
    GPSO_PSTAMP(kt, 0);                                                                            \
    if (kt + 1 < kt_end) issue_panel(kt + 1, b ^ 1);                                               \
    vecG na_nxt = na;                                                                              \
    if (kt + 2 < kt_end) {                                                                         \
      load_xs(kt + 2);                                                                             \
      na_nxt = xn4[(kt + 2) * 4 + (lane >> 4)] * C2;                                               \
    }                                                                                              \
    GPSO_PSTAMP(kt, 1);                                                                            \
    leaf_v2_step<T, TG, RT, CT, KERNEL, DIAGF>(                                                    \
        kt, kt_diag0, kt + 1 < kt_end, GEN_DIAG, lane, dp4, panel + (size_t)b * RT * FB,           \
        xsl + (size_t)b * dp4 * XB, xb, na, al4, nb, variance, acc, macc, p_cur);                  \
    na = na_nxt;                                                                                   \
    if (kt + 2 < kt_end) store_xs(b ^ 1);                                                          \
    GPSO_PSTAMP(kt, 4);                                                                            \
    __syncthreads(); /* panel(kt+1) / xs(kt+2) landed; buffers b free */                           \
    GPSO_PSTAMP(kt, 5);                                                                            \
  }
  for (int kt = 0; kt < kt_diag0; ++kt) GPSO_V2_STEP(false, kt + 1 >= kt_diag0)
  for (int kt = kt_diag0; kt < kt_end; ++kt) GPSO_V2_STEP(true, true)
#undef GPSO_V2_STEP

  // ---- epilogue: column sums of squares over the BM rows (double accumulation: the variance is
  // sigma^2 minus this sum, a cancellation at the reference's noise floor) and the mean partial ----
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    double sq = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sq = fma((double)acc[rt][t][r], (double)acc[rt][t][r], sq);
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    double mm = (double)macc[t];
    mm += __shfl_xor(mm, 16);
    mm += __shfl_xor(mm, 32);
    if (lane < 16) {
      const int64_t col = col0 + t * 16 + lane;
      part_var[(int64_t)bi * mpad + col] = sq;
      part_mean[(int64_t)bi * mpad + col] = mm;
    }
  }
  if constexpr (ONE) {
    __syncthreads();  // this workgroup's partial sums are stored
    one_launch_epilogue<4 * CT * 16>(one, tid, lds_raw);
  }
}

template <typename T, typename TG, int BM, int CT, int KERNEL>
__global__ __launch_bounds__(256, 2) void leaf_tiles_v2_kernel(
    const T* __restrict__ linv_p, const TG* __restrict__ xs_p, const TG* __restrict__ xnorm,
    const T* __restrict__ alpha, const TG* __restrict__ leaves_s, const TG* __restrict__ lnorm,
    double* __restrict__ part_var, double* __restrict__ part_mean, int dp4, int64_t mpad, int nbi,
    T variance, const int64_t* __restrict__ m_live) {
  leaf_tiles_v2_body<T, TG, BM, CT, KERNEL, false>(linv_p, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, dp4,
                                                   mpad, nbi, variance, m_live, OneLaunch{});
}
template <typename T, typename TG, int BM, int CT, int KERNEL>
__global__ __launch_bounds__(256, 2) void leaf_tiles_v2_one_kernel(
    const T* __restrict__ linv_p, const TG* __restrict__ xs_p, const TG* __restrict__ xnorm,
    const T* __restrict__ alpha, double* __restrict__ part_var, double* __restrict__ part_mean, int dp4, int64_t mpad,
    T variance, OneLaunch one) {
  leaf_tiles_v2_body<T, TG, BM, CT, KERNEL, true>(linv_p, xs_p, xnorm, alpha, static_cast<const TG*>(one.leaves_s),
                                                  static_cast<const TG*>(one.lnorm), part_var, part_mean, dp4, mpad, 1,
                                                  variance, nullptr, one);
}

// =============================================================================================
// leaf_tiles, split-bf16 apply (float contexts, opt-in "predict math" bf16x3 / bf16x6)
//
// The bf16 matrix cores run at 16x the rate of the f32 MFMA.  L^-1 (at fit time) and the generated
// cross-Gram tile (here) are split into NS bf16 pieces x = h0 + h1 (+ h2), each piece the bf16
// rounding of the remainder; the product is recovered from 3 (NS = 2: h0h0 + h0h1 + h1h0) or 6
// (NS = 3: + h1h1 + h0h2 + h2h0) v_mfma_f32_16x16x32_bf16 with f32 accumulation.  The x.x*
// contraction runs in TG (double by default, see above) and the Matern map in f32.
//
// Workgroup: 8 waves (one per CU, 2 per SIMD), 256 rows of L^-1 x 256 leaves; per k-step of 32
// training points the NS x 16 A fragments (1 KiB each, 8 bf16 per lane) arrive in LDS by
// global_load_lds, double buffered.  K-index convention of a fragment (same for A and B, so any
// assignment is valid): element j < 4 of lane l <-> point 32 q + 4 (l >> 4) + j, element j >= 4
// <-> point 32 q + 16 + 4 (l >> 4) + (j - 4): the two generated 16-point tiles of the step drop
// into the B operand without any lane movement.
// =============================================================================================

// bf16 pieces of L^-1 from the fit-type matrix (TF = float or double: the first piece rounds the
// full-precision value)
template <int NS, typename TF>
__global__ __launch_bounds__(256) void pack_linv_bf16_kernel(const TF* __restrict__ linv, int64_t n,
                                                             int64_t npad, u32x4* __restrict__ out, int64_t rt0) {
  const int64_t npad16 = npad / 16, npad32 = npad / 32;
  const int64_t idx = rt0 * npad32 * 64 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (rt, kq, lane)
  if (idx >= npad16 * npad32 * 64) return;
  const int lane = (int)(idx & 63);
  const int64_t kq = (idx >> 6) % npad32, rt = (idx >> 6) / npad32;
  const int64_t row = rt * 16 + (lane & 15);
  float v[8];
  float lo[8];  // TF = double: the part of the value float cannot hold goes into the later pieces
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t col = kq * 32 + (j >> 2) * 16 + 4 * (lane >> 4) + (j & 3);
    const TF x = (row < n && col <= row) ? linv[row * npad + col] : (TF)0;
    v[j] = (float)x;
    lo[j] = (float)(x - (TF)v[j]);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    u32x4 f;
#pragma unroll
    for (int h = 0; h < 4; ++h) f[h] = bf16_split_pair(v[2 * h], v[2 * h + 1]);
    if (s == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += lo[j];
    }
    out[((int64_t)s * npad16 + rt) * npad32 * 64 + kq * 64 + lane] = f;
  }
}

// ---- fp16 split ("f16x3"): x = h0 + h1 with fp16 pieces (11 significant bits each, round to nearest: |x - h0 - h1| <=
// 2^-23 |x|), a product = h0 h0' + h0 h1' + h1 h0' on v_mfma_f32_16x16x32_f16 -- THREE matrix instructions instead of
// six, the dropped h1 h1' is 2^-22 relative.  fp16 has 5 exponent bits, so both operands are scaled by powers of two
template <typename TF>
__global__ __launch_bounds__(256) void absmax_kernel(const TF* __restrict__ linv, int64_t n, int64_t npad,
                                                     unsigned* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  float m = 0.0f;
  if (r < n) {
    const TF* row = linv + r * npad;
    for (int64_t c = 4 * (threadIdx.x & 15); c <= r; c += 64) {  // (npad is a multiple of 64: c + 3 < npad)
      TF v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = row[c + j];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j <= r) m = fmaxf(m, fabsf((float)v[j]));
    }
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __builtin_bit_cast(unsigned, m));
}

// fp16 pieces of L^-1 2^sa in the fragment order of pack_linv_bf16_kernel<2>; scal[0] = max |L^-1| (absmax_kernel, or
// the fit's own pass over L^-1: fit.hip white_kernel / alpha_sum_kernel), scal[1] := 2^-sa (read by the predict kernel's epilogue)
// One launch, two block ranges (round 6; two launches before: the first one's 4 us were a launch that exits at once):
// blocks [0, nb_above) -- the tile rows [0, rt_split) ABOVE the rows a gpso_append wrote, grid-stride, and only when the
// scale they were packed with (scal[3], read-only here) is no longer the scale; blocks [nb_above, ..) -- the tile rows
// [rt_split, rt_hi), one (rt, kq, lane) per thread.  A full packing is rt_split = 0, nb_above = 0.
template <typename TF>
__global__ __launch_bounds__(256) void pack_linv_f16_kernel(const TF* __restrict__ linv, int64_t n, int64_t npad,
                                                            float* __restrict__ scal, u32x4* __restrict__ out, int64_t rt_split,
                                                            int64_t rt_hi, int nb_above) {
  const int64_t npad16 = npad / 16, npad32 = npad / 32;
  int e = 0;
  (void)frexpf(fmaxf(scal[0], 1e-30f), &e);  // max = m 2^e, m in [0.5, 1)
  const float up = ldexpf(1.0f, 14 - e);
  const bool above = (int)blockIdx.x < nb_above;
  if (above && scal[3] == ldexpf(1.0f, e - 14)) return;
  if (!above && (int)blockIdx.x == nb_above && threadIdx.x == 0) scal[1] = ldexpf(1.0f, e - 14);
  const int64_t rt_lo = above ? 0 : rt_split;
  const int64_t total = ((above ? rt_split : rt_hi) - rt_lo) * npad32 * 64;
  const int64_t first = above ? (int64_t)blockIdx.x * blockDim.x : ((int64_t)blockIdx.x - nb_above) * blockDim.x;
  const int64_t stride = above ? (int64_t)nb_above * blockDim.x : total;  // (the lower range: one element per thread)
  for (int64_t idx = first + threadIdx.x; idx < total; idx += stride) {  // (rt, kq, lane)
    const int lane = (int)(idx & 63);
    const int64_t kq = (idx >> 6) % npad32, rt = rt_lo + (idx >> 6) / npad32;
    const int64_t row = rt * 16 + (lane & 15);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t col = kq * 32 + (j >> 2) * 16 + 4 * (lane >> 4) + (j & 3);
      // (a double L^-1 is rounded to float first: the two fp16 pieces hold 22 bits, float's 24 are enough)
      v[j] = (row < n && col <= row) ? (float)linv[row * npad + col] * up : 0.0f;
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      u32x4 f;
#pragma unroll
      for (int h = 0; h < 4; ++h) f[h] = f16_split_pair(v[2 * h], v[2 * h + 1]);
      out[((int64_t)s_ * npad16 + rt) * npad32 * 64 + kq * 64 + lane] = f;
    }
  }
}

// ---- round 4: the x.x* contraction of the float generation on the fp16 pipe -----------------------------------------
// In float the contraction is D_pad / 4 v_mfma_f32_16x16x4_f32 per 16 x 16 tile at 32 clocks each: at D = 40 1 280 clocks
// of a k-step's 3 072-clock apply budget, and the measured fraction of the bound falls with D accordingly (0.455 at D = 6,
// 0.32 at D = 40 for N = 16 384: profiles/r04_sweep.jsonl).  The same split that carries L^-1 carries the scaled inputs:
// x / l = h0 + h1 in fp16 (|x / l - h0 - h1| <= 2^-24 |x / l|: float's own rounding), a product = h1 h0' + h0 h1' + h0 h0'
// on v_mfma_f32_16x16x32_f16 -- 12 ceil(D / 32) matrix instructions of 16 clocks per k-step: 192 clocks up to D = 32,
// 384 up to D = 64.  Both sides are scaled by 2^sx (exact) with the largest |x / l| of the TRAINING inputs in [2^7, 2^8):
// a leaf up to 255 times further out than any training input still fits fp16 (beyond that it saturates: its r^2 is
// dominated by its own norm, which stays float, and the kernel map underflows either way); the combine multiplies the
// contraction by -2 SC 2^-2sx.
// The TRAINING input's norm rides in the contraction: the chunks hold D_pad + 1 <= 32 C16 slots, slot D_pad of a training
// row is -|x / l|^2 2^2sx / 2^8 (the float norm, scaled by powers of two, split like every other entry; below 2^14 for
// D_pad <= 48) and slot D_pad of every leaf is 2^7 (exact in the first piece), so the accumulator arrives as
// 2^2sx (x.x* - |x|^2 / 2) and the combine is ONE fma per value, u = (-2 SC 2^-2sx) s + SC |x*|^2 -- no norm fetch, no norm
// scaling, no addition.  The leaf's norm stays a float outside the contraction: its range is not known at packing time.
// Both norms are sums over the values the PIECES represent (h0 + h1), not over the floats they came from.
// Fragment order of the training side: block (q, h, cc, piece) of 64 lanes x 16 bytes, lane l element j = row
// 32 q + 16 h + (l & 15), slot 32 cc + 8 (l >> 4) + j -- the A operand of the 16x16x32 instruction as it stands.
// scal: 4 device floats -- [0] max |x / l| (as float bits, atomicMax), [1] := 2^sx, [2] := 2^-2sx
__global__ __launch_bounds__(256) void absmax_f32_kernel(const float* __restrict__ v, int64_t count,
                                                         unsigned* __restrict__ out) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(v[i]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __builtin_bit_cast(unsigned, m));
}
__global__ __launch_bounds__(256) void pack_xs_f16_kernel(const float* __restrict__ xs, const float* __restrict__ xnorm,
                                                          int64_t npad, int dp, int nc, float* __restrict__ scal,
                                                          u32x4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // ((q, h), cc, lane)
  int e = 0;
  (void)frexpf(fmaxf(scal[0], 1e-30f), &e);  // max = m 2^e, m in [0.5, 1)
  const float up = ldexpf(1.0f, 8 - e);
  if (idx == 0) {
    scal[1] = up;
    scal[2] = ldexpf(1.0f, 2 * (e - 8));
  }
  if (idx >= (npad / 16) * nc * 64) return;
  const int lane = (int)(idx & 63);
  const int cc = (int)((idx >> 6) % nc);
  const int64_t kt = (idx >> 6) / nc;  // 16-row tile = 2 q + h
  const int64_t row = kt * 16 + (lane & 15);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 32 * cc + 8 * (lane >> 4) + j;
    v[j] = k < dp ? xs[row * dp + k] * up : 0.0f;
    if (k == dp) {
      // the norm slot: |x / l|^2 of the row AS THE PIECES REPRESENT IT (h0 + h1 differs from the float by up to 2^-22
      // relative).  With the float norm against the pieces' products, r^2 of a leaf AT a training input came out 3x
      // further from zero than with the f32 contraction (self-test readings, tools/micro/gen_readings.py); with norms
      // and products of the same numbers the distance of a point to itself is 2 h1.h1' = 2^-23 |x|^2 and the error at
      // general leaves drops below the f32 contraction's (profiles/r04_c16_check.jsonl)
      float acc = 0.0f;
      for (int kk = 0; kk < dp; ++kk) {
        const float a = xs[row * dp + kk] * up;
        const float h0 = (float)(_Float16)a;
        const float vp = h0 + (float)(_Float16)(a - h0);
        acc += vp * vp;
      }
      v[j] = -acc * (1.0f / 256.0f);
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    u32x4 f;
#pragma unroll
    for (int h = 0; h < 4; ++h) f[h] = f16_split_pair(v[2 * h], v[2 * h + 1]);
    out[((kt * nc + cc) * 2 + s_) * 64 + lane] = f;
  }
}
void launch_gen_inputs_f16(hipStream_t st, const float* xs32, const float* xnorm32, int64_t npad, int dp, float* scal,
                           void* xs_h16) {
  const int nc = leaf_c16_chunks(dp / 4);
  (void)hipMemsetAsync(scal, 0, 4, st);
  const int64_t count = npad * dp;
  hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 1024)), dim3(256), 0, st,
                     xs32, count, reinterpret_cast<unsigned*>(scal));
  const int64_t total = (npad / 16) * nc * 64;
  hipLaunchKernelGGL(pack_xs_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, xs32, xnorm32, npad,
                     dp, nc, scal, static_cast<u32x4*>(xs_h16));
}


template <typename TF>
void launch_pack_linv_bf16(hipStream_t st, int nsplit, const TF* linv, int64_t n, int64_t npad,
                           void* linv_b, int64_t rt0) {
  const int64_t total = (npad / 16 - rt0) * (npad / 32) * 64;
  if (total <= 0) return;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (nsplit == 3)
    hipLaunchKernelGGL((pack_linv_bf16_kernel<3, TF>), grid, dim3(256), 0, st, linv, n, npad, static_cast<u32x4*>(linv_b), rt0);
  else
    hipLaunchKernelGGL((pack_linv_bf16_kernel<2, TF>), grid, dim3(256), 0, st, linv, n, npad, static_cast<u32x4*>(linv_b), rt0);
}
template <typename TF>
void launch_pack_linv_f16(hipStream_t st, const TF* linv, int64_t n, int64_t npad, float* scal, void* linv_b,
                          bool have_max, int64_t rt0) {
  if (!have_max) {
    (void)hipMemsetAsync(scal, 0, 4, st);
    hipLaunchKernelGGL((absmax_kernel<TF>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, linv, n, npad,
                       reinterpret_cast<unsigned*>(scal));
  }
  const int64_t npad16 = npad / 16, per_rt = (npad / 32) * 64;
  // (the tile rows above the appended ones: only when the scale changed -- decided on the device; same launch)
  const int nb_above = rt0 > 0 ? (int)std::min<int64_t>((rt0 * per_rt + 255) / 256, 4096) : 0;
  const int64_t nb_rows = npad16 > rt0 ? ((npad16 - rt0) * per_rt + 255) / 256 : 0;
  if (nb_above + nb_rows > 0)
    hipLaunchKernelGGL((pack_linv_f16_kernel<TF>), dim3((unsigned)(nb_above + nb_rows)), dim3(256), 0, st, linv, n, npad, scal,
                       static_cast<u32x4*>(linv_b), rt0, npad16, nb_above);
}
template void launch_pack_linv_f16<float>(hipStream_t, const float*, int64_t, int64_t, float*, void*, bool, int64_t);
template void launch_pack_linv_f16<double>(hipStream_t, const double*, int64_t, int64_t, float*, void*, bool, int64_t);
template void launch_pack_linv_bf16<float>(hipStream_t, int, const float*, int64_t, int64_t, void*, int64_t);
template void launch_pack_linv_bf16<double>(hipStream_t, int, const double*, int64_t, int64_t, void*, int64_t);

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void leaf_finalize_kernel(const double* __restrict__ part_var,
                                                            const double* __restrict__ part_mean,
                                                            int nbi, int64_t mpad, int64_t m,
                                                            KernParams kp, double varsigma,
                                                            double* __restrict__ mean,
                                                            double* __restrict__ var,
                                                            double* __restrict__ ucb, const void* __restrict__ lnorm,
                                                            int lnorm_f64) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double v = 0, mu = 0;
  for (int b = 0; b < nbi; ++b) {
    v += part_var[(int64_t)b * mpad + j];
    mu += part_mean[(int64_t)b * mpad + j];
  }
  if (leaf_norm_is_nan(lnorm, lnorm_f64, j)) v = mu = __builtin_nan("");
  // [gpflow base_conditional] fvar = k** - sum A^2 ; predict_y adds the noise variance
  const double vy = __dadd_rn(__dsub_rn(kp.variance, v), kp.noise);
  const double my = __dadd_rn(mu, kp.mean_c);
  mean[j] = my;
  var[j] = vy;
  // gpso/gp_surrogate.py:326  ucb = mean + varsigma * var  (two roundings, as numpy does)
  if (ucb) {
    double prod = varsigma * vy;
    asm volatile("" : "+v"(prod));  // keep the product rounded on its own: no fma contraction
    ucb[j] = my + prod;
  }
}

__device__ __forceinline__ Best block_best(Best mine, Best* sh) {
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.u = __shfl_xor(mine.u, off);
    o.i = __shfl_xor(mine.i, off);
    if (better(o, mine)) mine = o;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (better(sh[w], mine)) mine = sh[w];
  }
  return mine;  // valid on thread 0
}

// stage 1: grid (nblk, nseg); each block scans a strided share of its segment.  FIN (round 4): the leaves arrive as
// the tile kernel's partial sums and are FINALISED here (leaf_finalize_kernel's arithmetic, leaf for leaf: every leaf
// of [0, M) lies in exactly one segment) -- one launch and one pass over mean / var / ucb less per call
template <bool FIN>
__global__ __launch_bounds__(256) void seg_argmax_stage1(const double* __restrict__ ucb,
                                                         const int64_t* __restrict__ seg_off,
                                                         Best* __restrict__ partial, LeafFinalize fin) {
  __shared__ Best sh[4];
  const int seg = blockIdx.y;
  const int64_t lo = seg_off[seg], hi = seg_off[seg + 1];
  Best mine{0.0, -1};
  for (int64_t j = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < hi;
       j += (int64_t)gridDim.x * blockDim.x) {
    Best c{FIN ? finalize_leaf(fin, j) : ucb[j], j};
    if (better(c, mine)) mine = c;
  }
  mine = block_best(mine, sh);
  if (threadIdx.x == 0) partial[(int64_t)seg * gridDim.x + blockIdx.x] = mine;
}

// stage 2: one block per segment; writes idx (relative to the segment) and the winner's values
__global__ __launch_bounds__(256) void seg_argmax_stage2(const Best* __restrict__ partial, int nblk,
                                                         const int64_t* __restrict__ seg_off,
                                                         const double* __restrict__ mean,
                                                         const double* __restrict__ var,
                                                         const double* __restrict__ ucb, int nseg,
                                                         double* __restrict__ out_vals /*[nseg*4 + 2]*/,
                                                         double* __restrict__ host_vals /* nullable: the same records
                                                         straight into pinned host memory (no copy operation behind) */,
                                                         double done_token) {
  __shared__ Best sh[4];
  const int seg = blockIdx.x;
  if (seg >= nseg) return;
  Best mine{0.0, -1};
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    Best c = partial[(int64_t)seg * nblk + b];
    if (better(c, mine)) mine = c;
  }
  mine = block_best(mine, sh);
  if (threadIdx.x == 0) {
    // one record per segment: mean, var, ucb and -- bit-cast into the 4th slot -- the winner's
    // index relative to the segment (a single device-to-host copy brings everything back)
    if (mine.i < 0) {  // empty segment
      out_vals[seg * 4 + 0] = out_vals[seg * 4 + 1] = out_vals[seg * 4 + 2] = __builtin_nan("");
      out_vals[seg * 4 + 3] = __builtin_bit_cast(double, (int64_t)-1);
    } else {
      out_vals[seg * 4 + 0] = mean[mine.i];
      out_vals[seg * 4 + 1] = var[mine.i];
      out_vals[seg * 4 + 2] = ucb[mine.i];
      out_vals[seg * 4 + 3] = __builtin_bit_cast(double, (int64_t)(mine.i - seg_off[seg]));
    }
    if (seg == 0) out_vals[nseg * 4 + 1] = 0.0;  // status slot of a group payload: this rank's half succeeded
    if (host_vals != nullptr) {
#pragma unroll
      for (int k = 0; k < 4; ++k) host_vals[seg * 4 + k] = out_vals[seg * 4 + k];
      if (seg == 0) host_vals[nseg * 4 + 1] = 0.0;
      if (nseg == 1) publish_done(host_vals, nseg * 4 + 2, done_token);  // (one workgroup, one writer)
    }
  }
}


// Keyed variant for de-duplicated on-device growth (grow.hip: grow_unique_kernel): row j of the compact
// list carries key[j] = seg * rows + reference row index.  Segment seg owns the compact range
// [seg * uniq, (seg + 1) * uniq) plus those rows of the appended tail [nseg * uniq, *live) whose key
// falls into it.  The order is np.argmax's on (ucb, reference row index), so the winner and its index
// are what scoring the full duplicated list would give.
template <bool FIN>
__global__ __launch_bounds__(256) void keyed_argmax_stage1(const double* __restrict__ ucb,
                                                           const int64_t* __restrict__ key, int64_t rows,
                                                           int64_t uniq, int nseg,
                                                           const int64_t* __restrict__ live,
                                                           Best* __restrict__ partial,
                                                           int64_t* __restrict__ pos /* compact row of each partial */,
                                                           LeafFinalize fin) {
  __shared__ Best sh[4];
  __shared__ int64_t shp[4];
  const int seg = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  Best mine{0.0, -1};
  int64_t mypos = -1;
  for (int64_t j = (int64_t)seg * uniq + t0; j < (int64_t)(seg + 1) * uniq; j += stride) {
    Best c{FIN ? finalize_leaf(fin, j) : ucb[j], key[j]};
    if (better(c, mine)) {
      mine = c;
      mypos = j;
    }
  }
  const int64_t lv = *live;
  for (int64_t j = (int64_t)nseg * uniq + t0; j < lv; j += stride) {
    const int64_t kj = key[j];
    if (kj / rows != seg) continue;
    Best c{FIN ? finalize_leaf(fin, j) : ucb[j], kj};
    if (better(c, mine)) {
      mine = c;
      mypos = j;
    }
  }
  // block reduction carrying the compact position along
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.u = __shfl_xor(mine.u, off);
    o.i = __shfl_xor(mine.i, off);
    const int64_t op = __shfl_xor(mypos, off);
    if (better(o, mine)) {
      mine = o;
      mypos = op;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sh[wave] = mine;
    shp[wave] = mypos;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (better(sh[w], mine)) {
        mine = sh[w];
        mypos = shp[w];
      }
    partial[(int64_t)seg * gridDim.x + blockIdx.x] = mine;
    pos[(int64_t)seg * gridDim.x + blockIdx.x] = mypos;
  }
}

__global__ __launch_bounds__(256) void keyed_argmax_stage2(const Best* __restrict__ partial,
                                                           const int64_t* __restrict__ pos, int nblk,
                                                           int64_t rows, const double* __restrict__ mean,
                                                           const double* __restrict__ var,
                                                           const double* __restrict__ ucb,
                                                           const int64_t* __restrict__ live, int nseg,
                                                           double* __restrict__ out_vals /*[nseg*4 + 2]*/,
                                                           double* __restrict__ host_vals /* nullable, see seg_argmax_stage2 */,
                                                           double done_token) {
  __shared__ Best sh[4];
  const int seg = blockIdx.x;
  Best mine{0.0, -1};
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    Best c = partial[(int64_t)seg * nblk + b];
    if (better(c, mine)) mine = c;
  }
  mine = block_best(mine, sh);
  if (threadIdx.x == 0) {
    if (mine.i < 0) {
      out_vals[seg * 4 + 0] = out_vals[seg * 4 + 1] = out_vals[seg * 4 + 2] = __builtin_nan("");
      out_vals[seg * 4 + 3] = __builtin_bit_cast(double, (int64_t)-1);
    } else {
      int64_t at = -1;  // keys are unique: the partial that holds the winner's key holds its position
      for (int b = 0; b < nblk; ++b)
        if (partial[(int64_t)seg * nblk + b].i == mine.i) at = pos[(int64_t)seg * nblk + b];
      out_vals[seg * 4 + 0] = mean[at];
      out_vals[seg * 4 + 1] = var[at];
      out_vals[seg * 4 + 2] = ucb[at];
      out_vals[seg * 4 + 3] = __builtin_bit_cast(double, (int64_t)(mine.i - (int64_t)seg * rows));
    }
    if (seg == 0) {
      out_vals[nseg * 4] = __builtin_bit_cast(double, *live);
      out_vals[nseg * 4 + 1] = 0.0;  // status slot of a group payload
    }
    if (host_vals != nullptr) {
#pragma unroll
      for (int k = 0; k < 4; ++k) host_vals[seg * 4 + k] = out_vals[seg * 4 + k];
      if (seg == 0) {
        host_vals[nseg * 4] = out_vals[nseg * 4];
        host_vals[nseg * 4 + 1] = 0.0;
      }
      if (nseg == 1) publish_done(host_vals, nseg * 4 + 2, done_token);
    }
  }
}

// Small batches (round 4): finalize + both arg-max stages in ONE launch of one workgroup.  Every live leaf is finalised
// exactly once (finalize_leaf), every segment reduced with np.argmax's rule on (ucb, index / reference key): the same
// records seg_argmax_* / keyed_argmax_* produce, with two launches (and the copy back) less.
template <bool KEYED>
__global__ __launch_bounds__(256) void small_best_kernel(SmallBest a) {
  __shared__ Best sh[4];
  __shared__ int64_t shp[4];
  const int tid = threadIdx.x;
  const int64_t live = KEYED ? a.base + (int64_t)*a.extra : a.m;
  for (int seg = 0; seg < a.nseg; ++seg) {
    Best mine{0.0, -1};
    int64_t mypos = -1;
    if constexpr (KEYED) {
      for (int64_t j = (int64_t)seg * a.uniq + tid; j < (int64_t)(seg + 1) * a.uniq; j += 256) {
        Best c{finalize_leaf(a.fin, j), a.key[j]};
        if (better(c, mine)) {
          mine = c;
          mypos = j;
        }
      }
      for (int64_t j = (int64_t)a.nseg * a.uniq + tid; j < live; j += 256) {
        const int64_t kj = a.key[j];
        if (kj / a.rows != seg) continue;
        Best c{finalize_leaf(a.fin, j), kj};
        if (better(c, mine)) {
          mine = c;
          mypos = j;
        }
      }
    } else {
      for (int64_t j = a.seg_off[seg] + tid; j < a.seg_off[seg + 1]; j += 256) {
        Best c{finalize_leaf(a.fin, j), j};
        if (better(c, mine)) {
          mine = c;
          mypos = j;
        }
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      Best o;
      o.u = __shfl_xor(mine.u, off);
      o.i = __shfl_xor(mine.i, off);
      const int64_t op = __shfl_xor(mypos, off);
      if (better(o, mine)) {
        mine = o;
        mypos = op;
      }
    }
    if ((tid & 63) == 0) {
      sh[tid >> 6] = mine;
      shp[tid >> 6] = mypos;
    }
    __syncthreads();  // (also: the means / variances stored above are visible to thread 0 below)
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (better(sh[w], mine)) {
          mine = sh[w];
          mypos = shp[w];
        }
      double* o = a.out_vals + seg * 4;
      if (mine.i < 0) {
        o[0] = o[1] = o[2] = __builtin_nan("");
        o[3] = __builtin_bit_cast(double, (int64_t)-1);
      } else {
        o[0] = a.fin.mean[mypos];
        o[1] = a.fin.var[mypos];
        o[2] = a.fin.ucb[mypos];
        o[3] = __builtin_bit_cast(double, KEYED ? (int64_t)(mine.i - (int64_t)seg * a.rows) : (int64_t)(mine.i - a.seg_off[seg]));
      }
      if (a.host_vals != nullptr)
        for (int k = 0; k < 4; ++k) a.host_vals[seg * 4 + k] = o[k];
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.out_vals[a.nseg * 4] = __builtin_bit_cast(double, live);
    a.out_vals[a.nseg * 4 + 1] = 0.0;  // status slot of a group payload: this rank's half succeeded
    if (a.host_vals != nullptr) {
      a.host_vals[a.nseg * 4] = __builtin_bit_cast(double, live);
      a.host_vals[a.nseg * 4 + 1] = 0.0;
    }
    if (KEYED) *a.extra = 0ull;  // (zero between calls: the next growth appends from its own base)
    publish_done(a.host_vals, a.nseg * 4 + 2, a.done_token);  // (thread 0 wrote every host record itself)
  }
}

void launch_small_best(hipStream_t st, const SmallBest& a, bool keyed) {
  if (keyed) hipLaunchKernelGGL(small_best_kernel<true>, dim3(1), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(small_best_kernel<false>, dim3(1), dim3(256), 0, st, a);
}

// multi-GPU: one thread per segment folds the ranks' winners in rank order with the same rule.  A rank's payload is
// `stride` doubles: nseg x (mean, var, ucb, bit-cast index), then -- stride == nseg * 4 + 2 -- a spare slot and the
// rank's STATUS of the call (0 or a negative GPSO_E_* code): the fold also takes the worst status over the ranks
// (out[nseg * 4 + 1], and the rank it came from bit-cast into out[nseg * 4]), so that every rank of a group returns the
// same verdict and none is left inside a collective by a peer that failed locally.
__global__ void reduce_winners_kernel(const double* __restrict__ gathered, const int64_t* __restrict__ base,
                                      int world, int nseg, int stride, double* __restrict__ out,
                                      double* __restrict__ host_out /* nullable: the same records into pinned host memory */,
                                      double done_token) {
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg == 0 && stride >= nseg * 4 + 2) {
    double worst = 0.0;
    int64_t who = -1;
    for (int r = 0; r < world; ++r) {
      // not a number (a payload that never arrived) counts as GPSO_E_RCCL; the verdict is the true minimum over the
      // ranks, and `who` only moves when the minimum does
      const double raw = gathered[(int64_t)r * stride + nseg * 4 + 1];
      const double st = (raw == raw) ? raw : -6.0 /* GPSO_E_RCCL */;
      if (st < worst) {
        worst = st;
        who = r;
      }
    }
    out[nseg * 4] = __builtin_bit_cast(double, who);
    out[nseg * 4 + 1] = worst;
    if (host_out != nullptr) {
      host_out[nseg * 4] = __builtin_bit_cast(double, who);
      host_out[nseg * 4 + 1] = worst;
    }
  }
  if (seg < nseg) {
  Best best{0.0, -1};
  int from = -1;
  for (int r = 0; r < world; ++r) {
    const double* row = gathered + (int64_t)r * stride + (int64_t)seg * 4;
    int64_t i = __builtin_bit_cast(int64_t, row[3]);
    if (i >= 0 && base != nullptr) i += base[(int64_t)r * nseg + seg];
    const Best c{row[2], i};
    if (better(c, best)) {
      best = c;
      from = r;
    }
  }
  double* o = out + (int64_t)seg * 4;
  if (from < 0) {
    o[0] = o[1] = o[2] = __builtin_nan("");
    o[3] = __builtin_bit_cast(double, (int64_t)-1);
  } else {
    const double* row = gathered + (int64_t)from * stride + (int64_t)seg * 4;
    o[0] = row[0];
    o[1] = row[1];
    o[2] = row[2];
    o[3] = __builtin_bit_cast(double, best.i);
  }
  if (host_out != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) host_out[(int64_t)seg * 4 + k] = o[k];
  }
  }
  if (host_out != nullptr && done_token != 0.0 && gridDim.x == 1) {  // one workgroup: every writer releases, then the token
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) publish_done(host_out, nseg * 4 + 2, done_token);
  }
}

// per-chunk live count of a leaf batch processed in chunks: out[c] = clamp(*live - c * chunk, 0, chunk)
__global__ void chunk_live_kernel(const int64_t* __restrict__ live, int64_t chunk, int nchunk,
                                  int64_t* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunk) return;
  const int64_t v = *live - (int64_t)c * chunk;
  out[c] = v < 0 ? 0 : (v > chunk ? chunk : v);
}


// ---------------------------------------------------------------------------------------------
// host-side launchers (declared in kernels.hpp)
template <typename TG, typename TIN>
void launch_prep_leaves(hipStream_t st, const TIN* xs, int64_t m, int64_t mpad, int d, int dp,
                        const double* ls, const int64_t* m_live, TG* out, TG* norm) {
  const int64_t blocks = (mpad + 255) / 256;
  hipLaunchKernelGGL((prep_leaves_kernel<TG, TIN>), dim3((unsigned)blocks), dim3(256), 0, st, xs, m,
                     mpad, d, dp, ls, m_live, out, norm);
}
template void launch_prep_leaves<float, float>(hipStream_t, const float*, int64_t, int64_t, int, int, const double*, const int64_t*, float*, float*);
template void launch_prep_leaves<float, double>(hipStream_t, const double*, int64_t, int64_t, int, int, const double*, const int64_t*, float*, float*);
template void launch_prep_leaves<double, float>(hipStream_t, const float*, int64_t, int64_t, int, int, const double*, const int64_t*, double*, double*);
template void launch_prep_leaves<double, double>(hipStream_t, const double*, int64_t, int64_t, int, int, const double*, const int64_t*, double*, double*);

// rows of L^-1 per workgroup.  float: 256 rows x 32 leaves per wave, or -- once D and the number of
// row blocks are large -- 512 rows x 16 leaves: every generated K* tile then feeds 32 row tiles
// instead of 16, which halves the regeneration share (measured in round 1: C5 +6.4 %, C4 +1.9 %,
// C3 -4 %).  double: 256 rows x 16 leaves.  128-row blocks when N_pad is an odd multiple of 128.
template <>
int leaf_tiles_bm<float>(int64_t npad, int dp4) {
  if (npad % 512 == 0 && ((npad >= 4096 && dp4 >= 5) || (npad >= 2048 && dp4 >= 9))) return 512;
  return (npad % 256 == 0) ? 256 : 128;
}
template <>
int leaf_tiles_bm<double>(int64_t npad, int dp4) {
  (void)dp4;
  return (npad % 256 == 0) ? 256 : 128;
}

template <typename T, typename TG, int BM, int CT, int KERNEL>
static int launch_leaf_tiles_v2(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm,
                                const T* alpha, const TG* leaves_s, const TG* lnorm, double* part_var,
                                double* part_mean, int64_t npad, int dp4, int64_t mpad,
                                const KernParams& kp, const int64_t* m_live) {
  constexpr int RT = BM / 16;
  const int nbi = (int)(npad / BM);
  const dim3 grid((unsigned)(mpad / (4 * CT * 16)), (unsigned)nbi);
  const size_t lds = leaf_v2_lds_bytes<T, TG, CT>(RT, dp4);
  const int rc = ensure_dyn_lds((const void*)leaf_tiles_v2_kernel<T, TG, BM, CT, KERNEL>, (int)lds);
  if (rc) return rc;
  hipLaunchKernelGGL((leaf_tiles_v2_kernel<T, TG, BM, CT, KERNEL>), grid, dim3(256), lds, st, linv_p, xs_p,
                     xnorm, alpha, leaves_s, lnorm, part_var, part_mean, dp4, mpad, nbi, (T)kp.variance,
                     m_live);
  return 0;
}

template <typename T, typename TG, int KERNEL>
static int launch_leaf_tiles_shape(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm,
                                   const T* alpha, const TG* leaves_s, const TG* lnorm, double* part_var,
                                   double* part_mean, int64_t npad, int dp4, int64_t mpad,
                                   const KernParams& kp, const int64_t* m_live) {
#define GPSO_ARGS st, linv_p, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live
  const int bm = leaf_tiles_bm<T>(npad, dp4);
  if constexpr (sizeof(T) == 4) {
    if (bm == 512) return launch_leaf_tiles_v2<T, TG, 512, 1, KERNEL>(GPSO_ARGS);
    if (bm == 256) return launch_leaf_tiles_v2<T, TG, 256, 2, KERNEL>(GPSO_ARGS);
    return launch_leaf_tiles_v2<T, TG, 128, 4, KERNEL>(GPSO_ARGS);
  } else {
    // float64: 16 accumulator tiles of 8 VGPRs; 256 (128) rows x 16 leaves per wave keeps the number
    // of generated entries per MFMA low (the f64 kernel map is VALU-expensive)
    if (bm == 256) return launch_leaf_tiles_v2<T, TG, 256, 1, KERNEL>(GPSO_ARGS);
    return launch_leaf_tiles_v2<T, TG, 128, 2, KERNEL>(GPSO_ARGS);
  }
}

// the one-launch small call: N_pad = BM (one row block), see leaf_tiles_v2_one_kernel
template <typename T, typename TG, int BM, int CT, int KERNEL>
static int launch_leaf_tiles_one_shape(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm, const T* alpha,
                                       double* part_var, double* part_mean, int dp4, int64_t mpad, const KernParams& kp,
                                       const OneLaunch& one) {
  constexpr int RT = BM / 16, LW = 4 * CT * 16;
  const size_t lds = leaf_v2_lds_bytes<T, TG, CT>(RT, dp4);
  if (one.mode == 1 && (size_t)2 * one.d * LW * 8 > lds) return 2;  // the box state of the prologue does not fit
  const int rc = ensure_dyn_lds((const void*)leaf_tiles_v2_one_kernel<T, TG, BM, CT, KERNEL>, (int)lds);
  if (rc) return rc;
  hipLaunchKernelGGL((leaf_tiles_v2_one_kernel<T, TG, BM, CT, KERNEL>), dim3((unsigned)(mpad / LW)), dim3(256), lds, st,
                     linv_p, xs_p, xnorm, alpha, part_var, part_mean, dp4, mpad, (T)kp.variance, one);
  return 0;
}
template <typename T, typename TG, int KERNEL>
static int launch_leaf_tiles_one_k(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm, const T* alpha,
                                   double* part_var, double* part_mean, int64_t npad, int dp4, int64_t mpad,
                                   const KernParams& kp, const OneLaunch& one) {
#define GPSO_ARGS1 st, linv_p, xs_p, xnorm, alpha, part_var, part_mean, dp4, mpad, kp, one
  if constexpr (sizeof(T) == 4) {
    if (npad == 256) return launch_leaf_tiles_one_shape<T, TG, 256, 2, KERNEL>(GPSO_ARGS1);
    return launch_leaf_tiles_one_shape<T, TG, 128, 4, KERNEL>(GPSO_ARGS1);
  } else {
    if (npad == 256) return launch_leaf_tiles_one_shape<T, TG, 256, 1, KERNEL>(GPSO_ARGS1);
    return launch_leaf_tiles_one_shape<T, TG, 128, 2, KERNEL>(GPSO_ARGS1);
  }
#undef GPSO_ARGS1
}
// returns 0 (launched), 2 (not applicable: the caller runs another sequence) or a negative status
template <typename T, typename TG>
int launch_leaf_tiles_one(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm, const T* alpha,
                          double* part_var, double* part_mean, int64_t npad, int dp4, int64_t mpad, const KernParams& kp,
                          const OneLaunch& one) {
  if (npad != 128 && npad != 256) return 2;
  switch (kp.kernel) {
    case 0: return launch_leaf_tiles_one_k<T, TG, 0>(st, linv_p, xs_p, xnorm, alpha, part_var, part_mean, npad, dp4, mpad, kp, one);
    case 1: return launch_leaf_tiles_one_k<T, TG, 1>(st, linv_p, xs_p, xnorm, alpha, part_var, part_mean, npad, dp4, mpad, kp, one);
    case 2: return launch_leaf_tiles_one_k<T, TG, 2>(st, linv_p, xs_p, xnorm, alpha, part_var, part_mean, npad, dp4, mpad, kp, one);
    default: return launch_leaf_tiles_one_k<T, TG, 3>(st, linv_p, xs_p, xnorm, alpha, part_var, part_mean, npad, dp4, mpad, kp, one);
  }
}
template int launch_leaf_tiles_one<float, float>(hipStream_t, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const OneLaunch&);
template int launch_leaf_tiles_one<float, double>(hipStream_t, const float*, const double*, const double*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const OneLaunch&);
template int launch_leaf_tiles_one<double, double>(hipStream_t, const double*, const double*, const double*, const double*, double*, double*, int64_t, int, int64_t, const KernParams&, const OneLaunch&);

template <typename T, typename TG>
int launch_leaf_tiles(hipStream_t st, const T* linv_p, const TG* xs_p, const TG* xnorm,
                      const T* alpha, const TG* leaves_s, const TG* lnorm, double* part_var,
                      double* part_mean, int64_t npad, int dp4, int64_t mpad, const KernParams& kp,
                      const int64_t* m_live) {
  switch (kp.kernel) {
    case 0: return launch_leaf_tiles_shape<T, TG, 0>(GPSO_ARGS);
    case 1: return launch_leaf_tiles_shape<T, TG, 1>(GPSO_ARGS);
    case 2: return launch_leaf_tiles_shape<T, TG, 2>(GPSO_ARGS);
    default: return launch_leaf_tiles_shape<T, TG, 3>(GPSO_ARGS);
  }
#undef GPSO_ARGS
}
template int launch_leaf_tiles<float, float>(hipStream_t, const float*, const float*, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*);
template int launch_leaf_tiles<float, double>(hipStream_t, const float*, const double*, const double*, const float*, const double*, const double*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*);
template int launch_leaf_tiles<double, double>(hipStream_t, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*);

void launch_leaf_finalize(hipStream_t st, const double* part_var, const double* part_mean, int nbi,
                          int64_t mpad, int64_t m, const KernParams& kp, double varsigma,
                          double* mean, double* var, double* ucb, const void* lnorm, int lnorm_f64) {
  const int64_t blocks = (m + 255) / 256;
  if (blocks == 0) return;
  hipLaunchKernelGGL(leaf_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part_var,
                     part_mean, nbi, mpad, m, kp, varsigma, mean, var, ucb, lnorm, lnorm_f64);
}

void launch_seg_argmax(hipStream_t st, const double* mean, const double* var, const double* ucb,
                       const int64_t* seg_off_dev, int nseg, int nblk, void* partial_dev,
                       double* out_vals_dev, const LeafFinalize* fin, double* host_vals, double done_token) {
  if (fin != nullptr)
    hipLaunchKernelGGL(seg_argmax_stage1<true>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb,
                       seg_off_dev, reinterpret_cast<Best*>(partial_dev), *fin);
  else
    hipLaunchKernelGGL(seg_argmax_stage1<false>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb,
                       seg_off_dev, reinterpret_cast<Best*>(partial_dev), LeafFinalize{});
  hipLaunchKernelGGL(seg_argmax_stage2, dim3((unsigned)nseg), dim3(256), 0, st,
                     reinterpret_cast<const Best*>(partial_dev), nblk, seg_off_dev, mean, var, ucb, nseg,
                     out_vals_dev, host_vals, done_token);
}

void launch_keyed_argmax(hipStream_t st, const double* mean, const double* var, const double* ucb,
                         const int64_t* key_dev, int64_t rows, int64_t uniq, int nseg, const int64_t* live_dev,
                         int nblk, void* partial_dev, int64_t* pos_dev, double* out_vals_dev, const LeafFinalize* fin,
                         double* host_vals, double done_token) {
  if (fin != nullptr)
    hipLaunchKernelGGL(keyed_argmax_stage1<true>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb, key_dev,
                       rows, uniq, nseg, live_dev, reinterpret_cast<Best*>(partial_dev), pos_dev, *fin);
  else
    hipLaunchKernelGGL(keyed_argmax_stage1<false>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb, key_dev,
                       rows, uniq, nseg, live_dev, reinterpret_cast<Best*>(partial_dev), pos_dev, LeafFinalize{});
  hipLaunchKernelGGL(keyed_argmax_stage2, dim3((unsigned)nseg), dim3(256), 0, st,
                     reinterpret_cast<const Best*>(partial_dev), pos_dev, nblk, rows, mean, var, ucb, live_dev,
                     nseg, out_vals_dev, host_vals, done_token);
}

void launch_reduce_winners(hipStream_t st, const double* gathered, const int64_t* base, int world, int nseg,
                           int stride, double* out, double* host_out, double done_token) {
  hipLaunchKernelGGL(reduce_winners_kernel, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, st, gathered, base,
                     world, nseg, stride, out, host_out, done_token);
}

// splitmix64-style finaliser of (word, position, salt); summed with wrap-around
__global__ __launch_bounds__(256) void hash_words_kernel(const uint64_t* __restrict__ w, size_t n, uint64_t salt,
                                                         unsigned long long* __restrict__ acc) {
  uint64_t sum = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t z = w[i] ^ ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull) ^ (salt * 0xD1B54A32D192ED03ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    sum += z ^ (z >> 31);
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(acc, (unsigned long long)sum);
}

void launch_hash_words(hipStream_t st, const void* words, size_t nwords, uint64_t salt, unsigned long long* acc) {
  if (nwords == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((nwords + 255) / 256, 2048);
  hipLaunchKernelGGL(hash_words_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const uint64_t*>(words), nwords, salt, acc);
}

void launch_chunk_live(hipStream_t st, const int64_t* live_dev, int64_t chunk, int nchunk, int64_t* out_dev) {
  hipLaunchKernelGGL(chunk_live_kernel, dim3((unsigned)((nchunk + 63) / 64)), dim3(64), 0, st, live_dev, chunk,
                     nchunk, out_dev);
}

}  // namespace gpso
