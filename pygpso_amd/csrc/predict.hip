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

namespace gpso {


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

// one leaf: sum the row blocks' partials, form mean / var (+ noise) / ucb, store them; returns ucb
__device__ __forceinline__ double finalize_leaf(const LeafFinalize& f, int64_t j) {
  double v = 0, mu = 0;
  for (int b = 0; b < f.nbi; ++b) {
    v += f.part_var[(int64_t)b * f.mpad + j];
    mu += f.part_mean[(int64_t)b * f.mpad + j];
  }
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

// ---------------------------------------------------------------------------------------------
// Packed operand layouts produced at fit time (fit.hip: pack_linv_kernel / scale_x_kernel):
//   linv_p : 16x16 tiles of L^-1 (lower tiles only, row-major over the triangle: tile (rt, kt <= rt)
//            at ((rt (rt + 1) / 2 + kt) * 256)); inside a tile element (row, k) sits at
//            lane * 4 + (k & 3) with lane = (row & 15) + 16 * ((k & 15) >> 2)
//            -> one wave reads a whole tile as 64 contiguous vec4 (1 KiB f32 / 2 KiB f64), and
//            lane l gets L^-1[row l&15][k = 4 (l>>4) + 0..3]: the A operand of MFMA k-step r is
//            element r.
//   xs_p   : scaled training inputs as MFMA A fragments IN THE GENERATION TYPE TG:
//            ((kt * dp4 + c) * 64 + lane) holds x~[16 kt + arow_for_k4(lane & 15)][4 c + (lane >> 4)]
// With those, accumulator register r of lane l of the generated tile S = x~ x~*^T corresponds to
// training row 16 kt + 4 (l >> 4) + r and leaf column (l & 15) for BOTH the f32 and f64 MFMA,
// which is exactly the B-operand shape (k = l >> 4 within k-step r) the second MFMA needs.
//
// Generation type TG vs apply type T.  float contexts generate the cross-Gram tile with TG = double
// by default ("accurate generation"): GPflow's GEMM-form r^2 = |x|^2 + |x*|^2 - 2 x.x* is a
// cancellation of terms of size |x / l|^2 (~100 at the reference's lengthscales), which in float
// leaves an absolute error ~1e-5 in r^2 -- and L^-1 (entries up to 1/sigma_n ~ 1e3 at the
// reference's noise floor) amplifies that into a variance error of 1e-3 sigma^2 (measured,
// profiles/r02a_precision_before.jsonl).  The x.x* contraction is D/4 MFMAs per 16x16 tile against
// 64..128 for the apply, so running it on v_mfma_f64_16x16x4_f64 costs a few percent; r^2 is
// combined in double and only then rounded to float for the Matern / SE map.
// =============================================================================================
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the same with the instruction's immediate offset (added to the global AND the LDS address)
template <int OFF>
__device__ __forceinline__ void glds16_off(const void* gsrc_lane, void* lds_wave_base) {
  static_assert(OFF >= -4096 && OFF < 4096, "13-bit signed immediate");
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}
template <int OFF>
__device__ __forceinline__ void glds4_off(const void* gsrc_lane, void* lds_wave_base) {
  static_assert(OFF >= -4096 && OFF < 4096, "13-bit signed immediate");
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, OFF, 0);
}
__device__ __forceinline__ void glds4(const void* gsrc_lane, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
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

#ifndef GPSO_BSTAMP
#define GPSO_BSTAMP(q, i)  // tools/micro/leaf_bf16_phases.hip defines this to record s_memtime stamps
#endif
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
    if (last_flag) __threadfence();
  }
  __syncthreads();
  if (!last_flag) return;
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
                                                             int64_t npad, u32x4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (rt, kq, lane)
  const int64_t npad16 = npad / 16, npad32 = npad / 32;
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
// (exact): L^-1 by 2^sa with max |L^-1| 2^sa in [2^13, 2^14) (found on the device at packing time, absmax_kernel), the
// generated tile by 2^sb with sigma^2 2^sb in [2^13, 2^14) (folded into the variance the kernel map multiplies with).
// An entry 2^-28 below its operand's maximum is still a normal fp16 number; below that the ABSOLUTE error per entry
// stays under 2^-25 of the scaled maximum -- far inside the 2^-22 the dropped product costs.  Accumulation is f32 as
// everywhere; the epilogue undoes the scales (sum of squares x 2^-2(sa+sb), mean x 2^-sb).
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

// max |linv| over the lower triangle of the first n rows -> out[0] (as float bits: non-negative floats order like
// unsigned integers); out[0] must be zero on entry.  One workgroup per 16 rows, 16 lanes per row, four columns per load.
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
template <typename TF>
__global__ __launch_bounds__(256) void pack_linv_f16_kernel(const TF* __restrict__ linv, int64_t n, int64_t npad,
                                                            float* __restrict__ scal, u32x4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (rt, kq, lane)
  const int64_t npad16 = npad / 16, npad32 = npad / 32;
  int e = 0;
  (void)frexpf(fmaxf(scal[0], 1e-30f), &e);  // max = m 2^e, m in [0.5, 1)
  const float up = ldexpf(1.0f, 14 - e);
  if (idx == 0) scal[1] = ldexpf(1.0f, e - 14);
  if (idx >= npad16 * npad32 * 64) return;
  const int lane = (int)(idx & 63);
  const int64_t kq = (idx >> 6) % npad32, rt = (idx >> 6) / npad32;
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

template <typename TG, int C16 = 0 /* chunks of 32 slots of the fp16 contraction; 0: the contraction in TG */>
struct Bf16Lds {
  // bytes of the X fragments of one k-step: 2 k-tiles x D_pad / 4 groups (TG), or 2 k-tiles x chunks of 32 dimensions x
  // 2 fp16 pieces x 1 KB (C16); a wave's leaf fragments take as much
  static __host__ __device__ constexpr int xfrag(int dp4) { return C16 ? C16 * 4096 : 2 * dp4 * 64 * (int)sizeof(TG); }
  // bytes of one X buffer: the fragments, 32 norms (TG, padded to 64), 32 alphas (float, padded to 64)
  static __host__ __device__ constexpr int xbytes(int dp4) { return xfrag(dp4) + 64 * (int)sizeof(TG) + 256; }
};
// the x.x* contraction of one k-step of a wave: s[h][t] += (16 training points of half h) x (16 leaves of column tile t).
// Float / double: software-pipelined over the groups of four dimensions -- the operands of group c + 1 are on their way
// from LDS while the MFMAs of group c issue.  C16: three fp16 products per chunk of 32 dimensions, small terms first.
template <typename TG, int C16, int CT>
__device__ __forceinline__ void leaf_contract(int lane, int dp4, const unsigned char* xs_b, const TG* xb,
                                              typename Mfma<TG>::vec4 (&s)[2][CT]) {
  if constexpr (C16 != 0) {
    static_assert(sizeof(TG) == 4, "the fp16 contraction belongs to float generation");
    constexpr int nc = C16;
    const u32x4* xa = reinterpret_cast<const u32x4*>(xs_b);  // [h][cc][piece][64]
    const u32x4* lb = reinterpret_cast<const u32x4*>(xb);    // [t][cc][piece][64]
#pragma unroll
    for (int cc = 0; cc < nc; ++cc) {
      u32x4 a[2][2], b[CT][2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) a[h][pc] = xa[((h * nc + cc) * 2 + pc) * 64 + lane];
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) b[t][pc] = lb[((t * nc + cc) * 2 + pc) * 64 + lane];
#define GPSO_XX(PA, PB)                                                                                              \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) _Pragma("unroll") for (int t = 0; t < CT; ++t) s[h][t] =             \
      __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[h][PA]), __builtin_bit_cast(f16x8, b[t][PB]), s[h][t], 0, 0, 0)
      GPSO_XX(1, 0);
      GPSO_XX(0, 1);
      GPSO_XX(0, 0);
#undef GPSO_XX
    }
  } else {
    using MG = Mfma<TG>;
    constexpr int XB = 64 * (int)sizeof(TG);
    TG x0 = reinterpret_cast<const TG*>(xs_b)[lane];
    TG x1 = reinterpret_cast<const TG*>(xs_b + dp4 * XB)[lane];
    TG l[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) l[t] = xb[(t * dp4) * 64 + lane];
    for (int c = 0; c < dp4; ++c) {
      TG x0n = x0, x1n = x1, ln[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) ln[t] = l[t];
      if (c + 1 < dp4) {
        x0n = reinterpret_cast<const TG*>(xs_b + (c + 1) * XB)[lane];
        x1n = reinterpret_cast<const TG*>(xs_b + (dp4 + c + 1) * XB)[lane];
#pragma unroll
        for (int t = 0; t < CT; ++t) ln[t] = xb[(t * dp4 + c + 1) * 64 + lane];
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        s[0][t] = MG::mma(x0, l[t], s[0][t]);
        s[1][t] = MG::mma(x1, l[t], s[1][t]);
      }
      x0 = x0n;
      x1 = x1n;
#pragma unroll
      for (int t = 0; t < CT; ++t) l[t] = ln[t];
    }
  }
}

// One k-step (32 training points) of the split-bf16 leaf tile is two stretches of very different kind: the
// GENERATION of this wave's 32 x 32 cross-Gram values (x.x* contraction, kernel map, split into bf16 pieces:
// vector ALU work) and the APPLY (192 bf16 MFMAs: 3072 clocks of the matrix pipe).  Two waves share a SIMD, and a
// workgroup barrier per step starts them together: run in the same order they fight for the vector ALU, then
// queue for the matrix pipe.  So the waves of a SIMD run the two stretches in OPPOSITE order (waves 0-3 generate
// step q, then apply it; waves 4-7 apply step q with the pieces they generated during step q - 1, then generate
// step q + 1): one wave's vector work runs under the other's MFMAs.
//
// Round 4 -- the map of the split kernels, written for the vector ALU's issue slots (a wave's k-step is as long as
// its OWN instruction stream: 14 vector instructions per generated value, issued value by value with every
// transcendental waiting on the instruction in front of it, were 3 400 of a step's 7 200 clocks):
//   * the scale of the exponent is folded into the norms and the contraction's multiplier: with
//     u = SC r^2, SC = C2 log2(e)^2 (SE: log2(e) / 2), t' = sqrt(u) = log2(e) sqrt(C2) r and k = exp2(-t') P(t'), where
//     P carries sigma^2 in its coefficients: sigma^2 (1 + ln2 t' + ln2^2 / 3 t'^2) for Matern-5/2 -- no separate
//     multiplications by log2(e) and by sigma^2;
//   * float generation takes sqrt(|u|) (a source modifier) instead of clamping: a GEMM-form r^2 that rounds to -1e-6
//     is as wrong as one that rounds to +1e-6, and the map's error is the same second-order term either way
//     (double generation keeps GPflow's clamp: its r^2 is exact to 1e-15);
//   * the second piece of the fp16 split comes from v_fma_mixlo / mixhi_f16 (a - (float)h, exact, rounded once to fp16);
//   * stage-major order: all combines, all square roots, all exponentials, all polynomials -- no instruction waits
//     on the one in front of it.
// 9 vector instructions per value instead of 14.
template <int KERNEL>
struct GenScale {
  static constexpr double kLog2e = 1.44269504088896340736;
  static constexpr double SC = (KERNEL == 3) ? 0.5 * kLog2e : KernScale<KERNEL>::C2 * kLog2e * kLog2e;
};
// (a, b) -> the packed fp16 pair AND the packed pair of the remainders' fp16 roundings (the second piece of a two-piece
// split): v_fma_mixlo / mixhi_f16 form a - (float)h exactly and round it once to fp16 into the low / high half -- the
// bits of f16_split_pair applied twice, three instructions per pair instead of four
__device__ __forceinline__ void f16_split_pair_both(float a, float b, unsigned& h, unsigned& l) {
  const f32x2 v = {a, b};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
}
// vc: sigma^2 (x 2^sb under the fp16 split) times the polynomial's coefficients in t' (see above)
template <int KERNEL>
__device__ __forceinline__ void gen_poly_coeffs(float variance, float (&vc)[3]) {
  constexpr float kLn2 = 0.69314718055994530942f;
  vc[0] = variance;
  vc[1] = variance * kLn2;
  vc[2] = variance * (kLn2 * kLn2 / 3.0f);
}
template <int NS, typename TG, int KERNEL, bool F16, bool DIAG, int C16 = 0, int CT = 2>
__device__ __forceinline__ void leaf_bf16_gen(int lane, int dp4,
                                              const unsigned char* xs_b /* [2][dp4] X fragments | norms | alpha */,
                                              const TG* xb, const TG (&nb)[CT] /* SC |x*|^2 */, const TG cm /* -2 SC (C16: x 2^-2sx) */,
                                              const float (&vc)[3], bf16x8 (&bfrag)[NS][CT], float (&macc)[CT]) {
  using MG = Mfma<TG>;
  using vecG = typename MG::vec4;
  constexpr TG SC = (TG)GenScale<KERNEL>::SC;
  const int XF = Bf16Lds<TG, C16>::xfrag(dp4);
  // ---- generate the two 16-point tiles of this k-step (TG) --------------------------------------
  vecG s[2][CT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < CT; ++t) s[h][t] = vecG{0, 0, 0, 0};
  // norms and alpha of the 32 points of this k-step arrived in LDS with the panel (no ordinary global
  // load inside the loop: one issued after the LDS-DMA makes hipcc drain the DMA queue at its use)
  const TG* nrm = reinterpret_cast<const TG*>(xs_b + XF);
  const float* alp = reinterpret_cast<const float*>(xs_b + XF + 64 * sizeof(TG));
  // the contraction (the norms are fetched in front of it: a generator wave's step is a latency chain, not an issue
  // budget -- stamps: tools/micro/leaf_spec_phases.hip)
  vecG nav[2];
  if constexpr (C16 == 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) nav[h] = *reinterpret_cast<const vecG*>(nrm + 16 * h + 4 * (lane >> 4));
  }
  leaf_contract<TG, C16, CT>(lane, dp4, xs_b, xb, s);
  float p[CT][8];
  // stage 0: u = SC r^2, GPflow's GEMM form combined in TG, rounded to float
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if constexpr (C16 != 0) {  // (the training input's norm arrived inside the contraction)
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[t][4 * h + r] = (float)fma_t(cm, s[h][t][r], nb[t]);
    } else {
      const vecG na = nav[h] * SC;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[t][4 * h + r] = (float)fma_t(cm, s[h][t][r], na[r] + nb[t]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float e[CT][8];
  if constexpr (KERNEL != 3) {  // stage 1: t' = sqrt(u)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (sizeof(TG) == 4) p[t][j] = __builtin_amdgcn_sqrtf(__builtin_fabsf(p[t][j]));
        else p[t][j] = __builtin_amdgcn_sqrtf(fmaxf(p[t][j], (float)(GenScale<KERNEL>::SC * 1e-36)));
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  // stage 2: e = exp2(-t')  (SE: exp2(-u))
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) e[t][j] = __builtin_amdgcn_exp2f(-p[t][j]);
  __builtin_amdgcn_sched_barrier(0);
  // stage 3: k = e P(t')
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (KERNEL == 0) p[t][j] = fmaf(p[t][j], fmaf(p[t][j], vc[2], vc[1]), vc[0]) * e[t][j];
      else if constexpr (KERNEL == 1) p[t][j] = fmaf(p[t][j], vc[1], vc[0]) * e[t][j];
      else p[t][j] = vc[0] * e[t][j];
    }
  if constexpr (DIAG) {  // this k-step lies in the diagonal block: its share of k*.alpha (f32, before the split)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(alp + 16 * h + 4 * (lane >> 4));
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          macc[t] = fma_t(p[t][4 * h + r], a4[r], macc[t]);
          // keeps the two tiles' means in separate registers: packed by the SLP vectoriser into one register pair
          // and a chain of dependent v_pk_fma_f32, the high half (t = 1) came back wrong now and then in waves
          // 4-7 -- 33 of 400 runs of a D = 3 posterior, 27 of 150 of a C3 posterior in the bf16x3 kernel; 0 with
          // this line (profiles/r02h_packed_mean_bug.txt; predict.hip is also built with -fno-slp-vectorize)
          asm volatile("" : "+v"(macc[t]));
        }
    }
  }
  // ---- split into bf16 / fp16 pieces: B operands --------------------------------------------------
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    if constexpr (F16) {
      u32x4 f0, f1;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        unsigned hh, ll;
        f16_split_pair_both(p[t][2 * h], p[t][2 * h + 1], hh, ll);
        f0[h] = hh;
        f1[h] = ll;
      }
      bfrag[0][t] = __builtin_bit_cast(bf16x8, f0);  // (fp16 pieces travel in the same 16-byte registers)
      bfrag[1][t] = __builtin_bit_cast(bf16x8, f1);
    } else {
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        u32x4 f;
#pragma unroll
        for (int h = 0; h < 4; ++h) f[h] = bf16_split_pair(p[t][2 * h], p[t][2 * h + 1]);
        bfrag[sp][t] = __builtin_bit_cast(bf16x8, f);
      }
    }
  }
}

// apply: acc[rt][t] += sum over the kept piece products, small terms first.  DIAG: k-steps of the diagonal block --
// the tiles above the diagonal are all zero and skipped per row tile; off-diagonal steps are one branch-free stretch
template <int NS, bool F16, bool DIAG>
__device__ __forceinline__ void leaf_bf16_apply(int q, int q_diag0, int lane, const u32x4* panel_b /* [NS][16][64] */,
                                                const bf16x8 (&bfrag)[NS][2], f32x4 (&acc)[16][2]) {
  static_assert(!F16 || NS == 2, "the fp16 split has two pieces");
  constexpr int RT = 16, CT = 2;
  u32x4 a[2][NS];
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) a[0][sp] = panel_b[(sp * RT + 0) * 64 + lane];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (rt + 1 < RT) {
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) a[(rt + 1) & 1][sp] = panel_b[(sp * RT + rt + 1) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (DIAG && 2 * (q - q_diag0) > rt) continue;  // all-zero tiles above the diagonal
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      f32x4 c = acc[rt][t];
#define GPSO_BF(SA, SB)                                                                                                   \
  c = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[rt & 1][SA]), __builtin_bit_cast(f16x8, bfrag[SB][t]), c, 0, 0, 0) \
          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[rt & 1][SA]), bfrag[SB][t], c, 0, 0, 0)
      if constexpr (NS == 3) {
        GPSO_BF(2, 0);
        GPSO_BF(0, 2);
        GPSO_BF(1, 1);
      }
      GPSO_BF(1, 0);
      GPSO_BF(0, 1);
      GPSO_BF(0, 0);
#undef GPSO_BF
      acc[rt][t] = c;
    }
  }
}

// ---- the FUSED step (round 4): apply of step q with the generation of step q + 1 dealt into its MFMA shadows -------
// The two-phase step above runs generation and apply as two stretches per wave and relies on the partner wave of the
// SIMD for overlap; a wave issues in order, so its step is the SUM of its stretches (stamps: DMA 650 + generation 2 700
// + apply 1 900 + barriers 750 .. 1 600 = 7 000 clocks per step for a matrix pipe that works 3 840 of them).  But an
// MFMA only holds the vector issue port for 8 of its 16 clocks: the SAME wave can issue one or two vector instructions
// behind every MFMA for free.  So, as leaf_tiles_v2_kernel does for the f32 kernel: the contraction MFMAs of step q + 1
// go first (their results mature under row tile 0's MFMAs), then the map and the split of step q + 1 -- in stage-major
// order: 16 combines, 16 square roots, 16 exponentials, 16 polynomials, 8 pair splits -- are dealt over row tiles
// 1 .. 15 of the apply of step q, a few instructions behind each tile's six MFMAs.  All eight waves run the same
// stream, one workgroup barrier per step.  Same operations on the same operands as the two-phase step: bit-identical
// partial sums (tests/test_gpu_parity.py compares the two kernels).
// GMODE: 0 = nothing to generate (last step), 1 = generate step q + 1, 2 = ... and accumulate its share of k*.alpha
// ASKIP: the first ASKIP row tiles of step q are all zero (step j of the diagonal block: 2 j tiles above the diagonal) --
// a compile-time count: the eight steps of the diagonal block are eight straight-line copies.  With the skip as a
// run-time test per row tile the compiler kept the accumulators of skipped tiles alive through 74 register-pair moves
// per step (disassembly), in the steps that already have the least matrix work to hide them behind.
template <int NS, typename TG, int KERNEL, bool F16, int ASKIP, int GMODE, int C16 = 0>
__device__ __forceinline__ void leaf_bf16_fused_step(int q, int q_diag0, int lane, int dp4,
                                                     const u32x4* panel_b /* [NS][16][64]: L^-1 pieces of step q */,
                                                     const unsigned char* xs_n /* inputs of step q + 1 */, const TG* xb,
                                                     const TG (&nb)[2], const TG cm, const float (&vc)[3],
                                                     const bf16x8 (&bcur)[NS][2], bf16x8 (&bnxt)[NS][2],
                                                     f32x4 (&acc)[16][2], float (&macc)[2]) {
  using MG = Mfma<TG>;
  using vecG = typename MG::vec4;
  constexpr int RT = 16, CT = 2;
  constexpr TG SC = (TG)GenScale<KERNEL>::SC;
  constexpr bool GEN = GMODE != 0;
  const int XF = Bf16Lds<TG, C16>::xfrag(dp4);
  // ---- contraction of step q + 1 (TG), software-pipelined over the groups of four dimensions ------------------------
  vecG s[2][CT];
  vecG nav[2];
  if constexpr (GEN) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < CT; ++t) s[h][t] = vecG{0, 0, 0, 0};
    if constexpr (C16 == 0) {
      const TG* nrm = reinterpret_cast<const TG*>(xs_n + XF);
#pragma unroll
      for (int h = 0; h < 2; ++h) nav[h] = *reinterpret_cast<const vecG*>(nrm + 16 * h + 4 * (lane >> 4));
    }
    leaf_contract<TG, C16, CT>(lane, dp4, xs_n, xb, s);
  }
  // ---- apply of step q, the map of step q + 1 dealt over row tiles 1 .. 15 --------------------------------------------
  // value e = 8 t + j, j = 4 h + r (column tile t, 16-point half h, accumulator register r); ops in stage-major order --
  // within a stage every op is independent of its neighbours, and an op's input is at least 16 ops old:
  //   norms x SC (8) | combine (16) | sqrt (16; none for the squared exponential) | exp2 (16) | polynomial x exponential
  //   (16) | k*.alpha (16; GMODE 2) | split of pair (t, j) (8)
  constexpr int E = 16;
  constexpr int O_COMB = C16 != 0 ? 0 : 8, O_SQRT = O_COMB + E, O_EXP = O_SQRT + (KERNEL == 3 ? 0 : E), O_POLY = O_EXP + E,
                O_MEAN = O_POLY + E, O_SPLIT = O_MEAN + (GMODE == 2 ? E : 0), NOPS = O_SPLIT + 8;
  float p[CT][8], ex[CT][8];
  TG na[2][4];
  f32x4 al4[2];
  u32x4 fr[NS][CT];
  if constexpr (GMODE == 2) {
    const float* alp = reinterpret_cast<const float*>(xs_n + XF + 64 * sizeof(TG));
#pragma unroll
    for (int h = 0; h < 2; ++h) al4[h] = *reinterpret_cast<const f32x4*>(alp + 16 * h + 4 * (lane >> 4));
  }
  auto op = [&](auto o_) {
    constexpr int o = decltype(o_)::value;
    if constexpr (o < O_COMB) {
      na[o >> 2][o & 3] = nav[o >> 2][o & 3] * SC;
    } else if constexpr (o < O_SQRT) {
      constexpr int e = o - O_COMB, t = e >> 3, h = (e >> 2) & 1, r = e & 3;
      if constexpr (C16 != 0) p[t][4 * h + r] = (float)fma_t(cm, s[h][t][r], nb[t]);
      else p[t][4 * h + r] = (float)fma_t(cm, s[h][t][r], na[h][r] + nb[t]);
    } else if constexpr (o < O_EXP) {
      constexpr int e = o - O_SQRT, t = e >> 3, j = e & 7;
      if constexpr (sizeof(TG) == 4) p[t][j] = __builtin_amdgcn_sqrtf(__builtin_fabsf(p[t][j]));
      else p[t][j] = __builtin_amdgcn_sqrtf(fmaxf(p[t][j], (float)(GenScale<KERNEL>::SC * 1e-36)));
    } else if constexpr (o < O_POLY) {
      constexpr int e = o - O_EXP, t = e >> 3, j = e & 7;
      ex[t][j] = __builtin_amdgcn_exp2f(-p[t][j]);
    } else if constexpr (o < O_MEAN) {
      constexpr int e = o - O_POLY, t = e >> 3, j = e & 7;
      if constexpr (KERNEL == 0) p[t][j] = fmaf(p[t][j], fmaf(p[t][j], vc[2], vc[1]), vc[0]) * ex[t][j];
      else if constexpr (KERNEL == 1) p[t][j] = fmaf(p[t][j], vc[1], vc[0]) * ex[t][j];
      else p[t][j] = vc[0] * ex[t][j];
    } else if constexpr (o < O_SPLIT) {  // (GMODE 2) k*.alpha in f32, before the split; per column tile in the order j = 0 .. 7
      constexpr int e = o - O_MEAN, t = e >> 3, j = e & 7;
      macc[t] = fma_t(p[t][j], al4[j >> 2][j & 3], macc[t]);
      asm volatile("" : "+v"(macc[t]));  // the two tiles' means stay in separate registers (see leaf_bf16_gen)
    } else {
      constexpr int e = o - O_SPLIT, t = e >> 2, j = e & 3;
      if constexpr (F16) {
        unsigned hh, ll;
        f16_split_pair_both(p[t][2 * j], p[t][2 * j + 1], hh, ll);
        fr[0][t][j] = hh;
        fr[1][t][j] = ll;
      } else {
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) fr[sp][t][j] = bf16_split_pair(p[t][2 * j], p[t][2 * j + 1]);
      }
    }
  };
  static_assert(ASKIP >= 0 && ASKIP < RT, "at least one live row tile");
  u32x4 a[2][NS];
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) a[ASKIP & 1][sp] = panel_b[(sp * RT + ASKIP) * 64 + lane];
  static_for<0, RT>([&](auto rt_) {
    constexpr int rt = decltype(rt_)::value;
    if constexpr (rt + 1 < RT && rt + 1 > ASKIP) {
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) a[(rt + 1) & 1][sp] = panel_b[(sp * RT + rt + 1) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (rt >= ASKIP) {  // (diagonal block: all-zero tiles above the diagonal)
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        f32x4 c = acc[rt][t];
#define GPSO_BF(SA, SB)                                                                                                   \
  c = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[rt & 1][SA]), __builtin_bit_cast(f16x8, bcur[SB][t]), c, 0, 0, 0) \
          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[rt & 1][SA]), bcur[SB][t], c, 0, 0, 0)
        if constexpr (NS == 3) {
          GPSO_BF(2, 0);
          GPSO_BF(0, 2);
          GPSO_BF(1, 1);
        }
        GPSO_BF(1, 0);
        GPSO_BF(0, 1);
        GPSO_BF(0, 0);
#undef GPSO_BF
        acc[rt][t] = c;
      }
    }
    if constexpr (GEN && rt >= 1) {  // this row tile's share of the map: ops [(rt - 1) NOPS / 15, rt NOPS / 15)
      static_for<(rt - 1) * NOPS / (RT - 1), rt * NOPS / (RT - 1)>(op);
    }
  });
  if constexpr (GEN) {
#pragma unroll
    for (int sp = 0; sp < NS; ++sp)
#pragma unroll
      for (int t = 0; t < CT; ++t) bnxt[sp][t] = __builtin_bit_cast(bf16x8, fr[sp][t]);
  }
}

// F16: the fp16 split (two pieces, three products); `variance` then arrives multiplied by 2^sb, inv_scale_a[1] is
// 2^-sa (device, written by pack_linv_f16_kernel) and inv_scale_b = 2^-sb
// FUSED: every wave runs the fused step (apply of step q with the generation of step q + 1 in its MFMA shadows, one
// barrier per step); otherwise round 3's two-phase step with the waves of a SIMD in opposite order
// C16: the contraction on the fp16 pipe -- xs_p then points at the fp16 piece pairs of the scaled inputs
// (pack_xs_f16_kernel's order) and c16_scale at their scale (device: [1] = 2^sx, [2] = 2^-2sx)
template <int NS, typename TG, int KERNEL, bool F16 = false, bool FUSED = false, int C16 = 0>
__global__ __launch_bounds__(512, 2) void leaf_tiles_bf16_kernel(
    const u32x4* __restrict__ linv_b, const TG* __restrict__ xs_p, const TG* __restrict__ xnorm,
    const float* __restrict__ alpha, const TG* __restrict__ leaves_s,
    const TG* __restrict__ lnorm, double* __restrict__ part_var, double* __restrict__ part_mean,
    int npad16, int dp4, int64_t mpad, int nbi, float variance, const int64_t* __restrict__ m_live,
    const float* __restrict__ inv_scale_a, float inv_scale_b, const float* __restrict__ c16_scale) {
  constexpr int RT = 16, CT = 2, NW = 8;
  constexpr TG SC = (TG)GenScale<KERNEL>::SC;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  if (m_live != nullptr && (int64_t)blockIdx.x * (NW * CT * 16) >= *m_live) return;  // workgroup-uniform
  u32x4* panel = reinterpret_cast<u32x4*>(lds_raw);                 // [2][NS][RT][64]
  unsigned char* xsl = reinterpret_cast<unsigned char*>(panel + 2 * NS * RT * 64);  // [3] X buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xstride = Bf16Lds<TG, C16>::xbytes(dp4);
  const int xfrag = Bf16Lds<TG, C16>::xfrag(dp4);
  // this wave's leaf fragments: [CT][dp4][64] TG, or (C16) [CT][chunk][piece][64] x 16 bytes
  TG* xb = reinterpret_cast<TG*>(xsl + 3 * xstride + (size_t)wave * xfrag);

  const int bi = nbi - 1 - (int)blockIdx.y;
  const int64_t col0 = ((int64_t)blockIdx.x * NW + wave) * (CT * 16);
  const int dp = dp4 * 4;
  const int npad32 = npad16 / 2;
  const int q_diag0 = bi * (RT / 2), q_end = q_diag0 + RT / 2;

  // ---- LDS-DMA duties, dealt EVENLY over the eight waves (round 4) -------------------------------------------
  // A step's DMAs are NS x 16 fragments of the L^-1 pieces (1 KB each), the X fragments of the step (256-byte pieces)
  // and its 32 norms and 32 alphas.  Until round 3 waves 0 .. NS*RT/8 - 1 carried eight fragments each, wave 6 all
  // X pieces and wave 7 norms and alphas: an LDS-DMA costs its wave ~90 clocks to issue, so the panel waves spent 740
  // clocks of a 6 750-clock step there, and wave 6 -- which generates and applies like every other wave -- arrived last
  // at every barrier (890 + 310 clocks of barrier wait on the others; stamps: tools/micro/leaf_bf16_phases.hip).  Now every
  // wave moves NS*RT/8 = 4 (6) fragments + the X pieces w, w + 8, w + 16, w + 24 + (waves 6 / 7) norms / alphas.
  // Each group is one LDS window addressed as M0 + the instruction's 13-bit immediate offset, which the hardware adds
  // to the global and the LDS address alike (the global base is pre-biased by the same amount): one M0 write and one
  // scalar add per DMA instead of ~25 scalar instructions of address arithmetic.
  // DW waves carry the DMA duties.  Fused step, two pieces: waves 0-3 ONLY -- the SIMD's arbiter serves its older wave
  // first, so waves 0-3 are through a fused step in 3 700 clocks and would wait 1 800 at the barrier for waves 4-7, which
  // need 5 000 (stamps, tools/micro/leaf_bf16_phases.hip): the DMA issue (~650 clocks per wave when dealt evenly) is the
  // work that can be moved, and it goes to the waves that have the time.  Otherwise all eight waves share it.
  constexpr int DW = (FUSED != 0 && NS == 2) ? NW / 2 : NW;
  constexpr int FPW = NS * RT / DW;  // fragments of the L^-1 pieces per DMA wave: 8 / 4 (two pieces) or 6 (three)
  static_assert(FPW * DW == NS * RT && FPW * 1024 <= 8192 && RT == 16, "window of a wave's fragments");
  const bool dma_wave = wave < DW;
  const int lane16 = lane * 16, lane4 = lane * 4;
  const unsigned char* pgb[FPW];
#pragma unroll
  for (int j = 0; j < FPW; ++j) {
    const int f = FPW * (wave % DW) + j, sp = f / RT, rt = f % RT;  // fragment f = piece sp, row tile rt
    pgb[j] = reinterpret_cast<const unsigned char*>(linv_b + ((size_t)sp * npad16 + (bi * RT + rt)) * npad32 * 64) -
             (j * 1024 - FPW * 512);
  }
  // (the empty asm keeps a wave-uniform address in scalar registers: left alone, the compiler hoists
  // base + lane offset out of the loop as per-lane 64-bit pointers and spills them)
  auto uniform = [](const unsigned char* p) {
    const unsigned long long g = (unsigned long long)p;
    unsigned lo = (unsigned)g, hi = (unsigned)(g >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
  };
  auto issue_panel = [&](int q, int buf) {
    if (!dma_wave) return;
    unsigned char* centre = reinterpret_cast<unsigned char*>(panel) + buf * (NS * RT * 1024) + wave * (FPW * 1024) + FPW * 512;
    static_for<0, FPW>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      glds16_off<j * 1024 - FPW * 512>(uniform(pgb[j] + (size_t)q * 1024) + lane16, centre);
    });
  };
  // the inputs of a k-step: a ring of three buffers (step q + 1 is generated during step q).  Piece r of
  // the X fragments is bytes [256 r, 256 r + 256) of the step's contiguous source block and of the buffer alike
  // (<= 32 pieces: D <= 48 in float, <= 32 in double -- checked at launch); DMA wave w moves pieces w + DW jj.  The last
  // but one DMA wave also moves the 32 norms (TG, as 64 dwords; float: lanes 32-63 fetch duplicates into the unused
  // half), the last one the 32 alphas 64 TG behind them.
  // C16: the fragments are 4 C16 <= 8 pieces of 1 KB, one 16-byte DMA each, dealt the same way; no norms (they ride in
  // the contraction).
  constexpr int XPB = C16 ? 1024 : 256;  // bytes of an X piece
  const int xpieces = xfrag / XPB;
  const unsigned char* xs_bytes = reinterpret_cast<const unsigned char*>(xs_p);
  const size_t xstep = (size_t)xfrag;
  const int xmine = dma_wave ? (xpieces - wave + DW - 1) / DW : 0;  // how many of the pieces w, w + DW, ... exist
  auto issue_x = [&](int q) {
    if (!dma_wave) return;
    unsigned char* xd = xsl + (q % 3) * xstride;
    if (xmine > 0) {
      const unsigned char* src = uniform(xs_bytes + (size_t)q * xstep + wave * XPB + 4096);
      unsigned char* centre = xd + wave * XPB + 4096;
      static_for<0, (C16 ? 8 : 32) / DW>([&](auto jj_) {
        constexpr int jj = decltype(jj_)::value;
        if constexpr (C16) {
          if (xmine > jj) glds16_off<jj * DW * 1024 - 4096>(src + lane16, centre);
        } else {
          if (xmine > jj) glds4_off<jj * DW * 256 - 4096>(src + lane4, centre);
        }
      });
    }
    unsigned char* nd = xd + xfrag;
    if (wave == DW - 2 && C16 == 0) {
      const int nlane = (sizeof(TG) == 8) ? lane4 : (lane & 31) * 4;
      glds4_off<0>(uniform(reinterpret_cast<const unsigned char*>(xnorm + 32 * q)) + nlane, nd);
    } else if (wave == DW - 1) {
      glds4_off<0>(uniform(reinterpret_cast<const unsigned char*>(alpha + 32 * q)) + (lane & 31) * 4, nd + 64 * sizeof(TG));
    }
  };

  issue_panel(0, 0);
  issue_x(0);
  if (1 < q_end) issue_x(1);
  TG cm = TG(-2) * SC;
  float nb_c16[CT] = {0, 0};
  if constexpr (C16) {
    // the leaves' side of the fp16 contraction: B operand of the 16x16x32 instruction, lane l element j = leaf
    // col0 + 16 t + (l & 15), dimension 32 cc + 8 (l >> 4) + j, scaled like the training side and split the same way
    const float up = c16_scale[1];
    cm *= (TG)c16_scale[2];
    constexpr int nc = C16;
    u32x4* xb16 = reinterpret_cast<u32x4*>(xb);
    for (int t = 0; t < CT; ++t) {
      float nrm2 = 0.0f;  // |x* / l|^2 2^2sx of the pieces (see pack_xs_f16_kernel: norms and products of the same numbers)
      for (int cc = 0; cc < nc; ++cc) {
        const TG* src = leaves_s + (col0 + t * 16 + (lane & 15)) * dp;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 32 * cc + 8 * (lane >> 4) + j;
          v[j] = k < dp ? fminf(fmaxf((float)src[k] * up, -60000.0f), 60000.0f) : (k == dp ? 128.0f : 0.0f);
        }
        // (the norm from scalar conversions of the values, ahead of the split: summed from bit casts of the packed pieces
        // hipcc 7.2 added element (0, e) of the first pair for every pair -- disassembly; wrong norms, caught by the tests)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float h0 = (float)(_Float16)v[j];
          const float vp = h0 + (float)(_Float16)(v[j] - h0);
          if (32 * cc + 8 * (lane >> 4) + j < dp) nrm2 += vp * vp;
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          u32x4 f;
#pragma unroll
          for (int h = 0; h < 4; ++h) f[h] = f16_split_pair(v[2 * h], v[2 * h + 1]);
          xb16[((t * nc + cc) * 2 + pc) * 64 + lane] = f;
        }
      }
      nrm2 += __shfl_xor(nrm2, 16);  // the four lane groups hold a row's slots 8 g .. 8 g + 7 of every chunk
      nrm2 += __shfl_xor(nrm2, 32);
      nb_c16[t] = nrm2 * c16_scale[2];
    }
  } else {
    for (int t = 0; t < CT; ++t)
      for (int c = 0; c < dp4; ++c)
        xb[(t * dp4 + c) * 64 + lane] =
            leaves_s[(col0 + t * 16 + (lane & 15)) * dp + 4 * c + (lane >> 4)];
  }
  TG nb[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    if constexpr (C16 != 0) nb[t] = (TG)nb_c16[t] * SC;
    else nb[t] = lnorm[col0 + t * 16 + (lane & 15)] * SC;
  }
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[rt][t] = f32x4{0, 0, 0, 0};
  float macc[CT] = {0, 0};
  bf16x8 bfrag[NS][CT];
  float vc[3];
  gen_poly_coeffs<KERNEL>(variance, vc);
  __syncthreads();

  auto issue_for = [&](int k) {
    if (k + 1 < q_end) issue_panel(k + 1, (k + 1) & 1);
    if (k + 2 < q_end) issue_x(k + 2);
  };
  if constexpr (FUSED) {
    // Interval q (between workgroup barriers q - 1 and q), every wave alike: the DMAs of the L^-1 pieces of step q + 1
    // (into the buffer step q - 1 was applied from) and of the inputs of step q + 2 (ring of three) are issued, then
    // the fused step applies step q and generates step q + 1.  Step 0 is generated on its own.
    bf16x8 bnxt[NS][CT];
    if (q_diag0 == 0) leaf_bf16_gen<NS, TG, KERNEL, F16, true, C16>(lane, dp4, xsl, xb, nb, cm, vc, bfrag, macc);
    else leaf_bf16_gen<NS, TG, KERNEL, F16, false, C16>(lane, dp4, xsl, xb, nb, cm, vc, bfrag, macc);
    // Measured and NOT kept (tools/ab_time.py, same box, f16x3 at C3: two-phase 0.850 | this 0.7955 ms): the step's DMAs
    // dealt behind the MFMAs of row tiles 1, 2, ... like the map (+2 %); a ring of THREE buffers of L^-1 pieces with the
    // DMAs of step q + 2 issued at the tail of step q behind a raw s_barrier (0.7973); the SIMD's issue priority handed
    // from waves 0-3 to waves 4-7 in the middle of every step (s_setprio; 0.8042 -- the arbiter serves the older wave
    // first: stamps show waves 0-3 through a step in 3 700 clocks and waiting 1 800 at the barrier for waves 4-7, which
    // need 5 000; flipping the priority flips who waits, the sum grows); with the fp16 contraction (a third fewer vector
    // instructions per step) the DMA duties dealt over all eight waves again (0.6976 against 0.6897 ms); a fourth product
    // h1 h1' in the fp16 contraction (self-test readings 20 % lower, kernel +2.5 %).
#define GPSO_FUSED_STEP(ASKIP, GMODE)                                                                                 \
  {                                                                                                                   \
    GPSO_BSTAMP(q, 0);                                                                                                \
    issue_for(q);                                                                                                     \
    GPSO_BSTAMP(q, 1);                                                                                                \
    leaf_bf16_fused_step<NS, TG, KERNEL, F16, ASKIP, GMODE, C16>(q, q_diag0, lane, dp4,                               \
                                                                 panel + (q & 1) * NS * RT * 64,                      \
                                                                 xsl + ((q + 1) % 3) * xstride, xb, nb, cm, vc,       \
                                                                 bfrag, bnxt, acc, macc);                             \
    GPSO_BSTAMP(q, 4);                                                                                                \
    __syncthreads();                                                                                                  \
    GPSO_BSTAMP(q, 5);                                                                                                \
    if (GMODE != 0) {                                                                                                 \
      for (int sp = 0; sp < NS; ++sp)                                                                                 \
        for (int t = 0; t < CT; ++t) bfrag[sp][t] = bnxt[sp][t];                                                      \
    }                                                                                                                 \
  }
    int q = 0;
    for (; q + 1 < q_diag0; ++q) GPSO_FUSED_STEP(0, 1)
    if (q < q_diag0) {  // the last step below the diagonal block generates the block's first step: with its k*.alpha
      GPSO_FUSED_STEP(0, 2)
      ++q;
    }
    static_for<0, RT / 2>([&](auto j_) {  // the diagonal block: step j skips its 2 j all-zero row tiles
      constexpr int j = decltype(j_)::value;
      if constexpr (j + 1 < RT / 2) GPSO_FUSED_STEP(2 * j, 2)
      else GPSO_FUSED_STEP(2 * j, 0)
      ++q;
    });
#undef GPSO_FUSED_STEP
  } else {
  // Interval k (between workgroup barriers k - 1 and k): waves 0-3 generate and apply step k; waves 4-7 apply step
  // k and generate step k + 1.  One copy of the code: every wave runs gen(q), apply(q) for q = 0, 1, ...; only the
  // place of the barrier differs -- after apply(q) for waves 0-3, after gen(q) (q >= 1) for waves 4-7, which
  // therefore meet one last barrier after the loop.  DMA issued in interval k (L^-1 pieces of step k + 1 into the
  // buffer step k - 1 was applied from; inputs of step k + 2 into the ring of three) has landed at barrier k.
  const bool ahead = wave >= NW / 2;
  // the k-steps below the diagonal block run a branch-free copy of the step (no per-row-tile skip, no mean); the
  // RT / 2 steps of the diagonal block the general one
#define GPSO_BF16_STEP(DIAGF)                                                                                         \
  {                                                                                                                   \
    GPSO_BSTAMP(q, 0);                                                                                                \
    if (!ahead || q == 0) issue_for(q);                                                                               \
    else if (q >= 2) issue_for(q - 1); /* (this wave's iteration q starts in interval q - 1) */                       \
    GPSO_BSTAMP(q, 1);                                                                                                \
    leaf_bf16_gen<NS, TG, KERNEL, F16, DIAGF, C16>(lane, dp4, xsl + (q % 3) * xstride, xb, nb, cm, vc, bfrag, macc);  \
    GPSO_BSTAMP(q, 2);                                                                                                \
    if (ahead && q > 0) __syncthreads();                                                                              \
    GPSO_BSTAMP(q, 3);                                                                                                \
    leaf_bf16_apply<NS, F16, DIAGF>(q, q_diag0, lane, panel + (q & 1) * NS * RT * 64, bfrag, acc);                    \
    GPSO_BSTAMP(q, 4);                                                                                                \
    if (!ahead) __syncthreads();                                                                                      \
    GPSO_BSTAMP(q, 5);                                                                                                \
  }
  for (int q = 0; q < q_diag0; ++q) GPSO_BF16_STEP(false)
  for (int q = q_diag0; q < q_end; ++q) GPSO_BF16_STEP(true)
#undef GPSO_BF16_STEP
  if (ahead) __syncthreads();

  }

  // (fp16 split: undo the power-of-two scales of the two operands -- exact)
  double unscale2 = 1.0, unscale_m = 1.0;
  if constexpr (F16) {
    const double ia = (double)inv_scale_a[1], ib = (double)inv_scale_b;  // ([0] is max |L^-1|, [1] = 2^-sa)
    unscale2 = (ia * ib) * (ia * ib);
    unscale_m = ib;
  }
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
    if constexpr (F16) {
      sq *= unscale2;
      mm *= unscale_m;
    }
    if (lane < 16) {
      const int64_t col = col0 + t * 16 + lane;
      part_var[(int64_t)bi * mpad + col] = sq;
      part_mean[(int64_t)bi * mpad + col] = mm;
    }
  }
}

template <int NS, typename TG, bool F16 = false, int C16 = 0>
static int launch_leaf_tiles_bf16_ns(hipStream_t st, const void* linv_b, const TG* xs_p,
                                     const TG* xnorm, const float* alpha, const TG* leaves_s,
                                     const TG* lnorm, double* part_var, double* part_mean,
                                     int64_t npad, int dp4, int64_t mpad, const KernParams& kp,
                                     const int64_t* m_live, const float* inv_scale_a = nullptr, int variant = 0,
                                     const float* c16_scale = nullptr) {
  const int nbi = (int)(npad / 256);
  const dim3 grid((unsigned)(mpad / 256), (unsigned)nbi);
  const size_t lds = leaf_bf16_lds_bytes(NS, dp4, (int)sizeof(TG), C16 != 0);
  if (C16 ? leaf_c16_chunks(dp4) != C16 : 2 * dp4 * (int)(sizeof(TG) / 4) > 32) {  // the X fragments of a k-step are one DMA window of 32 pieces (C16: 8)
    note_launch_error("launch_leaf_tiles_bf16: more than 32 X pieces per k-step");
    return 1;
  }
  // fp16 split: the generated tile is scaled by 2^sb, sigma^2 2^sb in [2^13, 2^14) (folded into the variance)
  int eb = 0;
  (void)frexp(kp.variance, &eb);
  const float var_arg = F16 ? (float)ldexp(kp.variance, 14 - eb) : (float)kp.variance;
  const float inv_b = F16 ? (float)ldexp(1.0, eb - 14) : 1.0f;
  // variant 0 (GPSO_SPLIT_KERNEL_AUTO): the fused step; 1: round 3's two-phase step.  Same bits either way.
#define GPSO_L2(K, FUSED)                                                                           \
  do {                                                                                              \
    const int rc = ensure_dyn_lds((const void*)leaf_tiles_bf16_kernel<NS, TG, K, F16, FUSED, C16>, (int)lds); \
    if (rc) return rc;                                                                              \
    hipLaunchKernelGGL((leaf_tiles_bf16_kernel<NS, TG, K, F16, FUSED, C16>), grid, dim3(512), lds, st, \
                       static_cast<const u32x4*>(linv_b), xs_p, xnorm, alpha, leaves_s, lnorm,      \
                       part_var, part_mean, (int)(npad / 16), dp4, mpad, nbi, var_arg, m_live,      \
                       inv_scale_a, inv_b, c16_scale);                                              \
  } while (0)
#define GPSO_L(K)                                                                                   \
  do {                                                                                              \
    if (variant == 0) GPSO_L2(K, true);                                                             \
    else GPSO_L2(K, false);                                                                         \
  } while (0)
  switch (kp.kernel) {
    case 0: GPSO_L(0); break;
    case 1: GPSO_L(1); break;
    case 2: GPSO_L(2); break;
    default: GPSO_L(3); break;
  }
#undef GPSO_L2
#undef GPSO_L
  return 0;
}

template <typename TG>
int launch_leaf_tiles_bf16(hipStream_t st, int nsplit, const void* linv_b, const TG* xs_p,
                           const TG* xnorm, const float* alpha, const TG* leaves_s,
                           const TG* lnorm, double* part_var, double* part_mean, int64_t npad,
                           int dp4, int64_t mpad, const KernParams& kp, const int64_t* m_live,
                           const float* f16_inv_scale_a, int variant, const void* xs_h16, const float* c16_scale) {
  if (f16_inv_scale_a != nullptr) {  // fp16 split (nsplit == 2 pieces)
    if constexpr (sizeof(TG) == 4) {
      if (xs_h16 != nullptr && c16_scale != nullptr) {  // ... with the contraction on the fp16 pipe as well
        if (leaf_c16_chunks(dp4) == 1)
          return launch_leaf_tiles_bf16_ns<2, TG, true, 1>(st, linv_b, static_cast<const TG*>(xs_h16), xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, f16_inv_scale_a, variant, c16_scale);
        return launch_leaf_tiles_bf16_ns<2, TG, true, 2>(st, linv_b, static_cast<const TG*>(xs_h16), xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, f16_inv_scale_a, variant, c16_scale);
      }
    }
    return launch_leaf_tiles_bf16_ns<2, TG, true>(st, linv_b, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, f16_inv_scale_a, variant);
  }
  if (nsplit == 3)
    return launch_leaf_tiles_bf16_ns<3, TG>(st, linv_b, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, nullptr, variant);
  return launch_leaf_tiles_bf16_ns<2, TG>(st, linv_b, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, nullptr, variant);
}
template int launch_leaf_tiles_bf16<float>(hipStream_t, int, const void*, const float*, const float*, const float*, const float*, const float*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, int, const void*, const float*);
template int launch_leaf_tiles_bf16<double>(hipStream_t, int, const void*, const double*, const double*, const float*, const double*, const double*, double*, double*, int64_t, int, int64_t, const KernParams&, const int64_t*, const float*, int, const void*, const float*);

template <typename TF>
void launch_pack_linv_bf16(hipStream_t st, int nsplit, const TF* linv, int64_t n, int64_t npad,
                           void* linv_b) {
  const int64_t total = (npad / 16) * (npad / 32) * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (nsplit == 3)
    hipLaunchKernelGGL((pack_linv_bf16_kernel<3, TF>), grid, dim3(256), 0, st, linv, n, npad, static_cast<u32x4*>(linv_b));
  else
    hipLaunchKernelGGL((pack_linv_bf16_kernel<2, TF>), grid, dim3(256), 0, st, linv, n, npad, static_cast<u32x4*>(linv_b));
}
template <typename TF>
void launch_pack_linv_f16(hipStream_t st, const TF* linv, int64_t n, int64_t npad, float* scal, void* linv_b,
                          bool have_max) {
  if (!have_max) {
    (void)hipMemsetAsync(scal, 0, 4, st);
    hipLaunchKernelGGL((absmax_kernel<TF>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, linv, n, npad,
                       reinterpret_cast<unsigned*>(scal));
  }
  const int64_t total = (npad / 16) * (npad / 32) * 64;
  hipLaunchKernelGGL((pack_linv_f16_kernel<TF>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, linv, n, npad,
                     scal, static_cast<u32x4*>(linv_b));
}
template void launch_pack_linv_f16<float>(hipStream_t, const float*, int64_t, int64_t, float*, void*, bool);
template void launch_pack_linv_f16<double>(hipStream_t, const double*, int64_t, int64_t, float*, void*, bool);
template void launch_pack_linv_bf16<float>(hipStream_t, int, const float*, int64_t, int64_t, void*);
template void launch_pack_linv_bf16<double>(hipStream_t, int, const double*, int64_t, int64_t, void*);

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void leaf_finalize_kernel(const double* __restrict__ part_var,
                                                            const double* __restrict__ part_mean,
                                                            int nbi, int64_t mpad, int64_t m,
                                                            KernParams kp, double varsigma,
                                                            double* __restrict__ mean,
                                                            double* __restrict__ var,
                                                            double* __restrict__ ucb) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double v = 0, mu = 0;
  for (int b = 0; b < nbi; ++b) {
    v += part_var[(int64_t)b * mpad + j];
    mu += part_mean[(int64_t)b * mpad + j];
  }
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
                                                         straight into pinned host memory (no copy operation behind) */) {
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
                                                           double* __restrict__ host_vals /* nullable, see seg_argmax_stage2 */) {
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
                                      int world, int nseg, int stride, double* __restrict__ out) {
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
  }
  if (seg >= nseg) return;
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
                          double* mean, double* var, double* ucb) {
  const int64_t blocks = (m + 255) / 256;
  if (blocks == 0) return;
  hipLaunchKernelGGL(leaf_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part_var,
                     part_mean, nbi, mpad, m, kp, varsigma, mean, var, ucb);
}

void launch_seg_argmax(hipStream_t st, const double* mean, const double* var, const double* ucb,
                       const int64_t* seg_off_dev, int nseg, int nblk, void* partial_dev,
                       double* out_vals_dev, const LeafFinalize* fin, double* host_vals) {
  if (fin != nullptr)
    hipLaunchKernelGGL(seg_argmax_stage1<true>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb,
                       seg_off_dev, reinterpret_cast<Best*>(partial_dev), *fin);
  else
    hipLaunchKernelGGL(seg_argmax_stage1<false>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb,
                       seg_off_dev, reinterpret_cast<Best*>(partial_dev), LeafFinalize{});
  hipLaunchKernelGGL(seg_argmax_stage2, dim3((unsigned)nseg), dim3(256), 0, st,
                     reinterpret_cast<const Best*>(partial_dev), nblk, seg_off_dev, mean, var, ucb, nseg,
                     out_vals_dev, host_vals);
}

void launch_keyed_argmax(hipStream_t st, const double* mean, const double* var, const double* ucb,
                         const int64_t* key_dev, int64_t rows, int64_t uniq, int nseg, const int64_t* live_dev,
                         int nblk, void* partial_dev, int64_t* pos_dev, double* out_vals_dev, const LeafFinalize* fin,
                         double* host_vals) {
  if (fin != nullptr)
    hipLaunchKernelGGL(keyed_argmax_stage1<true>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb, key_dev,
                       rows, uniq, nseg, live_dev, reinterpret_cast<Best*>(partial_dev), pos_dev, *fin);
  else
    hipLaunchKernelGGL(keyed_argmax_stage1<false>, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, st, ucb, key_dev,
                       rows, uniq, nseg, live_dev, reinterpret_cast<Best*>(partial_dev), pos_dev, LeafFinalize{});
  hipLaunchKernelGGL(keyed_argmax_stage2, dim3((unsigned)nseg), dim3(256), 0, st,
                     reinterpret_cast<const Best*>(partial_dev), pos_dev, nblk, rows, mean, var, ucb, live_dev,
                     nseg, out_vals_dev, host_vals);
}

void launch_reduce_winners(hipStream_t st, const double* gathered, const int64_t* base, int world, int nseg,
                           int stride, double* out) {
  hipLaunchKernelGGL(reduce_winners_kernel, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, st, gathered, base,
                     world, nseg, stride, out);
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
