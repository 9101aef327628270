// The split predict kernels (fp16 / bf16 pieces of L^-1 on the 16-bit matrix cores): device code and launchers, included
// by ONE translation unit per generation type (predict_split_f32.hip, predict_split_f64.hip -- the 64 instantiations
// compile in parallel) and by the stamp harnesses under tools/micro.  Moved here from predict.hip in round 4, unchanged.
#pragma once
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace gpso {

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
// The workgroup barrier of the split kernels: every barrier there publishes LDS-DMA data, so the wave's own DMAs are waited for
// EXPLICITLY in front of it.  hipcc 7.2 inserts that wait (s_waitcnt vmcnt(0)) itself when the DMA and the barrier share a basic
// block or an iteration -- and does not when the DMAs were issued before a loop's back edge and the barrier stands at the loop's
// header (round 6: the next row block's first DMAs issued before the epilogue; the barrier at the top of the block then had
// lgkmcnt(0) only, and the variances of the bf16 kernels at C5 came out wrong now and then -- found by the full GPU suite, read
// in the assembly).  The explicit wait costs nothing where the compiler's is already there.
__device__ __forceinline__ void leaf_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---- fp16 split ("f16x3"): x = h0 + h1 with fp16 pieces (11 significant bits each, round to nearest: |x - h0 - h1| <=
// 2^-23 |x|), a product = h0 h0' + h0 h1' + h1 h0' on v_mfma_f32_16x16x32_f16 -- THREE matrix instructions instead of
// six, the dropped h1 h1' is 2^-22 relative.  fp16 has 5 exponent bits, so both operands are scaled by powers of two
// (exact): L^-1 by 2^sa with max |L^-1| 2^sa in [2^13, 2^14) (found on the device at packing time, absmax_kernel), the
// generated tile by 2^sb with sigma^2 2^sb in [2^13, 2^14) (folded into the variance the kernel map multiplies with).
// An entry 2^-28 below its operand's maximum is still a normal fp16 number; below that the ABSOLUTE error per entry
// stays under 2^-25 of the scaled maximum -- far inside the 2^-22 the dropped product costs.  Accumulation is f32 as
// everywhere; the epilogue undoes the scales (sum of squares x 2^-2(sa+sb), mean x 2^-sb).
// (f16x2 / f16x8 and f16_split_pair: common.hpp -- the fit's fp16 planes use them too)

// Round 6: the half-step stagger of waves 4-7 (leaf_bf16_fused_half below) is bit-identical to the shipped step and SLOWER
// on the same box (tools/ab_time.py, alternating fresh processes: C3 0.7193 | 0.7379 ms, C4 share 5.339 | 5.490 ms; profiles/
// r06_predict_experiments.txt): the partners are already out of step -- the SIMD's arbiter serves the older wave first, waves
// 0-3 are through a step in 3 700 clocks and wait for waves 4-7 -- so the delay adds to the slower half's path instead of
// filling the faster half's wait.  Built only with -DGPSO_LEAF_STAGGER=1.
#ifndef GPSO_LEAF_STAGGER
#define GPSO_LEAF_STAGGER 0
#endif
constexpr bool kLeafStagger = GPSO_LEAF_STAGGER != 0;
#ifndef GPSO_EARLY_DMA
#define GPSO_EARLY_DMA 1
#endif
constexpr bool kEarlyDma = GPSO_EARLY_DMA != 0;  // the next row block's first DMAs issued before the epilogue (leaf_tiles_bf16_kernel)
// The fused step on the 32x32x16 matrix instruction (leaf_split_m32.hpp; round 6): correct (errors against float64 equal to the
// 16x16x32 step's), 4.9 % FEWER clocks per launch (GRBM_GUI_ACTIVE 1.056e7 against 1.111e7 at C3) -- and an 11 % LOWER clock
// under it (1.80 against 2.02 GHz: MI355X_MICROARCH.md, DVFS give-back, item 7), so 6.6-7.1 % SLOWER at C3 / C4 / C5
// (profiles/r06_step32_*.jsonl, r06_pmc_16_vs_32.txt).  Built only with -DGPSO_STEP32=1 (then GPSO_SPLIT_KERNEL_FUSED32 selects it).
#ifndef GPSO_STEP32
#define GPSO_STEP32 0
#endif
constexpr bool kLeafStep32 = GPSO_STEP32 != 0;
// GPSO_OPT_ROW_LOOP (process-wide switch of the split kernels' workgroup shape; api.hip sets it): predict.hip owns both
extern int g_leaf_row_loop;
extern int g_leaf_last_splits;
int leaf_cu_count();
#ifndef GPSO_BSTAMP
#define GPSO_BSTAMP(q, i)  // tools/micro/leaf_bf16_phases.hip defines this to record s_memtime stamps
#endif

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
// XP = 1 (round 6): the LAST chunk uses at most 16 of its 32 slots (D_pad + 1 - 32 (chunks - 1) <= 16: C3's D = 12, the second
// chunk of C5's D = 40) -- its two small products share ONE instruction: A = [h0 (slots 0-15) | h1 (slots 0-15)] against
// B = [h1' | h0'], then h0 h0' as before (the pieces' slots 16-31 are zeros): 8 contraction MFMAs per chunk instead of 12.  The
// operands are the same 16-byte words of the same fragments, fetched by other lanes.
template <typename TG, int C16, int CT, int XP = 0>
__device__ __forceinline__ void leaf_contract(int lane, int dp4, const unsigned char* xs_b, const TG* xb,
                                              typename Mfma<TG>::vec4 (&s)[2][CT]) {
  if constexpr (C16 != 0) {
    static_assert(sizeof(TG) == 4, "the fp16 contraction belongs to float generation");
    constexpr int nc = C16;
    const u32x4* xa = reinterpret_cast<const u32x4*>(xs_b);  // [h][cc][piece][64]
    const u32x4* lb = reinterpret_cast<const u32x4*>(xb);    // [t][cc][piece][64]
#pragma unroll
    for (int cc = 0; cc < nc; ++cc) {
      if constexpr (XP == 1) {
        if (cc == nc - 1) {  // (compile-time after unrolling)
          const int g1 = lane >> 5;                      // lanes 32-63 take the words of the OTHER piece
          const int w = (lane & 15) + 16 * ((lane >> 4) & 1);  // ... at slots 0-15: lane groups 0, 1 of the fragment
          u32x4 a1[2], a2[2], b1[CT], b2[CT];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            a1[h] = xa[((h * nc + cc) * 2 + g1) * 64 + w];
            a2[h] = xa[((h * nc + cc) * 2 + 0) * 64 + lane];
          }
#pragma unroll
          for (int t = 0; t < CT; ++t) {
            b1[t] = lb[((t * nc + cc) * 2 + 1 - g1) * 64 + w];
            b2[t] = lb[((t * nc + cc) * 2 + 0) * 64 + lane];
          }
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < CT; ++t)
              s[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[h]), __builtin_bit_cast(f16x8, b1[t]), s[h][t], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < CT; ++t)
              s[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2[h]), __builtin_bit_cast(f16x8, b2[t]), s[h][t], 0, 0, 0);
          continue;
        }
      }
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
template <int NS, typename TG, int KERNEL, bool F16, bool DIAG, int C16 = 0, int CT = 2, int XP = 0>
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
  leaf_contract<TG, C16, CT, XP>(lane, dp4, xs_b, xb, s);
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
// RTL: row tiles of the block that hold training rows (16, or 8 for a LAST row block with <= 128 of them: the tiles
// beyond are all zero and are not applied; the map of the next step is then dealt over row tiles 1 .. RTL - 1)
template <int NS, typename TG, int KERNEL, bool F16, int ASKIP, int GMODE, int C16 = 0, int RTL = 16, int XP = 0>
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
    leaf_contract<TG, C16, CT, XP>(lane, dp4, xs_n, xb, s);
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
  static_assert(ASKIP >= 0 && ASKIP < RTL && RTL <= RT, "at least one live row tile");
  // the L^-1 fragments of a row tile are read AD tiles ahead of their MFMAs (GPSO_LEAF_AHEAD, default 1: two register sets; 2:
  // three sets, a read has two tiles' MFMAs = 192 clocks to land instead of 96)
#ifndef GPSO_LEAF_AHEAD
#define GPSO_LEAF_AHEAD 1
#endif
  constexpr int AD = (NS == 2) ? GPSO_LEAF_AHEAD : 1, AB = AD + 1;
  u32x4 a[AB][NS];
  static_for<0, AD>([&](auto d_) {
    constexpr int rt0 = ASKIP + decltype(d_)::value;
    if constexpr (rt0 < RTL) {
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) a[rt0 % AB][sp] = panel_b[(sp * RT + rt0) * 64 + lane];
    }
  });
  static_for<0, RTL>([&](auto rt_) {
    constexpr int rt = decltype(rt_)::value;
    if constexpr (rt + AD < RTL && rt + AD >= ASKIP + AD && rt >= ASKIP) {
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) a[(rt + AD) % AB][sp] = panel_b[(sp * RT + rt + AD) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (rt >= ASKIP) {  // (diagonal block: all-zero tiles above the diagonal)
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        f32x4 c = acc[rt][t];
#define GPSO_BF(SA, SB)                                                                                                   \
  c = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[rt % AB][SA]), __builtin_bit_cast(f16x8, bcur[SB][t]), c, 0, 0, 0) \
          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[rt % AB][SA]), bcur[SB][t], c, 0, 0, 0)
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
    if constexpr (GEN && rt >= 1) {  // this row tile's share of the map: ops [(rt - 1) NOPS / (RTL - 1), rt NOPS / (RTL - 1))
      static_for<(rt - 1) * NOPS / (RTL - 1), rt * NOPS / (RTL - 1)>(op);
    }
  });
  if constexpr (GEN) {
#pragma unroll
    for (int sp = 0; sp < NS; ++sp)
#pragma unroll
      for (int t = 0; t < CT; ++t) bnxt[sp][t] = __builtin_bit_cast(bf16x8, fr[sp][t]);
  }
}

// ---- the fused step in two HALVES (round 6: the stagger) ------------------------------------------------------------------
// All eight waves run the same fused stream with one barrier per step, two waves per SIMD: partners reach their MFMA
// bursts, their LDS fragment reads and the barrier together (matrix pipe busy 0.62, the rest is waiting on each other:
// MI355X_MICROARCH.md, "two waves per SIMD", item 9).  The stagger delays waves 4-7 by HALF a step: the step is cut between
// row tiles RTL / 2 - 1 and RTL / 2, waves 0-3 meet the barrier at the end of a step, waves 4-7 in its middle -- one copy of
// the code, only the place of the barrier differs.  Waves 4-7 then read the L^-1 pieces of step q - 1 (second half) and of
// step q (first half) in the interval in which waves 0-3 issue the DMAs of step q + 1: the pieces live in a ring of THREE
// buffers.  Same operations on the same operands in the same order per wave: the same bits.
// The state a step carries from its first half to its second (registers):
template <int NS, typename TG, int CT = 2>
struct FusedRegs {
  typename Mfma<TG>::vec4 s[2][CT];
  typename Mfma<TG>::vec4 nav[2];
  float p[CT][8], ex[CT][8];
  TG na[2][4];
  f32x4 al4[2];
  u32x4 fr[NS][CT];
  u32x4 a[2][NS];
};
template <int NS, typename TG, int KERNEL, bool F16, int ASKIP, int GMODE, int C16, int RTL, int HALF>
__device__ __forceinline__ void leaf_bf16_fused_half(int lane, int dp4, const u32x4* panel_b, const unsigned char* xs_n,
                                                     const TG* xb, const TG (&nb)[2], const TG cm, const float (&vc)[3],
                                                     const bf16x8 (&bcur)[NS][2], bf16x8 (&bnxt)[NS][2], f32x4 (&acc)[16][2],
                                                     float (&macc)[2], FusedRegs<NS, TG>& R) {
  using MG = Mfma<TG>;
  using vecG = typename MG::vec4;
  constexpr int RT = 16, CT = 2;
  constexpr TG SC = (TG)GenScale<KERNEL>::SC;
  constexpr bool GEN = GMODE != 0;
  const int XF = Bf16Lds<TG, C16>::xfrag(dp4);
  constexpr int E = 16;
  constexpr int O_COMB = C16 != 0 ? 0 : 8, O_SQRT = O_COMB + E, O_EXP = O_SQRT + (KERNEL == 3 ? 0 : E), O_POLY = O_EXP + E,
                O_MEAN = O_POLY + E, O_SPLIT = O_MEAN + (GMODE == 2 ? E : 0), NOPS = O_SPLIT + 8;
  static_assert(ASKIP >= 0 && ASKIP < RTL && RTL <= RT && RTL % 2 == 0, "at least one live row tile");
  if constexpr (HALF == 0) {
    if constexpr (GEN) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < CT; ++t) R.s[h][t] = vecG{0, 0, 0, 0};
      if constexpr (C16 == 0) {
        const TG* nrm = reinterpret_cast<const TG*>(xs_n + XF);
#pragma unroll
        for (int h = 0; h < 2; ++h) R.nav[h] = *reinterpret_cast<const vecG*>(nrm + 16 * h + 4 * (lane >> 4));
      }
      leaf_contract<TG, C16, CT>(lane, dp4, xs_n, xb, R.s);
    }
    if constexpr (GMODE == 2) {
      const float* alp = reinterpret_cast<const float*>(xs_n + XF + 64 * sizeof(TG));
#pragma unroll
      for (int h = 0; h < 2; ++h) R.al4[h] = *reinterpret_cast<const f32x4*>(alp + 16 * h + 4 * (lane >> 4));
    }
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) R.a[ASKIP & 1][sp] = panel_b[(sp * RT + ASKIP) * 64 + lane];
  }
  auto op = [&](auto o_) {
    constexpr int o = decltype(o_)::value;
    if constexpr (o < O_COMB) {
      R.na[o >> 2][o & 3] = R.nav[o >> 2][o & 3] * SC;
    } else if constexpr (o < O_SQRT) {
      constexpr int e = o - O_COMB, t = e >> 3, h = (e >> 2) & 1, r = e & 3;
      if constexpr (C16 != 0) R.p[t][4 * h + r] = (float)fma_t(cm, R.s[h][t][r], nb[t]);
      else R.p[t][4 * h + r] = (float)fma_t(cm, R.s[h][t][r], R.na[h][r] + nb[t]);
    } else if constexpr (o < O_EXP) {
      constexpr int e = o - O_SQRT, t = e >> 3, j = e & 7;
      if constexpr (sizeof(TG) == 4) R.p[t][j] = __builtin_amdgcn_sqrtf(__builtin_fabsf(R.p[t][j]));
      else R.p[t][j] = __builtin_amdgcn_sqrtf(fmaxf(R.p[t][j], (float)(GenScale<KERNEL>::SC * 1e-36)));
    } else if constexpr (o < O_POLY) {
      constexpr int e = o - O_EXP, t = e >> 3, j = e & 7;
      R.ex[t][j] = __builtin_amdgcn_exp2f(-R.p[t][j]);
    } else if constexpr (o < O_MEAN) {
      constexpr int e = o - O_POLY, t = e >> 3, j = e & 7;
      if constexpr (KERNEL == 0) R.p[t][j] = fmaf(R.p[t][j], fmaf(R.p[t][j], vc[2], vc[1]), vc[0]) * R.ex[t][j];
      else if constexpr (KERNEL == 1) R.p[t][j] = fmaf(R.p[t][j], vc[1], vc[0]) * R.ex[t][j];
      else R.p[t][j] = vc[0] * R.ex[t][j];
    } else if constexpr (o < O_SPLIT) {  // (GMODE 2) k*.alpha in f32, before the split; per column tile in the order j = 0 .. 7
      constexpr int e = o - O_MEAN, t = e >> 3, j = e & 7;
      macc[t] = fma_t(R.p[t][j], R.al4[j >> 2][j & 3], macc[t]);
      asm volatile("" : "+v"(macc[t]));  // the two tiles' means stay in separate registers (see leaf_bf16_gen)
    } else {
      constexpr int e = o - O_SPLIT, t = e >> 2, j = e & 3;
      if constexpr (F16) {
        unsigned hh, ll;
        f16_split_pair_both(R.p[t][2 * j], R.p[t][2 * j + 1], hh, ll);
        R.fr[0][t][j] = hh;
        R.fr[1][t][j] = ll;
      } else {
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) R.fr[sp][t][j] = bf16_split_pair(R.p[t][2 * j], R.p[t][2 * j + 1]);
      }
    }
  };
  static_for<HALF * (RTL / 2), (HALF + 1) * (RTL / 2)>([&](auto rt_) {
    constexpr int rt = decltype(rt_)::value;
    if constexpr (rt + 1 < RTL && rt + 1 > ASKIP) {
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) R.a[(rt + 1) & 1][sp] = panel_b[(sp * RT + rt + 1) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (rt >= ASKIP) {  // (diagonal block: all-zero tiles above the diagonal)
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        f32x4 c = acc[rt][t];
#define GPSO_BF(SA, SB)                                                                                                   \
  c = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, R.a[rt & 1][SA]), __builtin_bit_cast(f16x8, bcur[SB][t]), c, 0, 0, 0) \
          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, R.a[rt & 1][SA]), bcur[SB][t], c, 0, 0, 0)
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
    if constexpr (GEN && rt >= 1) {  // this row tile's share of the map: ops [(rt - 1) NOPS / (RTL - 1), rt NOPS / (RTL - 1))
      static_for<(rt - 1) * NOPS / (RTL - 1), rt * NOPS / (RTL - 1)>(op);
    }
  });
  if constexpr (GEN && HALF == 1) {
#pragma unroll
    for (int sp = 0; sp < NS; ++sp)
#pragma unroll
      for (int t = 0; t < CT; ++t) bnxt[sp][t] = __builtin_bit_cast(bf16x8, R.fr[sp][t]);
  }
}

#include "leaf_split_m32.hpp"

// F16: the fp16 split (two pieces, three products); `variance` then arrives multiplied by 2^sb, inv_scale_a[1] is
// 2^-sa (device, written by pack_linv_f16_kernel) and inv_scale_b = 2^-sb
// FUSED: every wave runs the fused step (apply of step q with the generation of step q + 1 in its MFMA shadows, one
// barrier per step); otherwise round 3's two-phase step with the waves of a SIMD in opposite order
// C16: the contraction on the fp16 pipe -- xs_p then points at the fp16 piece pairs of the scaled inputs
// (pack_xs_f16_kernel's order) and c16_scale at their scale (device: [1] = 2^sx, [2] = 2^-2sx)
// M32P: the fused step on the 32x32x16 matrix instruction (leaf_split_m32.hpp; fp16 split with the fp16 contraction only)
// XP: the last chunk of the fp16 contraction packs its two small products into one instruction (leaf_contract)
template <int NS, typename TG, int KERNEL, bool F16 = false, bool FUSED = false, int C16 = 0, bool M32P = false, int XP = 0>
__global__ __launch_bounds__(512, 2) void leaf_tiles_bf16_kernel(
    const u32x4* __restrict__ linv_b, const TG* __restrict__ xs_p, const TG* __restrict__ xnorm,
    const float* __restrict__ alpha, const TG* __restrict__ leaves_s,
    const TG* __restrict__ lnorm, double* __restrict__ part_var, double* __restrict__ part_mean,
    int npad16, int dp4, int64_t mpad, int nbi, float variance, const int64_t* __restrict__ m_live,
    const float* __restrict__ inv_scale_a, float inv_scale_b, const float* __restrict__ c16_scale, int q_max,
    const float* __restrict__ raw /* nullable (C16 only): the caller's UNSCALED float leaves [raw_m][raw_d] -- the prologue
    scales them itself ((float)(x / l), rows beyond raw_m are padding): no prep launch in front of this kernel */,
    const double* __restrict__ raw_ls, int64_t raw_m, int raw_d) {
  constexpr int RT = 16, CT = 2, NW = 8;
  constexpr TG SC = (TG)GenScale<KERNEL>::SC;
  constexpr bool M32 = M32P && FUSED && F16 && NS == 2 && C16 != 0 && sizeof(TG) == 4;
  static_assert(M32 == M32P, "the 32x32x16 step exists for the fp16 split with the fp16 contraction");
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // ---- which (leaf tile, row blocks) this workgroup computes ---------------------------------------------------------------
  // Rounds 1-5: one workgroup = (leaf tile blockIdx.x, ONE row block, heaviest first over blockIdx.y): 2 048 workgroups at C3,
  // eight per CU one after the other, each paying the leaf prologue (the tile's fragments from global memory, split into fp16
  // pieces), a cold first DMA and a dispatch.  Round 6: a workgroup keeps its leaf tile and LOOPS over row blocks -- split
  // blockIdx.y of gridDim.y takes the row blocks of rank j S + (j even ? split : S - 1 - split) in heaviest-first order (a
  // zig-zag: the splits weigh the same within one row block) -- so the prologue is paid once per leaf tile.  gridDim.y = nbi is
  // rounds 1-5's kernel; the launcher picks the smallest S that still fills the chip evenly (C3: S = 1, 256 workgroups of 288
  // k-steps each).  Same tiles, same operations in the same order per (leaf tile, row block): the same bits.
  // (Also measured this round and not kept -- row blocks paired heaviest + lightest and dealt to the XCDs in runs so that a
  // block's planes are fetched by one or two L2s instead of all eight: a third of the HBM traffic, same bits, 54 % slower,
  // because workgroups are dispatched in order ACROSS the XCDs; commit b85fc33, profiles/r06_predict_experiments.txt.)
  const int S = (int)gridDim.y, split = (int)blockIdx.y;
  const int64_t leaf_tile = blockIdx.x;
  if (m_live != nullptr && leaf_tile * (NW * CT * 16) >= *m_live) return;  // workgroup-uniform
  // STAG (round 6): waves 4-7 run half a step behind waves 0-3 (leaf_bf16_fused_half); the L^-1 pieces then live in a ring of
  // three buffers -- where that fits the 160 KB (one chunk of the fp16 contraction: D <= 28)
  constexpr bool STAG = kLeafStagger && FUSED && NS == 2 && F16 && C16 == 1;
  constexpr int PBUF = STAG ? 3 : 2;
  u32x4* panel = reinterpret_cast<u32x4*>(lds_raw);                 // [PBUF][NS][RT][64]
  unsigned char* xsl = reinterpret_cast<unsigned char*>(panel + PBUF * NS * RT * 64);  // [3] X buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xstride = Bf16Lds<TG, C16>::xbytes(dp4);
  const int xfrag = Bf16Lds<TG, C16>::xfrag(dp4);
  // this wave's leaf fragments: [CT][dp4][64] TG, or (C16) [CT][chunk][piece][64] x 16 bytes
  TG* xb = reinterpret_cast<TG*>(xsl + 3 * xstride + (size_t)wave * xfrag);

  const int64_t col0 = (leaf_tile * NW + wave) * (CT * 16);
  const int dp = dp4 * 4;
  const int npad32 = npad16 / 2;
  // the row block in hand (set_row_block): bi, its diagonal k-steps [q_diag0, q_end), and q_lim --
  // q_max = ceil(N / 32): the k-steps from there on hold padding points only -- all-zero columns of L^-1.  The fused
  // loop stops applying there (the last row block of an N that is not a multiple of 256; adding exact zeros or not adding
  // them: the same partial sums).  q_diag0 < q_max always: a row block has at least one training row.
  int bi = 0, q_diag0 = 0, q_end = 0, q_lim = 0;

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
  // The fragments of a DMA wave: f = FPW (wave % DW) + j, piece f / RT, row tile f % RT.  Where FPW divides RT (two pieces: 8 or 4
  // fragments per wave) they are consecutive row tiles of ONE piece -- equidistant in memory (a row tile's pieces are N_pad / 32
  // KB apart): one 64-bit base per wave and a 32-bit stride instead of FPW 64-bit bases (up to 16 of the ~100 scalar registers
  // the kernel has).  Three pieces (FPW = 6) straddle pieces: those kernels keep the array.
  constexpr bool kStrided = RT % FPW == 0;
  const unsigned char* pgb[kStrided ? 1 : FPW];
  const unsigned pg_step = (unsigned)npad32 * 1024u - 1024u;  // (from fragment j to j + 1, net of the immediate offsets' 1 KB)
  bool has_next = false;
  auto rank_of = [&](int j) { return j * S + ((j & 1) ? S - 1 - split : split); };
  auto set_row_block = [&](int j_mine /* this workgroup's j-th row block */) {
    const int rank = rank_of(j_mine);  // (ranks in heaviest-first order)
    has_next = rank_of(j_mine + 1) < nbi;
    bi = nbi - 1 - rank;
    q_diag0 = bi * (RT / 2);
    q_end = q_diag0 + RT / 2;
    q_lim = FUSED ? min(q_end, q_max) : q_end;
#pragma unroll
    for (int j = 0; j < (kStrided ? 1 : FPW); ++j) {
      const int f = FPW * (wave % DW) + j, sp = f / RT, rt = f % RT;  // fragment f = piece sp, row tile rt
      pgb[j] = reinterpret_cast<const unsigned char*>(linv_b + ((size_t)sp * npad16 + (bi * RT + rt)) * npad32 * 64) -
               (j * 1024 - FPW * 512);
    }
  };
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
    if constexpr (kStrided) {
      const unsigned char* pq = pgb[0] + (size_t)q * 1024;
      static_for<0, FPW>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        glds16_off<j * 1024 - FPW * 512>(uniform(pq + (size_t)j * pg_step) + lane16, centre);
      });
    } else {
      static_for<0, FPW>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        glds16_off<j * 1024 - FPW * 512>(uniform(pgb[j] + (size_t)q * 1024) + lane16, centre);
      });
    }
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

  // the first DMAs of a row block: its L^-1 pieces of step 0 into buffer 0 and the inputs of steps 0 and 1 (every row block has
  // at least eight steps)
  auto issue_block_start = [&]() {
    issue_panel(0, 0);
    issue_x(0);
    issue_x(1);
  };
  set_row_block(0);
  issue_block_start();
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
          // (clamped into fp16's range; a NaN coordinate stays NaN -- fminf / fmaxf would turn it into -60000 and the
          // leaf into a far-away point with the prior's mean and variance, unlike every other predict path)
          float xv;
          if (raw != nullptr) {  // (workgroup-uniform) the arithmetic of prep_leaves_kernel<float, float>, bit for bit
            const int64_t jrow = col0 + t * 16 + (lane & 15);
            const int kk = k < dp ? k : 0;
            xv = (jrow < raw_m && kk < raw_d) ? (float)((double)raw[jrow * raw_d + kk] / raw_ls[kk]) : 0.0f;
            xv *= up;
          } else {
            xv = (float)src[k < dp ? k : 0] * up;
          }
          v[j] = k < dp ? (xv != xv ? xv : fminf(fmaxf(xv, -60000.0f), 60000.0f)) : (k == dp ? 128.0f : 0.0f);
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
  float macc[CT];
  bf16x8 bfrag[NS][CT];
  float vc[3];
  gen_poly_coeffs<KERNEL>(variance, vc);
  // (M32) the same tile as eight 32 x 32 accumulators; this lane's leaf is col0 + (lane & 31)
  f32x16 acc32[RT / 2];
  float macc32 = 0.0f;
  u32x4 bfrag32[2][2];
  const M32Lanes ml = m32_lanes<C16 != 0 ? C16 : 1>(lane);
  const float nb32 = M32 ? (float)(((lane >> 4) & 1) ? nb[1] : nb[0]) : 0.0f;
  (void)acc32; (void)macc32; (void)bfrag32; (void)ml; (void)nb32;
  // ---- the row blocks of this workgroup, heaviest first -------------------------------------------------------------------
  // (every step ends with a workgroup barrier, the last one included: when the DMAs of the next row block's first steps are
  // issued, no wave still reads the buffers they land in)
  // (Measured and not kept: the next row block's first DMAs issued during the LAST step of the one in hand -- its buffers are
  // free there.  The second set of wave-uniform pointers spills 87 scalar registers into vector lanes inside the hot loop: C3
  // 0.7193 | 0.7221 ms against 0.7464 | 0.7304 without the prefetch, C4 / C5 +2.5 %.  r06_predict_experiments.txt.)
  for (int rbj = 0;; ++rbj) {
  if (!kEarlyDma && rbj > 0) {
    set_row_block(rbj);
    issue_block_start();
  }
  if constexpr (M32) {
#pragma unroll
    for (int R = 0; R < RT / 2; ++R)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc32[R][i] = 0.0f;
    macc32 = 0.0f;
  } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[rt][t] = f32x4{0, 0, 0, 0};
    macc[0] = macc[1] = 0.0f;
  }
  leaf_sync();

  auto issue_for = [&](int k) {
    if (k + 1 < q_lim) issue_panel(k + 1, STAG ? (k + 1) % 3 : (k + 1) & 1);
    if (k + 2 < q_end && k + 2 <= q_lim) issue_x(k + 2);  // (step q_lim is still GENERATED by step q_lim - 1: padding points, alpha = 0)
  };
  if constexpr (FUSED) {
    // Interval q (between workgroup barriers q - 1 and q), every wave alike: the DMAs of the L^-1 pieces of step q + 1
    // (into the buffer step q - 1 was applied from) and of the inputs of step q + 2 (ring of three) are issued, then
    // the fused step applies step q and generates step q + 1.  Step 0 is generated on its own.
    bf16x8 bnxt[NS][CT];
    FusedRegs<NS, TG> fregs;
    const bool late = wave >= NW / 2;  // (wave-uniform) the staggered half
    (void)fregs;
    (void)late;
    u32x4 bnxt32[2][2];
    (void)bnxt32;
    if constexpr (M32) {
      if (q_diag0 == 0) leaf_gen32<KERNEL, true, C16>(dp4, xsl, xb, ml, nb32, (float)cm, vc, bfrag32, macc32);
      else leaf_gen32<KERNEL, false, C16>(dp4, xsl, xb, ml, nb32, (float)cm, vc, bfrag32, macc32);
    } else {
      if (q_diag0 == 0) leaf_bf16_gen<NS, TG, KERNEL, F16, true, C16, 2, XP>(lane, dp4, xsl, xb, nb, cm, vc, bfrag, macc);
      else leaf_bf16_gen<NS, TG, KERNEL, F16, false, C16, 2, XP>(lane, dp4, xsl, xb, nb, cm, vc, bfrag, macc);
    }
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
    if constexpr (STAG) {                                                                                             \
      leaf_bf16_fused_half<NS, TG, KERNEL, F16, ASKIP, GMODE, C16, RTL, 0>(lane, dp4, panel + (q % 3) * NS * RT * 64, \
                                                                      xsl + ((q + 1) % 3) * xstride, xb, nb, cm, vc,  \
                                                                      bfrag, bnxt, acc, macc, fregs);                 \
      GPSO_BSTAMP(q, 2);                                                                                              \
      if (late) leaf_sync(); /* waves 4-7 meet the barrier in the middle of their step */                         \
      GPSO_BSTAMP(q, 3);                                                                                              \
      leaf_bf16_fused_half<NS, TG, KERNEL, F16, ASKIP, GMODE, C16, RTL, 1>(lane, dp4, panel + (q % 3) * NS * RT * 64, \
                                                                      xsl + ((q + 1) % 3) * xstride, xb, nb, cm, vc,  \
                                                                      bfrag, bnxt, acc, macc, fregs);                 \
      GPSO_BSTAMP(q, 4);                                                                                              \
      if (!late) leaf_sync();                                                                                     \
    } else if constexpr (M32) {                                                                                       \
      leaf_fused_step32<KERNEL, ASKIP, GMODE, C16, RTL>(dp4, panel + (q & 1) * NS * RT * 64,                          \
                                                        xsl + ((q + 1) % 3) * xstride, xb, ml, nb32, (float)cm, vc,   \
                                                        bfrag32, bnxt32, acc32, macc32);                              \
      GPSO_BSTAMP(q, 4);                                                                                              \
      leaf_sync();                                                                                                \
    } else {                                                                                                          \
      leaf_bf16_fused_step<NS, TG, KERNEL, F16, ASKIP, GMODE, C16, RTL, XP>(q, q_diag0, lane, dp4,                    \
                                                                   panel + (q & 1) * NS * RT * 64,                    \
                                                                   xsl + ((q + 1) % 3) * xstride, xb, nb, cm, vc,     \
                                                                   bfrag, bnxt, acc, macc);                           \
      GPSO_BSTAMP(q, 4);                                                                                              \
      leaf_sync();                                                                                                \
    }                                                                                                                 \
    GPSO_BSTAMP(q, 5);                                                                                                \
    if (GMODE != 0) {                                                                                                 \
      if constexpr (M32) {                                                                                            \
        for (int sp = 0; sp < 2; ++sp)                                                                                \
          for (int sl = 0; sl < 2; ++sl) bfrag32[sp][sl] = bnxt32[sp][sl];                                            \
      } else {                                                                                                        \
        for (int sp = 0; sp < NS; ++sp)                                                                               \
          for (int t = 0; t < CT; ++t) bfrag[sp][t] = bnxt[sp][t];                                                    \
      }                                                                                                               \
    }                                                                                                                 \
  }
    auto fused_loops = [&](auto rtl_) {
      constexpr int RTL = decltype(rtl_)::value;
      int q = 0;
      for (; q + 1 < q_diag0; ++q) GPSO_FUSED_STEP(0, 1)
      if (q < q_diag0) {  // the last step below the diagonal block generates the block's first step: with its k*.alpha
        GPSO_FUSED_STEP(0, 2)
        ++q;
      }
      static_for<0, RTL / 2>([&](auto j_) {  // the diagonal block: step j skips its 2 j all-zero row tiles
        constexpr int j = decltype(j_)::value;
        if (q < q_lim) {  // (workgroup-uniform)
          if constexpr (j + 1 < RT / 2) GPSO_FUSED_STEP(2 * j, 2)
          else GPSO_FUSED_STEP(2 * j, 0)
        }
        ++q;
      });
    };
    // a LAST row block with at most 128 training rows (q_max <= q_diag0 + 4): its row tiles 8 .. 15 are padding -- a
    // second copy of the loops that applies row tiles 0 .. 7 only (N = 1100: 76 rows in the fifth block)
    if (q_max <= q_diag0 + RT / 4) fused_loops(std::integral_constant<int, RT / 2>{});
    else fused_loops(std::integral_constant<int, RT>{});
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
    leaf_bf16_gen<NS, TG, KERNEL, F16, DIAGF, C16, 2, XP>(lane, dp4, xsl + (q % 3) * xstride, xb, nb, cm, vc, bfrag, macc); \
    GPSO_BSTAMP(q, 2);                                                                                                \
    if (ahead && q > 0) leaf_sync();                                                                              \
    GPSO_BSTAMP(q, 3);                                                                                                \
    leaf_bf16_apply<NS, F16, DIAGF>(q, q_diag0, lane, panel + (q & 1) * NS * RT * 64, bfrag, acc);                    \
    GPSO_BSTAMP(q, 4);                                                                                                \
    if (!ahead) leaf_sync();                                                                                      \
    GPSO_BSTAMP(q, 5);                                                                                                \
  }
  for (int q = 0; q < q_diag0; ++q) GPSO_BF16_STEP(false)
  for (int q = q_diag0; q < q_end; ++q) GPSO_BF16_STEP(true)
#undef GPSO_BF16_STEP
  if (ahead) leaf_sync();

  }

  // The next row block's first DMAs fly under this block's epilogue (conversions and f64 sums that touch no LDS): every wave is
  // past the last step's barrier, so the buffers they land in are free (-0.2 ... -0.4 % at C3 / C4 / C5; GPSO_EARLY_DMA=0: at the
  // top of the next block as in rounds 1-5).  The barrier that publishes them stands at the loop's header: leaf_sync() waits for
  // them explicitly -- the compiler does not, across the back edge.  From here on bi, q_diag0, q_end, q_lim and has_next are the
  // NEXT block's.
  const int bi_out = bi;
  const bool more = has_next;
  if (kEarlyDma && more) {  // (workgroup-uniform)
    if constexpr (STAG) leaf_sync();  // (waves 4-7 meet their last barrier in the middle of a step)
    set_row_block(rbj + 1);
    issue_block_start();
  }
  // (fp16 split: undo the power-of-two scales of the two operands -- exact)
  double unscale2 = 1.0, unscale_m = 1.0;
  if constexpr (F16) {
    const double ia = (double)inv_scale_a[1], ib = (double)inv_scale_b;  // ([0] is max |L^-1|, [1] = 2^-sa)
    unscale2 = (ia * ib) * (ia * ib);
    unscale_m = ib;
  }
  if constexpr (M32) {
    double sq = 0;
#pragma unroll
    for (int R = 0; R < RT / 2; ++R)
#pragma unroll
      for (int i = 0; i < 16; ++i) sq = fma((double)acc32[R][i], (double)acc32[R][i], sq);
    sq += __shfl_xor(sq, 32);
    double mm = (double)macc32;
    mm += __shfl_xor(mm, 32);
    sq *= unscale2;
    mm *= unscale_m;
    if (lane < 32) {
      const int64_t col = col0 + lane;
      part_var[(int64_t)bi_out * mpad + col] = sq;
      part_mean[(int64_t)bi_out * mpad + col] = mm;
    }
  } else {
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
        part_var[(int64_t)bi_out * mpad + col] = sq;
        part_mean[(int64_t)bi_out * mpad + col] = mm;
      }
    }
  }
  if (!more) break;  // (workgroup-uniform)
  }  // row blocks
}

// How many workgroups share a leaf tile's row blocks (gridDim.y of the split kernels).  GPSO_OPT_ROW_LOOP = 0: nbi, one row block
// each (rounds 1-5); v >= 2: v (tests); 1 (default): the count with the shortest modelled makespan.  In k-steps: row block bi costs
// 8 (bi + 1), a workgroup kPrologue on top (the leaf prologue + a cold first DMA, from the C3 A/B: 2.7 % of 288 + 8 x that);
// with S < nbi equal-weight workgroups the chip runs ceil(tiles S / CUs) rounds of the heaviest split; with one row block per
// workgroup, heaviest first, the dispatcher packs them to within a light block of the mean.  C3 (256 tiles): 1; C4 share
// (128): 2; C5 share (512): 1; ragged batches (462 or 1 384 tiles): nbi -- a coarser grain there costs a whole round of tail
// (measured: +4.7 % on bench.py --leaves grow --depth 11 with the first version of this rule, which only looked at the
// workgroup count).  A batch whose live row count only the device knows (m_live) keeps one row block per workgroup.
inline int leaf_row_splits(int64_t ltiles, int nbi, int ncu, bool live_known) {
  const int mode = g_leaf_row_loop;
  if (mode == 0 || nbi <= 1) return nbi;
  if (mode >= 2) return std::min(mode, nbi);
  if (!live_known) return nbi;
  constexpr double kPrologue = 1.2;
  const double total = 4.0 * nbi * (nbi + 1);
  const double per_cu = (double)ltiles * (total + kPrologue * nbi) / ncu + 8.0;
  double best = std::max(per_cu, 8.0 * nbi + kPrologue);
  int best_s = nbi;
  for (int S = nbi / 2; S >= 1; S /= 2) {  // (down, so that ties go to the smaller count)
    double heaviest = 0.0;
    for (int split = 0; split < S; ++split) {
      double w = 0.0;
      for (int j = 0;; ++j) {
        const int rank = j * S + ((j & 1) ? S - 1 - split : split);
        if (rank >= nbi) break;
        w += 8.0 * (nbi - rank);
      }
      heaviest = std::max(heaviest, w);
    }
    const double t = (double)((ltiles * S + ncu - 1) / ncu) * (heaviest + kPrologue);
    if (t <= best) best = t, best_s = S;
  }
  return best_s;
}

// FUSED and KS (kernel families 0-1: Matern-5/2, -3/2 | 2-3: Matern-1/2, squared exponential) are template parameters of
// the launchers: each (TG, FUSED, KS) slice of the kernels is instantiated in a translation unit of its own
// (launch_leaf_tiles_bf16_v<TG, FUSED, KS>, predict_split_*.hip), and the slices compile in parallel
template <bool FUSED, int KS, int NS, typename TG, bool F16 = false, int C16 = 0>
static int launch_leaf_tiles_bf16_ns(hipStream_t st, const void* linv_b, const TG* xs_p,
                                     const TG* xnorm, const float* alpha, const TG* leaves_s,
                                     const TG* lnorm, double* part_var, double* part_mean,
                                     int64_t npad, int dp4, int64_t mpad, const KernParams& kp,
                                     const int64_t* m_live, const float* inv_scale_a = nullptr,
                                     const float* c16_scale = nullptr, int64_t n_rows = 0, const RawLeaves& rawl = RawLeaves{}) {
  const int nbi = (int)(npad / 256);
  const dim3 grid((unsigned)(mpad / 256), (unsigned)nbi);
  constexpr bool STAG = kLeafStagger && FUSED && NS == 2 && F16 && C16 == 1;
  const size_t lds = leaf_bf16_lds_bytes(NS, dp4, (int)sizeof(TG), C16 != 0) + (STAG ? (size_t)NS * 16 * 64 * 16 : 0);  // (a third buffer of L^-1 pieces)
  if (C16 ? leaf_c16_chunks(dp4) != C16 : 2 * dp4 * (int)(sizeof(TG) / 4) > 32) {  // the X fragments of a k-step are one DMA window of 32 pieces (C16: 8)
    note_launch_error("launch_leaf_tiles_bf16: more than 32 X pieces per k-step");
    return 1;
  }
  // fp16 split: the generated tile is scaled by 2^sb, sigma^2 2^sb in [2^13, 2^14) (folded into the variance)
  int eb = 0;
  (void)frexp(kp.variance, &eb);
  const float var_arg = F16 ? (float)ldexp(kp.variance, 14 - eb) : (float)kp.variance;
  const float inv_b = F16 ? (float)ldexp(1.0, eb - 14) : 1.0f;
  const int q_max = n_rows > 0 ? (int)((n_rows + 31) / 32) : (int)(npad / 32);
  const int S = leaf_row_splits(mpad / 256, nbi, leaf_cu_count(), m_live == nullptr);
  g_leaf_last_splits = S;
  const dim3 grid_s((unsigned)(mpad / 256), (unsigned)S);
  // FUSED (GPSO_SPLIT_KERNEL_AUTO): the fused step; otherwise round 3's two-phase step.  Same bits either way.
  // rawl.step32 (GPSO_SPLIT_KERNEL_AUTO): the fused step on the 32x32x16 instruction, where that kernel exists
  constexpr bool kHasM32 = kLeafStep32 && FUSED && F16 && NS == 2 && C16 != 0 && sizeof(TG) == 4;
  const bool m32 = kHasM32 && rawl.step32;
  // xp: the last chunk of the fp16 contraction uses <= 16 slots (the scaled inputs and the norm slot behind them)
  const bool xp = C16 != 0 && dp4 * 4 + 1 - 32 * (C16 - 1) <= 16;
#define GPSO_L1(K, M32, XP)                                                                         \
  do {                                                                                              \
    const int rc = ensure_dyn_lds((const void*)leaf_tiles_bf16_kernel<NS, TG, K, F16, FUSED, C16, M32, XP>, (int)lds); \
    if (rc) return rc;                                                                              \
    hipLaunchKernelGGL((leaf_tiles_bf16_kernel<NS, TG, K, F16, FUSED, C16, M32, XP>), grid_s, dim3(512), lds, st, \
                       static_cast<const u32x4*>(linv_b), xs_p, xnorm, alpha, leaves_s, lnorm,      \
                       part_var, part_mean, (int)(npad / 16), dp4, mpad, nbi, var_arg, m_live,      \
                       inv_scale_a, inv_b, c16_scale, q_max, C16 ? rawl.x : nullptr, rawl.ls, rawl.m, rawl.d); \
  } while (0)
#define GPSO_L(K)                                   \
  do {                                              \
    if constexpr (kHasM32) {                        \
      if (m32) GPSO_L1(K, true, 0);                 \
      else if (xp) GPSO_L1(K, false, 1);            \
      else GPSO_L1(K, false, 0);                    \
    } else if constexpr (C16 != 0) {                \
      if (xp) GPSO_L1(K, false, 1);                 \
      else GPSO_L1(K, false, 0);                    \
    } else {                                        \
      GPSO_L1(K, false, 0);                         \
    }                                               \
  } while (0)
  if constexpr (KS == 0) {
    if (kp.kernel == 0) GPSO_L(0);
    else GPSO_L(1);
  } else {
    if (kp.kernel == 2) GPSO_L(2);
    else GPSO_L(3);
  }
#undef GPSO_L
#undef GPSO_L1
  (void)m32;
  (void)xp;
  return 0;
}

template <typename TG, bool FUSED, int KS>
int launch_leaf_tiles_bf16_v(hipStream_t st, int nsplit, const void* linv_b, const TG* xs_p,
                           const TG* xnorm, const float* alpha, const TG* leaves_s,
                           const TG* lnorm, double* part_var, double* part_mean, int64_t npad,
                           int dp4, int64_t mpad, const KernParams& kp, const int64_t* m_live,
                           const float* f16_inv_scale_a, const void* xs_h16, const float* c16_scale,
                           int64_t n_rows, const RawLeaves& rawl) {
  if (f16_inv_scale_a != nullptr) {  // fp16 split (nsplit == 2 pieces)
    if constexpr (sizeof(TG) == 4) {
      if (xs_h16 != nullptr && c16_scale != nullptr) {  // ... with the contraction on the fp16 pipe as well
        if (leaf_c16_chunks(dp4) == 1)
          return launch_leaf_tiles_bf16_ns<FUSED, KS, 2, TG, true, 1>(st, linv_b, static_cast<const TG*>(xs_h16), xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, f16_inv_scale_a, c16_scale, n_rows, rawl);
        return launch_leaf_tiles_bf16_ns<FUSED, KS, 2, TG, true, 2>(st, linv_b, static_cast<const TG*>(xs_h16), xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, f16_inv_scale_a, c16_scale, n_rows, rawl);
      }
    }
    return launch_leaf_tiles_bf16_ns<FUSED, KS, 2, TG, true>(st, linv_b, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, f16_inv_scale_a, nullptr, n_rows);
  }
  if (nsplit == 3)
    return launch_leaf_tiles_bf16_ns<FUSED, KS, 3, TG>(st, linv_b, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, nullptr, nullptr, n_rows);
  return launch_leaf_tiles_bf16_ns<FUSED, KS, 2, TG>(st, linv_b, xs_p, xnorm, alpha, leaves_s, lnorm, part_var, part_mean, npad, dp4, mpad, kp, m_live, nullptr, nullptr, n_rows);
}
}  // namespace gpso
