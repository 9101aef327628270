// The fused step of the fp16-split predict kernel on the 32x32x16 matrix instruction (round 6; included by leaf_split.hpp).
//
// Why: the fused step on v_mfma_f32_16x16x32_f16 is bound by the SIMD's vector ISSUE port, not by the matrix pipe (DESIGN 4.1:
// per wave and k-step 108 MFMAs x 8 clocks of issue + 146 vector instructions + 41 LDS reads = 1 700 clocks of issue against
// 1 664 of pipe, two waves per SIMD).  An MFMA holds the issue port for 8 clocks whatever its shape (MI355X_MICROARCH.md, cycle
// constants): the same products as 54 instructions of 32 clocks instead of 108 of 16 take the same pipe time and HALF the issue
// slots -- 1 300 clocks of issue per wave and k-step against 1 728 of pipe.
//
// What changes, and what does not.  HBM and LDS layouts are the 16x16x32 kernel's: the L^-1 pieces (1 KB fragments of 16 rows x
// 32 k), the scaled inputs' piece pairs and the leaf fragments arrive exactly as before; the 32x32x16 operands are the SAME
// 16-byte words fetched by other lanes (a ds_read_b128 takes any address per lane):
//   * A operand of the apply, lane (rho = l & 31, g = l >> 5), k-slice sl of the step: rows 32 R + rho, the 16-byte word of
//     fragment 2 R + (rho >> 4) at its lane (rho & 15) + 16 (2 sl + g) -- which holds (pack_linv_f16_kernel's k order, made for
//     the 16x16x32 kernel's generated operand) the step's training points 16 hf + 4 (2 sl + g) + r, element 4 hf + r;
//   * contraction: rows = the step's 32 training points, fetched in the order phi(rho) = rho with bits 3 and 4 swapped, so that
//     output register i = 8 sl + 4 hf + r of lane (leaf l & 31, g) -- row r + 8 (2 sl + hf) + 4 g of the instruction's output --
//     is training point 16 hf + 8 sl + 4 g + r: registers 8 sl .. 8 sl + 7 are, in order, the B operand of k-slice sl.
// A wave's tile is 256 rows x 32 leaves as before: 8 accumulators of 32 x 32 (128 registers).  A "unit" u = 2 R + sl (16 per
// step) plays the part of the 16x16 kernel's row tile: two fragment reads, three MFMAs, a share of the next step's map; the
// diagonal block's step j skips its first 2 j units -- the same count as row tiles.  The sums differ from the 16x16 kernel's in
// the order the matrix instruction adds 32 products (two instructions of 16 instead of one of 32): same tolerance class, not
// the same bits.
// (no include guard: included once, inside namespace gpso, below the definitions it uses)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// per-lane offsets (in 16-byte words) into the 16x16x32-ordered fragment arrays
struct M32Lanes {
  int la;  // apply A operand: + (piece * 16 + 2 R) * 64 + 32 sl
  int xa;  // contraction A operand (training side): + (chunk * 2 + piece) * 64 + 32 sigma
  int lb;  // contraction B operand (leaf side):     + (chunk * 2 + piece) * 64 + 32 sigma
  int g;   // l >> 5
};
template <int C16>
__device__ __forceinline__ M32Lanes m32_lanes(int lane) {
  M32Lanes m;
  const int g = lane >> 5, rho = lane & 31;
  const int pi = (rho & 7) | ((rho & 8) << 1) | ((rho & 16) >> 1);  // phi: bits 3 and 4 swapped
  m.g = g;
  m.la = ((lane >> 4) & 1) * 64 + (lane & 15) + 16 * g;
  m.xa = (pi >> 4) * (C16 * 2 * 64) + (pi & 15) + 16 * g;
  m.lb = ((lane >> 4) & 1) * (C16 * 2 * 64) + (lane & 15) + 16 * g;
  return m;
}

#define GPSO_MFMA32(A, B, C) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)

// s += (the step's 32 training points, in the order phi) x (this wave's 32 leaves): three fp16 products per chunk of 32 slots,
// small terms first, each over the chunk's two 16-slot halves
template <int C16>
__device__ __forceinline__ void leaf_contract32(const unsigned char* xs_b, const void* xb, const M32Lanes& ml, f32x16& s) {
  const u32x4* xa = reinterpret_cast<const u32x4*>(xs_b) + ml.xa;
  const u32x4* lb = reinterpret_cast<const u32x4*>(xb) + ml.lb;
#pragma unroll
  for (int cc = 0; cc < C16; ++cc) {
    u32x4 a[2][2], b[2][2];  // [sigma][piece]
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        a[sg][pc] = xa[(cc * 2 + pc) * 64 + 32 * sg];
        b[sg][pc] = lb[(cc * 2 + pc) * 64 + 32 * sg];
      }
#define GPSO_XX(PA, PB) \
  _Pragma("unroll") for (int sg = 0; sg < 2; ++sg) s = GPSO_MFMA32(a[sg][PA], b[sg][PB], s)
    GPSO_XX(1, 0);
    GPSO_XX(0, 1);
    GPSO_XX(0, 0);
#undef GPSO_XX
  }
}

// alpha of the 16 points a lane's registers stand for: register 8 sl + 4 hf + r = point 16 hf + 8 sl + 4 g + r
__device__ __forceinline__ void m32_alpha(const float* alp, int g, float (&al)[16]) {
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(alp + 16 * hf + 8 * sl + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) al[8 * sl + 4 * hf + r] = a4[r];
    }
}

// a k-step generated on its own (the first step of a row block): contraction, map, split -- leaf_bf16_gen's arithmetic per value
template <int KERNEL, bool DIAG, int C16>
__device__ __forceinline__ void leaf_gen32(int dp4, const unsigned char* xs_b, const void* xb, const M32Lanes& ml, float nb,
                                           float cm, const float (&vc)[3], u32x4 (&bfrag)[2][2] /* [piece][k-slice] */,
                                           float& macc) {
  const int XF = Bf16Lds<float, C16>::xfrag(dp4);
  f32x16 s;
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = 0.0f;
  leaf_contract32<C16>(xs_b, xb, ml, s);
  float p[16], e[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = fmaf(cm, s[i], nb);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KERNEL != 3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = __builtin_amdgcn_sqrtf(__builtin_fabsf(p[i]));
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(-p[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if constexpr (KERNEL == 0) p[i] = fmaf(p[i], fmaf(p[i], vc[2], vc[1]), vc[0]) * e[i];
    else if constexpr (KERNEL == 1) p[i] = fmaf(p[i], vc[1], vc[0]) * e[i];
    else p[i] = vc[0] * e[i];
  }
  if constexpr (DIAG) {
    float al[16];
    m32_alpha(reinterpret_cast<const float*>(xs_b + XF + 64 * sizeof(float)), ml.g, al);
#pragma unroll
    for (int i = 0; i < 16; ++i) macc = fmaf(p[i], al[i], macc);
  }
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    u32x4 f0, f1;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      unsigned hh, ll;
      f16_split_pair_both(p[8 * sl + 2 * jj], p[8 * sl + 2 * jj + 1], hh, ll);
      f0[jj] = hh;
      f1[jj] = ll;
    }
    bfrag[0][sl] = f0;
    bfrag[1][sl] = f1;
  }
}

// The fused step: apply of step q (16 units of three MFMAs) with the contraction and the map of step q + 1 dealt behind them.
// GMODE, ASKIP, RTL as in leaf_bf16_fused_step (ASKIP, RTL count units = 16-row tiles).
template <int KERNEL, int ASKIP, int GMODE, int C16, int RTL>
__device__ __forceinline__ void leaf_fused_step32(int dp4, const u32x4* panel_b /* [2][16][64]: L^-1 pieces of step q */,
                                                  const unsigned char* xs_n /* inputs of step q + 1 */, const void* xb,
                                                  const M32Lanes& ml, float nb, float cm, const float (&vc)[3],
                                                  const u32x4 (&bcur)[2][2], u32x4 (&bnxt)[2][2], f32x16 (&acc)[8],
                                                  float& macc) {
  constexpr int RT = 16;
  constexpr bool GEN = GMODE != 0;
  const int XF = Bf16Lds<float, C16>::xfrag(dp4);
  f32x16 s;
  if constexpr (GEN) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.0f;
    leaf_contract32<C16>(xs_n, xb, ml, s);
  }
  // ops in stage-major order, register i = 8 sl + j: combine (16) | sqrt (16; none for the squared exponential) | exp2 (16) |
  // polynomial x exponential (16) | k*.alpha (16; GMODE 2) | split of pair (sl, jj) (8)
  constexpr int E = 16;
  constexpr int O_SQRT = E, O_EXP = O_SQRT + (KERNEL == 3 ? 0 : E), O_POLY = O_EXP + E, O_MEAN = O_POLY + E,
                O_SPLIT = O_MEAN + (GMODE == 2 ? E : 0), NOPS = O_SPLIT + 8;
  float p[16], ex[16], al[16];
  u32x4 fr[2][2];
  if constexpr (GMODE == 2) m32_alpha(reinterpret_cast<const float*>(xs_n + XF + 64 * sizeof(float)), ml.g, al);
  auto op = [&](auto o_) {
    constexpr int o = decltype(o_)::value;
    if constexpr (o < O_SQRT) {
      p[o] = fmaf(cm, s[o], nb);
    } else if constexpr (o < O_EXP) {
      constexpr int i = o - O_SQRT;
      p[i] = __builtin_amdgcn_sqrtf(__builtin_fabsf(p[i]));
    } else if constexpr (o < O_POLY) {
      constexpr int i = o - O_EXP;
      ex[i] = __builtin_amdgcn_exp2f(-p[i]);
    } else if constexpr (o < O_MEAN) {
      constexpr int i = o - O_POLY;
      if constexpr (KERNEL == 0) p[i] = fmaf(p[i], fmaf(p[i], vc[2], vc[1]), vc[0]) * ex[i];
      else if constexpr (KERNEL == 1) p[i] = fmaf(p[i], vc[1], vc[0]) * ex[i];
      else p[i] = vc[0] * ex[i];
    } else if constexpr (o < O_SPLIT) {
      constexpr int i = o - O_MEAN;
      macc = fmaf(p[i], al[i], macc);
    } else {
      constexpr int e = o - O_SPLIT, sl = e >> 2, jj = e & 3;
      unsigned hh, ll;
      f16_split_pair_both(p[8 * sl + 2 * jj], p[8 * sl + 2 * jj + 1], hh, ll);
      fr[0][sl][jj] = hh;
      fr[1][sl][jj] = ll;
    }
  };
  static_assert(ASKIP >= 0 && ASKIP < RTL && RTL <= RT && ASKIP % 2 == 0 && RTL % 2 == 0, "whole 32-row tiles");
  const u32x4* pa = panel_b + ml.la;
  u32x4 a[2][2];  // [unit parity][piece]
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) a[ASKIP & 1][sp] = pa[(sp * RT + (ASKIP & ~1)) * 64 + 32 * (ASKIP & 1)];
  static_for<0, RTL>([&](auto u_) {
    constexpr int u = decltype(u_)::value;
    if constexpr (u + 1 < RTL && u + 1 > ASKIP) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) a[(u + 1) & 1][sp] = pa[(sp * RT + ((u + 1) & ~1)) * 64 + 32 * ((u + 1) & 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (u >= ASKIP) {  // (diagonal block: all-zero tiles above the diagonal)
      constexpr int R = u >> 1, sl = u & 1;
      f32x16 c = acc[R];
      c = GPSO_MFMA32(a[u & 1][1], bcur[0][sl], c);
      c = GPSO_MFMA32(a[u & 1][0], bcur[1][sl], c);
      c = GPSO_MFMA32(a[u & 1][0], bcur[0][sl], c);
      acc[R] = c;
    }
    if constexpr (GEN && u >= 1) {  // this unit's share of the map
      static_for<(u - 1) * NOPS / (RTL - 1), u * NOPS / (RTL - 1)>(op);
    }
  });
  if constexpr (GEN) {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) bnxt[sp][sl] = fr[sp][sl];
  }
}
