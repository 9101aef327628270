// Minimal probe for the packed-mean failure (profiles/r02h_packed_mean_bug.txt): does a DEPENDENT chain of
// v_pk_fma_f32, spaced exactly as hipcc 7.2 spaces it (one wait state -- `s_nop 0` or one unrelated VALU -- between a
// link and the link that consumes its result), ever deliver a wrong HIGH half while the OTHER wave of its SIMD keeps the
// matrix pipe / the double-rate VALU busy?  Two hazards are probed, both taken from the failing build's disassembly
// (tools/micro/packed_mean_probe.hip, packed build, leaf_tiles_bf16_kernel<2, double, 0>, %.preheader108.i):
//   F  forwarding:  v_pk_fma acc, p, a, acc ; s_nop <N> ; v_pk_fma acc, p', a', acc ...   (N = 0 is the compiler's choice)
//   W  write-after-read:  v_pk_fma acc, p, v[x:x+1], acc op_sel_hi:[1,0,1] ; v_mov_b32 v[x], other
//      (both halves of the link read v[x]; the v_mov right behind it is what the compiler emitted; a high half that reads
//      its operand late would see `other`)
// Workgroup = 8 waves (two per SIMD, as the real kernel): waves 4-7 run the chains, waves 0-3 the PARTNER stream:
//   0 nothing | 1 v_mfma_f32_16x16x32_bf16 back to back | 2 v_mfma_f64_16x16x4_f64 back to back | 3 v_fma_f64 stream
// Every chain is computed twice -- packed, and with scalar v_fma_f32 on the same inputs -- and compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/pk_hazard_probe.hip -o tools/micro/pk_hazard_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8 dependent links, the real chain's shape; p[k] = (tile 0, tile 1) values, a = 8 multipliers (broadcast to both halves)
template <int MODE /* 0: F, 1: W */, int NOPS>
__device__ __forceinline__ f32x2 chain_packed(f32x2 acc, const f32x2 (&p)[8], f32x4 a0, f32x4 a1) {
  if constexpr (MODE == 0) {
#define GPSO_LINKS(GAP)                                                                              \
    asm volatile("v_pk_fma_f32 %0, %1, %9, %0 op_sel_hi:[1,0,1]\n\t" GAP                              \
                 "v_pk_fma_f32 %0, %2, %9, %0 op_sel:[0,1,0]\n\t" GAP                                 \
                 "v_pk_fma_f32 %0, %3, %10, %0 op_sel_hi:[1,0,1]\n\t" GAP                             \
                 "v_pk_fma_f32 %0, %4, %10, %0 op_sel:[0,1,0]\n\t" GAP                                \
                 "v_pk_fma_f32 %0, %5, %11, %0 op_sel_hi:[1,0,1]\n\t" GAP                             \
                 "v_pk_fma_f32 %0, %6, %11, %0 op_sel:[0,1,0]\n\t" GAP                                \
                 "v_pk_fma_f32 %0, %7, %12, %0 op_sel_hi:[1,0,1]\n\t" GAP                             \
                 "v_pk_fma_f32 %0, %8, %12, %0 op_sel:[0,1,0]\n\ts_nop 1"                             \
                 : "+v"(acc)                                                                         \
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), \
                   "v"(f32x2{a0[0], a0[1]}), "v"(f32x2{a0[2], a0[3]}), "v"(f32x2{a1[0], a1[1]}), "v"(f32x2{a1[2], a1[3]}))
    if constexpr (NOPS < 0) { GPSO_LINKS(""); }             // NO wait state: below the compiler's rule (sensitivity check)
    else if constexpr (NOPS == 0) { GPSO_LINKS("s_nop 0\n\t"); }  // hipcc's choice: one wait state
    else if constexpr (NOPS == 1) { GPSO_LINKS("s_nop 1\n\t"); }
    else { GPSO_LINKS("s_nop 3\n\t"); }
#undef GPSO_LINKS
  } else {
    // the compiler's own sequence around links 3 and 4: the multiplier register of a link is overwritten by the v_mov
    // right behind it.  v[100:101] is the scratch pair: v100 is read by BOTH halves of a link (op_sel_hi [1,0,1]), then
    // rewritten by the very next instruction
    asm volatile(
        "v_mov_b32 v100, %1\n\tv_mov_b32 v101, 0\n\ts_nop 1\n\t"
        "v_pk_fma_f32 %0, %9, v[100:101], %0 op_sel_hi:[1,0,1]\n\t"   // a0[0]
        "v_mov_b32 v100, %2\n\t"                                      // write-after-read on the link above
        "v_pk_fma_f32 %0, %10, v[100:101], %0 op_sel_hi:[1,0,1]\n\t"  // a0[2]
        "v_mov_b32 v100, %3\n\t"
        "v_pk_fma_f32 %0, %11, v[100:101], %0 op_sel_hi:[1,0,1]\n\t"  // a0[3]
        "v_mov_b32 v100, %4\n\t"
        "v_pk_fma_f32 %0, %12, v[100:101], %0 op_sel_hi:[1,0,1]\n\t"  // a1[0]
        "v_mov_b32 v100, %5\n\t"
        "v_pk_fma_f32 %0, %13, v[100:101], %0 op_sel_hi:[1,0,1]\n\t"  // a1[1]
        "v_mov_b32 v100, %6\n\t"
        "v_pk_fma_f32 %0, %14, v[100:101], %0 op_sel_hi:[1,0,1]\n\t"  // a1[2]
        "v_mov_b32 v100, %7\n\t"
        "v_pk_fma_f32 %0, %15, v[100:101], %0 op_sel_hi:[1,0,1]\n\t"  // a1[3]
        "v_mov_b32 v100, %8\n\t"
        "v_pk_fma_f32 %0, %16, v[100:101], %0 op_sel_hi:[1,0,1]\n\ts_nop 1"  // a0[3] again
        : "+v"(acc)
        : "v"(a0[0]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(a0[3]),
          "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "v100", "v101");
  }
  return acc;
}
template <int MODE>
__device__ __forceinline__ f32x2 chain_scalar(f32x2 acc, const f32x2 (&p)[8], f32x4 a0, f32x4 a1) {
  const float m0[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  const float m1[8] = {a0[0], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a0[3]};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float m = (MODE == 0) ? m0[k] : m1[k];
    acc.x = __builtin_fmaf(p[k].x, m, acc.x);
    acc.y = __builtin_fmaf(p[k].y, m, acc.y);
    asm volatile("" : "+v"(acc.x), "+v"(acc.y));  // scalar, not re-packed
  }
  return acc;
}

template <int MODE, int NOPS, int PARTNER>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, int iters, unsigned* __restrict__ bad,
                                             float* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ f32x4 alpha_lds[2 * 64];
  if (tid < 128) alpha_lds[tid] = f32x4{in[4 * tid], in[4 * tid + 1], in[4 * tid + 2], in[4 * tid + 3]};
  __syncthreads();
  if (wave < 4) {  // the partner stream of this SIMD
    if (PARTNER == 1) {
      f32x4 c[4] = {};
      bf16x8 a = {}, b = {};
      for (int it = 0; it < iters * 6; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u], 0, 0, 0);
      sink[blockIdx.x * 512 + tid] = c[0][0] + c[1][0] + c[2][0] + c[3][0];
    } else if (PARTNER == 2) {
      f64x4 c[4] = {};
      const double a = in[lane], b = in[lane + 64];
      for (int it = 0; it < iters * 2; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u], 0, 0, 0);
      sink[blockIdx.x * 512 + tid] = (float)(c[0][0] + c[1][0] + c[2][0] + c[3][0]);
    } else if (PARTNER == 3) {
      double c[4] = {1, 2, 3, 4};
      const double a = in[lane];
      for (int it = 0; it < iters * 16; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = __builtin_fma(c[u], 0.999, a);
      sink[blockIdx.x * 512 + tid] = (float)(c[0] + c[1] + c[2] + c[3]);
    }
    return;
  }
  unsigned nbad = 0;
  f32x2 accp = {0.f, 0.f}, accs = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x2 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = in[(it * 8 + k + lane) & 1023];
      p[k] = f32x2{v, 1.0f - v};
      asm volatile("" : "+v"(p[k]));
    }
    // multipliers straight from LDS, as in the kernel (two ds_read_b128)
    const f32x4 a0 = alpha_lds[(it + lane) & 63], a1 = alpha_lds[64 + ((it + lane) & 63)];
    accp = chain_packed<MODE, NOPS>(accp, p, a0, a1);
    accs = chain_scalar<MODE>(accs, p, a0, a1);
    if (__builtin_bit_cast(unsigned, accp.x) != __builtin_bit_cast(unsigned, accs.x)) nbad += 1;       // low half
    if (__builtin_bit_cast(unsigned, accp.y) != __builtin_bit_cast(unsigned, accs.y)) nbad += 0x10000;  // high half
    accp = accs;  // (keep both chains on the same trajectory after a fault)
    accp.x *= 0.5f; accp.y *= 0.5f; accs = accp;
  }
  atomicAdd(&bad[0], nbad & 0xffff);
  atomicAdd(&bad[1], nbad >> 16);
  sink[blockIdx.x * 512 + tid] = accp.x + accp.y;
}

template <int MODE, int NOPS, int PARTNER>
static void run(const float* din, unsigned* dbad, float* dsink, int iters, int blocks, int launches) {
  unsigned tot[2] = {0, 0};
  for (int l = 0; l < launches; ++l) {
    hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL((probe<MODE, NOPS, PARTNER>), dim3(blocks), dim3(512), 0, 0, din, iters, dbad, dsink);
    unsigned h[2];
    hipMemcpy(h, dbad, 8, hipMemcpyDeviceToHost);
    tot[0] += h[0];
    tot[1] += h[1];
  }
  static const char* partner[] = {"idle", "bf16 MFMA", "f64 MFMA", "v_fma_f64"};
  printf("%s  s_nop between links: %2d (-1: none)  partner wave: %-9s  chains %lld  wrong low halves %u  wrong HIGH halves %u\n",
         MODE == 0 ? "F (forwarding)    " : "W (write-after-read)", NOPS, partner[PARTNER],
         (long long)launches * blocks * 256 * iters, tot[0], tot[1]);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, blocks = argc > 2 ? atoi(argv[2]) : 256, launches = argc > 3 ? atoi(argv[3]) : 3;
  std::vector<float> in(1024 + 64);
  unsigned s = 7;
  for (auto& v : in) { s = s * 1664525u + 1013904223u; v = 0.25f + (float)(s >> 8) / (1 << 24); }
  float *din, *dsink; unsigned* dbad;
  hipMalloc(&din, in.size() * 4); hipMalloc(&dsink, (size_t)blocks * 512 * 4); hipMalloc(&dbad, 8);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
#define ROW(M, N) run<M, N, 0>(din, dbad, dsink, iters, blocks, launches); run<M, N, 1>(din, dbad, dsink, iters, blocks, launches); \
                  run<M, N, 2>(din, dbad, dsink, iters, blocks, launches); run<M, N, 3>(din, dbad, dsink, iters, blocks, launches);
  ROW(0, -1) ROW(0, 0) ROW(0, 1) ROW(0, 3) ROW(1, 0)
  return 0;
}
