// Issue-rate probe for the split predict kernel's map (round 4): how long do independent v_fma_f32, v_exp_f32,
// v_sqrt_f32 and the 16x16x32 fp16 MFMA take per instruction, alone and MIXED (one MFMA followed by k vector
// instructions), with one and with two waves per SIMD?  Whole-kernel times by HIP events over long loops (s_memtime
// is a 100 MHz counter here); one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/issue_probe.hip -o tools/micro/issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// MODE 0: 16 independent v_fma | 1: 16 independent v_exp | 2: 16 independent v_sqrt | 3: 16 independent MFMAs
// 10 + k: 16 x (MFMA, k x v_fma) | 20 + k: 16 x (MFMA, k x v_exp)
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* sink, int iters, float seed) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(0.5f + i); }
  const float c1 = 1.0001f, c2 = 0.25f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    } else {
      constexpr int K = MODE % 10;
      constexpr bool TRANS = MODE >= 20;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if constexpr (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i + 4 * k) & 15]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 4 * k) & 15]) : "v"(c1), "v"(c2));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + acc[i][0] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* sink, int per_iter_mfma, int per_iter_valu) {
  const int iters = 20000;
  for (int threads : {256, 512}) {  // one / two waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<256, threads>>>(sink, 100, 1.5f);
    hipEventRecord(e0);
    probe<MODE><<<256, threads>>>(sink, iters, 1.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_iter = ms * 1e6 / iters;
    printf("%-34s %d wave(s)/SIMD: %8.1f ns per 16-group  (%d MFMA + %d vector per wave)\n", name, threads / 256, ns_iter,
           per_iter_mfma, per_iter_valu);
  }
}

int main() {
  float* sink;
  hipMalloc(&sink, 256 * 512 * 4);
  run<0>("16 v_fma_f32", sink, 0, 16);
  run<1>("16 v_exp_f32", sink, 0, 16);
  run<2>("16 v_sqrt_f32", sink, 0, 16);
  run<3>("16 mfma_16x16x32_f16", sink, 16, 0);
  run<11>("16 x (mfma + 1 v_fma)", sink, 16, 16);
  run<12>("16 x (mfma + 2 v_fma)", sink, 16, 32);
  run<13>("16 x (mfma + 3 v_fma)", sink, 16, 48);
  run<14>("16 x (mfma + 4 v_fma)", sink, 16, 64);
  run<21>("16 x (mfma + 1 v_exp)", sink, 16, 16);
  run<22>("16 x (mfma + 2 v_exp)", sink, 16, 32);
  return 0;
}
