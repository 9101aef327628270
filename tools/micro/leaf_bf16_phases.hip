// Where do the waves of leaf_tiles_bf16_kernel (split-bf16 x6, float generation) spend a k-step?  s_memtime
// stamps of wave 0 (generates step q, then applies it) and wave 4 (applies step q, then generates q + 1) of
// the heaviest workgroup, inside the real kernel at the C3 shape:
//   0 step begin | 1 LDS-DMA issued | 2 generation done | 3 (waves 4-7: barrier) | 4 apply issued | 5 (waves 0-3: barrier)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I pygpso_amd/csrc tools/micro/leaf_bf16_phases.hip -o tools/micro/leaf_bf16_phases.bin
__device__ long long g_bst[2 * 64 * 8];
#ifndef GPSO_NOSTAMP  // -DGPSO_NOSTAMP: the kernel as shipped (clean kernel times of the ablation builds)
#define GPSO_BSTAMP(q, i)                                                                              \
  do {                                                                                                  \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && (q) < 64)    \
      g_bst[((threadIdx.x >> 8) * 64 + (q)) * 8 + (i)] = __builtin_amdgcn_s_memtime();                  \
  } while (0)
#endif
__device__ long long g_cst[2 * 64 * 8];
#ifndef GPSO_NOSTAMP
#define GPSO_CSTAMP(q, i)                                                                              \
  do {                                                                                                  \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && (q) < 64 && (i) < 8) \
      g_cst[((threadIdx.x >> 8) * 64 + (q)) * 8 + (i)] = __builtin_amdgcn_s_memtime();                  \
  } while (0)
#endif
#include "../../pygpso_amd/csrc/predict.hip"
#ifndef GPSO_CSTAMP
#define GPSO_CSTAMP(q, i)
#endif
#include "../attic/leaf_tiles_wide.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
namespace gpso {
int ensure_dyn_lds(const void* fn, int bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return 0;
}
void note_launch_error(const char*) {}
}  // namespace gpso
using namespace gpso;

int main(int argc, char** argv) {
  const int64_t npad = 2048, m = 65536;
  const bool f16 = argc > 1 && argv[1][0] == 'f';    // "f": the fp16 split (two planes, three products)
  const bool x3 = f16 || (argc > 1 && argv[1][0] == '3') || (argc > 2 && argv[2][0] == '3');  // "3" (also as a second argument, behind w / s): bf16 x3
  const int dp4 = 3, dp = 12, ns = x3 ? 2 : 3;
  std::vector<float> linv((size_t)npad * npad, 0.f), xsp(npad * dp), xn(npad), al(npad), lv(m * dp), ln(m);
  unsigned s = 1; auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24); };
  for (int64_t i = 0; i < npad; ++i)
    for (int64_t j = 0; j <= i; ++j) linv[i * npad + j] = 0.01f * (rnd() - 0.5f);
  for (auto& v : xsp) v = rnd();
  for (auto& v : xn) v = 3.0f + rnd();
  for (auto& v : al) v = rnd() - 0.5f;
  for (auto& v : lv) v = rnd();
  for (auto& v : ln) v = 3.0f + rnd();
  float *dl, *dx, *dn, *da, *dlv, *dln; double *pv, *pm; void* lb;
  hipMalloc(&dl, linv.size() * 4); hipMalloc(&dx, xsp.size() * 4); hipMalloc(&dn, npad * 4); hipMalloc(&da, npad * 4);
  hipMalloc(&dlv, lv.size() * 4); hipMalloc(&dln, m * 4); hipMalloc(&pv, 8 * m * 8); hipMalloc(&pm, 8 * m * 8);
  hipMalloc(&lb, (size_t)ns * npad * npad * 2 + 256);
  float* f16_scal = reinterpret_cast<float*>(static_cast<char*>(lb) + (size_t)ns * npad * npad * 2);
  hipMemcpy(dl, linv.data(), linv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, xsp.data(), xsp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, xn.data(), npad * 4, hipMemcpyHostToDevice); hipMemcpy(da, al.data(), npad * 4, hipMemcpyHostToDevice);
  hipMemcpy(dlv, lv.data(), lv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dln, ln.data(), m * 4, hipMemcpyHostToDevice);
  const bool two_phase = argc > 2 && argv[2][0] == '2';  // second argument "2": round 3's two-phase step instead of the fused step
  const bool stream = argc > 1 && argv[1][0] == 's';  // "s": the one-wave-per-SIMD stream kernel (leaf_tiles_bf16s_kernel)
  const bool wide = stream || (argc > 1 && argv[1][0] == 'w');  // "w": the 32x32x16 kernel (leaf_tiles_bf16w_kernel)
  float* dxw = nullptr;
  if (wide) {
    std::vector<double> xs64((size_t)npad * dp);
    for (auto& v : xs64) v = rnd();
    double* dxs; hipMalloc(&dxs, xs64.size() * 8); hipMemcpy(dxs, xs64.data(), xs64.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&dxw, (size_t)(npad / 32) * leaf_bf16w_dpw(dp) * 64 * 4);
    launch_gen_inputs_wide(0, dxs, npad, dp, dp, 0, dxw);
    launch_pack_linv_bf16w<float>(0, ns, dl, npad, npad, lb);
  } else if (f16) {
    launch_pack_linv_f16<float>(0, dl, npad, npad, f16_scal, lb);
  } else {
    launch_pack_linv_bf16<float>(0, ns, dl, npad, npad, lb);
  }
  KernParams kp{0, 1.0, 1e-3, 0.0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    if (stream) launch_leaf_tiles_bf16s(0, ns, lb, dxw, da, dlv, dln, pv, pm, npad, dp, dp, m, kp, nullptr);
    else if (wide) launch_leaf_tiles_bf16w(0, ns, lb, dxw, da, dlv, dln, pv, pm, npad, dp, dp, m, kp, nullptr);
    else launch_leaf_tiles_bf16<float>(0, ns, lb, dx, dn, da, dlv, dln, pv, pm, npad, dp4, m, kp, nullptr, f16 ? f16_scal : nullptr, two_phase ? 1 : 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> g(2 * 64 * 8);
    hipMemcpyFromSymbol(g.data(), HIP_SYMBOL(g_bst), g.size() * 8);
    printf("%s%s kernel %.3f ms (with stamps)\n", wide ? "32x32x16" : "16x16x32", f16 ? " fp16 x3" : (x3 ? " bf16 x3" : " bf16 x6"), ms);
    if (stream) {  // wave 0 of the heaviest workgroup: begin | contraction done | group 7 done | DMA issued | group 15 done | DMA landed | (barrier)
      double seg[6] = {0, 0, 0, 0, 0, 0}; int cnt = 0;
      for (int q = 8; q < 52; ++q) {
        const long long* a = g.data() + q * 8;
        for (int i = 0; i < 5; ++i) seg[i] += (double)(a[i + 1] - a[i]);
        seg[5] += (double)(a[8] - a[5]);
        ++cnt;
      }
      printf("  stream, clocks per k-step: contraction %.0f | groups 0-7 %.0f | groups 8-15 %.0f | input DMA issue %.0f | wait for the DMA %.0f | barrier %.0f | step %.0f\n",
             seg[0] / cnt, seg[1] / cnt, seg[2] / cnt, seg[3] / cnt, seg[4] / cnt, seg[5] / cnt, (seg[0] + seg[1] + seg[2] + seg[3] + seg[4] + seg[5]) / cnt);
      continue;
    }
    for (int w = 0; w < 2; ++w) {  // heaviest workgroup: bi = 7 -> 64 k-steps (56 full + 8 diagonal)
      double seg[5] = {0, 0, 0, 0, 0}, tot = 0; int cnt = 0;
      for (int q = 8; q < 52; ++q) {
        const long long* a = g.data() + (w * 64 + q) * 8;
        for (int i = 0; i < 5; ++i) seg[i] += (double)(a[i + 1] - a[i]);
        tot += (double)(a[8] - a[0]);
        ++cnt;
      }
      if (!two_phase && !wide) {  // fused step: stamps 0 | 1 DMA issued | 4 fused step issued | 5 behind the barrier
        double d = 0, f = 0, b = 0;
        for (int q = 8; q < 52; ++q) {
          const long long* a = g.data() + (w * 64 + q) * 8;
          d += (double)(a[1] - a[0]); f += (double)(a[4] - a[1]); b += (double)(a[5] - a[4]);
        }
        printf("  wave %d, clocks per k-step (fused): issue DMA %.0f | apply(q) + generate(q + 1) %.0f | barrier %.0f | step %.0f\n", 4 * w, d / cnt, f / cnt, b / cnt, tot / cnt);
        continue;
      }
      printf("  wave %d, clocks per k-step: issue DMA %.0f | generation %.0f | %s %.0f | apply %.0f | %s %.0f | step %.0f\n", 4 * w,
             seg[0] / cnt, seg[1] / cnt, w ? "barrier" : "-", seg[2] / cnt, seg[3] / cnt, w ? "-" : "barrier", seg[4] / cnt, tot / cnt);
      if (wide) {  // the MFMA stream = contraction of the next step (stamp 3 -> 6), then the apply (6 -> 4)
        double g0 = 0;
        for (int q = 8; q < 52; ++q) {
          const long long* a = g.data() + (w * 64 + q) * 8;
          g0 += (double)(a[6] - a[3]);
        }
        printf("          of the apply column: contraction of the next step %.0f\n", g0 / cnt);
      }
    }
  }
  return 0;
}
