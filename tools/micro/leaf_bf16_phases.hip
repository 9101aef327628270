// Where do the waves of leaf_tiles_bf16_kernel (split-bf16 x6, float generation) spend a k-step?  s_memtime
// stamps of wave 0 (generates step q, then applies it) and wave 4 (applies step q, then generates q + 1) of
// the heaviest workgroup, inside the real kernel at the C3 shape:
//   0 step begin | 1 LDS-DMA issued | 2 generation done | 3 (waves 4-7: barrier) | 4 apply issued | 5 (waves 0-3: barrier)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I pygpso_amd/csrc tools/micro/leaf_bf16_phases.hip -o tools/micro/leaf_bf16_phases.bin
__device__ long long g_bst[2 * 64 * 8];
#define GPSO_BSTAMP(q, i)                                                                              \
  do {                                                                                                  \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && (q) < 64)    \
      g_bst[((threadIdx.x >> 8) * 64 + (q)) * 8 + (i)] = __builtin_amdgcn_s_memtime();                  \
  } while (0)
#include "../../pygpso_amd/csrc/predict.hip"
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
  const int dp4 = 3, dp = 12, ns = 3;
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
  hipMalloc(&lb, (size_t)ns * npad * npad * 2);
  hipMemcpy(dl, linv.data(), linv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, xsp.data(), xsp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, xn.data(), npad * 4, hipMemcpyHostToDevice); hipMemcpy(da, al.data(), npad * 4, hipMemcpyHostToDevice);
  hipMemcpy(dlv, lv.data(), lv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dln, ln.data(), m * 4, hipMemcpyHostToDevice);
  launch_pack_linv_bf16<float>(0, ns, dl, npad, npad, lb);
  KernParams kp{0, 1.0, 1e-3, 0.0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    launch_leaf_tiles_bf16<float>(0, ns, lb, dx, dn, da, dlv, dln, pv, pm, npad, dp4, m, kp, nullptr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> g(2 * 64 * 8);
    hipMemcpyFromSymbol(g.data(), HIP_SYMBOL(g_bst), g.size() * 8);
    printf("kernel %.3f ms (with stamps)\n", ms);
    for (int w = 0; w < 2; ++w) {  // heaviest workgroup: bi = 7 -> 64 k-steps (56 full + 8 diagonal)
      double seg[5] = {0, 0, 0, 0, 0}, tot = 0; int cnt = 0;
      for (int q = 8; q < 52; ++q) {
        const long long* a = g.data() + (w * 64 + q) * 8;
        for (int i = 0; i < 5; ++i) seg[i] += (double)(a[i + 1] - a[i]);
        tot += (double)(a[8] - a[0]);
        ++cnt;
      }
      printf("  wave %d, clocks per k-step: issue DMA %.0f | generation %.0f | %s %.0f | apply %.0f | %s %.0f | step %.0f\n", 4 * w,
             seg[0] / cnt, seg[1] / cnt, w ? "barrier" : "-", seg[2] / cnt, seg[3] / cnt, w ? "-" : "barrier", seg[4] / cnt, tot / cnt);
    }
  }
  return 0;
}
