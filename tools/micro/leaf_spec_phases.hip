// Where do the waves of leaf_tiles_spec_kernel (fp16 x3, float generation) spend a k-step?  s_memtime stamps of the
// heaviest workgroup, inside the real kernel at the C3 shape, for one wave of every role and SIMD slot:
//   0 step begin | 1 LDS-DMA issued | 2 work done (apply: 48 MFMAs; generator: the next step's pieces) | 3 behind the barrier
// plus HW_REG_HW_ID of every wave (which SIMD / CU it landed on).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops \
//         -I pygpso_amd/csrc tools/micro/leaf_spec_phases.hip -o tools/micro/leaf_spec_phases.bin
__device__ long long g_sst[16 * 64 * 4];
__device__ unsigned g_hwid[16];
#define GPSO_SSTAMP(q, i)                                                                              \
  do {                                                                                                  \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (q) < 64)                      \
      g_sst[((threadIdx.x >> 6) * 64 + (q)) * 4 + (i)] = __builtin_amdgcn_s_memtime();                  \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (q) == 8 && (i) == 0)          \
      g_hwid[threadIdx.x >> 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));           \
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
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  const int dp4 = 3, dp = 12, ns = 2;
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
  float* f16_scal = reinterpret_cast<float*>(lb);
  void* planes = static_cast<char*>(lb) + 256;
  hipMemcpy(dl, linv.data(), linv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, xsp.data(), xsp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, xn.data(), npad * 4, hipMemcpyHostToDevice); hipMemcpy(da, al.data(), npad * 4, hipMemcpyHostToDevice);
  hipMemcpy(dlv, lv.data(), lv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dln, ln.data(), m * 4, hipMemcpyHostToDevice);
  launch_pack_linv_f16<float>(0, dl, npad, npad, f16_scal, planes);
  KernParams kp{0, 1.0, 1e-3, 0.0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    launch_leaf_tiles_bf16<float>(0, ns, planes, dx, dn, da, dlv, dln, pv, pm, npad, dp4, m, kp, nullptr, f16_scal, variant);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel %.3f ms (with stamps), variant %d\n", ms, variant);
  }
  if (variant != 0) return 0;
  std::vector<long long> g(16 * 64 * 4);
  std::vector<unsigned> hw(16);
  hipMemcpyFromSymbol(g.data(), HIP_SYMBOL(g_sst), g.size() * 8);
  hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(g_hwid), hw.size() * 4);
  for (int w = 0; w < 16; ++w) {  // heaviest workgroup: bi = 7 -> 64 k-steps (56 below the diagonal block)
    double seg[3] = {0, 0, 0}, tot = 0; int cnt = 0;
    for (int q = 8; q < 52; ++q) {
      const long long* a = g.data() + (w * 64 + q) * 4;
      for (int i = 0; i < 3; ++i) seg[i] += (double)(a[i + 1] - a[i]);
      tot += (double)(a[4] - a[0]);
      ++cnt;
    }
    // HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13]
    printf("  wave %2d (%s %d, slot %d) simd %u cu %u: issue DMA %.0f | %s %.0f | barrier %.0f | step %.0f\n", w,
           w < 8 ? "apply" : "gener", (w >> 2) & 1, w & 3, (hw[w] >> 4) & 3, (hw[w] >> 8) & 15,
           seg[0] / cnt, w < 8 ? "apply" : "generate", seg[1] / cnt, seg[2] / cnt, tot / cnt);
  }
  return 0;
}
