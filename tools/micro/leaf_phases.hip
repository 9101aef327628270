// Where does a wave of leaf_tiles_v2_kernel spend a k-step?  s_memtime stamps of wave 0 of the heaviest
// workgroup (blockIdx = (0, 0)) at six points of every k-step, inside the real kernel at the C3 shape:
//   0 step begin | 1 LDS-DMA of the next k-tile issued | 2 generation MFMAs issued | 3 apply loop (+ map) done |
//   4 before the barrier | 5 after the barrier
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I pygpso_amd/csrc tools/micro/leaf_phases.hip -o tools/micro/leaf_phases.bin
__device__ long long g_pst[256 * 8];
#define GPSO_PSTAMP(kt, i)                                                                      \
  do {                                                                                          \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (kt) < 256)                   \
      g_pst[(kt) * 8 + (i)] = __builtin_amdgcn_s_memtime();                                     \
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
  const int dp4 = 3, dp = 12;
  const size_t tiles = packed_linv_elems(npad);
  std::vector<float> linv(tiles), xsp(npad * dp), xn(npad), al(npad), lv(m * dp), ln(m);
  unsigned s = 1; auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24); };
  for (auto& v : linv) v = 0.01f * (rnd() - 0.5f);
  for (auto& v : xsp) v = rnd();
  for (auto& v : xn) v = 3.0f + rnd();
  for (auto& v : al) v = rnd() - 0.5f;
  for (auto& v : lv) v = rnd();
  for (auto& v : ln) v = 3.0f + rnd();
  float *dl, *dx, *dn, *da, *dlv, *dln; double *pv, *pm;
  hipMalloc(&dl, tiles * 4); hipMalloc(&dx, xsp.size() * 4); hipMalloc(&dn, npad * 4); hipMalloc(&da, npad * 4);
  hipMalloc(&dlv, lv.size() * 4); hipMalloc(&dln, m * 4); hipMalloc(&pv, 8 * m * 8); hipMalloc(&pm, 8 * m * 8);
  hipMemcpy(dl, linv.data(), tiles * 4, hipMemcpyHostToDevice); hipMemcpy(dx, xsp.data(), xsp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dn, xn.data(), npad * 4, hipMemcpyHostToDevice); hipMemcpy(da, al.data(), npad * 4, hipMemcpyHostToDevice);
  hipMemcpy(dlv, lv.data(), lv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dln, ln.data(), m * 4, hipMemcpyHostToDevice);
  KernParams kp{0, 1.0, 1e-3, 0.0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    launch_leaf_tiles<float, float>(0, dl, dx, dn, da, dlv, dln, pv, pm, npad, dp4, m, kp, nullptr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> g(256 * 8);
    hipMemcpyFromSymbol(g.data(), HIP_SYMBOL(g_pst), g.size() * 8);
    // heaviest workgroup: bi = 7 -> 128 k-steps (112 full + 16 diagonal)
    double seg[5] = {0, 0, 0, 0, 0}, tot = 0; int cnt = 0;
    for (int kt = 8; kt < 104; ++kt) {  // steady state, off-diagonal steps
      for (int i = 0; i < 5; ++i) seg[i] += (double)(g[kt * 8 + i + 1] - g[kt * 8 + i]);
      tot += (double)(g[(kt + 1) * 8] - g[kt * 8]);
      ++cnt;
    }
    printf("kernel %.3f ms (with stamps) | per k-step of wave 0, clocks: issue DMA %.0f | generation %.0f | apply+map %.0f | tail %.0f | barrier %.0f | step %.0f\n",
           ms, seg[0] / cnt, seg[1] / cnt, seg[2] / cnt, seg[3] / cnt, seg[4] / cnt, tot / cnt);
    if (rep == 2) {
      printf("steps 40..47 (clocks since step begin):");
      for (int kt = 40; kt < 48; ++kt) {
        printf("\n  kt %d:", kt);
        for (int i = 1; i < 6; ++i) printf(" %lld", g[kt * 8 + i] - g[kt * 8]);
      }
      printf("\n");
    }
  }
  return 0;
}
