// Phase timing of the Cholesky step kernel's diagonal role (s_memtime stamps from inside the real
// launch_potrf at N = 2048) plus the wall time of the whole factorisation.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pygpso_amd/csrc tools/micro/diag_phases.hip -o tools/micro/diag_phases.bin
__device__ long long g_stamps[24];
#define GPSO_STAMP(i) do { if (threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#include "../../pygpso_amd/csrc/fit.hip"
#include <cstdio>
#include <cmath>
#include <vector>
namespace gpso {
int ensure_dyn_lds(const void* fn, int bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return 0;
}
void note_launch_error(const char* m) { fprintf(stderr, "launch error: %s\n", m); }
}  // namespace gpso
using namespace gpso;

int main(int argc, char** argv) {
  const int64_t n = (argc > 1) ? atoll(argv[1]) : 2048;
  std::vector<float> h((size_t)n * n);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j)
      h[(size_t)i * n + j] = (float)(std::exp(-std::fabs((double)(i - j)) / 40.0) + (i == j ? 1e-2 : 0.0));
  float *K, *Lf, *X, *Wk; double* dg; int* info;
  hipMalloc(&K, h.size() * 4); hipMalloc(&Lf, h.size() * 4); hipMalloc(&X, h.size() * 4); hipMalloc(&Wk, h.size() * 4);
  hipMalloc(&dg, 8 * n); hipMalloc(&info, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipMemcpy(K, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(X, 0, h.size() * 4);
    int imax = 2147483647; hipMemcpy(info, &imax, 4, hipMemcpyHostToDevice);
    hipEventRecord(e0, 0);
    if (!(1 & launch_potrf<float>(0, K, Lf, X, Wk, (float*)nullptr, n, n, dg, info, (argc > 2) ? atoll(argv[2]) : -1, (const FitPlanes*)nullptr))) launch_trtri<float>(0, Lf, X, Wk, n, fit_outer_panel(n));
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long g[24]; hipMemcpyFromSymbol(g, HIP_SYMBOL(g_stamps), sizeof(g));
    int inf; hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost);
    printf("potrf+inverse %.1f us (%lld steps, info %d) | last step, shader clocks: load %lld L10 %lld S %lld | panels/updates",
           ms * 1e3, (long long)(n / 64), inf == imax ? -1 : inf, g[17] - g[14], g[18] - g[17], g[15] - g[18]);
    for (int i = 1; i < 7; ++i) printf(" %lld", g[i] - g[i - 1]);
    printf(" | last panel+check %lld | trinv:", g[7 + 1] - g[6] - (g[8] - g[7]));
    // (the step kernel's diagonal role forms part of the inverse beside the panels -- chol64_lds<.., EARLY>: stamps 10 / 11 do not exist)
    printf(" %lld %lld %lld %lld", g[8] - g[7], g[9] - g[8], g[12] - g[9], g[13] - g[12]);
    printf(" | store %lld | total %lld\n", g[16] - g[13], g[16] - g[14]);
  }
  return 0;
}
