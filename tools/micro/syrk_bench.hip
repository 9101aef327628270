// split-bf16 SYRK (fit.hip: syrk_bf16_kernel) against the f32 tile GEMM on the trailing-update shape of the
// two-level Cholesky: C (m x m, lower 128-tiles) -= A A^T, A = m x k.  Checks the result and times both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pygpso_amd/csrc tools/micro/syrk_bench.hip -o tools/micro/syrk_bench.bin
__device__ long long g_gst[4 * 40 * 4];
__device__ int g_sel[2];
#ifdef GPSO_BENCH_STAMPS
#define GPSO_GSTAMP(st, i) do { if ((threadIdx.x & 63) == 0 && ((int)blockIdx.x == g_sel[0] || (int)blockIdx.x == g_sel[1]) && st < 40 && i < 3) g_gst[(((int)blockIdx.x == g_sel[1]) * 2 + (threadIdx.x >> 7)) * 160 + st * 4 + i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
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
  const int64_t m = (argc > 1) ? atoll(argv[1]) : 6144, k = (argc > 2) ? atoll(argv[2]) : 1024, ld = m + k;
  std::vector<float> hA((size_t)m * ld), hC((size_t)m * ld);
  unsigned s = 7; auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24) - 0.5f; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hC) v = 10.0f * rnd();
  float *A, *C1, *C2; unsigned short* P;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&C1, hC.size() * 4); hipMalloc(&C2, hC.size() * 4); hipMalloc(&P, 3 * hA.size() * 2);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((m * k / 2 + 255) / 256)), dim3(256), 0, 0, A, ld, m, k, P, (int64_t)hA.size());
  GemmDesc u{};
  u.A = A; u.sai = ld; u.sak = 1; u.B = A; u.sbk = 1; u.sbj = ld; u.ldc = ld;
  u.m = (int)m; u.n = (int)m; u.k = (int)k; u.m_last = (int)m; u.nbatch = 1; u.alpha = -1.0; u.beta = 1.0; u.lower_only = 1;
  GemmBf16Desc b{}; b.A = b.B = Bf16Planes{P, (int64_t)hA.size(), (int)(k / 32)}; b.C = C2; b.ldc = ld; b.m = b.n = (int)m; b.k = (int)k; b.alpha = -1.0f; b.beta = 1; b.lower_only = 1; b.nbatch = 1;
  int sel[2] = {17, 100};  // two of the persistent workgroups
#ifdef GPSO_BENCH_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(g_sel), sel, sizeof(sel));
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gf = (double)m * m * k / 1e9;
  double last_us = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(C1, hC.data(), hC.size() * 4, hipMemcpyHostToDevice); hipMemcpy(C2, hC.data(), hC.size() * 4, hipMemcpyHostToDevice);
    float t1, t2;
    u.C = C1;
    hipEventRecord(e0, 0); launch_gemm<float>(0, u); hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&t1, e0, e1);
    hipEventRecord(e0, 0); launch_gemm_bf16(0, b); hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&t2, e0, e1);
    last_us = t2 * 1e3;
    std::vector<float> r1(hC.size()), r2(hC.size());
    hipMemcpy(r1.data(), C1, r1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), C2, r2.size() * 4, hipMemcpyDeviceToHost);
    double maxd = 0, maxv = 0;
    for (int64_t i = 0; i < m; ++i)
      for (int64_t j = 0; j <= i; ++j) {
        maxd = std::max(maxd, (double)std::fabs(r1[i * ld + j] - r2[i * ld + j]));
        maxv = std::max(maxv, (double)std::fabs(r1[i * ld + j]));
      }
    printf("m %lld k %lld: f32 gemm %.0f us (%.0f TF/s) | split-bf16 syrk %.0f us (%.0f TF/s f32-equivalent) | max |diff| %.3g of max |C| %.3g\n",
           (long long)m, (long long)k, t1 * 1e3, gf / t1, t2 * 1e3, gf / t2, maxd, maxv);
  }
#ifdef GPSO_BENCH_STAMPS
  long long gs[641] = {0}; hipMemcpyFromSymbol(gs, HIP_SYMBOL(g_gst), 640 * 8);
  for (int w = 0; w < 4; ++w) {
    printf("workgroup %d wave %d, per tile [decode + first two steps .. k-loop | wait + C + stores]:", w < 2 ? sel[0] : sel[1], (w & 1) * 2);
    const long long* q = gs + w * 160;
    int nt = 0;
    for (int t = 0; t < 39 && q[4 * t + 2] != 0; ++t, ++nt) printf(" %lld|%lld", q[4 * t + 1] - q[4 * t], q[4 * t + 2] - q[4 * t + 1]);
    if (nt > 0) printf("\n   %d tiles in %lld clocks (kernel %.0f us)\n", nt, q[4 * (nt - 1) + 2] - q[0], last_us);
  }
#endif
  return 0;
}
