// Phase timing of the fused small fit (small_fit_kernel, N <= 128): s_memtime stamps from inside the real
// kernel plus its HIP-event time.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pygpso_amd/csrc tools/micro/small_phases.hip -o tools/micro/small_phases.bin
//   tools/micro/small_phases.bin [N] [D]
__device__ long long g_sst[16];
#define GPSO_SSTAMP(i) do { if (threadIdx.x == 0) g_sst[i] = __builtin_amdgcn_s_memtime(); } while (0)
#include "../../pygpso_amd/csrc/fit.hip"
#include <cstdio>
#include <cmath>
#include <string>
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
  const int n = (argc > 1) ? atoi(argv[1]) : 52, d = (argc > 2) ? atoi(argv[2]) : 2, dp = (d + 3) / 4 * 4;
  std::vector<double> X((size_t)n * d), y(n), ls(48, 0.25 * std::sqrt((double)d));
  unsigned s = 12345;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24); };
  for (auto& v : X) v = rnd();
  for (int i = 0; i < n; ++i) y[i] = std::sin(3.0 * X[(size_t)i * d]) + 0.1 * rnd();
  double *x64, *y64, *lsd, *xs64, *xn, *xsp, *dg, *kd, *scal;
  void *Lf, *linv, *kinv, *white, *alf, *alp, *linvp;
  hipMalloc(&x64, X.size() * 8); hipMalloc(&y64, n * 8); hipMalloc(&lsd, 48 * 8);
  hipMalloc(&xs64, 128 * dp * 8); hipMalloc(&xn, 128 * 8); hipMalloc(&xsp, 128 * dp * 8);
  hipMalloc(&dg, 128 * 8); hipMalloc(&kd, 128 * 8); hipMalloc(&scal, 128 * 8);
  hipMalloc(&Lf, 128 * 128 * 8); hipMalloc(&linv, 128 * 128 * 8); hipMalloc(&kinv, 128 * 128 * 8);
  hipMalloc(&white, 128 * 8); hipMalloc(&alf, 128 * 8); hipMalloc(&alp, 128 * 8); hipMalloc(&linvp, 36 * 256 * 8);
  hipMemcpy(x64, X.data(), X.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(y64, y.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(lsd, ls.data(), 48 * 8, hipMemcpyHostToDevice);
  SmallFitArgs a{};
  a.x64 = x64; a.y64 = y64; a.hyper = scal + 64; for (int k = 0; k < 48; ++k) a.ls[k] = ls[k]; a.n = n; a.d = d; a.dp = dp; a.kernel = 0; a.n_ls = 1; a.want_grad = 1;
  a.variance = 1.0; a.noise = 1e-3; a.mean_c = 0.1;
  a.xs64 = xs64; a.xnorm64 = xn; a.xs_p64 = xsp; a.Lf = Lf; a.linv = linv; a.kinv = kinv; a.white = white;
  a.alpha_f = alf; a.alpha_p = alp; a.linv_p = linvp; a.diag64 = dg; a.kinv_diag = kd; a.scal = scal;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // stamp i is taken at the END of phase names[i - 1] (stamps 5..7 only exist for N > 64)
  const char* names[] = {"scale", "gram+zero", "chol00", "trinv00", "L10+S", "chol11", "trinv11", "X10", "store+pack", "alpha+nlml", "kinv+grad"};
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    launch_small_fit<double, double>(0, a);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long g[16]; hipMemcpyFromSymbol(g, HIP_SYMBOL(g_sst), sizeof(g));
    double h[16]; hipMemcpy(h, scal, 16 * 8, hipMemcpyDeviceToHost);
    printf("n=%d d=%d: %.1f us (nlml %.6f) | clocks:", n, d, ms * 1e3, h[0]);
    const bool two = n > 64;
    int prev = 0;
    for (int i = 1; i < 12; ++i) {
      if (!two && i >= 5 && i <= 7) continue;
      printf(" %s %lld", names[i - 1], g[i] - g[prev]);
      prev = i;
    }
    printf(" | total %lld\n", g[11] - g[0]);
  }
  return 0;
}
