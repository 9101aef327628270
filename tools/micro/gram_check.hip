// debug: run gram_kernel<double> for every kernel id on a small random problem and print K[0][0..3]
#include <cstdio>
#include <vector>
#include <random>
#include "../../pygpso_amd/csrc/fit.hip"
namespace gpso {
int ensure_dyn_lds(const void*, int) { return 0; }
void note_launch_error(const char* m) { printf("launch error: %s\n", m); }
}
using namespace gpso;
int main() {
  const int n = 50, npad = 128, d = 2, dp = 4;
  std::mt19937 rng(1);
  std::uniform_real_distribution<double> U(0, 1);
  std::vector<double> x(n * d), ls(48, 0.3535);
  for (auto& v : x) v = U(rng);
  double *dx, *dls, *xs, *xn, *xp, *K;
  hipMalloc(&dx, n * d * 8); hipMalloc(&dls, 48 * 8); hipMalloc(&xs, npad * dp * 8); hipMalloc(&xn, npad * 8);
  hipMalloc(&xp, npad * dp * 8); hipMalloc(&K, npad * npad * 8);
  hipMemcpy(dx, x.data(), n * d * 8, hipMemcpyHostToDevice);
  hipMemcpy(dls, ls.data(), 48 * 8, hipMemcpyHostToDevice);
  launch_scale_x<double>(0, dx, n, npad, d, dp, dls, xs, xn, xp);
  for (int kern = 0; kern < 4; ++kern) {
    KernParams kp{kern, 1.3, 1e-3, 0.0};
    launch_gram<double>(0, xs, xn, n, npad, dp, kp, K);
    std::vector<double> h(npad * npad);
    hipMemcpy(h.data(), K, npad * npad * 8, hipMemcpyDeviceToHost);
    printf("kernel %d: K[0][0]=%g K[1][0]=%g K[1][1]=%g K[49][49]=%g K[49][3]=%g K[50][50]=%g %s\n", kern, h[0], h[npad], h[npad + 1],
           h[49 * npad + 49], h[49 * npad + 3], h[50 * npad + 50], hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
