// debug: evaluate the lean double kernel functions on a few arguments
#include <cstdio>
#include "../../pygpso_amd/csrc/common.hpp"
using namespace gpso;
__global__ void k(const double* r2, int n, double* out, const int* kid) {
  int i = threadIdx.x;
  if (i >= n) return;
  for (int kern = 0; kern < 4; ++kern) out[kern * n + i] = kern_from_r2_lean(kid[kern], r2[i], 1.3);
  out[4 * n + i] = sqrt_lean(fmax(r2[i], 1e-36));
  out[5 * n + i] = exp_lean(-sqrt_lean(fmax(r2[i], 1e-36)));
}
int main() {
  double h[8] = {0.0, -1e-16, 1e-16, 1e-36, 0.5, 4.0, 30.0, 1e-10};
  double *d, *o, ho[48]; int* kd; int hk[4] = {0, 1, 2, 3}; hipMalloc(&kd, 16); hipMemcpy(kd, hk, 16, hipMemcpyHostToDevice);
  hipMalloc(&d, 64); hipMalloc(&o, 48 * 8);
  hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, 8, o, kd);
  hipMemcpy(ho, o, 48 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("r2=%g: m52 %g m32 %g m12 %g se %g sqrt %g exp(-sqrt) %g\n", h[i], ho[i], ho[8 + i], ho[16 + i], ho[24 + i], ho[32 + i], ho[40 + i]);
  return 0;
}
