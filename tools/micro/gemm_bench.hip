// Throughput of the LDS-DMA tile GEMM on plain square products (no triangular trimming):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pygpso_amd/csrc tools/micro/gemm_bench.hip -o tools/micro/gemm_bench.bin
#include "../../pygpso_amd/csrc/fit.hip"
#include <cstdio>
#include <vector>
using namespace gpso;

template <typename T>
static void run(int n, bool a_kc, bool b_kc, double beta, int lower = 0, int kmode = 0, int kk = 0);

template <typename T>
static void run(int n, bool a_kc, bool b_kc, double beta, int lower, int kmode, int kk) {
  T *A, *B, *C;
  const size_t bytes = (size_t)n * n * sizeof(T);
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&C, bytes);
  hipMemset(A, 0, bytes); hipMemset(B, 0, bytes); hipMemset(C, 0, bytes);
  GemmDesc g{};
  g.A = A; g.sai = a_kc ? n : 1; g.sak = a_kc ? 1 : n;
  g.B = B; g.sbk = b_kc ? 1 : n; g.sbj = b_kc ? n : 1;
  g.C = C; g.ldc = n; g.m = n; g.n = n; g.k = n; g.m_last = n; g.nbatch = 1;
  g.alpha = 1.0; g.beta = beta; g.lower_only = lower; g.kmode = kmode;
  if (kk) g.k = kk;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch_gemm<T>(0, g);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) launch_gemm<T>(0, g);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  // flops actually issued: tiles kept by lower_only, k range kept by kmode (128-tile granularity)
  const int nt = n / 128;
  double tiles_k = 0;
  for (int ti = 0; ti < nt; ++ti)
    for (int tj = 0; tj < nt; ++tj) {
      if (lower && tj > ti) continue;
      int klo = 0, khi = g.k;
      if (kmode == 1) klo = 128 * tj;
      if (kmode == 2) klo = 128 * ti;
      if (kmode == 3) khi = std::min(g.k, 128 * (ti + 1));
      tiles_k += std::max(0, khi - klo);
    }
  printf("%s n=%d k=%d A:%s B:%s beta=%g lower=%d kmode=%d  %.3f ms  %.1f TFLOP/s (issued work)\n",
         sizeof(T) == 4 ? "f32" : "f64", n, g.k, a_kc ? "KC" : "RC", b_kc ? "KC" : "RC", beta, lower, kmode,
         ms, 2.0 * 128 * 128 * tiles_k / ms / 1e9);
  hipFree(A); hipFree(B); hipFree(C);
}

int main() {
  run<float>(8192, true, true, 0.0);
  run<float>(8192, false, false, 0.0);
  run<float>(8192, false, false, 0.0, 1, 0);      // lower only, full k
  run<float>(8192, false, false, 0.0, 1, 2);      // K^-1 shape: lower, k >= 128 ti
  run<float>(8192, false, false, 0.0, 0, 2);      // full grid, k >= 128 ti
  run<float>(7680, true, true, 1.0, 1, 0, 256);   // SYRK shape
  run<float>(7680, true, true, 1.0, 0, 0, 256);   // same, full grid
  run<float>(4096, true, false, 0.0, 0, 1);       // TRTRI a shape
  run<float>(4096, true, false, 0.0, 0, 3);       // TRTRI b shape
  run<double>(4096, true, true, 0.0);
  return 0;
}
