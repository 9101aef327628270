// Does the Cholesky chain of a diagonal block (potrf_block: one potrf_step_kernel launch per 64 columns) run at
// full speed on another stream BESIDE the persistent split-bf16 GEMM when that GEMM leaves R compute units
// unused?  Times chain and GEMM alone and together for several R, and a one-workgroup spin kernel (72 KB of
// LDS, ~100 us of dependent FMAs) in place of the chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pygpso_amd/csrc tools/micro/lookahead_probe.hip -o tools/micro/lookahead_probe.bin
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

__global__ __launch_bounds__(256) void spin_kernel(float* out, int iters) {
  extern __shared__ float sm[];
  float x = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;
  sm[threadIdx.x] = x;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = sm[255 - threadIdx.x];
}

int main(int argc, char** argv) {
  const int64_t n = 8192, w = 1024, m = (argc > 1) ? atoll(argv[1]) : 6144, ld = n;
  std::vector<float> h((size_t)n * n, 0.f);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = std::max<int64_t>(0, i - 200); j <= i; ++j)
      h[(size_t)i * n + j] = (float)(std::exp(-std::fabs((double)(i - j)) / 40.0) + (i == j ? 1e-1 : 0.0));
  float *K, *Lf, *X, *Wk, *C, *A, *so; double* dg; int* info; unsigned short* P;
  hipMalloc(&K, h.size() * 4); hipMalloc(&Lf, h.size() * 4); hipMalloc(&X, h.size() * 4); hipMalloc(&Wk, h.size() * 4);
  hipMalloc(&C, h.size() * 4); hipMalloc(&A, h.size() * 4); hipMalloc(&dg, 8 * n); hipMalloc(&info, 4); hipMalloc(&so, 4096 * 4);
  hipMalloc(&P, 3 * h.size() * 2);
  hipMemset(C, 0, h.size() * 4);
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((m * w / 2 + 255) / 256)), dim3(256), 0, 0, A, ld, m, w, P, (int64_t)h.size());
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
  GemmBf16Desc g{}; g.A = g.B = Bf16Planes{P, (int64_t)h.size(), (int)(w / 32)}; g.C = C; g.ldc = ld; g.m = g.n = (int)m; g.k = (int)w;
  g.alpha = -1.0f; g.beta = 1; g.lower_only = 1; g.nbatch = 1;
  (void)hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  auto reset = [&] {
    hipMemcpy(K, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(X, 0, h.size() * 4);
    int imax = 2147483647; hipMemcpy(info, &imax, 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
  };
  auto chain = [&](hipStream_t s) { potrf_block<float>(s, K, Lf, X, Wk, (float*)nullptr, n, (int)(w / 64), 0, n, dg, info); };
  auto spin = [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), 72 * 1024, s, so, 60000); };
  for (int rep = 0; rep < 2; ++rep) {
    float tc, ts, tg;
    reset(); hipEventRecord(b0, sb); chain(sb); hipEventRecord(b1, sb); hipDeviceSynchronize(); hipEventElapsedTime(&tc, b0, b1);
    hipEventRecord(b0, sb); spin(sb); hipEventRecord(b1, sb); hipDeviceSynchronize(); hipEventElapsedTime(&ts, b0, b1);
    hipEventRecord(a0, sa); launch_gemm_bf16(sa, g, 0); hipEventRecord(a1, sa); hipDeviceSynchronize(); hipEventElapsedTime(&tg, a0, a1);
    printf("alone: chain %.0f us | spin %.0f us | gemm %.0f us\n", tc * 1e3, ts * 1e3, tg * 1e3);
    for (int R : {0, 8, 16, 32, 64}) {
      float tg1, tc1, tg2, ts2;
      reset();
      hipEventRecord(a0, sa); launch_gemm_bf16(sa, g, R); hipEventRecord(a1, sa);
      hipEventRecord(b0, sb); chain(sb); hipEventRecord(b1, sb);
      hipDeviceSynchronize(); hipEventElapsedTime(&tg1, a0, a1); hipEventElapsedTime(&tc1, b0, b1);
      hipEventRecord(a0, sa); launch_gemm_bf16(sa, g, R); hipEventRecord(a1, sa);
      hipEventRecord(b0, sb); spin(sb); hipEventRecord(b1, sb);
      hipDeviceSynchronize(); hipEventElapsedTime(&tg2, a0, a1); hipEventElapsedTime(&ts2, b0, b1);
      printf("  R = %2d: gemm %.0f beside chain %.0f | gemm %.0f beside spin %.0f\n", R, tg1 * 1e3, tc1 * 1e3, tg2 * 1e3, ts2 * 1e3);
    }
  }
  return 0;
}
