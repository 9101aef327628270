// Can the Cholesky chain (potrf_block: one potrf_step_kernel launch per 64 columns, 73 KB-LDS workgroups)
// run BESIDE a bulk GEMM on another stream?  A gemm128 workgroup takes 64 KB of LDS and two share a CU, so a
// chain workgroup finds no room; launched with its LDS request inflated to 82 KB ("thin") only one GEMM
// workgroup fits a CU and 78 KB stay free for a chain workgroup.  Measures chain and GEMM alone and together.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pygpso_amd/csrc tools/micro/overlap_probe.hip -o tools/micro/overlap_probe.bin
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

static void launch_syrk(hipStream_t st, const GemmDesc& g, int lds_bytes) {
  const int64_t nti = g.m / 128;
  const dim3 grid((unsigned)(nti * (nti + 1) / 2), 1, 1);
  hipLaunchKernelGGL((gemm128_kernel<float, 128, true, true>), grid, dim3(256), lds_bytes, st, g);
}

int main(int argc, char** argv) {
  const int64_t n = 8192, w = (argc > 1) ? atoll(argv[1]) : 1024, m2 = (argc > 2) ? atoll(argv[2]) : 6144;
  std::vector<float> h((size_t)n * n, 0.f);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = std::max<int64_t>(0, i - 200); j <= i; ++j)
      h[(size_t)i * n + j] = (float)(std::exp(-std::fabs((double)(i - j)) / 40.0) + (i == j ? 1e-1 : 0.0));
  float *K, *Lf, *X, *Wk, *C; double* dg; int* info;
  hipMalloc(&K, h.size() * 4); hipMalloc(&Lf, h.size() * 4); hipMalloc(&X, h.size() * 4); hipMalloc(&Wk, h.size() * 4);
  hipMalloc(&C, h.size() * 4); hipMalloc(&dg, 8 * n); hipMalloc(&info, 4);
  hipMemset(C, 0, h.size() * 4);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipStream_t sp; int plo, phi; hipDeviceGetStreamPriorityRange(&plo, &phi); hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, phi);
  hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
  constexpr int kLds = G128<float, 128>::kNbuf * G128<float, 128>::kBufBytes;
  (void)hipFuncSetAttribute((const void*)gemm128_kernel<float, 128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  GemmDesc u{};  // C22 -= A A^T, A = m2 x w (k-contiguous), lower tiles
  u.A = Lf; u.sai = n; u.sak = 1; u.B = Lf; u.sbk = 1; u.sbj = n; u.C = C; u.ldc = n;
  u.m = (int)m2; u.n = (int)m2; u.k = (int)w; u.m_last = (int)m2; u.nbatch = 1; u.alpha = -1.0; u.beta = 1.0; u.lower_only = 1;
  auto reset = [&] {
    hipMemcpy(K, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(X, 0, h.size() * 4);
    int imax = 2147483647; hipMemcpy(info, &imax, 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
  };
  auto chain = [&](hipStream_t s) { potrf_block<float>(s, K, Lf, X, Wk, (float*)nullptr, n, (int)(w / 64), 0, n, dg, info); };
  const double gf = (double)m2 * m2 * w / 1e9;  // lower half of 2 m2^2 w
  printf("chain: %lld-wide diagonal block (%lld steps); GEMM: lower tiles of %lld x %lld x %lld = %.1f GFLOP; gemm LDS %d B\n",
         (long long)w, (long long)(w / 64), (long long)m2, (long long)m2, (long long)w, gf, kLds);
  for (int rep = 0; rep < 3; ++rep) {
    float tc, tg, tt, tcc, tgc, tcf, tgf;
    reset(); hipEventRecord(b0, sb); chain(sb); hipEventRecord(b1, sb); hipDeviceSynchronize(); hipEventElapsedTime(&tc, b0, b1);
    hipEventRecord(a0, sa); launch_syrk(sa, u, kLds); hipEventRecord(a1, sa); hipDeviceSynchronize(); hipEventElapsedTime(&tg, a0, a1);
    hipEventRecord(a0, sa); launch_syrk(sa, u, 82 * 1024); hipEventRecord(a1, sa); hipDeviceSynchronize(); hipEventElapsedTime(&tt, a0, a1);
    // thin GEMM beside the chain
    reset();
    hipEventRecord(a0, sa); launch_syrk(sa, u, 82 * 1024); hipEventRecord(a1, sa);
    hipEventRecord(b0, sb); chain(sb); hipEventRecord(b1, sb);
    hipDeviceSynchronize(); hipEventElapsedTime(&tgc, a0, a1); hipEventElapsedTime(&tcc, b0, b1);
    // full GEMM beside the chain
    reset();
    hipEventRecord(a0, sa); launch_syrk(sa, u, kLds); hipEventRecord(a1, sa);
    hipEventRecord(b0, sb); chain(sb); hipEventRecord(b1, sb);
    hipDeviceSynchronize(); hipEventElapsedTime(&tgf, a0, a1); hipEventElapsedTime(&tcf, b0, b1);
    // chain launched FIRST, thin GEMM second
    float tg2, tc2, tg3, tc3;
    reset();
    hipEventRecord(b0, sb); chain(sb); hipEventRecord(b1, sb);
    hipEventRecord(a0, sa); launch_syrk(sa, u, 82 * 1024); hipEventRecord(a1, sa);
    hipDeviceSynchronize(); hipEventElapsedTime(&tg2, a0, a1); hipEventElapsedTime(&tc2, b0, b1);
    // chain on a high-priority stream, launched first
    reset();
    hipEventRecord(b0, sp); chain(sp); hipEventRecord(b1, sp);
    hipEventRecord(a0, sa); launch_syrk(sa, u, 82 * 1024); hipEventRecord(a1, sa);
    hipDeviceSynchronize(); hipEventElapsedTime(&tg3, a0, a1); hipEventElapsedTime(&tc3, b0, b1);
    printf("  chain first: gemm %.0f chain %.0f | chain first, high priority: gemm %.0f chain %.0f\n", tg2 * 1e3, tc2 * 1e3, tg3 * 1e3, tc3 * 1e3);
    int inf; hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost);
    printf("alone: chain %.0f us, gemm %.0f us (%.0f TF/s), thin gemm %.0f us (%.0f TF/s) | thin+chain: gemm %.0f chain %.0f | full+chain: gemm %.0f chain %.0f | info %d\n",
           tc * 1e3, tg * 1e3, gf / tg, tt * 1e3, gf / tt, tgc * 1e3, tcc * 1e3, tgf * 1e3, tcf * 1e3, inf == 2147483647 ? -1 : inf);
  }
  return 0;
}
