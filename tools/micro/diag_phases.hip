// Phase timing of the 64x64 diagonal-block kernel (s_memtime stamps); build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pygpso_amd/csrc tools/micro/diag_phases.hip -o /tmp/diag_phases
__device__ long long g_stamps[16];
#define GPSO_STAMP(i) do { if (threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#include "../../pygpso_amd/csrc/fit.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace gpso;

template <typename T>
__global__ __launch_bounds__(256) void diag_phases_kernel(T* K, T* linv, int64_t ld, int64_t n,
                                                          double* logdet_part, int* info,
                                                          long long* stamps) {
  __shared__ double Ls[kFitBlock * kDS];
  __shared__ double Xs[kFitBlock * kDS];
  __shared__ double Ts[3 * kPB * 17];
  __shared__ double inv_diag[kFitBlock];
  const int tid = threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  T* A = K;
  for (int e = tid; e < kFitBlock * kFitBlock; e += 256) {
    const int r = e >> 6, c = e & 63;
    Ls[r * kDS + c] = (c <= r) ? (double)A[(int64_t)r * ld + c] : 0.0;
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  chol64_lds(Ls, inv_diag, 0, n, info);
  long long t2 = __builtin_amdgcn_s_memtime();
  if (tid < kFitBlock) {
    double lg = log(Ls[tid * kDS + tid]);
    lg = wave_sum(lg);
    if (tid == 0) logdet_part[0] = lg;
  }
  long long t3 = __builtin_amdgcn_s_memtime();
  trinv64_lds(Ls, inv_diag, Xs, Ts);
  long long t4 = __builtin_amdgcn_s_memtime();
  T* Xo = linv;
  for (int e = tid; e < kFitBlock * kFitBlock; e += 256) {
    const int r = e >> 6, c = e & 63;
    A[(int64_t)r * ld + c] = (T)Ls[r * kDS + c];
    Xo[(int64_t)r * ld + c] = (T)Xs[r * kDS + c];
  }
  __syncthreads();
  long long t5 = __builtin_amdgcn_s_memtime();
  if (tid == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = t4 - t3; stamps[4] = t5 - t4; }
}

int main() {
  const int ld = 2048;
  std::vector<float> h((size_t)64 * ld, 0.f);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd;
  std::vector<float> B(64 * 64);
  for (auto& v : B) v = nd(rng);
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < 64; ++j) {
      float s = (i == j) ? 64.f : 0.f;
      for (int k = 0; k < 64; ++k) s += B[i * 64 + k] * B[j * 64 + k];
      h[(size_t)i * ld + j] = s;
    }
  float *K, *X; double* lg; int* info; long long* st;
  hipMalloc(&K, h.size() * 4); hipMalloc(&X, h.size() * 4); hipMalloc(&lg, 8 * 64); hipMalloc(&info, 4); hipMalloc(&st, 8 * 8);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(K, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((diag_phases_kernel<float>), dim3(1), dim3(256), 0, 0, K, X, (int64_t)ld, (int64_t)64, lg, info, st);
    hipDeviceSynchronize();
    long long s[5]; hipMemcpy(s, st, 40, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx9
    long long g[16]; hipMemcpyFromSymbol(g, HIP_SYMBOL(g_stamps), sizeof(g));
    printf("chol panels/updates:"); for (int i = 1; i < 8; ++i) printf(" %lld", g[i] - g[i - 1]); printf("\n");
    printf("load %lld chol %lld logdet %lld trinv %lld store %lld   (shader clocks)\n", s[0], s[1], s[2], s[3], s[4]);
  }
  return 0;
}
