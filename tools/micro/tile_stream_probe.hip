// How fast can workgroups stream 64-row tiles of a row-major float matrix (leading dimension N, a power of two) when a tile row
// is a 256-byte run (64 columns: gpso_append's passes) or a 1 KB run (256 columns)?  Reads only: every thread sums what it
// loads, two tiles in flight per workgroup, lower triangle only, the same 8-chunk grid as append_pass_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tile_stream_probe tools/micro/tile_stream_probe.hip && /tmp/tile_stream_probe 16384
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W /* tile width in floats: 64 or 256 */>
__global__ __launch_bounds__(256) void stream_rows(const float* __restrict__ a, int64_t n, int ct, float* __restrict__ out) {
  // workgroup (tb, q): row block tb (64 rows), its tiles [q ct, min((q + 1) ct, tiles of the triangle)) of width W
  const int tb = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
  const int ntile_row = (int)((tb * 64 + 64 + W - 1) / W);  // tiles that touch the lower triangle of this row block
  const int lo = q * ct, hi = min((q + 1) * ct, ntile_row);
  if (lo >= hi) return;
  constexpr int VPR = W / 4;         // 16-byte loads per tile row
  constexpr int RPP = 256 / VPR;     // rows per pass (VPR <= 256)
  constexpr int NP = 64 / RPP;
  const int vrow = tid / VPR, vcol = (tid % VPR) * 4;
  f4 v[2][NP];
  float s = 0.f;
  auto fetch = [&](int o, f4* vv) {
#pragma unroll
    for (int p = 0; p < NP; ++p) vv[p] = *reinterpret_cast<const f4*>(a + ((int64_t)tb * 64 + p * RPP + vrow) * n + (int64_t)o * W + vcol);
  };
  fetch(lo, v[0]);
  if (lo + 1 < hi) fetch(lo + 1, v[1]);
  for (int o = lo; o < hi; o += 2) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (o + k < hi) {
#pragma unroll
        for (int p = 0; p < NP; ++p) s += v[k][p][0] + v[k][p][1] + v[k][p][2] + v[k][p][3];
        if (o + k + 2 < hi) fetch(o + k + 2, v[k]);
      }
  }
  if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 16384;
  float *a, *out;
  hipMalloc(&a, n * n * 4);
  hipMalloc(&out, 4);
  hipMemset(a, 0, n * n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nrb = (int)(n / 64);
  auto run = [&](int w) {
    const int tiles = (int)(n / w), ct = (tiles + 7) / 8, nq = (tiles + ct - 1) / ct;
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      hipEventRecord(e0);
      if (w == 64) hipLaunchKernelGGL(stream_rows<64>, dim3(nrb, nq), dim3(256), 0, 0, a, n, ct, out);
      else hipLaunchKernelGGL(stream_rows<256>, dim3(nrb, nq), dim3(256), 0, 0, a, n, ct, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    // bytes: the tiles that touch the lower triangle
    double bytes = 0;
    for (int tb = 0; tb < nrb; ++tb) bytes += (double)((tb * 64 + 64 + w - 1) / w) * w * 64 * 4;
    printf("N %lld  tile 64 x %3d (%4d-byte runs): %.1f us  %.2f TB/s\n", (long long)n, w, w * 4, best * 1e3, bytes / best / 1e9);
  };
  run(64);
  run(256);
  run(64);
  run(256);
  return 0;
}
