// Issue / latency probe for the single-wave chains of the fit's diagonal block (chol64_lds, trinv64_lds):
// one workgroup of 4 waves (one per SIMD, like role D), s_memtime around 64-long sequences.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/lat_probe.hip -o tools/micro/lat_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(long long* out, float* sink, float seed) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  long long t[24];
  float a = seed + lane, b = 1.0001f, c = 0.5f, d = seed * 2, e = seed * 3, f = seed * 5;
  double da = a, db = 1.0001, dc = 0.5;
  int n = 0;
  lds[tid] = a;
  __syncthreads();
#define STAMP() t[n++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0)
  STAMP();
  REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)  // dependent f32 fma
  STAMP();
  REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)  // 4 independent chains
  STAMP();
  REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(da) : "v"(db), "v"(dc));)  // dependent f64 fma
  STAMP();
  REP64(asm volatile("v_rsq_f32 %0, %0" : "+v"(a));)  // dependent rsq f32
  STAMP();
  REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(da));)  // dependent rsq f64
  STAMP();
  {
    int s;
    REP64(asm volatile("v_readlane_b32 %1, %0, 3\n v_mul_f32 %0, %1, %0" : "+v"(a), "=s"(s));)  // readlane -> VALU hop
  }
  STAMP();
  {
    int s;
    REP64(asm volatile("v_readlane_b32 %1, %0, 3\n s_nop 0" : "+v"(a), "=s"(s));)  // readlane alone (+ nop)
  }
  STAMP();
  REP64(asm volatile("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(tid * 4));)  // LDS round trip
  STAMP();
  REP64(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"((int)(a) & 1020));)  // dependent LDS read
  STAMP();
  REP16(__syncthreads();)
  STAMP();
  {
    f64x4 acc{0, 0, 0, 0};
    REP16(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, acc, 0, 0, 0);)  // dependent f64 MFMA
    da += acc[0];
  }
  STAMP();
  {
    f32x4 acc{0, 0, 0, 0};
    REP16(acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);)  // dependent f32 MFMA
    a += acc[0];
  }
  STAMP();
  REP64(asm volatile("s_nop 0");)
  STAMP();
  REP64(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(da) : "v"(a));)
  STAMP();
  if (tid == 0)
    for (int i = 0; i < n; ++i) out[i] = t[i];
  sink[tid] = a + d + e + f + (float)da;
}

int main() {
  long long* out; float* sink;
  hipMalloc(&out, 24 * 8); hipMalloc(&sink, 256 * 4);
  for (int it = 0; it < 3; ++it) probe<<<1, 256>>>(out, sink, 1.5f);
  hipDeviceSynchronize();
  long long h[24];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"64 dependent v_fma_f32", "64 v_fma_f32 in 4 independent chains", "64 dependent v_fma_f64",
                         "64 dependent v_rsq_f32", "64 dependent v_rsq_f64", "64 x (v_readlane -> v_mul reading the SGPR)",
                         "64 x (v_readlane, s_nop)", "64 x LDS write -> read -> wait", "64 dependent LDS reads",
                         "16 x __syncthreads (4 waves)", "16 dependent v_mfma_f64_16x16x4", "16 dependent v_mfma_f32_16x16x4",
                         "64 s_nop 0", "64 v_cvt_f64_f32 (independent)"};
  // s_memtime counts the 100 MHz constant clock on this part: report raw ticks and ticks per item
  for (int i = 0; i < 14; ++i) {
    const int items = (i == 9 || i == 10 || i == 11) ? 16 : 64;
    printf("%-48s %6lld ticks  %.2f per item\n", names[i], h[i + 1] - h[i], (double)(h[i + 1] - h[i]) / items);
  }
  return 0;
}
