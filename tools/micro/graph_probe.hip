// Does a hipGraph shorten a short, synchronous launch sequence?  The shape of a small best-UCB call (section 4.4 of DESIGN.md):
// three dependent kernels of ~5 / ~15 / ~5 us, the last one writing pinned host memory, the host waiting for it -- once per
// call.  Variants: (a) three stream launches + event spin (what the library does), (b) one hipGraphLaunch of a captured
// graph + the same wait, (c) the graph with its kernel parameters updated before every launch
// (hipGraphExecKernelNodeSetParams x 3: what a call with new arguments would need).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/graph_probe.hip -o tools/micro/graph_probe.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void spin_kernel(float* buf, int iters, float seed) {
  float v = seed + threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  buf[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void last_kernel(const float* buf, double* host_out, int iters, double tag) {
  float v = buf[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  if (threadIdx.x == 0) host_out[0] = tag + (v > 1e30f ? 1.0 : 0.0);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  float* buf;
  double* host;
  CK(hipMalloc(&buf, 64 * 256 * 4));
  CK(hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocDefault));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int it_a = 600, it_b = 2500, it_c = 600;  // ~ 5 / 15 / 5 us of dependent fmas
  auto wait = [&]() {
    (void)hipEventRecord(ev, st);
    while (hipEventQuery(ev) == hipErrorNotReady) {
    }
  };
  auto run_stream = [&](double tag) {
    hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(256), 0, st, buf, it_a, 1.0f);
    hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, st, buf, it_b, 2.0f);
    hipLaunchKernelGGL(last_kernel, dim3(1), dim3(256), 0, st, buf, host, it_c, tag);
    wait();
  };
  // capture the same sequence
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(256), 0, st, buf, it_a, 1.0f);
  hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, st, buf, it_b, 2.0f);
  hipLaunchKernelGGL(last_kernel, dim3(1), dim3(256), 0, st, buf, host, it_c, 0.0);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  size_t nn = 0;
  CK(hipGraphGetNodes(graph, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  CK(hipGraphGetNodes(graph, nodes.data(), &nn));
  auto run_graph = [&]() {
    (void)hipGraphLaunch(exec, st);
    wait();
  };
  // parameter update of the three kernel nodes (new scalar arguments every call)
  int a_it = it_a, b_it = it_b, c_it = it_c;
  float s1 = 1.0f, s2 = 2.0f;
  double tagv = 0.0;
  void* args_a[] = {&buf, &a_it, &s1};
  void* args_b[] = {&buf, &b_it, &s2};
  void* args_c[] = {&buf, &host, &c_it, &tagv};
  hipKernelNodeParams pa{}, pb{}, pc{};
  pa.func = reinterpret_cast<void*>(spin_kernel); pa.gridDim = dim3(8); pa.blockDim = dim3(256); pa.kernelParams = args_a;
  pb.func = reinterpret_cast<void*>(spin_kernel); pb.gridDim = dim3(64); pb.blockDim = dim3(256); pb.kernelParams = args_b;
  pc.func = reinterpret_cast<void*>(last_kernel); pc.gridDim = dim3(1); pc.blockDim = dim3(256); pc.kernelParams = args_c;
  auto run_graph_update = [&](double tag) {
    tagv = tag;
    s1 += 1e-3f;
    (void)hipGraphExecKernelNodeSetParams(exec, nodes[0], &pa);
    (void)hipGraphExecKernelNodeSetParams(exec, nodes[1], &pb);
    (void)hipGraphExecKernelNodeSetParams(exec, nodes[2], &pc);
    (void)hipGraphLaunch(exec, st);
    wait();
  };
  const int reps = 3000;
  for (int round = 0; round < 3; ++round) {
    for (int i = 0; i < 300; ++i) run_stream(i);
    double t0 = now_us();
    for (int i = 0; i < reps; ++i) run_stream(i);
    const double t_stream = (now_us() - t0) / reps;
    for (int i = 0; i < 300; ++i) run_graph();
    t0 = now_us();
    for (int i = 0; i < reps; ++i) run_graph();
    const double t_graph = (now_us() - t0) / reps;
    for (int i = 0; i < 300; ++i) run_graph_update(i);
    t0 = now_us();
    for (int i = 0; i < reps; ++i) run_graph_update(i);
    const double t_upd = (now_us() - t0) / reps;
    printf("round %d: three stream launches %.1f us | one hipGraphLaunch %.1f us | graph with 3 node updates %.1f us per call (%zu nodes)\n",
           round, t_stream, t_graph, t_upd, nn);
  }
  return 0;
}
