// tools/graphbound.hip -- dev microbenchmark: what does a hipGraphLaunch BOUNDARY cost on the device?  Graphs of n
// empty kernels replayed back to back: us per kernel = floor + boundary / n.
//   hipcc --offload-arch=gfx950 -O3 tools/graphbound.hip -o tools/bin/graphbound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(512) void empty_kernel(int) {}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int n : {1, 2, 4, 8, 16, 32, 64, 128}) {
    hipGraph_t g; hipGraphExec_t ge[2];
    for (int k = 0; k < 2; ++k) {
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, i);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge[k], g, nullptr, nullptr, 0));
    }
    const int launches = 4096 / n > 64 ? 4096 / n : 64;
    for (int i = 0; i < 8; ++i) CK(hipGraphLaunch(ge[i & 1], st));
    CK(hipStreamSynchronize(st));
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < launches; ++i) CK(hipGraphLaunch(ge[i & 1], st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms * 1e3f / launches);
    }
    std::sort(ts.begin(), ts.end());
    printf("graph of %3d empty kernels: %8.2f us per graph launch = %6.3f us per kernel\n", n, ts[2], ts[2] / n);
  }
  return 0;
}
