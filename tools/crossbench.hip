// tools/crossbench.hip -- dev-only: time of the 40x40 Euclid forward (cross geometry), optionally with a
// timing ablation compiled in (-DMMS_XABL=1 no sqrt/divide, 2 no arithmetic loop, 3 no stores; outputs WRONG).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I mms_answer_selection_amd/csrc \
//         tools/crossbench.hip -o /tmp/crossbench
#include "../mms_answer_selection_amd/csrc/simcross_elementwise.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
  const int W = 40, D = 50;
  for (int N : {1517, 8192}) {
    float *q, *a, *top;
    CK(hipMalloc(&q, (size_t)N * W * D * 4)); CK(hipMalloc(&a, (size_t)N * W * D * 4)); CK(hipMalloc(&top, (size_t)N * W * W * 4));
    std::vector<float> h((size_t)N * W * D);
    for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(q, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) mms::simcross_elementwise_forward(1, N, W, W, D, q, a, top, nullptr, nullptr, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 50; ++r) mms::simcross_elementwise_forward(1, N, W, W, D, q, a, top, nullptr, nullptr, 0);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("N=%5d  %.2f us per forward\n", N, ms * 1e3 / 50);
    CK(hipFree(q)); CK(hipFree(a)); CK(hipFree(top));
  }
  return 0;
}
