// tools/f16bench.hip -- dev-only: the fp16-storage fused Euclid kernel at cfg 5's shard (8192 x 1024) and at
// 65536 x 1024, optionally with a timing ablation compiled in (-DMMS_F16ABL=1 no chain, 2 no stores).
#include "../mms_answer_selection_amd/csrc/simcross_elementwise.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
  const int D = 1024;
  for (int N : {8192, 65536}) {
    _Float16 *q, *a, *dq, *da; float *dT, *top;
    const size_t nb = (size_t)N * D * 2;
    CK(hipMalloc(&q, nb)); CK(hipMalloc(&a, nb)); CK(hipMalloc(&dq, nb)); CK(hipMalloc(&da, nb));
    CK(hipMalloc(&dT, N * 4)); CK(hipMalloc(&top, N * 4));
    std::vector<_Float16> h((size_t)N * D);
    for (auto& v : h) { float u = 0.f; for (int k = 0; k < 12; ++k) u += rand() / (float)RAND_MAX; v = (_Float16)((u - 6.0f) * 0.4f); }
    CK(hipMemcpy(q, h.data(), nb, hipMemcpyHostToDevice));
    for (auto& v : h) { float u = 0.f; for (int k = 0; k < 12; ++k) u += rand() / (float)RAND_MAX; v = (_Float16)((u - 6.0f) * 0.4f); }
    CK(hipMemcpy(a, h.data(), nb, hipMemcpyHostToDevice));
    std::vector<float> g(N, 1.0f);
    CK(hipMemcpy(dT, g.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) mms::simcross_euclid_rows_f16(N, D, q, a, dT, top, dq, da, true, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 30; ++r) mms::simcross_euclid_rows_f16(N, D, q, a, dT, top, dq, da, true, 0);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("N=%6d  %.2f us per fused fwd+bwd\n", N, ms * 1e3 / 30);
#ifdef MMS_STAMPS
    unsigned miss = 0;
    CK(hipMemcpyFromSymbol(&miss, HIP_SYMBOL(mms::mms_miss_count), sizeof(miss)));
    printf("          window misses so far: %u in %d pair evaluations\n", miss, 33 * N);
#endif
    CK(hipFree(q)); CK(hipFree(a)); CK(hipFree(dq)); CK(hipFree(da)); CK(hipFree(dT)); CK(hipFree(top));
  }
  return 0;
}
