// tools/stampbench.hip -- dev-only: phase timeline of the fused rows kernel.
// Builds the kernel source with -DMMS_STAMPS (per-wave s_memtime stamps into a
// side buffer) and prints, for an HBM-cold and a cache-warm launch, when each
// phase boundary is reached across the 2048 waves (ns after the earliest wave
// started; percentiles).  Stamps add fences: read SHARES, not the total.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMMS_STAMPS -I include \
//         -I mms_answer_selection_amd/csrc tools/stampbench.hip -o /tmp/stampbench
#include "../mms_answer_selection_amd/csrc/simcross_elementwise.hip"

#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
  const int N = 4096, D = 300, ring = 64;
  const size_t nb = (size_t)N * D * 4;
  std::vector<float*> q(ring), a(ring), dq(ring), da(ring);
  float *dT, *top;
  CK(hipMalloc(&dT, N * 4)); CK(hipMalloc(&top, N * 4));
  std::vector<float> h((size_t)N * D);
  for (int s = 0; s < ring; ++s) {
    CK(hipMalloc(&q[s], nb)); CK(hipMalloc(&a[s], nb)); CK(hipMalloc(&dq[s], nb)); CK(hipMalloc(&da[s], nb));
    for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(q[s], h.data(), nb, hipMemcpyHostToDevice));
    for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(a[s], h.data(), nb, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(dT, h.data(), N * 4, hipMemcpyHostToDevice));
  const int waves = N / 2;
  unsigned long long* buf;
  CK(hipMalloc(&buf, (size_t)waves * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(mms::mms_stamp_buf), &buf, sizeof(buf)));
  const char* names[8] = {"wave start", "loads issued", "data arrived", "squares in LDS + pred", "chain done",
                          "T, coefficients", "stores issued", "stores retired"};
  for (int warm = 0; warm < 2; ++warm) {
    // pre-condition the caches: cold = walk the ring once, warm = run the same slot repeatedly
    for (int r = 0; r < (warm ? 4 : ring); ++r) {
      const int s = warm ? 0 : r;
      mms::simcross_elementwise_forward_backward(1, N, 1, 1, D, q[s], a[s], dT, top, nullptr, nullptr, dq[s], da[s], 0);
    }
    CK(hipMemset(buf, 0, (size_t)waves * 64));
    mms::simcross_elementwise_forward_backward(1, N, 1, 1, D, q[0], a[0], dT, top, nullptr, nullptr, dq[0], da[0], 0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)waves * 8);
    CK(hipMemcpy(st.data(), buf, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < waves; ++w) t0 = std::min(t0, st[(size_t)w * 8]);
    printf("=== %s launch (ticks of s_memtime after the earliest wave start; 2048 waves)\n", warm ? "cache-warm" : "HBM-cold");
    printf("%-24s %8s %8s %8s %8s %8s\n", "phase boundary", "min", "p10", "p50", "p90", "max");
    for (int k = 0; k < 8; ++k) {
      std::vector<double> v(waves);
      for (int w = 0; w < waves; ++w) v[w] = (double)(st[(size_t)w * 8 + k] - t0);
      std::sort(v.begin(), v.end());
      printf("%-24s %8.0f %8.0f %8.0f %8.0f %8.0f\n", names[k], v[0], v[waves / 10], v[waves / 2], v[waves * 9 / 10], v[waves - 1]);
    }
    // per-wave phase durations (median)
    unsigned miss = 0;
    CK(hipMemcpyFromSymbol(&miss, HIP_SYMBOL(mms::mms_miss_count), sizeof(miss)));
    printf("speculation misses so far (all launches): %u\n", miss);
    printf("median per-wave durations:");
    for (int k = 1; k < 8; ++k) {
      std::vector<double> v(waves);
      for (int w = 0; w < waves; ++w) v[w] = (double)(st[(size_t)w * 8 + k] - st[(size_t)w * 8 + k - 1]);
      std::sort(v.begin(), v.end());
      printf("  [%d->%d] %.0f", k - 1, k, v[waves / 2]);
    }
    printf("\n");
  }
  return 0;
}
