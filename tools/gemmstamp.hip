// tools/gemmstamp.hip -- dev-only: phase timeline of the 64x64 fp32-MFMA GEMM at cfg 3 (Q.W, 16384x300x300).
// Builds bilinear.hip with -DMMS_GEMM_STAMPS: thread 0 of each workgroup stamps s_memtime at
// start / first tile in LDS / k-loop done / stores retired, plus XCC_ID and HW_ID.  s_memtime is per XCD
// and unsynchronised across XCDs, so stamps are normalised to the earliest start ON THE SAME XCD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMMS_GEMM_STAMPS -I include \
//         -I mms_answer_selection_amd/csrc tools/gemmstamp.hip -o /tmp/gemmstamp
#include "../mms_answer_selection_amd/csrc/bilinear.hip"

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
  const int N = 16384, K = 300;
  float *q, *a, *W, *top, *qw;
  CK(hipMalloc(&q, (size_t)N * K * 4)); CK(hipMalloc(&a, (size_t)N * K * 4)); CK(hipMalloc(&qw, (size_t)N * K * 4));
  CK(hipMalloc(&W, (size_t)K * K * 4)); CK(hipMalloc(&top, N * 4));
  std::vector<float> h((size_t)N * K);
  for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
  CK(hipMemcpy(q, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)K * K * 4, hipMemcpyHostToDevice));
  const int wgs = 5 * 256;
  unsigned long long* buf;
  CK(hipMalloc(&buf, (size_t)wgs * 64));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(mms::mms_gemm_stamp_buf), &buf, sizeof(buf)));
  for (int r = 0; r < 3; ++r) mms::simmatrix_forward(N, K, K, q, a, W, top, qw, 0);
  CK(hipDeviceSynchronize());
  CK(hipMemset(buf, 0, (size_t)wgs * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  mms::simmatrix_forward(N, K, K, q, a, W, top, qw, 0);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("forward (GEMM + rowdot) %.1f us by events\n", ms * 1e3);
  std::vector<unsigned long long> st((size_t)wgs * 8);
  CK(hipMemcpy(st.data(), buf, st.size() * 8, hipMemcpyDeviceToHost));
  std::map<unsigned, unsigned long long> t0;   // per XCC
  for (int w = 0; w < wgs; ++w) {
    const unsigned xcc = (unsigned)(st[(size_t)w * 8 + 7] >> 32) & 0xf;
    auto it = t0.find(xcc);
    if (it == t0.end() || st[(size_t)w * 8] < it->second) t0[xcc] = st[(size_t)w * 8];
  }
  const char* names[4] = {"workgroup start", "first tile in LDS", "k loop done", "stores retired"};
  printf("%-20s %8s %8s %8s %8s %8s   (s_memtime ticks after the earliest start on the same XCD)\n", "phase boundary", "min", "p10", "p50", "p90", "max");
  for (int k = 0; k < 4; ++k) {
    std::vector<double> v(wgs);
    for (int w = 0; w < wgs; ++w) v[w] = (double)(st[(size_t)w * 8 + k] - t0[(unsigned)(st[(size_t)w * 8 + 7] >> 32) & 0xf]);
    std::sort(v.begin(), v.end());
    printf("%-20s %8.0f %8.0f %8.0f %8.0f %8.0f\n", names[k], v[0], v[wgs / 10], v[wgs / 2], v[wgs * 9 / 10], v[wgs - 1]);
  }
  {  // synchronised 100 MHz clock (s_memrealtime): when workgroups start and end, 10-ns ticks after the earliest start
    unsigned long long r0 = ~0ull;
    for (int w = 0; w < wgs; ++w) r0 = std::min(r0, st[(size_t)w * 8 + 4]);
    for (int k = 4; k < 6; ++k) {
      std::vector<double> v(wgs);
      for (int w = 0; w < wgs; ++w) v[w] = (double)(st[(size_t)w * 8 + k] - r0) * 0.01;
      std::sort(v.begin(), v.end());
      printf("%-20s %8.2f %8.2f %8.2f %8.2f %8.2f  us (s_memrealtime)\n", k == 4 ? "start" : "end", v[0], v[wgs / 10], v[wgs / 2], v[wgs * 9 / 10], v[wgs - 1]);
    }
    std::vector<double> life(wgs), tk(wgs);
    for (int w = 0; w < wgs; ++w) { life[w] = (double)(st[(size_t)w * 8 + 5] - st[(size_t)w * 8 + 4]) * 0.01; tk[w] = (double)(st[(size_t)w * 8 + 3] - st[(size_t)w * 8]); }
    double sl = 0, stt = 0; for (int w = 0; w < wgs; ++w) { sl += life[w]; stt += tk[w]; }
    printf("mean workgroup life %.2f us = %.0f s_memtime ticks -> %.1f ticks per us\n", sl / wgs, stt / wgs, stt / sl);
    std::vector<double> d1(wgs), d2(wgs), d3(wgs);
    for (int w = 0; w < wgs; ++w) { d1[w] = st[(size_t)w*8+1]-st[(size_t)w*8]; d2[w] = st[(size_t)w*8+2]-st[(size_t)w*8+1]; d3[w] = st[(size_t)w*8+3]-st[(size_t)w*8+2]; }
    std::sort(d1.begin(), d1.end()); std::sort(d2.begin(), d2.end()); std::sort(d3.begin(), d3.end());
    printf("phase durations in ticks (p10 p50 p90): prologue %.0f %.0f %.0f | k loop %.0f %.0f %.0f | epilogue %.0f %.0f %.0f\n",
           d1[wgs/10], d1[wgs/2], d1[wgs*9/10], d2[wgs/10], d2[wgs/2], d2[wgs*9/10], d3[wgs/10], d3[wgs/2], d3[wgs*9/10]);
  }
  // workgroups per CU
  std::map<unsigned long long, int> per_cu;
  for (int w = 0; w < wgs; ++w) {
    const unsigned long long wh = st[(size_t)w * 8 + 7];
    per_cu[((wh >> 32) & 0xf) << 16 | ((wh >> 8) & 0xff)]++;
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second]++;
  printf("distinct (XCC, SE/SH/CU) keys: %zu; workgroups per key -> how many keys:", per_cu.size());
  for (auto& kv : hist) printf("  %d:%d", kv.first, kv.second);
  printf("\nper-XCD workgroups:");
  std::map<unsigned, int> px;
  for (int w = 0; w < wgs; ++w) px[(unsigned)(st[(size_t)w * 8 + 7] >> 32) & 0xf]++;
  for (auto& kv : px) printf(" %u:%d", kv.first, kv.second);
  printf("\nper XCD: mean k-loop ticks, mean end (us), max end (us)\n");
  {
    unsigned long long r0 = ~0ull;
    for (int w = 0; w < wgs; ++w) r0 = std::min(r0, st[(size_t)w * 8 + 4]);
    for (unsigned x = 0; x < 8; ++x) {
      double sl = 0, se = 0, me = 0; int n = 0;
      for (int w = 0; w < wgs; ++w) if (((unsigned)(st[(size_t)w * 8 + 7] >> 32) & 0xf) == x) {
        sl += (double)(st[(size_t)w*8+2] - st[(size_t)w*8+1]); const double e = (double)(st[(size_t)w*8+5] - r0) * 0.01; se += e; me = std::max(me, e); ++n; }
      printf("  xcd %u: n %d  loop %.0f  end mean %.2f max %.2f\n", x, n, sl / n, se / n, me);
    }
    // spread inside a CU vs between CUs
    std::map<unsigned long long, std::vector<double>> cu;
    for (int w = 0; w < wgs; ++w) { const unsigned long long wh = st[(size_t)w * 8 + 7]; cu[((wh >> 32) & 0xf) << 16 | ((wh >> 8) & 0xff)].push_back((double)(st[(size_t)w*8+5] - r0) * 0.01); }
    std::vector<double> cumax, curange;
    for (auto& kv : cu) { auto mm = std::minmax_element(kv.second.begin(), kv.second.end()); cumax.push_back(*mm.second); curange.push_back(*mm.second - *mm.first); }
    std::sort(cumax.begin(), cumax.end()); std::sort(curange.begin(), curange.end());
    printf("per CU: last end p10 %.2f p50 %.2f p90 %.2f max %.2f us; (last - first end) inside a CU p50 %.2f p90 %.2f us\n",
           cumax[25], cumax[128], cumax[230], cumax[255], curange[128], curange[230]);
  }
  printf("first 12 workgroups (linear id: xcc hw_id start first_tile loop_done retired):\n");
  for (int w = 0; w < 12; ++w) {
    const unsigned xcc = (unsigned)(st[(size_t)w * 8 + 7] >> 32) & 0xf;
    printf("  %4d: xcc %u hw %08x  %8llu %8llu %8llu %8llu\n", w, xcc, (unsigned)st[(size_t)w * 8 + 7],
           st[(size_t)w * 8] - t0[xcc], st[(size_t)w * 8 + 1] - t0[xcc], st[(size_t)w * 8 + 2] - t0[xcc], st[(size_t)w * 8 + 3] - t0[xcc]);
  }
  return 0;
}
