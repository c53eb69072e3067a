// tools/panel32bench.hip -- dev harness (not product): the 32-row panel GEMM (tools/panel32_gemm.h) checked against
// an fp64 product on the device and timed next to round 2's 64-row kernel, on cfg 3's three products
// (SimMatrix 16384 x 300 x 300).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I mms_answer_selection_amd/csrc -I tools \
//         tools/panel32bench.hip -o /tmp/panel32bench && /tmp/panel32bench [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "panel32_gemm.h"
using namespace mms;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// C(i,n) = rs[i] * sum_k A(i,k) * ks[k] * B(k,n), fp64 accumulate; generic strides
__global__ void ref_gemm(int M, int N, int K, const float* A, long long a_i, long long a_k, const float* B, long long b_k,
                         long long b_n, const float* rs, const float* ks, double* C) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)M * N) return;
  const int i = (int)(e / N), n = (int)(e % N);
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)A[i * a_i + k * a_k] * (ks ? (double)ks[k] : 1.0) * (double)B[k * b_k + n * b_n];
  C[e] = (rs ? (double)rs[i] : 1.0) * s;
}
__global__ void slab_sum(const float* part, int splits, long long n, float* out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(long long)k * n + e];
  out[e] = s;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16384, K = 300;
  float *q, *a, *W, *qw, *dq, *top, *dT, *part, *da, *dWo;
  double* ref;
  CK(hipMalloc(&q, (size_t)N * K * 4)); CK(hipMalloc(&a, (size_t)N * K * 4)); CK(hipMalloc(&W, (size_t)K * K * 4));
  CK(hipMalloc(&qw, (size_t)N * K * 4)); CK(hipMalloc(&dq, (size_t)N * K * 4)); CK(hipMalloc(&top, N * 4));
  CK(hipMalloc(&dT, N * 4)); CK(hipMalloc(&da, (size_t)N * K * 4)); CK(hipMalloc(&part, (size_t)64 * K * K * 4));
  CK(hipMalloc(&dWo, (size_t)K * K * 4)); CK(hipMalloc(&ref, (size_t)N * K * 8));
  std::vector<float> h((size_t)N * K), hq, ha, hW((size_t)K * K), hdT(N);
  srand(7);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  hq = h; CK(hipMemcpy(q, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  ha = h; CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : hW) v = 0.16f * (rand() / (float)RAND_MAX - 0.5f);
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : hdT) v = 2.f * (rand() / (float)RAND_MAX - 0.5f);
  CK(hipMemcpy(dT, hdT.data(), N * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto check = [&](const char* name, const float* got, long long cnt) {
    std::vector<float> g(cnt); std::vector<double> r(cnt);
    CK(hipMemcpy(g.data(), got, cnt * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), ref, cnt * 8, hipMemcpyDeviceToHost));
    double maxabs = 0, maxref = 0; long long bad = -1;
    for (long long i = 0; i < cnt; ++i) {
      const double d = std::fabs((double)g[i] - r[i]);
      if (!(d <= maxabs)) { maxabs = d; bad = i; }
      maxref = std::max(maxref, std::fabs(r[i]));
    }
    printf("   check %-28s max|err| %.3e  max|ref| %.3e  rel %.2e  %s (worst at %lld)\n", name, maxabs, maxref,
           maxabs / std::max(1.0, maxref), maxabs <= 1e-5 * std::max(1.0, maxref) ? "OK" : "** FAIL **", bad);
  };
  auto run = [&](const char* name, auto&& body, double flop) {
    body(); CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 8; ++k) body();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f / 8);
    }
    std::sort(t.begin(), t.end());
    printf("%-44s median %8.2f us  min %8.2f   %6.1f TFLOP/s\n", name, t[2], t[0], flop / t[2] / 1e6);
  };
#ifdef MMS_P32_STAMPS
  unsigned long long* sb; CK(hipMalloc(&sb, 1024 * 16 * 8)); CK(hipMemset(sb, 0, 1024 * 16 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(p32_stamp_buf), &sb, sizeof(sb)));
  auto stamps = [&](const char* name, int nwg) {
    std::vector<unsigned long long> h(1024 * 16);
    CK(hipMemcpy(h.data(), sb, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < nwg; ++b) if (h[b * 16]) tmin = std::min(tmin, h[b * 16]);
    std::vector<double> start, b0, loop, epi, clk; int percu[8][64] = {}; int late = 0;
    for (int b = 0; b < nwg; ++b) {
      const unsigned long long* x = &h[(size_t)b * 16];
      if (!x[0]) continue;
      start.push_back((x[0] - tmin) / 100.0); b0.push_back((x[2] - x[0]) / 100.0); loop.push_back((x[3] - x[2]) / 100.0);
      epi.push_back((x[6] - x[3]) / 100.0); clk.push_back((double)(x[5] - x[4]) / (double)(x[3] - x[2]) * 100.0);
      const unsigned hw = (unsigned)x[1], xcc = (unsigned)(x[1] >> 32) & 15;
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      percu[xcc & 7][(se * 2 + sh) * 16 + cu & 63]++;
      if ((x[0] - tmin) / 100.0 > 5.0) ++late;
    }
    int hist[8] = {};
    for (int x = 0; x < 8; ++x) for (int c = 0; c < 64; ++c) hist[std::min(percu[x][c], 7)]++;
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    auto mx = [](std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
    printf("   stamps %-22s start med %.2f max %.2f us (%d start later than 5 us); launch->B_0 med %.2f; main loop med %.2f max %.2f us @ %.0f MHz; epilogue med %.2f max %.2f\n",
           name, med(start), mx(start), late, med(b0), med(loop), mx(loop), med(clk), med(epi), mx(epi));
    printf("          workgroups per (xcc, cu) slot: 0:%d 1:%d 2:%d 3:%d 4+:%d\n", hist[0], hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
    CK(hipMemset(sb, 0, 1024 * 16 * 8));
  };
#else
  auto stamps = [&](const char*, int) {};
#endif
  const double fl = 2.0 * N * K * K;
  const unsigned rblk = (unsigned)(((long long)N * K + 255) / 256), wblk = (unsigned)((K * K + 255) / 256);

  // ---- forward: qw = Q W, top_i = qw_i . a_i -------------------------------------------------------------------
  CK(hipMemset(qw, 0xff, (size_t)N * K * 4)); CK(hipMemset(top, 0xff, N * 4));
  {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K); p.Y = a; p.ldy = K; p.rowdot = top;
    if (!panel32_eligible(p, true, false)) printf("fwd not eligible\n");
    panel32_launch_t<19, true, false>(p, st); CK(hipStreamSynchronize(st));
  }
  hipLaunchKernelGGL(ref_gemm, dim3(rblk), dim3(256), 0, st, N, K, K, q, (long long)K, 1LL, W, (long long)K, 1LL,
                     (const float*)nullptr, (const float*)nullptr, ref);
  CK(hipStreamSynchronize(st));
  check("fwd qw = Q.W", qw, (long long)N * K);
  {
    std::vector<double> r((size_t)N * K); std::vector<float> tg(N);
    CK(hipMemcpy(r.data(), ref, r.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(tg.data(), top, N * 4, hipMemcpyDeviceToHost));
    double me = 0, mr = 0;
    for (int i = 0; i < N; ++i) {
      double s = 0; for (int n = 0; n < K; ++n) s += r[(size_t)i * K + n] * (double)ha[(size_t)i * K + n];
      me = std::max(me, std::fabs(s - tg[i])); mr = std::max(mr, std::fabs(s));
      if (!(std::fabs(s - tg[i]) <= 1e30)) me = 1e30;
    }
    printf("   check %-28s max|err| %.3e  max|ref| %.3e  %s\n", "fwd row dot", me, mr, me <= 1e-5 * std::max(1.0, mr) ? "OK" : "** FAIL **");
  }
  // ---- dq = diag(dT) A W^T, B read n-major straight from W; side job da = diag(dT) qw ---------------------------
  CK(hipMemset(dq, 0xff, (size_t)N * K * 4)); CK(hipMemset(da, 0xff, (size_t)N * K * 4));
  {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 1;
    p.side_in = qw; p.side_out = da; p.side_scale = dT; p.side_ld = K; p.side_cols = K;
    if (!panel32_eligible(p, true, true)) printf("dq not eligible\n");
    panel32_launch_t<19, true, true>(p, st); CK(hipStreamSynchronize(st));
  }
  hipLaunchKernelGGL(ref_gemm, dim3(rblk), dim3(256), 0, st, N, K, K, a, (long long)K, 1LL, W, 1LL, (long long)K,
                     dT, (const float*)nullptr, ref);
  CK(hipStreamSynchronize(st));
  check("dq = dT . A W^T (n-major B)", dq, (long long)N * K);
  {
    std::vector<float> g1((size_t)N * K), g2((size_t)N * K);
    CK(hipMemcpy(g1.data(), da, g1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g2.data(), qw, g2.size() * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (size_t i = 0; i < g1.size(); ++i) if (g1[i] != hdT[i / K] * g2[i]) ++bad;
    printf("   check %-28s %lld mismatching elements %s\n", "side job da = dT . qw", bad, bad ? "** FAIL **" : "OK");
  }
  // ---- dW = Q^T diag(dT) A, split-K slabs + ordered sum ---------------------------------------------------------
  {
    PanelArgs p = panel_args(K, K, N, q, K, a, K, part, K); p.kscale = dT;
    p.ksplit = panel32_pick_ksplit(K, 1, N, &p.kchunk); p.c_ks = (long long)K * K;
    printf("dW: ksplit %d kchunk %d eligible %d\n", p.ksplit, p.kchunk, (int)panel32_eligible(p, false, false));
    CK(hipMemset(part, 0xff, (size_t)p.ksplit * K * K * 4));
    panel32_launch_t<19, false, false>(p, st);
    hipLaunchKernelGGL(slab_sum, dim3(wblk), dim3(256), 0, st, part, p.ksplit, (long long)K * K, dWo);
    CK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(ref_gemm, dim3(wblk), dim3(256), 0, st, K, K, N, q, 1LL, (long long)K, a, (long long)K, 1LL,
                       (const float*)nullptr, dT, ref);
    CK(hipStreamSynchronize(st));
    check("dW = Q^T diag(dT) A", dWo, (long long)K * K);
  }

  {
    int nb = -1;
    auto k1 = panel32_kernel<19, true, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P32Geom<19, true, false>::kLdsBytes));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k1, 256, P32Geom<19, true, false>::kLdsBytes));
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k1)));
    printf("occupancy: panel32_kernel<19,true,false> %d workgroups per CU (256 threads, %zu B dynamic LDS, %d VGPRs, %zu B static LDS, %zu B scratch)\n",
           nb, (size_t)P32Geom<19, true, false>::kLdsBytes, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
    for (size_t l : {(size_t)32768, (size_t)65536, (size_t)70000, (size_t)77056, (size_t)81920}) {
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k1, 256, l));
      printf("   with %zu B of dynamic LDS: %d\n", l, nb);
    }
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("   device: %d CUs, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerBlock %zu, regsPerMultiprocessor %d\n",
           pr.multiProcessorCount, pr.sharedMemPerMultiprocessor, pr.sharedMemPerBlock, pr.regsPerMultiprocessor);
  }
  // ---- timing -----------------------------------------------------------------------------------------------------
  run("r2  fwd  Q.W + rowdot            (64-row)", [&] {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K); p.Y = a; p.ldy = K; p.rowdot = top;
    panel_launch_t<19, true>(p, st); }, fl);
  run("r3  fwd  Q.W + rowdot            (32-row x2)", [&] {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K); p.Y = a; p.ldy = K; p.rowdot = top;
    panel32_launch_t<19, true, false>(p, st); }, fl);
  stamps("fwd + rowdot", 512);
  run("r3  fwd  Q.W only                (32-row x2)", [&] {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K);
    panel32_launch_t<19, true, false>(p, st); }, fl);
  stamps("fwd Q.W only", 512);
  run("r3  fwd  no epilogue at all      (32-row x2)", [&] {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, nullptr, K);
    panel32_launch_t<19, true, false>(p, st); }, fl);
  run("r3  fwd  Q.W only, ONE workgroup per CU (LDS padded)", [&] {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K);
    auto kern = panel32_kernel<19, true, false>;
    static bool once = false;
    const size_t lds = 100 * 1024;
    if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
    p.row_blocks = (p.M + 31) / 32;
    hipLaunchKernelGGL(kern, dim3(p.row_blocks, 1), dim3(256), lds, st, p); }, fl);
  stamps("fwd, LDS padded", 512);
  run("r2  dq + side job da             (64-row)", [&] {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 1;
    p.side_in = qw; p.side_out = da; p.side_scale = dT; p.side_ld = K; p.side_cols = K;
    panel_launch_t<19, true>(p, st); }, fl);
  run("r3  dq + side job da, n-major B  (32-row x2)", [&] {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 1;
    p.side_in = qw; p.side_out = da; p.side_scale = dT; p.side_ld = K; p.side_cols = K;
    panel32_launch_t<19, true, true>(p, st); }, fl);
  run("r3  dq alone, n-major B          (32-row x2)", [&] {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 1;
    panel32_launch_t<19, true, true>(p, st); }, fl);
  run("r3  dq alone, k-major B          (32-row x2)", [&] {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 1;
    panel32_launch_t<19, true, false>(p, st); }, fl);
  {   // the grouped backward launch: correctness, then time
    CK(hipMemset(dq, 0xff, (size_t)N * K * 4)); CK(hipMemset(da, 0xff, (size_t)N * K * 4));
    PanelArgs p1 = panel_args(N, K, K, a, K, W, K, dq, K); p1.rowscale = dT; p1.stream_c = 1;
    p1.side_in = qw; p1.side_out = da; p1.side_scale = dT; p1.side_ld = K; p1.side_cols = K;
    PanelArgs p2 = panel_args(K, K, N, q, K, a, K, part, K); p2.kscale = dT;
    p2.ksplit = panel32_pick_ksplit(K, 1, N, &p2.kchunk); p2.c_ks = (long long)K * K;
    CK(hipMemset(part, 0xff, (size_t)p2.ksplit * K * K * 4));
    panel32_pair_launch(p1, p2, st);
    hipLaunchKernelGGL(slab_sum, dim3(wblk), dim3(256), 0, st, part, p2.ksplit, (long long)K * K, dWo);
    CK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(ref_gemm, dim3(rblk), dim3(256), 0, st, N, K, K, a, (long long)K, 1LL, W, 1LL, (long long)K,
                       dT, (const float*)nullptr, ref);
    CK(hipStreamSynchronize(st));
    check("pair: dq", dq, (long long)N * K);
    hipLaunchKernelGGL(ref_gemm, dim3(wblk), dim3(256), 0, st, K, K, N, q, 1LL, (long long)K, a, (long long)K, 1LL,
                       (const float*)nullptr, dT, ref);
    CK(hipStreamSynchronize(st));
    check("pair: dW", dWo, (long long)K * K);
    run("r3  PAIR: dq + side job  ||  dW split-K, ONE launch", [&] { panel32_pair_launch(p1, p2, st); }, 2 * fl);
    run("r3  dq + side job, then dW: two launches", [&] {
      panel32_launch_t<19, true, true>(p1, st); panel32_launch_t<19, false, false>(p2, st); }, 2 * fl);
    run("r2  dW, reduce+transpose, dq + side job: three launches", [&] {
      PanelArgs o2 = panel_args(K, K, N, q, K, a, K, part, K); o2.kscale = dT;
      o2.ksplit = panel_pick_ksplit(o2.row_blocks, 1, N, &o2.kchunk); o2.c_ks = (long long)K * K;
      panel_launch_t<19, false>(o2, st);
      PanelArgs o1 = p1; o1.row_blocks = (N + 63) / 64;
      panel_launch_t<19, true>(o1, st); }, 2 * fl);
  }
  run("r2  dW   Q^T diag(dT) A split-K  (64-row)", [&] {
    PanelArgs p = panel_args(K, K, N, q, K, a, K, part, K); p.kscale = dT;
    p.ksplit = panel_pick_ksplit(p.row_blocks, 1, N, &p.kchunk); p.c_ks = (long long)K * K;
    panel_launch_t<19, false>(p, st); }, fl);
  run("r3  dW   Q^T diag(dT) A split-K  (32-row x2)", [&] {
    PanelArgs p = panel_args(K, K, N, q, K, a, K, part, K); p.kscale = dT;
    p.ksplit = panel32_pick_ksplit(K, 1, N, &p.kchunk); p.c_ks = (long long)K * K;
    panel32_launch_t<19, false, false>(p, st); }, fl);
  return 0;
}
