// tools/bx3bench.hip -- dev harness (not product): the split-bf16 panel GEMM (csrc/bx3_gemm.h) checked against an fp64
// product on the device and timed next to the fp32-MFMA panel kernel, on cfg 3's products (SimMatrix 16384 x 300 x 300).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I mms_answer_selection_amd/csrc \
//         tools/bx3bench.hip -o /tmp/bx3bench && /tmp/bx3bench [N] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "bx3_gemm.h"
using namespace mms;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void ref_tn(int M, int N, int K, const float* A, const float* B, const float* ks, double* C) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= M * N) return;
  const int i = e / N, j = e % N;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)A[(long long)k * M + i] * (double)(ks[k] * B[(long long)k * N + j]);
  C[e] = s;
}
__global__ void slab_sum(const float* part, int splits, long long n, float* out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(long long)k * n + e];
  out[e] = s;
}
__global__ void ref_gemm(int M, int N, int K, const float* A, long long a_i, long long a_k, const float* B, long long b_k,
                         long long b_n, const float* rs, double* C) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)M * N) return;
  const int i = (int)(e / N), n = (int)(e % N);
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)A[i * a_i + k * a_k] * (double)B[k * b_k + n * b_n];
  C[e] = (rs ? (double)rs[i] : 1.0) * s;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16384, K = argc > 2 ? atoi(argv[2]) : 300;
  float *q, *a, *W, *qw, *dq, *top, *dT;
  double* ref; bx3_u4* img;
  CK(hipMalloc(&q, (size_t)N * K * 4)); CK(hipMalloc(&a, (size_t)N * K * 4)); CK(hipMalloc(&W, (size_t)K * K * 4));
  CK(hipMalloc(&qw, (size_t)N * K * 4)); CK(hipMalloc(&dq, (size_t)N * K * 4)); CK(hipMalloc(&top, N * 4));
  CK(hipMalloc(&dT, N * 4)); CK(hipMalloc(&ref, (size_t)N * K * 8)); CK(hipMalloc(&img, bx3_image_bytes(K, K)));
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
    double maxabs = 0, maxref = 0, sq = 0; long long bad = -1;
    for (long long i = 0; i < cnt; ++i) {
      const double d = std::fabs((double)g[i] - r[i]);
      if (!(d <= maxabs)) { maxabs = d; bad = i; }
      sq += d * d;
      maxref = std::max(maxref, std::fabs(r[i]));
    }
    printf("   check %-28s max|err| %.3e rms %.3e  max|ref| %.3e  rel %.2e  %s (worst at %lld)\n", name, maxabs,
           std::sqrt(sq / cnt), maxref, maxabs / std::max(1.0, maxref), maxabs <= 1e-5 * std::max(1.0, maxref) ? "OK" : "** FAIL **", bad);
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
    printf("%-52s median %8.2f us  min %8.2f   %6.1f TFLOP/s (fp32-equivalent)\n", name, t[2], t[0], flop / t[2] / 1e6);
  };
#ifdef MMS_BX3_STAMPS
  unsigned long long* sb; CK(hipMalloc(&sb, 1024 * 16 * 8)); CK(hipMemset(sb, 0, 1024 * 16 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(bx3_stamp_buf), &sb, sizeof(sb)));
  auto stamps = [&](const char* name, int nwg) {
    std::vector<unsigned long long> h(1024 * 16);
    CK(hipMemcpy(h.data(), sb, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < nwg; ++b) if (h[b * 16]) tmin = std::min(tmin, h[b * 16]);
    std::vector<double> start, b0, loop, epi, clk, end;
    for (int b = 0; b < nwg; ++b) {
      const unsigned long long* x = &h[(size_t)b * 16];
      if (!x[0]) continue;
      start.push_back((x[0] - tmin) / 100.0); b0.push_back((x[1] - x[0]) / 100.0); loop.push_back((x[2] - x[1]) / 100.0);
      epi.push_back((x[3] - x[2]) / 100.0); clk.push_back((double)(x[5] - x[4]) / (double)(x[2] - x[1]) * 100.0);
      end.push_back((x[3] - tmin) / 100.0);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    auto mx = [](std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
    printf("   stamps %-22s start med %.2f max %.2f us; launch->barrier 0 med %.2f max %.2f; main loop med %.2f max %.2f us @ %.0f MHz; epilogue med %.2f max %.2f; last end %.2f\n",
           name, med(start), mx(start), med(b0), mx(b0), med(loop), mx(loop), med(clk), med(epi), mx(epi), mx(end));
    CK(hipMemset(sb, 0, 1024 * 16 * 8));
  };
#else
  auto stamps = [&](const char*, int) {};
#endif
  const double fl = 2.0 * N * K * K;
  const unsigned rblk = (unsigned)(((long long)N * K + 255) / 256);
  auto fwd_args = [&] {
    Bx3Args p{}; p.M = N; p.N = K; p.K = K; p.A = q; p.lda = K; p.img = img; p.C = qw; p.ldc = K; p.Y = a; p.ldy = K;
    p.rowdot = top; p.rd_stride = 1; return p; };
  auto dq_args = [&] {
    Bx3Args p{}; p.M = N; p.N = K; p.K = K; p.A = a; p.lda = K; p.img = img; p.C = dq; p.ldc = K; p.rowscale = dT;
    p.stream_c = 1; return p; };

  // ---- forward: qw = Q W, top_i = qw_i . a_i
  CK(hipMemset(qw, 0xff, (size_t)N * K * 4)); CK(hipMemset(top, 0xff, N * 4));
  { Bx3Args p = fwd_args(); if (!bx3_eligible(p)) printf("fwd not eligible\n");
    bx3_split_b(W, K, 1, K, K, img, st, top, 1, N); bx3_launch(p, st); CK(hipStreamSynchronize(st)); CK(hipGetLastError()); }
  hipLaunchKernelGGL(ref_gemm, dim3(rblk), dim3(256), 0, st, N, K, K, q, (long long)K, 1LL, W, (long long)K, 1LL, (const float*)nullptr, ref);
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
  // ---- dq = diag(dT) A W^T
  CK(hipMemset(dq, 0xff, (size_t)N * K * 4));
  { Bx3Args p = dq_args(); bx3_split_b(W, 1, K, K, K, img, st); bx3_launch(p, st); CK(hipStreamSynchronize(st)); CK(hipGetLastError()); }
  hipLaunchKernelGGL(ref_gemm, dim3(rblk), dim3(256), 0, st, N, K, K, a, (long long)K, 1LL, W, 1LL, (long long)K, dT, ref);
  CK(hipStreamSynchronize(st));
  check("dq = dT . A W^T", dq, (long long)N * K);
  // ---- the side job of the dq launch: da = dT . qw
  { float* da; CK(hipMalloc(&da, (size_t)N * K * 4)); CK(hipMemset(da, 0xff, (size_t)N * K * 4));
    Bx3Args p = dq_args(); p.side_in = qw; p.side_out = da; p.side_scale = dT; p.side_ld = K; p.side_cols = K;
    if (!bx3_eligible(p)) printf("dq + side job not eligible\n");
    bx3_launch(p, st); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    std::vector<float> g1((size_t)N * K), g2((size_t)N * K);
    CK(hipMemcpy(g1.data(), da, g1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g2.data(), qw, g2.size() * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (size_t i = 0; i < g1.size(); ++i) if (g1[i] != hdT[i / K] * g2[i]) ++bad;
    printf("   check %-28s %lld mismatching elements %s\n", "side job da = dT . qw", bad, bad ? "** FAIL **" : "OK");
    check("dq beside the side job", dq, (long long)N * K);
    CK(hipFree(da)); }
  // ---- dW = Q^T diag(dT) A: split-K slabs + ordered sum
  float *part, *dWo; CK(hipMalloc(&part, (size_t)72 * K * K * 4)); CK(hipMalloc(&dWo, (size_t)K * K * 4));
  Bx3TnArgs tn{}; tn.M = K; tn.N = K; tn.K = N; tn.A = q; tn.lda = K; tn.B = a; tn.ldb = K; tn.kscale = dT; tn.C = part;
  tn.c_ks = (long long)K * K; tn.nchunks = bx3_tn_pick_chunks(N, bx3_tn_quads(K, K), &tn.kchunk);
  printf("dW: %d quadrants, %d chunks of %d pairs, eligible %d\n", bx3_tn_quads(K, K), tn.nchunks, tn.kchunk, (int)bx3_tn_eligible(tn));
  const unsigned wblk = (unsigned)((K * K + 255) / 256);
  { CK(hipMemset(part, 0xff, (size_t)tn.nchunks * K * K * 4));
    bx3_tn_launch(tn, st);
    hipLaunchKernelGGL(slab_sum, dim3(wblk), dim3(256), 0, st, part, tn.nchunks, (long long)K * K, dWo);
    CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    hipLaunchKernelGGL(ref_tn, dim3(wblk), dim3(256), 0, st, K, K, N, q, a, dT, ref);
    CK(hipStreamSynchronize(st));
    check("dW = Q^T diag(dT) A", dWo, (long long)K * K); }
  // the fp32 panel kernel's error on the same product, for scale
  { PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K); panel_launch(p, true, st); CK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(ref_gemm, dim3(rblk), dim3(256), 0, st, N, K, K, q, (long long)K, 1LL, W, (long long)K, 1LL, (const float*)nullptr, ref);
    CK(hipStreamSynchronize(st)); check("(fp32 panel kernel, Q.W)", qw, (long long)N * K); }

  { hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&bx3_kernel<5>)));
    printf("bx3_kernel<5>: %d VGPRs, %zu B scratch, %zu B LDS\n", fa.numRegs, fa.localSizeBytes, (size_t)Bx3Geom<5>::kLdsBytes); }
  run("fp32 panel  fwd Q.W + rowdot", [&] {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K); p.Y = a; p.ldy = K; p.rowdot = top; p.rd_stride = 1;
    panel_launch(p, true, st); }, fl);
  run("bx3 split W only", [&] { bx3_split_b(W, K, 1, K, K, img, st); }, 0);
  run("bx3 fwd Q.W + rowdot (image ready)", [&] { bx3_launch(fwd_args(), st); }, fl);
  stamps("fwd + rowdot", 256);
  run("bx3 fwd Q.W + rowdot (split + product)", [&] { bx3_split_b(W, K, 1, K, K, img, st, top, 1, N); bx3_launch(fwd_args(), st); }, fl);
  run("bx3 fwd Q.W, no store (rowdot only)", [&] { Bx3Args p = fwd_args(); p.C = nullptr; bx3_launch(p, st); }, fl);
  run("bx3 Q.W, store only (no rowdot)", [&] { Bx3Args p = fwd_args(); p.Y = nullptr; p.rowdot = nullptr; bx3_launch(p, st); }, fl);
  run("bx3 Q.W, no store, no rowdot", [&] { Bx3Args p = fwd_args(); p.Y = nullptr; p.rowdot = nullptr; p.C = nullptr; bx3_launch(p, st); }, fl);
  stamps("no store no rowdot", 256);
  run("bx3 dq = dT . A W^T (split W^T + product)", [&] { bx3_split_b(W, 1, K, K, K, img, st); bx3_launch(dq_args(), st); }, fl);
  run("bx3 dW slabs (split-K product only)", [&] { bx3_tn_launch(tn, st); }, fl);
  run("bx3 dW slabs + slab sum", [&] { bx3_tn_launch(tn, st);
    hipLaunchKernelGGL(slab_sum, dim3(wblk), dim3(256), 0, st, part, tn.nchunks, (long long)K * K, dWo); }, fl);
  { float* da; CK(hipMalloc(&da, (size_t)N * K * 4));
    run("bx3 dq + side job da = dT . qw (split W^T + product)", [&] {
      bx3_split_b(W, 1, K, K, K, img, st); Bx3Args p = dq_args(); p.side_in = qw; p.side_out = da; p.side_scale = dT; p.side_ld = K; p.side_cols = K;
      bx3_launch(p, st); }, fl);
    stamps("dq + side job", 256); }
  return 0;
}
