// tools/panelbench.hip -- dev microbenchmark (not product): the three panel-GEMM products of cfg 3
// (SimMatrix 16384 x 300 x 300) timed alone with HIP events, hipGraph-replayed.  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I mms_answer_selection_amd/csrc \
//         [-DMMS_PG_ABLATE=1] tools/panelbench.hip -o /tmp/panelbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "panel_gemm.h"
using namespace mms;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16384, K = 300;
  float *q, *a, *W, *qw, *dq, *top, *dT, *part, *part2;
  CK(hipMalloc(&q, (size_t)N * K * 4)); CK(hipMalloc(&a, (size_t)N * K * 4)); CK(hipMalloc(&W, (size_t)K * K * 4));
  CK(hipMalloc(&qw, (size_t)N * K * 4)); CK(hipMalloc(&dq, (size_t)N * K * 4)); CK(hipMalloc(&top, N * 4));
  CK(hipMalloc(&dT, N * 4)); CK(hipMalloc(&part2, (size_t)N * K * 4)); CK(hipMalloc(&part, (size_t)64 * K * K * 4));
  std::vector<float> h((size_t)N * K);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  CK(hipMemcpy(q, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)K * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dT, h.data(), N * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef MMS_PG_STAMPS
  unsigned long long* sb; CK(hipMalloc(&sb, 1024 * 8 * 8)); CK(hipMemset(sb, 0, 1024 * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(pg_stamp_buf), &sb, sizeof(sb)));
  auto dump = [&](const char* name) {
    std::vector<unsigned long long> h(1024 * 8);
    CK(hipMemcpy(h.data(), sb, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, loop_us, epi_us, wl, wb;
    for (int b = 0; b < 256; ++b) {
      const unsigned long long* x = &h[(size_t)b * 8];
      if (!x[0] || !x[2]) continue;
      clk.push_back((double)(x[2] - x[0]) / (double)(x[3] - x[1]) * 100.0);      // MHz: s_memrealtime ticks at 100 MHz
      loop_us.push_back((double)(x[3] - x[1]) / 100.0);
      epi_us.push_back((double)(x[5] - x[3]) / 100.0);
      wl.push_back((double)x[6]); wb.push_back((double)x[7]);
    }
    std::sort(clk.begin(), clk.end()); std::sort(loop_us.begin(), loop_us.end()); std::sort(epi_us.begin(), epi_us.end());
    {
      std::vector<double> a0, a1, a2;
      for (int b = 0; b < 512; ++b) {                 // loader stamps live at (gridDim.x + block): find non-zero triples past the compute ones
        const unsigned long long* x = &h[(size_t)b * 8];
        if (x[0] + x[1] + x[2] && !x[3] && !x[4] && !x[5] && !x[6] && !x[7]) { a0.push_back((double)x[0]); a1.push_back((double)x[1]); a2.push_back((double)x[2]); }
      }
      std::sort(a0.begin(), a0.end()); std::sort(a1.begin(), a1.end()); std::sort(a2.begin(), a2.end());
      if (!a0.empty()) printf("   loader wave 4 (cycles over tiles 1..): vmcnt wait median %.0f, barrier wait median %.0f, issue+side median %.0f\n", a0[a0.size() / 2], a1[a1.size() / 2], a2[a2.size() / 2]);
    }
    std::sort(wl.begin(), wl.end()); std::sort(wb.begin(), wb.end());
    if (!wl.empty()) printf("   in-loop waits of wave 0 (cycles over all tiles): lgkmcnt(0) before the barrier median %.0f, barrier median %.0f max %.0f\n", wl[wl.size() / 2], wb[wb.size() / 2], wb.back());
    if (!clk.empty())
      printf("   stamps %-30s shader clock median %.0f MHz (min %.0f max %.0f); main loop median %.2f us max %.2f; epilogue median %.2f us max %.2f\n",
             name, clk[clk.size() / 2], clk.front(), clk.back(), loop_us[loop_us.size() / 2], loop_us.back(),
             epi_us[epi_us.size() / 2], epi_us.back());
    CK(hipMemset(sb, 0, 1024 * 8 * 8));
  };
#else
  auto dump = [&](const char*) {};
#endif
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
    printf("%-34s median %8.2f us  min %8.2f   %6.1f TFLOP/s\n", name, t[2], t[0], flop / t[2] / 1e6);
    dump(name);
  };
  const double fl = 2.0 * N * K * K;

  run("fwd  Q.W + rowdot", [&] {
    PanelArgs p = panel_args(N, K, K, q, K, W, K, qw, K); p.Y = a; p.ldy = K; p.rowdot = top;
    panel_launch_t<19, true>(p, st); }, fl);
  run("dq   rowscale(A.Wt), streamed C", [&] {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 1;
    panel_launch_t<19, true>(p, st); }, fl);
  run("dq   rowscale(A.Wt), plain stores", [&] {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 0;
    panel_launch_t<19, true>(p, st); }, fl);
  run("dq + side job da = dT . qw", [&] {
    PanelArgs p = panel_args(N, K, K, a, K, W, K, dq, K); p.rowscale = dT; p.stream_c = 1;
    p.side_in = qw; p.side_out = part2; p.side_scale = dT; p.side_ld = K; p.side_cols = K;
    panel_launch_t<19, true>(p, st); }, fl);
  run("dW   Q^T diag(dT) A split-K", [&] {
    PanelArgs p = panel_args(K, K, N, q, K, a, K, part, K); p.kscale = dT;
    p.ksplit = panel_pick_ksplit(p.row_blocks, 1, N, &p.kchunk); p.c_ks = (long long)K * K;
    panel_launch_t<19, false>(p, st); }, fl);
  return 0;
}
