// tools/membench.hip -- dev microbenchmark (not part of the product or the tests):
// what does a pure copy with the SAME footprint as cfg 2 (read q,a 4096x300 fp32,
// write dq,da) achieve on this box, HBM-cold (ring of batches) and cache-warm,
// next to the library's own entry points?  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -I include tools/membench.hip -o /tmp/membench \
//         -L mms_answer_selection_amd -lmms_hip -Wl,-rpath,$PWD/mms_answer_selection_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "mms.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int N = 4096, D = 300, D4 = 75;

__global__ __launch_bounds__(256) void copy2_kernel(const float4* __restrict__ q, const float4* __restrict__ a,
                                                    float4* __restrict__ dq, float4* __restrict__ da, int n4) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n4) { float4 x = q[i], y = a[i]; dq[i] = x; da[i] = y; }
}
// same per-wave structure as the library kernel: a wave owns 150 float4 per operand, 3 loads per lane
__global__ __launch_bounds__(256) void copy2_wave_kernel(const float4* __restrict__ q, const float4* __restrict__ a,
                                                         float4* __restrict__ dq, float4* __restrict__ da, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + wave) * 2;
  if (row0 >= n) return;
  const size_t b = (size_t)row0 * D4;
  float4 x[3], y[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) { int i = lane + 64 * it; int ii = i < 150 ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
#pragma unroll
  for (int it = 0; it < 3; ++it) { int i = lane + 64 * it; if (i < 150) { dq[b + i] = x[it]; da[b + i] = y[it]; } }
}
// the same copy with non-temporal (streaming) stores, 8 waves per workgroup like the library kernel
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void copy2_wave_nt_kernel(const float4* __restrict__ q, const float4* __restrict__ a,
                                                            float4* __restrict__ dq, float4* __restrict__ da, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = min((blockIdx.x * 8 + wave) * 2 + (lane >> 5), n - 1), j = lane & 31;
  const size_t b = (size_t)row * D4;
  float4 x[3], y[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) { int i = j + 32 * it; int ii = i < D4 ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    int i = j + 32 * it;
    if (i < D4) {
      __builtin_nontemporal_store((v4f_nt){x[it].x, x[it].y, x[it].z, x[it].w}, (v4f_nt*)(dq + b + i));
      __builtin_nontemporal_store((v4f_nt){y[it].x, y[it].y, y[it].z, y[it].w}, (v4f_nt*)(da + b + i));
    }
  }
}
__global__ __launch_bounds__(256) void read2_kernel(const float4* __restrict__ q, const float4* __restrict__ a,
                                                    float* __restrict__ out, int n4) {
  int i = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if (i < n4) { float4 x = q[i], y = a[i]; s = x.x + y.x + x.y + y.y + x.z + y.z + x.w + y.w; }
  if (s == 12345.678f) out[i] = s;
}

__global__ __launch_bounds__(512) void empty_kernel(int n) {}
// read q, a (cold), write one float per pair: the data movement of a forward-only launch
__global__ __launch_bounds__(512) void read2_pair_kernel(const float4* __restrict__ q, const float4* __restrict__ a,
                                                         float* __restrict__ top, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = min((blockIdx.x * 8 + wave) * 2 + (lane >> 5), n - 1), j = lane & 31;
  const size_t b = (size_t)row * D4;
  float4 x[3], y[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) { int i = j + 32 * it; int ii = i < D4 ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < 3; ++it) s += (x[it].x - y[it].x) + (x[it].y - y[it].y) + (x[it].z - y[it].z) + (x[it].w - y[it].w);
  for (int o = 16; o; o >>= 1) s += __shfl_xor(s, o);
  if (j == 0) top[row] = s;
}
// read q, a (just read by the launch before: L2 / Infinity Cache), top, dT; write dq, da with streaming stores:
// the data movement of a backward-only launch
__global__ __launch_bounds__(512) void bwd_like_kernel(const float4* __restrict__ q, const float4* __restrict__ a,
                                                       const float* __restrict__ top, const float* __restrict__ dT,
                                                       float4* __restrict__ dq, float4* __restrict__ da, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = min((blockIdx.x * 8 + wave) * 2 + (lane >> 5), n - 1), j = lane & 31;
  const size_t b = (size_t)row * D4;
  float4 x[3], y[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) { int i = j + 32 * it; int ii = i < D4 ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
  const float c = top[row] * dT[row];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    int i = j + 32 * it;
    if (i < D4) {
      __builtin_nontemporal_store((v4f_nt){c * (x[it].x - y[it].x), c * (x[it].y - y[it].y), c * (x[it].z - y[it].z), c * (x[it].w - y[it].w)}, (v4f_nt*)(dq + b + i));
      __builtin_nontemporal_store((v4f_nt){c * (y[it].x - x[it].x), c * (y[it].y - x[it].y), c * (y[it].z - x[it].z), c * (y[it].w - x[it].w)}, (v4f_nt*)(da + b + i));
    }
  }
}

struct Slot { float *q, *a, *dT, *top, *dq, *da; };

int main(int argc, char** argv) {
  const int ring = 64, G = 16, reps = 64;
  std::vector<Slot> s(ring);
  const size_t nb = (size_t)N * D * sizeof(float);
  for (auto& x : s) {
    CK(hipMalloc(&x.q, nb)); CK(hipMalloc(&x.a, nb)); CK(hipMalloc(&x.dq, nb)); CK(hipMalloc(&x.da, nb));
    CK(hipMalloc(&x.dT, N * 4)); CK(hipMalloc(&x.top, N * 4));
    std::vector<float> h((size_t)N * D);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(x.q, h.data(), nb, hipMemcpyHostToDevice));
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(x.a, h.data(), nb, hipMemcpyHostToDevice));
    CK(hipMemcpy(x.dT, h.data(), N * 4, hipMemcpyHostToDevice));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n4 = N * D4;

  auto run = [&](const char* name, auto&& body, double bytes) {
    for (int warm = 0; warm < 2; ++warm) {
      const int nslots = warm ? 1 : ring;
      std::vector<hipGraphExec_t> gs;
      for (int g0 = 0; g0 < nslots; g0 += G) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < G; ++k) body(s[(g0 + k) % nslots], st);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        gs.push_back(ge);
      }
      for (size_t i = 0; i < gs.size() * 2; ++i) CK(hipGraphLaunch(gs[i % gs.size()], st));
      CK(hipStreamSynchronize(st));
      std::vector<float> times;
      for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(gs[i % gs.size()], st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        times.push_back(ms * 1e3f / (reps * G));
      }
      std::sort(times.begin(), times.end());
      printf("%-28s %-5s median %7.3f us/step  min %7.3f   %7.1f GB/s (bytes %.1f MB)\n", name, warm ? "warm" : "cold",
             times[2], times[0], bytes / (times[2] * 1e-6) / 1e9, bytes / 1e6);
    }
  };

  // EAGER launches (no hipGraph) of the Layer-API pair for 20 steps, the host running ahead of the device behind an
  // untimed lead-in: what a native host that launches kernel by kernel sees (no 6.5-us graph-launch boundary)
  if (argc > 1 && atoi(argv[1]) == 2) {
    auto pair = [&](Slot& x) {
      mms_simcross_forward_f32(1, N, 1, 1, D, 1, x.q, x.a, nullptr, nullptr, x.top, nullptr, nullptr, nullptr, 0, st);
      mms_simcross_backward_f32(1, N, 1, 1, D, 1, x.q, x.a, nullptr, 0, x.top, x.dT, nullptr, nullptr, 1, 1, x.dq, x.da, nullptr, nullptr, nullptr, 0, st);
    };
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 32; ++k) pair(s[32 + k]);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int K : {20, 64, 256}) {
      std::vector<float> times;
      for (int rep = 0; rep < 7; ++rep) {
        CK(hipStreamSynchronize(st));
        for (int i = 0; i < 8; ++i) CK(hipGraphLaunch(ge, st));      // ~2 ms of lead-in
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < K; ++k) pair(s[(rep * 7 + k) % 32]);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        times.push_back(ms * 1e3f / K);
      }
      std::sort(times.begin(), times.end());
      printf("EAGER mms fwd + mms bwd, %3d steps per region: median %7.3f us/step  min %7.3f\n", K, times[3], times[0]);
    }
    return 0;
  }
  run("copy2 (1 float4/thread)", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(copy2_kernel, dim3((n4 + 255) / 256), dim3(256), 0, t, (const float4*)x.q, (const float4*)x.a, (float4*)x.dq, (float4*)x.da, n4); }, 4.0 * nb);
  run("copy2 pair-structured, nt stores", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(copy2_wave_nt_kernel, dim3((N + 15) / 16), dim3(512), 0, t, (const float4*)x.q, (const float4*)x.a, (float4*)x.dq, (float4*)x.da, N); }, 4.0 * nb);
  run("copy2 wave-structured", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(copy2_wave_kernel, dim3((N + 7) / 8), dim3(256), 0, t, (const float4*)x.q, (const float4*)x.a, (float4*)x.dq, (float4*)x.da, N); }, 4.0 * nb);
  run("read2", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(read2_kernel, dim3((n4 + 255) / 256), dim3(256), 0, t, (const float4*)x.q, (const float4*)x.a, x.top, n4); }, 2.0 * nb);
  run("mms fwd+bwd fused", [&](Slot& x, hipStream_t t) {
    mms_simcross_forward_backward_f32(1, N, 1, 1, D, 1, x.q, x.a, nullptr, nullptr, x.dT, x.top, nullptr, nullptr, x.dq, x.da, nullptr, nullptr, nullptr, 0, t); }, 4.0 * nb);
  run("mms fwd", [&](Slot& x, hipStream_t t) {
    mms_simcross_forward_f32(1, N, 1, 1, D, 1, x.q, x.a, nullptr, nullptr, x.top, nullptr, nullptr, nullptr, 0, t); }, 2.0 * nb);
  run("mms bwd", [&](Slot& x, hipStream_t t) {
    mms_simcross_backward_f32(1, N, 1, 1, D, 1, x.q, x.a, nullptr, 0, x.top, x.dT, nullptr, nullptr, 1, 1, x.dq, x.da, nullptr, nullptr, nullptr, 0, t); }, 4.0 * nb);
  run("2 empty launches", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, t, N);
    hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, t, N); }, 0.0);
  run("read2 pair-structured", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(read2_pair_kernel, dim3((N + 15) / 16), dim3(512), 0, t, (const float4*)x.q, (const float4*)x.a, x.top, N); }, 2.0 * nb);
  run("SEQ read2 + bwd-like (2 launches)", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(read2_pair_kernel, dim3((N + 15) / 16), dim3(512), 0, t, (const float4*)x.q, (const float4*)x.a, x.top, N);
    hipLaunchKernelGGL(bwd_like_kernel, dim3((N + 15) / 16), dim3(512), 0, t, (const float4*)x.q, (const float4*)x.a, x.top, x.dT, (float4*)x.dq, (float4*)x.da, N); }, 6.0 * nb);
  run("SEQ mms fwd + mms bwd (2 launches)", [&](Slot& x, hipStream_t t) {
    mms_simcross_forward_f32(1, N, 1, 1, D, 1, x.q, x.a, nullptr, nullptr, x.top, nullptr, nullptr, nullptr, 0, t);
    mms_simcross_backward_f32(1, N, 1, 1, D, 1, x.q, x.a, nullptr, 0, x.top, x.dT, nullptr, nullptr, 1, 1, x.dq, x.da, nullptr, nullptr, nullptr, 0, t); }, 6.0 * nb);
  return 0;
}
