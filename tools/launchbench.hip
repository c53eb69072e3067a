// tools/launchbench.hip -- dev microbenchmark (not part of the product or the tests):
// what bounds the Layer-API sequence "Forward launch, Backward launch" at cfg 2's footprint
// (q, a (4096,300) fp32)?  Pure data-movement kernels in several thread layouts, graph-replayed over
// an HBM-cold ring like bench.py, next to chains of EMPTY kernels of several grid shapes (the launch
// floor).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/launchbench.hip -o /tmp/launchbench && /tmp/launchbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int N = 4096, D = 300, D4 = 75;
typedef float v4f_nt __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sum4(float4 x, float4 y) { return (x.x - y.x) + (x.y - y.y) + (x.z - y.z) + (x.w - y.w); }

template <int T> __global__ __launch_bounds__(T) void empty_kernel(int n) {}

// ---- forward-like: read q, a; (almost) no output ------------------------------------------------
template <int T> __global__ __launch_bounds__(T) void f_flat(const float4* __restrict__ q, const float4* __restrict__ a,
                                                            float* __restrict__ out, int n4) {
  int i = blockIdx.x * T + threadIdx.x;
  float s = 0.f;
  if (i < n4) s = sum4(q[i], a[i]);
  if (s == 12345.678f) out[i] = s;
}
// pair-structured: 32 lanes per row, two rows per wave, WPB waves per workgroup (the library's layout)
template <int WPB> __global__ __launch_bounds__(64 * WPB) void f_pair(const float4* __restrict__ q, const float4* __restrict__ a,
                                                                     float* __restrict__ top, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = min((blockIdx.x * WPB + wave) * 2 + (lane >> 5), n - 1), j = lane & 31;
  const size_t b = (size_t)row * D4;
  float4 x[3], y[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) { int i = j + 32 * it; int ii = i < D4 ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < 3; ++it) s += sum4(x[it], y[it]);
  for (int o = 16; o; o >>= 1) s += __shfl_xor(s, o);
  if (j == 0) top[row] = s;
}
// workgroup-blocked linear: a workgroup of T threads owns R rows = R*75 consecutive float4 per operand
template <int T, int R> __global__ __launch_bounds__(T) void f_block(const float4* __restrict__ q, const float4* __restrict__ a,
                                                                    float* __restrict__ top, int n) {
  constexpr int C = R * D4, NIT = (C + T - 1) / T;
  const size_t b = (size_t)blockIdx.x * C;
  float4 x[NIT], y[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) { int i = threadIdx.x + T * it; int ii = i < C ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) s += sum4(x[it], y[it]);
  if (s == 12345.678f) top[blockIdx.x] = s;
}

// ---- backward-like: read q, a, top, dT; write dq, da (streaming stores) -------------------------
template <int T, bool NT> __global__ __launch_bounds__(T) void b_flat(const float4* __restrict__ q, const float4* __restrict__ a,
                                                                     const float* __restrict__ top, const float* __restrict__ dT,
                                                                     float4* __restrict__ dq, float4* __restrict__ da, int n4) {
  int i = blockIdx.x * T + threadIdx.x;
  if (i >= n4) return;
  float4 x = q[i], y = a[i];
  const int row = i / D4;
  const float c = top[row] * dT[row];
  v4f_nt o0 = {c * (x.x - y.x), c * (x.y - y.y), c * (x.z - y.z), c * (x.w - y.w)};
  v4f_nt o1 = {-o0.x, -o0.y, -o0.z, -o0.w};
  if (NT) { __builtin_nontemporal_store(o0, (v4f_nt*)(dq + i)); __builtin_nontemporal_store(o1, (v4f_nt*)(da + i)); }
  else { *(v4f_nt*)(dq + i) = o0; *(v4f_nt*)(da + i) = o1; }
}
template <int WPB> __global__ __launch_bounds__(64 * WPB) void b_pair(const float4* __restrict__ q, const float4* __restrict__ a,
                                                                     const float* __restrict__ top, const float* __restrict__ dT,
                                                                     float4* __restrict__ dq, float4* __restrict__ da, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = min((blockIdx.x * WPB + wave) * 2 + (lane >> 5), n - 1), j = lane & 31;
  const size_t b = (size_t)row * D4;
  float4 x[3], y[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) { int i = j + 32 * it; int ii = i < D4 ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
  const float c = top[row] * dT[row];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    int i = j + 32 * it;
    if (i < D4) {
      v4f_nt o0 = {c * (x[it].x - y[it].x), c * (x[it].y - y[it].y), c * (x[it].z - y[it].z), c * (x[it].w - y[it].w)};
      v4f_nt o1 = {-o0.x, -o0.y, -o0.z, -o0.w};
      __builtin_nontemporal_store(o0, (v4f_nt*)(dq + b + i));
      __builtin_nontemporal_store(o1, (v4f_nt*)(da + b + i));
    }
  }
}
template <int T, int R> __global__ __launch_bounds__(T) void b_block(const float4* __restrict__ q, const float4* __restrict__ a,
                                                                    const float* __restrict__ top, const float* __restrict__ dT,
                                                                    float4* __restrict__ dq, float4* __restrict__ da, int n) {
  constexpr int C = R * D4, NIT = (C + T - 1) / T;
  const size_t b = (size_t)blockIdx.x * C;
  float4 x[NIT], y[NIT];
  float c[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) { int i = threadIdx.x + T * it; int ii = i < C ? i : 0; x[it] = q[b + ii]; y[it] = a[b + ii]; }
#pragma unroll
  for (int it = 0; it < NIT; ++it) { int i = threadIdx.x + T * it; int row = blockIdx.x * R + (i < C ? i : 0) / D4; c[it] = top[row] * dT[row]; }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int i = threadIdx.x + T * it;
    if (i < C) {
      v4f_nt o0 = {c[it] * (x[it].x - y[it].x), c[it] * (x[it].y - y[it].y), c[it] * (x[it].z - y[it].z), c[it] * (x[it].w - y[it].w)};
      v4f_nt o1 = {-o0.x, -o0.y, -o0.z, -o0.w};
      __builtin_nontemporal_store(o0, (v4f_nt*)(dq + b + i));
      __builtin_nontemporal_store(o1, (v4f_nt*)(da + b + i));
    }
  }
}

struct Slot { float *q, *a, *dT, *top, *dq, *da; };

int main(int argc, char** argv) {
  const int ring = 64, G = 16, reps = 64;
  const bool only_empty = argc > 1 && atoi(argv[1]) == 1;
  std::vector<Slot> s(ring);
  const size_t nb = (size_t)N * D * sizeof(float);
  {
    std::vector<float> h((size_t)N * D);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& x : s) {
      CK(hipMalloc(&x.q, nb)); CK(hipMalloc(&x.a, nb)); CK(hipMalloc(&x.dq, nb)); CK(hipMalloc(&x.da, nb));
      CK(hipMalloc(&x.dT, N * 4)); CK(hipMalloc(&x.top, N * 4));
      CK(hipMemcpy(x.q, h.data(), nb, hipMemcpyHostToDevice));
      CK(hipMemcpy(x.a, h.data() + 7, nb - 28, hipMemcpyHostToDevice));
      CK(hipMemcpy(x.dT, h.data(), N * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(x.top, h.data() + 11, N * 4, hipMemcpyHostToDevice));
    }
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n4 = N * D4;

  auto run = [&](const char* name, auto&& body, double bytes, bool both = true) {
    for (int warm = 0; warm < (both ? 2 : 1); ++warm) {
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
      printf("%-40s %-5s median %7.3f us/step  min %7.3f   %7.1f GB/s\n", name, warm ? "warm" : "cold",
             times[2], times[0], bytes / (times[2] * 1e-6) / 1e9);
      for (auto ge : gs) CK(hipGraphExecDestroy(ge));
    }
    fflush(stdout);
  };
#define EMPTY(G_, T_) run("empty x1 grid " #G_ " x " #T_, [&](Slot& x, hipStream_t t) { \
    hipLaunchKernelGGL(empty_kernel<T_>, dim3(G_), dim3(T_), 0, t, N); }, 0.0, false);
  EMPTY(1, 64) EMPTY(256, 64) EMPTY(256, 256) EMPTY(256, 512) EMPTY(256, 1024) EMPTY(1200, 256) EMPTY(4800, 64) EMPTY(2048, 512)
  if (only_empty) return 0;

#define Q4 (const float4*)x.q
#define A4 (const float4*)x.a
  // forward-like
  run("F flat 256", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(f_flat<256>, dim3((n4 + 255) / 256), dim3(256), 0, t, Q4, A4, x.top, n4); }, 2.0 * nb);
  run("F flat 512", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(f_flat<512>, dim3((n4 + 511) / 512), dim3(512), 0, t, Q4, A4, x.top, n4); }, 2.0 * nb);
  run("F flat 1024", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(f_flat<1024>, dim3((n4 + 1023) / 1024), dim3(1024), 0, t, Q4, A4, x.top, n4); }, 2.0 * nb);
  run("F flat 64", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(f_flat<64>, dim3((n4 + 63) / 64), dim3(64), 0, t, Q4, A4, x.top, n4); }, 2.0 * nb);
  run("F pair 8 waves (library layout)", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(f_pair<8>, dim3(N / 16), dim3(512), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  run("F pair 4 waves", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(f_pair<4>, dim3(N / 8), dim3(256), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  run("F pair 1 wave", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(f_pair<1>, dim3(N / 2), dim3(64), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  run("F block 512 thr x 16 rows", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((f_block<512, 16>), dim3(N / 16), dim3(512), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  run("F block 256 thr x 8 rows", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((f_block<256, 8>), dim3(N / 8), dim3(256), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  run("F block 256 thr x 4 rows", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((f_block<256, 4>), dim3(N / 4), dim3(256), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  run("F block 128 thr x 2 rows", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((f_block<128, 2>), dim3(N / 2), dim3(128), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  run("F block 1024 thr x 16 rows", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((f_block<1024, 16>), dim3(N / 16), dim3(1024), 0, t, Q4, A4, x.top, N); }, 2.0 * nb);
  // backward-like alone (reads HBM-cold)
  run("B flat 256 nt", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((b_flat<256, true>), dim3((n4 + 255) / 256), dim3(256), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, n4); }, 4.0 * nb);
  run("B flat 256 plain stores", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((b_flat<256, false>), dim3((n4 + 255) / 256), dim3(256), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, n4); }, 4.0 * nb);
  run("B pair 8 waves (library layout)", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL(b_pair<8>, dim3(N / 16), dim3(512), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, N); }, 4.0 * nb);
  run("B block 512 thr x 16 rows", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((b_block<512, 16>), dim3(N / 16), dim3(512), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, N); }, 4.0 * nb);
  run("B block 256 thr x 4 rows", [&](Slot& x, hipStream_t t) { hipLaunchKernelGGL((b_block<256, 4>), dim3(N / 4), dim3(256), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, N); }, 4.0 * nb);
  // the sequence
  run("SEQ F flat 256 + B flat 256 nt", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(f_flat<256>, dim3((n4 + 255) / 256), dim3(256), 0, t, Q4, A4, x.top, n4);
    hipLaunchKernelGGL((b_flat<256, true>), dim3((n4 + 255) / 256), dim3(256), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, n4); }, 6.0 * nb);
  run("SEQ F pair 8 + B pair 8", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(f_pair<8>, dim3(N / 16), dim3(512), 0, t, Q4, A4, x.top, N);
    hipLaunchKernelGGL(b_pair<8>, dim3(N / 16), dim3(512), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, N); }, 6.0 * nb);
  run("SEQ F block 256x4 + B block 256x4", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL((f_block<256, 4>), dim3(N / 4), dim3(256), 0, t, Q4, A4, x.top, N);
    hipLaunchKernelGGL((b_block<256, 4>), dim3(N / 4), dim3(256), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, N); }, 6.0 * nb);
  run("SEQ F flat 256 + B block 256x4", [&](Slot& x, hipStream_t t) {
    hipLaunchKernelGGL(f_flat<256>, dim3((n4 + 255) / 256), dim3(256), 0, t, Q4, A4, x.top, n4);
    hipLaunchKernelGGL((b_block<256, 4>), dim3(N / 4), dim3(256), 0, t, Q4, A4, x.top, x.dT, (float4*)x.dq, (float4*)x.da, N); }, 6.0 * nb);
  return 0;
}
