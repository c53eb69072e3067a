// tools/mfmarate.hip -- dev microbenchmark (not product): issue cadence of the fp32 MFMAs the panel GEMM is
// built from, one wave per SIMD on every CU, with and without LDS operand reads beside them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfmarate.hip -o /tmp/mfmarate && /tmp/mfmarate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// MODE 0: 19 x 16x16x4, operands in registers.  1: + one ds_read2_b32-style pair of LDS reads per two MFMAs,
// consumed one step later.  2: 5 x 32x32x2 in registers.  3: 5 x 32x32x2 + one LDS read per MFMA.
// 4: 19 x 16x16x4 with one ds_read_b128 per four MFMAs.
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rate_kernel(float* out, unsigned long long* stamps, int iters) {
  __shared__ float lds[8192];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 8192; i += 64 * WAVES) lds[i] = 0.001f * (i & 15);
  __syncthreads();
  float a = 0.5f + lane * 0.001f, b = 0.25f;
  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  if (MODE == 0 || MODE == 1 || MODE == 4) {
    v4f acc[19];
#pragma unroll
    for (int i = 0; i < 19; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float bv[2][20];
#pragma unroll
    for (int i = 0; i < 20; ++i) { bv[0][i] = b; bv[1][i] = b; }
    const float* p = lds + (lane & 15) + 308 * 4 * (lane >> 4);
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        if (MODE == 1) {
#pragma unroll
          for (int tt = 0; tt < 19; ++tt) bv[(st + 1) & 1][tt] = p[st * 308 + 16 * tt];
        }
        if (MODE == 4 && (st & 3) == 0) {
#pragma unroll
          for (int tt = 0; tt < 5; ++tt) {
            const v4f x = *reinterpret_cast<const v4f*>(lds + 4 * (lane & 15) + 64 * tt + 1024 * (st >> 2) + 40 * 4 * (lane >> 4));
            bv[(st + 1) & 1][4 * tt] = x[0]; bv[(st + 1) & 1][4 * tt + 1] = x[1];
            bv[(st + 1) & 1][4 * tt + 2] = x[2]; bv[(st + 1) & 1][4 * tt + 3] = x[3];
          }
        }
#pragma unroll
        for (int tt = 0; tt < 19; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[st & 1][tt], acc[tt], 0, 0, 0);
        if (MODE == 1) {
#pragma unroll
          for (int u = 0; u < 10; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 19; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[t] = s;
  } else {
    v16f acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float bv[2][5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { bv[0][i] = b; bv[1][i] = b; }
    const float* p = lds + (lane & 31) + 164 * (lane >> 5);
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int st = 0; st < 16; ++st) {       // 16 k-steps of 2 = the same 32-deep tile
        if (MODE == 3) {
#pragma unroll
          for (int tt = 0; tt < 5; ++tt) bv[(st + 1) & 1][tt] = p[st * 328 + 32 * tt];
        }
#pragma unroll
        for (int tt = 0; tt < 5; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[st & 1][tt], acc[tt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[t] = s;
  }
  if (lane == 0) {
    const size_t w = (size_t)blockIdx.x * WAVES + (t >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int MODE, int WAVES>
void run(const char* name, double macs_per_iter, int nmfma_per_iter) {
  const int iters = 200, blocks = 256;
  float* out; unsigned long long* st;
  CK(hipMalloc(&out, 1 << 16)); CK(hipMalloc(&st, blocks * WAVES * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * WAVES * 2);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (int w = 0; w < blocks * WAVES; ++w) { cyc.push_back((double)h[2 * w] / iters / nmfma_per_iter); clk.push_back((double)h[2 * w] / h[2 * w + 1] * 100.0); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double tf = 2.0 * macs_per_iter * iters * blocks * WAVES / (ms * 1e-3) / 1e12;
  printf("%-58s %6.2f cycles/MFMA (median wave; min %.2f max %.2f)  clock %4.0f MHz  %6.1f TFLOP/s  kernel %.1f us\n", name,
         cyc[cyc.size() / 2], cyc.front(), cyc.back(), clk[clk.size() / 2], tf, ms * 1e3);
}

int main() {
  run<0, 4>("19 x 16x16x4 f32, operands in registers, 1 wave/SIMD", 19.0 * 8 * 1024, 152);
  run<1, 4>("19 x 16x16x4 + 19 ds_read_b32 per step (interleaved)", 19.0 * 8 * 1024, 152);
  run<4, 4>("19 x 16x16x4 + 5 ds_read_b128 per four steps", 19.0 * 8 * 1024, 152);
  run<2, 4>("5 x 32x32x2 f32, operands in registers, 1 wave/SIMD", 5.0 * 16 * 2048, 80);
  run<3, 4>("5 x 32x32x2 + 5 ds_read_b32 per step", 5.0 * 16 * 2048, 80);
  run<0, 8>("19 x 16x16x4 registers, 2 waves/SIMD", 19.0 * 8 * 1024, 152);
  run<1, 8>("19 x 16x16x4 + LDS reads, 2 waves/SIMD", 19.0 * 8 * 1024, 152);
  return 0;
}
