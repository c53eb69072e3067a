// tools/bx3struct.hip -- dev microbenchmark (not product): what one k-step of csrc/bx3_gemm.h costs, piece by piece.
// 256 workgroups of 8 waves (4 "compute" + 4 idle partners at the barrier), each compute wave runs STEPS steps of
// 30 v_mfma_f32_32x32x16_bf16 (5 accumulator tiles x 6) on random-ish operands:
//   MODE 0: MFMAs only (operands in registers)            1: + one s_barrier per step (8 waves)
//   MODE 2: + 17 ds_read_b128 per step (next step's frags) 3: 1 + 2                4: 3 + 60 VALU per step
//   MODE 5: 3 + the PARTNER wave of each SIMD issues 180 dependent-free VALU per step between the barriers
//   MODE 6: 3 + 180 VALU per step in the compute wave's OWN stream (6 behind each MFMA)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bx3struct.hip -o /tmp/bx3struct && /tmp/bx3struct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __bf16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* stamps, int steps, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 32768; i += 512) reinterpret_cast<unsigned*>(lds)[i] = (seed * 2654435761u + i * 40503u) & 0x3fff3fffu | 0x3c003c00u;
  __syncthreads();
  constexpr bool BAR = MODE == 1 || MODE >= 3, RD = MODE >= 2, VAL = MODE == 4, OWN = MODE == 6;
  if (wave >= 4) {
    float pv[12];
    for (int i = 0; i < 12; ++i) pv[i] = lane * 0.01f + i;
    if (BAR) for (int s = 0; s < steps; ++s) {
      if (MODE == 5) {
#pragma unroll
        for (int r = 0; r < 15; ++r)
#pragma unroll
          for (int i = 0; i < 12; ++i) pv[i] = pv[i] * 1.0001f + 0.5f;
      }
      __builtin_amdgcn_s_barrier();
    }
    if (MODE == 5) { float z = 0; for (int i = 0; i < 12; ++i) z += pv[i]; out[blockIdx.x * 256 + (threadIdx.x & 255)] = z; }
    return;
  }
  f16v acc[5];
  for (int t = 0; t < 5; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const u4* L = reinterpret_cast<const u4*>(lds) + lane;
  h8 a0 = __builtin_bit_cast(h8, L[0]), a1 = __builtin_bit_cast(h8, L[64]), a2 = __builtin_bit_cast(h8, L[128]);
  h8 b[5][3];
  for (int t = 0; t < 5; ++t) for (int p = 0; p < 3; ++p) b[t][p] = __builtin_bit_cast(h8, L[(3 * t + p) * 64 + 192]);
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
    const u4* Ls = L + ((s & 3) * 1472);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      __builtin_amdgcn_sched_barrier(0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[t][0], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[t][2], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[t][1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[t][0], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[t][1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[t][0], acc[t], 0, 0, 0);
      if (VAL) {
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i & 7] = v[i & 7] * 1.0001f + 0.5f;
#pragma unroll
        for (int g = 0; g < 6; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
      }
      if (OWN) {
#pragma unroll
        for (int i = 0; i < 36; ++i) v[i & 7] = v[i & 7] * 1.0001f + 0.5f;
#pragma unroll
        for (int g = 0; g < 6; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (RD) {
#pragma unroll
        for (int p = 0; p < 3; ++p) b[t][p] = __builtin_bit_cast(h8, Ls[(3 * t + p) * 64]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (BAR) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int t = 0; t < 5; ++t) for (int i = 0; i < 16; ++i) sum += acc[t][i];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* st, int steps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 131072, 0, out, st, steps, 7u + r);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(512);
  CK(hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, us;
  for (int b = 0; b < 256; ++b) { cyc.push_back((double)h[2 * b] / steps); us.push_back(h[2 * b + 1] / 100.0); }
  std::sort(cyc.begin(), cyc.end()); std::sort(us.begin(), us.end());
  printf("%-58s %7.1f cycles per step (30 MFMAs: %.2f per MFMA), %.2f us for %d steps, %.0f MHz\n", name, cyc[128], cyc[128] / 30,
         us[128], steps, cyc[128] * steps / us[128]);
}

int main() {
  float* out; unsigned long long* st;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&st, 512 * 8));
  const int steps = 190;
  run<0>("MFMAs only", out, st, steps);
  run<1>("+ barrier per step", out, st, steps);
  run<2>("+ 15 ds_read_b128 per step (no barrier)", out, st, steps);
  run<3>("+ barrier + reads", out, st, steps);
  run<4>("+ barrier + reads + 60 VALU", out, st, steps);
  run<5>("+ barrier + reads + 180 VALU per step in the PARTNER wave", out, st, steps);
  run<6>("+ barrier + reads + 180 VALU per step in the SAME wave", out, st, steps);
  run<0>("MFMAs only (again)", out, st, steps);
  return 0;
}
