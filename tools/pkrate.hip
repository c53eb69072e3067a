// tools/pkrate.hip -- dev-only: issue cost of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32
// (wave64, inline asm so the compiler cannot re-vectorise either loop).
//   hipcc --offload-arch=gfx950 -O3 tools/pkrate.hip -o /tmp/pkrate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
  float2v a[8];
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (float2v){seed + i, seed - i}; s[i] = seed + i; }
  float2v inc = (float2v){seed, seed * 0.5f};
  float f = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(f));
        if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(inc));
        if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "v"(inc));
        if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(f));
        if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(inc));
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  float* out; CK(hipMalloc(&out, 4096 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4096, wgs = 2048;   // 8 workgroups per CU = 8 waves per SIMD
  const char* names[5] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32 (op_sel, neg)", "v_fma_f32", "v_pk_fma_f32"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(rate_kernel<0>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f); break;
        case 1: hipLaunchKernelGGL(rate_kernel<1>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f); break;
        case 2: hipLaunchKernelGGL(rate_kernel<2>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f); break;
        case 3: hipLaunchKernelGGL(rate_kernel<3>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f); break;
        default: hipLaunchKernelGGL(rate_kernel<4>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f); break;
      }
      CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)wgs * 4 * iters * 32 / 1024;   // wave-instructions per SIMD
    printf("%-28s %.3f ms: %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", names[mode], ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  }
  return 0;
}
