// tools/chainbench.hip -- dev microbenchmark: cycles per DEPENDENT v_add_f32
// on gfx950 under different conditions (what the bit-exact d-ascending chain
// of the Euclidean forward is bound by).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ACTIVE>
__global__ void chain_regs(const float* in, float* out, unsigned long long* cyc, int reps) {
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = in[i];
  float s = in[32];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if ((int)(threadIdx.x & 63) < ACTIVE) {
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 32; ++i) s += v[i];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef float float2v __attribute__((ext_vector_type(2)));
// packed: one v_pk_add_f32 advances TWO independent chains per lane
template <int ACTIVE>
__global__ void chain_regs_pk(const float* in, float* out, unsigned long long* cyc, int reps) {
  float2v v[32];
  for (int i = 0; i < 32; ++i) { v[i].x = in[i]; v[i].y = in[i + 1]; }
  float2v s; s.x = in[32]; s.y = in[33];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if ((int)(threadIdx.x & 63) < ACTIVE) {
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 32; ++i) s = s + v[i];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef float v4f_t __attribute__((ext_vector_type(4)));
// dependent v_mfma_f32_16x16x4_f32 with B == 1: four sequential fp32 adds per instruction
__global__ void chain_mfma(const float* in, float* out, unsigned long long* cyc, int reps) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = in[(threadIdx.x + i) & 31];
  v4f_t acc = {in[32], in[33], in[34], in[35]};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], 1.0f, acc, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void chain_lds(const float* in, float* out, unsigned long long* cyc, int D4) {
  __shared__ float4 sq[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 2 * D4; i += 64) { float x = in[i % 33]; sq[wave][i] = make_float4(x, x, x, x); }
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float dist = 0.f;
  if (lane < 2) {
    const float4* r4 = &sq[wave][lane * D4];
    for (int d = 0; d < D4; ++d) { float4 v = r4[d]; dist += v.x; dist += v.y; dist += v.z; dist += v.w; }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = dist;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float *in, *out; unsigned long long* cyc;
  CK(hipMalloc(&in, 64 * 4)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 8192 * 8));
  std::vector<float> h(64, 1.0f / 3); CK(hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice));
  std::vector<unsigned long long> c(8192);
  auto report = [&](const char* name, int blocks, double adds) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < blocks; ++i) s += c[i];
    // s_memtime counts at 100 MHz on gfx9 (constant clock): report ns
    printf("%-44s blocks %5d: %8.1f ticks/wave = %6.2f ns per dependent add\n", name, blocks, s / blocks, s / blocks * 10.0 / adds);
  };
  const int reps = 10;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(chain_regs<64>, dim3(1), dim3(64), 0, 0, in, out, cyc, reps); report("regs, 1 wave on chip, 64 lanes", 1, 320);
    hipLaunchKernelGGL(chain_regs<2>, dim3(1), dim3(64), 0, 0, in, out, cyc, reps); report("regs, 1 wave on chip, 2 lanes", 1, 320);
    hipLaunchKernelGGL(chain_regs<2>, dim3(256), dim3(256), 0, 0, in, out, cyc, reps); report("regs, 1 wave/SIMD all CUs, 2 lanes", 256, 320);
    hipLaunchKernelGGL(chain_regs<2>, dim3(512), dim3(256), 0, 0, in, out, cyc, reps); report("regs, 2 waves/SIMD all CUs, 2 lanes", 512, 320);
    hipLaunchKernelGGL(chain_regs<2>, dim3(1024), dim3(256), 0, 0, in, out, cyc, reps); report("regs, 4 waves/SIMD all CUs, 2 lanes", 1024, 320);
    hipLaunchKernelGGL(chain_regs_pk<64>, dim3(1), dim3(64), 0, 0, in, out, cyc, reps); report("PACKED regs, 1 wave on chip", 1, 320);
    hipLaunchKernelGGL(chain_regs_pk<64>, dim3(256), dim3(256), 0, 0, in, out, cyc, reps); report("PACKED regs, 1 wave/SIMD all CUs", 256, 320);
    hipLaunchKernelGGL(chain_regs_pk<64>, dim3(512), dim3(256), 0, 0, in, out, cyc, reps); report("PACKED regs, 2 waves/SIMD all CUs", 512, 320);
    hipLaunchKernelGGL(chain_mfma, dim3(1), dim3(64), 0, 0, in, out, cyc, 40); report("MFMA 16x16x4 dependent, 1 wave (per MFMA = 4 adds)", 1, 320);
    hipLaunchKernelGGL(chain_mfma, dim3(256), dim3(256), 0, 0, in, out, cyc, 40); report("MFMA 16x16x4 dependent, 1 wave/SIMD all CUs", 256, 320);
    hipLaunchKernelGGL(chain_mfma, dim3(512), dim3(256), 0, 0, in, out, cyc, 40); report("MFMA 16x16x4 dependent, 2 waves/SIMD all CUs", 512, 320);
    hipLaunchKernelGGL(chain_lds, dim3(1), dim3(64), 0, 0, in, out, cyc, 75); report("lds-fed D4=75, 1 wave", 1, 300);
    hipLaunchKernelGGL(chain_lds, dim3(512), dim3(256), 0, 0, in, out, cyc, 75); report("lds-fed D4=75, 2 waves/SIMD all CUs", 512, 300);
  }
  return 0;
}
