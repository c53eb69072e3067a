// tools/mfmastruct.hip -- dev microbenchmark (not product): which structural element of the panel GEMM's main loop
// costs matrix-pipe time?  Start from the stream tools/mfmarate.hip shows running at 32.2 cycles per MFMA (19 x
// v_mfma_f32_16x16x4_f32 per k-step with their 19 LDS operand reads interleaved, one wave per SIMD) and add, one at
// a time: the per-tile workgroup barrier, four partner waves that only take the barriers, VALU work in the partner
// waves (what round 2's loader waves spend on DMA addresses), real LDS-DMAs by the partners, a three-stage ring.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfmastruct.hip -o /tmp/mfmastruct && /tmp/mfmastruct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// FLAGS: 1 = barrier per tile (8 k-steps = 152 MFMAs); 2 = four partner waves (waves 4-7) taking the barriers;
// 4 = partners do 84 dependent-free VALU ops per tile at priority 3; 8 = partners issue 14 LDS-DMAs (1 KB each) per
// tile from an L2-resident buffer; 16 = operand reads walk a three-stage ring (stage = tile % 3) instead of one image;
// 32 = partners' VALU ops at priority 0; 64 = compute waves at priority 2; 128 = the compute wave itself issues 4 VALU
// ops per k-step (LDS address updates); 256 = accumulators ping-pong between two register sets (vDst != SrcC, as the
// compiler allocates them in the product kernels); 512 = A operand re-read from LDS every 4 k-steps (ds_read_b128);
// 1024 = the operand reads are hand-issued `ds_read_b32 v, vbase offset:IMM` from ONE per-stage base register (no
// address arithmetic in the loop at all), each MFMA behind `s_waitcnt lgkmcnt(18)` tied to its operand register
template <int FLAGS>
__global__ __launch_bounds__(512) void struct_kernel(float* out, unsigned long long* stamps, const float* src, int iters) {
  extern __shared__ float lds[];                       // 3 x 13376 floats
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr bool BAR = FLAGS & 1, PART = FLAGS & 2, PVALU = FLAGS & 4, PDMA = FLAGS & 8, RING = FLAGS & 16;
  constexpr int STAGE = 13376;
  for (int i = t; i < 3 * STAGE; i += blockDim.x) lds[i] = src[i & 0x3ffff];
  __syncthreads();
  if (wave >= 4) {
    if (!PART) return;
    if (!(FLAGS & 32)) __builtin_amdgcn_s_setprio(3);
    float x = lane * 0.5f, y = 1.0001f;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    for (int it = 0; it < iters; ++it) {
      if (BAR) asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory");
      if (PVALU) {
#pragma unroll
        for (int u = 0; u < 84; ++u) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
      }
      if (PDMA) {
        const unsigned sb = lds_base + (unsigned)(((it + 2) % 3) * STAGE) * 4u;
#pragma unroll
        for (int u = 0; u < 14; ++u) {
          const float* gp = src + ((wave - 4) * 14 + u) * 256 + lane * 4;
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gp), "s"(sb + (unsigned)(((wave - 4) * 14 + u) % 52) * 1024u) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (x == 12345.678f) out[t] = x;
    return;
  }
  if (FLAGS & 64) __builtin_amdgcn_s_setprio(2);
  float a = 0.5f + lane * 0.001f;
  v4f acc[19];
#pragma unroll
  for (int i = 0; i < 19; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  float bv[2][20];
#pragma unroll
  for (int i = 0; i < 20; ++i) { bv[0][i] = 0.25f; bv[1][i] = 0.25f; }
  const float* p0 = lds + (lane & 15) + 308 * 4 * (lane >> 4);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  v4f acc2[19];
#pragma unroll
  for (int i = 0; i < 19; ++i) acc2[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  int vx = lane;
  v4f a4 = (v4f){a, a, a, a};
  if (FLAGS & 1024) {
    // hand-issued stream: stage bases in three registers, everything else immediate
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    unsigned sbase[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) sbase[s3] = lds_base + (unsigned)(s3 * STAGE + (lane & 15) + 308 * 4 * (lane >> 4)) * 4u;
    const unsigned long long t0h = __builtin_amdgcn_s_memtime(), r0h = __builtin_amdgcn_s_memrealtime();
    // operand n of a tile (n = 19 st + tt) lives in slot n mod 38; its read is issued right behind MFMA n - 16, so
    // that in front of MFMA n exactly the reads n+1 .. n+15 are younger: `s_waitcnt lgkmcnt(15)` (a 4-bit counter)
    float bb[38];
#pragma unroll
    for (int i = 0; i < 38; ++i) bb[i] = 0.25f;
    for (int it = 0; it < iters; ++it) {
      const int s3 = it % 3, n3 = (it + 1) % 3;
      const unsigned base = RING ? (s3 == 0 ? sbase[0] : s3 == 1 ? sbase[1] : sbase[2]) : sbase[0];
      const unsigned nbase = RING ? (n3 == 0 ? sbase[0] : n3 == 1 ? sbase[1] : sbase[2]) : sbase[0];
#pragma unroll
      for (int n = 0; n < 152; ++n) {
        if (BAR && n == 133) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc[n % 19] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb[n % 38], acc[n % 19], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int m = n + 16;
        if (m < 152) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bb[m % 38]) : "v"(base), "n"((((m / 19) & 7) * 308 + 16 * (m % 19)) * 4) : "memory");
        else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bb[m % 38]) : "v"(nbase), "n"(((((m - 152) / 19) & 7) * 308 + 16 * ((m - 152) % 19)) * 4) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1h = __builtin_amdgcn_s_memtime(), r1h = __builtin_amdgcn_s_memrealtime();
    float sh = 0.f;
#pragma unroll
    for (int i = 0; i < 19; ++i) sh += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + bb[i] + bb[i + 19];
    if (sh == 12345.678f) out[t] = sh;
    if (lane == 0) {
      const size_t w = (size_t)blockIdx.x * 4 + wave;
      stamps[2 * w] = t1h - t0h;
      stamps[2 * w + 1] = r1h - r0h;
    }
    return;
  }
  for (int it = 0; it < iters; ++it) {
    const float* p = RING ? p0 + (it % 3) * STAGE : p0;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (BAR && st == 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int tt = 0; tt < 19; ++tt) bv[(st + 1) & 1][tt] = p[((st + 1) & 7) * 308 + 16 * tt];
      if ((FLAGS & 512) && (st & 3) == 3) a4 = *reinterpret_cast<const v4f*>(lds + 2 * STAGE + 40 * (lane & 15) + 4 * (lane >> 4) + 16 * (st >> 2));
      if (FLAGS & 128) {
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vx) : "v"(lane));
      }
      if (FLAGS & 256) {
        if (st & 1) {
#pragma unroll
          for (int tt = 0; tt < 19; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[st & 3], bv[st & 1][tt], acc2[tt], 0, 0, 0);
        } else {
#pragma unroll
          for (int tt = 0; tt < 19; ++tt) acc2[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[st & 3], bv[st & 1][tt], acc[tt], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int tt = 0; tt < 19; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[st & 3], bv[st & 1][tt], acc[tt], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 19; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc2[i][0] + acc2[i][3];
  if (s == 12345.678f) out[t] = s + vx;
  if (lane == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + wave;
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int FLAGS>
void run(const char* name, const float* src) {
  const int iters = 200, blocks = 256;
  float* out; unsigned long long* st;
  CK(hipMalloc(&out, 1 << 16)); CK(hipMalloc(&st, blocks * 4 * 16));
  auto kern = struct_kernel<FLAGS>;
  const size_t lds = 3 * 13376 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3((FLAGS & 2) ? 512 : 256), lds, 0, out, st, src, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * 4 * 2);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w] / iters / 152); clk.push_back((double)h[2 * w] / h[2 * w + 1] * 100.0); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  printf("%-92s %6.2f cycles/MFMA (median wave; min %.2f max %.2f)  clock %4.0f MHz  kernel %.1f us\n", name,
         cyc[cyc.size() / 2], cyc.front(), cyc.back(), clk[clk.size() / 2], ms * 1e3);
  CK(hipFree(out)); CK(hipFree(st));
}

int main() {
  float* src; CK(hipMalloc(&src, 1 << 20));
  { std::vector<float> h(1 << 18); for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f; CK(hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice)); }
  run<0>("stream: 19 MFMA + 19 LDS operand reads per k-step, 1 wave/SIMD", src);
  run<1>("+ barrier per tile (4 waves)", src);
  run<1 | 2>("+ four partner waves that only take the barriers", src);
  run<1 | 2 | 4>("+ partners: 84 VALU ops per tile, priority 3", src);
  run<1 | 2 | 4 | 32>("+ partners: 84 VALU ops per tile, priority 0", src);
  run<1 | 2 | 8>("+ partners: 14 LDS-DMAs per tile each (no VALU)", src);
  run<1 | 2 | 4 | 8>("+ partners: 84 VALU ops + 14 LDS-DMAs per tile, priority 3", src);
  run<1 | 2 | 4 | 8 | 16>("+ ... and the operand reads walk a three-stage ring", src);
  run<1 | 2 | 4 | 8 | 16 | 64>("+ ... compute waves at priority 2", src);
  run<16>("stream with the ring alone", src);
  run<128>("stream + 4 own VALU ops per k-step", src);
  run<256>("stream, accumulators ping-pong between two register sets", src);
  run<512>("stream + A operand re-read from LDS", src);
  run<128 | 256 | 512>("stream + own VALU + ping-pong + A re-read", src);
  run<1 | 2 | 8 | 16 | 128 | 256 | 512>("all of the above + barrier + partner DMAs (no partner VALU) + ring", src);
  run<1024>("HAND-ISSUED stream: ds_read_b32 base + immediate, waits tied to operands", src);
  run<1024 | 16>("hand-issued + ring (base selected per tile)", src);
  run<1024 | 16 | 1 | 2 | 8>("hand-issued + ring + barrier + partner DMAs (no partner VALU)", src);
  return 0;
}
