// tools/dmarate.hip -- dev microbenchmark (not product): how fast can ONE CU pull a k-tile stream into LDS
// (LDS-DMA) or registers, when every CU reads the SAME bytes (the weight matrix of the panel GEMM: L2 hits,
// same lines wanted by 32 CUs of an XCD at once) or its own bytes; 4 or 8 loader waves; 1 KiB per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dmarate.hip -o /tmp/dmarate && /tmp/dmarate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

// Each wave issues PER instructions per "tile", waits until only the newest PER are outstanding, repeats TILES times.
// SHARED: every workgroup reads the same 360 KB region (tile t at offset t * tile bytes, wrapping); else its own region.
template <int WAVES, int PER, bool SHARED, bool TO_REGS>
__global__ __launch_bounds__(64 * WAVES) void dma_kernel(const float* src, size_t region_floats, float* out,
                                                         unsigned long long* stamps, int tiles) {
  extern __shared__ float4 lds4[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const float* base = SHARED ? src : src + (size_t)blockIdx.x * region_floats;
  const size_t tile_floats = (size_t)WAVES * PER * 256;        // floats per tile (all waves)
  const size_t ntile_region = region_floats / tile_floats;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int T = 0; T < tiles; ++T) {
    const float* tb = base + (size_t)(T % ntile_region) * tile_floats;
    const unsigned stage = (unsigned)(T % 3) * (unsigned)(WAVES * PER * 1024);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const float* p = tb + (size_t)((wave + WAVES * j) * 64 + lane) * 4;
      if (TO_REGS) {
        const v4f x = *reinterpret_cast<const v4f*>(p);
        acc += x;
      } else {
        dma16(p, stage + (unsigned)(wave + WAVES * j) * 1024u);
      }
    }
    if (!TO_REGS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  }
  if (!TO_REGS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[t] = acc[0];
  if (lane == 0) stamps[(size_t)blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int WAVES, int PER, bool SHARED, bool TO_REGS>
void run(const char* name, const float* src, size_t region_floats) {
  const int blocks = 256, tiles = 200;
  float* out; unsigned long long* st;
  CK(hipMalloc(&out, 1 << 16)); CK(hipMalloc(&st, blocks * WAVES * 8));
  auto kern = dma_kernel<WAVES, PER, SHARED, TO_REGS>;
  const size_t lds = (size_t)3 * WAVES * PER * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), lds, 0, src, region_floats, out, st, tiles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * WAVES);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double bytes_cu = (double)tiles * WAVES * PER * 1024;
  printf("%-72s %6.2f B/clk/CU (median wave)  %7.1f GB/s per CU  chip %6.2f TB/s  kernel %.1f us\n", name,
         bytes_cu / (double)h[h.size() / 2], bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * blocks / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main() {
  const size_t region = 92160;                   // floats per region: 360 KB (a 300 x 300 fp32 weight, padded to tiles)
  float* src; CK(hipMalloc(&src, (size_t)256 * region * 4 + (1 << 20)));
  CK(hipMemset(src, 0, (size_t)256 * region * 4 + (1 << 20)));
  run<4, 10, true, false>("LDS-DMA, 4 waves x 10 KiB per tile, every CU the SAME 360 KB", src, region);
  run<4, 10, false, false>("LDS-DMA, 4 waves x 10 KiB per tile, each CU its OWN 360 KB", src, region);
  run<8, 5, true, false>("LDS-DMA, 8 waves x 5 KiB per tile, every CU the SAME 360 KB", src, region);
  run<4, 10, true, true>("global_load_dwordx4 to registers, 4 waves, SAME 360 KB", src, region);
  run<4, 10, false, true>("global_load_dwordx4 to registers, 4 waves, OWN 360 KB", src, region);
  run<1, 10, true, false>("LDS-DMA, 1 wave x 10 KiB per tile, SAME 360 KB", src, region);
  run<2, 10, true, false>("LDS-DMA, 2 waves x 10 KiB per tile, SAME 360 KB", src, region);
  return 0;
}
