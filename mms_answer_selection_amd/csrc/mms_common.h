// csrc/mms_common.h -- shared device helpers and launch plumbing (gfx950 only).
#ifndef MMS_COMMON_H_
#define MMS_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mms.h"

namespace mms {

constexpr int kWave = 64;  // CDNA4 wavefront

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? MMS_OK : MMS_ERR_LAUNCH;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// Sum across the 64 lanes of a wave in a fixed butterfly order (deterministic).
// Every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Sum across a workgroup of THREADS (multiple of 64) threads; result valid in
// every thread.  `red` is LDS scratch of THREADS/64 floats.  Fixed order.
template <int THREADS>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from a previous use
  if ((threadIdx.x & 63) == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) t += red[w];
  return t;
}

}  // namespace mms
#endif  // MMS_COMMON_H_
