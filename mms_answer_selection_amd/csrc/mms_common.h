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

// Write-once streaming stores (bottom diffs: written here, read by another kernel much
// later): the non-temporal form does not allocate in L2, so the NEXT launch's reads are not
// queued behind this launch's write-backs.  Measured on cfg 2, HBM-cold: 6.1 -> 5.35 us/step.
typedef float mms_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(float4* p, const float4& v) {
  __builtin_nontemporal_store((mms_v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<mms_v4f*>(p));
}
template <typename V>
__device__ __forceinline__ void stream_store_vec(V* p, const V& v) {   // V: an ext_vector_type
  __builtin_nontemporal_store(v, p);
}

// Sum across the 64 lanes of a wave in a fixed order (deterministic); every
// lane ends with the total.  DPP cross-lane adds, no LDS round trips: xor-1 and
// xor-2 inside quads, half-mirror and mirror inside each row of 16, then
// row_bcast:15 / row_bcast:31 accumulate the four row totals into row 3, whose
// lane 63 is broadcast.  (__shfl_xor lowers to ds_bpermute: ~100 cycles of LDS
// latency per step, six dependent steps.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);   // row_mirror      -> every lane holds its row's total
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> row 3 holds the wave total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sum across each 32-lane HALF of a wave (lanes 0-31 and 32-63 separately); every
// lane ends with its half's total.  Four DPP steps give each row of 16 its total;
// gfx950's v_permlane16_swap then exchanges odd and even rows so that one more add
// completes both halves at once.
__device__ __forceinline__ float half_wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);
  v = dpp_add<0x4E, 0xf>(v);
  v = dpp_add<0x141, 0xf>(v);
  v = dpp_add<0x140, 0xf>(v);
  const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// Per-lane (a, b) -> lanes 0-31 get a[l] + a[l+32], lanes 32-63 get b[l-32] + b[l]: the first
// step of reducing two quantities of a whole wave into one half each (v_permlane32_swap).
__device__ __forceinline__ float swap_halves_add(float a, float b) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// Bitwise OR across each 32-lane half of a wave, same data movement as half_wave_sum.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_or(unsigned v) {
  return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned half_wave_or(unsigned v) {
  v = dpp_or<0xB1>(v);
  v = dpp_or<0x4E>(v);
  v = dpp_or<0x141>(v);
  v = dpp_or<0x140>(v);
  const auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return sw[0] | sw[1];
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
