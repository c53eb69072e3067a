// csrc/euclid_math.h -- the Euclidean SimCross backward term, bit-exact with
// the reference CPU expression (src/caffe/layers/sim_cross_layer.cpp:216-217)
//
//   Dtype tt = top_diff*top*top*top*(q - a) / (top - 1 + 1e-9);
//
// For Dtype = float the numerator is a float product evaluated left to right,
// `top - 1` is a float, `+ 1e-9` promotes to double, the quotient is a double
// division, and the assignment rounds it to float.
#ifndef MMS_EUCLID_MATH_H_
#define MMS_EUCLID_MATH_H_

#include <hip/hip_runtime.h>

namespace mms {

// Per-(pair,j,k) coefficients.  c = ((g*T)*T)*T ; den = (double)(T-1) + 1e-9 ;
// rcp ~ 1/den (about 1 ulp), used only by the self-checking fast path below.
struct EuclidCoef {
  float c;
  double den;
  double rcp;
};

// 1/den to ~1 ulp without the ~35-instruction IEEE division: hardware
// reciprocal seed + two Newton steps (each squares the relative error; the
// seed is good to > 20 bits, so two steps leave only the final roundings).
// Only euclid_tt's fast path consumes it, and that path re-checks itself.
__device__ __forceinline__ double rcp_newton(double den) {
  double r = __builtin_amdgcn_rcp(den);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ EuclidCoef euclid_coef(float T, float g) {
  EuclidCoef k;
  k.c = g * T * T * T;
  k.den = (double)(T - 1.0f) + 1e-9;
  k.rcp = rcp_newton(k.den);
  return k;
}

// The reference expression, literally.
__device__ __forceinline__ float euclid_tt_exact(float c, double den, float diff) {
  return (float)((double)(c * diff) / den);
}

// Same bits, ~4x fewer instructions: one fp64 multiply by the per-pair
// reciprocal instead of an fp64 division per element.
//   Q = num/den (real).  The reference returns fl32(fl64(Q)).
//   p = fl64(num * rcp), rcp = 1/den to ~1 ulp, is within a few double-ulps
//   of fl64(Q) (rcp error <= ~2 ulp, product rounding 0.5 ulp, each up to 2x
//   when expressed in ulps of p).  fl32(p) can differ from fl32(fl64(Q)) only
//   if a float rounding boundary lies between them.  Below float precision a
//   double carries 29 mantissa bits; the boundary (midpoint of two floats) is
//   the pattern 1000...0 = 2^28 in those bits.  If p's low 29 bits are
//   farther than 64 from 2^28 there is no boundary within reach and the cheap
//   result is the reference's result; otherwise (probability ~2.4e-7) the lane
//   redoes the division exactly.  Results in the float-subnormal range round
//   at a different bit and also take the exact path.
__device__ __forceinline__ float euclid_tt(const EuclidCoef& k, float diff) {
  const float num = k.c * diff;
  const double p = (double)num * k.rcp;
  float res = (float)p;
  const unsigned lo = (unsigned)__double_as_longlong(p) & 0x1fffffffu;
  const bool near_boundary = (lo - 0x0fffffc0u) <= 128u;
  const bool tiny = (fabsf(res) <= 1.17549435e-38f) && (num != 0.0f);
  if (near_boundary || tiny) res = (float)((double)num / k.den);
  return res;
}

// sum_d sq[d], d ascending, fp32, adds only -- the reference's accumulation
// order for `dist += diff*diff` (sim_cross_layer.cpp:100-105) once the
// squares are formed.  `r4` points at D4 float4 of squares in LDS.  The LDS
// reads run one 8 x 16 B batch ahead of the adds, and the wave is prioritised
// while it is latency-bound on this dependent chain.
__device__ __forceinline__ float chain_sum_lds(const float4* r4, int D4) {
  __builtin_amdgcn_s_setprio(3);
  float dist = 0.f;
  const int nb = D4 >> 3;
  float4 va[8], vb[8];
  if (nb > 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) va[u] = r4[u];
  }
  int b = 0;
  for (; b + 2 <= nb; b += 2) {
#pragma unroll
    for (int u = 0; u < 8; ++u) vb[u] = r4[(b + 1) * 8 + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      dist += va[u].x; dist += va[u].y; dist += va[u].z; dist += va[u].w;
    }
    if (b + 2 < nb) {
#pragma unroll
      for (int u = 0; u < 8; ++u) va[u] = r4[(b + 2) * 8 + u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      dist += vb[u].x; dist += vb[u].y; dist += vb[u].z; dist += vb[u].w;
    }
  }
  if (b < nb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      dist += va[u].x; dist += va[u].y; dist += va[u].z; dist += va[u].w;
    }
  }
  for (int d = nb * 8; d < D4; ++d) {
    const float4 v = r4[d];
    dist += v.x; dist += v.y; dist += v.z; dist += v.w;
  }
  __builtin_amdgcn_s_setprio(0);
  return dist;
}

// Compiler-level ordering between LDS writes of some lanes and LDS reads of
// other lanes of the SAME wave (the hardware executes a wave's LDS
// instructions in order; no workgroup barrier is needed or wanted).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace mms
#endif  // MMS_EUCLID_MATH_H_
