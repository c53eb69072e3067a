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

// s = init; for i in [0, n4): s += r4[i].x, .y, .z, .w  (fp32, adds only, in
// that order) -- the reference's accumulation order for `dist += diff*diff`
// (sim_cross_layer.cpp:100-105) once the squares are formed.  `r4` points at
// float4 squares in LDS.  The chain is bound by VALU issue (~4.5 cycles per
// wave-instruction whatever the number of active lanes) as much as by add
// latency, so the loop carries NOTHING but the adds: LDS reads run one
// 8 x 16 B batch ahead with immediate offsets, and no per-element predicate.
__device__ __forceinline__ float chain_sum_lds(const float4* r4, int n4, float init) {
  float s = init;
  const int nb = n4 >> 3;
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
      s += va[u].x; s += va[u].y; s += va[u].z; s += va[u].w;
    }
    if (b + 2 < nb) {
#pragma unroll
      for (int u = 0; u < 8; ++u) va[u] = r4[(b + 2) * 8 + u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s += vb[u].x; s += vb[u].y; s += vb[u].z; s += vb[u].w;
    }
  }
  if (b < nb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s += va[u].x; s += va[u].y; s += va[u].z; s += va[u].w;
    }
  }
  for (int d = nb * 8; d < n4; ++d) {
    const float4 v = r4[d];
    s += v.x; s += v.y; s += v.z; s += v.w;
  }
  return s;
}

// ---------------------------------------------------------------------------
// Speculative two-segment evaluation of the d-ascending chain, bit-exact.
//
// A dependent fp32 add costs ~6.6 cycles of latency and ~4.5 issue cycles of
// its SIMD no matter how few lanes are active (profiles/r01_chainbench_*), so
// a D-long chain per pair is the critical path of the forward pass.  The chain
// cannot be re-associated (it must round like the reference), but its second
// half can be started before the first half has finished if the value the
// first half will end on is GUESSED: the sequential fp32 partial sum differs
// from a tree sum of the same terms by a few ulps only (random-walk rounding
// error, sigma ~ 0.22*sqrt(n) ulp; 2.3 ulp at n = 150).  So each pair gets
// LPR = 64/RW lanes:
//   lane j = 0        walks segment 0 from 0           (exact prefix),
//   lane j = 1..LPR-1 walks segment 1 from pred + (j - LPR/2) ulps,
// where pred is a tree sum of segment 0.  When lane 0 finishes, its end value
// s1 selects the candidate lane whose start value IS s1 -- that lane has
// computed exactly the reference's continuation.  If s1 fell outside the
// window (never observed at +-15 ulps for D = 300: > 6 sigma) the segment is
// simply re-walked from s1.  The result is the reference's sum bit for bit in
// every case; only the time varies.
//
// LDS layout of one pair ("split image", spec_lds_index below): segment 0
// = squares [0, b4) padded with all-zero float4 up to h4 = D4 - b4 entries,
// then segment 1 = squares [b4, D4) (h4 entries).  Adding +0 to the
// non-negative running sum is exact, so both kinds of lane run the same h4
// steps with no predicate in the loop.
__device__ __forceinline__ int spec_b4(int D4) { return D4 >> 1; }
__device__ __forceinline__ int spec_h4(int D4) { return D4 - (D4 >> 1); }
__device__ __forceinline__ int spec_stride4(int D4) { return 2 * spec_h4(D4); }
// position of square-float4 `i` (0 <= i < D4) inside the pair's split image
__device__ __forceinline__ int spec_lds_index(int i, int D4) {
  return i < spec_b4(D4) ? i : i + (spec_h4(D4) - spec_b4(D4));
}

// `img4`   this pair's split image in LDS (spec_stride4(D4) float4; the pad
//          entries must already be zero)
// `pred`   tree sum of segment 0 (any summation order; it only centres the window)
// `j`      this lane's index within the pair's lane group; `lead` = lane id of j = 0
// returns  the full sum, valid in EVERY lane of the group.
template <int LPR>
__device__ __forceinline__ float chain_sum_speculative(const float4* img4, int D4, float pred,
                                                       int j, int lead) {
  const int h4 = spec_h4(D4);
  const bool exact_lane = (j == 0);
  const float start = exact_lane ? 0.0f : __int_as_float(__float_as_int(pred) + (j - LPR / 2));
  __builtin_amdgcn_s_setprio(3);
  const float end = chain_sum_lds(img4 + (exact_lane ? 0 : h4), h4, start);
  const float s1 = __shfl(end, lead, 64);                       // exact prefix sum
  const int k = __float_as_int(s1) - __float_as_int(pred) + LPR / 2;   // candidate lane index
  const bool hit = (k >= 1) && (k <= LPR - 1);
  float total = __shfl(end, lead + (hit ? k : 0), 64);
  if (!hit) total = chain_sum_lds(img4 + h4, h4, s1);           // re-walk, exact
  __builtin_amdgcn_s_setprio(0);
  return total;
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
