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

#include "mms_common.h"

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

// Four elements that share one coefficient set (a float4 of one pair's row):
// same arithmetic as euclid_tt, with the rare-case tests made cheaper and
// RETURNED instead of branched on, so that a caller can run several float4s
// as one straight-line block (independent instruction chains interleave) and
// take a single wave-level branch for all of them:
//   * "result could be float-subnormal" needs |num * rcp| < 2^-126; when
//     |rcp| >= 1 (always true for T in (0,1]: |den| <= 1) that requires num
//     itself to be subnormal -- one v_cmp_class per element; a pair whose
//     |rcp| < 1 (only possible for caller-supplied T outside (0,1]) takes the
//     exact path wholesale.
__device__ __forceinline__ float4 euclid_tt4_fast(const EuclidCoef& k, const float4& d, bool& risky) {
  const float n0 = k.c * d.x, n1 = k.c * d.y, n2 = k.c * d.z, n3 = k.c * d.w;
  const double p0 = (double)n0 * k.rcp, p1 = (double)n1 * k.rcp;
  const double p2 = (double)n2 * k.rcp, p3 = (double)n3 * k.rcp;
  float4 r;
  r.x = (float)p0; r.y = (float)p1; r.z = (float)p2; r.w = (float)p3;
  auto near = [](double p) {
    const unsigned lo = (unsigned)__double_as_longlong(p) & 0x1fffffffu;
    return (lo - 0x0fffffc0u) <= 128u;
  };
  auto subn = [](float x) -> bool { return __builtin_isfpclass(x, 0x0090 /* +-subnormal */); };
  risky = near(p0) || near(p1) || near(p2) || near(p3) || subn(n0) || subn(n1) || subn(n2) ||
          subn(n3) || !(fabs(k.rcp) >= 1.0);
  return r;
}
__device__ __forceinline__ float4 euclid_tt4_exact(const EuclidCoef& k, const float4& d) {
  float4 r;
  r.x = euclid_tt_exact(k.c, k.den, d.x); r.y = euclid_tt_exact(k.c, k.den, d.y);
  r.z = euclid_tt_exact(k.c, k.den, d.z); r.w = euclid_tt_exact(k.c, k.den, d.w);
  return r;
}
__device__ __forceinline__ float4 euclid_tt4(const EuclidCoef& k, const float4& d) {
  bool risky;
  float4 r = euclid_tt4_fast(k, d, risky);
  if (risky) r = euclid_tt4_exact(k, d);
  return r;
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

// Packed variant: one v_pk_add_f32 advances TWO running sums (the two halves of
// `s`) by the same addend.  op_sel picks which dword of the 64-bit source pair
// feeds both halves, so no broadcast moves are needed (plain vector code: hipcc
// emits the op_sel forms itself and keeps its counted LDS waits).  Measured
// (profiles/r01_chainbench…): ~10.5 ticks per dependent packed add however many
// waves share the SIMD, against ~14 for a plain add at two waves per SIMD.
typedef float float2v __attribute__((ext_vector_type(2)));
#ifdef MMS_STAMPS   // dev-only: speculation-miss counter (tools/stampbench.hip)
__device__ unsigned mms_miss_count = 0;
#define MMS_COUNT_MISS() do { if ((threadIdx.x & 31) == 0) atomicAdd(&mms_miss_count, 1u); } while (0)
#else
#define MMS_COUNT_MISS() do {} while (0)
#endif
__device__ __forceinline__ float2v chain_sum_lds_pk(const float4* r4, int n4, float2v init) {
  float2v s = init;
  auto step = [&](const float4& v) {
    // hipcc lowers each line to one v_pk_add_f32 with op_sel broadcasting the addend
    s = s + (float2v){v.x, v.x};
    s = s + (float2v){v.y, v.y};
    s = s + (float2v){v.z, v.z};
    s = s + (float2v){v.w, v.w};
  };
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
    for (int u = 0; u < 8; ++u) step(va[u]);
    if (b + 2 < nb) {
#pragma unroll
      for (int u = 0; u < 8; ++u) va[u] = r4[(b + 2) * 8 + u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) step(vb[u]);
  }
  if (b < nb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) step(va[u]);
  }
  for (int d = nb * 8; d < n4; ++d) step(r4[d]);
  return s;
}

// ---------------------------------------------------------------------------
// Speculative three-segment evaluation of the d-ascending chain, bit-exact.
//
// A dependent fp32 add costs ~6.6 cycles of latency and ~4.5 issue cycles of
// its SIMD no matter how few lanes are active (profiles/r01_chainbench_*), so
// a D-long chain per pair is the critical path of the forward pass.  The chain
// cannot be re-associated (it must round like the reference), but a later
// segment can be started before the earlier ones have finished if the value
// they will end on is GUESSED: the sequential fp32 partial sum differs from a
// tree sum of the same terms by a few ulps only (random-walk rounding error,
// sigma ~ 0.22*sqrt(n) ulp: 1.8 ulp at n = 100, 2.5 at n = 200, measured).
// Each pair gets LPR = 64/RW lanes and, thanks to packed adds, 2*LPR running
// sums ("slots"):
//   lane 0                 walks segment 0 from 0                       (exact prefix)
//   lanes 1 .. L1          walk segment 1 from pred1 + k ulps, |k| <= H1
//   lanes L1+1 .. LPR-1    walk segment 2 from pred2 + k ulps, |k| <= H2
// (LPR = 32: H1 = 12, H2 = 15; LPR = 64: H1 = 25, H2 = 36), where pred1/pred2
// are tree sums of segment 0 / segments 0-1.  When the walks finish, the exact
// end of segment 0 selects the segment-1 slot whose start value IS that end --
// that slot has computed exactly the reference's continuation -- and its end
// selects the segment-2 slot the same way.  A miss (the true value outside the
// window; < 1e-6 per pair on embedding-like data, forced in the tests) re-walks
// that segment from the exact value.  The result is the reference's sum bit for
// bit in every case; only the time varies.
//
// LDS image of one pair: its D4 float4 squares followed by all-zero float4 up
// to 3*h4, h4 = ceil(D4/3); segment g is [g*h4, (g+1)*h4).  Adding +0 to the
// non-negative running sum is exact, so every lane runs the same h4 steps.
__device__ __forceinline__ int spec_h4(int D4) { return (D4 + 2) / 3; }
__device__ __forceinline__ int spec_stride4(int D4) { return 3 * spec_h4(D4); }

template <int LPR> struct SpecPlan;
template <> struct SpecPlan<32> { static constexpr int H1 = 12, L1 = 13, H2 = 15; };
template <> struct SpecPlan<64> { static constexpr int H1 = 25, L1 = 26, H2 = 36; };

// `img4`   this pair's image in LDS (pad entries already zero)
// `pred1`, `pred2`  tree sums of segment 0 and of segments 0-1 (they only centre the windows)
// `j`      this lane's index within the pair's lane group; `lead` = lane id of j = 0
// returns  the full sum, valid in EVERY lane of the group.
template <int LPR>
__device__ __forceinline__ float chain_sum_speculative(const float4* img4, int D4, float pred1,
                                                       float pred2, int j, int lead) {
  typedef SpecPlan<LPR> P;
  const int h4 = spec_h4(D4);
  const int seg = (j == 0) ? 0 : (j <= P::L1 ? 1 : 2);
  const int c0 = (seg == 1) ? 2 * (j - 1) - P::H1 : 2 * (j - 1 - P::L1) - P::H2;   // ulp offset of slot 0
  const int pbits = __float_as_int(seg == 1 ? pred1 : pred2);
  float2v start;
  start.x = (seg == 0) ? 0.0f : __int_as_float(pbits + c0);
  start.y = (seg == 0) ? 0.0f : __int_as_float(pbits + c0 + 1);
  __builtin_amdgcn_s_setprio(3);
  const float2v end = chain_sum_lds_pk(img4 + seg * h4, h4, start);
  // segment 0 -> 1
  const float s1 = __shfl(end.x, lead, 64);
  const int k1 = __float_as_int(s1) - __float_as_int(pred1) + P::H1;     // slot index in segment 1
  const bool hit1 = (k1 >= 0) && (k1 <= 2 * P::H1);
  const int l1 = lead + 1 + ((hit1 ? k1 : 0) >> 1);
  const float e1x = __shfl(end.x, l1, 64), e1y = __shfl(end.y, l1, 64);
  float s2 = (k1 & 1) ? e1y : e1x;
  if (!hit1) { MMS_COUNT_MISS(); s2 = chain_sum_lds(img4 + h4, h4, s1); }   // re-walk, exact
  // segment 1 -> 2
  const int k2 = __float_as_int(s2) - __float_as_int(pred2) + P::H2;
  const bool hit2 = (k2 >= 0) && (k2 <= 2 * P::H2);
  const int l2 = lead + 1 + P::L1 + ((hit2 ? k2 : 0) >> 1);
  const float e2x = __shfl(end.x, l2, 64), e2y = __shfl(end.y, l2, 64);
  float s3 = (k2 & 1) ? e2y : e2x;
  if (!hit2) { MMS_COUNT_MISS(); s3 = chain_sum_lds(img4 + 2 * h4, h4, s2); }
  __builtin_amdgcn_s_setprio(0);
  return s3;
}

// ---- quad-shared variant for WIDE rows (one pair per wave; cfg 5: D = 1024) --------------------------------
// With a whole wave on one pair, every lane of chain_sum_speculative reads its entire segment from LDS
// (identical addresses within a segment: 86 ds_read_b128 per lane per pair) and the LDS -> VGPR return path
// -- 8 LDS cycles per wave-wide 16-byte read, whatever is broadcast -- paces the kernel: 2,752 LDS cycles per
// four pairs per CU against 1,376 cycles of packed adds (tools/f16abl.sh).  Here the four lanes
// of a QUAD share one copy of the addends: lane ql of a quad reads only the float4s u = ql (mod 4) of its
// segment (a quarter of the LDS traffic) and every add takes its addend from the owning lane through the
// DPP quad_perm broadcast of the add instruction itself (v_add_f32_dpp: no extra instruction, no LDS).
// Packed adds have no DPP form, so a lane's two running sums cost two adds per step instead of one packed
// add: 30 % more VALU issue, a quarter of the LDS return traffic.
//   quad 0           walks segment 0 from 0 (four identical copies)
//   quads 1 .. 6     24 lanes, 48 start values pred1 + k ulps, -24 <= k <= 23
//   quads 7 .. 15    36 lanes, 72 start values pred2 + k ulps, -36 <= k <= 35
// Segments are quad_h4(D4) float4 long (a multiple of 4, so that every lane of a quad runs the same number
// of reads); the image is padded with zeros to three segments as before.  Same stitching rule, same exact
// re-walk on a miss: the result is the reference's sum bit for bit.
__device__ __host__ __forceinline__ int quad_h4(int D4) { return 4 * ((D4 + 11) / 12); }
struct QuadPlan { static constexpr int H1 = 24, Q1 = 1, N1 = 6, H2 = 36; };   // quads of segment 1: [Q1, Q1+N1)

// s += (value of `a` in quad lane O), one instruction.  Inline asm: left to the compiler, the two running sums
// of a lane are packed into v_pk_add_f32 (which has no DPP form) behind a v_mov_b32_dpp per step.  The DPP
// source operand is never written by a VALU instruction (the addends come straight from ds_read), so the
// VALU-write -> DPP-read hazard cannot arise between these statements.
#define MMS_QUAD_STEP4(P)                                                                      \
  asm("v_add_f32_dpp %0, %2, %0 quad_perm:[" P "] row_mask:0xf bank_mask:0xf\n\t"             \
      "v_add_f32_dpp %1, %2, %1 quad_perm:[" P "] row_mask:0xf bank_mask:0xf\n\t"             \
      "v_add_f32_dpp %0, %3, %0 quad_perm:[" P "] row_mask:0xf bank_mask:0xf\n\t"             \
      "v_add_f32_dpp %1, %3, %1 quad_perm:[" P "] row_mask:0xf bank_mask:0xf\n\t"             \
      "v_add_f32_dpp %0, %4, %0 quad_perm:[" P "] row_mask:0xf bank_mask:0xf\n\t"             \
      "v_add_f32_dpp %1, %4, %1 quad_perm:[" P "] row_mask:0xf bank_mask:0xf\n\t"             \
      "v_add_f32_dpp %0, %5, %0 quad_perm:[" P "] row_mask:0xf bank_mask:0xf\n\t"             \
      "v_add_f32_dpp %1, %5, %1 quad_perm:[" P "] row_mask:0xf bank_mask:0xf"                  \
      : "+v"(sx), "+v"(sy) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w))
// Both running sums of a lane advance by the four values of float4 `v` held by quad lane O: eight
// v_add_f32_dpp in ONE asm statement (left to the compiler, the two sums are packed into v_pk_add_f32 --
// which has no DPP form -- behind a v_mov_b32_dpp per step; one statement per add drew a padding s_nop each).
// The DPP source operands are never VALU results (the addends come straight from ds_read) and the
// accumulators are ordinary src1 operands, so no VALU-write -> DPP-read hazard exists inside the string.
template <int O>
__device__ __forceinline__ void quad_step4(float& sx, float& sy, const float4& v) {
  static_assert(O >= 0 && O < 4, "quad lane");
  if (O == 0) MMS_QUAD_STEP4("0,0,0,0");
  if (O == 1) MMS_QUAD_STEP4("1,1,1,1");
  if (O == 2) MMS_QUAD_STEP4("2,2,2,2");
  if (O == 3) MMS_QUAD_STEP4("3,3,3,3");
}
#undef MMS_QUAD_STEP4
// r4: this lane's segment in LDS; the lane reads float4s ql, ql + 4, ...; h4 % 4 == 0
__device__ __forceinline__ float2v chain_sum_lds_quad(const float4* r4, int h4, int ql, float2v init) {
  float sx = init.x, sy = init.y;
  const float4* mine = r4 + ql;
  const int nm = h4 >> 2;                          // reads per lane
  float4 va[8], vb[8];
  auto steps = [&](const float4 (&v)[8], int cnt) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < cnt) {
        quad_step4<0>(sx, sy, v[u]);
        quad_step4<1>(sx, sy, v[u]);
        quad_step4<2>(sx, sy, v[u]);
        quad_step4<3>(sx, sy, v[u]);
      }
    }
  };
  const int nb = nm >> 3;
  if (nb > 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) va[u] = mine[4 * u];
  }
  int b = 0;
  for (; b + 2 <= nb; b += 2) {
#pragma unroll
    for (int u = 0; u < 8; ++u) vb[u] = mine[4 * ((b + 1) * 8 + u)];
    steps(va, 8);
    if (b + 2 < nb) {
#pragma unroll
      for (int u = 0; u < 8; ++u) va[u] = mine[4 * ((b + 2) * 8 + u)];
    }
    steps(vb, 8);
  }
  if (b < nb) steps(va, 8);
  // remainder (nm % 8 reads): EVERY lane of the quad executes the same steps (the broadcasts need all four)
  const int rem = nm - nb * 8;
  if (rem > 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) va[u] = mine[4 * (nb * 8 + (u < rem ? u : 0))];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < rem) {                               // wave-uniform
        quad_step4<0>(sx, sy, va[u]);
        quad_step4<1>(sx, sy, va[u]);
        quad_step4<2>(sx, sy, va[u]);
        quad_step4<3>(sx, sy, va[u]);
      }
    }
  }
  return (float2v){sx, sy};
}

// `img4` the pair's image (3 * h4 float4, zero padded), `pred1` / `pred2` tree sums of segment 0 and segments
// 0-1, `lane` 0..63 (one pair per wave).  Returns the full sum in every lane.
__device__ __forceinline__ float chain_sum_speculative_quad(const float4* img4, int h4, float pred1, float pred2,
                                                            int lane) {
  typedef QuadPlan P;
  const int qd = lane >> 2, ql = lane & 3;
  const int seg = (qd < P::Q1) ? 0 : (qd < P::Q1 + P::N1 ? 1 : 2);
  const int first1 = 4 * P::Q1, first2 = 4 * (P::Q1 + P::N1);          // first lane of segments 1 and 2
  const int c0 = (seg == 1) ? 2 * (lane - first1) - P::H1 : 2 * (lane - first2) - P::H2;   // ulp offset of slot 0
  const int pbits = __float_as_int(seg == 1 ? pred1 : pred2);
  float2v start;
  start.x = (seg == 0) ? 0.0f : __int_as_float(pbits + c0);
  start.y = (seg == 0) ? 0.0f : __int_as_float(pbits + c0 + 1);
  __builtin_amdgcn_s_setprio(3);
  const float2v end = chain_sum_lds_quad(img4 + seg * h4, h4, ql, start);
  // segment 0 -> 1
  const float s1 = __shfl(end.x, 0, 64);
  const int k1 = __float_as_int(s1) - __float_as_int(pred1) + P::H1;     // slot index in segment 1
  const bool hit1 = (k1 >= 0) && (k1 < 8 * P::N1);
  const int l1 = first1 + ((hit1 ? k1 : 0) >> 1);
  const float e1x = __shfl(end.x, l1, 64), e1y = __shfl(end.y, l1, 64);
  float s2 = (k1 & 1) ? e1y : e1x;
  if (!hit1) { MMS_COUNT_MISS(); s2 = chain_sum_lds(img4 + h4, h4, s1); }   // re-walk, exact
  // segment 1 -> 2
  const int k2 = __float_as_int(s2) - __float_as_int(pred2) + P::H2;
  const bool hit2 = (k2 >= 0) && (k2 < 2 * (64 - first2));
  const int l2 = first2 + ((hit2 ? k2 : 0) >> 1);
  const float e2x = __shfl(end.x, l2, 64), e2y = __shfl(end.y, l2, 64);
  float s3 = (k2 & 1) ? e2y : e2x;
  if (!hit2) { MMS_COUNT_MISS(); s3 = chain_sum_lds(img4 + 2 * h4, h4, s2); }
  __builtin_amdgcn_s_setprio(0);
  return s3;
}

// Compile-time-length variants for the widths the wave-pair kernel is specialised
// for (simcross_elementwise.hip: euclid_pair32_kernel).  H4 (float4s per segment)
// is a constant, so
//  * the segment's LDS reads are all issued up front into registers (two waves
//    per SIMD leave 256 VGPRs per lane) -- and BEFORE the window centres are
//    reduced, so that LDS latency hides behind the DPP sums;
//  * the chain is H4*4 straight-line packed adds: no loop control, no register
//    rotation, and op_sel broadcasts each addend lane-locally;
//  * the two look-ups that stitch the segments together run on the SCALAR unit
//    (v_readlane with a computed lane, s_sub/s_cmp) instead of three dependent
//    ds_bpermute round trips;
//  * a miss anywhere (never observed on real data; forced by the adversarial
//    tests) sends the whole wave to ONE exact re-walk of the full image.
template <int H4>
struct SpecSegment {
  float4 v[H4];
  __device__ __forceinline__ void load(const float4* r4) {
#pragma unroll
    for (int u = 0; u < H4; ++u) v[u] = r4[u];
  }
  __device__ __forceinline__ float2v chain(float2v s) const {
#pragma unroll
    for (int u = 0; u < H4; ++u) {
      const float2v lo = {v[u].x, v[u].y}, hi = {v[u].z, v[u].w};
      s = s + __builtin_shufflevector(lo, lo, 0, 0);
      s = s + __builtin_shufflevector(lo, lo, 1, 1);
      s = s + __builtin_shufflevector(hi, hi, 0, 0);
      s = s + __builtin_shufflevector(hi, hi, 1, 1);
    }
    return s;
  }
};

// segment of lane j of a 32-lane group: j = 0 walks segment 0 from 0.0f exactly
__device__ __forceinline__ int spec_seg32(int j) { return (j == 0) ? 0 : (j <= SpecPlan<32>::L1 ? 1 : 2); }

__device__ __forceinline__ float2v spec_start32(float pred1, float pred2, int j) {
  typedef SpecPlan<32> P;
  const int seg = spec_seg32(j);
  const int c0 = (seg == 1) ? 2 * (j - 1) - P::H1 : 2 * (j - 1 - P::L1) - P::H2;   // ulp offset of slot 0
  const int pbits = __float_as_int(seg == 1 ? pred1 : pred2);
  float2v start;
  start.x = (seg == 0) ? 0.0f : __int_as_float(pbits + c0);
  start.y = (seg == 0) ? 0.0f : __int_as_float(pbits + c0 + 1);
  return start;
}

// Stitch the three segments of BOTH pairs of a wave at once (lanes 0-31 / 32-63), on the
// VALU: the lane whose start value equals the running total contributes its end value, and
// an OR-reduction over the half (DPP, no LDS) hands it to every lane.  Sums of squares are
// never negative, so bit 31 is free to carry "some lane matched".  Three reductions of six
// dependent DPP steps replace three ds_bpermute round trips (or ~40 dependent scalar ops).
// Returns the pair's total in every lane of its half; `hit` is false on a window miss.
__device__ __forceinline__ float spec_resolve_halves(float2v start, float2v end, int j, bool* hit) {
  constexpr unsigned F = 0x80000000u;
  const int seg = spec_seg32(j);
  const unsigned sx = __float_as_uint(start.x), sy = __float_as_uint(start.y);
  const unsigned ex = __float_as_uint(end.x) | F, ey = __float_as_uint(end.y) | F;
  const unsigned s1 = half_wave_or(j == 0 ? ex : 0u) & ~F;                 // segment 0 ends here
  const unsigned m1 = (seg == 1) ? (sx == s1 ? ex : (sy == s1 ? ey : 0u)) : 0u;
  const unsigned s2f = half_wave_or(m1);
  const unsigned s2 = s2f & ~F;
  const unsigned m2 = (seg == 2) ? (sx == s2 ? ex : (sy == s2 ? ey : 0u)) : 0u;
  const unsigned s3f = half_wave_or(m2);
  *hit = ((s2f & s3f) & F) != 0;
  return __uint_as_float(s3f & ~F);
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
