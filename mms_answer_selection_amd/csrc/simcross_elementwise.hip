// csrc/simcross_elementwise.hip -- SimCross dist_mode 0 (cosine) and 1
// (Euclidean) forward/backward for gfx950.  HBM-bound: no MFMA here.
//
// Reference semantics (all file:line in src/caffe/layers/sim_cross_layer.cpp):
//   Euclid fwd  :96-111   T = 1/(1+sqrt(sum_d (q-a)^2)), d ascending, fp32.
//   Euclid bwd  :208-225  tt = dT*T*T*T*(q-a)/(T-1+1e-9) (double divide);
//                         dq[j,d] = sum_k tt (k ascending from 0),
//                         da[k,d] = sum_j -tt (j ascending from 0).
//   Cosine fwd  :112-139  n0,n1 = sqrt(dot) cached; T = dot/n0/n1.
//   Cosine bwd  :226-250.
//
// Two geometries get their own kernels:
//   "rows"  W1 == W2 == 1 (sentence-vector pairs; BASELINE cfg 2/5): a
//           workgroup owns ROWS consecutive pairs, streams them with 16-byte
//           loads into LDS, and ONE lane per pair walks d ascending so the sum
//           has the reference's order bit for bit.  The forward+backward
//           fusion keeps q-a in LDS so q and a are read from HBM once.
//   "cross" general W1 x W2 word grids (TREC-QA 40x40): one wave per
//           (pair, j-tile, k-tile), q/a d-chunks staged in LDS, an RJ x RK
//           register tile per lane, d ascending per output.
//
// Compiled with -ffp-contract=off: the reference CPU build has no FMA
// contraction, so mul and add must round separately to match it bitwise.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "euclid_math.h"
#include "mms_common.h"

namespace mms {

// Dev-only phase stamps (tools/stampbench.hip builds this file with -DMMS_STAMPS): lane 0 of
// every wave of the fused rows kernel records s_memtime at its phase boundaries into a
// buffer no other code reads.  Compiled out of the product.
#ifdef MMS_STAMPS
__device__ unsigned long long* mms_stamp_buf = nullptr;
#define MMS_STAMP(k)                                                                        \
  do {                                                                                      \
    if (mms_stamp_buf && (threadIdx.x & 63) == 0)                                           \
      mms_stamp_buf[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define MMS_STAMP(k) do {} while (0)
#endif

// =============================== rows geometry ==============================

// ---- wave-centric kernel (the fast path) ------------------------------------
// A wave owns RW (1 or 2) consecutive pairs (RW*D/4 float4 per operand); the
// four waves of a workgroup are independent (no workgroup barrier).  Timeline
// of one wave:
//   1. ALL its 16-byte loads of q and a are issued back to back (NIT per
//      operand per lane, predicated) -- nothing waits inside a loop;
//   2. diff = q-a stays in registers; diff^2 goes to the wave's LDS slice;
//   3. the d-ascending sum of each pair's squares -- the reference's summation
//      order -- is evaluated by the pair's 64/RW lanes with the speculative
//      two-segment scheme of euclid_math.h (bit-exact, half the chain length);
//   4. every lane derives T and the backward coefficients of its pair
//      (wave-uniform per lane group; no LDS round trip);
//   5. every lane turns its register-resident diffs into dq / da and stores
//      16 bytes per lane.
// FWD only stops after 3; BWD only skips 2-3 and reads T from memory.
template <int NIT, int RW, bool FWD, bool BWD>
__global__ __launch_bounds__(256) void euclid_rows_wave_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top_in, const float* __restrict__ top_diff,
    float* __restrict__ top_out, float* __restrict__ dq, float* __restrict__ da, int N,
    int D4) {
  constexpr int LPR = 64 / RW;                   // lanes per pair
  extern __shared__ float4 lds4[];               // [4 waves][RW split images] (euclid_math.h)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + wave) * RW;
  if (row0 >= N) return;                         // whole wave leaves; no block barrier below
  const int rows = min(RW, N - row0);
  const int n4 = rows * D4;
  const size_t base4 = (size_t)row0 * D4;
  const float4* q4 = reinterpret_cast<const float4*>(q) + base4;
  const float4* a4 = reinterpret_cast<const float4*>(a) + base4;
  const int st4 = spec_stride4(D4);
  float4* sq4 = lds4 + (size_t)wave * RW * st4;

  MMS_STAMP(0);
  float4 x[NIT], y[NIT], df[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    const int ii = i < n4 ? i : 0;               // clamp: keep the load unconditional
    x[it] = q4[ii];
    y[it] = a4[ii];
  }
  MMS_STAMP(1);
  // this lane's pair for the chain / coefficient work
  const int grp = lane / LPR, j = lane % LPR;
  const int grow = min(grp, rows - 1);           // a missing 2nd pair mirrors the 1st (results unused)
  float T = 0.f;
  if (!FWD) T = top_in[row0 + grow];
  float g = 0.f;
  if (BWD) g = top_diff[row0 + grow];

  float2v pred[RW];                              // per pair: (pred1, pred2) partial sums
#pragma unroll
  for (int r = 0; r < RW; ++r) pred[r] = (float2v){0.f, 0.f};
  const int h4 = spec_h4(D4);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    df[it].x = x[it].x - y[it].x; df[it].y = x[it].y - y[it].y;
    df[it].z = x[it].z - y[it].z; df[it].w = x[it].w - y[it].w;
    if (FWD) {
      const int i = lane + 64 * it;
      float4 s;
      s.x = df[it].x * df[it].x; s.y = df[it].y * df[it].y;
      s.z = df[it].z * df[it].z; s.w = df[it].w * df[it].w;
      // image slot of this float4 (pair r's image starts at r*st4) and its tree-sum
      // contribution to the two predictions of the pair it belongs to
      const bool r1 = (RW == 2) && (i >= D4);
      const int ir = r1 ? i - D4 : i;
      if (i < n4) sq4[r1 ? i + (st4 - D4) : i] = s;
      const float s4 = (i < n4) ? (s.x + s.y) + (s.z + s.w) : 0.f;
      float2v c;                                   // (segment-0 part, segments-0-1 part)
      c.x = (ir < h4) ? s4 : 0.f;
      c.y = (ir < 2 * h4) ? s4 : 0.f;
      const float2v z2 = {0.f, 0.f};
      pred[0] += r1 ? z2 : c;
      if (RW == 2) pred[RW - 1] += r1 ? c : z2;
    }
  }
#ifdef MMS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MMS_STAMP(2);
  if (FWD) {
    // zero pad at the end of each image (0..2 entries)
    const int npad = st4 - D4;
    if (lane < RW * npad) sq4[(lane / npad) * st4 + D4 + (lane % npad)] = make_float4(0.f, 0.f, 0.f, 0.f);
    float my1 = 0.f, my2 = 0.f;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float p1 = wave_sum(pred[r].x), p2 = wave_sum(pred[r].y);
      if (r == grow) { my1 = p1; my2 = p2; }
    }
    wave_lds_sync();
    MMS_STAMP(3);
#if defined(MMS_ABLATE) && MMS_ABLATE >= 1   // dev-only timing ablation (tools/ablate.sh): no chain
    const float dist = my2;
#else
    const float dist = chain_sum_speculative<LPR>(sq4 + grow * st4, D4, my1, my2, j, grp * LPR);
#endif
    MMS_STAMP(4);
    T = 1.0f / (1.0f + sqrtf(dist));            // :106-107
    if (BWD) asm volatile("" : "+v"(g));        // in a register before the store of T (see euclid_pair32_kernel)
    if (j == 0 && grp < rows) top_out[row0 + grp] = T;
  }
  if (!BWD) return;

  // coefficients of this lane group's pair, then of the pairs this lane's elements belong to
  const EuclidCoef mine = euclid_coef(T, g);
  EuclidCoef kr[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    // lane r*LPR is a compile-time lane: v_readlane (no LDS round trip as with ds_bpermute)
    kr[r].c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.c), r * LPR));
    {
      const long long dn = __double_as_longlong(mine.den), rc = __double_as_longlong(mine.rcp);
      const unsigned dlo = __builtin_amdgcn_readlane((int)(unsigned)dn, r * LPR);
      const unsigned dhi = __builtin_amdgcn_readlane((int)(unsigned)(dn >> 32), r * LPR);
      const unsigned rlo = __builtin_amdgcn_readlane((int)(unsigned)rc, r * LPR);
      const unsigned rhi = __builtin_amdgcn_readlane((int)(unsigned)(rc >> 32), r * LPR);
      kr[r].den = __longlong_as_double((long long)(((unsigned long long)dhi << 32) | dlo));
      kr[r].rcp = __longlong_as_double((long long)(((unsigned long long)rhi << 32) | rlo));
    }
  }
  MMS_STAMP(5);
  float4* dq4 = reinterpret_cast<float4*>(dq) + base4;
  float4* da4 = reinterpret_cast<float4*>(da) + base4;
  // all NIT float4s as ONE straight-line block (their instruction chains interleave),
  // then a single wave-level branch for the rare exact re-computation
  float4 t[NIT];
  bool any_risky = false;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    const EuclidCoef& k = (RW == 2 && i >= D4) ? kr[RW - 1] : kr[0];
    bool risky;
    t[it] = euclid_tt4_fast(k, df[it], risky);
    any_risky |= risky && (i < n4);
  }
  if (any_risky) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = lane + 64 * it;
      const EuclidCoef& k = (RW == 2 && i >= D4) ? kr[RW - 1] : kr[0];
      t[it] = euclid_tt4_exact(k, df[it]);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    if (i >= n4) break;
    // dq = 0 + tt ; da = 0 + (-tt)   (:176-177 zero, :219-220 accumulate once)
    float4 o0, o1;
    o0.x = 0.f + t[it].x; o0.y = 0.f + t[it].y; o0.z = 0.f + t[it].z; o0.w = 0.f + t[it].w;
    o1.x = 0.f + (-t[it].x); o1.y = 0.f + (-t[it].y); o1.z = 0.f + (-t[it].z); o1.w = 0.f + (-t[it].w);
    stream_store(dq4 + i, o0);
    stream_store(da4 + i, o1);
  }
  MMS_STAMP(6);
#ifdef MMS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MMS_STAMP(7);
}

// ---- width-specialised wave-pair kernel (the headline configuration) ----------
// Same algorithm as euclid_rows_wave_kernel<.., RW = 2, ..> for D/4 = D4C known at
// compile time, with the layout made ROW-ALIGNED: lanes 0-31 own pair 2w, lanes
// 32-63 pair 2w+1, lane j holds float4s j, j+32, j+64 of its pair.  A lane then
// belongs to ONE pair for everything it does (loads, squares, speculation lane,
// coefficients, stores): no per-slot pair masks, no cross-lane broadcast of the
// coefficients, half-wave DPP sums, and every loop bound is a constant, so the
// chain is straight-line code.  About 40 % fewer wave-instructions than the
// generic kernel, which is what bounds this kernel.
//
// EXACT selects the arithmetic of the backward term tt = dT*T^3*(q-a)/(T-1+1e-9):
//   true : the reference's bits (fp32 product, DOUBLE divisor, one rounding to
//          float) through the checked reciprocal fast path of euclid_math.h;
//   false: fp32 throughout, tt = (c*(q-a)) * fl32(1/den): at most 2 ulp from the
//          reference's value (1.2e-7 relative against the 1e-5 bar), a third of
//          the instructions.  The FORWARD value T is bit-exact in both.
template <int D4C, bool FWD, bool BWD, bool EXACT, int WPB>
__global__ __launch_bounds__(64 * WPB) void euclid_pair32_kernel(
    int N, const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top_in, const float* __restrict__ top_diff,
    float* __restrict__ top_out, float* __restrict__ dq, float* __restrict__ da) {
  // N first: with -amdgpu-kernarg-preload-count the leading arguments arrive in SGPRs at wave
  // start, so the loads below do not wait on a scalar fetch of the argument block
  constexpr int NIT = (D4C + 31) / 32;
  constexpr int LASTN = D4C - 32 * (NIT - 1);    // lanes with a float4 in the last slot
  constexpr int H4 = (D4C + 2) / 3, ST4 = 3 * H4;
  __shared__ float4 lds4[FWD ? WPB * 2 * ST4 : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane >> 5, j = lane & 31;
  // No early exit: a wave past the end works on the last pair and stores nothing, so that no
  // branch (and no wait on the kernel arguments) stands between wave start and the loads.
  const int want = (blockIdx.x * WPB + wave) * 2 + grp;
  const bool have = want < N;
  const int row = have ? want : N - 1;
  const float4* q4 = reinterpret_cast<const float4*>(q) + (size_t)row * D4C;
  const float4* a4 = reinterpret_cast<const float4*>(a) + (size_t)row * D4C;
  const bool last_ok = (LASTN >= 32) || (j < LASTN);

  MMS_STAMP(0);
  float T = 0.f;
  if (!FWD) T = top_in[row];
  float g = 0.f;
  if (BWD) g = top_diff[row];
  float4 x[NIT], y[NIT], df[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = (it < NIT - 1 || last_ok) ? j + 32 * it : 0;   // clamp: keep the load unconditional
    x[it] = q4[i];
    y[it] = a4[i];
  }
  MMS_STAMP(1);

  float p1 = 0.f, p2 = 0.f;
  float4* img = lds4 + (wave * 2 + grp) * ST4;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    df[it].x = x[it].x - y[it].x; df[it].y = x[it].y - y[it].y;
    df[it].z = x[it].z - y[it].z; df[it].w = x[it].w - y[it].w;
    if (FWD) {
      const bool valid = (it < NIT - 1) || last_ok;
      float4 s;
      s.x = df[it].x * df[it].x; s.y = df[it].y * df[it].y;
      s.z = df[it].z * df[it].z; s.w = df[it].w * df[it].w;
      if (valid) img[j + 32 * it] = s;
      const float s4 = valid ? (s.x + s.y) + (s.z + s.w) : 0.f;
      const int i = j + 32 * it;
      // tree-sum contributions to the two window centres (segment 0; segments 0-1)
      if (32 * it + 31 < H4) p1 += s4;
      else if (32 * it < H4) p1 += (i < H4) ? s4 : 0.f;
      if (32 * it + 31 < 2 * H4) p2 += s4;
      else if (32 * it < 2 * H4) p2 += (i < 2 * H4) ? s4 : 0.f;
    }
  }
#ifdef MMS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MMS_STAMP(2);
  if (FWD) {
    if (ST4 > D4C && j < ST4 - D4C) img[D4C + j] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_sync();
    SpecSegment<H4> sg;
    sg.load(img + spec_seg32(j) * H4);          // in flight while the window centres are reduced
    p1 = half_wave_sum(p1);
    p2 = half_wave_sum(p2);
    MMS_STAMP(3);
    __builtin_amdgcn_s_setprio(3);
    const float2v start = spec_start32(p1, p2, j);
#if defined(MMS_ABLATE) && MMS_ABLATE >= 12     // dev-only timing ablations (tools/ablate.sh)
    float dist = p2 + sg.v[0].x;
#elif defined(MMS_ABLATE) && MMS_ABLATE == 11
    const float2v end = sg.chain(start);
    float dist = end.x + end.y;
#else
    const float2v end = sg.chain(start);
    bool hit;
    float dist = spec_resolve_halves(start, end, j, &hit);
    if (!hit) {                                 // uniform per half; exact re-walk of this lane's pair
      MMS_COUNT_MISS();
      dist = chain_sum_lds(img, ST4, 0.0f);
    }
#endif
    __builtin_amdgcn_s_setprio(0);
    MMS_STAMP(4);
#if defined(MMS_ABLATE) && MMS_ABLATE >= 13
    T = dist;
#else
    T = 1.0f / (1.0f + sqrtf(dist));            // :106-107
#endif
    // g is pinned as "in a register" HERE, before the store of T: left to the compiler, its first use came
    // after that store (issued under a lane mask, so the wait could not be counted) and every wave sat in
    // s_waitcnt vmcnt(0) until the store was acknowledged
    if (BWD) asm volatile("" : "+v"(g));
    if (j == 0 && have) top_out[row] = T;
  }
  if (!BWD) return;

  float4* dq4 = reinterpret_cast<float4*>(dq) + (size_t)row * D4C;
  float4* da4 = reinterpret_cast<float4*>(da) + (size_t)row * D4C;
  if (!FWD) {                                   // all requests landed before the first store (see euclid_block_kernel)
    asm volatile("" : "+v"(T), "+v"(g));
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      asm volatile("" : "+v"(df[it].x), "+v"(df[it].y), "+v"(df[it].z), "+v"(df[it].w));
  }
  float4 t[NIT];
  if (EXACT) {
    const EuclidCoef k = euclid_coef(T, g);
    bool any_risky = false;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      bool risky;
      t[it] = euclid_tt4_fast(k, df[it], risky);
      any_risky |= risky && ((it < NIT - 1) || last_ok);
    }
    if (any_risky) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) t[it] = euclid_tt4_exact(k, df[it]);
    }
  } else {
    const float c = g * T * T * T;
    const float r = (float)rcp_newton((double)(T - 1.0f) + 1e-9);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      t[it].x = (c * df[it].x) * r; t[it].y = (c * df[it].y) * r;
      t[it].z = (c * df[it].z) * r; t[it].w = (c * df[it].w) * r;
    }
  }
  MMS_STAMP(5);
  if (have) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (!((it < NIT - 1) || last_ok)) break;
    // dq = 0 + tt ; da = 0 + (-tt)   (:176-177 zero, :219-220 accumulate once)
    float4 o0, o1;
    o0.x = 0.f + t[it].x; o0.y = 0.f + t[it].y; o0.z = 0.f + t[it].z; o0.w = 0.f + t[it].w;
    o1.x = 0.f + (-t[it].x); o1.y = 0.f + (-t[it].y); o1.z = 0.f + (-t[it].z); o1.w = 0.f + (-t[it].w);
    stream_store(dq4 + j + 32 * it, o0);
    stream_store(da4 + j + 32 * it, o1);
  }
  }
  MMS_STAMP(6);
#ifdef MMS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MMS_STAMP(7);
}

// ---- the same algorithm with the global-memory side laid out by WORKGROUP, not by pair ----------
// Measured (tools/launchbench.hip, profiles/r02_launchbench.txt): HBM-cold, a launch whose waves each read
// two 512-byte pieces 1200 bytes apart per load instruction (the row-aligned layout above) takes 4.4 us for
// q and a of cfg 2, a launch whose workgroup reads its rows as ONE dense run (instruction `it` of all its
// waves covers a contiguous span) takes 3.6 us -- the speed of a flat one-float4-per-thread read; the
// write side behaves the same (6.5 vs 5.5 us for a backward-shaped launch).  What costs is the order in
// which a CU's requests reach a DRAM page: three visits at different times against one.
// So: a workgroup of WPB waves owns R = 2*WPB consecutive pairs = C = R*D4C consecutive float4 of q and of
// a; thread t loads float4s t, t+T, t+2T ... of that run (and stores dq / da the same way).  The squares
// go to LDS at (pair, column) -- the image layout the chain wants -- and after ONE workgroup barrier each
// half-wave walks its pair's chain exactly as in euclid_pair32_kernel (same predictions, same windows,
// same bits).  A thread's float4s belong to whatever pairs they fall in, so the backward reads T (and the
// per-pair coefficients) by pair index: from LDS when this launch computed T, from top_in otherwise -- a
// backward-only launch has no LDS traffic and no barrier at all.
// SPAN = waves that share one dense run: WPB (the whole workgroup, one barrier) or 1 (each wave reads its own two
// rows as a dense run and synchronises with nobody: the chain phases of a CU's waves then start as their own data
// arrives instead of all at once).
template <int D4C, bool FWD, bool BWD, bool EXACT, int WPB, int SPAN = WPB>
__global__ __launch_bounds__(64 * WPB) void euclid_block_kernel(
    int N, const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top_in, const float* __restrict__ top_diff,
    float* __restrict__ top_out, float* __restrict__ dq, float* __restrict__ da) {
  static_assert(SPAN == WPB || SPAN == 1, "a dense run belongs to the workgroup or to one wave");
  constexpr int T = 64 * SPAN, R = 2 * SPAN, C = R * D4C;
  constexpr int NSPAN = WPB / SPAN;                          // runs per workgroup
  constexpr int NIT = (C + T - 1) / T;                       // float4s per operand per thread
  constexpr int PNIT = (D4C + 31) / 32, LASTN = D4C - 32 * (PNIT - 1);
  constexpr int H4 = (D4C + 2) / 3, ST4 = 3 * H4;
  __shared__ float4 lds4_all[FWD ? NSPAN * R * ST4 : 1];
  __shared__ float Tl_all[(FWD && BWD) ? NSPAN * R : 1];
  const int span = (SPAN == WPB) ? 0 : (int)(threadIdx.x >> 6);
  const int tid = (SPAN == WPB) ? (int)threadIdx.x : (int)(threadIdx.x & 63);
  float4* lds4 = lds4_all + (FWD ? span * R * ST4 : 0);
  float* Tl = Tl_all + ((FWD && BWD) ? span * R : 0);
  const long long run = (long long)blockIdx.x * NSPAN + span;   // index of this dense run
  const long long total4 = (long long)N * D4C;
  const long long b = run * C;
  const float4* q4 = reinterpret_cast<const float4*>(q);
  const float4* a4 = reinterpret_cast<const float4*>(a);

  float4 x[NIT], y[NIT], df[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + T * it;
    long long gi = b + ((NIT * T == C || i < C) ? i : 0);     // clamp: keep the load unconditional
    gi = gi < total4 ? gi : total4 - 1;                       // a run past the end reads the last float4
    x[it] = q4[gi];
    y[it] = a4[gi];
  }
  // per-float4 pair coefficients of a backward-only launch: requested with the operands
  float Tg[NIT], gg[NIT];
  if (BWD) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + T * it;
      long long row = run * R + ((NIT * T == C || i < C) ? i : 0) / D4C;
      row = row < N ? row : N - 1;
      gg[it] = top_diff[row];
      if (!FWD) Tg[it] = top_in[row];
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    df[it].x = x[it].x - y[it].x; df[it].y = x[it].y - y[it].y;
    df[it].z = x[it].z - y[it].z; df[it].w = x[it].w - y[it].w;
    if (FWD) {
      const int i = tid + T * it;
      float4 s;
      s.x = df[it].x * df[it].x; s.y = df[it].y * df[it].y;
      s.z = df[it].z * df[it].z; s.w = df[it].w * df[it].w;
      if (NIT * T == C || i < C) lds4[(ST4 == D4C) ? i : (i / D4C) * ST4 + (i % D4C)] = s;
    }
  }
  if (FWD) {
    if constexpr (ST4 > D4C) {                              // zero tail of each image
      if (tid < R * (ST4 - D4C))
        lds4[(tid / (ST4 - D4C)) * ST4 + D4C + tid % (ST4 - D4C)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (SPAN == WPB) __syncthreads(); else wave_lds_sync();
    const int wave = tid >> 6, lane = tid & 63;
    const int grp = lane >> 5, j = lane & 31;
    const int lp = wave * 2 + grp;
    const long long row = run * R + lp;
    const bool have = row < N;
    const float4* img = lds4 + lp * ST4;
    SpecSegment<H4> sg;
    sg.load(img + spec_seg32(j) * H4);          // in flight while the window centres are formed
    // tree-sum contributions to the two window centres (segment 0; segments 0-1): the same per-lane
    // terms and the same reduction as euclid_pair32_kernel, read back from the image
    const bool last_ok = (LASTN >= 32) || (j < LASTN);
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int it = 0; it < PNIT; ++it) {
      const bool valid = (it < PNIT - 1) || last_ok;
      const float4 s = img[valid ? j + 32 * it : 0];
      const float s4 = valid ? (s.x + s.y) + (s.z + s.w) : 0.f;
      const int i = j + 32 * it;
      if (32 * it + 31 < H4) p1 += s4;
      else if (32 * it < H4) p1 += (i < H4) ? s4 : 0.f;
      if (32 * it + 31 < 2 * H4) p2 += s4;
      else if (32 * it < 2 * H4) p2 += (i < 2 * H4) ? s4 : 0.f;
    }
    p1 = half_wave_sum(p1);
    p2 = half_wave_sum(p2);
    __builtin_amdgcn_s_setprio(3);
    const float2v start = spec_start32(p1, p2, j);
    const float2v end = sg.chain(start);
    bool hit;
    float dist = spec_resolve_halves(start, end, j, &hit);
    if (!hit) {                                 // uniform per half; exact re-walk of this lane's pair
      MMS_COUNT_MISS();
      dist = chain_sum_lds(img, ST4, 0.0f);
    }
    __builtin_amdgcn_s_setprio(0);
    const float Tp = 1.0f / (1.0f + sqrtf(dist));            // :106-107
    if (BWD) {                                  // in registers before the store of T (see euclid_pair32_kernel)
#pragma unroll
      for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(gg[it]));
    }
    if (j == 0) {
      if (have) top_out[row] = Tp;
      if (BWD) Tl[lp] = Tp;
    }
    if (BWD) { if (SPAN == WPB) __syncthreads(); else wave_lds_sync(); }
  }
  if (!BWD) return;

  float4* dq4 = reinterpret_cast<float4*>(dq);
  float4* da4 = reinterpret_cast<float4*>(da);
  if (!FWD) {
    // Everything this thread requested is in registers before its first store.  The stores sit under lane masks
    // (the end of the batch), so the compiler cannot count them: a load consumed after a store became
    // s_waitcnt vmcnt(0), i.e. a wait for the ACKNOWLEDGEMENT of the stores already issued -- in the middle of
    // the store phase of the launch that bounds the headline.
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      asm volatile("" : "+v"(gg[it]), "+v"(Tg[it]));
      asm volatile("" : "+v"(df[it].x), "+v"(df[it].y), "+v"(df[it].z), "+v"(df[it].w));
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + T * it;
    const bool live = (NIT * T == C || i < C) && (b + i < total4);
    const float Tp = FWD ? Tl[((NIT * T == C || i < C) ? i : 0) / D4C] : Tg[it];
    float4 t;
    if (EXACT) {
      const EuclidCoef k = euclid_coef(Tp, gg[it]);
      t = euclid_tt4(k, df[it]);
    } else {
      const float c = gg[it] * Tp * Tp * Tp;
      const float r = (float)rcp_newton((double)(Tp - 1.0f) + 1e-9);
      t.x = (c * df[it].x) * r; t.y = (c * df[it].y) * r;
      t.z = (c * df[it].z) * r; t.w = (c * df[it].w) * r;
    }
    if (live) {
      // dq = 0 + tt ; da = 0 + (-tt)   (:176-177 zero, :219-220 accumulate once)
      float4 o0, o1;
      o0.x = 0.f + t.x; o0.y = 0.f + t.y; o0.z = 0.f + t.z; o0.w = 0.f + t.w;
      o1.x = 0.f + (-t.x); o1.y = 0.f + (-t.y); o1.z = 0.f + (-t.z); o1.w = 0.f + (-t.w);
      stream_store(dq4 + b + i, o0);
      stream_store(da4 + b + i, o1);
    }
  }
}

// How the fp16-storage kernels sum a pair's squares (include/mms.h: mms_set_f16_distance_mode); per calling thread.
static thread_local int t_f16_distance_mode = MMS_F16_DISTANCE_ORDERED;
int f16_distance_mode() { return t_f16_distance_mode; }
void set_f16_distance_mode(int m) { t_f16_distance_mode = m; }

// ---- fp16 storage, fp32 arithmetic (BASELINE cfg 5) --------------------------
// Same wave-centric structure with one pair per wave (cfg 5 is D = 1024): q, a,
// dq, da live in HBM as IEEE half (half the bytes per pair: s = 2 in SURVEY
// 8d's formulas); every half is widened exactly to fp32 on load, ALL arithmetic
// is the fp32 reference arithmetic in the reference order, and only the final
// dq / da are rounded (RNE) to half.  The scores stay fp32.  Hence:
//   top == oracle(fp32(q_half), fp32(a_half)) bit for bit, and
//   dq  == half(oracle dq) bit for bit.
// The reference has no fp16 instantiation (common.hpp:41-44); this is an
// MI355X-side storage format, not a change of the layer's numerics.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// RW pairs per wave (64 / RW lanes each).  At D = 1024 the ordered chain is 55 % of the kernel
// (tools/f16abl.sh: 18.6 us with, 8.3 us without it at 8192 pairs): two pairs per wave would halve its
// VALU issue cost per pair, but the narrower speculation windows (+-12 / +-15 ulp instead of +-25 / +-36)
// miss too often there (see simcross_euclid_rows_f16); a miss re-walks one segment exactly, so results
// never change, only time.
// QUAD (RW == 1 only): the quad-shared chain of euclid_math.h (a quarter of the LDS operand traffic).
// TREE (RW == 1 only): the distance is the TREE sum of the squares that the ordered variants use only to centre
// their speculation windows -- no ordered chain, 8.3 instead of 19 us at cfg 5's shard.  The reference has no
// fp16 instantiation, so there is no reference rounding to reproduce; SURVEY 8(d) holds cfg 5 to 1e-3 relative
// against the fp32 oracle on the fp16-rounded inputs, and this sum is within ~1e-6 of it.  Opt-in
// (mms_set_f16_distance_mode): the default stays the ordered sum, bit-identical to the fp32 layer's.
template <int NIT, int RW, bool BWD, bool QUAD = false, bool TREE = false>
__global__ __launch_bounds__(256) void euclid_rows_wave_f16_kernel(
    const _Float16* __restrict__ q, const _Float16* __restrict__ a,
    const float* __restrict__ top_diff, float* __restrict__ top_out,
    _Float16* __restrict__ dq, _Float16* __restrict__ da, int N, int D8) {
  constexpr int LPR = 64 / RW;
  extern __shared__ float4 lds4[];               // [4 waves][RW images] (euclid_math.h)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + wave) * RW;
  if (row0 >= N) return;
  const int rows = min(RW, N - row0);
  const int n8 = rows * D8;
  const int D4 = 2 * D8;
  const size_t base8 = (size_t)row0 * D8;
  const half8* q8 = reinterpret_cast<const half8*>(q) + base8;
  const half8* a8 = reinterpret_cast<const half8*>(a) + base8;
  const int h4 = QUAD ? quad_h4(D4) : spec_h4(D4), st4 = 3 * h4;
  float4* sq4 = lds4 + (size_t)wave * RW * st4;

  half8 x[NIT], y[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    const int ii = i < n8 ? i : 0;
    x[it] = q8[ii];
    y[it] = a8[ii];
  }
  const int grp = lane / LPR, j = lane % LPR;
  const int grow = min(grp, rows - 1);           // a missing 2nd pair mirrors the 1st (results unused)
  float g = 0.f;
  if (BWD) g = top_diff[row0 + grow];

  float4 df[2 * NIT];
  float tree_total = 0.f;
  float2v pred[RW];                              // per pair: (pred1, pred2) partial sums
#pragma unroll
  for (int r = 0; r < RW; ++r) pred[r] = (float2v){0.f, 0.f};
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    const bool r1 = (RW == 2) && (i >= D8);
    const int ir = r1 ? i - D8 : i;              // half8 index inside its pair
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float4 d;
      d.x = (float)x[it][4 * hh + 0] - (float)y[it][4 * hh + 0];
      d.y = (float)x[it][4 * hh + 1] - (float)y[it][4 * hh + 1];
      d.z = (float)x[it][4 * hh + 2] - (float)y[it][4 * hh + 2];
      d.w = (float)x[it][4 * hh + 3] - (float)y[it][4 * hh + 3];
      df[2 * it + hh] = d;
      float4 s;
      s.x = d.x * d.x; s.y = d.y * d.y; s.z = d.z * d.z; s.w = d.w * d.w;
      const int i4 = 2 * ir + hh;                // float4 index inside the pair's image
      if (!TREE && i < n8) sq4[(r1 ? st4 : 0) + i4] = s;
      const float s4 = (i < n8) ? (s.x + s.y) + (s.z + s.w) : 0.f;
      if (TREE) tree_total += s4;
      float2v c;
      c.x = (i4 < h4) ? s4 : 0.f;
      c.y = (i4 < 2 * h4) ? s4 : 0.f;
      const float2v z2 = {0.f, 0.f};
      pred[0] += r1 ? z2 : c;
      if (RW == 2) pred[RW - 1] += r1 ? c : z2;
    }
  }
  const int npad = st4 - D4;
  if (lane < RW * npad) sq4[(lane / npad) * st4 + D4 + (lane % npad)] = make_float4(0.f, 0.f, 0.f, 0.f);
  float my1 = 0.f, my2 = 0.f;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const float p1 = wave_sum(pred[r].x), p2 = wave_sum(pred[r].y);
    if (r == grow) { my1 = p1; my2 = p2; }
  }
  wave_lds_sync();
#if defined(MMS_F16ABL) && MMS_F16ABL == 1     // dev-only timing ablation (tools/f16bench.hip): no chain
  const float dist = my2;
#else
  const float dist = TREE ? wave_sum(tree_total)
                   : QUAD ? chain_sum_speculative_quad(sq4, h4, my1, my2, lane)
                          : chain_sum_speculative<LPR>(sq4 + grow * st4, D4, my1, my2, j, grp * LPR);
#endif
  const float T = 1.0f / (1.0f + sqrtf(dist));
  if (BWD) asm volatile("" : "+v"(g));          // in a register before the store of T (see euclid_pair32_kernel)
  if (j == 0 && grp < rows) top_out[row0 + grp] = T;
  if (!BWD) return;

  const EuclidCoef mine = euclid_coef(T, g);
  EuclidCoef kr[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {                 // lane r*LPR is a compile-time lane: v_readlane
    kr[r].c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.c), r * LPR));
    const long long dn = __double_as_longlong(mine.den), rc = __double_as_longlong(mine.rcp);
    const unsigned dlo = __builtin_amdgcn_readlane((int)(unsigned)dn, r * LPR);
    const unsigned dhi = __builtin_amdgcn_readlane((int)(unsigned)(dn >> 32), r * LPR);
    const unsigned rlo = __builtin_amdgcn_readlane((int)(unsigned)rc, r * LPR);
    const unsigned rhi = __builtin_amdgcn_readlane((int)(unsigned)(rc >> 32), r * LPR);
    kr[r].den = __longlong_as_double((long long)(((unsigned long long)dhi << 32) | dlo));
    kr[r].rcp = __longlong_as_double((long long)(((unsigned long long)rhi << 32) | rlo));
  }
  half8* dq8 = reinterpret_cast<half8*>(dq) + base8;
  half8* da8 = reinterpret_cast<half8*>(da) + base8;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    if (i >= n8) break;
    const EuclidCoef& k = (RW == 2 && i >= D8) ? kr[RW - 1] : kr[0];
    half8 o0, o1;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const float4 t = euclid_tt4(k, df[2 * it + hh]);
      o0[4 * hh + 0] = (_Float16)(0.f + t.x); o0[4 * hh + 1] = (_Float16)(0.f + t.y);
      o0[4 * hh + 2] = (_Float16)(0.f + t.z); o0[4 * hh + 3] = (_Float16)(0.f + t.w);
      o1[4 * hh + 0] = (_Float16)(0.f + (-t.x)); o1[4 * hh + 1] = (_Float16)(0.f + (-t.y));
      o1[4 * hh + 2] = (_Float16)(0.f + (-t.z)); o1[4 * hh + 3] = (_Float16)(0.f + (-t.w));
    }
#if defined(MMS_F16ABL) && MMS_F16ABL == 2     // dev-only timing ablation: no stores
    if (T == 12345.0f)
#endif
    {
      stream_store_vec(dq8 + i, o0);
      stream_store_vec(da8 + i, o1);
    }
  }
}

// The ORDERED distance at large D without speculation (round 3; D > 400, where the kernel above runs one pair per wave
// and its quad-shared speculative chain is 55 % of the launch: 18.9 us at cfg 5's shard).  One wave per pair loads,
// squares and later differentiates its pair exactly as above; the squares go to LDS as the pair's image, and after one
// workgroup barrier LANE p of wave 0 walks pair p's image front to back -- the reference's d-ascending fp32 sum
// (sim_cross_layer.cpp:100-106) as 4 D4 dependent adds fed by D4 ds_read_b128, about 2.4 us for D = 1024 whatever the
// number of lanes walking.  Eight pairs per workgroup, several workgroups per CU: one workgroup's walk hides behind the
// others' loads and stores (the launch is HBM-bound: 67 MB).  No windows, no misses, no re-walks; bit-identical by
// construction.
template <int NIT, bool BWD, int WPB = 8>
__global__ __launch_bounds__(64 * WPB) void euclid_rows_lanechain_f16_kernel(
    const _Float16* __restrict__ q, const _Float16* __restrict__ a, const float* __restrict__ top_diff,
    float* __restrict__ top_out, _Float16* __restrict__ dq, _Float16* __restrict__ da, int N, int D8) {
  extern __shared__ float4 lc_lds[];             // [WPB pairs][D4 + 1] float4, then WPB floats (the distances)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + wave;
  const int rowc = row < N ? row : N - 1;        // a missing pair mirrors the last one (results unused)
  const int D4 = 2 * D8, st4 = D4 + 1;
  const half8* q8 = reinterpret_cast<const half8*>(q) + (size_t)rowc * D8;
  const half8* a8 = reinterpret_cast<const half8*>(a) + (size_t)rowc * D8;
  float4* img = lc_lds + (size_t)wave * st4;
  float* dist_lds = reinterpret_cast<float*>(lc_lds + (size_t)WPB * st4);

  half8 x[NIT], y[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it, ii = i < D8 ? i : 0;
    x[it] = q8[ii];
    y[it] = a8[ii];
  }
  float g = 0.f;
  if (BWD) g = top_diff[rowc];
  float4 df[2 * NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float4 d;
      d.x = (float)x[it][4 * hh + 0] - (float)y[it][4 * hh + 0];
      d.y = (float)x[it][4 * hh + 1] - (float)y[it][4 * hh + 1];
      d.z = (float)x[it][4 * hh + 2] - (float)y[it][4 * hh + 2];
      d.w = (float)x[it][4 * hh + 3] - (float)y[it][4 * hh + 3];
      df[2 * it + hh] = d;
      float4 sq;
      sq.x = d.x * d.x; sq.y = d.y * d.y; sq.z = d.z * d.z; sq.w = d.w * d.w;
      if (i < D8) img[2 * i + hh] = sq;
    }
  }
  __syncthreads();
  if (wave == 0 && lane < WPB) {
    const float4* mine = lc_lds + (size_t)lane * st4;
    float acc = 0.f;
#pragma unroll 8
    for (int i4 = 0; i4 < D4; ++i4) {
      const float4 v = mine[i4];
      acc += v.x; acc += v.y; acc += v.z; acc += v.w;
    }
    dist_lds[lane] = acc;
  }
  __syncthreads();
  if (row >= N) return;
  const float dist = dist_lds[wave];
  const float T = 1.0f / (1.0f + sqrtf(dist));
  if (BWD) asm volatile("" : "+v"(g));          // in a register before the store of T (see euclid_pair32_kernel)
  if (lane == 0) top_out[row] = T;
  if (!BWD) return;
  const EuclidCoef k = euclid_coef(T, g);
  half8* dq8 = reinterpret_cast<half8*>(dq) + (size_t)row * D8;
  half8* da8 = reinterpret_cast<half8*>(da) + (size_t)row * D8;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    if (i >= D8) break;
    half8 o0, o1;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const float4 t = euclid_tt4(k, df[2 * it + hh]);
      o0[4 * hh + 0] = (_Float16)(0.f + t.x); o0[4 * hh + 1] = (_Float16)(0.f + t.y);
      o0[4 * hh + 2] = (_Float16)(0.f + t.z); o0[4 * hh + 3] = (_Float16)(0.f + t.w);
      o1[4 * hh + 0] = (_Float16)(0.f + (-t.x)); o1[4 * hh + 1] = (_Float16)(0.f + (-t.y));
      o1[4 * hh + 2] = (_Float16)(0.f + (-t.z)); o1[4 * hh + 3] = (_Float16)(0.f + (-t.w));
    }
    stream_store_vec(dq8 + i, o0);
    stream_store_vec(da8 + i, o1);
  }
}

// fp16-STORAGE cosine, W1 = W2 = 1 (round 3; cfg 5's "multi-modal concat embeddings, fp16" with dist_mode 0): one
// wave per pair, a lane holds NIT half8 of q and of a (all 16-byte loads up front, kept for the backward), fp32
// arithmetic on the exactly-widened inputs: three tree sums (the reference's cblas_sdot has no defined order), the
// reference's T = q.a / nq / na with its two successive divisions (sim_cross_layer.cpp:135) and its cached NORMS
// (:118); backward through per-pair factors as cosine_pair32_kernel (c1 = g / n0 / n1, c2 = g T / n0^2, c3 = g T / n1^2,
// IEEE divisions once per pair), gradients stored as RNE halves.  Bytes per pair: reads 2 D s + 4, writes 2 D s + 12.
template <int NIT, bool BWD>
__global__ __launch_bounds__(256) void cosine_rows_wave_f16_kernel(
    const _Float16* __restrict__ q, const _Float16* __restrict__ a, const float* __restrict__ top_diff,
    float* __restrict__ top, float* __restrict__ norm0, float* __restrict__ norm1,
    _Float16* __restrict__ dq, _Float16* __restrict__ da, int N, int D8) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const half8* q8 = reinterpret_cast<const half8*>(q) + (size_t)row * D8;
  const half8* a8 = reinterpret_cast<const half8*>(a) + (size_t)row * D8;
  half8 x[NIT], y[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it, ic = i < D8 ? i : D8 - 1;
    x[it] = q8[ic];
    y[it] = a8[ic];
  }
  float g = BWD ? top_diff[row] : 0.f;
  float sqq = 0.f, saa = 0.f, sqa = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (lane + 64 * it < D8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xf = (float)x[it][e], yf = (float)y[it][e];
        sqq += xf * xf; saa += yf * yf; sqa += xf * yf;
      }
    }
  }
  sqq = wave_sum(sqq); saa = wave_sum(saa); sqa = wave_sum(sqa);
  const float n0 = sqrtf(sqq), n1 = sqrtf(saa);
  const float T = sqa / n0 / n1;                       // two successive divisions (:135)
  if (BWD) asm volatile("" : "+v"(g));
  if (lane == 0) {
    top[row] = T;
    if (norm0) norm0[row] = n0;
    if (norm1) norm1[row] = n1;
  }
  if (!BWD) return;
  const float c1 = g / n0 / n1, c2 = g * T / (n0 * n0), c3 = g * T / (n1 * n1);
  half8* dq8 = reinterpret_cast<half8*>(dq) + (size_t)row * D8;
  half8* da8 = reinterpret_cast<half8*>(da) + (size_t)row * D8;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    if (i < D8) {
      half8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xf = (float)x[it][e], yf = (float)y[it][e];
        o0[e] = (_Float16)(0.f + (c1 * yf - c2 * xf));   // dq = g (a / n0 / n1 - q T / n0^2)   (:239-241)
        o1[e] = (_Float16)(0.f + (c1 * xf - c3 * yf));   // da = g (q / n0 / n1 - a T / n1^2)   (:243-245)
      }
      __builtin_nontemporal_store(o0, dq8 + i);
      __builtin_nontemporal_store(o1, da8 + i);
    }
  }
}

int simcross_cosine_rows_f16(int N, int D, const void* q, const void* a, const float* top_diff, float* top,
                             float* norm0, float* norm1, void* dq, void* da, bool bwd, hipStream_t s) {
  if (N == 0) return MMS_OK;
  if (D % 8 != 0 || D > 2048 || !aligned16(q) || !aligned16(a) || (bwd && (!aligned16(dq) || !aligned16(da))))
    return MMS_ERR_UNSUPPORTED;
  const int D8 = D / 8, nit = (D8 + 63) / 64;
  const unsigned grid = (unsigned)((N + 3) / 4);
  const _Float16* qh = static_cast<const _Float16*>(q);
  const _Float16* ah = static_cast<const _Float16*>(a);
  _Float16* dqh = static_cast<_Float16*>(dq);
  _Float16* dah = static_cast<_Float16*>(da);
#define MMS_COS16(n)                                                                                     \
  case n:                                                                                                \
    if (bwd) hipLaunchKernelGGL((cosine_rows_wave_f16_kernel<n, true>), dim3(grid), dim3(256), 0, s, qh, ah, top_diff, top, \
                                norm0, norm1, dqh, dah, N, D8);                                          \
    else hipLaunchKernelGGL((cosine_rows_wave_f16_kernel<n, false>), dim3(grid), dim3(256), 0, s, qh, ah, top_diff, top,    \
                            norm0, norm1, dqh, dah, N, D8);                                              \
    break;
  switch (nit) { MMS_COS16(1) MMS_COS16(2) MMS_COS16(3) MMS_COS16(4) default: return MMS_ERR_UNSUPPORTED; }
#undef MMS_COS16
  return launch_status();
}

int simcross_euclid_rows_f16(int N, int D, const void* q, const void* a, const float* top_diff,
                             float* top, void* dq, void* da, bool bwd, hipStream_t s) {
  if (N == 0) return MMS_OK;
  if (D % 8 != 0 || D > 2048 || !aligned16(q) || !aligned16(a) || (bwd && (!aligned16(dq) || !aligned16(da))))
    return MMS_ERR_UNSUPPORTED;
  const int D8 = D / 8;
  // Two pairs per wave (32 candidate lanes each) only while the narrower windows hold: at D = 1024 the
  // ordered fp32 sum of a 344-term segment strays sigma ~ 6 ulp from its tree-sum prediction, the +-12 / +-15
  // windows of 32 lanes missed on 3.4 % of pairs (tools/f16abl.sh) and the exact re-walks ate the gain at
  // cfg 5's shard size (19.5 vs 18.6 us; 125 vs 149 us at 65536 pairs).  Same limit as the fp32 kernels.
  const int rw = D <= 400 ? 2 : 1;
  const int nit = (rw * D8 + 63) / 64;
  const unsigned grid = (unsigned)((N + 4 * rw - 1) / (4 * rw));
  // one pair per wave: the quad-shared chain (MMS_F16_CHAIN=lanes selects the per-lane chain, for A/B timing)
  static const bool quad_off = [] { const char* e = std::getenv("MMS_F16_CHAIN"); return e && !std::strcmp(e, "lanes"); }();   // "quad": the quad chain
  const bool quad = rw == 1 && !quad_off;
  static const bool chain_env = [] { return std::getenv("MMS_F16_CHAIN") != nullptr; }();
  const bool chain_old = rw == 2 || chain_env;      // (rw == 2 is handled above this branch)
  const bool tree = f16_distance_mode() == MMS_F16_DISTANCE_TREE;
  const size_t lds = (size_t)4 * rw * 3 * (quad ? quad_h4(2 * D8) : (2 * D8 + 2) / 3) * sizeof(float4);
  const _Float16* qh = static_cast<const _Float16*>(q);
  const _Float16* ah = static_cast<const _Float16*>(a);
  _Float16* dqh = static_cast<_Float16*>(dq);
  _Float16* dah = static_cast<_Float16*>(da);
#define MMS_F16_LAUNCH(n, r)                                                                          \
  do {                                                                                                \
    if (bwd)                                                                                          \
      hipLaunchKernelGGL((euclid_rows_wave_f16_kernel<n, r, true>), dim3(grid), dim3(256), lds, s,    \
                         qh, ah, top_diff, top, dqh, dah, N, D8);                                     \
    else                                                                                              \
      hipLaunchKernelGGL((euclid_rows_wave_f16_kernel<n, r, false>), dim3(grid), dim3(256), lds, s,   \
                         qh, ah, top_diff, top, dqh, dah, N, D8);                                     \
  } while (0)
  if (tree) {
    // one pair per wave whatever D: no speculation windows to fit
    const int nit1 = (D8 + 63) / 64;
    const unsigned grid1 = (unsigned)((N + 3) / 4);
#define MMS_F16_TREE(n)                                                                               \
  do {                                                                                                \
    if (bwd)                                                                                          \
      hipLaunchKernelGGL((euclid_rows_wave_f16_kernel<n, 1, true, false, true>), dim3(grid1),         \
                         dim3(256), 16, s, qh, ah, top_diff, top, dqh, dah, N, D8);                   \
    else                                                                                              \
      hipLaunchKernelGGL((euclid_rows_wave_f16_kernel<n, 1, false, false, true>), dim3(grid1),        \
                         dim3(256), 16, s, qh, ah, top_diff, top, dqh, dah, N, D8);                   \
  } while (0)
    switch (nit1) {
      case 1: MMS_F16_TREE(1); break;
      case 2: MMS_F16_TREE(2); break;
      case 3: MMS_F16_TREE(3); break;
      default: MMS_F16_TREE(4); break;
    }
#undef MMS_F16_TREE
  } else if (rw == 2) {
    switch (nit) {
      case 1: MMS_F16_LAUNCH(1, 2); break;
      case 2: MMS_F16_LAUNCH(2, 2); break;
      case 3: MMS_F16_LAUNCH(3, 2); break;
      default: MMS_F16_LAUNCH(4, 2); break;
    }
  } else if (!chain_old) {
    // D > 400, ordered: lane p of wave 0 walks pair p's image (MMS_F16_CHAIN=quad / lanes select round 2's speculative
    // chains, for A/B timing)
    static const int wpb = [] { const char* e = std::getenv("MMS_F16_LC_WPB"); const int v = e ? std::atoi(e) : 8;
                                return v == 4 || v == 16 ? v : 8; }();        // dev-only A/B of the workgroup size
    const unsigned gridw = (unsigned)((N + wpb - 1) / wpb);
    const size_t ldsw = ((size_t)wpb * (2 * D8 + 1) + 4) * sizeof(float4);
#define MMS_F16_LCW(n, w)                                                                             \
  do {                                                                                                \
    static bool once = [] {                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&euclid_rows_lanechain_f16_kernel<n, true, w>),  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&euclid_rows_lanechain_f16_kernel<n, false, w>), \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      return true;                                                                                    \
    }();                                                                                              \
    (void)once;                                                                                       \
    if (bwd)                                                                                          \
      hipLaunchKernelGGL((euclid_rows_lanechain_f16_kernel<n, true, w>), dim3(gridw), dim3(64 * w), ldsw, \
                         s, qh, ah, top_diff, top, dqh, dah, N, D8);                                  \
    else                                                                                              \
      hipLaunchKernelGGL((euclid_rows_lanechain_f16_kernel<n, false, w>), dim3(gridw), dim3(64 * w), ldsw, \
                         s, qh, ah, top_diff, top, dqh, dah, N, D8);                                  \
  } while (0)
#define MMS_F16_LC(n)                                                                                 \
  do {                                                                                                \
    if (wpb == 4) MMS_F16_LCW(n, 4); else if (wpb == 16) MMS_F16_LCW(n, 16); else MMS_F16_LCW(n, 8);  \
  } while (0)
    switch (nit) {
      case 1: MMS_F16_LC(1); break;
      case 2: MMS_F16_LC(2); break;
      case 3: MMS_F16_LC(3); break;
      default: MMS_F16_LC(4); break;
    }
#undef MMS_F16_LCW
#undef MMS_F16_LC
  } else if (!quad) {
    switch (nit) {
      case 1: MMS_F16_LAUNCH(1, 1); break;
      case 2: MMS_F16_LAUNCH(2, 1); break;
      case 3: MMS_F16_LAUNCH(3, 1); break;
      default: MMS_F16_LAUNCH(4, 1); break;
    }
  } else {
#define MMS_F16_QUAD(n)                                                                               \
  do {                                                                                                \
    if (bwd)                                                                                          \
      hipLaunchKernelGGL((euclid_rows_wave_f16_kernel<n, 1, true, true>), dim3(grid), dim3(256), lds, \
                         s, qh, ah, top_diff, top, dqh, dah, N, D8);                                  \
    else                                                                                              \
      hipLaunchKernelGGL((euclid_rows_wave_f16_kernel<n, 1, false, true>), dim3(grid), dim3(256),     \
                         lds, s, qh, ah, top_diff, top, dqh, dah, N, D8);                             \
  } while (0)
    switch (nit) {
      case 1: MMS_F16_QUAD(1); break;
      case 2: MMS_F16_QUAD(2); break;
      case 3: MMS_F16_QUAD(3); break;
      default: MMS_F16_QUAD(4); break;
    }
#undef MMS_F16_QUAD
  }
#undef MMS_F16_LAUNCH
  return launch_status();
}

// ---- generic fallback (any D, any alignment): workgroup of ROWS pairs --------
// Forward (BWD=false) or forward+backward (BWD=true) for W1=W2=1, Euclidean.
// LDS: diff[ROWS*D] floats (dynamic) + per-row coefficient slots.
template <int ROWS, int THREADS, bool BWD>
__global__ __launch_bounds__(THREADS) void euclid_rows_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top_diff, float* __restrict__ top,
    float* __restrict__ dq, float* __restrict__ da, int N, int D) {
  extern __shared__ float4 lds_raw[];
  float* diff = reinterpret_cast<float*>(lds_raw);
  __shared__ float cs[ROWS];
  __shared__ double dens[ROWS];

  const int row0 = blockIdx.x * ROWS;
  const int rows = min(ROWS, N - row0);
  const size_t base = (size_t)row0 * D;
  const int total = rows * D;

  for (int i = threadIdx.x; i < total; i += THREADS) diff[i] = q[base + i] - a[base + i];
  __syncthreads();

  // One lane per pair: the reference's d-ascending fp32 chain (:100-106).
  if (threadIdx.x < rows) {
    const float* r = diff + threadIdx.x * D;
    float dist = 0.f;
    for (int d = 0; d < D; ++d) dist += r[d] * r[d];
    dist = sqrtf(dist);
    const float T = 1.0f / (1.0f + dist);
    top[row0 + threadIdx.x] = T;
    if (BWD) {
      const EuclidCoef k = euclid_coef(T, top_diff[row0 + threadIdx.x]);
      cs[threadIdx.x] = k.c;
      dens[threadIdx.x] = k.den;
    }
  }
  if (!BWD) return;
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += THREADS) {
    const int r = i / D;
    const float t = euclid_tt_exact(cs[r], dens[r], diff[i]);
    dq[base + i] = 0.f + t;
    da[base + i] = 0.f + (-t);
  }
}

// Backward alone, generic fallback: pure streaming.
__global__ __launch_bounds__(256) void euclid_rows_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top, const float* __restrict__ top_diff,
    float* __restrict__ dq, float* __restrict__ da, int total, int D) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int r = i / D;
    const EuclidCoef k = euclid_coef(top[r], top_diff[r]);
    const float t = euclid_tt_exact(k.c, k.den, q[i] - a[i]);
    dq[i] = 0.f + t;
    da[i] = 0.f + (-t);
  }
}

// Cosine, W1=W2=1: one wave per pair; three dot products reduced with a fixed
// butterfly (the reference's order here is whatever its BLAS does).
// BWD fuses the backward with a known top_diff.
template <bool VEC4, bool FWD, bool BWD>
__global__ __launch_bounds__(256) void cosine_rows_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top_diff, float* __restrict__ top,
    float* __restrict__ norm0, float* __restrict__ norm1,
    float* __restrict__ dq, float* __restrict__ da, int N, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* qr = q + (size_t)row * D;
  const float* ar = a + (size_t)row * D;
  float g = 0.f;
  if (BWD) g = top_diff[row];                    // requested up front; pinned before the stores of the forward
  float T, n0, n1;
  if (FWD) {
    float sqq = 0.f, saa = 0.f, sqa = 0.f;
    if (VEC4) {
      const float4* q4 = reinterpret_cast<const float4*>(qr);
      const float4* a4 = reinterpret_cast<const float4*>(ar);
      for (int i = lane; i < (D >> 2); i += 64) {
        const float4 x = q4[i], y = a4[i];
        sqq += x.x * x.x; sqq += x.y * x.y; sqq += x.z * x.z; sqq += x.w * x.w;
        saa += y.x * y.x; saa += y.y * y.y; saa += y.z * y.z; saa += y.w * y.w;
        sqa += x.x * y.x; sqa += x.y * y.y; sqa += x.z * y.z; sqa += x.w * y.w;
      }
    } else {
      for (int i = lane; i < D; i += 64) {
        const float x = qr[i], y = ar[i];
        sqq += x * x; saa += y * y; sqa += x * y;
      }
    }
    sqq = wave_sum(sqq); saa = wave_sum(saa); sqa = wave_sum(sqa);
    n0 = sqrtf(sqq);
    n1 = sqrtf(saa);
    T = sqa / n0 / n1;  // two successive divisions (:135)
    if (BWD) asm volatile("" : "+v"(g));
    if (lane == 0) { top[row] = T; norm0[row] = n0; norm1[row] = n1; }
  } else {
    T = top[row]; n0 = norm0[row]; n1 = norm1[row];
  }
  if (!BWD) return;
  float* dqr = dq + (size_t)row * D;
  float* dar = da + (size_t)row * D;
  // :239-245   dq += g*(a/n0/n1 - q*T/(n0*n0)) ; da += g*(q/n0/n1 - a*T/(n1*n1))
  const float n00 = n0 * n0, n11 = n1 * n1;
  if (VEC4) {
    const float4* q4 = reinterpret_cast<const float4*>(qr);
    const float4* a4 = reinterpret_cast<const float4*>(ar);
    float4* dq4 = reinterpret_cast<float4*>(dqr);
    float4* da4 = reinterpret_cast<float4*>(dar);
    for (int i = lane; i < (D >> 2); i += 64) {
      const float4 x = q4[i], y = a4[i];
      float4 o0, o1;
      o0.x = 0.f + g * (y.x / n0 / n1 - x.x * T / n00);
      o0.y = 0.f + g * (y.y / n0 / n1 - x.y * T / n00);
      o0.z = 0.f + g * (y.z / n0 / n1 - x.z * T / n00);
      o0.w = 0.f + g * (y.w / n0 / n1 - x.w * T / n00);
      o1.x = 0.f + g * (x.x / n0 / n1 - y.x * T / n11);
      o1.y = 0.f + g * (x.y / n0 / n1 - y.y * T / n11);
      o1.z = 0.f + g * (x.z / n0 / n1 - y.z * T / n11);
      o1.w = 0.f + g * (x.w / n0 / n1 - y.w * T / n11);
      stream_store(dq4 + i, o0);
      stream_store(da4 + i, o1);
    }
  } else {
    for (int i = lane; i < D; i += 64) {
      const float x = qr[i], y = ar[i];
      dqr[i] = 0.f + g * (y / n0 / n1 - x * T / n00);
      dar[i] = 0.f + g * (x / n0 / n1 - y * T / n11);
    }
  }
}

// Cosine, W1=W2=1, the GloVe widths (D = 100 / 200 / 300): the data movement of
// euclid_pair32_kernel -- 32 lanes per pair, two pairs per wave, every 16-byte load of q and a issued
// up front and kept in registers for the backward, half-wave DPP reductions, streaming stores --
// without the ordered chain (the reference's dot products are cblas_sdot: no defined order, 1e-5
// contract).  The backward multiplies by per-pair factors 1/n0/n1, T/n0^2, T/n1^2 computed once
// (IEEE divisions) instead of dividing per element (:239-245 written out costs six divisions per
// (q_d, a_d)): a few ulp from the reference's expression, inside the same 1e-5.
template <int D4C, bool FWD, bool BWD, int WPB>
__global__ __launch_bounds__(64 * WPB) void cosine_pair32_kernel(
    int N, const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top_diff, float* __restrict__ top, float* __restrict__ norm0,
    float* __restrict__ norm1, float* __restrict__ dq, float* __restrict__ da) {
  constexpr int NIT = (D4C + 31) / 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 5, j = lane & 31;
  const int want = (blockIdx.x * WPB + wave) * 2 + grp;
  const bool have = want < N;
  const int row = have ? want : N - 1;
  const float4* q4 = reinterpret_cast<const float4*>(q) + (size_t)row * D4C;
  const float4* a4 = reinterpret_cast<const float4*>(a) + (size_t)row * D4C;
  float4 x[NIT], y[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = j + 32 * it;
    const int ii = i < D4C ? i : 0;              // clamp: keep the load unconditional
    x[it] = q4[ii];
    y[it] = a4[ii];
  }
  float T, n0, n1;
  if (FWD) {
    float sqq = 0.f, saa = 0.f, sqa = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (j + 32 * it < D4C) {
        const float4 u = x[it], v = y[it];
        sqq += u.x * u.x; sqq += u.y * u.y; sqq += u.z * u.z; sqq += u.w * u.w;
        saa += v.x * v.x; saa += v.y * v.y; saa += v.z * v.z; saa += v.w * v.w;
        sqa += u.x * v.x; sqa += u.y * v.y; sqa += u.z * v.z; sqa += u.w * v.w;
      }
    }
    sqq = half_wave_sum(sqq); saa = half_wave_sum(saa); sqa = half_wave_sum(sqa);
    n0 = sqrtf(sqq);                             // the NORM is cached, as on the CPU (:118)
    n1 = sqrtf(saa);
    T = sqa / n0 / n1;                           // two successive divisions (:135)
    if (j == 0 && have) { top[row] = T; norm0[row] = n0; norm1[row] = n1; }
  } else {
    T = top[row]; n0 = norm0[row]; n1 = norm1[row];
  }
  if (!BWD) return;
  const float g = top_diff[row];
  const float inv01 = 1.0f / n0 / n1, cq = T / (n0 * n0), ca = T / (n1 * n1);
  float4* dq4 = reinterpret_cast<float4*>(dq) + (size_t)row * D4C;
  float4* da4 = reinterpret_cast<float4*>(da) + (size_t)row * D4C;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = j + 32 * it;
    if (i < D4C && have) {
      const float4 u = x[it], v = y[it];
      float4 o0, o1;
      o0.x = 0.f + g * (v.x * inv01 - u.x * cq); o0.y = 0.f + g * (v.y * inv01 - u.y * cq);
      o0.z = 0.f + g * (v.z * inv01 - u.z * cq); o0.w = 0.f + g * (v.w * inv01 - u.w * cq);
      o1.x = 0.f + g * (u.x * inv01 - v.x * ca); o1.y = 0.f + g * (u.y * inv01 - v.y * ca);
      o1.z = 0.f + g * (u.z * inv01 - v.z * ca); o1.w = 0.f + g * (u.w * inv01 - v.w * ca);
      stream_store(dq4 + i, o0);
      stream_store(da4 + i, o1);
    }
  }
}

static bool pair32_width(int D);
template <bool FWD, bool BWD>
static void launch_cosine_pair32(const float* q, const float* a, const float* top_diff, float* top,
                                 float* norm0, float* norm1, float* dq, float* da, int N, int D,
                                 hipStream_t s) {
  constexpr int WPB = 8;
  const unsigned grid = (unsigned)((N + 2 * WPB - 1) / (2 * WPB));
#define MMS_C32(d4)                                                                                 \
  case 4 * d4:                                                                                      \
    hipLaunchKernelGGL((cosine_pair32_kernel<d4, FWD, BWD, WPB>), dim3(grid), dim3(64 * WPB), 0, s, \
                       N, q, a, top_diff, top, norm0, norm1, dq, da);                               \
    break;
  switch (D) { MMS_C32(25) MMS_C32(50) MMS_C32(75) }
#undef MMS_C32
}

// ============================== cross geometry ==============================

// L2 norms of `rows` rows of length D: one wave per row (cosine, general W).
// Word id stored as a float (Caffe feeds ids as Dtype), clamped into the table like embed_fwd_kernel.
__device__ __forceinline__ int gather_id(float v, int K) {
  const int i = (int)v;
  return i < 0 ? 0 : (i >= K ? K - 1 : i);
}

// index != nullptr: row `row` is table row index[row] of x (K rows) -- the Embed gather fused in.
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ x,
                                                       float* __restrict__ nrm,
                                                       long long rows, int D,
                                                       const float* __restrict__ index, int K,
                                                       const float* __restrict__ ebias = nullptr) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* r = x + (index ? (long long)gather_id(index[row], K) : row) * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) { const float v = ebias ? ebias[i] + r[i] : r[i]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) nrm[row] = sqrtf(s);
}

// Forward for general W1 x W2, MODE 0 (cosine; norms precomputed) or 1.
// One wave per (pair, j-tile, k-tile); tile = (8*RJ) x (8*RK) outputs,
// lane (lj = lane>>3, lk = lane&7) owns outputs j = j0+lj+8*rj, k = k0+lk+8*rk.
//
// CrossAcc: the register tile of one lane.  Accumulators live in packed pairs (v_pk_add_f32 /
// v_pk_mul_f32 work on two fp32 per lane and per issue slot; each half is an ordinary IEEE op, so the
// d-ascending sums keep their bits): columns (2p, 2p+1) of a row pair up; with RK odd the last column
// pairs rows (2p, 2p+1); with both odd one scalar is left.
template <int RJ, int RK, int MODE>
struct CrossAcc {
  static constexpr int PK = RK / 2, PJ = (RK & 1) ? RJ / 2 : 0;
  static constexpr bool LAST = (RK & 1) && (RJ & 1);
  float2v accp[RJ][PK > 0 ? PK : 1], accq[PJ > 0 ? PJ : 1];
  float accs;

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int x = 0; x < RJ; ++x)
#pragma unroll
      for (int y = 0; y < (PK > 0 ? PK : 1); ++y) accp[x][y] = (float2v){0.f, 0.f};
#pragma unroll
    for (int x = 0; x < (PJ > 0 ? PJ : 1); ++x) accq[x] = (float2v){0.f, 0.f};
    accs = 0.f;
  }
  // dn steps of d: qrow / arow point at this lane's first row of each operand in LDS (column 0 of
  // the staged span), rows 8 apart are 8*ls floats apart
  __device__ __forceinline__ void accumulate(const float* qrow, const float* arow, int ls, int dn) {
    for (int dd = 0; dd < dn; ++dd) {
      float qv[RJ], av[RK];
#pragma unroll
      for (int x = 0; x < RJ; ++x) qv[x] = qrow[8 * x * ls + dd];
#pragma unroll
      for (int y = 0; y < RK; ++y) av[y] = arow[8 * y * ls + dd];
#pragma unroll
      for (int x = 0; x < RJ; ++x) {
        const float2v q2 = (float2v){qv[x], qv[x]};
#pragma unroll
        for (int y = 0; y < PK; ++y) {
          const float2v a2 = (float2v){av[2 * y], av[2 * y + 1]};
          if (MODE == 1) {
            const float2v df = q2 - a2;
            accp[x][y] += df * df;
          } else {
            accp[x][y] += q2 * a2;
          }
        }
      }
      if (PJ > 0) {
        const float2v a2 = (float2v){av[RK - 1], av[RK - 1]};
#pragma unroll
        for (int x = 0; x < PJ; ++x) {
          const float2v q2 = (float2v){qv[2 * x], qv[2 * x + 1]};
          if (MODE == 1) {
            const float2v df = q2 - a2;
            accq[x] += df * df;
          } else {
            accq[x] += q2 * a2;
          }
        }
      }
      if (LAST) {
        if (MODE == 1) {
          const float df = qv[RJ - 1] - av[RK - 1];
          accs += df * df;
        } else {
          accs += qv[RJ - 1] * av[RK - 1];
        }
      }
    }
  }
  __device__ __forceinline__ float get(int x, int y) const {
    if (y < 2 * PK) return (y & 1) ? accp[x][y / 2].y : accp[x][y / 2].x;
    if (x < 2 * PJ) return (x & 1) ? accq[x / 2].y : accq[x / 2].x;
    return accs;
  }
  // T from the sums (:106-107 / :131-136) and the stores of this lane's outputs
  __device__ __forceinline__ void finish(float* __restrict__ top, const float* __restrict__ norm0,
                                         const float* __restrict__ norm1, int n, int j0, int k0,
                                         int lj, int lk, int W1, int W2) const {
    // cosine: this lane's RJ + RK norms are in registers before its first store (a load between two stores
    // waits, with vmcnt(0), for the acknowledgement of the store in front of it)
    float n0v[RJ], n1v[RK];
    if (MODE != 1) {
#pragma unroll
      for (int x = 0; x < RJ; ++x) n0v[x] = norm0[(size_t)n * W1 + min(j0 + lj + 8 * x, W1 - 1)];
#pragma unroll
      for (int y = 0; y < RK; ++y) n1v[y] = norm1[(size_t)n * W2 + min(k0 + lk + 8 * y, W2 - 1)];
#pragma unroll
      for (int x = 0; x < RJ; ++x) asm volatile("" : "+v"(n0v[x]));
#pragma unroll
      for (int y = 0; y < RK; ++y) asm volatile("" : "+v"(n1v[y]));
    }
#pragma unroll
    for (int x = 0; x < RJ; ++x) {
      const int j = j0 + lj + 8 * x;
      if (j >= W1) continue;
#pragma unroll
      for (int y = 0; y < RK; ++y) {
        const int k = k0 + lk + 8 * y;
        if (k >= W2) continue;
        float T;
#if defined(MMS_XABL) && MMS_XABL == 1   // dev-only timing ablation (tools/crossbench.hip): no sqrt / divide
        if (MODE == 1) {
          T = get(x, y);
        } else
#endif
        if (MODE == 1) {
          T = 1.0f / (1.0f + sqrtf(get(x, y)));
        } else {
          T = get(x, y) / n0v[x] / n1v[y];
        }
#if defined(MMS_XABL) && MMS_XABL == 3   // dev-only timing ablation: no stores
        if (T != T + 1.0f && T == 12345.678f)
#endif
        top[((size_t)n * W1 + j) * W2 + k] = T;
      }
    }
  }
};

// Embed fused into the load: with iq != nullptr, q and a are both the embedding
// TABLE (K x D) and row j of pair n is table row iq[n*W1 + j] (ia likewise) -- the (N, W, D) blobs
// the Embed layer would write and SimCross read back never exist.
struct CrossGather {
  const float* iq;
  const float* ia;
  int K;
  const float* bias;     // the Embed layer's bias (D floats) or nullptr: row value = bias[d] + table[id][d], the
                         // one rounding of embed_layer.cpp:146-151 (gemm with alpha = beta = 1)
};

// Generic staging: q/a are staged DC floats of d at a time in LDS with stride DC+1 (bank =
// (row + d) mod 32: conflict-free across rows, broadcast within a row).
template <int RJ, int RK, int MODE>
__global__ __launch_bounds__(256) void cross_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ norm0, const float* __restrict__ norm1,
    float* __restrict__ top, int N, int W1, int W2, int D, int tilesJ, int tilesK, CrossGather gt) {
  constexpr int TJ = 8 * RJ, TK = 8 * RK, DC = 32, LS = DC + 1;
  __shared__ float qs[4][TJ * LS];
  __shared__ float as[4][TK * LS];
  __shared__ int rowoff[4][TJ + TK];               // gather: element offset of each tile row in the table
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long work = (long long)blockIdx.x * 4 + wave;
  const long long total = (long long)N * tilesJ * tilesK;
  const bool valid = work < total;
  const long long w = valid ? work : 0;
  const int n = (int)(w / (tilesJ * tilesK));
  const int rem = (int)(w % (tilesJ * tilesK));
  const int j0 = (rem / tilesK) * TJ, k0 = (rem % tilesK) * TK;
  const int lj = lane >> 3, lk = lane & 7;
  const float* qn = q + (size_t)n * W1 * D;
  const float* an = a + (size_t)n * W2 * D;

  CrossAcc<RJ, RK, MODE> acc;
  acc.clear();

  const int lrow = lane >> 5, lcol = lane & 31;
  // Staging: every load of a chunk is issued (clamped, hence unconditional, addresses) before the
  // first LDS write; out-of-range elements are zeroed when written.  Written as `ok ? load : 0` the
  // compiler emitted load / wait / write per row: 40 serialized round trips per chunk (18 of 38 us
  // at 1517 x 40 x 40 x 50).  The NEXT chunk's loads are issued right after the LDS writes of the
  // current one, so they are in flight behind its arithmetic.
  float rq[TJ / 2], ra[TK / 2];
  const bool gather = gt.iq != nullptr;
  if (gather) {                                  // word ids of this tile's rows -> table offsets, once
    for (int r = lane; r < TJ + TK; r += 64) {
      const bool isq = r < TJ;
      const int rr = isq ? min(j0 + r, W1 - 1) : min(k0 + r - TJ, W2 - 1);
      const float id = isq ? gt.iq[(size_t)n * W1 + rr] : gt.ia[(size_t)n * W2 + rr];
      rowoff[wave][r] = gather_id(id, gt.K) * D;
    }
    wave_lds_sync();
  }
  auto fetch = [&](int d0) {
    const int col = min(d0 + lcol, D - 1);
    if (gather) {
#pragma unroll
      for (int r = 0; r < TJ; r += 2) rq[r / 2] = q[(size_t)rowoff[wave][r + lrow] + col];
#pragma unroll
      for (int r = 0; r < TK; r += 2) ra[r / 2] = a[(size_t)rowoff[wave][TJ + r + lrow] + col];
      if (gt.bias) {
        const float bv = gt.bias[col];
#pragma unroll
        for (int r = 0; r < TJ; r += 2) rq[r / 2] = bv + rq[r / 2];
#pragma unroll
        for (int r = 0; r < TK; r += 2) ra[r / 2] = bv + ra[r / 2];
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < TJ; r += 2) rq[r / 2] = qn[(size_t)min(j0 + r + lrow, W1 - 1) * D + col];
#pragma unroll
    for (int r = 0; r < TK; r += 2) ra[r / 2] = an[(size_t)min(k0 + r + lrow, W2 - 1) * D + col];
  };
  fetch(0);
  for (int d0 = 0; d0 < D; d0 += DC) {
    const int dn = min(DC, D - d0);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TJ; r += 2)
      qs[wave][(r + lrow) * LS + lcol] = (valid && j0 + r + lrow < W1 && lcol < dn) ? rq[r / 2] : 0.f;
#pragma unroll
    for (int r = 0; r < TK; r += 2)
      as[wave][(r + lrow) * LS + lcol] = (valid && k0 + r + lrow < W2 && lcol < dn) ? ra[r / 2] : 0.f;
    __syncthreads();
    if (d0 + DC < D) fetch(d0 + DC);
#if defined(MMS_XABL) && MMS_XABL == 2   // dev-only timing ablation: no arithmetic
    if (dn > 0) continue;
#endif
    acc.accumulate(&qs[wave][lj * LS], &as[wave][lk * LS], LS, dn);
  }
  if (!valid) return;
  acc.finish(top, norm0, norm1, n, j0, k0, lj, lk, W1, W2);
}

// "Pair image" staging for small D (the driver's default 50-d vectors): a wave owns one whole pair,
// W1 = 8*RJ and W2 = 8*RK exactly, and the (W1 x D) and (W2 x D) blocks of q and a -- contiguous in
// memory -- are COPIED to LDS as they are, 16 bytes per lane per load, row stride D (no padding, no
// chunking over d, no index arithmetic).  Rows 8 apart must fall into different banks for the
// broadcast reads of accumulate(): gcd(D, 64) <= 8 (launch_cross_fwd checks).  Against the generic
// staging at 1517 x 40 x 40 x 50 this replaces 80 4-byte load instructions and 80 predicated LDS
// writes per wave by 16 + 16.  Waves are independent: 2 per workgroup, 32 KB of LDS each.
// D is a template parameter: with a run-time row stride the ten row addresses of accumulate() are
// recomputed per d step (VALU-bound loop: +15 % time); compiled in, they are immediate offsets.
template <int RJ, int RK, int MODE, int D>
__global__ __launch_bounds__(128) void cross_fwd_image_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ norm0, const float* __restrict__ norm1,
    float* __restrict__ top, int N, CrossGather gt) {
  constexpr int W1 = 8 * RJ, W2 = 8 * RK;
  extern __shared__ float4 img4[];                 // [2 waves][(W1 + W2) * D / 4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int work = blockIdx.x * 2 + wave;
  const bool valid = work < N;
  const int n = valid ? work : N - 1;
  const int nq4 = W1 * D / 4, na4 = W2 * D / 4;
  float4* qs4 = img4 + (size_t)wave * (nq4 + na4);
  float4* as4 = qs4 + nq4;
  const float4* q4 = reinterpret_cast<const float4*>(q + (size_t)n * W1 * D);
  const float4* a4 = reinterpret_cast<const float4*>(a + (size_t)n * W2 * D);
  if (gt.iq != nullptr) {
    // Embed fused in: image row r is table row id[r]; rows are D floats = 8-byte aligned for even D, so
    // the copy runs in float2.  Per batch: the ids of 8 + 8 elements, then their 16 loads, then the writes.
    constexpr int R2 = D / 2;                      // float2 per row
    float2* qs2 = reinterpret_cast<float2*>(qs4);
    float2* as2 = reinterpret_cast<float2*>(as4);
    const float2* t2 = reinterpret_cast<const float2*>(q);
    const float* iq = gt.iq + (size_t)n * W1;
    const float* ia = gt.ia + (size_t)n * W2;
    constexpr int NQ2 = W1 * R2, NA2 = W2 * R2, NMAX = NQ2 > NA2 ? NQ2 : NA2;
    for (int e0 = 0; e0 < NMAX; e0 += 512) {
      float idq[8], ida[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 64 * u + lane;
        idq[u] = iq[min(e, NQ2 - 1) / R2];
        ida[u] = ia[min(e, NA2 - 1) / R2];
      }
      float2 rq[8], ra[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 64 * u + lane;
        rq[u] = t2[(size_t)gather_id(idq[u], gt.K) * R2 + min(e, NQ2 - 1) % R2];
        ra[u] = t2[(size_t)gather_id(ida[u], gt.K) * R2 + min(e, NA2 - 1) % R2];
      }
      if (gt.bias) {
        const float2* b2 = reinterpret_cast<const float2*>(gt.bias);   // D even: the row pitch makes this 8-byte aligned with the table
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + 64 * u + lane;
          const float2 bq = b2[min(e, NQ2 - 1) % R2], ba = b2[min(e, NA2 - 1) % R2];
          rq[u].x = bq.x + rq[u].x; rq[u].y = bq.y + rq[u].y;
          ra[u].x = ba.x + ra[u].x; ra[u].y = ba.y + ra[u].y;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 64 * u + lane;
        if (e < NQ2) qs2[e] = rq[u];
        if (e < NA2) as2[e] = ra[u];
      }
    }
  } else
  // copy in batches of 4 + 4 loads (all issued before the first LDS write of the batch)
  for (int i0 = 0; i0 < max(nq4, na4); i0 += 256) {
    float4 rq[4], ra[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 64 * u + lane;
      rq[u] = q4[min(i, nq4 - 1)];
      ra[u] = a4[min(i, na4 - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 64 * u + lane;
      if (i < nq4) qs4[i] = rq[u];
      if (i < na4) as4[i] = ra[u];
    }
  }
  wave_lds_sync();
  const int lj = lane >> 3, lk = lane & 7;
  CrossAcc<RJ, RK, MODE> acc;
  acc.clear();
#if !(defined(MMS_XABL) && MMS_XABL == 2)
  acc.accumulate(reinterpret_cast<const float*>(qs4) + lj * D, reinterpret_cast<const float*>(as4) + lk * D, D, D);
#endif
  if (!valid) return;
  acc.finish(top, norm0, norm1, n, 0, 0, lj, lk, W1, W2);
}

// Backward for general W1 x W2, MODE 0/1: one workgroup per pair n.  Thread
// owns one (j,d) of dq and walks k ascending, then one (k,d) of da and walks
// j ascending -- the reference's accumulation order (:209-223), so Euclidean
// is bit-exact.  q/a/top rows of one n stay L1/L2-resident across the walk.
template <int MODE>
__global__ __launch_bounds__(256) void cross_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top, const float* __restrict__ top_diff,
    const float* __restrict__ norm0, const float* __restrict__ norm1,
    float* __restrict__ dq, float* __restrict__ da, int W1, int W2, int D) {
  const int n = blockIdx.x;
  const float* qn = q + (size_t)n * W1 * D;
  const float* an = a + (size_t)n * W2 * D;
  const float* Tn = top + (size_t)n * W1 * W2;
  const float* gn = top_diff + (size_t)n * W1 * W2;
  float* dqn = dq + (size_t)n * W1 * D;
  float* dan = da + (size_t)n * W2 * D;
  const float* n0n = MODE == 0 ? norm0 + (size_t)n * W1 : nullptr;
  const float* n1n = MODE == 0 ? norm1 + (size_t)n * W2 : nullptr;

  for (int e = threadIdx.x; e < W1 * D; e += 256) {
    const int j = e / D, d = e - j * D;
    const float qv = qn[e];
    float acc = 0.f;
    if (MODE == 1) {
      for (int k = 0; k < W2; ++k) {
        const EuclidCoef kc = euclid_coef(Tn[j * W2 + k], gn[j * W2 + k]);
        acc += euclid_tt_exact(kc.c, kc.den, qv - an[(size_t)k * D + d]);
      }
    } else {
      const float nrm0 = n0n[j];
      for (int k = 0; k < W2; ++k) {
        const float nrm1 = n1n[k];
        acc += gn[j * W2 + k] * (an[(size_t)k * D + d] / nrm0 / nrm1 -
                                 qv * Tn[j * W2 + k] / (nrm0 * nrm0));
      }
    }
    dqn[e] = acc;
  }
  for (int e = threadIdx.x; e < W2 * D; e += 256) {
    const int k = e / D, d = e - k * D;
    const float av = an[e];
    float acc = 0.f;
    if (MODE == 1) {
      for (int j = 0; j < W1; ++j) {
        const EuclidCoef kc = euclid_coef(Tn[j * W2 + k], gn[j * W2 + k]);
        acc += -euclid_tt_exact(kc.c, kc.den, qn[(size_t)j * D + d] - av);
      }
    } else {
      const float nrm1 = n1n[k];
      for (int j = 0; j < W1; ++j) {
        const float nrm0 = n0n[j];
        acc += gn[j * W2 + k] * (qn[(size_t)j * D + d] / nrm0 / nrm1 -
                                 av * Tn[j * W2 + k] / (nrm1 * nrm1));
      }
    }
    dan[e] = acc;
  }
}

// Tiled backward for general W1 x W2 (the fast path when the per-pair tables fit
// LDS).  One workgroup per (pair, 32-wide d chunk):
//   * per-(j,k) coefficient tables are built ONCE per workgroup in LDS
//     (Euclid: c, den, 1/den; cosine: g, 1/(n0 n1), T/n0^2, T/n1^2);
//   * the q / a chunk is staged in LDS (stride 33: conflict-free);
//   * a thread owns one (j,d) of dq and walks k ascending, then one (k,d) of da
//     walking j ascending -- the reference's accumulation order (:209-223).
// Euclid stays bit-exact (euclid_tt's self-checking reciprocal path).  Cosine
// multiplies by precomputed reciprocals instead of dividing per term: it is
// held to 1e-5 like everything that is BLAS-ordered in the reference.
constexpr int kBwdDC = 32;

template <int MODE, bool EXACT>
__global__ __launch_bounds__(256) void cross_bwd_tiled_kernel(
    const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ top, const float* __restrict__ top_diff,
    const float* __restrict__ norm0, const float* __restrict__ norm1,
    float* __restrict__ dq, float* __restrict__ da, int W1, int W2, int D, int nchunks, int split) {
  extern __shared__ double lds_d[];
  constexpr int LS = kBwdDC + 1;
  // split: blockIdx.x = (n * nchunks + chunk) * 2 + pass -- the dq pass and the da pass of one
  // (pair, chunk) run as separate workgroups (twice the parallelism; small batches).
  // Otherwise one workgroup does both and the tables are built once (large batches).
  const int bid = split ? (blockIdx.x >> 1) : blockIdx.x;
  const bool do_dq = !split || (blockIdx.x & 1) == 0, do_da = !split || (blockIdx.x & 1) == 1;
  const int n = bid / nchunks, chunk = bid % nchunks;
  const int d0 = chunk * kBwdDC, dn = min(kBwdDC, D - d0);
  const int JK = W1 * W2;
  // carve: doubles first (8-byte aligned), then floats
  double* t_den = lds_d;                          // MODE 1: [JK]
  double* t_rcp = lds_d + (MODE == 1 ? JK : 0);   // MODE 1: [JK]
  float* fbase = reinterpret_cast<float*>(lds_d + (MODE == 1 ? 2 * JK : 0));
  float* t_c = fbase;                             // MODE 1: c       MODE 0: g
  float* t_i01 = fbase + JK;                      // MODE 0: 1/(n0 n1)
  float* t_b1 = fbase + 2 * JK;                   // MODE 0: T/n0^2
  float* t_b2 = fbase + 3 * JK;                   // MODE 0: T/n1^2
  float* qs = fbase + (MODE == 1 ? JK : 4 * JK);
  float* as = qs + W1 * LS;
  // fp32 backward arithmetic (include/mms.h): the double tables are not needed; fl32(1/den)
  // lives in their place
  float* t_r = reinterpret_cast<float*>(lds_d);

  const float* qn = q + (size_t)n * W1 * D;
  const float* an = a + (size_t)n * W2 * D;
  const float* Tn = top + (size_t)n * JK;
  const float* gn = top_diff + (size_t)n * JK;

  for (int e0 = threadIdx.x; MODE == 1 && e0 < JK; e0 += 256 * 8) {
    float tv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int e = min(e0 + 256 * u, JK - 1); tv[u] = Tn[e]; gv[u] = gn[e]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      if (e >= JK) break;
      const EuclidCoef k = euclid_coef(tv[u], gv[u]);
      t_c[e] = k.c;
      if (EXACT) { t_den[e] = k.den; t_rcp[e] = k.rcp; } else { t_r[e] = (float)k.rcp; }
    }
  }
  for (int e = threadIdx.x; MODE != 1 && e < JK; e += 256) {
    if (MODE == 1) {
    } else {
      const int j = e / W2, kk = e - j * W2;
      const float n0 = norm0[(size_t)n * W1 + j], n1 = norm1[(size_t)n * W2 + kk];
      t_c[e] = gn[e];
      t_i01[e] = 1.0f / n0 / n1;
      t_b1[e] = Tn[e] / (n0 * n0);
      t_b2[e] = Tn[e] / (n1 * n1);
    }
  }
  // eight unconditional (clamped) loads per thread in flight before the first LDS write; a rolled
  // `ok ? load : 0` loop paid one memory round trip per iteration
  auto stage = [&](const float* src, float* dst, int W) {
    for (int base = 0; base < W * kBwdDC; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + threadIdx.x + 256 * u;
        v[u] = src[(size_t)min(e >> 5, W - 1) * D + min(d0 + (e & 31), D - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + threadIdx.x + 256 * u;
        if (e < W * kBwdDC) dst[(e >> 5) * LS + (e & 31)] = (e & 31) < dn ? v[u] : 0.f;
      }
    }
  };
  stage(qn, qs, W1);
  stage(an, as, W2);
  __syncthreads();

  float* dqn = dq + (size_t)n * W1 * D;
  float* dan = da + (size_t)n * W2 * D;
  if (MODE == 1 && !EXACT) {
    // fp32 arithmetic: a thread owns FOUR consecutive d of one row, so the per-(j,k) coefficients
    // c and fl32(1/den) are read from LDS once per four terms (6 LDS reads per 4 terms instead of
    // 12: this loop is LDS-bandwidth-bound); every sum still runs over k (or j) ascending.
    for (int e = threadIdx.x; do_dq && e < W1 * (kBwdDC / 4); e += 256) {
      const int j = e >> 3, dd0 = (e & 7) * 4;
      if (dd0 >= dn) continue;
      float qv[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) qv[u] = qs[j * LS + dd0 + u];
      for (int k = 0; k < W2; ++k) {
        const float c = t_c[j * W2 + k], r = t_r[j * W2 + k];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += (c * (qv[u] - as[k * LS + dd0 + u])) * r;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dd0 + u < dn) dqn[(size_t)j * D + d0 + dd0 + u] = acc[u];
    }
    for (int e = threadIdx.x; do_da && e < W2 * (kBwdDC / 4); e += 256) {
      const int k = e >> 3, dd0 = (e & 7) * 4;
      if (dd0 >= dn) continue;
      float av[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) av[u] = as[k * LS + dd0 + u];
      for (int j = 0; j < W1; ++j) {
        const float c = t_c[j * W2 + k], r = t_r[j * W2 + k];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += -((c * (qs[j * LS + dd0 + u] - av[u])) * r);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dd0 + u < dn) dan[(size_t)k * D + d0 + dd0 + u] = acc[u];
    }
    return;
  }
  for (int e = threadIdx.x; do_dq && e < W1 * kBwdDC; e += 256) {
    const int j = e >> 5, dd = e & 31;
    if (dd >= dn) continue;
    const float qv = qs[j * LS + dd];
    float acc = 0.f;
    for (int k = 0; k < W2; ++k) {
      const int t = j * W2 + k;
      const float av = as[k * LS + dd];
      if (MODE == 1 && EXACT) {
        EuclidCoef kc;
        kc.c = t_c[t]; kc.den = t_den[t]; kc.rcp = t_rcp[t];
        acc += euclid_tt(kc, qv - av);
      } else if (MODE == 1) {
        acc += (t_c[t] * (qv - av)) * t_r[t];
      } else {
        acc += t_c[t] * (av * t_i01[t] - qv * t_b1[t]);
      }
    }
    dqn[(size_t)j * D + d0 + dd] = acc;
  }
  for (int e = threadIdx.x; do_da && e < W2 * kBwdDC; e += 256) {
    const int k = e >> 5, dd = e & 31;
    if (dd >= dn) continue;
    const float av = as[k * LS + dd];
    float acc = 0.f;
    for (int j = 0; j < W1; ++j) {
      const int t = j * W2 + k;
      const float qv = qs[j * LS + dd];
      if (MODE == 1 && EXACT) {
        EuclidCoef kc;
        kc.c = t_c[t]; kc.den = t_den[t]; kc.rcp = t_rcp[t];
        acc += -euclid_tt(kc, qv - av);
      } else if (MODE == 1) {
        acc += -((t_c[t] * (qv - av)) * t_r[t]);
      } else {
        acc += t_c[t] * (qv * t_i01[t] - av * t_b2[t]);
      }
    }
    dan[(size_t)k * D + d0 + dd] = acc;
  }
}

static size_t cross_bwd_tiled_lds(int mode, int W1, int W2) {
  const size_t JK = (size_t)W1 * W2;
  const size_t tables = mode == 1 ? JK * (8 + 8 + 4) : JK * 16;
  return tables + (size_t)(W1 + W2) * (kBwdDC + 1) * sizeof(float) + 16;
}

// ---- Euclidean cross-geometry backward for MANY pairs of narrow word grids (cfg 4's 1517 x 40 x 40 x 50) ------
// cross_bwd_tiled_kernel computes every term tt[j,k,d] twice (once in the k-ordered sum of dq[j,d], once in
// the j-ordered sum of da[k,d]) and wastes 44 % of its second 32-wide d chunk at D = 50.  Here ONE WAVE owns a
// pair and a LANE owns a column d: the lane walks j (outer) and k (inner) over all W1*W2 terms of its column,
// each computed ONCE; dq[j,d] is the running sum over k inside one j (k ascending, as :209-223), and the W2
// accumulators da[k,d] stay in registers across the j loop (j ascending) -- the reference's accumulation
// orders, so the Euclidean results keep their bits.  The per-(j,k) coefficients are wave-uniform: built once
// into LDS (c and fl32(1/den), or c, den, 1/den for the reference rounding) and read back as broadcasts.
// W2 is a compile-time constant (even; the widths the dispatcher instantiates -- the reference pads sentences to
// one length, 40 in network_v4): the k loop is straight-line code, two k per packed sub / mul / mul, with all
// of a row's coefficient reads in flight together.  Other widths keep cross_bwd_tiled_kernel.
// NW = 2 (fp32 arithmetic only): TWO waves per pair, wave w takes the rows j of half w -- its own rows of the
// coefficient table, its own rows of dq (complete, k ascending as before), and a partial da over its rows; da is
// the sum of the two partials (one association away from the j-ascending sum: inside the mode's 2-ulp-per-term
// contract, not bit-identical -- the reference-rounding mode keeps one wave per pair).  1517 pairs are 1.5 waves
// per SIMD with one wave per pair -- the launch takes as long as the SIMDs that hold two -- and 3 with two.
template <int W2C, bool EXACT, int NW = 1>
__global__ __launch_bounds__(64 * NW) void cross_bwd_lane_kernel(
    const float* __restrict__ q, const float* __restrict__ a, const float* __restrict__ top,
    const float* __restrict__ top_diff, float* __restrict__ dq, float* __restrict__ da, int W1, int D) {
  static_assert(W2C % 2 == 0, "two k per packed operation");
  static_assert(NW == 1 || !EXACT, "the reference-rounding mode sums da in j order: one wave per pair");
  constexpr int KP = W2C / 2;
  extern __shared__ __attribute__((aligned(16))) double lds_lane[];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int JH = (W1 + NW - 1) / NW;                       // rows per wave
  const int jb = min(wv * JH, W1), je = min(jb + JH, W1);
  const int JK = W1 * W2C;
  // tables (index e = j*W2C + k, the blob's own order): EXACT: double den[], double rcp[], float c[];
  // otherwise float2 (c, fl32(1/den))[]
  double* t_den = lds_lane;
  double* t_rcp = lds_lane + (EXACT ? JK : 0);
  float* t_c = reinterpret_cast<float*>(lds_lane + (EXACT ? 2 * JK : 0));
  // fp32 mode: W2C/2 float4 (c_k, c_k+1, r_k, r_k+1) per row j
  const float* Tn = top + (size_t)n * JK;
  const float* gn = top_diff + (size_t)n * JK;
  const int eb = jb * W2C, ee = je * W2C;                  // this wave's rows of the table
  for (int e0 = eb + lane; e0 < ee; e0 += 64 * 8) {
    float tv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int e = min(e0 + 64 * u, JK - 1); tv[u] = Tn[e]; gv[u] = gn[e]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 64 * u;
      if (e >= ee) break;
      const EuclidCoef kc = euclid_coef(tv[u], gv[u]);
      if (EXACT) { t_c[e] = kc.c; t_den[e] = kc.den; t_rcp[e] = kc.rcp; }
      else {                                             // per k pair: (c_k, c_k+1, r_k, r_k+1)
        float* f = reinterpret_cast<float*>(lds_lane);
        const int p4 = (e >> 1) * 4 + (e & 1);           // W2C even: pairs never straddle rows
        f[p4] = kc.c;
        f[p4 + 2] = (float)kc.rcp;
      }
    }
  }
  const bool live = lane < D;
  const int d = live ? lane : D - 1;
  const float* qn = q + (size_t)n * W1 * D + d;
  const float* an = a + (size_t)n * W2C * D + d;
  float2v av[KP], acc[KP];
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) {
    av[kp].x = an[(size_t)(2 * kp) * D];
    av[kp].y = an[(size_t)(2 * kp + 1) * D];
    acc[kp] = (float2v){0.f, 0.f};
  }
  wave_lds_sync();
  float* dqn = dq + (size_t)n * W1 * D + d;
  if (EXACT) {
    float qv = qn[0];
    for (int j = 0; j < W1; ++j) {
      const float qnext = qn[(size_t)min(j + 1, W1 - 1) * D];
      float accq = 0.f;
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        EuclidCoef k0, k1;
        const int e = j * W2C + 2 * kp;
        k0.c = t_c[e]; k0.den = t_den[e]; k0.rcp = t_rcp[e];
        k1.c = t_c[e + 1]; k1.den = t_den[e + 1]; k1.rcp = t_rcp[e + 1];
        float2v tt;
        tt.x = euclid_tt(k0, qv - av[kp].x);
        tt.y = euclid_tt(k1, qv - av[kp].y);
        accq += tt.x;
        accq += tt.y;
        acc[kp] = acc[kp] - tt;                            // da += -tt, j ascending
      }
      if (live) dqn[(size_t)j * D] = accq;
      qv = qnext;
    }
  } else {
    typedef float float4v __attribute__((ext_vector_type(4)));
    const float4v* tab = reinterpret_cast<const float4v*>(lds_lane);
    // a row's W2C/2 (c0, c1, r0, r1) entries are read together at the top of its iteration (16-byte broadcast
    // reads); the next q value is requested one iteration ahead.  Measured alternatives at 1517 x 40 x 40 x 50
    // (this form: 39.5 us; cross_bwd_tiled_kernel: 62): a second row buffer (ping-pong) needs 290 VGPRs = one
    // wave per SIMD, 61 us; re-loading each entry right after its use 44 us; q staged through LDS as well 42.5 us.
    // What bounds it is the LDS return path -- every wave reads its whole 12.8 KB table for all 64 lanes, 18 us
    // of LDS cycles per CU -- next to the VALU issue of the SIMDs that hold two of the 1517 waves.  The
    // coefficients on the SCALAR path instead (a table in a workspace written by a first launch, read through
    // wave-uniform addresses: five s_load_dwordx16 per row, SGPR operands of the packed multiplies) was built and
    // measured: 61 us -- 80 SGPRs per row leave no room to run the loads ahead, so each row exposes its
    // scalar-cache misses (the 19 MB table streams through once).
    float qv = qn[(size_t)min(jb, W1 - 1) * D];
    for (int j = jb; j < je; ++j) {
      const float qnext = qn[(size_t)min(j + 1, W1 - 1) * D];
      const float2v qq = {qv, qv};
      float4v cr[KP];
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) cr[kp] = tab[(size_t)j * KP + kp];
      float accq = 0.f;
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        const float2v c = {cr[kp].x, cr[kp].y}, r = {cr[kp].z, cr[kp].w};
        const float2v tt = (c * (qq - av[kp])) * r;
        accq += tt.x;
        accq += tt.y;
        acc[kp] = acc[kp] - tt;                            // da += -tt, j ascending
      }
      if (live) dqn[(size_t)j * D] = accq;
      qv = qnext;
    }
  }
  float* dan = da + (size_t)n * W2C * D + d;
  if (NW == 2) {                                             // wave 1 hands its partial da to wave 0 through LDS
    float2v* part = reinterpret_cast<float2v*>(reinterpret_cast<float*>(lds_lane) + 2 * JK);   // behind the table
    if (wv == 1) {
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) part[kp * 64 + lane] = acc[kp];
    }
    __syncthreads();
    if (wv == 1) return;
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) acc[kp] = acc[kp] + part[kp * 64 + lane];
  }
  if (live) {
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      dan[(size_t)(2 * kp) * D] = acc[kp].x;
      dan[(size_t)(2 * kp + 1) * D] = acc[kp].y;
    }
  }
}
static size_t cross_bwd_lane_lds(bool exact, int W1, int W2) {
  // fp32 mode: the table, then the second wave's partial da (W2 floats per lane)
  return (size_t)W1 * W2 * (exact ? 8 + 8 + 4 : 8) + (exact ? 0 : (size_t)W2 * 64 * sizeof(float)) + 16;
}
// the lane kernel pays when there are enough pairs to give every SIMD a wave and the grids are narrow
static bool cross_bwd_lane_ok(bool exact, int N, int W1, int W2, int D) {
  const bool width = W2 == 8 || W2 == 16 || W2 == 20 || W2 == 24 || W2 == 32 || W2 == 40 || W2 == 48;
  // reference rounding: 20 bytes of coefficients per (j,k); beyond ~16 KB per wave the table limits occupancy and
  // the tiled kernel wins (1517 x 40 x 40 x 50: 236 vs 183 us; 4096 x 20 x 20 x 50: 50 vs 92 us)
  return width && D <= 64 && N >= 512 && cross_bwd_lane_lds(exact, W1, W2) <= (exact ? 16 : 64) * 1024;
}

// ================================ dispatch ==================================

template <int MODE>
static void launch_cross_fwd(const float* q, const float* a, const float* n0,
                             const float* n1, float* top, int N, int W1, int W2,
                             int D, hipStream_t s, CrossGather gt = CrossGather{nullptr, nullptr, 0, nullptr}) {
  // Register tile per lane: as large as possible (fewer LDS reads per flop) while the
  // launch still has enough waves to occupy the chip (small N: smaller tiles, more waves).
  auto r_cap = [](int w, int cap) { int r = (w + 7) / 8; return r > cap ? cap : r; };
  int rj = 1, rk = 1, tilesJ = 1, tilesK = 1;
  for (int cap = 5; cap >= 1; --cap) {
    rj = r_cap(W1, cap); rk = r_cap(W2, cap);
    tilesJ = (W1 + 8 * rj - 1) / (8 * rj); tilesK = (W2 + 8 * rk - 1) / (8 * rk);
    if ((long long)N * tilesJ * tilesK >= 1024) break;
  }
  const long long work = (long long)N * tilesJ * tilesK;
  // small D: the whole pair as one LDS image per wave (cross_fwd_image_kernel)
  {
    auto gcd64 = [](int d) { int g = 64; while (d % g) g >>= 1; return g; };
    const size_t img = (size_t)(W1 + W2) * D * sizeof(float);
    constexpr int DI = 50;   // the driver's default embedding width (do_trec_qa_clean.py -d 50)
    const bool fits = W1 % 8 == 0 && W2 % 8 == 0 && W1 / 8 <= 5 && W2 / 8 <= 5 && N >= 1024 &&
                      D == DI && aligned16(q) && aligned16(a) && gcd64(D) <= 8 && 2 * img <= 64 * 1024;
    static_assert(DI % 2 == 0, "gather copies float2");
    if (fits) {
      const unsigned g2 = (unsigned)((N + 1) / 2);
#define MMS_IMG_CASE(J, K)                                                                      \
  if (W1 == 8 * J && W2 == 8 * K) {                                                             \
    hipLaunchKernelGGL((cross_fwd_image_kernel<J, K, MODE, DI>), dim3(g2), dim3(128), 2 * img,  \
                       s, q, a, n0, n1, top, N, gt);                                            \
    return;                                                                                     \
  }
#define MMS_IMG_ROW(J) MMS_IMG_CASE(J, 1) MMS_IMG_CASE(J, 2) MMS_IMG_CASE(J, 3) MMS_IMG_CASE(J, 4) MMS_IMG_CASE(J, 5)
      MMS_IMG_ROW(1) MMS_IMG_ROW(2) MMS_IMG_ROW(3) MMS_IMG_ROW(4) MMS_IMG_ROW(5)
#undef MMS_IMG_ROW
#undef MMS_IMG_CASE
    }
  }
  const unsigned grid = (unsigned)((work + 3) / 4);
#define MMS_CROSS_CASE(J, K)                                                        \
  if (rj == J && rk == K) {                                                         \
    hipLaunchKernelGGL((cross_fwd_kernel<J, K, MODE>), dim3(grid), dim3(256), 0, s, \
                       q, a, n0, n1, top, N, W1, W2, D, tilesJ, tilesK, gt);        \
    return;                                                                         \
  }
#define MMS_CROSS_ROW(J) MMS_CROSS_CASE(J, 1) MMS_CROSS_CASE(J, 2) MMS_CROSS_CASE(J, 3) \
                         MMS_CROSS_CASE(J, 4) MMS_CROSS_CASE(J, 5)
  MMS_CROSS_ROW(1) MMS_CROSS_ROW(2) MMS_CROSS_ROW(3) MMS_CROSS_ROW(4) MMS_CROSS_ROW(5)
#undef MMS_CROSS_ROW
#undef MMS_CROSS_CASE
}

constexpr int kRows = 8;       // pairs per workgroup in the generic rows kernels
constexpr int kRowsThreads = 256;

static bool all_aligned16(const void* p0, const void* p1, const void* p2, const void* p3) {
  return aligned16(p0) && aligned16(p1) && (!p2 || aligned16(p2)) && (!p3 || aligned16(p3));
}
static bool vec4_ok(int D, const void* p0, const void* p1, const void* p2, const void* p3) {
  return (D % 4 == 0) && all_aligned16(p0, p1, p2, p3);
}

// generic rows kernels: LDS needed; fall back to the cross kernels above ~64 KB.
static size_t rows_lds_bytes(int D) { return (size_t)kRows * D * sizeof(float); }
static bool rows_fit(int D) { return rows_lds_bytes(D) <= 64 * 1024; }

// wave kernel: RW pairs per wave (2 up to D = 400, 1 beyond: the speculation
// window in ulps must grow with sqrt(D), see euclid_math.h), NIT = ceil(RW*D/4 / 64)
// 16-byte loads per operand per lane.
static int wave_rw(int D) { return D <= 400 ? 2 : 1; }
static int wave_nit(int D) { return (wave_rw(D) * (D / 4) + 63) / 64; }
static bool wave_ok(int D, const void* p0, const void* p1, const void* p2, const void* p3) {
  return vec4_ok(D, p0, p1, p2, p3) && D <= 1024;
}

// Backward arithmetic of the Euclidean term (include/mms.h: mms_set_euclid_backward_mode).  The mode
// belongs to the calling thread (Caffe drives each GPU from its own thread); a thread that never set it
// takes the process default from the environment.
static std::atomic<int> g_euclid_bwd_default{-1};
static thread_local int t_euclid_bwd_mode = -1;
int euclid_backward_mode() {
  if (t_euclid_bwd_mode >= 0) return t_euclid_bwd_mode;
  int m = g_euclid_bwd_default.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = std::getenv("MMS_EUCLID_BWD");
    m = (e && (!std::strcmp(e, "reference") || !std::strcmp(e, "exact") || !std::strcmp(e, "1")))
            ? MMS_EUCLID_BWD_REFERENCE : MMS_EUCLID_BWD_FP32;
    g_euclid_bwd_default.store(m, std::memory_order_relaxed);
  }
  return m;
}
void set_euclid_backward_mode(int m) { t_euclid_bwd_mode = m; }

// widths with a specialised kernel: 100-d, 200-d and 300-d GloVe (D4 = 25, 50, 75)
static bool pair32_width(int D) { return D == 300 || D == 200 || D == 100; }

template <bool FWD, bool BWD, int WPB>
static void launch_pair32w(const float* q, const float* a, const float* top_in, const float* top_diff,
                           float* top_out, float* dq, float* da, int N, int D, hipStream_t s) {
  const unsigned grid = (unsigned)((N + 2 * WPB - 1) / (2 * WPB));
  const bool exact = BWD && euclid_backward_mode() == MMS_EUCLID_BWD_REFERENCE;
  // Which global-memory layout (same results bit for bit; tests/test_gpu_parity.py runs both for every kind
  // of launch).  Measured at cfg 2, HBM-cold, graph-replayed (tools/layoutab.sh, profiles/r02_layout_ab.txt):
  //   backward-only launch      row-aligned 6.27 us, workgroup-dense 4.87 us  -> dense
  //   forward-only launch       4.87 vs 4.92 us: the chain's tail, not the read pattern, bounds it -> row-aligned
  //   fused forward+backward    5.65 vs 6.0 us: waves free of workgroup barriers spread the store phase -> row-aligned
  //   Forward launch then Backward launch (what a Net issues): 8.80 us both row-aligned -> 7.97 us forward
  //   row-aligned + backward dense.  Both must map workgroup b to the SAME pairs (same waves per workgroup):
  //   the backward then finds q and a in the L2 of the XCD that read them in the forward; mismatched maps cost
  //   0.6 us.
  // Dev switch for A/B timing: MMS_EUCLID_LAYOUT_{FWD,BWD,FUSED} = pair | block | wave (dense run per wave).
  static const int layout = [] {
    const char* e = std::getenv(FWD && BWD ? "MMS_EUCLID_LAYOUT_FUSED" : FWD ? "MMS_EUCLID_LAYOUT_FWD" : "MMS_EUCLID_LAYOUT_BWD");
    if (e) return !std::strcmp(e, "pair") ? 0 : (!std::strcmp(e, "wave") ? 2 : 1);
    return FWD ? 0 : 1;
  }();
#define MMS_P32_GO(K, ...)                                                                          \
  do {                                                                                               \
    if (exact)                                                                                       \
      hipLaunchKernelGGL((K<__VA_ARGS__, FWD, BWD, true, WPB>), dim3(grid), dim3(64 * WPB), 0, s, N, \
                         q, a, top_in, top_diff, top_out, dq, da);                                   \
    else                                                                                             \
      hipLaunchKernelGGL((K<__VA_ARGS__, FWD, BWD, false, WPB>), dim3(grid), dim3(64 * WPB), 0, s,   \
                         N, q, a, top_in, top_diff, top_out, dq, da);                                \
  } while (0)
#define MMS_P32(d4)                                                                                  \
  case 4 * d4:                                                                                       \
    if (layout == 0) MMS_P32_GO(euclid_pair32_kernel, d4);                                           \
    else if (layout == 1) MMS_P32_GO(euclid_block_kernel, d4);                                       \
    else if (exact)                                                                                  \
      hipLaunchKernelGGL((euclid_block_kernel<d4, FWD, BWD, true, WPB, 1>), dim3(grid),              \
                         dim3(64 * WPB), 0, s, N, q, a, top_in, top_diff, top_out, dq, da);          \
    else                                                                                             \
      hipLaunchKernelGGL((euclid_block_kernel<d4, FWD, BWD, false, WPB, 1>), dim3(grid),             \
                         dim3(64 * WPB), 0, s, N, q, a, top_in, top_diff, top_out, dq, da);          \
    break;
  switch (D) { MMS_P32(25) MMS_P32(50) MMS_P32(75) }
#undef MMS_P32
#undef MMS_P32_GO
}

// Eight waves (16 pairs) per workgroup: N = 4096 is then 256 workgroups, one per CU, two waves
// per SIMD -- measured 3 % faster HBM-cold than 512 workgroups of four waves (dispatch ramp).
template <bool FWD, bool BWD>
static void launch_pair32(const float* q, const float* a, const float* top_in, const float* top_diff,
                          float* top_out, float* dq, float* da, int N, int D, hipStream_t s) {
  // dev switch for A/B timing (tools/layers_probe.py): MMS_PAIR32_WPB_FWD / _BWD / _FUSED = 2, 4, 8 or 16
  static const int wpb = [] {
    const char* e = std::getenv(FWD && BWD ? "MMS_PAIR32_WPB_FUSED" : FWD ? "MMS_PAIR32_WPB_FWD" : "MMS_PAIR32_WPB_BWD");
    return e ? std::atoi(e) : 8;
  }();
  switch (wpb) {
    case 2: launch_pair32w<FWD, BWD, 2>(q, a, top_in, top_diff, top_out, dq, da, N, D, s); break;
    case 4: launch_pair32w<FWD, BWD, 4>(q, a, top_in, top_diff, top_out, dq, da, N, D, s); break;
    case 16: launch_pair32w<FWD, BWD, 16>(q, a, top_in, top_diff, top_out, dq, da, N, D, s); break;
    default: launch_pair32w<FWD, BWD, 8>(q, a, top_in, top_diff, top_out, dq, da, N, D, s); break;
  }
}

template <bool FWD, bool BWD>
static void launch_rows_wave(const float* q, const float* a, const float* top_in,
                             const float* top_diff, float* top_out, float* dq, float* da, int N,
                             int D, hipStream_t s) {
  if (pair32_width(D)) {
    launch_pair32<FWD, BWD>(q, a, top_in, top_diff, top_out, dq, da, N, D, s);
    return;
  }
  const int D4 = D / 4;
  const int rw = wave_rw(D);
  const unsigned grid = (unsigned)((N + 4 * rw - 1) / (4 * rw));
  const size_t lds = FWD ? (size_t)4 * rw * 3 * ((D4 + 2) / 3) * sizeof(float4) : 0;
#define MMS_NIT_CASE(n, r)                                                                      \
  case n:                                                                                       \
    hipLaunchKernelGGL((euclid_rows_wave_kernel<n, r, FWD, BWD>), dim3(grid), dim3(256), lds, s, \
                       q, a, top_in, top_diff, top_out, dq, da, N, D4);                         \
    break;
  if (rw == 2) {
    switch (wave_nit(D)) { MMS_NIT_CASE(1, 2) MMS_NIT_CASE(2, 2) MMS_NIT_CASE(3, 2) MMS_NIT_CASE(4, 2) }
  } else {
    switch (wave_nit(D)) { MMS_NIT_CASE(1, 1) MMS_NIT_CASE(2, 1) MMS_NIT_CASE(3, 1) MMS_NIT_CASE(4, 1) }
  }
#undef MMS_NIT_CASE
}

int simcross_elementwise_forward(int mode, int N, int W1, int W2, int D,
                                 const float* q, const float* a, float* top,
                                 float* norm0, float* norm1, hipStream_t s) {
  if (N == 0) return MMS_OK;
  const bool rows = (W1 == 1 && W2 == 1);
  if (mode == 1) {
    if (rows && wave_ok(D, q, a, nullptr, nullptr)) {
      launch_rows_wave<true, false>(q, a, nullptr, nullptr, top, nullptr, nullptr, N, D, s);
    } else if (rows && rows_fit(D)) {
      const unsigned grid = (N + kRows - 1) / kRows;
      hipLaunchKernelGGL((euclid_rows_kernel<kRows, kRowsThreads, false>), dim3(grid),
                         dim3(kRowsThreads), rows_lds_bytes(D), s, q, a, nullptr, top, nullptr,
                         nullptr, N, D);
    } else {
      launch_cross_fwd<1>(q, a, nullptr, nullptr, top, N, W1, W2, D, s);
    }
  } else {
    if (rows && pair32_width(D) && vec4_ok(D, q, a, nullptr, nullptr)) {
      launch_cosine_pair32<true, false>(q, a, nullptr, top, norm0, norm1, nullptr, nullptr, N, D, s);
    } else if (rows) {
      const unsigned grid = (N + 3) / 4;
      if (vec4_ok(D, q, a, nullptr, nullptr))
        hipLaunchKernelGGL((cosine_rows_kernel<true, true, false>), dim3(grid), dim3(256), 0, s,
                           q, a, nullptr, top, norm0, norm1, nullptr, nullptr, N, D);
      else
        hipLaunchKernelGGL((cosine_rows_kernel<false, true, false>), dim3(grid), dim3(256), 0, s,
                           q, a, nullptr, top, norm0, norm1, nullptr, nullptr, N, D);
    } else {
      const long long r0 = (long long)N * W1, r1 = (long long)N * W2;
      hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)((r0 + 3) / 4)), dim3(256), 0, s, q, norm0, r0, D, nullptr, 0);
      hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)((r1 + 3) / 4)), dim3(256), 0, s, a, norm1, r1, D, nullptr, 0);
      launch_cross_fwd<0>(q, a, norm0, norm1, top, N, W1, W2, D, s);
    }
  }
  return launch_status();
}

// top = SimCross(Embed(index_q), Embed(index_a)) for dist_mode 0 / 1; embed_bias = the Embed layers' bias blob
// (the driver's layers have one: `bias_term` stays at its default, do_trec_qa_clean.py:462-467) or null:
// embed_layer.cpp:135-152 followed by
// sim_cross_layer.cpp:96-139, with the gather done by SimCross's own loads.
int embed_simcross_forward(int mode, int N, int W1, int W2, int D, int K, const float* index_q,
                           const float* index_a, const float* weight, const float* embed_bias, float* top,
                           float* norm0, float* norm1, hipStream_t s) {
  if (N == 0) return MMS_OK;
  const CrossGather gt{index_q, index_a, K, embed_bias};
  if (mode == 1) {
    launch_cross_fwd<1>(weight, weight, nullptr, nullptr, top, N, W1, W2, D, s, gt);
  } else {
    const long long r0 = (long long)N * W1, r1 = (long long)N * W2;
    hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)((r0 + 3) / 4)), dim3(256), 0, s, weight, norm0, r0, D, index_q, K, embed_bias);
    hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)((r1 + 3) / 4)), dim3(256), 0, s, weight, norm1, r1, D, index_a, K, embed_bias);
    launch_cross_fwd<0>(weight, weight, norm0, norm1, top, N, W1, W2, D, s, gt);
  }
  return launch_status();
}

int simcross_elementwise_backward(int mode, int N, int W1, int W2, int D,
                                  const float* q, const float* a, const float* top,
                                  const float* top_diff, const float* norm0,
                                  const float* norm1, float* dq, float* da,
                                  hipStream_t s) {
  if (N == 0) return MMS_OK;
  const bool rows = (W1 == 1 && W2 == 1);
  if (rows && mode == 1) {
    if (wave_ok(D, q, a, dq, da)) {
      launch_rows_wave<false, true>(q, a, top, top_diff, nullptr, dq, da, N, D, s);
    } else {
      const int total = N * D;
      int blocks = (total + 255) / 256;
      if (blocks > 256 * 16) blocks = 256 * 16;
      hipLaunchKernelGGL(euclid_rows_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q, a,
                         top, top_diff, dq, da, total, D);
    }
  } else if (rows && mode == 0 && pair32_width(D) && vec4_ok(D, q, a, dq, da)) {
    launch_cosine_pair32<false, true>(q, a, top_diff, const_cast<float*>(top), const_cast<float*>(norm0),
                                      const_cast<float*>(norm1), dq, da, N, D, s);
  } else if (rows && mode == 0) {
    const unsigned grid = (N + 3) / 4;
    if (vec4_ok(D, q, a, dq, da))
      hipLaunchKernelGGL((cosine_rows_kernel<true, false, true>), dim3(grid), dim3(256), 0, s, q, a,
                         top_diff, const_cast<float*>(top), const_cast<float*>(norm0),
                         const_cast<float*>(norm1), dq, da, N, D);
    else
      hipLaunchKernelGGL((cosine_rows_kernel<false, false, true>), dim3(grid), dim3(256), 0, s, q, a,
                         top_diff, const_cast<float*>(top), const_cast<float*>(norm0),
                         const_cast<float*>(norm1), dq, da, N, D);
  } else if (mode == 1 && cross_bwd_lane_ok(euclid_backward_mode() == MMS_EUCLID_BWD_REFERENCE, N, W1, W2, D)) {
    const bool exact = euclid_backward_mode() == MMS_EUCLID_BWD_REFERENCE;
    const size_t lds = cross_bwd_lane_lds(exact, W1, W2);
#define MMS_LANE(W2_)                                                                                     \
  case W2_:                                                                                               \
    if (exact)                                                                                            \
      hipLaunchKernelGGL((cross_bwd_lane_kernel<W2_, true>), dim3((unsigned)N), dim3(64), lds, s, q, a,   \
                         top, top_diff, dq, da, W1, D);                                                   \
    else                                                                                                  \
      hipLaunchKernelGGL((cross_bwd_lane_kernel<W2_, false, 2>), dim3((unsigned)N), dim3(128), lds, s, q, \
                         a, top, top_diff, dq, da, W1, D);                                                \
    break;
    switch (W2) { MMS_LANE(8) MMS_LANE(16) MMS_LANE(20) MMS_LANE(24) MMS_LANE(32) MMS_LANE(40) MMS_LANE(48) }
#undef MMS_LANE
  } else if (cross_bwd_tiled_lds(mode, W1, W2) <= 64 * 1024 &&
             2LL * N * ((D + kBwdDC - 1) / kBwdDC) <= 0x7fffffffLL) {
    const int nchunks = (D + kBwdDC - 1) / kBwdDC;
    const size_t lds = cross_bwd_tiled_lds(mode, W1, W2);
    const int split = ((long long)N * nchunks < 1024) ? 1 : 0;     // fill the chip when the batch is small
    const unsigned grid = (unsigned)((split ? 2LL : 1LL) * N * nchunks);
    if (mode == 1 && euclid_backward_mode() == MMS_EUCLID_BWD_REFERENCE)
      hipLaunchKernelGGL((cross_bwd_tiled_kernel<1, true>), dim3(grid), dim3(256), lds, s, q, a, top,
                         top_diff, nullptr, nullptr, dq, da, W1, W2, D, nchunks, split);
    else if (mode == 1)
      hipLaunchKernelGGL((cross_bwd_tiled_kernel<1, false>), dim3(grid), dim3(256), lds, s, q, a, top,
                         top_diff, nullptr, nullptr, dq, da, W1, W2, D, nchunks, split);
    else
      hipLaunchKernelGGL((cross_bwd_tiled_kernel<0, true>), dim3(grid), dim3(256), lds, s, q, a, top,
                         top_diff, norm0, norm1, dq, da, W1, W2, D, nchunks, split);
  } else if (mode == 1) {
    hipLaunchKernelGGL((cross_bwd_kernel<1>), dim3(N), dim3(256), 0, s, q, a, top, top_diff,
                       nullptr, nullptr, dq, da, W1, W2, D);
  } else {
    hipLaunchKernelGGL((cross_bwd_kernel<0>), dim3(N), dim3(256), 0, s, q, a, top, top_diff,
                       norm0, norm1, dq, da, W1, W2, D);
  }
  return launch_status();
}

// Forward+backward in one launch where the geometry allows (rows); otherwise
// the two passes back to back.
int simcross_elementwise_forward_backward(int mode, int N, int W1, int W2, int D,
                                          const float* q, const float* a,
                                          const float* top_diff, float* top,
                                          float* norm0, float* norm1, float* dq,
                                          float* da, hipStream_t s) {
  if (N == 0) return MMS_OK;
  const bool rows = (W1 == 1 && W2 == 1);
  if (rows && mode == 1 && wave_ok(D, q, a, dq, da)) {
    launch_rows_wave<true, true>(q, a, nullptr, top_diff, top, dq, da, N, D, s);
    return launch_status();
  }
  if (rows && mode == 1 && rows_fit(D)) {
    const unsigned grid = (N + kRows - 1) / kRows;
    hipLaunchKernelGGL((euclid_rows_kernel<kRows, kRowsThreads, true>), dim3(grid),
                       dim3(kRowsThreads), rows_lds_bytes(D), s, q, a, top_diff, top, dq, da, N, D);
    return launch_status();
  }
  if (rows && mode == 0 && pair32_width(D) && vec4_ok(D, q, a, dq, da)) {
    launch_cosine_pair32<true, true>(q, a, top_diff, top, norm0, norm1, dq, da, N, D, s);
    return launch_status();
  }
  if (rows && mode == 0) {
    const unsigned grid = (N + 3) / 4;
    if (vec4_ok(D, q, a, dq, da))
      hipLaunchKernelGGL((cosine_rows_kernel<true, true, true>), dim3(grid), dim3(256), 0, s, q, a,
                         top_diff, top, norm0, norm1, dq, da, N, D);
    else
      hipLaunchKernelGGL((cosine_rows_kernel<false, true, true>), dim3(grid), dim3(256), 0, s, q, a,
                         top_diff, top, norm0, norm1, dq, da, N, D);
    return launch_status();
  }
  int rc = simcross_elementwise_forward(mode, N, W1, W2, D, q, a, top, norm0, norm1, s);
  if (rc != MMS_OK) return rc;
  return simcross_elementwise_backward(mode, N, W1, W2, D, q, a, top, top_diff, norm0, norm1,
                                       dq, da, s);
}

}  // namespace mms
