// csrc/pairrank.hip -- PairRankLoss forward/backward and the fused
// (q, a+, a-) training step for gfx950.
//
// Reference (src/caffe/layers/pair_rank_loss_layer.cpp):
//   forward  :26-52  diff = a-b ; similar = diff ; ordered = margin - y*diff
//                    (built as sub, mul, axpby(-1,0), add_scalar);
//                    loss = sum_i [max(0,ordered_i) + |(1-y_i)*similar_i|] / count
//   backward :55-84  sign = (i==0 ? -1 : +1) * top_diff/count ;
//                    diff_i = sign*(1[ordered>0]*y - ((1-y)*similar>0 ? 1 : -1)*(1-y))
// Per-element values are bit-identical to the CPU code.  The loss scalar is a
// fixed-shape tree sum (deterministic; the reference's 1-thread running sum
// is not reproduced -- tests hold it to 1e-5 relative).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "euclid_math.h"
#include "pairrank_math.h"
#include "mms_common.h"

namespace mms {

int euclid_backward_mode();   // simcross_elementwise.hip

// Which comparison gates the hinge term of the backward (include/mms.h: mms_set_pairrank_hinge_mode):
// the reference's Backward_cpu uses `ordered > 0` (pair_rank_loss_layer.cpp:76), its Backward_gpu kernel
// `ordered >= 0` (pair_rank_loss_layer.cu:51).  They differ only where margin - y*(a-b) is exactly 0.
// Per calling thread (a Caffe host runs one thread per GPU); default: the CPU code's strict `>`.
constexpr int kTicketTop = 1024;      // words 0..1023 of a ticket slot: one per group of kTicketGroup workgroups; then the top word
constexpr int kTicketStride = kTicketTop + 32;
constexpr int kTicketGroup = 16;
// An arrival word carries the arrivals AND what arrived, so that ONE atomic both hands over a partial loss
// and tells its issuer whether it was the last: [63] poison, [52..62] arrivals, [0..51] sum of the terms in
// units of 2^-S (S chosen by the host from N so that the field cannot overflow while every term < 2^kFxTermBits).
constexpr int kFxSumBits = 52;
constexpr int kFxTermBits = 10;
constexpr unsigned long long kFxPoison = 1ull << 63;
constexpr unsigned long long kFxOne = 1ull << kFxSumBits;
constexpr unsigned long long kFxSumMask = kFxOne - 1;

static thread_local int t_hinge_mode = MMS_PAIRRANK_HINGE_CPU;
int pairrank_hinge_mode() { return t_hinge_mode; }
void set_pairrank_hinge_mode(int m) { t_hinge_mode = m; }
static thread_local int t_loss_sum = MMS_LOSS_SUM_FAST;
int loss_sum_mode() { return t_loss_sum; }
void set_loss_sum_mode(int m) { t_loss_sum = m; }
static thread_local int t_triplet_finish = MMS_TRIPLET_FINISH_INLAUNCH;
int triplet_finish_mode() { return t_triplet_finish; }
void set_triplet_finish_mode(int m) { t_triplet_finish = m; }

constexpr int kPairThreads = 256;

// Each block reduces a grid-strided slice; block b writes partials[b], or the
// final loss when it is the only block.
__global__ __launch_bounds__(kPairThreads) void pairrank_fwd_kernel(
    int count, float margin, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ y, float* __restrict__ ordered, float* __restrict__ similar,
    float* __restrict__ partials, float* __restrict__ loss) {
  __shared__ float red[kPairThreads / 64];
  float s = 0.f;
  const int stride = gridDim.x * kPairThreads;
  for (int i = blockIdx.x * kPairThreads + threadIdx.x; i < count; i += stride) {
    const PairTerm p = pair_term(a[i], b[i], y[i], margin);
    ordered[i] = p.ordered;
    similar[i] = p.similar;
    s += p.term;
  }
  s = block_sum<kPairThreads>(s, red);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) *loss = s / (float)count;  // :49
    else partials[blockIdx.x] = s;
  }
}

// One block: sums `n` partials in a fixed order and writes sum/count.
__global__ __launch_bounds__(kPairThreads) void loss_finish_kernel(
    const float* __restrict__ partials, int n, int count, float* __restrict__ loss) {
  __shared__ float red[kPairThreads / 64];
  float s = 0.f;
  // eight independent loads in flight per thread (a one-block kernel is pure latency: a
  // load-add-load-add loop costs one memory round trip per element), added in index order
  for (int base = threadIdx.x; base < n; base += 8 * kPairThreads) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * kPairThreads;
      v[u] = partials[i < n ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (base + u * kPairThreads < n) ? v[u] : 0.f;
  }
  s = block_sum<kPairThreads>(s, red);
  if (threadIdx.x == 0) *loss = s / (float)count;
}

// MMS_LOSS_SUM_REFERENCE: the reference's own sum -- ONE running fp32 accumulator over the terms in index order
// (pair_rank_loss_layer.cpp:41-49) -- so that the loss scalar carries the CPU code's bits, drift and all.  A
// dependent chain of `count` adds by one lane (~3 ns each); the other threads only stage the next terms in LDS.
// terms != nullptr: the per-element terms as the fused step stored them; otherwise they are formed from the
// layer's own outputs, max(0, ordered) + |(1 - y) * similar| (:43-44).
constexpr int kRunChunk = 4096;
__global__ __launch_bounds__(256) void loss_running_sum_kernel(
    const float* __restrict__ terms, const float* __restrict__ ordered, const float* __restrict__ similar,
    const float* __restrict__ y, int count, float* __restrict__ loss) {
  __shared__ float buf[2][kRunChunk];
  auto stage = [&](int b, int base, int first, int nthreads) {   // threads [first, first + nthreads) fill buf[b]
    for (int e = (int)threadIdx.x - first; e < kRunChunk; e += nthreads) {
      const int i = base + e;
      float t = 0.f;
      if (i < count) {
        if (terms) t = terms[i];
        else {
          const float o = ordered[i];
          const float hinge = (0.0f < o) ? o : 0.0f;
          t = hinge + fabsf((1.0f - y[i]) * similar[i]);
        }
      }
      buf[b][e] = t;
    }
  };
  stage(0, 0, 0, 256);
  __syncthreads();
  float l = 0.f;
  for (int base = 0, b = 0; base < count; base += kRunChunk, b ^= 1) {
    if (threadIdx.x >= 64) {                       // the other waves fetch the next chunk meanwhile
      if (base + kRunChunk < count) stage(b ^ 1, base + kRunChunk, 64, 192);
    } else if (threadIdx.x == 0) {
      const int n = min(kRunChunk, count - base);
      int e = 0;
      for (; e + 8 <= n; e += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = buf[b][e + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) l += v[u];
      }
      for (; e < n; ++e) l += buf[b][e];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = l / (float)count;  // :49
}

__global__ __launch_bounds__(256) void pairrank_bwd_kernel(
    int count, float s0, float s1, const float* __restrict__ y,
    const float* __restrict__ ordered, const float* __restrict__ similar,
    float* __restrict__ da, float* __restrict__ db, int hinge_ge) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    float ga, gb;
    pair_grad(y[i], ordered[i], similar[i], s0, s1, ga, gb, hinge_ge != 0);
    if (da) da[i] = ga;
    if (db) db[i] = gb;
  }
}

static int pair_blocks(int count) {
  int b = (count + kPairThreads * 4 - 1) / (kPairThreads * 4);  // ~4 elements per thread
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return b;
}
// Batches of at most 8192 elements (the reference's use: one score column of 50 .. 4096 pairs) in ONE
// launch: a single 1024-thread workgroup, up to eight elements per thread with all 24 loads issued
// before the first use, fixed-order block sum.  Saves the separate loss-finish launch.
constexpr int kSmallThreads = 1024, kSmallMax = 8 * kSmallThreads;
__global__ __launch_bounds__(kSmallThreads) void pairrank_fwd_small_kernel(
    int count, float margin, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ y, float* __restrict__ ordered, float* __restrict__ similar,
    float* __restrict__ loss) {
  __shared__ float red[kSmallThreads / 64];
  float va[8], vb[8], vy[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = threadIdx.x + u * kSmallThreads;
    const int ii = i < count ? i : 0;
    va[u] = a[ii]; vb[u] = b[ii]; vy[u] = y[ii];
  }
  // all 24 values in registers before the first store: the stores sit under `i < count`, so the compiler cannot
  // count them and the wait for the NEXT value became vmcnt(0) -- the acknowledgement of the stores just issued
#pragma unroll
  for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(va[u]), "+v"(vb[u]), "+v"(vy[u]));
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = threadIdx.x + u * kSmallThreads;
    if (i < count) {
      const PairTerm p = pair_term(va[u], vb[u], vy[u], margin);
      ordered[i] = p.ordered;
      similar[i] = p.similar;
      s += p.term;
    }
  }
  s = block_sum<kSmallThreads>(s, red);
  if (threadIdx.x == 0) *loss = s / (float)count;   // :49
}

size_t pairrank_workspace_bytes(int count) {
  const int b = pair_blocks(count);
  return b > 1 ? (size_t)b * sizeof(float) : 0;
}

int pairrank_forward(int count, float margin, const float* a, const float* b, const float* y,
                     float* ordered, float* similar, float* loss, void* ws, size_t ws_bytes,
                     hipStream_t s) {
  if (count <= kSmallMax) {
    hipLaunchKernelGGL(pairrank_fwd_small_kernel, dim3(1), dim3(kSmallThreads), 0, s, count, margin, a,
                       b, y, ordered, similar, loss);
    if (loss_sum_mode() == MMS_LOSS_SUM_REFERENCE)
      hipLaunchKernelGGL(loss_running_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)nullptr, ordered,
                         similar, y, count, loss);
    return launch_status();
  }
  const int blocks = pair_blocks(count);
  if (blocks > 1 && (ws == nullptr || ws_bytes < (size_t)blocks * sizeof(float)))
    return MMS_ERR_WORKSPACE;
  float* partials = static_cast<float*>(ws);
  hipLaunchKernelGGL(pairrank_fwd_kernel, dim3(blocks), dim3(kPairThreads), 0, s, count, margin,
                     a, b, y, ordered, similar, partials, loss);
  if (loss_sum_mode() == MMS_LOSS_SUM_REFERENCE)
    hipLaunchKernelGGL(loss_running_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)nullptr, ordered,
                       similar, y, count, loss);
  else if (blocks > 1)
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kPairThreads), 0, s, partials, blocks,
                       count, loss);
  return launch_status();
}

int pairrank_backward(int count, float top_diff, const float* y, const float* ordered,
                      const float* similar, float* da, float* db, hipStream_t s) {
  if (count == 0 || (da == nullptr && db == nullptr)) return MMS_OK;
  // :64  sign *= top[0]->cpu_diff()[0] / bottom[0]->count()   (float / int -> float)
  const float scale = top_diff / (float)count;
  const float s0 = -1.0f * scale, s1 = 1.0f * scale;
  int blocks = (count + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pairrank_bwd_kernel, dim3(blocks), dim3(256), 0, s, count, s0, s1, y,
                     ordered, similar, da, db, pairrank_hinge_mode() == MMS_PAIRRANK_HINGE_GPU ? 1 : 0);
  return launch_status();
}

// ======================= fused (q, a+, a-) training step =====================
// Euclidean SimCross on (q,a+) and (q,a-), PairRankLoss on the two score
// columns, and the whole backward, in one launch (+ a one-block loss finish).
// Same wave-centric structure as euclid_rows_wave_kernel: a wave owns ONE
// triplet, issues all its 16-byte loads of q, a+, a- up front, keeps q-a+ and
// q-a- in registers; lanes 0-31 evaluate the positive branch's d-ascending sum
// and lanes 32-63 the negative branch's (speculative two-segment scheme of
// euclid_math.h when SPEC, plain walk by lanes 0 / 32 otherwise; both are the
// reference order, sim_cross_layer.cpp:100-106).  Each input is read once and
// each gradient written once.
template <int NIT, bool SPEC>
__global__ __launch_bounds__(256) void triplet_wave_kernel(
    int N, int D4, float margin, float s0, float s1, const float* __restrict__ q,
    const float* __restrict__ ap, const float* __restrict__ an, const float* __restrict__ y,
    float* __restrict__ s_pos, float* __restrict__ s_neg, float* __restrict__ partials,
    float* __restrict__ dq, float* __restrict__ dap, float* __restrict__ dan, int hinge_ge) {
  extern __shared__ float4 lds4[];               // [4 waves][2 branches] split images (euclid_math.h)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= N) return;
  const size_t base4 = (size_t)row * D4;
  const float4* q4 = reinterpret_cast<const float4*>(q) + base4;
  const float4* p4 = reinterpret_cast<const float4*>(ap) + base4;
  const float4* m4 = reinterpret_cast<const float4*>(an) + base4;
  const int st4 = spec_stride4(D4);
  float4* sqp = lds4 + (size_t)wave * 2 * st4;
  float4* sqn = sqp + st4;

  float4 x[NIT], u[NIT], v[NIT], dp[NIT], dn[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    const int ii = i < D4 ? i : 0;
    x[it] = q4[ii]; u[it] = p4[ii]; v[it] = m4[ii];
  }
  float yy = y[row];
  float predp1 = 0.f, predp2 = 0.f, predn1 = 0.f, predn2 = 0.f;
  const int h4 = spec_h4(D4);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    dp[it].x = x[it].x - u[it].x; dp[it].y = x[it].y - u[it].y;
    dp[it].z = x[it].z - u[it].z; dp[it].w = x[it].w - u[it].w;
    dn[it].x = x[it].x - v[it].x; dn[it].y = x[it].y - v[it].y;
    dn[it].z = x[it].z - v[it].z; dn[it].w = x[it].w - v[it].w;
    float4 a, b;
    a.x = dp[it].x * dp[it].x; a.y = dp[it].y * dp[it].y;
    a.z = dp[it].z * dp[it].z; a.w = dp[it].w * dp[it].w;
    b.x = dn[it].x * dn[it].x; b.y = dn[it].y * dn[it].y;
    b.z = dn[it].z * dn[it].z; b.w = dn[it].w * dn[it].w;
    if (i < D4) { sqp[i] = a; sqn[i] = b; }
    const float a4 = (i < D4) ? (a.x + a.y) + (a.z + a.w) : 0.f;
    const float b4 = (i < D4) ? (b.x + b.y) + (b.z + b.w) : 0.f;
    predp1 += (i < h4) ? a4 : 0.f; predp2 += (i < 2 * h4) ? a4 : 0.f;
    predn1 += (i < h4) ? b4 : 0.f; predn2 += (i < 2 * h4) ? b4 : 0.f;
  }
  {
    const int npad = st4 - D4;                      // zero pad at the end of both images
    if (lane < 2 * npad) sqp[(lane / npad) * st4 + D4 + (lane % npad)] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int br = lane >> 5, j = lane & 31;       // branch handled by this half-wave
  float dist;
  if (SPEC) {
    predp1 = wave_sum(predp1); predp2 = wave_sum(predp2);
    predn1 = wave_sum(predn1); predn2 = wave_sum(predn2);
    wave_lds_sync();
    dist = chain_sum_speculative<32>(br ? sqn : sqp, D4, br ? predn1 : predp1, br ? predn2 : predp2,
                                     j, br * 32);
  } else {
    wave_lds_sync();
    dist = 0.f;
    if (j == 0) {   // plain walk over the whole image (the pad adds +0: exact)
      dist = chain_sum_lds(br ? sqn : sqp, st4, 0.f);
    }
    dist = __shfl(dist, br * 32, 64);
  }
  const float Tmine = 1.0f / (1.0f + sqrtf(dist));
  const float Tp = __shfl(Tmine, 0, 64), Tn = __shfl(Tmine, 32, 64);
  asm volatile("" : "+v"(yy));   // in a register before the stores below (see triplet32_kernel)
  if (lane == 0) { s_pos[row] = Tp; s_neg[row] = Tn; }

  // PairRankLoss on (Tp, Tn, y): every lane computes the same scalars
  const PairTerm pt = pair_term(Tp, Tn, yy, margin);
  float ga, gb;
  pair_grad(yy, pt.ordered, pt.similar, s0, s1, ga, gb, hinge_ge != 0);
  if (lane == 0) partials[row] = pt.term;
  const EuclidCoef k0 = euclid_coef(Tp, ga), k1 = euclid_coef(Tn, gb);

  // Layer-by-layer semantics: each SimCross backward produces dq_branch = 0 + tt and
  // da = 0 + (-tt); Caffe's Split layer then adds the two dq_branch blobs.
  float4* dq4 = reinterpret_cast<float4*>(dq) + base4;
  float4* dp4 = reinterpret_cast<float4*>(dap) + base4;
  float4* dn4 = reinterpret_cast<float4*>(dan) + base4;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = lane + 64 * it;
    if (i >= D4) break;
    const float4 tp = euclid_tt4(k0, dp[it]), tn = euclid_tt4(k1, dn[it]);
    float4 oq, op, on;
    oq.x = (0.f + tp.x) + (0.f + tn.x); oq.y = (0.f + tp.y) + (0.f + tn.y);
    oq.z = (0.f + tp.z) + (0.f + tn.z); oq.w = (0.f + tp.w) + (0.f + tn.w);
    op.x = 0.f + (-tp.x); op.y = 0.f + (-tp.y); op.z = 0.f + (-tp.z); op.w = 0.f + (-tp.w);
    on.x = 0.f + (-tn.x); on.y = 0.f + (-tn.y); on.z = 0.f + (-tn.z); on.w = 0.f + (-tn.w);
    stream_store(dq4 + i, oq);
    stream_store(dp4 + i, op);
    stream_store(dn4 + i, on);
  }
}

// Width-specialised variant (D = 100 / 200 / 300), the triplet counterpart of
// euclid_pair32_kernel (simcross_elementwise.hip): D4C known at compile time, all 64 lanes
// hold float4s lane, lane+64 of q, a+ and a-; the window centres of the positive branch are
// reduced INTO lanes 0-31 and those of the negative branch into lanes 32-63 with one
// v_permlane32_swap + a half-wave DPP sum each; the chain is straight-line packed adds fed
// by LDS reads issued before the reductions; the stitch is the DPP OR-reduction; eight
// waves per workgroup, no early exit, N first for the kernarg preload, streaming stores.
// EXACT as in euclid_pair32_kernel (include/mms.h: mms_set_euclid_backward_mode).
template <int D4C, bool EXACT, int WPB, bool INL>
__global__ __launch_bounds__(64 * WPB) void triplet32_kernel(
    int N, float margin, float s0, float s1, const float* __restrict__ q,
    const float* __restrict__ ap, const float* __restrict__ an, const float* __restrict__ y,
    float* __restrict__ s_pos, float* __restrict__ s_neg, float* __restrict__ partials,
    float* __restrict__ dq, float* __restrict__ dap, float* __restrict__ dan, int hinge_ge,
    unsigned long long* __restrict__ ticket, float* __restrict__ loss, double fx_scale) {
  constexpr int NIT = (D4C + 63) / 64;
  constexpr int LASTN = D4C - 64 * (NIT - 1);
  constexpr int H4 = (D4C + 2) / 3, ST4 = 3 * H4;
  __shared__ float4 lds4[WPB * 2 * ST4];
  __shared__ unsigned long long wg_arrivals;       // INL: [60..63] waves arrived, [52..59] poisoned waves, [0..51] sum
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (INL) {
    // zeroed before anything is in flight: a raw barrier here waits for nothing but the eight wave starts
    if (threadIdx.x == 0) wg_arrivals = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const int want = blockIdx.x * WPB + wave;
  const bool have = want < N;
  const int row = have ? want : N - 1;           // a wave past the end recomputes the last triplet, stores nothing
  const size_t base4 = (size_t)row * D4C;
  const float4* q4 = reinterpret_cast<const float4*>(q) + base4;
  const float4* p4 = reinterpret_cast<const float4*>(ap) + base4;
  const float4* m4 = reinterpret_cast<const float4*>(an) + base4;
  const bool last_ok = (LASTN >= 64) || (lane < LASTN);
  float4* sqp = lds4 + (size_t)wave * 2 * ST4;
  float4* sqn = sqp + ST4;

  float yy = y[row];
  float4 x[NIT], u[NIT], v[NIT], dp[NIT], dn[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = (it < NIT - 1 || last_ok) ? lane + 64 * it : 0;
    x[it] = q4[i]; u[it] = p4[i]; v[it] = m4[i];
  }
  float pp1 = 0.f, pp2 = 0.f, pn1 = 0.f, pn2 = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const bool valid = (it < NIT - 1) || last_ok;
    const int i = lane + 64 * it;
    dp[it].x = x[it].x - u[it].x; dp[it].y = x[it].y - u[it].y;
    dp[it].z = x[it].z - u[it].z; dp[it].w = x[it].w - u[it].w;
    dn[it].x = x[it].x - v[it].x; dn[it].y = x[it].y - v[it].y;
    dn[it].z = x[it].z - v[it].z; dn[it].w = x[it].w - v[it].w;
    float4 a, b;
    a.x = dp[it].x * dp[it].x; a.y = dp[it].y * dp[it].y;
    a.z = dp[it].z * dp[it].z; a.w = dp[it].w * dp[it].w;
    b.x = dn[it].x * dn[it].x; b.y = dn[it].y * dn[it].y;
    b.z = dn[it].z * dn[it].z; b.w = dn[it].w * dn[it].w;
    if (valid) { sqp[i] = a; sqn[i] = b; }
    const float a4 = valid ? (a.x + a.y) + (a.z + a.w) : 0.f;
    const float b4 = valid ? (b.x + b.y) + (b.z + b.w) : 0.f;
    if (64 * it + 63 < H4) { pp1 += a4; pn1 += b4; }
    else if (64 * it < H4) { pp1 += (i < H4) ? a4 : 0.f; pn1 += (i < H4) ? b4 : 0.f; }
    if (64 * it + 63 < 2 * H4) { pp2 += a4; pn2 += b4; }
    else if (64 * it < 2 * H4) { pp2 += (i < 2 * H4) ? a4 : 0.f; pn2 += (i < 2 * H4) ? b4 : 0.f; }
  }
  if (ST4 > D4C && lane < 2 * (ST4 - D4C))
    sqp[(lane / (ST4 - D4C)) * ST4 + D4C + (lane % (ST4 - D4C))] = make_float4(0.f, 0.f, 0.f, 0.f);
  wave_lds_sync();
  const int br = lane >> 5, j = lane & 31;       // branch handled by this half-wave
  SpecSegment<H4> sg;
  sg.load((br ? sqn : sqp) + spec_seg32(j) * H4);
  // positive totals into lanes 0-31, negative totals into lanes 32-63
  const float p1 = half_wave_sum(swap_halves_add(pp1, pn1));
  const float p2 = half_wave_sum(swap_halves_add(pp2, pn2));
  __builtin_amdgcn_s_setprio(3);
  const float2v start = spec_start32(p1, p2, j);
  const float2v end = sg.chain(start);
  bool hit;
  float dist = spec_resolve_halves(start, end, j, &hit);
  if (!hit) {
    MMS_COUNT_MISS();
    dist = chain_sum_lds(br ? sqn : sqp, ST4, 0.0f);
  }
  __builtin_amdgcn_s_setprio(0);
  const float Tmine = 1.0f / (1.0f + sqrtf(dist));
  const float Tp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tmine), 0));
  const float Tn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tmine), 32));
  asm volatile("" : "+v"(yy));   // in a register before the stores below, or its wait becomes vmcnt(0) behind them (see euclid_pair32_kernel)
  if (lane == 0 && have) { s_pos[row] = Tp; s_neg[row] = Tn; }

  // PairRankLoss on (Tp, Tn, y): every lane computes the same scalars
  const PairTerm pt = pair_term(Tp, Tn, yy, margin);
  float ga, gb;
  pair_grad(yy, pt.ordered, pt.similar, s0, s1, ga, gb, hinge_ge != 0);
  // ---- loss scalar ------------------------------------------------------------------------------------------
  // INL: ONE launch.  The terms are added as integers (units of 2^-S), so the sum does not depend on the order
  // of arrival, and the arrival count travels in the same 64-bit word as the sum: an atomic's return value
  // tells its issuer both that it was the last and what the others brought, with no store whose visibility
  // would have to be waited for first.  Three hops: waves -> workgroup word in LDS -> one word per
  // kTicketGroup workgroups -> top word; the wave that completes the top word writes the loss.  All of it is
  // issued BEFORE this wave's gradient stores (one wave per workgroup waits one round trip for its group word,
  // one per group issues the top atomic and reads its return after its stores), so the round trips run under
  // the launch's store drain instead of behind it (the first in-launch form -- write-through term stores,
  // arrival tickets, then a 16 KB read of the terms by the last workgroup -- had four dependent round trips
  // behind the terms and measured 12.6 us against 11.3 for a second launch).
  // A term outside [0, 2^kFxTermBits) (labels or a margin in the hundreds, a NaN input) poisons the words it
  // passes through and the loss comes out NaN; the two-launch mode has no such limit (include/mms.h).
  bool top_wait = false;
  unsigned long long top_old = 0, top_pay = 0;
  if (INL) {
    if (lane == 0) {
      const float t = have ? pt.term : 0.f;
      const bool ok = t >= 0.f && t < (float)(1 << kFxTermBits);
      const unsigned long long fx = ok ? (unsigned long long)((double)t * fx_scale) : 0ull;
      const unsigned long long pay = (1ull << 60) | (ok ? 0ull : kFxOne) | fx;
      const unsigned long long old = __hip_atomic_fetch_add(&wg_arrivals, pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((old >> 60) == (unsigned long long)(WPB - 1)) {            // this wave completes the workgroup
        const unsigned long long wg = old + pay;
        const unsigned grp = blockIdx.x / kTicketGroup;
        const unsigned gsize = min((unsigned)kTicketGroup, gridDim.x - (unsigned)kTicketGroup * grp);
        const unsigned long long gpay = kFxOne | (wg & kFxSumMask);
        if ((wg >> kFxSumBits) & 0xffull)
          __hip_atomic_fetch_or(ticket + grp, kFxPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long gold = __hip_atomic_fetch_add(ticket + grp, gpay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((gold >> kFxSumBits) & 0x7ffull) == (unsigned long long)(gsize - 1)) {   // ... and its group
          const unsigned long long g = gold + gpay;
          __hip_atomic_store(ticket + grp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the slot's next launch
          top_pay = kFxOne | (g & kFxSumMask);
          if (g & kFxPoison) __hip_atomic_fetch_or(ticket + kTicketTop, kFxPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          top_old = __hip_atomic_fetch_add(ticket + kTicketTop, top_pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          top_wait = true;                                            // consumed after this wave's stores
        }
      }
    }
  } else {
    // the term leaves as a plain store: a second, one-workgroup launch sums all N of them
    if (lane == 0 && have) partials[row] = pt.term;
  }

  float4* dq4 = reinterpret_cast<float4*>(dq) + base4;
  float4* dp4 = reinterpret_cast<float4*>(dap) + base4;
  float4* dn4 = reinterpret_cast<float4*>(dan) + base4;
  float4 tp[NIT], tn[NIT];
  if (EXACT) {
    const EuclidCoef k0 = euclid_coef(Tp, ga), k1 = euclid_coef(Tn, gb);
#pragma unroll
    for (int it = 0; it < NIT; ++it) { tp[it] = euclid_tt4(k0, dp[it]); tn[it] = euclid_tt4(k1, dn[it]); }
  } else {
    const float c0 = ga * Tp * Tp * Tp, c1 = gb * Tn * Tn * Tn;
    const float r0 = (float)rcp_newton((double)(Tp - 1.0f) + 1e-9);
    const float r1 = (float)rcp_newton((double)(Tn - 1.0f) + 1e-9);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      tp[it].x = (c0 * dp[it].x) * r0; tp[it].y = (c0 * dp[it].y) * r0;
      tp[it].z = (c0 * dp[it].z) * r0; tp[it].w = (c0 * dp[it].w) * r0;
      tn[it].x = (c1 * dn[it].x) * r1; tn[it].y = (c1 * dn[it].y) * r1;
      tn[it].z = (c1 * dn[it].z) * r1; tn[it].w = (c1 * dn[it].w) * r1;
    }
  }
  if (have) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (!((it < NIT - 1) || last_ok)) break;
      const int i = lane + 64 * it;
      float4 oq, op, on;
      oq.x = (0.f + tp[it].x) + (0.f + tn[it].x); oq.y = (0.f + tp[it].y) + (0.f + tn[it].y);
      oq.z = (0.f + tp[it].z) + (0.f + tn[it].z); oq.w = (0.f + tp[it].w) + (0.f + tn[it].w);
      op.x = 0.f + (-tp[it].x); op.y = 0.f + (-tp[it].y); op.z = 0.f + (-tp[it].z); op.w = 0.f + (-tp[it].w);
      on.x = 0.f + (-tn[it].x); on.y = 0.f + (-tn[it].y); on.z = 0.f + (-tn[it].z); on.w = 0.f + (-tn[it].w);
      stream_store(dq4 + i, oq);
      stream_store(dp4 + i, op);
      stream_store(dn4 + i, on);
    }
  }

  if (!INL) return;
  if (top_wait) {                                  // lane 0 of one wave per kTicketGroup workgroups
    const unsigned ngrp = (gridDim.x + kTicketGroup - 1) / kTicketGroup;
    if (((top_old >> kFxSumBits) & 0x7ffull) == (unsigned long long)(ngrp - 1)) {
      const unsigned long long all = top_old + top_pay;
      __hip_atomic_store(ticket + kTicketTop, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float sum = (float)((double)(all & kFxSumMask) / fx_scale);
      *loss = (all & kFxPoison) ? __builtin_nanf("") : sum / (float)N;                       // pair_rank_loss_layer.cpp:49
    }
  }
}

// ---- the same step with TWO triplets per wave --------------------------------------------------------------
// A wave owns triplets 2w and 2w+1: their rows of q, a+ and a- are ONE dense run of 2*D4C float4 per array
// (lane l holds float4s l, l+64, l+128 of the run, whatever triplet they fall in), all requested up front.
// The two triplets then go through the chain phase one after the other -- pass 0, pass 1, each exactly the
// half-wave scheme of triplet32_kernel on images in LDS -- and a pass stores its own triplet's gradients as
// soon as its scores are known: pass 1's LDS round trip and packed-add chains run while pass 0's stores drain,
// instead of every wave of the launch chaining and then every wave storing.  Half as many waves to dispatch,
// and a CU has half as many chains in its LDS return path at a time.  Same arithmetic, same bits.
template <int D4C, bool EXACT, int WPB, bool INL>
__global__ __launch_bounds__(64 * WPB) void triplet32x2_kernel(
    int N, float margin, float s0, float s1, const float* __restrict__ q,
    const float* __restrict__ ap, const float* __restrict__ an, const float* __restrict__ y,
    float* __restrict__ s_pos, float* __restrict__ s_neg, float* __restrict__ partials,
    float* __restrict__ dq, float* __restrict__ dap, float* __restrict__ dan, int hinge_ge,
    unsigned long long* __restrict__ ticket, float* __restrict__ loss, double fx_scale) {
  constexpr int C = 2 * D4C;                       // float4 per array per wave
  constexpr int NIT = (C + 63) / 64;
  constexpr int PNIT = (D4C + 31) / 32, LASTN = D4C - 32 * (PNIT - 1);
  constexpr int H4 = (D4C + 2) / 3, ST4 = 3 * H4;
  __shared__ float4 lds4[WPB * 4 * ST4];           // per wave: (triplet 0, +), (0, -), (1, +), (1, -)
  __shared__ unsigned long long wg_arrivals;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (INL) {
    if (threadIdx.x == 0) wg_arrivals = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const long long r0 = ((long long)blockIdx.x * WPB + wave) * 2;
  const long long total4 = (long long)N * D4C;
  const long long b4 = r0 * D4C;
  const float4* q4 = reinterpret_cast<const float4*>(q);
  const float4* p4 = reinterpret_cast<const float4*>(ap);
  const float4* m4 = reinterpret_cast<const float4*>(an);
  float yy[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) yy[t] = y[min(r0 + t, (long long)N - 1)];
  float4 dp[NIT], dn[NIT];
  {
    float4 x[NIT], u[NIT], v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = lane + 64 * it;
      long long gi = b4 + ((NIT * 64 == C || i < C) ? i : 0);          // clamp: keep the load unconditional
      gi = gi < total4 ? gi : total4 - 1;                              // a run past the end reads the last float4
      x[it] = q4[gi]; u[it] = p4[gi]; v[it] = m4[gi];
    }
    float4* img = lds4 + (size_t)wave * 4 * ST4;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = lane + 64 * it;
      dp[it].x = x[it].x - u[it].x; dp[it].y = x[it].y - u[it].y;
      dp[it].z = x[it].z - u[it].z; dp[it].w = x[it].w - u[it].w;
      dn[it].x = x[it].x - v[it].x; dn[it].y = x[it].y - v[it].y;
      dn[it].z = x[it].z - v[it].z; dn[it].w = x[it].w - v[it].w;
      float4 a, b;
      a.x = dp[it].x * dp[it].x; a.y = dp[it].y * dp[it].y;
      a.z = dp[it].z * dp[it].z; a.w = dp[it].w * dp[it].w;
      b.x = dn[it].x * dn[it].x; b.y = dn[it].y * dn[it].y;
      b.z = dn[it].z * dn[it].z; b.w = dn[it].w * dn[it].w;
      if (NIT * 64 == C || i < C) {
        const int t = i >= D4C ? 1 : 0, c = i - t * D4C;
        img[(2 * t) * ST4 + c] = a;
        img[(2 * t + 1) * ST4 + c] = b;
      }
    }
    if constexpr (ST4 > D4C) {                                         // zero tail of each of the four images
      if (lane < 4 * (ST4 - D4C))
        img[(lane / (ST4 - D4C)) * ST4 + D4C + lane % (ST4 - D4C)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  asm volatile("" : "+v"(yy[0]), "+v"(yy[1]));    // in registers before any store (see euclid_pair32_kernel)
  wave_lds_sync();

  const int br = lane >> 5, j = lane & 31;         // branch walked by this half-wave
  float4* dq4 = reinterpret_cast<float4*>(dq);
  float4* dp4 = reinterpret_cast<float4*>(dap);
  float4* dn4 = reinterpret_cast<float4*>(dan);
  // In-launch loss: both passes first, then the arrival atomics, then ALL gradient stores, so that the atomics
  // enter the memory queues ahead of the wave's 7 KB of stores.  Measured (rocprofv3): the launch takes 9.6 us
  // with the in-launch sum against 7.6 us without, wherever the atomics are issued -- two DEPENDENT device-scope
  // atomic round trips (group word, then top word; they execute at the memory side of the fabric, not in an
  // XCD's L2) cost ~1.9 us, about what the second launch costs (1.9-2.3 us): 10.0 vs 10.2 us per step.
  // With the second launch a pass stores as soon as its scores are known.
  constexpr bool LATE = INL;
  unsigned long long fx_sum = 0, fx_bad = 0;
  bool top_wait = false;
  unsigned long long top_old = 0, top_pay = 0;
  EuclidCoef k0[2], k1[2];
  float c0[2] = {0.f, 0.f}, c1[2] = {0.f, 0.f}, rr0[2] = {0.f, 0.f}, rr1[2] = {0.f, 0.f};
  auto store_pass = [&](int t) {                   // gradients of triplet t's float4s (a slot can hold both triplets': masked)
    const bool have = r0 + t < N;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (64 * it >= (t + 1) * D4C || 64 * it + 63 < t * D4C) continue;   // no float4 of triplet t in this slot
      const int i = lane + 64 * it;
      const bool mine = have && i >= t * D4C && i < (t + 1) * D4C;
      float4 tp, tn;
      if (EXACT) { tp = euclid_tt4(k0[t], dp[it]); tn = euclid_tt4(k1[t], dn[it]); }
      else {
        tp.x = (c0[t] * dp[it].x) * rr0[t]; tp.y = (c0[t] * dp[it].y) * rr0[t];
        tp.z = (c0[t] * dp[it].z) * rr0[t]; tp.w = (c0[t] * dp[it].w) * rr0[t];
        tn.x = (c1[t] * dn[it].x) * rr1[t]; tn.y = (c1[t] * dn[it].y) * rr1[t];
        tn.z = (c1[t] * dn[it].z) * rr1[t]; tn.w = (c1[t] * dn[it].w) * rr1[t];
      }
      if (mine) {
        float4 oq, op, on;
        oq.x = (0.f + tp.x) + (0.f + tn.x); oq.y = (0.f + tp.y) + (0.f + tn.y);
        oq.z = (0.f + tp.z) + (0.f + tn.z); oq.w = (0.f + tp.w) + (0.f + tn.w);
        op.x = 0.f + (-tp.x); op.y = 0.f + (-tp.y); op.z = 0.f + (-tp.z); op.w = 0.f + (-tp.w);
        on.x = 0.f + (-tn.x); on.y = 0.f + (-tn.y); on.z = 0.f + (-tn.z); on.w = 0.f + (-tn.w);
        stream_store(dq4 + b4 + i, oq);
        stream_store(dp4 + b4 + i, op);
        stream_store(dn4 + b4 + i, on);
      }
    }
  };
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const bool have = r0 + t < N;
    const long long row = have ? r0 + t : (long long)N - 1;
    const float4* im = lds4 + ((size_t)wave * 4 + 2 * t + br) * ST4;
    SpecSegment<H4> sg;
    sg.load(im + spec_seg32(j) * H4);
    // window centres: tree sums of segment 0 / segments 0-1, read back from the image (as euclid_block_kernel)
    const bool last_ok = (LASTN >= 32) || (j < LASTN);
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int it = 0; it < PNIT; ++it) {
      const bool valid = (it < PNIT - 1) || last_ok;
      const float4 sq = im[valid ? j + 32 * it : 0];
      const float s4 = valid ? (sq.x + sq.y) + (sq.z + sq.w) : 0.f;
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
    if (!hit) {
      MMS_COUNT_MISS();
      dist = chain_sum_lds(im, ST4, 0.0f);
    }
    __builtin_amdgcn_s_setprio(0);
    const float Tmine = 1.0f / (1.0f + sqrtf(dist));
    const float Tp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tmine), 0));
    const float Tn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tmine), 32));
    const PairTerm pt = pair_term(Tp, Tn, yy[t], margin);
    float ga, gb;
    pair_grad(yy[t], pt.ordered, pt.similar, s0, s1, ga, gb, hinge_ge != 0);
    if (EXACT) { k0[t] = euclid_coef(Tp, ga); k1[t] = euclid_coef(Tn, gb); }
    else {
      c0[t] = ga * Tp * Tp * Tp; c1[t] = gb * Tn * Tn * Tn;
      rr0[t] = (float)rcp_newton((double)(Tp - 1.0f) + 1e-9);
      rr1[t] = (float)rcp_newton((double)(Tn - 1.0f) + 1e-9);
    }
    if (INL) {
      const float tm = have ? pt.term : 0.f;
      const bool ok = tm >= 0.f && tm < (float)(1 << kFxTermBits);
      fx_sum += ok ? (unsigned long long)((double)tm * fx_scale) : 0ull;
      fx_bad += ok ? 0ull : 1ull;
      if (t == 1 && lane == 0) {                   // the wave's two terms arrive together (see triplet32_kernel)
        const unsigned long long pay = (1ull << 60) | (fx_bad ? kFxOne : 0ull) | fx_sum;
        const unsigned long long old = __hip_atomic_fetch_add(&wg_arrivals, pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((old >> 60) == (unsigned long long)(WPB - 1)) {
          const unsigned long long wg = old + pay;
          const unsigned grp = blockIdx.x / kTicketGroup;
          const unsigned gsize = min((unsigned)kTicketGroup, gridDim.x - (unsigned)kTicketGroup * grp);
          const unsigned long long gpay = kFxOne | (wg & kFxSumMask);
          if ((wg >> kFxSumBits) & 0xffull)
            __hip_atomic_fetch_or(ticket + grp, kFxPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long gold = __hip_atomic_fetch_add(ticket + grp, gpay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (((gold >> kFxSumBits) & 0x7ffull) == (unsigned long long)(gsize - 1)) {
            const unsigned long long g = gold + gpay;
            __hip_atomic_store(ticket + grp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            top_pay = kFxOne | (g & kFxSumMask);
            if (g & kFxPoison) __hip_atomic_fetch_or(ticket + kTicketTop, kFxPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            top_old = __hip_atomic_fetch_add(ticket + kTicketTop, top_pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            top_wait = true;
          }
        }
      }
    } else {
      if (lane == 0 && have) partials[row] = pt.term;
    }
    if (lane == 0 && have) { s_pos[row] = Tp; s_neg[row] = Tn; }
    if (!LATE) store_pass(t);
  }
  if (LATE) { store_pass(0); store_pass(1); }
  if (!INL) return;
  if (top_wait) {
    const unsigned ngrp = (gridDim.x + kTicketGroup - 1) / kTicketGroup;
    if (((top_old >> kFxSumBits) & 0x7ffull) == (unsigned long long)(ngrp - 1)) {
      const unsigned long long all = top_old + top_pay;
      __hip_atomic_store(ticket + kTicketTop, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float sum = (float)((double)(all & kFxSumMask) / fx_scale);
      *loss = (all & kFxPoison) ? __builtin_nanf("") : sum / (float)N;                       // pair_rank_loss_layer.cpp:49
    }
  }
}

// Generic fallback (any D / alignment): a workgroup owns ROWS triplets.
template <int ROWS, int THREADS>
__global__ __launch_bounds__(THREADS) void triplet_generic_kernel(
    int N, int D, float margin, float s0, float s1, const float* __restrict__ q,
    const float* __restrict__ ap, const float* __restrict__ an, const float* __restrict__ y,
    float* __restrict__ s_pos, float* __restrict__ s_neg, float* __restrict__ partials,
    float* __restrict__ dq, float* __restrict__ dap, float* __restrict__ dan, int hinge_ge) {
  extern __shared__ float4 lds_raw[];
  float* dpos = reinterpret_cast<float*>(lds_raw);   // [ROWS*D]
  float* dneg = dpos + (size_t)ROWS * D;             // [ROWS*D]
  __shared__ float Ts[2][ROWS];
  __shared__ float cs[2][ROWS];
  __shared__ double dens[2][ROWS];
  __shared__ float terms[ROWS];

  const int row0 = blockIdx.x * ROWS;
  const int rows = min(ROWS, N - row0);
  if (rows <= 0) return;                             // uniform per workgroup: no barrier is skipped by part of one
  const size_t base = (size_t)row0 * D;
  const int total = rows * D;
  for (int i = threadIdx.x; i < total; i += THREADS) {
    const float x = q[base + i];
    dpos[i] = x - ap[base + i];
    dneg[i] = x - an[base + i];
  }
  __syncthreads();
  if (threadIdx.x < 2 * ROWS) {
    const int br = threadIdx.x / ROWS, r = threadIdx.x % ROWS;
    if (r < rows) {
      const float* src = (br ? dneg : dpos) + r * D;
      float dist = 0.f;
      for (int d = 0; d < D; ++d) dist += src[d] * src[d];
      const float T = 1.0f / (1.0f + sqrtf(dist));
      Ts[br][r] = T;
      (br ? s_neg : s_pos)[row0 + r] = T;
    }
  }
  __syncthreads();
  if (threadIdx.x < ROWS) {
    const int r = threadIdx.x;
    float t = 0.f;
    if (r < rows) {
      const float yy = y[row0 + r];
      const PairTerm p = pair_term(Ts[0][r], Ts[1][r], yy, margin);
      float ga, gb;
      pair_grad(yy, p.ordered, p.similar, s0, s1, ga, gb, hinge_ge != 0);
      const EuclidCoef k0 = euclid_coef(Ts[0][r], ga), k1 = euclid_coef(Ts[1][r], gb);
      cs[0][r] = k0.c; dens[0][r] = k0.den;
      cs[1][r] = k1.c; dens[1][r] = k1.den;
      t = p.term;
    }
    terms[r] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < rows) partials[row0 + threadIdx.x] = terms[threadIdx.x];   // one term per triplet, like the wave kernels
  for (int i = threadIdx.x; i < total; i += THREADS) {
    const int r = i / D;
    const float tp = euclid_tt_exact(cs[0][r], dens[0][r], dpos[i]);
    const float tn = euclid_tt_exact(cs[1][r], dens[1][r], dneg[i]);
    dq[base + i] = (0.f + tp) + (0.f + tn);
    dap[base + i] = 0.f + (-tp);
    dan[base + i] = 0.f + (-tn);
  }
}

constexpr int kTripRows = 8;
constexpr int kTripThreads = 256;

// Arrival words of the in-launch loss reduction live at the HEAD of the caller's triplet workspace (kTicketStride
// 64-bit words, then one float per triplet).  They are zero whenever no launch is using the workspace:
// mms_triplet_workspace_init zeroes them once (and again after a launch that died mid-way), and the wave that
// completes a word resets it.  A launch -- eager or as a node of a captured graph -- therefore owns the words of the
// workspace it was given: the ABI's "one workspace per call in flight" rule covers them, and nothing about them is
// chosen at call time by host state.
constexpr size_t kTicketBytes = (size_t)kTicketStride * sizeof(unsigned long long);

size_t triplet_workspace_bytes(int N) { return kTicketBytes + (size_t)N * sizeof(float); }

int triplet_workspace_init(void* ws, size_t ws_bytes, hipStream_t s) {
  if (ws == nullptr || ws_bytes < kTicketBytes || (reinterpret_cast<uintptr_t>(ws) & 7u)) return MMS_ERR_WORKSPACE;
  return hipMemsetAsync(ws, 0, kTicketBytes, s) == hipSuccess ? MMS_OK : MMS_ERR_LAUNCH;
}

int triplet_euclid_step(int N, int D, float margin, float loss_weight, const float* q,
                        const float* ap, const float* an, const float* y, float* s_pos,
                        float* s_neg, float* loss, float* dq, float* dap, float* dan, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  if (N == 0) return MMS_OK;
  if (ws == nullptr || ws_bytes < triplet_workspace_bytes(N)) return MMS_ERR_WORKSPACE;
  const float scale = loss_weight / (float)N;  // pair_rank_loss_layer.cpp:64, count = N*1
  const float s0 = -1.0f * scale, s1 = 1.0f * scale;
  if (reinterpret_cast<uintptr_t>(ws) & 7u) return MMS_ERR_WORKSPACE;
  unsigned long long* const tickets = static_cast<unsigned long long*>(ws);
  float* partials = reinterpret_cast<float*>(static_cast<char*>(ws) + kTicketBytes);
  const int hge = pairrank_hinge_mode() == MMS_PAIRRANK_HINGE_GPU ? 1 : 0;
  const bool v = (D % 4 == 0) && aligned16(q) && aligned16(ap) && aligned16(an) &&
                 aligned16(dq) && aligned16(dap) && aligned16(dan);
  int nparts;
  if (v && (D == 300 || D == 200 || D == 100)) {
    constexpr int WPB = 8;
    nparts = N;
    // dev switch for A/B timing (tools/triplet_probe.py): MMS_TRIPLET_TPW = 1 | 2 triplets per wave
    static const int tpw = [] { const char* e = std::getenv("MMS_TRIPLET_TPW"); return e && std::atoi(e) == 1 ? 1 : 2; }();
    const unsigned grid = (unsigned)((N + WPB * tpw - 1) / (WPB * tpw));
    const bool exact = euclid_backward_mode() == MMS_EUCLID_BWD_REFERENCE;
#define MMS_T32_GO(d4, ex, inl)                                                                        \
  do {                                                                                                     \
    if (tpw == 2)                                                                                          \
      hipLaunchKernelGGL((triplet32x2_kernel<d4, ex, WPB, inl>), dim3(grid), dim3(64 * WPB), 0, s, N,      \
                         margin, s0, s1, q, ap, an, y, s_pos, s_neg, partials, dq, dap, dan, hge, tk,      \
                         loss, fx_scale);                                                                  \
    else                                                                                                   \
      hipLaunchKernelGGL((triplet32_kernel<d4, ex, WPB, inl>), dim3(grid), dim3(64 * WPB), 0, s, N,        \
                         margin, s0, s1, q, ap, an, y, s_pos, s_neg, partials, dq, dap, dan, hge, tk,      \
                         loss, fx_scale);                                                                  \
  } while (0)
#define MMS_T32(d4)                                                       \
  case 4 * d4:                                                            \
    if (exact) { if (tk) MMS_T32_GO(d4, true, true); else MMS_T32_GO(d4, true, false); }     \
    else       { if (tk) MMS_T32_GO(d4, false, true); else MMS_T32_GO(d4, false, false); }   \
    break;
    // MMS_TRIPLET_FINISH_INLAUNCH: the loss is summed inside the launch (integer terms, arrival words that carry
    // the sum: see the kernel); otherwise, and for batches beyond what a ticket slot covers, the per-triplet
    // terms are summed by a second, one-workgroup launch.
    const unsigned ngrp = (grid + kTicketGroup - 1) / kTicketGroup;
    unsigned long long* tk = (triplet_finish_mode() != MMS_TRIPLET_FINISH_INLAUNCH || ngrp > (unsigned)kTicketTop ||
                              loss_sum_mode() == MMS_LOSS_SUM_REFERENCE || loss == nullptr)
                                 ? nullptr : tickets;
    int lg = 0;
    while (((long long)1 << lg) < (long long)N) ++lg;
    const double fx_scale = std::ldexp(1.0, kFxSumBits - kFxTermBits - lg);
    switch (D) { MMS_T32(25) MMS_T32(50) MMS_T32(75) }
#undef MMS_T32
#undef MMS_T32_GO
    if (tk) return launch_status();               // the loss was reduced inside the launch
  } else if (v && D <= 1024) {
    const int D4 = D / 4;
    const int nit = (D4 + 63) / 64;
    nparts = N;
    const unsigned grid = (unsigned)((N + 3) / 4);
    const size_t lds = (size_t)4 * 2 * 3 * ((D4 + 2) / 3) * sizeof(float4);
    const bool spec = D <= 400;   // +-15 ulp window: see euclid_math.h
#define MMS_NIT_CASE(n)                                                                         \
  case n:                                                                                       \
    if (spec)                                                                                   \
      hipLaunchKernelGGL((triplet_wave_kernel<n, true>), dim3(grid), dim3(256), lds, s, N, D4,  \
                         margin, s0, s1, q, ap, an, y, s_pos, s_neg, partials, dq, dap, dan, hge);   \
    else                                                                                        \
      hipLaunchKernelGGL((triplet_wave_kernel<n, false>), dim3(grid), dim3(256), lds, s, N, D4, \
                         margin, s0, s1, q, ap, an, y, s_pos, s_neg, partials, dq, dap, dan, hge);   \
    break;
    switch (nit) { MMS_NIT_CASE(1) MMS_NIT_CASE(2) MMS_NIT_CASE(3) MMS_NIT_CASE(4) }
#undef MMS_NIT_CASE
  } else {
    const size_t lds = 2 * (size_t)kTripRows * D * sizeof(float);
    if (lds > 96 * 1024) return MMS_ERR_UNSUPPORTED;
    nparts = N;                                    // one term per triplet; a workgroup owns kTripRows triplets
    hipLaunchKernelGGL((triplet_generic_kernel<kTripRows, kTripThreads>), dim3((unsigned)((N + kTripRows - 1) / kTripRows)),
                       dim3(kTripThreads), lds, s, N, D, margin, s0, s1, q, ap, an, y, s_pos,
                       s_neg, partials, dq, dap, dan, hge);
  }
  if (loss == nullptr) return launch_status();     // the caller does not want the scalar: no reduction at all
  if (loss_sum_mode() == MMS_LOSS_SUM_REFERENCE)   // nparts == N on every path: one term per triplet
    hipLaunchKernelGGL(loss_running_sum_kernel, dim3(1), dim3(256), 0, s, partials, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, N, loss);
  else
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kPairThreads), 0, s, partials, nparts, N,
                       loss);
  return launch_status();
}

// loss = (sum of the N per-triplet terms) / N from a device array of terms: the tail of both fused steps
// (pair_rank_loss_layer.cpp:41-49).  MMS_LOSS_SUM_REFERENCE: the reference's running fp32 sum, else the fixed tree.
int triplet_loss_from_terms(const float* terms, int N, float* loss, hipStream_t s) {
  if (loss_sum_mode() == MMS_LOSS_SUM_REFERENCE)
    hipLaunchKernelGGL(loss_running_sum_kernel, dim3(1), dim3(256), 0, s, terms, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, N, loss);
  else
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kPairThreads), 0, s, terms, N, N, loss);
  return launch_status();
}

}  // namespace mms
