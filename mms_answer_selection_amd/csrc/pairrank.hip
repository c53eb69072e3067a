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
#include "mms_common.h"

namespace mms {

struct PairTerm {
  float ordered, similar, term;
};

// :28-37 and the summand of :43-44, in the reference's operation order.
__device__ __forceinline__ PairTerm pair_term(float a, float b, float y, float margin) {
  PairTerm p;
  const float diff = a - b;          // caffe_sub
  p.similar = diff;                  // caffe_copy
  float o = diff * y;                // caffe_mul
  o = -1.0f * o + 0.0f * o;          // caffe_cpu_axpby(-1, x, 0, y = x) (MKL semantics)
  o = o + margin;                    // caffe_add_scalar
  p.ordered = o;
  const float hinge = (0.0f < o) ? o : 0.0f;  // std::max(Dtype(0), o)
  p.term = hinge + fabsf((1.0f - y) * diff);
  return p;
}

// :72-79 for one element; s0/s1 are the two `sign` values.
__device__ __forceinline__ void pair_grad(float y, float ordered, float similar, float s0,
                                          float s1, float& ga, float& gb) {
  const float ordered_t = ordered > 0.0f ? 1.0f : 0.0f;
  const float similar_t = (1.0f - y) * similar > 0.0f ? 1.0f : -1.0f;
  const float inner = ordered_t * y - similar_t * (1.0f - y);
  ga = s0 * inner;
  gb = s1 * inner;
}

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
  for (int i = threadIdx.x; i < n; i += kPairThreads) s += partials[i];
  s = block_sum<kPairThreads>(s, red);
  if (threadIdx.x == 0) *loss = s / (float)count;
}

__global__ __launch_bounds__(256) void pairrank_bwd_kernel(
    int count, float s0, float s1, const float* __restrict__ y,
    const float* __restrict__ ordered, const float* __restrict__ similar,
    float* __restrict__ da, float* __restrict__ db) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    float ga, gb;
    pair_grad(y[i], ordered[i], similar[i], s0, s1, ga, gb);
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

size_t pairrank_workspace_bytes(int count) {
  const int b = pair_blocks(count);
  return b > 1 ? (size_t)b * sizeof(float) : 0;
}

int pairrank_forward(int count, float margin, const float* a, const float* b, const float* y,
                     float* ordered, float* similar, float* loss, void* ws, size_t ws_bytes,
                     hipStream_t s) {
  const int blocks = pair_blocks(count);
  if (blocks > 1 && (ws == nullptr || ws_bytes < (size_t)blocks * sizeof(float)))
    return MMS_ERR_WORKSPACE;
  float* partials = static_cast<float*>(ws);
  hipLaunchKernelGGL(pairrank_fwd_kernel, dim3(blocks), dim3(kPairThreads), 0, s, count, margin,
                     a, b, y, ordered, similar, partials, loss);
  if (blocks > 1)
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
                     ordered, similar, da, db);
  return launch_status();
}

// ======================= fused (q, a+, a-) training step =====================
// Euclidean SimCross on (q,a+) and (q,a-), PairRankLoss on the two score
// columns, and the whole backward, in one launch (+ a one-block loss finish).
// A workgroup owns ROWS triplets: q-a+ and q-a- live in LDS from the first
// HBM read to the gradient write, so each input is read once and each
// gradient written once.  One lane per (triplet, branch) walks d ascending
// (reference order, sim_cross_layer.cpp:100-106).
__device__ __forceinline__ void triplet_coef(float T, float g, float& c, double& den) {
  c = g * T * T * T;                  // sim_cross_layer.cpp:216
  den = (double)(T - 1.0f) + 1e-9;    // :217
}

template <int ROWS, int THREADS, bool VEC4>
__global__ __launch_bounds__(THREADS) void triplet_euclid_kernel(
    int N, int D, float margin, float s0, float s1, const float* __restrict__ q,
    const float* __restrict__ ap, const float* __restrict__ an, const float* __restrict__ y,
    float* __restrict__ s_pos, float* __restrict__ s_neg, float* __restrict__ partials,
    float* __restrict__ dq, float* __restrict__ dap, float* __restrict__ dan) {
  extern __shared__ float4 lds_raw[];
  float* dpos = reinterpret_cast<float*>(lds_raw);   // [ROWS*D]
  float* dneg = dpos + (size_t)ROWS * D;             // [ROWS*D]
  __shared__ float Ts[2][ROWS];
  __shared__ float cs[2][ROWS];
  __shared__ double dens[2][ROWS];
  __shared__ float terms[ROWS];

  const int row0 = blockIdx.x * ROWS;
  const int rows = min(ROWS, N - row0);
  const size_t base = (size_t)row0 * D;
  const int total = rows * D;

  if (VEC4) {
    const float4* q4 = reinterpret_cast<const float4*>(q + base);
    const float4* p4 = reinterpret_cast<const float4*>(ap + base);
    const float4* n4 = reinterpret_cast<const float4*>(an + base);
    float4* dp4 = reinterpret_cast<float4*>(dpos);
    float4* dn4 = reinterpret_cast<float4*>(dneg);
    for (int i = threadIdx.x; i < (total >> 2); i += THREADS) {
      const float4 x = q4[i], u = p4[i], v = n4[i];
      float4 a, b;
      a.x = x.x - u.x; a.y = x.y - u.y; a.z = x.z - u.z; a.w = x.w - u.w;
      b.x = x.x - v.x; b.y = x.y - v.y; b.z = x.z - v.z; b.w = x.w - v.w;
      dp4[i] = a;
      dn4[i] = b;
    }
  } else {
    for (int i = threadIdx.x; i < total; i += THREADS) {
      const float x = q[base + i];
      dpos[i] = x - ap[base + i];
      dneg[i] = x - an[base + i];
    }
  }
  __syncthreads();

  // lanes [0,ROWS): positive branch; lanes [ROWS,2*ROWS): negative branch.
  if (threadIdx.x < 2 * ROWS) {
    const int br = threadIdx.x / ROWS, r = threadIdx.x % ROWS;
    if (r < rows) {
      const float* src = (br ? dneg : dpos) + r * D;
      float dist = 0.f;
      if (VEC4) {
        const float4* r4 = reinterpret_cast<const float4*>(src);
#pragma unroll 4
        for (int d = 0; d < (D >> 2); ++d) {
          const float4 v = r4[d];
          dist += v.x * v.x; dist += v.y * v.y; dist += v.z * v.z; dist += v.w * v.w;
        }
      } else {
        for (int d = 0; d < D; ++d) dist += src[d] * src[d];
      }
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
      pair_grad(yy, p.ordered, p.similar, s0, s1, ga, gb);
      float c; double den;
      triplet_coef(Ts[0][r], ga, c, den); cs[0][r] = c; dens[0][r] = den;
      triplet_coef(Ts[1][r], gb, c, den); cs[1][r] = c; dens[1][r] = den;
      t = p.term;
    }
    terms[r] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) s += terms[r];
    partials[blockIdx.x] = s;
  }

  // Gradients.  Layer-by-layer semantics: each SimCross backward produces
  // dq_branch = 0 + tt, da = 0 + (-tt); Caffe's Split layer then adds the two
  // dq_branch blobs (split_layer.cpp: caffe_add(bottom_diff = top0 + top1)).
  if (VEC4) {
    const int D4 = D >> 2;
    const float4* dp4 = reinterpret_cast<const float4*>(dpos);
    const float4* dn4 = reinterpret_cast<const float4*>(dneg);
    float4* dq4 = reinterpret_cast<float4*>(dq + base);
    float4* dap4 = reinterpret_cast<float4*>(dap + base);
    float4* dan4 = reinterpret_cast<float4*>(dan + base);
    for (int i = threadIdx.x; i < (total >> 2); i += THREADS) {
      const int r = i / D4;
      const float c0 = cs[0][r], c1 = cs[1][r];
      const double e0 = dens[0][r], e1 = dens[1][r];
      const float4 a = dp4[i], b = dn4[i];
      float tp[4], tn[4];
      tp[0] = (float)((double)(c0 * a.x) / e0); tp[1] = (float)((double)(c0 * a.y) / e0);
      tp[2] = (float)((double)(c0 * a.z) / e0); tp[3] = (float)((double)(c0 * a.w) / e0);
      tn[0] = (float)((double)(c1 * b.x) / e1); tn[1] = (float)((double)(c1 * b.y) / e1);
      tn[2] = (float)((double)(c1 * b.z) / e1); tn[3] = (float)((double)(c1 * b.w) / e1);
      float4 oq, op, on;
      oq.x = (0.f + tp[0]) + (0.f + tn[0]); oq.y = (0.f + tp[1]) + (0.f + tn[1]);
      oq.z = (0.f + tp[2]) + (0.f + tn[2]); oq.w = (0.f + tp[3]) + (0.f + tn[3]);
      op.x = 0.f + (-tp[0]); op.y = 0.f + (-tp[1]); op.z = 0.f + (-tp[2]); op.w = 0.f + (-tp[3]);
      on.x = 0.f + (-tn[0]); on.y = 0.f + (-tn[1]); on.z = 0.f + (-tn[2]); on.w = 0.f + (-tn[3]);
      dq4[i] = oq;
      dap4[i] = op;
      dan4[i] = on;
    }
  } else {
    for (int i = threadIdx.x; i < total; i += THREADS) {
      const int r = i / D;
      const float tp = (float)((double)(cs[0][r] * dpos[i]) / dens[0][r]);
      const float tn = (float)((double)(cs[1][r] * dneg[i]) / dens[1][r]);
      dq[base + i] = (0.f + tp) + (0.f + tn);
      dap[base + i] = 0.f + (-tp);
      dan[base + i] = 0.f + (-tn);
    }
  }
}

constexpr int kTripRows = 8;
constexpr int kTripThreads = 256;

size_t triplet_workspace_bytes(int N) {
  return (size_t)((N + kTripRows - 1) / kTripRows) * sizeof(float);
}

int triplet_euclid_step(int N, int D, float margin, float loss_weight, const float* q,
                        const float* ap, const float* an, const float* y, float* s_pos,
                        float* s_neg, float* loss, float* dq, float* dap, float* dan, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  if (N == 0) return MMS_OK;
  const size_t lds = 2 * (size_t)kTripRows * D * sizeof(float);
  if (lds > 96 * 1024) return MMS_ERR_UNSUPPORTED;
  if (ws == nullptr || ws_bytes < triplet_workspace_bytes(N)) return MMS_ERR_WORKSPACE;
  const int blocks = (N + kTripRows - 1) / kTripRows;
  const float scale = loss_weight / (float)N;  // pair_rank_loss_layer.cpp:64, count = N*1
  const float s0 = -1.0f * scale, s1 = 1.0f * scale;
  float* partials = static_cast<float*>(ws);
  const bool v = (D % 4 == 0) && aligned16(q) && aligned16(ap) && aligned16(an) &&
                 aligned16(dq) && aligned16(dap) && aligned16(dan);
  if (v)
    hipLaunchKernelGGL((triplet_euclid_kernel<kTripRows, kTripThreads, true>), dim3(blocks),
                       dim3(kTripThreads), lds, s, N, D, margin, s0, s1, q, ap, an, y, s_pos,
                       s_neg, partials, dq, dap, dan);
  else
    hipLaunchKernelGGL((triplet_euclid_kernel<kTripRows, kTripThreads, false>), dim3(blocks),
                       dim3(kTripThreads), lds, s, N, D, margin, s0, s1, q, ap, an, y, s_pos,
                       s_neg, partials, dq, dap, dan);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kPairThreads), 0, s, partials, blocks, N,
                     loss);
  return launch_status();
}

}  // namespace mms
