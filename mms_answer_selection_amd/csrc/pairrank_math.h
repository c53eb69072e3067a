// csrc/pairrank_math.h -- PairRankLoss for ONE element, in the reference's operation order
// (src/caffe/layers/pair_rank_loss_layer.cpp:28-37, 43-44, 72-79).  Shared by pairrank.hip (the layer's kernels and
// the fused Euclidean triplet step) and by the panel GEMM's epilogue (the fused learned-metric triplet step).
#ifndef MMS_PAIRRANK_MATH_H_
#define MMS_PAIRRANK_MATH_H_

#include <hip/hip_runtime.h>

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
                                          float s1, float& ga, float& gb, bool ge = false) {
  const float ordered_t = (ge ? ordered >= 0.0f : ordered > 0.0f) ? 1.0f : 0.0f;
  const float similar_t = (1.0f - y) * similar > 0.0f ? 1.0f : -1.0f;
  const float inner = ordered_t * y - similar_t * (1.0f - y);
  ga = s0 * inner;
  gb = s1 * inner;
}

}  // namespace mms
#endif  // MMS_PAIRRANK_MATH_H_
