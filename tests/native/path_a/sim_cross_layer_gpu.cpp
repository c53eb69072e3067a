// Path A: replaces src/caffe/layers/sim_cross_layer.cu:128-243 in the reference's tree.
#include "caffe/layers/sim_cross_layer.hpp"
#include "mms.h"

namespace caffe {

// The bilinear mode needs scratch (U = dT.A, V = dT^T.Q, split-K slabs): measure_temp0_, which the reference's
// Reshape sizes for one (W, D) product (sim_cross_layer.cpp:70-79), is re-sized to what the library asks for.
static float* simcross_workspace(Blob<float>* scratch, int mode, int N, int W1, int W2, int D, int M, size_t* bytes) {
  *bytes = mms_simcross_workspace_bytes(mode, N, W1, W2, D, M);
  if (*bytes == 0) return NULL;
  const int elems = (int)((*bytes + sizeof(float) - 1) / sizeof(float));
  if (scratch->count() < elems) scratch->Reshape(vector<int>(1, elems));
  return scratch->mutable_gpu_data();
}

template <>
void SimCrossLayer<float>::Forward_gpu(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
  const int N = bottom[0]->num(), W1 = bottom[0]->channels(), W2 = bottom[1]->channels(), D = bottom[0]->height();
  const int M = top[0]->channels();
  const bool m0 = dist_mode_ == 0, m2 = dist_mode_ == 2;
  size_t ws_bytes;
  float* ws = simcross_workspace(&measure_temp0_, dist_mode_, N, W1, W2, D, M, &ws_bytes);
  const int rc = mms_simcross_forward_f32(
      dist_mode_, N, W1, W2, D, M, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
      m2 ? this->blobs_[0]->gpu_data() : NULL, (m2 && this->blobs_.size() > 1) ? this->blobs_[1]->gpu_data() : NULL,
      top[0]->mutable_gpu_data(), m0 ? data0_norm_.mutable_gpu_data() : NULL,
      m0 ? data1_norm_.mutable_gpu_data() : NULL, ws, ws_bytes, /*stream=*/NULL);   // Caffe runs on the null stream
  CHECK_EQ(rc, (int)MMS_OK) << mms_error_string(rc);
}

template <>
void SimCrossLayer<float>::Backward_gpu(const vector<Blob<float>*>& top, const vector<bool>& propagate_down,
                                        const vector<Blob<float>*>& bottom) {
  const int N = bottom[0]->num(), W1 = bottom[0]->channels(), W2 = bottom[1]->channels(), D = bottom[0]->height();
  const int M = top[0]->channels();
  const bool m0 = dist_mode_ == 0, m2 = dist_mode_ == 2;
  const bool bias_term = m2 && this->blobs_.size() > 1;
  size_t ws_bytes;
  float* ws = simcross_workspace(&measure_temp0_, dist_mode_, N, W1, W2, D, M, &ws_bytes);
  const int rc = mms_simcross_backward_f32(
      dist_mode_, N, W1, W2, D, M, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
      m2 ? this->blobs_[0]->gpu_data() : NULL, bias_term, top[0]->gpu_data(), top[0]->gpu_diff(),
      m0 ? data0_norm_.gpu_data() : NULL, m0 ? data1_norm_.gpu_data() : NULL,
      propagate_down[0], propagate_down[1], bottom[0]->mutable_gpu_diff(), bottom[1]->mutable_gpu_diff(),
      m2 ? this->blobs_[0]->mutable_gpu_diff() : NULL, bias_term ? this->blobs_[1]->mutable_gpu_diff() : NULL,
      ws, ws_bytes, /*stream=*/NULL);
  CHECK_EQ(rc, (int)MMS_OK) << mms_error_string(rc);
}

}  // namespace caffe
