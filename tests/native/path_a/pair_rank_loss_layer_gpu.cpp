// Path A: replaces src/caffe/layers/pair_rank_loss_layer.cu:10-83 in the reference's tree.
// One member is added to include/caffe/layers/pair_rank_loss_layer.hpp: `Blob<Dtype> workspace_;`.
#include "caffe/layers/pair_rank_loss_layer.hpp"
#include "mms.h"

namespace caffe {

template <>
void PairRankLossLayer<float>::Forward_gpu(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
  const int count = bottom[0]->count();
  const size_t ws_bytes = mms_pairrank_workspace_bytes(count);
  const int elems = (int)((ws_bytes + sizeof(float) - 1) / sizeof(float));
  if (workspace_.count() < elems) workspace_.Reshape(vector<int>(1, elems));
  const int rc = mms_pairrank_forward_f32(count, margin_, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
                                          bottom[2]->gpu_data(), ordered_diff_.mutable_gpu_data(),
                                          similar_diff_.mutable_gpu_data(), top[0]->mutable_gpu_data(),
                                          elems ? workspace_.mutable_gpu_data() : NULL, ws_bytes, /*stream=*/NULL);
  CHECK_EQ(rc, (int)MMS_OK) << mms_error_string(rc);
}

template <>
void PairRankLossLayer<float>::Backward_gpu(const vector<Blob<float>*>& top, const vector<bool>& propagate_down,
                                            const vector<Blob<float>*>& bottom) {
  if (propagate_down[2]) LOG(FATAL) << this->type() << " Layer cannot backpropagate to label inputs.";   // :58-61
  // the loss weight: the reference reads it on the host too (pair_rank_loss_layer.cu:62 / .cpp:64)
  const float loss_weight = top[0]->cpu_diff()[0];
  const int rc = mms_pairrank_backward_f32(bottom[0]->count(), loss_weight, bottom[2]->gpu_data(),
                                           ordered_diff_.gpu_data(), similar_diff_.gpu_data(), propagate_down[0],
                                           propagate_down[1], bottom[0]->mutable_gpu_diff(),
                                           bottom[1]->mutable_gpu_diff(), /*stream=*/NULL);
  CHECK_EQ(rc, (int)MMS_OK) << mms_error_string(rc);
}

}  // namespace caffe
