// Path A: replaces src/caffe/layers/sim_matrix_layer.cu:21-46 in the reference's tree.
// One member is added to include/caffe/layers/sim_matrix_layer.hpp: `Blob<Dtype> workspace_;`.
#include "caffe/layers/sim_matrix_layer.hpp"
#include "mms.h"

namespace caffe {

template <>
void SimMatrixLayer<float>::Forward_gpu(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
  // Q*W lands in bottom[1]'s diff, where the reference's forward puts it (sim_matrix_layer.cpp:58).
  // With the layer's workspace (mms_simmatrix_workspace_bytes, the backward's) the product runs on the bf16 matrix
  // pipe at fp32 accuracy for >= 2048 rows (include/mms.h); mms_simmatrix_forward_f32 (no workspace) stays on fp32 MFMA.
  const size_t ws_bytes = mms_simmatrix_workspace_bytes(M_, K1_, K2_);
  const int elems = (int)((ws_bytes + sizeof(float) - 1) / sizeof(float));
  if (workspace_.count() < elems) workspace_.Reshape(vector<int>(1, elems));
  const int rc = mms_simmatrix_forward_ws_f32(M_, K1_, K2_, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
                                              this->blobs_[0]->gpu_data(), top[0]->mutable_gpu_data(),
                                              bottom[1]->mutable_gpu_diff(), workspace_.mutable_gpu_data(), ws_bytes,
                                              /*stream=*/NULL);
  CHECK_EQ(rc, (int)MMS_OK) << mms_error_string(rc);
}

template <>
void SimMatrixLayer<float>::Backward_gpu(const vector<Blob<float>*>& top, const vector<bool>& propagate_down,
                                         const vector<Blob<float>*>& bottom) {
  const size_t ws_bytes = mms_simmatrix_workspace_bytes(M_, K1_, K2_);
  const int elems = (int)((ws_bytes + sizeof(float) - 1) / sizeof(float));
  if (workspace_.count() < elems) workspace_.Reshape(vector<int>(1, elems));
  // recomputes W^T q_j like the reference (:88); a host that knows bottom[1]'s diff still holds the forward's Q*W
  // calls mms_simmatrix_backward_cached_f32(..., qw = bottom[1]->gpu_diff(), ...) instead: one GEMM fewer
  const int rc = mms_simmatrix_backward_f32(
      M_, K1_, K2_, bottom[0]->gpu_data(), bottom[1]->gpu_data(), this->blobs_[0]->gpu_data(), top[0]->gpu_diff(),
      this->param_propagate_down_[0], propagate_down[0], propagate_down[1],
      propagate_down[0] ? bottom[0]->mutable_gpu_diff() : NULL, propagate_down[1] ? bottom[1]->mutable_gpu_diff() : NULL,
      this->param_propagate_down_[0] ? this->blobs_[0]->mutable_gpu_diff() : NULL,
      elems ? workspace_.mutable_gpu_data() : NULL, ws_bytes, /*stream=*/NULL);
  CHECK_EQ(rc, (int)MMS_OK) << mms_error_string(rc);
}

}  // namespace caffe
