// tests/native/path_a/reference_layer_decls.hpp -- TEST SCAFFOLD, not product code.
// The class declarations the three replacement translation units of INTEGRATION.md (path A) are compiled
// against: the layer classes as the reference declares them -- same names, same protected members
// (include/caffe/layers/sim_cross_layer.hpp:36-43, sim_matrix_layer.hpp:36-38, pair_rank_loss_layer.hpp:58-60,
// loss_layer.hpp:22-49) -- on top of this repository's mirror of Layer / Blob (csrc/caffe_api.hpp), plus the ONE
// member path A adds to two of them (marked ADDED).  In the reference's tree the replacement files include the
// reference's own headers instead; here `caffe/layers/*.hpp` next to this file forward to it.
#ifndef MMS_TESTS_PATH_A_DECLS_HPP_
#define MMS_TESTS_PATH_A_DECLS_HPP_
#include "caffe_api.hpp"

namespace caffe {

template <typename Dtype>
class SimCrossLayer : public Layer<Dtype> {
 public:
  explicit SimCrossLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  const char* type() const override { return "SimCross"; }
 protected:
  void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override;
  void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override;
  int dist_mode_;
  Blob<Dtype> data0_norm_;
  Blob<Dtype> data1_norm_;
  Blob<Dtype> measure_temp0_;
  Blob<Dtype> measure_temp1_;
};

template <typename Dtype>
class SimMatrixLayer : public Layer<Dtype> {
 public:
  explicit SimMatrixLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  const char* type() const override { return "SimMatrix"; }
 protected:
  void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override;
  void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override;
  int K1_;
  int K2_;
  int M_;
  Blob<Dtype> workspace_;   // ADDED by path A (split-K slabs, W^T)
};

template <typename Dtype>
class LossLayer : public Layer<Dtype> {
 public:
  explicit LossLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
};

template <typename Dtype>
class PairRankLossLayer : public LossLayer<Dtype> {
 public:
  explicit PairRankLossLayer(const LayerParameter& param) : LossLayer<Dtype>(param) {}
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  const char* type() const override { return "PairRankLoss"; }
 protected:
  void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override;
  void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override;
  void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override;
  Dtype margin_;
  Blob<Dtype> ordered_diff_;
  Blob<Dtype> similar_diff_;
  Blob<Dtype> workspace_;   // ADDED by path A (loss partials of batches beyond 8192 elements)
};

}  // namespace caffe
#endif
