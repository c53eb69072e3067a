// csrc/caffe_layers.cpp -- SimCross / SimMatrix / PairRankLoss as Caffe layers
// whose Forward_gpu / Backward_gpu are the HIP kernels behind include/mms.h,
// plus fillers, the prototxt-subset reader and the C handle API of
// include/mms_layer.h.  Builds into libmms_caffe.so (links libmms_hip.so).
//
// Layer set-up, shape rules, parameter blob order and every reference-visible
// quirk follow the reference sources cited next to each method; the compute
// bodies are not restated here at all -- they are calls into the C ABI.
// CPU mode is deliberately absent: this library is the GPU implementation,
// and a silent host fallback would void every parity claim.
#include <map>
#include <set>
#include <type_traits>
#include "caffe_api.hpp"

#include <cctype>

#include <fstream>

#include "hdf5_io.hpp"
#include "mms.h"
#include "mms_layer.h"

namespace caffe {

// --------------------------------- RNG + fillers -----------------------------
std::mt19937& caffe_rng() {
  static std::mt19937 g(std::random_device{}());
  return g;
}
void caffe_set_random_seed(unsigned seed) { caffe_rng().seed(seed); }

template <typename Dtype>
class ConstantFiller : public Filler<Dtype> {
 public:
  using Filler<Dtype>::Filler;
  void Fill(Blob<Dtype>* blob) override {
    Dtype* data = blob->mutable_cpu_data();
    const Dtype v = this->filler_param_.value();
    for (int i = 0; i < blob->count(); ++i) data[i] = v;
  }
};
template <typename Dtype>
class UniformFiller : public Filler<Dtype> {
 public:
  using Filler<Dtype>::Filler;
  void Fill(Blob<Dtype>* blob) override {
    CHECK(blob->count());
    std::uniform_real_distribution<Dtype> d(this->filler_param_.min(), this->filler_param_.max());
    Dtype* data = blob->mutable_cpu_data();
    for (int i = 0; i < blob->count(); ++i) data[i] = d(caffe_rng());
  }
};
template <typename Dtype>
class GaussianFiller : public Filler<Dtype> {
 public:
  using Filler<Dtype>::Filler;
  void Fill(Blob<Dtype>* blob) override {
    CHECK(blob->count());
    std::normal_distribution<Dtype> d(this->filler_param_.mean(), this->filler_param_.std());
    Dtype* data = blob->mutable_cpu_data();
    for (int i = 0; i < blob->count(); ++i) data[i] = d(caffe_rng());
  }
};
template <typename Dtype>
class XavierFiller : public Filler<Dtype> {  // fan-in variance norm
 public:
  using Filler<Dtype>::Filler;
  void Fill(Blob<Dtype>* blob) override {
    CHECK(blob->count());
    const int fan_in = blob->count() / blob->num();
    const Dtype scale = std::sqrt(Dtype(3) / fan_in);
    std::uniform_real_distribution<Dtype> d(-scale, scale);
    Dtype* data = blob->mutable_cpu_data();
    for (int i = 0; i < blob->count(); ++i) data[i] = d(caffe_rng());
  }
};
template <typename Dtype>
Filler<Dtype>* GetFiller(const FillerParameter& param) {
  const string& type = param.type();
  if (type == "constant") return new ConstantFiller<Dtype>(param);
  if (type == "uniform") return new UniformFiller<Dtype>(param);
  if (type == "gaussian") return new GaussianFiller<Dtype>(param);
  if (type == "xavier") return new XavierFiller<Dtype>(param);
  CHECK(false) << "Unknown filler name: " << type;
  return nullptr;
}
template Filler<float>* GetFiller<float>(const FillerParameter&);
template Filler<double>* GetFiller<double>(const FillerParameter&);

static void mms_check(int rc, const char* what) {
  CHECK_EQ(rc, (int)MMS_OK) << what << ": " << mms_error_string(rc);
}

// A Layer host built against one include/mms.h must not run on a libmms_hip.so built from another (argument lists
// changed between ABI versions): checked once when this library is loaded, fatal like every other CHECK here.
namespace {
struct AbiVersionCheck {
  AbiVersionCheck() {
    CHECK_EQ(mms_version(), (int)MMS_VERSION) << "libmms_hip.so ABI version differs from the include/mms.h this "
                                                 "Layer library was compiled against; rebuild both";
  }
} g_abi_version_check;
}  // namespace

// caffe_gpu_dot (src/caffe/util/math_functions.cu): the product is formed on the device, one scalar comes back
template <typename T, typename F>
static T gpu_dot_impl(int n, const T* x, const T* y, F fn) {
  static thread_local T* dev = nullptr;
  if (!dev && hipMalloc(reinterpret_cast<void**>(&dev), sizeof(T)) != hipSuccess) MMS_FATAL("") << "hipMalloc failed";
  mms_check(fn(n, x, y, dev, nullptr), "mms_dot");
  T host = 0;
  if (hipMemcpy(&host, dev, sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) MMS_FATAL("") << "hipMemcpy failed";
  return host;
}
float caffe_gpu_dot(int n, const float* x, const float* y) { return gpu_dot_impl<float>(n, x, y, mms_dot_f32); }
double caffe_gpu_dot(int n, const double* x, const double* y) { return gpu_dot_impl<double>(n, x, y, mms_dot_f64); }
// The C ABI by element type: what lets one layer template serve Layer<float> and Layer<double>.
namespace abi {
inline size_t simcross_ws(float, int mode, int N, int W1, int W2, int D, int M) { return mms_simcross_workspace_bytes(mode, N, W1, W2, D, M); }
inline size_t simcross_ws(double, int mode, int N, int W1, int W2, int D, int M) { return mms_simcross_workspace_bytes_f64(mode, N, W1, W2, D, M); }
inline int simcross_forward(int mode, int N, int W1, int W2, int D, int M, const float* q, const float* a, const float* W,
                            const float* bias, float* top, float* n0, float* n1, void* ws, size_t wsb) {
  return mms_simcross_forward_f32(mode, N, W1, W2, D, M, q, a, W, bias, top, n0, n1, ws, wsb, nullptr);
}
inline int simcross_forward(int mode, int N, int W1, int W2, int D, int M, const double* q, const double* a, const double* W,
                            const double* bias, double* top, double* n0, double* n1, void* ws, size_t wsb) {
  return mms_simcross_forward_f64(mode, N, W1, W2, D, M, q, a, W, bias, top, n0, n1, ws, wsb, nullptr);
}
inline int simcross_backward(int mode, int N, int W1, int W2, int D, int M, const float* q, const float* a, const float* W,
                             int bias_term, const float* top, const float* dT, const float* n0, const float* n1, int pd0,
                             int pd1, float* dq, float* da, float* dW, float* db, void* ws, size_t wsb) {
  return mms_simcross_backward_f32(mode, N, W1, W2, D, M, q, a, W, bias_term, top, dT, n0, n1, pd0, pd1, dq, da, dW, db, ws,
                                   wsb, nullptr);
}
inline int simcross_backward(int mode, int N, int W1, int W2, int D, int M, const double* q, const double* a, const double* W,
                             int bias_term, const double* top, const double* dT, const double* n0, const double* n1, int pd0,
                             int pd1, double* dq, double* da, double* dW, double* db, void* ws, size_t wsb) {
  return mms_simcross_backward_f64(mode, N, W1, W2, D, M, q, a, W, bias_term, top, dT, n0, n1, pd0, pd1, dq, da, dW, db, ws,
                                   wsb, nullptr);
}
inline size_t simmatrix_ws(float, int N, int K1, int K2) { return mms_simmatrix_workspace_bytes(N, K1, K2); }
inline size_t simmatrix_ws(double, int, int, int) { return 0; }
inline int simmatrix_forward(int N, int K1, int K2, const float* q, const float* a, const float* W, float* top, float* scr,
                             void* ws, size_t wsb) {
  if (ws) return mms_simmatrix_forward_ws_f32(N, K1, K2, q, a, W, top, scr, ws, wsb, nullptr);   // the layer's own workspace
  return mms_simmatrix_forward_f32(N, K1, K2, q, a, W, top, scr, nullptr);
}
inline int simmatrix_forward(int N, int K1, int K2, const double* q, const double* a, const double* W, double* top, double* scr,
                             void*, size_t) {
  return mms_simmatrix_forward_f64(N, K1, K2, q, a, W, top, scr, nullptr);
}
inline int simmatrix_backward_cached(int N, int K1, int K2, const float* q, const float* a, const float* W, const float* qw,
                                     const float* dT, int ppd, int pd0, int pd1, float* dq, float* da, float* dW, void* ws, size_t wsb) {
  return mms_simmatrix_backward_cached_f32(N, K1, K2, q, a, W, qw, dT, ppd, pd0, pd1, dq, da, dW, ws, wsb, nullptr);
}
inline int simmatrix_backward_cached(int N, int K1, int K2, const double* q, const double* a, const double* W, const double*,
                                     const double* dT, int ppd, int pd0, int pd1, double* dq, double* da, double* dW, void*, size_t) {
  return mms_simmatrix_backward_f64(N, K1, K2, q, a, W, dT, ppd, pd0, pd1, dq, da, dW, nullptr);   // functional path: recomputes
}
inline int simmatrix_backward(int N, int K1, int K2, const float* q, const float* a, const float* W, const float* dT, int ppd,
                              int pd0, int pd1, float* dq, float* da, float* dW, void* ws, size_t wsb) {
  return mms_simmatrix_backward_f32(N, K1, K2, q, a, W, dT, ppd, pd0, pd1, dq, da, dW, ws, wsb, nullptr);
}
inline int simmatrix_backward(int N, int K1, int K2, const double* q, const double* a, const double* W, const double* dT,
                              int ppd, int pd0, int pd1, double* dq, double* da, double* dW, void*, size_t) {
  return mms_simmatrix_backward_f64(N, K1, K2, q, a, W, dT, ppd, pd0, pd1, dq, da, dW, nullptr);
}
inline size_t pairrank_ws(float, int count) { return mms_pairrank_workspace_bytes(count); }
inline size_t pairrank_ws(double, int) { return 0; }
inline int pairrank_forward(int count, float margin, const float* a, const float* b, const float* y, float* o, float* s,
                            float* loss, void* ws, size_t wsb) {
  return mms_pairrank_forward_f32(count, margin, a, b, y, o, s, loss, ws, wsb, nullptr);
}
inline int pairrank_forward(int count, double margin, const double* a, const double* b, const double* y, double* o, double* s,
                            double* loss, void*, size_t) {
  return mms_pairrank_forward_f64(count, margin, a, b, y, o, s, loss, nullptr);
}
inline int pairrank_backward(int count, float td, const float* y, const float* o, const float* s, int pd0, int pd1, float* da,
                             float* db) {
  return mms_pairrank_backward_f32(count, td, y, o, s, pd0, pd1, da, db, nullptr);
}
inline int pairrank_backward(int count, double td, const double* y, const double* o, const double* s, int pd0, int pd1,
                             double* da, double* db) {
  return mms_pairrank_backward_f64(count, td, y, o, s, pd0, pd1, da, db, nullptr);
}
}  // namespace abi

#define NO_CPU_MODE MMS_FATAL("") << this->type() << " Layer: libmms is the GPU (HIP) implementation; " \
  "CPU mode is served by the reference's own Forward_cpu/Backward_cpu, not by this library."

// ===================================== SimCross ==============================
// Reference: include/caffe/layers/sim_cross_layer.hpp, src/caffe/layers/sim_cross_layer.cpp
template <typename Dtype>
class SimCrossLayer : public Layer<Dtype> {
 public:
  explicit SimCrossLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  const char* type() const override { return "SimCross"; }
  int ExactNumBottomBlobs() const override { return 2; }
  int ExactNumTopBlobs() const override { return 1; }

  // sim_cross_layer.cpp:10-47.  Quirks kept: pre-loaded blobs_ are NOT honoured
  // (no "Skipping parameter initialization" branch) and param_propagate_down_
  // is never sized.
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK(bottom.size() == 2);
    CHECK(bottom[0]->num() == bottom[1]->num());
    CHECK(bottom[0]->height() == bottom[1]->height());
    const SimCrossParameter& p = this->layer_param_.sim_cross_param();
    dist_mode_ = p.dist_mode();
    CHECK(dist_mode_ >= 0 && dist_mode_ <= 2) << "dist_mode must be 0 (cosine), 1 (euclid) or 2 (bilinear)";
    if (dist_mode_ == 2) {
      const bool bias_term = p.bias_term();
      this->blobs_.resize(bias_term ? 2 : 1);
      this->blobs_[0].reset(new Blob<Dtype>(vector<int>{p.mesure_count(), bottom[0]->height(), bottom[1]->height()}));
      shared_ptr<Filler<Dtype> > wf(GetFiller<Dtype>(p.weight_filler()));
      wf->Fill(this->blobs_[0].get());
      if (bias_term) {
        this->blobs_[1].reset(new Blob<Dtype>(vector<int>{p.mesure_count(), bottom[0]->channels(), bottom[1]->channels()}));
        shared_ptr<Filler<Dtype> > bf(GetFiller<Dtype>(p.bias_filler()));
        bf->Fill(this->blobs_[1].get());
      }
    }
  }

  // sim_cross_layer.cpp:50-80.  top = (N, M|1, W1, W2); data{0,1}_norm_ for mode 0.
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    const int M = dist_mode_ == 2 ? this->layer_param_.sim_cross_param().mesure_count() : 1;
    top[0]->Reshape(vector<int>{bottom[0]->num(), M, bottom[0]->channels(), bottom[1]->channels()});
    if (dist_mode_ == 0) {
      data0_norm_.Reshape(vector<int>{bottom[0]->num(), bottom[0]->channels()});
      data1_norm_.Reshape(vector<int>{bottom[1]->num(), bottom[1]->channels()});
    }
    const size_t ws = abi::simcross_ws(Dtype(0), dist_mode_, bottom[0]->num(), bottom[0]->channels(),
                                       bottom[1]->channels(), bottom[0]->height(), M);
    if (ws) workspace_.Reshape(vector<int>{(int)((ws + sizeof(Dtype) - 1) / sizeof(Dtype))});
  }

  // "euclid_backward_mode": MMS_EUCLID_BWD_FP32 (0) / MMS_EUCLID_BWD_REFERENCE (1) for THIS layer; -1 = the thread's
  bool SetOption(const std::string& key, int value) override {
    if (key != "euclid_backward_mode" || value < -1 || value > 1) return false;
    euclid_bwd_mode_ = value;
    return true;
  }

  // Scoring straight from word ids (PathNet's "fuse_embed_scoring"): top = this layer applied to
  // (Embed(ids_q), Embed(ids_a)) with the Embed layers' shared table and bias, the gather done by the forward
  // kernels' own loads (include/mms.h: mms_embed_simcross_forward_f32 / _bilinear_forward_f32).  `bottom` are the
  // layer's ordinary bottoms -- shapes only, their data is neither read nor written.  Returns false when the
  // geometry has no fused kernel (the caller then runs the Embed layers and Forward as usual).
  bool ForwardFromWordIds(const Blob<Dtype>& ids_q, const Blob<Dtype>& ids_a, const Blob<Dtype>& table,
                          const Blob<Dtype>* embed_bias, const vector<Blob<Dtype>*>& bottom,
                          const vector<Blob<Dtype>*>& top) {
    if constexpr (std::is_same<Dtype, float>::value) {
      this->Reshape(bottom, top);
      const int N = bottom[0]->num(), W1 = bottom[0]->channels(), W2 = bottom[1]->channels(), D = bottom[0]->height();
      if (ids_q.count() != N * W1 || ids_a.count() != N * W2 || table.num_axes() != 2 || table.shape(1) != D) return false;
      const int K = table.shape(0), M = top[0]->channels();
      const float* eb = embed_bias ? embed_bias->gpu_data() : nullptr;
      int rc;
      if (dist_mode_ == 2)
        rc = mms_embed_simcross_bilinear_forward_f32(N, W1, W2, D, M, K, ids_q.gpu_data(), ids_a.gpu_data(),
                                                     table.gpu_data(), eb, this->blobs_[0]->gpu_data(),
                                                     this->blobs_.size() > 1 ? this->blobs_[1]->gpu_data() : nullptr,
                                                     top[0]->mutable_gpu_data(), nullptr);
      else
        rc = mms_embed_simcross_forward_f32(dist_mode_, N, W1, W2, D, K, ids_q.gpu_data(), ids_a.gpu_data(),
                                            table.gpu_data(), eb, top[0]->mutable_gpu_data(),
                                            dist_mode_ == 0 ? data0_norm_.mutable_gpu_data() : nullptr,
                                            dist_mode_ == 0 ? data1_norm_.mutable_gpu_data() : nullptr, nullptr);
      if (rc == MMS_ERR_UNSUPPORTED) return false;
      mms_check(rc, "mms_embed_simcross_forward");
      return true;
    } else {
      return false;                                    // the fused entry points are fp32
    }
  }

 protected:
  void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }
  void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }

  // replaces sim_cross_layer.cpp:83-163 / sim_cross_layer.cu:128-194
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    const int M = top[0]->channels();
    const bool mode2 = dist_mode_ == 2, mode0 = dist_mode_ == 0;
    mms_check(abi::simcross_forward(
                  dist_mode_, bottom[0]->num(), bottom[0]->channels(), bottom[1]->channels(),
                  bottom[0]->height(), M, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
                  mode2 ? this->blobs_[0]->gpu_data() : nullptr,
                  (mode2 && this->blobs_.size() > 1) ? this->blobs_[1]->gpu_data() : nullptr,
                  top[0]->mutable_gpu_data(), mode0 ? data0_norm_.mutable_gpu_data() : nullptr,
                  mode0 ? data1_norm_.mutable_gpu_data() : nullptr,
                  workspace_.count() ? workspace_.mutable_gpu_data() : nullptr,
                  (size_t)workspace_.count() * sizeof(Dtype)),
              "mms_simcross_forward");
  }

  // replaces sim_cross_layer.cpp:166-307 / sim_cross_layer.cu:197-243
  void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override {
    const int M = top[0]->channels();
    const bool mode2 = dist_mode_ == 2, mode0 = dist_mode_ == 0;
    const bool bias_term = mode2 && this->blobs_.size() > 1;
    // this layer's arithmetic of the Euclidean backward term, if one was pinned (SetOption): the C ABI's mode
    // belongs to the calling thread, so it is set for the duration of this call and put back
    struct ModeGuard {
      int saved;
      explicit ModeGuard(int m) : saved(-1) {
        if (m >= 0) { saved = mms_get_euclid_backward_mode(); mms_set_euclid_backward_mode(m); }
      }
      ~ModeGuard() { if (saved >= 0) mms_set_euclid_backward_mode(saved); }
    } guard(euclid_bwd_mode_);
    mms_check(abi::simcross_backward(
                  dist_mode_, bottom[0]->num(), bottom[0]->channels(), bottom[1]->channels(),
                  bottom[0]->height(), M, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
                  mode2 ? this->blobs_[0]->gpu_data() : nullptr, bias_term, top[0]->gpu_data(),
                  top[0]->gpu_diff(), mode0 ? data0_norm_.gpu_data() : nullptr,
                  mode0 ? data1_norm_.gpu_data() : nullptr, propagate_down[0], propagate_down[1],
                  bottom[0]->mutable_gpu_diff(), bottom[1]->mutable_gpu_diff(),
                  mode2 ? this->blobs_[0]->mutable_gpu_diff() : nullptr,
                  bias_term ? this->blobs_[1]->mutable_gpu_diff() : nullptr,
                  workspace_.count() ? workspace_.mutable_gpu_data() : nullptr,
                  (size_t)workspace_.count() * sizeof(Dtype)),
              "mms_simcross_backward");
  }

  int dist_mode_ = 1;  // 1 euclid, 0 cosine, 2 bilinear (sim_cross_layer.hpp:36)
  int euclid_bwd_mode_ = -1;  // -1: the calling thread's mode; else MMS_EUCLID_BWD_FP32 / _REFERENCE for this layer
  Blob<Dtype> data0_norm_, data1_norm_;
  Blob<Dtype> workspace_;  // replaces measure_temp{0,1}_
};
INSTANTIATE_CLASS_FD(SimCrossLayer);
REGISTER_LAYER_CLASS_FD(SimCross);

// ===================================== SimMatrix =============================
// Reference: include/caffe/layers/sim_matrix_layer.hpp, src/caffe/layers/sim_matrix_layer.cpp
template <typename Dtype>
class SimMatrixLayer : public Layer<Dtype> {
 public:
  explicit SimMatrixLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  const char* type() const override { return "SimMatrix"; }
  int ExactNumBottomBlobs() const override { return 2; }
  int ExactNumTopBlobs() const override { return 1; }

  // sim_matrix_layer.cpp:10-34 (honours pre-loaded blobs)
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK_EQ(bottom[0]->num(), bottom[1]->num());
    K1_ = bottom[0]->count(1);
    K2_ = bottom[1]->count(1);
    if (this->blobs_.size() > 0) {
      LOG_INFO << "Skipping parameter initialization";
    } else {
      this->blobs_.resize(1);
      this->blobs_[0].reset(new Blob<Dtype>(vector<int>{K1_, K2_}));
      shared_ptr<Filler<Dtype> > wf(GetFiller<Dtype>(this->layer_param_.sim_matrix_param().weight_filler()));
      wf->Fill(this->blobs_[0].get());
    }
    this->param_propagate_down_.resize(this->blobs_.size(), true);
  }

  // sim_matrix_layer.cpp:37-50
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK_EQ(K1_, bottom[0]->count(1)) << "Input size incompatible with inner product parameters.";
    CHECK_EQ(K2_, bottom[1]->count(1)) << "Input size incompatible with inner product parameters.";
    M_ = bottom[0]->count(0, 1);
    top[0]->Reshape(vector<int>{bottom[0]->shape(0), 1});
    const size_t ws = abi::simmatrix_ws(Dtype(0), M_, K1_, K2_);
    workspace_.Reshape(vector<int>{(int)((ws + sizeof(Dtype) - 1) / sizeof(Dtype))});
    if (qw_elems_ != M_ * K2_) { qw_valid_ = false; qw_elems_ = M_ * K2_; }
    if (private_qw_) qw_.Reshape(vector<int>{M_, K2_});
  }

 protected:
  void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }
  void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }

  // replaces sim_matrix_layer.cpp:53-65.  Like the reference, Forward leaves Q*W in bottom[1]'s DIFF buffer
  // (:58 takes bottom[1]->mutable_cpu_diff() as its scratch) -- an observable side effect, kept by default.
  // Backward then finds the product where the reference's forward left it and scales it in place into
  // da_j = dT_j * (W^T q_j) (:88) instead of running the same GEMM again: nothing in a Caffe net writes a
  // bottom's diff between a layer's Forward and its own Backward.  A host that does touch that buffer in
  // between can switch the layer to a private copy: SetOption("private_qw", 1) (then Forward leaves
  // bottom[1]'s diff alone, which is the only difference).
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    Dtype* qw = private_qw_ ? qw_.mutable_gpu_data() : bottom[1]->mutable_gpu_diff();
    mms_check(abi::simmatrix_forward(M_, K1_, K2_, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
                                     this->blobs_[0]->gpu_data(), top[0]->mutable_gpu_data(), qw,
                                     workspace_.count() ? workspace_.mutable_gpu_data() : nullptr,
                                     (size_t)workspace_.count() * sizeof(Dtype)),
              "mms_simmatrix_forward");
    qw_valid_ = true;
  }
  // replaces sim_matrix_layer.cpp:68-95
  void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override {
    const bool ppd = this->param_propagate_down_[0];
    if (qw_valid_) {
      const Dtype* qw = private_qw_ ? qw_.gpu_data() : bottom[1]->gpu_diff();   // in place when it is the diff
      mms_check(abi::simmatrix_backward_cached(
                    M_, K1_, K2_, bottom[0]->gpu_data(), bottom[1]->gpu_data(), this->blobs_[0]->gpu_data(),
                    qw, top[0]->gpu_diff(), ppd, propagate_down[0], propagate_down[1],
                    propagate_down[0] ? bottom[0]->mutable_gpu_diff() : nullptr,
                    propagate_down[1] ? bottom[1]->mutable_gpu_diff() : nullptr,
                    ppd ? this->blobs_[0]->mutable_gpu_diff() : nullptr,
                    workspace_.count() ? workspace_.mutable_gpu_data() : nullptr,
                    (size_t)workspace_.count() * sizeof(Dtype)),
                "mms_simmatrix_backward_cached");
      if (!private_qw_ && propagate_down[1]) qw_valid_ = false;   // the diff now holds da, not Q*W
      return;
    }
    mms_check(abi::simmatrix_backward(
                  M_, K1_, K2_, bottom[0]->gpu_data(), bottom[1]->gpu_data(), this->blobs_[0]->gpu_data(),
                  top[0]->gpu_diff(), ppd, propagate_down[0], propagate_down[1],
                  propagate_down[0] ? bottom[0]->mutable_gpu_diff() : nullptr,
                  propagate_down[1] ? bottom[1]->mutable_gpu_diff() : nullptr,
                  ppd ? this->blobs_[0]->mutable_gpu_diff() : nullptr,
                  workspace_.count() ? workspace_.mutable_gpu_data() : nullptr,
                  (size_t)workspace_.count() * sizeof(Dtype)),
              "mms_simmatrix_backward");
  }
 public:
  bool SetOption(const std::string& key, int value) override {
    if (key != "private_qw") return false;
    private_qw_ = value != 0;
    qw_valid_ = false;
    return true;
  }
 protected:
  int K1_ = 0, K2_ = 0, M_ = 0;
  Blob<Dtype> workspace_;
  Blob<Dtype> qw_;             // private_qw: Q*W of the last Forward (M_, K2_)
  bool qw_valid_ = false;
  bool private_qw_ = false;
  int qw_elems_ = 0;
};
INSTANTIATE_CLASS_FD(SimMatrixLayer);
REGISTER_LAYER_CLASS_FD(SimMatrix);

// ================================ LossLayer / PairRankLoss ===================
// Reference: include/caffe/layers/loss_layer.hpp:22-49, src/caffe/layers/loss_layer.cpp:8-23
template <typename Dtype>
class LossLayer : public Layer<Dtype> {
 public:
  explicit LossLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    if (this->layer_param_.loss_weight_size() == 0) this->layer_param_.add_loss_weight(Dtype(1));
  }
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK_EQ(bottom[0]->num(), bottom[1]->num()) << "The data and label should have the same number.";
    top[0]->Reshape(vector<int>());  // scalar, 0 axes
  }
  int ExactNumBottomBlobs() const override { return 2; }
  bool AutoTopBlobs() const override { return true; }
  int ExactNumTopBlobs() const override { return 1; }
  bool AllowForceBackward(int bottom_index) const override { return bottom_index != 1; }
};

// Reference: include/caffe/layers/pair_rank_loss_layer.hpp, src/caffe/layers/pair_rank_loss_layer.cpp
template <typename Dtype>
class PairRankLossLayer : public LossLayer<Dtype> {
 public:
  explicit PairRankLossLayer(const LayerParameter& param) : LossLayer<Dtype>(param) {}
  const char* type() const override { return "PairRankLoss"; }
  int ExactNumBottomBlobs() const override { return 3; }
  int ExactNumTopBlobs() const override { return -1; }
  int MinTopBlobs() const override { return 1; }
  int MaxTopBlobs() const override { return 2; }  // declared; only top[0] is ever written

  // pair_rank_loss_layer.cpp:10-23.  The caches are shaped ONCE here (there is no
  // Reshape override), so the batch must not grow after SetUp -- as in the reference.
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    LossLayer<Dtype>::LayerSetUp(bottom, top);
    CHECK_EQ(bottom[0]->num(), bottom[1]->num());
    CHECK_EQ(bottom[0]->num(), bottom[2]->num());
    CHECK_EQ(bottom[0]->count(1), bottom[2]->count(1));
    CHECK_EQ(bottom[0]->count(1), bottom[1]->count(1));
    margin_ = (Dtype)this->layer_param_.pair_rank_loss_param().margin();
    ordered_diff_.Reshape(bottom[0]->num(), bottom[0]->channels(), 1, 1);
    similar_diff_.Reshape(bottom[0]->num(), bottom[0]->channels(), 1, 1);
    const size_t ws = abi::pairrank_ws(Dtype(0), ordered_diff_.count());
    if (ws) workspace_.Reshape(vector<int>{(int)((ws + sizeof(Dtype) - 1) / sizeof(Dtype))});
  }

 protected:
  void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }
  void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }

  // replaces pair_rank_loss_layer.cpp:26-52 / .cu:10-43
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    const int count = bottom[0]->count();
    CHECK_LE(count, ordered_diff_.count()) << "PairRankLoss caches are sized in LayerSetUp; batch grew";
    mms_check(abi::pairrank_forward(count, margin_, bottom[0]->gpu_data(), bottom[1]->gpu_data(),
                                    bottom[2]->gpu_data(), ordered_diff_.mutable_gpu_data(),
                                    similar_diff_.mutable_gpu_data(), top[0]->mutable_gpu_data(),
                                    workspace_.count() ? workspace_.mutable_gpu_data() : nullptr,
                                    (size_t)workspace_.count() * sizeof(Dtype)),
              "mms_pairrank_forward");
  }
  // replaces pair_rank_loss_layer.cpp:55-84 (CPU semantics: strict '>')
  void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override {
    if (propagate_down[2]) LOG_FATAL << this->type() << " Layer cannot backpropagate to label inputs.";
    if (!propagate_down[0] && !propagate_down[1]) return;
    mms_check(abi::pairrank_backward(bottom[0]->count(), top[0]->cpu_diff()[0], bottom[2]->gpu_data(),
                                     ordered_diff_.gpu_data(), similar_diff_.gpu_data(),
                                     propagate_down[0], propagate_down[1],
                                     propagate_down[0] ? bottom[0]->mutable_gpu_diff() : nullptr,
                                     propagate_down[1] ? bottom[1]->mutable_gpu_diff() : nullptr),
              "mms_pairrank_backward");
  }
  Dtype margin_ = 1;
  Blob<Dtype> ordered_diff_, similar_diff_, workspace_;
};
INSTANTIATE_CLASS_FD(PairRankLossLayer);
REGISTER_LAYER_CLASS_FD(PairRankLoss);

// ======================================= Embed ===============================
// Reference: include/caffe/layers/embed_layer.hpp, src/caffe/layers/embed_layer.cpp (the fork
// adds `weight_source`, :46-113: word vectors loaded at set-up from a text / "all" / word2vec
// binary file).  Blob order [weight (K,N), bias (N)].
template <typename Dtype>
class EmbedLayer : public Layer<Dtype> {
 public:
  explicit EmbedLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  const char* type() const override { return "Embed"; }
  int ExactNumBottomBlobs() const override { return 1; }
  int ExactNumTopBlobs() const override { return 1; }

  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    const EmbedParameter& p = this->layer_param_.embed_param();
    N_ = p.num_output();
    CHECK_GT(N_, 0) << "EmbedLayer num_output must be positive.";
    K_ = p.input_dim();
    CHECK_GT(K_, 0) << "EmbedLayer input_dim must be positive.";
    bias_term_ = p.bias_term();
    if (this->blobs_.size() > 0) {
      LOG_INFO << "Skipping parameter initialization";
    } else {
      this->blobs_.resize(bias_term_ ? 2 : 1);
      this->blobs_[0].reset(new Blob<Dtype>(vector<int>{K_, N_}));
      shared_ptr<Filler<Dtype> > wf(GetFiller<Dtype>(p.weight_filler()));
      wf->Fill(this->blobs_[0].get());
      if (bias_term_) {
        this->blobs_[1].reset(new Blob<Dtype>(vector<int>{N_}));
        shared_ptr<Filler<Dtype> > bf(GetFiller<Dtype>(p.bias_filler()));
        bf->Fill(this->blobs_[1].get());
      }
      if (!p.weight_source().empty()) LoadWeightSource(p.weight_source());
    }
    this->param_propagate_down_.resize(this->blobs_.size(), true);
  }

  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    M_ = bottom[0]->count();
    vector<int> top_shape = bottom[0]->shape();
    top_shape.push_back(N_);
    top[0]->Reshape(top_shape);
    const size_t ws = mms_embed_workspace_bytes(M_, N_);
    workspace_.Reshape(vector<int>{(int)((ws + sizeof(Dtype) - 1) / sizeof(Dtype))});
  }

 protected:
  // embed_layer.cpp:46-113.  Rows are filled from row 0 in file order; the formats are
  // told apart by the last three characters of the file name, as in the reference.
  void LoadWeightSource(const string& path) {
    Dtype* w = this->blobs_[0]->mutable_cpu_data();
    const size_t cap = (size_t)K_ * N_;
    size_t wi = 0;
    const string ext = path.size() >= 3 ? path.substr(path.size() - 3) : string();
    FILE* f = std::fopen(path.c_str(), ext == "txt" || ext == "all" ? "r" : "rb");
    CHECK(f != nullptr) << "cannot open weight_source " << path;
    char word[256];
    if (ext == "txt") {                       // "<word> v1 ... vN" per line
      while (std::fscanf(f, "%255s ", word) != EOF)
        for (int i = 0; i < N_; ++i) {
          float v = 0;
          CHECK_EQ(std::fscanf(f, "%f ", &v), 1) << "truncated vector in " << path;
          CHECK_LT(wi, cap) << "weight_source has more rows than input_dim";
          w[wi++] = v;
        }
    } else if (ext == "all") {                // header "<float> <K-1> <N-1>", then "<id> v1 ... vN <word>"
      float b1; int t1, t2;
      CHECK_EQ(std::fscanf(f, "%f %d %d", &b1, &t1, &t2), 3);
      CHECK_EQ(t1, K_ - 1);
      CHECK_EQ(t2, N_ - 1);
      while (std::fscanf(f, "%d ", &t1) != EOF) {
        for (int i = 0; i < N_; ++i) {
          float v = 0;
          CHECK_EQ(std::fscanf(f, "%f ", &v), 1);
          CHECK_LT(wi, cap);
          w[wi++] = v;
        }
        CHECK_EQ(std::fscanf(f, "%255s", word), 1);
      }
    } else {                                  // word2vec binary: "<vocab> <dim>", then "<word> " + dim floats
      long long vsize = 0, dim = 0;
      CHECK_EQ(std::fscanf(f, "%lld", &vsize), 1);
      CHECK_EQ(std::fscanf(f, "%lld", &dim), 1);
      CHECK_EQ(dim, (long long)N_);
      for (long long b = 0; b < vsize; ++b) {
        int a = 0;
        while (true) {
          const int c = std::fgetc(f);
          if (c == EOF || c == ' ') break;
          if (a < 255 && c != '\n') word[a++] = (char)c;
        }
        for (long long i = 0; i < dim; ++i) {
          float v = 0;
          CHECK_EQ(std::fread(&v, sizeof(float), 1, f), (size_t)1);
          CHECK_LT(wi, cap);
          w[wi++] = v;
        }
      }
    }
    std::fclose(f);
  }

  void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }
  void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }

  // replaces embed_layer.cpp:135-152 / embed_layer.cu:42-62
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    mms_check(mms_embed_forward_f32(M_, N_, K_, bottom[0]->gpu_data(), this->blobs_[0]->gpu_data(),
                                    bias_term_ ? this->blobs_[1]->gpu_data() : nullptr,
                                    top[0]->mutable_gpu_data(), nullptr),
              "mms_embed_forward_f32");
  }
  // replaces embed_layer.cpp:155-180 / embed_layer.cu:64-88 (atomicAdd there; ordered sums here)
  void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                    const vector<Blob<Dtype>*>& bottom) override {
    CHECK(!propagate_down[0]) << "Can't backpropagate to EmbedLayer input.";
    const bool pw = this->param_propagate_down_[0];
    const bool pb = bias_term_ && this->param_propagate_down_[1];
    if (!pw && !pb) return;
    mms_check(mms_embed_backward_f32(M_, N_, K_, bottom[0]->gpu_data(), top[0]->gpu_diff(),
                                     pw ? this->blobs_[0]->mutable_gpu_diff() : nullptr,
                                     pb ? this->blobs_[1]->mutable_gpu_diff() : nullptr,
                                     workspace_.mutable_gpu_data(),
                                     (size_t)workspace_.count() * sizeof(Dtype), nullptr),
              "mms_embed_backward_f32");
  }
  int M_ = 0, K_ = 0, N_ = 0;
  bool bias_term_ = true;
  Blob<Dtype> workspace_;

 public:
  // Two Embed layers over ONE table and bias (network_v4's w2v_q / w2v_a, shared by parameter name), run as a pair by
  // PathNet: `later` is the layer that comes later in the net file -- Net::Backward reaches it first, so it is layer 0
  // of both calls (include/mms.h: mms_embed_forward_pair_f32 / mms_embed_backward_pair[_indexed]_f32; the bits are those
  // of the two layers run one after the other).  index_ws: the pair's inverted index, built beside the forward's gathers.
  static bool PairForward(EmbedLayer<float>& later, EmbedLayer<float>& earlier, const vector<Blob<float>*>& bl,
                          const vector<Blob<float>*>& tl, const vector<Blob<float>*>& be, const vector<Blob<float>*>& te,
                          Blob<float>* index_ws, bool* indexed) {
    later.Reshape(bl, tl);
    earlier.Reshape(be, te);
    if (later.N_ != earlier.N_ || later.K_ != earlier.K_ || later.bias_term_ != earlier.bias_term_) return false;
    const size_t need = mms_embed_workspace_bytes(later.M_ + earlier.M_, later.N_);
    index_ws->Reshape(vector<int>{(int)((need + sizeof(float) - 1) / sizeof(float))});
    *indexed = mms_embed_pair_index_supported(later.M_, earlier.M_, later.K_) != 0;
    mms_check(mms_embed_forward_pair_f32(later.M_, earlier.M_, later.N_, later.K_, bl[0]->gpu_data(), be[0]->gpu_data(),
                                         later.blobs_[0]->gpu_data(),
                                         later.bias_term_ ? later.blobs_[1]->gpu_data() : nullptr,
                                         tl[0]->mutable_gpu_data(), te[0]->mutable_gpu_data(),
                                         *indexed ? index_ws->mutable_gpu_data() : nullptr,
                                         *indexed ? (size_t)index_ws->count() * sizeof(float) : 0, nullptr),
              "mms_embed_forward_pair_f32");
    return true;
  }
  static void PairBackward(EmbedLayer<float>& later, EmbedLayer<float>& earlier, const vector<Blob<float>*>& bl,
                           const vector<Blob<float>*>& tl, const vector<Blob<float>*>& be, const vector<Blob<float>*>& te,
                           Blob<float>* index_ws, bool indexed) {
    const bool pw = later.param_propagate_down_[0];
    const bool pb = later.bias_term_ && later.param_propagate_down_[1];
    if (!pw && !pb) return;
    float* wd = pw ? later.blobs_[0]->mutable_gpu_diff() : nullptr;
    float* bd = pb ? later.blobs_[1]->mutable_gpu_diff() : nullptr;
    const size_t wsb = (size_t)index_ws->count() * sizeof(float);
    if (indexed)
      mms_check(mms_embed_backward_pair_indexed_f32(later.M_, earlier.M_, later.N_, later.K_, bl[0]->gpu_data(),
                                                    tl[0]->gpu_diff(), be[0]->gpu_data(), te[0]->gpu_diff(), wd, bd,
                                                    index_ws->mutable_gpu_data(), wsb, nullptr),
                "mms_embed_backward_pair_indexed_f32");
    else
      mms_check(mms_embed_backward_pair_f32(later.M_, earlier.M_, later.N_, later.K_, bl[0]->gpu_data(), tl[0]->gpu_diff(),
                                            be[0]->gpu_data(), te[0]->gpu_diff(), wd, bd, index_ws->mutable_gpu_data(), wsb,
                                            nullptr),
                "mms_embed_backward_pair_f32");
  }
};
INSTANTIATE_CLASS(EmbedLayer);
REGISTER_LAYER_CLASS(Embed);

// ===================== MAP / MRR / AUC / RankAccuracy (forward only) =========
// Reference: src/caffe/layers/{map,mrr,auc,rank_accuracy}_layer.cpp and their headers.
// The reference has no GPU code for these (Forward_gpu falls back to Forward_cpu through
// the base class, i.e. a D2H copy of the whole score blob); here they run on the device.
template <typename Dtype>
class RankMetricLayerBase : public Layer<Dtype> {
 public:
  explicit RankMetricLayerBase(const LayerParameter& param) : Layer<Dtype>(param) {}
  int ExactNumTopBlobs() const override { return 1; }
 protected:
  void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }
  // map_layer.cpp / mrr_layer.cpp / auc_layer.cpp: Backward is a no-op for metric layers
  void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override {}
  void Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override {}
  void size_workspace(int n) {
    const size_t ws = mms_rank_workspace_bytes(n);
    workspace_.Reshape(vector<int>{(int)((ws + sizeof(Dtype) - 1) / sizeof(Dtype))});
  }
  Blob<Dtype> workspace_;
};

template <typename Dtype>
class MAPLayer : public RankMetricLayerBase<Dtype> {
 public:
  explicit MAPLayer(const LayerParameter& param) : RankMetricLayerBase<Dtype>(param) {}
  const char* type() const override { return "MAP"; }
  int ExactNumBottomBlobs() const override { return 3; }
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    fixed_axis_ = this->layer_param_.map_param().fixed_axis();           // map_layer.cpp:13
  }
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK_LE(fixed_axis_, bottom[0]->count() / bottom[1]->count())
        << "top_k must be less than or equal to the number of classes.";
    const int outer = bottom[0]->count(0, 1), inner = bottom[0]->count(2);
    CHECK_EQ(outer * inner, bottom[1]->count()) << "Number of labels must match number of predictions";
    CHECK_EQ(outer * inner, bottom[2]->count());
    top[0]->Reshape(vector<int>());
    this->size_workspace(bottom[0]->num());
  }
 protected:
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    mms_check(mms_rank_map_mrr_f32(bottom[0]->num(), fixed_axis_, bottom[0]->gpu_data(),
                                   bottom[1]->gpu_data(), bottom[2]->gpu_data(),
                                   top[0]->mutable_gpu_data(), nullptr, nullptr,
                                   this->workspace_.mutable_gpu_data(),
                                   (size_t)this->workspace_.count() * sizeof(Dtype), nullptr),
              "mms_rank_map_mrr_f32");
  }
  int fixed_axis_ = 1;
};
INSTANTIATE_CLASS(MAPLayer);
REGISTER_LAYER_CLASS(MAP);

template <typename Dtype>
class MRRLayer : public RankMetricLayerBase<Dtype> {
 public:
  explicit MRRLayer(const LayerParameter& param) : RankMetricLayerBase<Dtype>(param) {}
  const char* type() const override { return "MRR"; }
  int ExactNumBottomBlobs() const override { return 3; }
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    fixed_axis_ = this->layer_param_.mrr_param().fixed_axis();           // mrr_layer.cpp:13
  }
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK_LE(fixed_axis_, bottom[0]->count() / bottom[1]->count())
        << "top_k must be less than or equal to the number of classes.";
    const int outer = bottom[0]->count(0, 1), inner = bottom[0]->count(2);
    CHECK_EQ(outer * inner, bottom[1]->count()) << "Number of labels must match number of predictions";
    CHECK_EQ(outer * inner, bottom[2]->count());
    top[0]->Reshape(vector<int>());
    this->size_workspace(bottom[0]->num());
  }
 protected:
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    mms_check(mms_rank_map_mrr_f32(bottom[0]->num(), fixed_axis_, bottom[0]->gpu_data(),
                                   bottom[1]->gpu_data(), bottom[2]->gpu_data(), nullptr,
                                   top[0]->mutable_gpu_data(), nullptr,
                                   this->workspace_.mutable_gpu_data(),
                                   (size_t)this->workspace_.count() * sizeof(Dtype), nullptr),
              "mms_rank_map_mrr_f32");
  }
  int fixed_axis_ = 1;
};
INSTANTIATE_CLASS(MRRLayer);
REGISTER_LAYER_CLASS(MRR);

template <typename Dtype>
class AUCLayer : public RankMetricLayerBase<Dtype> {
 public:
  explicit AUCLayer(const LayerParameter& param) : RankMetricLayerBase<Dtype>(param) {}
  const char* type() const override { return "AUC"; }
  int ExactNumBottomBlobs() const override { return 2; }
  int ExactNumTopBlobs() const override { return -1; }
  int MinTopBlobs() const override { return 1; }
  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    const AUCParameter& p = this->layer_param_.auc_param();               // auc_layer.cpp:14-20
    fixed_axis_ = p.fixed_axis();
    has_ignore_label_ = p.has_ignore_label();
    if (has_ignore_label_) ignore_label_ = p.ignore_label();
  }
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK_LE(fixed_axis_, bottom[0]->count() / bottom[1]->count())
        << "top_k must be less than or equal to the number of classes.";
    label_axis_ = bottom[0]->CanonicalAxisIndex(this->layer_param_.auc_param().axis());
    outer_num_ = bottom[0]->count(0, label_axis_);
    inner_num_ = bottom[0]->count(label_axis_ + 1);
    CHECK_EQ(outer_num_ * inner_num_, bottom[1]->count()) << "Number of labels must match number of predictions";
    top[0]->Reshape(vector<int>());
    this->size_workspace(outer_num_ * inner_num_);
  }
 protected:
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    mms_check(mms_rank_auc_nd_f32(outer_num_, bottom[0]->shape(label_axis_), inner_num_, fixed_axis_,
                                  bottom[0]->gpu_data(), bottom[1]->gpu_data(), has_ignore_label_,
                                  ignore_label_, top[0]->mutable_gpu_data(),
                                  this->workspace_.mutable_gpu_data(),
                                  (size_t)this->workspace_.count() * sizeof(Dtype), nullptr),
              "mms_rank_auc_nd_f32");
  }
  int fixed_axis_ = 1, label_axis_ = 1, outer_num_ = 0, inner_num_ = 1, ignore_label_ = 0;
  bool has_ignore_label_ = false;
};
INSTANTIATE_CLASS(AUCLayer);
REGISTER_LAYER_CLASS(AUC);

template <typename Dtype>
class RankAccuracyLayer : public RankMetricLayerBase<Dtype> {
 public:
  explicit RankAccuracyLayer(const LayerParameter& param) : RankMetricLayerBase<Dtype>(param) {}
  const char* type() const override { return "RankAccuracy"; }
  int ExactNumBottomBlobs() const override { return 3; }
  int ExactNumTopBlobs() const override { return -1; }
  int MinTopBlobs() const override { return 1; }
  void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK_EQ(bottom[0]->count(), bottom[1]->count()) << "two pairs have the same dimension!.";
    CHECK_EQ(bottom[0]->count(), bottom[2]->count()) << "pair should have the same dimension with the label!.";
    top[0]->Reshape(vector<int>());
    this->size_workspace(1);
  }
 protected:
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    mms_check(mms_rank_accuracy_f32(bottom[0]->count(), bottom[0]->gpu_data(), bottom[1]->gpu_data(),
                                    bottom[2]->gpu_data(), top[0]->mutable_gpu_data(),
                                    this->workspace_.mutable_gpu_data(),
                                    (size_t)this->workspace_.count() * sizeof(Dtype), nullptr),
              "mms_rank_accuracy_f32");
  }
};
INSTANTIATE_CLASS(RankAccuracyLayer);
REGISTER_LAYER_CLASS(RankAccuracy);

// ===================================== HDF5Data ==============================
// Reference: include/caffe/layers/hdf5_data_layer.hpp, src/caffe/layers/hdf5_data_layer.cpp
// :27-151 and .cu:19-51.  `source` lists one .h5 file per line; every top is the dataset of
// the same name, loaded whole and converted to float (util/hdf5.cpp:10-73); a batch is the
// next batch_size rows, moving on to the next file (and wrapping) when a file is used up,
// so a batch can straddle two files.  No libhdf5 here: csrc/hdf5_io.cpp decodes the files.
//
// MI355X shape of it: the reference keeps the file's blobs on the host and issues batch_size
// x top_size small copies per Forward (H2D in its GPU path).  Here a file's datasets are
// uploaded to HBM once, when the file is opened, and a batch is one gather launch per top
// and per contiguous run of rows (mms_feed_gather_rows_f32): no per-step PCIe traffic.
//
// shuffle: the reference uses std::random_shuffle, i.e. libstdc++'s `rand() % (i+1)` swaps
// driven by the C library's rand() state, which nothing in Caffe seeds.  The same loop and
// the same rand() are used here, so a process with the same rand() history shuffles alike.
template <typename Dtype>
class HDF5DataLayer : public Layer<Dtype> {
 public:
  explicit HDF5DataLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  const char* type() const override { return "HDF5Data"; }
  int ExactNumBottomBlobs() const override { return 0; }
  int MinTopBlobs() const override { return 1; }

  void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    CHECK(!this->layer_param_.has_transform_param()) << this->type() << " does not transform data.";
    const string& source = this->layer_param_.hdf5_data_param().source();
    LOG_INFO << "Loading list of HDF5 filenames from: " << source;
    hdf_filenames_.clear();
    std::ifstream source_file(source.c_str());
    CHECK(source_file.is_open()) << "Failed to open source file: " << source;
    string line;
    while (source_file >> line) hdf_filenames_.push_back(line);
    source_file.close();
    num_files_ = (int)hdf_filenames_.size();
    current_file_ = 0;
    CHECK_GE(num_files_, 1) << "Must have at least 1 HDF5 filename listed in " << source;
    file_permutation_.resize(num_files_);
    for (int i = 0; i < num_files_; ++i) file_permutation_[i] = i;
    if (this->layer_param_.hdf5_data_param().shuffle()) RandomShuffle(&file_permutation_);
    LoadHDF5FileData(hdf_filenames_[file_permutation_[current_file_]].c_str());
    current_row_ = 0;
    const int batch_size = this->layer_param_.hdf5_data_param().batch_size();
    for (size_t i = 0; i < top.size(); ++i) {
      vector<int> top_shape = hdf_blobs_[i]->shape();
      top_shape[0] = batch_size;
      top[i]->Reshape(top_shape);
    }
  }
  void Reshape(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) override {}

 protected:
  // libstdc++'s std::random_shuffle(first, last) (removed in C++17), spelled out.
  static void RandomShuffle(vector<int>* v) {
    for (size_t i = 1; i < v->size(); ++i) {
      const size_t j = (size_t)std::rand() % (i + 1);
      if (i != j) std::swap((*v)[i], (*v)[j]);
    }
  }
  void UploadPermutation() {
    const int n = (int)data_permutation_.size();
    perm_mem_.reset(new SyncedMemory((size_t)n * sizeof(int)));
    std::memcpy(perm_mem_->mutable_cpu_data(), data_permutation_.data(), (size_t)n * sizeof(int));
  }
  void LoadHDF5FileData(const char* filename) {
    mms_h5::File file;
    string err;
    CHECK(file.Open(filename, &err)) << err;
    const int top_size = this->layer_param_.top_size();
    hdf_blobs_.resize(top_size);
    for (int i = 0; i < top_size; ++i) {
      const string& name = this->layer_param_.top(i);
      CHECK(file.Find(name)) << "Failed to find HDF5 dataset " << name;
      mms_h5::DatasetInfo info;
      std::vector<float> values;
      CHECK(file.ReadFloat(name, &info, &values, &err)) << "Failed to read float dataset " << name << ": " << err;
      CHECK_GE((int)info.dims.size(), 1) << "Input must have at least 1 axis.";
      vector<int> shape;
      for (int64_t d : info.dims) shape.push_back((int)d);
      hdf_blobs_[i].reset(new Blob<Dtype>(shape));
      std::memcpy(hdf_blobs_[i]->mutable_cpu_data(), values.data(), values.size() * sizeof(float));
    }
    const int num = hdf_blobs_[0]->shape(0);
    for (int i = 1; i < top_size; ++i) CHECK_EQ(hdf_blobs_[i]->shape(0), num);
    data_permutation_.resize(num);
    for (int i = 0; i < num; ++i) data_permutation_[i] = i;
    if (this->layer_param_.hdf5_data_param().shuffle()) RandomShuffle(&data_permutation_);
    UploadPermutation();
  }
  void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) override { NO_CPU_MODE; }
  void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override {}
  void Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) override {}

  // Same row/file walk as hdf5_data_layer.cpp:124-151, taken a run of rows at a time.
  void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) override {
    const int batch_size = this->layer_param_.hdf5_data_param().batch_size();
    const bool shuffle = this->layer_param_.hdf5_data_param().shuffle();
    int i = 0;
    while (i < batch_size) {
      if (current_row_ == hdf_blobs_[0]->shape(0)) {
        if (num_files_ > 1) {
          ++current_file_;
          if (current_file_ == num_files_) {
            current_file_ = 0;
            if (shuffle) RandomShuffle(&file_permutation_);
          }
          // launches reading the old file's buffers are on the null stream; hipFree in
          // ~SyncedMemory synchronises with them before the memory goes away
          LoadHDF5FileData(hdf_filenames_[file_permutation_[current_file_]].c_str());
        }
        current_row_ = 0;
        if (shuffle) { RandomShuffle(&data_permutation_); HIP_CHECK(hipDeviceSynchronize()); UploadPermutation(); }
      }
      const int rows = hdf_blobs_[0]->shape(0);
      CHECK_GT(rows, 0) << "HDF5 file with no rows";
      const int run = std::min(batch_size - i, rows - current_row_);
      for (size_t j = 0; j < top.size(); ++j) {
        const int data_dim = top[j]->count() / top[j]->shape(0);
        mms_check(mms_feed_gather_rows_f32(run, data_dim, rows, hdf_blobs_[j]->gpu_data(),
                                           shuffle ? static_cast<const int*>(perm_mem_->gpu_data()) : nullptr,
                                           current_row_, top[j]->mutable_gpu_data() + (size_t)i * data_dim, nullptr),
                  "mms_feed_gather_rows_f32");
      }
      i += run;
      current_row_ += run;
    }
  }

  vector<string> hdf_filenames_;
  int num_files_ = 0, current_file_ = 0, current_row_ = 0;
  vector<shared_ptr<Blob<Dtype> > > hdf_blobs_;
  vector<int> data_permutation_, file_permutation_;
  std::unique_ptr<SyncedMemory> perm_mem_;
};
INSTANTIATE_CLASS(HDF5DataLayer);
REGISTER_LAYER_CLASS(HDF5Data);

// ============================ prototxt (text format) subset ==================
namespace {
struct Tok {
  enum Kind { END, IDENT, STRING, NUMBER, LBRACE, RBRACE, COLON } kind;
  string text;
};
class Lexer {
 public:
  explicit Lexer(const string& s) : s_(s) {}
  bool next(Tok* t, string* err) {
    while (i_ < s_.size()) {
      const char c = s_[i_];
      if (std::isspace((unsigned char)c) || c == ',' || c == ';') { ++i_; continue; }
      if (c == '#') { while (i_ < s_.size() && s_[i_] != '\n') ++i_; continue; }
      break;
    }
    if (i_ >= s_.size()) { t->kind = Tok::END; return true; }
    const char c = s_[i_];
    if (c == '{' || c == '<') { ++i_; t->kind = Tok::LBRACE; return true; }
    if (c == '}' || c == '>') { ++i_; t->kind = Tok::RBRACE; return true; }
    if (c == ':') { ++i_; t->kind = Tok::COLON; return true; }
    if (c == '"' || c == '\'') {
      const char q = c;
      size_t j = ++i_;
      string out;
      while (j < s_.size() && s_[j] != q) {
        if (s_[j] == '\\' && j + 1 < s_.size()) ++j;
        out += s_[j++];
      }
      if (j >= s_.size()) { *err = "unterminated string"; return false; }
      i_ = j + 1;
      t->kind = Tok::STRING; t->text = out;
      return true;
    }
    if (std::isalpha((unsigned char)c) || c == '_') {
      size_t j = i_;
      while (j < s_.size() && (std::isalnum((unsigned char)s_[j]) || s_[j] == '_')) ++j;
      t->kind = Tok::IDENT; t->text = s_.substr(i_, j - i_); i_ = j;
      return true;
    }
    if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
      size_t j = i_ + 1;
      while (j < s_.size() && (std::isalnum((unsigned char)s_[j]) || s_[j] == '.' || s_[j] == '-' || s_[j] == '+')) ++j;
      t->kind = Tok::NUMBER; t->text = s_.substr(i_, j - i_); i_ = j;
      return true;
    }
    *err = string("unexpected character '") + c + "'";
    return false;
  }
 private:
  const string& s_;
  size_t i_ = 0;
};

class Parser {
 public:
  explicit Parser(const string& s) : lex_(s) {}
  bool fail(const string& m) { if (err_.empty()) err_ = m; return false; }
  const string& err() const { return err_; }
  bool advance() { return lex_.next(&cur_, &err_); }
  Tok cur_;

  // after a field name: scalar value
  bool scalar(Tok* v) {
    if (cur_.kind != Tok::COLON) return fail("expected ':'");
    if (!advance()) return false;
    if (cur_.kind != Tok::STRING && cur_.kind != Tok::NUMBER && cur_.kind != Tok::IDENT) return fail("expected a value");
    *v = cur_;
    return advance();
  }
  bool as_float(const Tok& t, float* f) {
    char* end = nullptr;
    *f = std::strtof(t.text.c_str(), &end);
    if (t.kind != Tok::NUMBER || end == t.text.c_str()) return fail("expected a number, got '" + t.text + "'");
    return true;
  }
  bool as_int(const Tok& t, int* i) {
    char* end = nullptr;
    const long v = std::strtol(t.text.c_str(), &end, 0);
    if (t.kind != Tok::NUMBER || *end != '\0') return fail("expected an integer, got '" + t.text + "'");
    *i = (int)v;
    return true;
  }
  bool as_bool(const Tok& t, bool* b) {
    if (t.text == "true" || t.text == "1") { *b = true; return true; }
    if (t.text == "false" || t.text == "0") { *b = false; return true; }
    return fail("expected a bool, got '" + t.text + "'");
  }
  // after a field name: `{ ... }` or `: { ... }`, body handled by f(field_name)
  template <typename F>
  bool message(F f) {
    if (cur_.kind == Tok::COLON && !advance()) return false;
    if (cur_.kind != Tok::LBRACE) return fail("expected '{'");
    if (!advance()) return false;
    while (cur_.kind == Tok::IDENT) {
      const string name = cur_.text;
      if (!advance()) return false;
      if (!f(name)) return fail("unknown or malformed field '" + name + "'");
    }
    if (cur_.kind != Tok::RBRACE) return fail("expected '}'");
    return advance();
  }
  bool skip_message() { return message([&](const string&) { return skip_value(); }); }
  bool skip_value() {
    if (cur_.kind == Tok::LBRACE) return skip_message();
    if (cur_.kind != Tok::COLON) return fail("expected ':' or '{'");
    if (!advance()) return false;
    if (cur_.kind == Tok::LBRACE) return skip_message();
    return advance();
  }
  bool filler(FillerParameter* fp) {
    return message([&](const string& n) {
      Tok v;
      if (!scalar(&v)) return false;
      if (n == "type") { fp->type_ = v.text; return true; }
      if (n == "value") return as_float(v, &fp->value_);
      if (n == "min") return as_float(v, &fp->min_);
      if (n == "max") return as_float(v, &fp->max_);
      if (n == "mean") return as_float(v, &fp->mean_);
      if (n == "std") return as_float(v, &fp->std_);
      if (n == "sparse" || n == "variance_norm") return true;
      return false;
    });
  }
  bool layer_field(const string& n, LayerParameter* lp) {
    Tok v;
    if (n == "name") { if (!scalar(&v)) return false; lp->name_ = v.text; return true; }
    if (n == "type") { if (!scalar(&v)) return false; lp->type_ = v.text; return true; }
    if (n == "bottom") { if (!scalar(&v)) return false; lp->bottom_.push_back(v.text); return true; }
    if (n == "top") { if (!scalar(&v)) return false; lp->top_.push_back(v.text); return true; }
    if (n == "loss_weight") { float f; if (!scalar(&v) || !as_float(v, &f)) return false; lp->loss_weight_.push_back(f); return true; }
    if (n == "propagate_down") { return scalar(&v); }
    if (n == "phase") { if (!scalar(&v)) return false; lp->phase_ = (v.text == "TEST" || v.text == "1") ? 1 : 0; return true; }
    if (n == "include" || n == "exclude") {
      vector<int>* dst = n == "include" ? &lp->include_phase_ : &lp->exclude_phase_;
      bool saw_phase = false;
      if (!message([&](const string& m) {
            if (m == "phase") {
              Tok w;
              if (!scalar(&w)) return false;
              dst->push_back((w.text == "TEST" || w.text == "1") ? 1 : 0);
              saw_phase = true;
              return true;
            }
            return skip_value();                      // min_level / max_level / stage / not_stage: not interpreted
          })) return false;
      if (!saw_phase) dst->push_back(-1);              // a rule without a phase matches every phase
      return true;
    }
    if (n == "param") {
      ParamSpec ps;
      if (!message([&](const string& m) {
            Tok w;
            if (!scalar(&w)) return false;
            if (m == "name") { ps.name = w.text; return true; }
            if (m == "lr_mult") return as_float(w, &ps.lr_mult);
            if (m == "decay_mult") return as_float(w, &ps.decay_mult);
            if (m == "share_mode") return true;
            return false;
          })) return false;
      lp->param_.push_back(ps);
      return true;
    }
    if (n == "sim_cross_param") {
      SimCrossParameter* p = &lp->sim_cross_param_;
      return message([&](const string& m) {
        if (m == "weight_filler") return filler(&p->weight_filler_);
        if (m == "bias_filler") return filler(&p->bias_filler_);
        Tok w;
        if (!scalar(&w)) return false;
        if (m == "dist_mode") return as_int(w, &p->dist_mode_);
        if (m == "mesure_count") return as_int(w, &p->mesure_count_);
        if (m == "bias_term") return as_bool(w, &p->bias_term_);
        return false;
      });
    }
    if (n == "sim_matrix_param") {
      return message([&](const string& m) {
        if (m == "weight_filler") return filler(&lp->sim_matrix_param_.weight_filler_);
        return false;
      });
    }
    if (n == "embed_param") {
      EmbedParameter* p = &lp->embed_param_;
      return message([&](const string& m) {
        if (m == "weight_filler") return filler(&p->weight_filler_);
        if (m == "bias_filler") return filler(&p->bias_filler_);
        Tok w;
        if (!scalar(&w)) return false;
        if (m == "num_output") return as_int(w, &p->num_output_);
        if (m == "input_dim") return as_int(w, &p->input_dim_);
        if (m == "bias_term") return as_bool(w, &p->bias_term_);
        if (m == "weight_source") { p->weight_source_ = w.text; return true; }
        return false;
      });
    }
    if (n == "hdf5_data_param") {
      HDF5DataParameter* p = &lp->hdf5_data_param_;
      return message([&](const string& m) {
        Tok w;
        if (!scalar(&w)) return false;
        if (m == "source") { p->source_ = w.text; return true; }
        if (m == "batch_size") return as_int(w, &p->batch_size_);
        if (m == "shuffle") return as_bool(w, &p->shuffle_);
        return false;
      });
    }
    if (n == "transform_param") { lp->has_transform_param_ = true; return skip_value(); }
    if (n == "map_param" || n == "mrr_param") {
      int* fa = n == "map_param" ? &lp->map_param_.fixed_axis_ : &lp->mrr_param_.fixed_axis_;
      return message([&](const string& m) {
        Tok w;
        if (!scalar(&w)) return false;
        if (m == "fixed_axis") return as_int(w, fa);
        return false;
      });
    }
    if (n == "auc_param") {
      AUCParameter* p = &lp->auc_param_;
      return message([&](const string& m) {
        Tok w;
        if (!scalar(&w)) return false;
        if (m == "fixed_axis") return as_int(w, &p->fixed_axis_);
        if (m == "axis") return as_int(w, &p->axis_);
        if (m == "ignore_label") { p->has_ignore_label_ = true; return as_int(w, &p->ignore_label_); }
        return false;
      });
    }
    if (n == "pair_rank_loss_param") {
      return message([&](const string& m) {
        Tok w;
        if (!scalar(&w)) return false;
        if (m == "margin") return as_float(w, &lp->pair_rank_loss_param_.margin_);
        return false;
      });
    }
    // parameter messages of layer types this library does not implement (whole-net files only)
    if (tolerant_ && n.size() > 6 && n.compare(n.size() - 6, 6, "_param") == 0) {
      lp->skipped_fields_.push_back(n);
      return skip_value();
    }
    if (tolerant_ && (n == "blobs" || n == "blobs_lr" || n == "weight_decay")) return skip_value();   // V1 leftovers
    return false;
  }
  bool tolerant_ = false;
 private:
  Lexer lex_;
  string err_;
};
}  // namespace

bool ReadNetParameterFromText(const string& text, NetParameter* out, string* err) {
  Parser p(text);
  p.tolerant_ = true;
  *out = NetParameter();
  bool ok = p.advance();
  while (ok && p.cur_.kind == Tok::IDENT) {
    const string name = p.cur_.text;
    ok = p.advance();
    if (!ok) break;
    Tok v;
    if (name == "layer" || name == "layers") {
      LayerParameter lp;
      ok = p.message([&](const string& n) { return p.layer_field(n, &lp); });
      if (ok && lp.type_.empty()) ok = p.fail("layer '" + lp.name_ + "' has no type");
      if (ok) out->layer_.push_back(lp);
    } else if (name == "name") {
      ok = p.scalar(&v);
      if (ok) out->name_ = v.text;
    } else if (name == "input") {
      ok = p.scalar(&v);
      if (ok) out->input_.push_back(v.text);
    } else if (name == "input_shape") {
      vector<int> shape;
      ok = p.message([&](const string& m) {
        Tok w;
        int d;
        if (m != "dim" || !p.scalar(&w) || !p.as_int(w, &d)) return false;
        shape.push_back(d);
        return true;
      });
      if (ok) out->input_shape_.push_back(shape);
    } else if (name == "input_dim") {
      int d;
      ok = p.scalar(&v) && p.as_int(v, &d);
      if (ok) {
        if (out->input_shape_.empty() || out->input_shape_.back().size() == 4) out->input_shape_.push_back(vector<int>());
        out->input_shape_.back().push_back(d);
      }
    } else if (name == "force_backward") {
      ok = p.scalar(&v) && p.as_bool(v, &out->force_backward_);
    } else if (name == "state" || name == "debug_info") {
      ok = p.skip_value();
    } else {
      ok = p.fail("unknown net field '" + name + "'");
    }
  }
  if (ok && p.cur_.kind != Tok::END) ok = p.fail("trailing input after the net");
  if (!ok && err) *err = p.err().empty() ? "parse error" : p.err();
  return ok;
}

bool ReadLayerParameterFromText(const string& text, LayerParameter* out, string* err) {
  Parser p(text);
  *out = LayerParameter();
  bool ok = p.advance();
  if (ok && p.cur_.kind == Tok::IDENT && (p.cur_.text == "layer" || p.cur_.text == "layers")) {
    ok = p.advance() && p.message([&](const string& n) { return p.layer_field(n, out); });
  } else if (ok) {
    while (ok && p.cur_.kind == Tok::IDENT) {
      const string name = p.cur_.text;
      ok = p.advance() && (p.layer_field(name, out) || p.fail("unknown or malformed field '" + name + "'"));
    }
  }
  if (ok && p.cur_.kind != Tok::END) ok = p.fail("trailing input after the layer message");
  if (ok && out->type_.empty()) ok = p.fail("layer has no type");
  if (!ok && err) *err = p.err().empty() ? "parse error" : p.err();
  return ok;
}

}  // namespace caffe


// ================================ PathNet: the path's layers of a whole net =================================
// What Net::Init / ForwardFromTo / BackwardFromTo (src/caffe/net.cpp:40-270, 535-591) do for the layers of a
// generated net file that THIS library implements: phase filtering (net.cpp:272-330 StateMeetsRule, phase
// only), blobs wired by name with in-place tops sharing their bottom's blob (net.cpp:385-420), parameters
// shared by `param { name }` (net.cpp:450-530 AppendParam), layers run in file order.  Layers of other types
// are listed and skipped; a blob only they produce is an input the host fills before SetUp.
namespace caffe {
struct PathNetLayer {
  LayerParameter param;
  shared_ptr<Layer<float> > layer;        // null: the type is outside this library
  vector<Blob<float>*> bottom, top;
  vector<string> bottom_names, top_names;
  bool runnable = false;
  string why_not;                          // for a supported layer that cannot run: what is missing
};
class PathNet {
 public:
  PathNet(const NetParameter& np, int phase) : name_(np.name_), phase_(phase) {
    for (size_t i = 0; i < np.input_.size(); ++i) {
      Blob<float>* b = blob(np.input_[i], true);
      if (i < np.input_shape_.size() && !np.input_shape_[i].empty()) b->Reshape(np.input_shape_[i]);
    }
    for (const LayerParameter& lp : np.layer_) {
      if (!included(lp)) continue;
      PathNetLayer L;
      L.param = lp;
      for (const string& bn : lp.bottom_) { L.bottom.push_back(blob(bn, true)); L.bottom_names.push_back(bn); }
      for (const string& tn : lp.top_) { L.top.push_back(blob(tn, true)); L.top_names.push_back(tn); }   // same name = same blob (in place)
      if (LayerRegistry<float>::Registry().count(lp.type())) L.layer = LayerRegistry<float>::CreateLayer(lp);
      layers_.push_back(L);
    }
  }
  // SetUp of every supported layer whose bottoms have a shape, in file order; returns the number that can run
  int SetUp() {
    int n = 0;
    for (PathNetLayer& L : layers_) {
      L.runnable = false;
      L.why_not.clear();
      if (!L.layer) { L.why_not = "layer type '" + L.param.type() + "' is not implemented by this library"; continue; }
      for (size_t b = 0; b < L.bottom.size(); ++b)
        if (L.bottom[b]->count() == 0 && L.why_not.empty()) L.why_not = "bottom '" + L.bottom_names[b] + "' has no shape (fill it before SetUp)";
      if (!L.why_not.empty()) continue;
      // AutoTopBlobs (layer.hpp:67-74 / net.cpp:150-166): anonymous tops up to the layer's need
      if (L.layer->AutoTopBlobs()) {
        const int need = std::max(L.layer->MinTopBlobs(), L.layer->ExactNumTopBlobs());
        while ((int)L.top.size() < need) {
          owned_.emplace_back(new Blob<float>());
          L.top.push_back(owned_.back().get());
          L.top_names.push_back("(automatic)");
        }
      }
      L.layer->SetUp(L.bottom, L.top);
      share_params(L);
      L.runnable = true;
      ++n;
    }
    insert_splits();
    plan_fusion();
    return n;
  }
  // "fuse_embed_scoring" (forward-only nets: evaluation, do_trec_qa_clean.py:617-650): a SimCross layer whose two
  // bottoms come from Embed layers that read ONE table (shared by parameter name, as network_v4's do) and feed
  // nothing else is run straight from the word ids -- the (N, W, D) blobs are never written.
  bool SetOption(const string& key, int value) {
    if (key == "pair_embed") { pair_embed_ = value != 0; return true; }
    if (key != "fuse_embed_scoring") return false;
    fuse_embed_ = value != 0;
    plan_fusion();
    return true;
  }
  int num_fused() const { return (int)fused_.size(); }
  float Forward() {
    float loss = 0;
    pair_done_.clear();
    for (auto& kv : pair_state_) kv.second.ran = false;
    for (size_t i = 0; i < layers_.size(); ++i) {
      PathNetLayer& L = layers_[i];
      if (!L.runnable || skipped_.count(i)) continue;
      for (Split& sp : splits_)                                    // SplitLayer::Reshape: the tops share the bottom's data
        for (size_t c = 0; c < sp.consumers.size(); ++c)
          if (sp.consumers[c].first == i) { sp.alias[c]->ReshapeLike(*sp.orig); sp.alias[c]->ShareData(*sp.orig); }
      if (pair_done_.count(i)) continue;                 // the later Embed of a pair: written with the earlier one
      auto ep = embed_pairs_.find(i);
      if (ep != embed_pairs_.end() && pair_embed_) {
        // two Embed layers over one table: both gathers in one launch, the backward's inverted index built beside them
        PathNetLayer& Lj = layers_[ep->second];
        EmbedPairState& st = pair_state_[i];
        if (EmbedLayer<float>::PairForward(*static_cast<EmbedLayer<float>*>(Lj.layer.get()),
                                           *static_cast<EmbedLayer<float>*>(L.layer.get()), Lj.bottom, Lj.top, L.bottom, L.top,
                                           &st.index_ws, &st.indexed)) {
          st.ran = true;
          pair_done_.insert(ep->second);
          continue;
        }
        st.ran = false;
      }
      auto f = fused_.find(i);
      if (f != fused_.end()) {
        PathNetLayer& Eq = layers_[f->second.first];
        PathNetLayer& Ea = layers_[f->second.second];
        auto* sim = static_cast<SimCrossLayer<float>*>(L.layer.get());
        auto& eb = Eq.layer->blobs();
        if (sim->ForwardFromWordIds(*Eq.bottom[0], *Ea.bottom[0], *eb[0], eb.size() > 1 ? eb[1].get() : nullptr,
                                    L.bottom, L.top))
          continue;
        Eq.layer->Forward(Eq.bottom, Eq.top);            // no fused kernel for this geometry: the usual three
        Ea.layer->Forward(Ea.bottom, Ea.top);
      }
      loss += L.layer->Forward(L.bottom, L.top);
    }
    ran_fused_ = !fused_.empty();
    return loss;
  }
  void Backward() {
    CHECK(!ran_fused_) << "Backward on a net whose last Forward scored from word ids (fuse_embed_scoring): the "
                          "Embed tops were never written";
    for (size_t i = layers_.size(); i-- > 0;) {
      PathNetLayer& L = layers_[i];
      for (Split& sp : splits_) if (sp.producer == (int)i) split_backward(sp);     // every consumer has run by now
      if (!L.runnable) continue;
      if (!has_backward(L)) continue;
      if (pair_done_.count(i)) continue;                 // the later Embed of a pair: its Backward runs with the earlier one's,
      auto ep = embed_pairs_.find(i);                    // when every consumer of BOTH tops has run
      if (ep != embed_pairs_.end() && pair_state_[i].ran) {
        PathNetLayer& Lj = layers_[ep->second];
        EmbedLayer<float>::PairBackward(*static_cast<EmbedLayer<float>*>(Lj.layer.get()),
                                        *static_cast<EmbedLayer<float>*>(L.layer.get()), Lj.bottom, Lj.top, L.bottom, L.top,
                                        &pair_state_[i].index_ws, pair_state_[i].indexed);
        continue;
      }
      vector<bool> pd(L.bottom.size(), true);
      for (size_t b = 0; b < L.bottom.size(); ++b) pd[b] = propagates(L, b);
      L.layer->Backward(L.top, pd, L.bottom);
    }
    for (Split& sp : splits_) if (sp.producer < 0) split_backward(sp);             // net inputs
  }
  int num_splits() const { return (int)splits_.size(); }
  Blob<float>* find_blob(const string& n) { auto it = blobs_.find(n); return it == blobs_.end() ? nullptr : it->second; }
  vector<PathNetLayer>& layers() { return layers_; }
  const vector<string>& blob_names() const { return blob_order_; }
  const string& name() const { return name_; }
 private:
  bool included(const LayerParameter& lp) const {
    // net.cpp:272-296 FilterNet: no include rules = included unless an exclude rule matches; with include
    // rules, included iff one matches
    auto matches = [&](int rule) { return rule < 0 || rule == phase_; };
    if (lp.include_phase_.empty()) {
      for (int r : lp.exclude_phase_) if (matches(r)) return false;
      return true;
    }
    for (int r : lp.include_phase_) if (matches(r)) return true;
    return false;
  }
  Blob<float>* blob(const string& n, bool create) {
    auto it = blobs_.find(n);
    if (it != blobs_.end()) return it->second;
    if (!create) return nullptr;
    owned_.emplace_back(new Blob<float>());
    blobs_[n] = owned_.back().get();
    blob_order_.push_back(n);
    return owned_.back().get();
  }
  void share_params(PathNetLayer& L) {
    const string t = L.param.type();
    if (t == "HDF5Data") for (const string& tn : L.top_names) produced_by_data_.insert(tn);
    auto& blobs = L.layer->blobs();
    for (size_t i = 0; i < L.param.param_.size() && i < blobs.size(); ++i) {
      const string& pn = L.param.param_[i].name;
      if (pn.empty()) continue;
      auto it = shared_params_.find(pn);
      if (it == shared_params_.end()) { shared_params_[pn] = blobs[i]; continue; }
      CHECK_EQ(it->second->count(), blobs[i]->count()) << "Shared parameter '" << pn << "' has mismatched sizes";
      blobs[i] = it->second;                                         // ShareData + ShareDiff with the owner
    }
  }
  static bool has_backward(const PathNetLayer& L) {
    const string t = L.param.type();
    return !(t == "HDF5Data" || t == "MAP" || t == "MRR" || t == "AUC" || t == "RankAccuracy");
  }
  bool propagates(const PathNetLayer& L, size_t b) const {
    const string t = L.param.type();
    if (t == "Embed") return false;                                // indices (embed_layer.cpp:156)
    if (t == "PairRankLoss" && b == 2) return false;               // labels (pair_rank_loss_layer.cpp:58-61)
    if (produced_by_data_.count(L.bottom_names[b])) return false;
    return true;
  }
  // What Net::Init's InsertSplits does (src/caffe/util/insert_splits.cpp:13-88, net.cpp:57): a blob that feeds more
  // than one layer gets a Split layer, whose tops share its data and own their diffs, and whose Backward SUMS those
  // diffs into the blob's (split_layer.cpp:38-57).  Without it each consumer's Backward overwrites the blob's diff
  // and the last one wins.  Here: every consumer that propagates down into such a blob reads it through a blob of
  // its own (data shared, diff private) and the sum is taken -- in consumer order, like the reference -- right
  // before the producing layer's Backward.  This also gives SimMatrix's bottom[1] a diff no sibling touches: its
  // Forward parks Q.W there (sim_matrix_layer.cpp:58).
  struct Split {
    Blob<float>* orig;
    int producer;                                                  // layer index, -1: a net input
    vector<std::pair<size_t, size_t> > consumers;                  // (layer, bottom index), file order
    vector<Blob<float>*> alias;
  };
  void insert_splits() {
    if (!splits_.empty()) return;                                  // SetUp may run again; the wiring is done once
    for (const string& name : blob_order_) {
      Blob<float>* b = blobs_[name];
      Split sp;
      sp.orig = b;
      sp.producer = -1;
      bool in_place = false;
      for (size_t i = 0; i < layers_.size(); ++i) {
        const PathNetLayer& L = layers_[i];
        for (size_t k = 0; k < L.top.size(); ++k) if (L.top[k] == b) sp.producer = (int)i;
        if (!L.runnable || !has_backward(L)) continue;
        for (size_t k = 0; k < L.bottom.size(); ++k) {
          if (L.bottom[k] != b || !propagates(L, k)) continue;
          sp.consumers.emplace_back(i, k);
          for (Blob<float>* t : L.top) if (t == b) in_place = true;
        }
      }
      if (sp.consumers.size() < 2 || in_place || b->count() == 0) continue;
      for (auto& c : sp.consumers) {
        owned_.emplace_back(new Blob<float>());
        Blob<float>* a = owned_.back().get();
        a->ReshapeLike(*b);
        a->ShareData(*b);
        sp.alias.push_back(a);
        layers_[c.first].bottom[c.second] = a;
      }
      splits_.push_back(sp);
    }
  }
  void split_backward(Split& sp) {
    vector<const float*> tops;
    for (Blob<float>* a : sp.alias) tops.push_back(a->gpu_diff());
    mms_check(mms_split_backward_f32(sp.orig->count(), (int)tops.size(), tops.data(), sp.orig->mutable_gpu_diff(), nullptr),
              "mms_split_backward_f32");
  }
  // Pairs of Embed layers that read ONE table (and bias) by parameter name, the later one's word ids available when
  // the earlier one runs: forward as one launch, backward as one pass (EmbedLayer::PairForward / PairBackward; option
  // "pair_embed", on by default -- the results are the two layers' bits, so nothing observable changes but the time).
  void plan_embed_pairs() {
    embed_pairs_.clear();
    pair_state_.clear();
    std::set<size_t> taken;
    for (size_t i = 0; i < layers_.size(); ++i) {
      PathNetLayer& Li = layers_[i];
      if (!Li.runnable || Li.param.type() != "Embed" || taken.count(i) || Li.bottom.size() != 1 || Li.top.size() != 1) continue;
      for (size_t j = i + 1; j < layers_.size(); ++j) {
        PathNetLayer& Lj = layers_[j];
        if (!Lj.runnable || Lj.param.type() != "Embed" || taken.count(j) || Lj.bottom.size() != 1 || Lj.top.size() != 1) continue;
        auto &bi = Li.layer->blobs(), &bj = Lj.layer->blobs();
        if (bi.empty() || bi.size() != bj.size() || bi[0].get() != bj[0].get()) continue;
        if (bi.size() > 1 && bi[1].get() != bj[1].get()) continue;
        bool ready = true;                               // the later layer's ids must exist when the earlier layer runs
        for (size_t k = i; k < layers_.size() && ready; ++k)
          for (Blob<float>* t : layers_[k].top) if (t == Lj.bottom[0]) ready = false;
        if (!ready) continue;
        embed_pairs_[i] = j;
        pair_state_[i];
        taken.insert(i); taken.insert(j);
        break;
      }
    }
  }
  void plan_fusion() {
    plan_embed_pairs();
    fused_.clear();
    skipped_.clear();
    if (!fuse_embed_) return;
    auto producer = [&](size_t before, Blob<float>* b) -> int {
      for (size_t i = before; i-- > 0;)
        for (Blob<float>* t : layers_[i].top) if (t == b) return (int)i;
      return -1;
    };
    auto consumers = [&](Blob<float>* b) {
      int c = 0;
      for (const PathNetLayer& L : layers_) for (Blob<float>* x : L.bottom) if (x == b) ++c;
      return c;
    };
    for (size_t i = 0; i < layers_.size(); ++i) {
      PathNetLayer& L = layers_[i];
      if (!L.runnable || L.param.type() != "SimCross" || L.bottom.size() != 2) continue;
      const int eq = producer(i, L.bottom[0]), ea = producer(i, L.bottom[1]);
      if (eq < 0 || ea < 0 || eq == ea) continue;
      const PathNetLayer &Eq = layers_[eq], &Ea = layers_[ea];
      if (!Eq.runnable || !Ea.runnable || Eq.param.type() != "Embed" || Ea.param.type() != "Embed") continue;
      if (Eq.top.size() != 1 || Ea.top.size() != 1 || Eq.bottom.size() != 1 || Ea.bottom.size() != 1) continue;
      if (consumers(L.bottom[0]) != 1 || consumers(L.bottom[1]) != 1) continue;        // someone else reads q or a
      auto &bq = Eq.layer->blobs(), &ba = Ea.layer->blobs();
      if (bq.empty() || bq.size() != ba.size() || bq[0].get() != ba[0].get()) continue;  // two tables
      if (bq.size() > 1 && bq[1].get() != ba[1].get()) continue;                          // two biases
      fused_[i] = std::make_pair((size_t)eq, (size_t)ea);
      skipped_.insert((size_t)eq);
      skipped_.insert((size_t)ea);
    }
  }
  string name_;
  int phase_;
  bool fuse_embed_ = false, ran_fused_ = false, pair_embed_ = true;
  struct EmbedPairState { Blob<float> index_ws; bool indexed = false, ran = false; };
  std::map<size_t, size_t> embed_pairs_;                 // earlier Embed layer -> the later one over the same table
  std::map<size_t, EmbedPairState> pair_state_;
  std::set<size_t> pair_done_;                           // later layers of pairs that ran as pairs in this Forward
  std::map<size_t, std::pair<size_t, size_t> > fused_;   // SimCross layer -> its two Embed producers
  std::set<size_t> skipped_;
  vector<PathNetLayer> layers_;
  std::map<string, Blob<float>*> blobs_;
  vector<string> blob_order_;
  vector<std::unique_ptr<Blob<float> > > owned_;
  std::map<string, shared_ptr<Blob<float> > > shared_params_;
  std::set<string> produced_by_data_;
  vector<Split> splits_;
};
}  // namespace caffe

// ================================== C handle API =============================
struct mms_blob {
  caffe::Blob<float>* b;
  std::shared_ptr<caffe::Blob<float> > keep;  // set for borrowed parameter blobs
  bool owned;
};
struct mms_layer {
  std::shared_ptr<caffe::Layer<float> > l;
  std::vector<std::unique_ptr<mms_blob> > param_handles;
};
struct mms_net {
  std::unique_ptr<caffe::PathNet> net;
  std::map<std::string, std::unique_ptr<mms_blob> > blob_handles;
  std::vector<std::unique_ptr<mms_layer> > layer_handles;
};

namespace {
std::vector<caffe::Blob<float>*> unwrap(mms_blob_t* const* v, int n) {
  std::vector<caffe::Blob<float>*> out(n);
  for (int i = 0; i < n; ++i) out[i] = v[i]->b;
  return out;
}
std::vector<int> shape_vec(const int* shape, int n) { return std::vector<int>(shape, shape + n); }
}  // namespace

extern "C" {

mms_blob_t* mms_blob_create(const int* shape, int num_axes) {
  // no axes = the default constructor (count 0, nothing allocated: blob.hpp:26-27), as `new Blob<Dtype>()`
  mms_blob_t* h = new mms_blob{num_axes > 0 ? new caffe::Blob<float>(shape_vec(shape, num_axes)) : new caffe::Blob<float>(), nullptr, true};
  return h;
}
void mms_blob_destroy(mms_blob_t* b) {
  if (!b) return;
  if (b->owned) delete b->b;
  delete b;
}
void mms_blob_reshape(mms_blob_t* b, const int* shape, int num_axes) { b->b->Reshape(shape_vec(shape, num_axes)); }
int mms_blob_num_axes(const mms_blob_t* b) { return b->b->num_axes(); }
int mms_blob_shape(const mms_blob_t* b, int axis) { return b->b->shape(axis); }
int mms_blob_count(const mms_blob_t* b) { return b->b->count(); }
const float* mms_blob_cpu(mms_blob_t* b, int which) { return which ? b->b->cpu_diff() : b->b->cpu_data(); }
float* mms_blob_mutable_cpu(mms_blob_t* b, int which) { return which ? b->b->mutable_cpu_diff() : b->b->mutable_cpu_data(); }
const float* mms_blob_gpu(mms_blob_t* b, int which) { return which ? b->b->gpu_diff() : b->b->gpu_data(); }
float* mms_blob_mutable_gpu(mms_blob_t* b, int which) { return which ? b->b->mutable_gpu_diff() : b->b->mutable_gpu_data(); }

mms_layer_t* mms_layer_create(const char* prototxt, char* err, int err_len) {
  caffe::LayerParameter lp;
  std::string e;
  if (!caffe::ReadLayerParameterFromText(prototxt ? prototxt : "", &lp, &e)) {
    if (err && err_len > 0) std::snprintf(err, err_len, "%s", e.c_str());
    return nullptr;
  }
  mms_layer_t* h = new mms_layer;
  h->l = caffe::LayerRegistry<float>::CreateLayer(lp);
  return h;
}
void mms_layer_destroy(mms_layer_t* l) { delete l; }
const char* mms_layer_type(const mms_layer_t* l) { return l->l->type(); }
void mms_layer_setup(mms_layer_t* l, mms_blob_t* const* bottom, int nbottom, mms_blob_t* const* top, int ntop) {
  l->l->SetUp(unwrap(bottom, nbottom), unwrap(top, ntop));
}
float mms_layer_forward(mms_layer_t* l, mms_blob_t* const* bottom, int nbottom, mms_blob_t* const* top, int ntop) {
  return l->l->Forward(unwrap(bottom, nbottom), unwrap(top, ntop));
}
void mms_layer_backward(mms_layer_t* l, mms_blob_t* const* top, int ntop, const int* propagate_down,
                        mms_blob_t* const* bottom, int nbottom) {
  std::vector<bool> pd(nbottom);
  for (int i = 0; i < nbottom; ++i) pd[i] = propagate_down[i] != 0;
  l->l->Backward(unwrap(top, ntop), pd, unwrap(bottom, nbottom));
}
int mms_layer_num_param_blobs(mms_layer_t* l) { return (int)l->l->blobs().size(); }
mms_blob_t* mms_layer_param_blob(mms_layer_t* l, int i) {
  auto& blobs = l->l->blobs();
  if (i < 0 || i >= (int)blobs.size()) return nullptr;
  if ((int)l->param_handles.size() < (int)blobs.size()) l->param_handles.resize(blobs.size());
  if (!l->param_handles[i] || l->param_handles[i]->b != blobs[i].get())
    l->param_handles[i].reset(new mms_blob{blobs[i].get(), blobs[i], false});
  return l->param_handles[i].get();
}
void mms_layer_set_param_propagate_down(mms_layer_t* l, int i, int v) { l->l->set_param_propagate_down(i, v != 0); }
int mms_layer_set_option(mms_layer_t* l, const char* key, int value) { return (key && l->l->SetOption(key, value)) ? 0 : 1; }
// ---- whole nets ----
mms_net_t* mms_net_create(const char* prototxt, int phase, char* err, int err_len) {
  caffe::NetParameter np;
  std::string e;
  if (!caffe::ReadNetParameterFromText(prototxt ? prototxt : "", &np, &e)) {
    if (err && err_len > 0) std::snprintf(err, err_len, "%s", e.c_str());
    return nullptr;
  }
  mms_net_t* h = new mms_net;
  h->net.reset(new caffe::PathNet(np, phase ? 1 : 0));
  h->layer_handles.resize(h->net->layers().size());
  return h;
}
void mms_net_destroy(mms_net_t* n) { delete n; }
int mms_net_set_option(mms_net_t* n, const char* key, int value) { return (key && n->net->SetOption(key, value)) ? 0 : 1; }
int mms_net_num_fused(const mms_net_t* n) { return n->net->num_fused(); }
int mms_net_num_splits(const mms_net_t* n) { return n->net->num_splits(); }
const char* mms_net_name(const mms_net_t* n) { return n->net->name().c_str(); }
int mms_net_num_layers(const mms_net_t* n) { return (int)n->net->layers().size(); }
const char* mms_net_layer_name(const mms_net_t* n, int i) { return n->net->layers()[i].param.name().c_str(); }
const char* mms_net_layer_type(const mms_net_t* n, int i) { return n->net->layers()[i].param.type().c_str(); }
int mms_net_layer_supported(const mms_net_t* n, int i) { return n->net->layers()[i].layer ? 1 : 0; }
int mms_net_layer_runnable(const mms_net_t* n, int i) { return n->net->layers()[i].runnable ? 1 : 0; }
const char* mms_net_layer_why_not(const mms_net_t* n, int i) { return n->net->layers()[i].why_not.c_str(); }
mms_layer_t* mms_net_layer(mms_net_t* n, int i) {
  if (i < 0 || i >= (int)n->net->layers().size() || !n->net->layers()[i].layer) return nullptr;
  if (!n->layer_handles[i]) { n->layer_handles[i].reset(new mms_layer); n->layer_handles[i]->l = n->net->layers()[i].layer; }
  return n->layer_handles[i].get();
}
int mms_net_num_blobs(const mms_net_t* n) { return (int)n->net->blob_names().size(); }
const char* mms_net_blob_name(const mms_net_t* n, int i) { return n->net->blob_names()[i].c_str(); }
mms_blob_t* mms_net_blob(mms_net_t* n, const char* name) {
  caffe::Blob<float>* b = name ? n->net->find_blob(name) : nullptr;
  if (!b) return nullptr;
  auto& h = n->blob_handles[name];
  if (!h) h.reset(new mms_blob{b, nullptr, false});
  return h.get();
}
int mms_net_setup(mms_net_t* n) { return n->net->SetUp(); }
float mms_net_forward(mms_net_t* n) { return n->net->Forward(); }
void mms_net_backward(mms_net_t* n) { n->net->Backward(); }

void mms_caffe_set_mode(int gpu) { caffe::Caffe::set_mode(gpu ? caffe::Caffe::GPU : caffe::Caffe::CPU); }
void mms_caffe_set_random_seed(unsigned seed) { caffe::caffe_set_random_seed(seed); }
const char* mms_layer_registry_types(void) {
  static std::string s;
  s.clear();
  for (auto& t : caffe::LayerRegistry<float>::LayerTypeList()) s += (s.empty() ? "" : ",") + t;
  return s.c_str();
}

// ---- HDF5 files (host only) ----
struct mms_h5_file { mms_h5::File f; std::vector<std::string> names; };
static void set_err(char* err, int err_len, const std::string& m) {
  if (err && err_len > 0) std::snprintf(err, err_len, "%s", m.c_str());
}
mms_h5_file_t* mms_h5_open(const char* path, char* err, int err_len) {
  std::unique_ptr<mms_h5_file> h(new mms_h5_file);
  std::string e;
  if (!path || !h->f.Open(path, &e)) { set_err(err, err_len, e.empty() ? "null path" : e); return nullptr; }
  h->names = h->f.DatasetNames();
  return h.release();
}
void mms_h5_close(mms_h5_file_t* h) { delete h; }
int mms_h5_num_datasets(const mms_h5_file_t* h) { return (int)h->names.size(); }
const char* mms_h5_dataset_name(const mms_h5_file_t* h, int i) { return h->names[i].c_str(); }
int mms_h5_dataset_info(const mms_h5_file_t* h, const char* name, long long* dims, int max_axes,
                        int* type_class, int* elem_size, char* err, int err_len) {
  mms_h5::DatasetInfo info;
  std::string e;
  if (!h->f.Info(name, &info, &e)) { set_err(err, err_len, e); return -1; }
  for (int a = 0; a < (int)info.dims.size() && a < max_axes; ++a) dims[a] = info.dims[a];
  if (type_class) *type_class = info.type_class;
  if (elem_size) *elem_size = info.elem_size;
  return (int)info.dims.size();
}
int mms_h5_read_float(const mms_h5_file_t* h, const char* name, float* out, long long capacity,
                      char* err, int err_len) {
  mms_h5::DatasetInfo info;
  std::vector<float> v;
  std::string e;
  if (!h->f.ReadFloat(name, &info, &v, &e)) { set_err(err, err_len, e); return 1; }
  if ((long long)v.size() > capacity) { set_err(err, err_len, "output buffer too small"); return 2; }
  std::memcpy(out, v.data(), v.size() * sizeof(float));
  return 0;
}
struct mms_h5_writer { std::vector<mms_h5::WriteDataset> sets; };
mms_h5_writer_t* mms_h5_writer_create(void) { return new mms_h5_writer; }
void mms_h5_writer_destroy(mms_h5_writer_t* w) { delete w; }
void mms_h5_writer_add(mms_h5_writer_t* w, const char* name, const long long* dims, int num_axes,
                       int elem_size, const double* values) {
  mms_h5::WriteDataset d;
  d.name = name;
  d.elem_size = elem_size;
  long long c = 1;
  for (int a = 0; a < num_axes; ++a) { d.dims.push_back(dims[a]); c *= dims[a]; }
  d.values.assign(values, values + c);
  w->sets.push_back(std::move(d));
}
int mms_h5_writer_save(const mms_h5_writer_t* w, const char* path, char* err, int err_len) {
  std::string e;
  if (!mms_h5::WriteContiguous(path, w->sets, &e)) { set_err(err, err_len, e); return 1; }
  return 0;
}

// ---- Layer<double>: one-shot run (create by type string, SetUp, Forward, Backward) ----
// The handle API above is float; this single entry point drives the double instantiation of the
// three path layers end to end so that it can be checked against the oracle's double code.
int mms_layer_run_f64(const char* prototxt, int nbottom, const int* bottom_axes, const int* bottom_dims,
                      const double* const* bottom_data, int nparam, const double* const* param_data,
                      const double* top_diff, const int* propagate_down, double* top_out, long long top_capacity,
                      int* top_dims_out, int* top_axes_out, double* const* bottom_diff_out,
                      double* const* param_diff_out, char* err, int err_len) {
  caffe::LayerParameter lp;
  std::string e;
  if (!prototxt || !caffe::ReadLayerParameterFromText(prototxt, &lp, &e)) { set_err(err, err_len, e); return 1; }
  if (caffe::LayerRegistry<double>::Registry().count(lp.type()) == 0) {
    set_err(err, err_len, "no Layer<double> registered for type " + lp.type());
    return 2;
  }
  std::shared_ptr<caffe::Layer<double> > layer = caffe::LayerRegistry<double>::CreateLayer(lp);
  std::vector<std::unique_ptr<caffe::Blob<double> > > bots;
  std::vector<caffe::Blob<double>*> bottom, top;
  int at = 0;
  for (int b = 0; b < nbottom; ++b) {
    std::vector<int> shape(bottom_dims + at, bottom_dims + at + bottom_axes[b]);
    at += bottom_axes[b];
    bots.emplace_back(new caffe::Blob<double>(shape));
    std::memcpy(bots.back()->mutable_cpu_data(), bottom_data[b], sizeof(double) * bots.back()->count());
    bottom.push_back(bots.back().get());
  }
  caffe::Blob<double> top0;
  top.push_back(&top0);
  layer->SetUp(bottom, top);
  if ((int)layer->blobs().size() != nparam && nparam != 0) { set_err(err, err_len, "parameter blob count mismatch"); return 3; }
  for (int i = 0; i < nparam; ++i)
    if (param_data && param_data[i])
      std::memcpy(layer->blobs()[i]->mutable_cpu_data(), param_data[i], sizeof(double) * layer->blobs()[i]->count());
  layer->Forward(bottom, top);
  if (top0.count() > top_capacity) { set_err(err, err_len, "top buffer too small"); return 4; }
  std::memcpy(top_out, top0.cpu_data(), sizeof(double) * top0.count());
  *top_axes_out = top0.num_axes();
  for (int a = 0; a < top0.num_axes() && a < 8; ++a) top_dims_out[a] = top0.shape(a);
  if (top_diff) std::memcpy(top0.mutable_cpu_diff(), top_diff, sizeof(double) * top0.count());
  else top0.mutable_cpu_diff()[0] = 1.0;                        // a loss layer's loss_weight
  std::vector<bool> pd(nbottom);
  for (int b = 0; b < nbottom; ++b) pd[b] = propagate_down ? propagate_down[b] != 0 : true;
  for (int i = 0; i < nparam; ++i)                                // start the parameter diffs from the caller's values
    if (param_diff_out && param_diff_out[i])
      std::memcpy(layer->blobs()[i]->mutable_cpu_diff(), param_diff_out[i], sizeof(double) * layer->blobs()[i]->count());
  layer->Backward(top, pd, bottom);
  for (int b = 0; b < nbottom; ++b)
    if (bottom_diff_out && bottom_diff_out[b])
      std::memcpy(bottom_diff_out[b], bottom[b]->cpu_diff(), sizeof(double) * bottom[b]->count());
  for (int i = 0; i < nparam; ++i)
    if (param_diff_out && param_diff_out[i])
      std::memcpy(param_diff_out[i], layer->blobs()[i]->cpu_diff(), sizeof(double) * layer->blobs()[i]->count());
  return 0;
}

}  // extern "C"
