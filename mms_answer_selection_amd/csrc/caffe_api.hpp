// csrc/caffe_api.hpp -- host-side mirror of the slice of the Caffe plugin
// interface the three MMS layers sit behind, restated minimally for HIP.
//
// What it mirrors (reference file:line, include/caffe/...):
//   Blob<Dtype>          blob.hpp:24-277   shape/count/legacy accessors, cpu/gpu data+diff
//   SyncedMemory         syncedmem.hpp:62  UNINITIALIZED / HEAD_AT_CPU / HEAD_AT_GPU / SYNCED
//   Layer<Dtype>         layer.hpp:32-445  SetUp = CheckBlobCounts, LayerSetUp, Reshape,
//                                          SetLossWeights (:67-74); Forward/Backward wrappers
//                                          dispatching on Caffe::mode() (:451-503)
//   LayerRegistry        layer_factory.hpp:56-137   type string -> creator
//   LayerParameter & co  src/caffe/proto/caffe.proto:310-416, 430-432, 471-481 (subset,
//                        plain structs with protobuf-style accessors; no protoc here)
//   Filler               filler.hpp   constant / uniform / gaussian / xavier
//   CHECK / LOG(FATAL)   glog semantics: print and abort()  (device_alternate.hpp:48-76)
// Same names, argument meaning and error behaviour as the reference so a
// layer body written against Caffe compiles against this header unchanged.
// Everything else of Caffe (Net, Solver, data layers, ...) is out of scope.
#ifndef MMS_CAFFE_API_HPP_
#define MMS_CAFFE_API_HPP_

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace caffe {

using std::shared_ptr;
using std::string;
using std::vector;

// ------------------------------- glog-style checks --------------------------
class FatalStream {
 public:
  FatalStream(const char* file, int line, const char* what) {
    os_ << "F " << file << ":" << line << "] " << what;
  }
  [[noreturn]] ~FatalStream() {
    std::fprintf(stderr, "%s\n*** Check failure: aborting (Caffe semantics) ***\n", os_.str().c_str());
    std::fflush(stderr);
    std::abort();
  }
  template <typename T>
  FatalStream& operator<<(const T& v) { os_ << v; return *this; }
 private:
  std::ostringstream os_;
};
struct NullStream {
  template <typename T>
  NullStream& operator<<(const T&) { return *this; }
};
#define MMS_FATAL(what) ::caffe::FatalStream(__FILE__, __LINE__, what)
#define CHECK(c) if (!(c)) MMS_FATAL("Check failed: " #c " ")
#define CHECK_OP(a, b, op) if (!((a)op(b))) MMS_FATAL("Check failed: " #a " " #op " " #b " ") << "(" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP(a, b, !=)
#define CHECK_LE(a, b) CHECK_OP(a, b, <=)
#define CHECK_LT(a, b) CHECK_OP(a, b, <)
#define CHECK_GE(a, b) CHECK_OP(a, b, >=)
#define CHECK_GT(a, b) CHECK_OP(a, b, >)
#define LOG_FATAL MMS_FATAL("")
#define LOG_INFO ::caffe::NullStream()
#define LOG(severity) LOG_##severity   // glog spelling: LOG(FATAL) << ..., LOG(INFO) << ...
#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) MMS_FATAL("HIP: ") << hipGetErrorString(e_) << " in " #expr; } while (0)
#define NOT_IMPLEMENTED MMS_FATAL("Not Implemented Yet")

// ------------------------------- Caffe singleton -----------------------------
class Caffe {
 public:
  enum Brew { CPU, GPU };
  static Brew mode() { return mode_(); }
  static void set_mode(Brew m) { mode_() = m; }
 private:
  static Brew& mode_() { static thread_local Brew m = GPU; return m; }
};

// --------------------------------- SyncedMemory ------------------------------
class SyncedMemory {
 public:
  enum SyncedHead { UNINITIALIZED, HEAD_AT_CPU, HEAD_AT_GPU, SYNCED };
  explicit SyncedMemory(size_t size) : size_(size) {}
  ~SyncedMemory() {
    if (cpu_ptr_) (void)hipHostFree(cpu_ptr_);
    if (gpu_ptr_) (void)hipFree(gpu_ptr_);
  }
  SyncedMemory(const SyncedMemory&) = delete;
  SyncedMemory& operator=(const SyncedMemory&) = delete;
  const void* cpu_data() { to_cpu(); return cpu_ptr_; }
  const void* gpu_data() { to_gpu(); return gpu_ptr_; }
  void* mutable_cpu_data() { to_cpu(); head_ = HEAD_AT_CPU; return cpu_ptr_; }
  void* mutable_gpu_data() { to_gpu(); head_ = HEAD_AT_GPU; return gpu_ptr_; }
  SyncedHead head() const { return head_; }
  size_t size() const { return size_; }

 private:
  void alloc_cpu() {  // pinned, zero-filled on first touch (syncedmem.cpp:28-29)
    if (!cpu_ptr_) { HIP_CHECK(hipHostMalloc(&cpu_ptr_, size_ ? size_ : 1)); std::memset(cpu_ptr_, 0, size_); }
  }
  void alloc_gpu() {
    if (!gpu_ptr_) { HIP_CHECK(hipMalloc(&gpu_ptr_, size_ ? size_ : 1)); HIP_CHECK(hipMemset(gpu_ptr_, 0, size_)); }
  }
  void to_cpu() {
    switch (head_) {
      case UNINITIALIZED: alloc_cpu(); head_ = HEAD_AT_CPU; break;
      case HEAD_AT_GPU:
        alloc_cpu();
        HIP_CHECK(hipMemcpy(cpu_ptr_, gpu_ptr_, size_, hipMemcpyDeviceToHost));  // synchronous, like syncedmem.cpp:39
        head_ = SYNCED;
        break;
      default: break;
    }
  }
  void to_gpu() {
    switch (head_) {
      case UNINITIALIZED: alloc_gpu(); head_ = HEAD_AT_GPU; break;
      case HEAD_AT_CPU:
        alloc_gpu();
        HIP_CHECK(hipMemcpy(gpu_ptr_, cpu_ptr_, size_, hipMemcpyHostToDevice));
        head_ = SYNCED;
        break;
      default: break;
    }
  }
  void* cpu_ptr_ = nullptr;
  void* gpu_ptr_ = nullptr;
  size_t size_;
  SyncedHead head_ = UNINITIALIZED;
};

// ------------------------------------- Blob ----------------------------------
constexpr int kMaxBlobAxes = 32;

template <typename Dtype>
class Blob {
 public:
  Blob() {}
  explicit Blob(const vector<int>& shape) { Reshape(shape); }
  Blob(int num, int channels, int height, int width) { Reshape(num, channels, height, width); }
  Blob(const Blob&) = delete;
  Blob& operator=(const Blob&) = delete;

  void Reshape(int num, int channels, int height, int width) {
    Reshape(vector<int>{num, channels, height, width});
  }
  // Never shrinks the allocation (blob.cpp:23-45).
  void Reshape(const vector<int>& shape) {
    CHECK_LE(shape.size(), (size_t)kMaxBlobAxes);
    count_ = 1;
    shape_.resize(shape.size());
    for (size_t i = 0; i < shape.size(); ++i) {
      CHECK_GE(shape[i], 0);
      if (count_ != 0) CHECK_LE(shape[i], 0x7fffffff / count_) << "blob size exceeds INT_MAX";
      count_ *= shape[i];
      shape_[i] = shape[i];
    }
    if (count_ > capacity_) {
      capacity_ = count_;
      data_.reset(new SyncedMemory(capacity_ * sizeof(Dtype)));
      diff_.reset(new SyncedMemory(capacity_ * sizeof(Dtype)));
    }
  }
  void ReshapeLike(const Blob& other) { Reshape(other.shape()); }

  const vector<int>& shape() const { return shape_; }
  int shape(int index) const { return shape_[CanonicalAxisIndex(index)]; }
  int num_axes() const { return (int)shape_.size(); }
  int count() const { return count_; }
  int count(int start_axis, int end_axis) const {
    CHECK_LE(start_axis, end_axis);
    CHECK_GE(start_axis, 0);
    CHECK_LE(end_axis, num_axes());
    int c = 1;
    for (int i = start_axis; i < end_axis; ++i) c *= shape_[i];
    return c;
  }
  int count(int start_axis) const { return count(start_axis, num_axes()); }
  int CanonicalAxisIndex(int axis_index) const {
    CHECK_GE(axis_index, -num_axes());
    CHECK_LT(axis_index, num_axes());
    return axis_index < 0 ? axis_index + num_axes() : axis_index;
  }
  // Legacy 4-D accessors: valid for <= 4 axes, missing axes read as 1
  // (blob.hpp:132-151).  SimCross relies on height() of a 3-axis blob being D.
  int num() const { return LegacyShape(0); }
  int channels() const { return LegacyShape(1); }
  int height() const { return LegacyShape(2); }
  int width() const { return LegacyShape(3); }
  int LegacyShape(int index) const {
    CHECK_LE(num_axes(), 4) << "Cannot use legacy accessors on Blobs with > 4 axes.";
    CHECK_LT(index, 4);
    CHECK_GE(index, -4);
    if (index >= num_axes() || index < -num_axes()) return 1;
    return shape(index);
  }

  const Dtype* cpu_data() const { CHECK(data_); return (const Dtype*)data_->cpu_data(); }
  const Dtype* gpu_data() const { CHECK(data_); return (const Dtype*)data_->gpu_data(); }
  const Dtype* cpu_diff() const { CHECK(diff_); return (const Dtype*)diff_->cpu_data(); }
  const Dtype* gpu_diff() const { CHECK(diff_); return (const Dtype*)diff_->gpu_data(); }
  Dtype* mutable_cpu_data() { CHECK(data_); return (Dtype*)data_->mutable_cpu_data(); }
  Dtype* mutable_gpu_data() { CHECK(data_); return (Dtype*)data_->mutable_gpu_data(); }
  Dtype* mutable_cpu_diff() { CHECK(diff_); return (Dtype*)diff_->mutable_cpu_data(); }
  Dtype* mutable_gpu_diff() { CHECK(diff_); return (Dtype*)diff_->mutable_gpu_data(); }
  const shared_ptr<SyncedMemory>& data() const { return data_; }
  const shared_ptr<SyncedMemory>& diff() const { return diff_; }
  // blob.cpp:148-158: this blob's data (diff) becomes the other's -- what SplitLayer::Reshape does to its tops
  void ShareData(const Blob& other) { CHECK_EQ(count_, other.count()); data_ = other.data(); }
  void ShareDiff(const Blob& other) { CHECK_EQ(count_, other.count()); diff_ = other.diff(); }

 private:
  shared_ptr<SyncedMemory> data_, diff_;
  vector<int> shape_;
  int count_ = 0;
  int capacity_ = 0;
};

// ------------------------------- parameter messages --------------------------
// Plain structs with the accessor names protoc would generate.
struct FillerParameter {  // caffe.proto:43-61
  string type_ = "constant";
  float value_ = 0, min_ = 0, max_ = 1, mean_ = 0, std_ = 1;
  const string& type() const { return type_; }
  float value() const { return value_; }
  float min() const { return min_; }
  float max() const { return max_; }
  float mean() const { return mean_; }
  float std() const { return std_; }
};
struct SimCrossParameter {  // caffe.proto:471-477
  int dist_mode_ = 1;
  int mesure_count_ = 1;  // sic: the reference's spelling
  bool bias_term_ = true;
  FillerParameter weight_filler_, bias_filler_;
  int dist_mode() const { return dist_mode_; }
  int mesure_count() const { return mesure_count_; }
  bool bias_term() const { return bias_term_; }
  const FillerParameter& weight_filler() const { return weight_filler_; }
  const FillerParameter& bias_filler() const { return bias_filler_; }
};
struct SimMatrixParameter {  // caffe.proto:430-432
  FillerParameter weight_filler_;
  const FillerParameter& weight_filler() const { return weight_filler_; }
};
struct PairRankLossParameter {  // caffe.proto:479-481
  float margin_ = 1.0f;
  float margin() const { return margin_; }
};
struct EmbedParameter {  // caffe.proto:788-802 (+ the fork's weight_source, :801)
  int num_output_ = 0, input_dim_ = 0;
  bool bias_term_ = true;
  FillerParameter weight_filler_, bias_filler_;
  string weight_source_;
  int num_output() const { return num_output_; }
  int input_dim() const { return input_dim_; }
  bool bias_term() const { return bias_term_; }
  const FillerParameter& weight_filler() const { return weight_filler_; }
  const FillerParameter& bias_filler() const { return bias_filler_; }
  const string& weight_source() const { return weight_source_; }
};
struct MAPParameter {  // caffe.proto:422-424
  int fixed_axis_ = 1;
  int fixed_axis() const { return fixed_axis_; }
};
struct MRRParameter {  // caffe.proto:426-428
  int fixed_axis_ = 1;
  int fixed_axis() const { return fixed_axis_; }
};
struct AUCParameter {  // caffe.proto:465-469
  int fixed_axis_ = 1, axis_ = 1, ignore_label_ = 0;
  bool has_ignore_label_ = false;
  int fixed_axis() const { return fixed_axis_; }
  int axis() const { return axis_; }
  bool has_ignore_label() const { return has_ignore_label_; }
  int ignore_label() const { return ignore_label_; }
};
struct HDF5DataParameter {  // caffe.proto:827-839
  string source_;
  int batch_size_ = 0;
  bool shuffle_ = false;
  const string& source() const { return source_; }
  int batch_size() const { return batch_size_; }
  bool shuffle() const { return shuffle_; }
};
struct ParamSpec {  // caffe.proto:281-308 (subset)
  string name;
  float lr_mult = 1, decay_mult = 1;
};
struct LayerParameter {  // caffe.proto:310-416 (subset)
  string name_, type_;
  vector<string> bottom_, top_;
  vector<float> loss_weight_;
  vector<ParamSpec> param_;
  SimCrossParameter sim_cross_param_;
  SimMatrixParameter sim_matrix_param_;
  PairRankLossParameter pair_rank_loss_param_;
  EmbedParameter embed_param_;
  const EmbedParameter& embed_param() const { return embed_param_; }
  HDF5DataParameter hdf5_data_param_;
  const HDF5DataParameter& hdf5_data_param() const { return hdf5_data_param_; }
  bool has_transform_param_ = false;
  bool has_transform_param() const { return has_transform_param_; }
  const string& top(int i) const { return top_[i]; }
  MAPParameter map_param_;
  MRRParameter mrr_param_;
  AUCParameter auc_param_;
  const MAPParameter& map_param() const { return map_param_; }
  const MRRParameter& mrr_param() const { return mrr_param_; }
  const AUCParameter& auc_param() const { return auc_param_; }
  const string& name() const { return name_; }
  const string& type() const { return type_; }
  int bottom_size() const { return (int)bottom_.size(); }
  int top_size() const { return (int)top_.size(); }
  int loss_weight_size() const { return (int)loss_weight_.size(); }
  float loss_weight(int i) const { return loss_weight_[i]; }
  void add_loss_weight(float w) { loss_weight_.push_back(w); }
  const SimCrossParameter& sim_cross_param() const { return sim_cross_param_; }
  const SimMatrixParameter& sim_matrix_param() const { return sim_matrix_param_; }
  const PairRankLossParameter& pair_rank_loss_param() const { return pair_rank_loss_param_; }
  // caffe.proto:326-341: phase / include / exclude (NetStateRule: only its `phase` is interpreted here)
  int phase_ = -1;                       // -1 unset, 0 TRAIN, 1 TEST
  vector<int> include_phase_, exclude_phase_;
  vector<string> skipped_fields_;        // *_param messages of layer types this library does not implement
};

// caffe.proto:63-110 (subset): what the driver's generated net files hold
struct NetParameter {
  string name_;
  vector<LayerParameter> layer_;
  vector<string> input_;
  vector<vector<int> > input_shape_;
  bool force_backward_ = false;
};
// Reads a whole net in protobuf text format (`name: ... layer { ... } layer { ... } ...`; the V1 spelling
// `layers { ... }` is accepted for the fields it shares).  Parameter messages of layer types outside this
// library (convolution_param, pooling_param, ...) are skipped and recorded, not rejected: the net file
// python/caffe/net_spec.py writes for network_v4 (examples/trec_qa_w2v_mms/do_trec_qa_clean.py:608-615)
// must load unmodified.
bool ReadNetParameterFromText(const string& text, NetParameter* out, string* err);

// Reads ONE `layer { ... }` message (or its body) in protobuf text format.
// Unknown fields are a fatal error, as with protobuf's TextFormat.
bool ReadLayerParameterFromText(const string& text, LayerParameter* out, string* err);

// ------------------------------------ Filler ---------------------------------
std::mt19937& caffe_rng();
void caffe_set_random_seed(unsigned seed);

template <typename Dtype>
class Filler {
 public:
  explicit Filler(const FillerParameter& p) : filler_param_(p) {}
  virtual ~Filler() {}
  virtual void Fill(Blob<Dtype>* blob) = 0;
 protected:
  FillerParameter filler_param_;
};
template <typename Dtype>
Filler<Dtype>* GetFiller(const FillerParameter& param);

// ------------------------------------- Layer ---------------------------------
// include/caffe/util/math_functions.hpp:152 (caffe_gpu_dot): x . y on the device, result on the host
float caffe_gpu_dot(int n, const float* x, const float* y);
double caffe_gpu_dot(int n, const double* x, const double* y);

template <typename Dtype>
class Layer {
 public:
  explicit Layer(const LayerParameter& param) : layer_param_(param) {}
  virtual ~Layer() {}

  void SetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CheckBlobCounts(bottom, top);
    LayerSetUp(bottom, top);
    Reshape(bottom, top);
    SetLossWeights(top);
  }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;

  inline Dtype Forward(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  inline void Backward(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                       const vector<Blob<Dtype>*>& bottom);

  vector<shared_ptr<Blob<Dtype> > >& blobs() { return blobs_; }
  const LayerParameter& layer_param() const { return layer_param_; }
  inline Dtype loss(int top_index) const { return ((int)loss_.size() > top_index) ? loss_[top_index] : Dtype(0); }
  inline void set_loss(int top_index, Dtype value) {
    if ((int)loss_.size() <= top_index) loss_.resize(top_index + 1, Dtype(0));
    loss_[top_index] = value;
  }
  virtual inline const char* type() const { return ""; }
  virtual inline int ExactNumBottomBlobs() const { return -1; }
  virtual inline int MinBottomBlobs() const { return -1; }
  virtual inline int MaxBottomBlobs() const { return -1; }
  virtual inline int ExactNumTopBlobs() const { return -1; }
  virtual inline int MinTopBlobs() const { return -1; }
  virtual inline int MaxTopBlobs() const { return -1; }
  virtual inline bool AutoTopBlobs() const { return false; }
  virtual inline bool AllowForceBackward(int) const { return true; }
  inline bool param_propagate_down(int id) {
    return ((int)param_propagate_down_.size() > id) ? param_propagate_down_[id] : false;
  }
  inline void set_param_propagate_down(int id, bool v) {
    if ((int)param_propagate_down_.size() <= id) param_propagate_down_.resize(id + 1, true);
    param_propagate_down_[id] = v;
  }
  // NOT in the reference's Layer: per-layer switches of this implementation (include/mms_layer.h:
  // mms_layer_set_option).  Returns false for a key the layer does not know.
  virtual bool SetOption(const std::string& key, int value) { return false; }

 protected:
  LayerParameter layer_param_;
  vector<shared_ptr<Blob<Dtype> > > blobs_;
  vector<bool> param_propagate_down_;
  vector<Dtype> loss_;

  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    return Forward_cpu(bottom, top);
  }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom) = 0;
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                            const vector<Blob<Dtype>*>& bottom) {
    Backward_cpu(top, propagate_down, bottom);
  }

  virtual void CheckBlobCounts(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    if (ExactNumBottomBlobs() >= 0)
      CHECK_EQ(ExactNumBottomBlobs(), (int)bottom.size()) << type() << " Layer takes " << ExactNumBottomBlobs() << " bottom blob(s) as input.";
    if (MinBottomBlobs() >= 0) CHECK_LE(MinBottomBlobs(), (int)bottom.size()) << type() << " Layer takes at least " << MinBottomBlobs() << " bottom blob(s) as input.";
    if (MaxBottomBlobs() >= 0) CHECK_GE(MaxBottomBlobs(), (int)bottom.size()) << type() << " Layer takes at most " << MaxBottomBlobs() << " bottom blob(s) as input.";
    if (ExactNumTopBlobs() >= 0)
      CHECK_EQ(ExactNumTopBlobs(), (int)top.size()) << type() << " Layer produces " << ExactNumTopBlobs() << " top blob(s) as output.";
    if (MinTopBlobs() >= 0) CHECK_LE(MinTopBlobs(), (int)top.size()) << type() << " Layer produces at least " << MinTopBlobs() << " top blob(s) as output.";
    if (MaxTopBlobs() >= 0) CHECK_GE(MaxTopBlobs(), (int)top.size()) << type() << " Layer produces at most " << MaxTopBlobs() << " top blob(s) as output.";
  }

  // Loss weights are stored in the top blobs' diff (layer.hpp:405-421).
  inline void SetLossWeights(const vector<Blob<Dtype>*>& top) {
    const int num_loss_weights = layer_param_.loss_weight_size();
    if (num_loss_weights) {
      CHECK_EQ((int)top.size(), num_loss_weights) << "loss_weight must be unspecified or specified once per top blob.";
      for (size_t top_id = 0; top_id < top.size(); ++top_id) {
        const Dtype loss_weight = layer_param_.loss_weight((int)top_id);
        if (loss_weight == Dtype(0)) continue;
        this->set_loss((int)top_id, loss_weight);
        const int count = top[top_id]->count();
        Dtype* loss_multiplier = top[top_id]->mutable_cpu_diff();
        for (int i = 0; i < count; ++i) loss_multiplier[i] = loss_weight;
      }
    }
  }
};

template <typename Dtype>
inline Dtype Layer<Dtype>::Forward(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  Dtype loss = 0;
  Reshape(bottom, top);
  switch (Caffe::mode()) {
    case Caffe::CPU:
      Forward_cpu(bottom, top);
      break;
    case Caffe::GPU:
      Forward_gpu(bottom, top);
      break;
  }
  // loss = sum_top dot(top.data, top.diff) for tops that carry a loss weight (layer.hpp:462-481: a host loop
  // in CPU mode, caffe_gpu_dot on the device in GPU mode -- the blob's head does not move to the host)
  for (size_t top_id = 0; top_id < top.size(); ++top_id) {
    if (!this->loss((int)top_id)) continue;
    const int count = top[top_id]->count();
    Dtype blob_loss = 0;
    if (Caffe::mode() == Caffe::GPU) {
      blob_loss = caffe_gpu_dot(count, top[top_id]->gpu_data(), top[top_id]->gpu_diff());
    } else {
      const Dtype* data = top[top_id]->cpu_data();
      const Dtype* loss_weights = top[top_id]->cpu_diff();
      for (int i = 0; i < count; ++i) blob_loss += data[i] * loss_weights[i];
    }
    loss += blob_loss;
  }
  return loss;
}

template <typename Dtype>
inline void Layer<Dtype>::Backward(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                                   const vector<Blob<Dtype>*>& bottom) {
  switch (Caffe::mode()) {
    case Caffe::CPU:
      Backward_cpu(top, propagate_down, bottom);
      break;
    case Caffe::GPU:
      Backward_gpu(top, propagate_down, bottom);
      break;
  }
}

// ---------------------------------- LayerRegistry ----------------------------
template <typename Dtype>
class LayerRegistry {
 public:
  typedef shared_ptr<Layer<Dtype> > (*Creator)(const LayerParameter&);
  typedef std::map<string, Creator> CreatorRegistry;
  static CreatorRegistry& Registry() {
    static CreatorRegistry* g_registry_ = new CreatorRegistry();
    return *g_registry_;
  }
  static void AddCreator(const string& type, Creator creator) {
    CreatorRegistry& registry = Registry();
    CHECK_EQ(registry.count(type), (size_t)0) << "Layer type " << type << " already registered.";
    registry[type] = creator;
  }
  static shared_ptr<Layer<Dtype> > CreateLayer(const LayerParameter& param) {
    const string& type = param.type();
    CreatorRegistry& registry = Registry();
    CHECK_EQ(registry.count(type), (size_t)1) << "Unknown layer type: " << type << " (known types: " << LayerTypeListString() << ")";
    return registry[type](param);
  }
  static vector<string> LayerTypeList() {
    vector<string> v;
    for (auto& kv : Registry()) v.push_back(kv.first);
    return v;
  }
 private:
  LayerRegistry() {}
  static string LayerTypeListString() {
    string s;
    for (auto& t : LayerTypeList()) s += (s.empty() ? "" : ", ") + t;
    return s;
  }
};

template <typename Dtype>
class LayerRegisterer {
 public:
  LayerRegisterer(const string& type, shared_ptr<Layer<Dtype> > (*creator)(const LayerParameter&)) {
    LayerRegistry<Dtype>::AddCreator(type, creator);
  }
};

// float for every layer; float AND double (common.hpp:41-44) for the three layers of the path,
// whose C ABI has _f64 entry points (the *_FD macros below).
#define REGISTER_LAYER_CREATOR(type, creator) \
  static LayerRegisterer<float> g_creator_f_##type(#type, creator<float>)
#define REGISTER_LAYER_CLASS(type)                                                    \
  template <typename Dtype>                                                           \
  shared_ptr<Layer<Dtype> > Creator_##type##Layer(const LayerParameter& param) {      \
    return shared_ptr<Layer<Dtype> >(new type##Layer<Dtype>(param));                  \
  }                                                                                   \
  REGISTER_LAYER_CREATOR(type, Creator_##type##Layer)
#define INSTANTIATE_CLASS(classname) template class classname<float>
#define REGISTER_LAYER_CLASS_FD(type)                                                 \
  REGISTER_LAYER_CLASS(type);                                                         \
  static LayerRegisterer<double> g_creator_d_##type(#type, Creator_##type##Layer<double>)
#define INSTANTIATE_CLASS_FD(classname) \
  template class classname<float>;      \
  template class classname<double>

}  // namespace caffe
#endif  // MMS_CAFFE_API_HPP_
