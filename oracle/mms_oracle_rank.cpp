// oracle/mms_oracle_rank.cpp -- TEST INFRASTRUCTURE ONLY (see mms_oracle.c).
//
// CPU restatement of the reference's forward-only ranking metrics, which
// define what "ranking output" means for the bit-identical check:
//   MAPLayer<Dtype>::Forward_cpu   src/caffe/layers/map_layer.cpp:41-100
//   MRRLayer<Dtype>::Forward_cpu   src/caffe/layers/mrr_layer.cpp:38-79
//   AUCLayer<Dtype>::Forward_cpu   src/caffe/layers/auc_layer.cpp:47-136
// Written in C++ (not C) on purpose: the reference buckets with std::map<int,…>
// and orders with std::sort (unstable) under a `lhs.first > rhs.first`
// comparator on std::pair<float,int>; using the same library calls keeps the
// implementation-defined tie order identical to a libstdc++ build of the
// reference.  PARITY UNPINNED -- the reference holds no fixture for these.
#include <algorithm>
#include <map>
#include <utility>
#include <vector>

namespace {
bool score_desc(const std::pair<float, int>& l, const std::pair<float, int>& r) {
  return l.first > r.first;  // map_layer.cpp:34-38, mrr_layer.cpp:33-35
}

// score of item i is prob[i*(fixed_axis+1)+fixed_axis] (map_layer.cpp:50):
// the reference assumes exactly fixed_axis+1 columns.
template <typename Dtype>
std::map<int, std::vector<std::pair<float, int> > > bucket(
    int n, int fixed_axis, const Dtype* prob, const Dtype* label,
    const Dtype* group) {
  std::map<int, std::vector<std::pair<float, int> > > all;
  for (int i = 0; i < n; ++i)
    all[group[i]].push_back(
        std::make_pair(prob[(size_t)i * (fixed_axis + 1) + fixed_axis], label[i]));
  return all;
}

template <typename Dtype>
Dtype map_impl(int n, int fixed_axis, const Dtype* prob, const Dtype* label,
               const Dtype* group, int* effective) {
  auto all = bucket(n, fixed_axis, prob, label, group);
  Dtype map_ = Dtype(0);
  int effect_sample = 0;
  for (auto it = all.begin(); it != all.end(); ++it) {
    std::sort(it->second.begin(), it->second.end(), score_desc);
    Dtype ap = 0;
    int map_rank = 0;
    int neg_exist = 0;
    for (size_t i = 0; i < it->second.size(); ++i) {
      if (it->second[i].second == 1) ap += (++map_rank) / (Dtype)(i + 1);
      else if (neg_exist == 0) neg_exist = 1;
    }
    if (map_rank < 1 || neg_exist == 0) continue;  // map_layer.cpp:90-92
    ++effect_sample;
    map_ += ap / map_rank;
  }
  if (effective) *effective = effect_sample;
  return map_ / effect_sample;  // map_layer.cpp:99 (NaN if no group counts)
}

template <typename Dtype>
Dtype mrr_impl(int n, int fixed_axis, const Dtype* prob, const Dtype* label,
               const Dtype* group, int* effective) {
  auto all = bucket(n, fixed_axis, prob, label, group);
  Dtype mrr = Dtype(0);
  int effect_sample = 0;
  for (auto it = all.begin(); it != all.end(); ++it) {
    std::sort(it->second.begin(), it->second.end(), score_desc);
    int mrr_rank = -1;
    int neg_exist = 0;
    for (size_t i = 0; i < it->second.size(); ++i) {
      if (mrr_rank < 0 && it->second[i].second == 1) mrr_rank = (int)i;
      if (neg_exist == 0 && it->second[i].second == 0) neg_exist = 1;
      if (neg_exist && mrr_rank > -1) break;
    }
    if (mrr_rank < 0 || neg_exist == 0) continue;
    ++effect_sample;
    mrr += 1.0 / (mrr_rank + 1);  // double literal: sum promoted, then stored
  }
  if (effective) *effective = effect_sample;
  return mrr / effect_sample;
}

// AUC over all items (outer_num = n, inner_num = 1 as the driver uses it,
// do_trec_qa_clean.py:496): sort desc, high += label, auc += high*(1-label).
template <typename Dtype>
Dtype auc_impl(int n, int dim, int fixed_axis, const Dtype* prob,
               const Dtype* label) {
  Dtype auc_value = 0;
  int high = 0, count = 0;
  std::vector<std::pair<Dtype, int> > v;
  for (int i = 0; i < n; ++i) {
    v.push_back(std::make_pair(prob[(size_t)i * dim + fixed_axis],
                               static_cast<int>(label[i])));
    ++count;
  }
  // the reference passes a pair<float,int> comparator to a pair<Dtype,int>
  // vector (auc_layer.cpp:42-44,93-95): for double this converts through float.
  std::sort(v.begin(), v.end(),
            [](const std::pair<Dtype, int>& l, const std::pair<Dtype, int>& r) {
              return score_desc(std::pair<float, int>(l), std::pair<float, int>(r));
            });
  for (size_t i = 0; i < v.size(); ++i) {
    high += v[i].second;
    auc_value += high * (1 - v[i].second);
  }
  if (high > 0) return auc_value / high / (count - high);
  return 0;
}
// auc_layer.cpp:47-136 with its general indexing: items (i, j), i < outer, j < inner; score
// prob[i*dim + fixed_axis*inner + j] with dim = channels*inner (:75-76), label label[i*inner + j] (:67-68),
// ignore_label items skipped before they are counted (:69-71).
template <typename Dtype>
Dtype auc_nd_impl(int outer, int channels, int inner, int fixed_axis, const Dtype* prob, const Dtype* label,
                  int has_ignore, int ignore_label) {
  Dtype auc_value = 0;
  int high = 0, count = 0;
  const int dim = channels * inner;
  std::vector<std::pair<Dtype, int> > v;
  for (int i = 0; i < outer; ++i) {
    for (int j = 0; j < inner; ++j) {
      const int label_value = static_cast<int>(label[(size_t)i * inner + j]);
      if (has_ignore && label_value == ignore_label) continue;
      v.push_back(std::make_pair(prob[(size_t)i * dim + (size_t)fixed_axis * inner + j], label_value));
      ++count;
    }
  }
  std::sort(v.begin(), v.end(),
            [](const std::pair<Dtype, int>& l, const std::pair<Dtype, int>& r) {
              return score_desc(std::pair<float, int>(l), std::pair<float, int>(r));
            });
  for (size_t i = 0; i < v.size(); ++i) {
    high += v[i].second;
    auc_value += high * (1 - v[i].second);
  }
  if (high > 0) return auc_value / high / (count - high);
  return 0;
}
}  // namespace

extern "C" {
float oracle_auc_nd_f32(int outer, int channels, int inner, int fixed_axis, const float* prob,
                        const float* label, int has_ignore, int ignore_label) {
  return auc_nd_impl<float>(outer, channels, inner, fixed_axis, prob, label, has_ignore, ignore_label);
}
float oracle_map_f32(int n, int fixed_axis, const float* prob,
                     const float* label, const float* group, int* effective) {
  return map_impl<float>(n, fixed_axis, prob, label, group, effective);
}
float oracle_mrr_f32(int n, int fixed_axis, const float* prob,
                     const float* label, const float* group, int* effective) {
  return mrr_impl<float>(n, fixed_axis, prob, label, group, effective);
}
float oracle_auc_f32(int n, int dim, int fixed_axis, const float* prob,
                     const float* label) {
  return auc_impl<float>(n, dim, fixed_axis, prob, label);
}
double oracle_map_f64(int n, int fixed_axis, const double* prob,
                      const double* label, const double* group, int* effective) {
  return map_impl<double>(n, fixed_axis, prob, label, group, effective);
}
double oracle_mrr_f64(int n, int fixed_axis, const double* prob,
                      const double* label, const double* group, int* effective) {
  return mrr_impl<double>(n, fixed_axis, prob, label, group, effective);
}
}
