// tests/native/layer_host.cpp -- a C++ host written against csrc/caffe_api.hpp exactly as INTEGRATION.md
// section B shows (LayerRegistry::CreateLayer from prototxt text, SetUp / Forward / Backward over Blob).
// Built with hipcc against libmms_caffe.so and run on the GPU by tests/test_native_hosts.py.
#include <cmath>
#include <cstdio>
#include <string>

#include "caffe_api.hpp"

static int fail(const char* what) { std::printf("FAIL: %s\n", what); return 1; }

int main() {
  caffe::Caffe::set_mode(caffe::Caffe::GPU);
  caffe::LayerParameter p;
  std::string err;
  if (!caffe::ReadLayerParameterFromText(
          "layer { name: 'sim' type: 'SimCross' bottom: 'q' bottom: 'a' top: 's' }", &p, &err))
    return fail(err.c_str());
  auto layer = caffe::LayerRegistry<float>::CreateLayer(p);               // layer_factory.hpp:56-84
  if (std::string(layer->type()) != "SimCross") return fail("type()");
  const int N = 6, W = 3, D = 8;
  caffe::Blob<float> q({N, W, D}), a({N, W, D}), top;
  float* qd = q.mutable_cpu_data();
  float* ad = a.mutable_cpu_data();
  for (int i = 0; i < N * W * D; ++i) { qd[i] = 0.25f * (float)(i % 7); ad[i] = qd[i]; }
  // word (n = 0, k = 1) of `a` is 3 away from its q twin in one coordinate: distance 3 -> T = 1 / (1 + 3)
  ad[1 * D + 2] += 3.0f;
  layer->SetUp({&q, &a}, {&top});                                         // layer.hpp:67-74
  if (top.num_axes() != 4 || top.shape(0) != N || top.shape(1) != 1 || top.shape(2) != W || top.shape(3) != W)
    return fail("top shape (N,1,W1,W2)");
  layer->Forward({&q, &a}, {&top});
  const float* t = top.cpu_data();
  if (t[1 * W + 1] != 0.25f) return fail("T = 1/(1+3) for the displaced word (default dist_mode 1: Euclid)");
  if (t[(1 * W + 0) * W + 0] != 1.0f) return fail("T = 1 for identical words");
  float* td = top.mutable_cpu_diff();
  for (int i = 0; i < top.count(); ++i) td[i] = 1.0f;
  layer->Backward({&top}, {true, true}, {&q, &a});
  const float* dq = q.cpu_diff();
  const float* da = a.cpu_diff();
  for (int i = 0; i < N * W * D; ++i)
    if (!(std::isfinite(dq[i]) && std::isfinite(da[i]))) return fail("finite gradients");
  // moving a's displaced coordinate further away lowers T: d top / d a < 0 there, and dq mirrors it
  if (!(da[1 * D + 2] < 0.f)) return fail("sign of da at the displaced coordinate");
  std::printf("layer_host ok: SimCross via LayerRegistry, T = %.4f, da = %.6f\n", t[1 * W + 1], da[1 * D + 2]);
  return 0;
}
