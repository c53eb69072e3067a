// csrc/caffemodel_io.cpp -- reading and writing layer parameters in the
// reference's snapshot format (the ".caffemodel" half).
//
// A .caffemodel is a binary-serialised NetParameter (src/caffe/proto/caffe.proto:64-96):
// Solver::Snapshot -> Net::ToProto -> Layer::ToProto writes each layer's
// LayerParameter with its `blobs` (include/caffe/layer.hpp:506-514, blob data only,
// no diffs), and Net::CopyTrainedLayersFrom (src/caffe/net.cpp) matches layers BY
// NAME and copies blobs whose shapes agree.  The driver relies on exactly that:
// it reloads `qa_iter_<n>.caffemodel` into a TEST net (do_trec_qa_clean.py:840).
//
// No protoc / libprotobuf in this image, so the subset of the wire format that
// matters is decoded and encoded by hand (proto2 wire format: varint keys,
// wire types 0/1/2/5):
//   NetParameter   { name = 1 (string), layer = 100 (LayerParameter, repeated) }
//   LayerParameter { name = 1, type = 2, bottom = 3, top = 4, blobs = 7 (BlobProto) }
//   BlobProto      { shape = 7 (BlobShape), data = 5 (packed float), diff = 6,
//                    double_data = 8 (packed double), num/channels/height/width = 1..4 }
//   BlobShape      { dim = 1 (packed int64) }
//   V1LayerParameter (NetParameter.layers = 2, pre-2015 snapshots; caffe.proto:1286-1345)
//                  { name = 4, type = 5 (enum -> the type string UpgradeV1LayerType gives it,
//                    util/upgrade_proto.cpp:865-950), blobs = 6, layer = 1 (V0LayerParameter:
//                    name = 1, type = 2, blobs = 50) }
// Unknown fields are skipped, as a protobuf parser does.  Both lists are read -- Caffe upgrades
// V0/V1 snapshots to `layer` before matching names (Net::CopyTrainedLayersFrom sees only names and blobs).
// HDF5-format snapshots (snapshot_format: HDF5; Net::ToHDF5 / CopyTrainedLayersFromHDF5, net.cpp:797-844,
// 893-960: /data/<layer name>/<param index>, float datasets) go through csrc/hdf5_io.cpp; a file is taken for
// HDF5 by its signature, whatever its name (the reference looks at the ".h5" suffix, net.cpp:778-785).
// tests/test_snapshot.py cross-checks both directions against google.protobuf.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <algorithm>
#include <cstdlib>

#include "caffe_api.hpp"
#include "hdf5_io.hpp"
#include "mms_layer.h"

namespace {

struct SnapBlob {
  std::vector<int> shape;
  std::vector<float> data;
};
struct SnapLayer {
  std::string name, type;
  std::vector<SnapBlob> blobs;
};

// ------------------------------- wire-format reader --------------------------
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  Reader sub() {  // length-delimited payload
    const uint64_t n = varint();
    if (!ok || n > (uint64_t)(end - p)) { ok = false; return Reader{p, p}; }
    Reader r{p, p + n};
    p += n;
    return r;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: if (end - p >= 8) p += 8; else ok = false; break;
      case 2: sub(); break;
      case 5: if (end - p >= 4) p += 4; else ok = false; break;
      default: ok = false;
    }
  }
};

bool parse_blob(Reader r, SnapBlob* b) {
  int legacy[4] = {0, 0, 0, 0};
  bool has_legacy = false, has_shape = false;
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field == 7 && wt == 2) {                      // BlobShape
      Reader s = r.sub();
      has_shape = true;
      while (s.ok && !s.done()) {
        const uint64_t k2 = s.varint();
        if ((k2 >> 3) == 1 && (k2 & 7) == 2) {        // packed dims
          Reader d = s.sub();
          while (d.ok && !d.done()) b->shape.push_back((int)d.varint());
        } else if ((k2 >> 3) == 1 && (k2 & 7) == 0) { // unpacked dim
          b->shape.push_back((int)s.varint());
        } else {
          s.skip((int)(k2 & 7));
        }
      }
      if (!s.ok) return false;
    } else if (field == 5 && wt == 2) {               // packed float data
      Reader d = r.sub();
      const size_t n = (size_t)(d.end - d.p) / 4;
      const size_t at = b->data.size();
      b->data.resize(at + n);
      std::memcpy(b->data.data() + at, d.p, n * 4);
    } else if (field == 5 && wt == 5) {               // unpacked float
      float f;
      if (r.end - r.p < 4) return false;
      std::memcpy(&f, r.p, 4); r.p += 4;
      b->data.push_back(f);
    } else if (field == 8 && wt == 2) {               // packed double_data
      Reader d = r.sub();
      const size_t n = (size_t)(d.end - d.p) / 8;
      for (size_t i = 0; i < n; ++i) {
        double v;
        std::memcpy(&v, d.p + 8 * i, 8);
        b->data.push_back((float)v);
      }
    } else if (field >= 1 && field <= 4 && wt == 0) {  // legacy num/channels/height/width
      legacy[field - 1] = (int)r.varint();
      has_legacy = true;
    } else {
      r.skip(wt);
    }
  }
  if (!has_shape && has_legacy) b->shape.assign(legacy, legacy + 4);
  return r.ok;
}

bool parse_layer(Reader r, SnapLayer* l) {
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if ((field == 1 || field == 2) && wt == 2) {
      Reader s = r.sub();
      (field == 1 ? l->name : l->type).assign((const char*)s.p, (size_t)(s.end - s.p));
    } else if (field == 7 && wt == 2) {
      SnapBlob b;
      if (!parse_blob(r.sub(), &b)) return false;
      l->blobs.push_back(std::move(b));
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

// V1LayerParameter.LayerType -> LayerParameter.type (the table of util/upgrade_proto.cpp:865-950, as data)
const char* v1_type_name(int t) {
  static const struct { int v; const char* n; } kTab[] = {
      {35, "AbsVal"}, {1, "Accuracy"}, {30, "ArgMax"}, {2, "BNLL"}, {3, "Concat"}, {37, "ContrastiveLoss"},
      {4, "Convolution"}, {39, "Deconvolution"}, {5, "Data"}, {6, "Dropout"}, {32, "DummyData"},
      {7, "EuclideanLoss"}, {25, "Eltwise"}, {38, "Exp"}, {8, "Flatten"}, {9, "HDF5Data"}, {10, "HDF5Output"},
      {28, "HingeLoss"}, {11, "Im2col"}, {12, "ImageData"}, {13, "InfogainLoss"}, {14, "InnerProduct"},
      {15, "LRN"}, {29, "MemoryData"}, {16, "MultinomialLogisticLoss"}, {34, "MVN"}, {17, "Pooling"},
      {26, "Power"}, {18, "ReLU"}, {19, "Sigmoid"}, {27, "SigmoidCrossEntropyLoss"}, {36, "Silence"},
      {20, "Softmax"}, {21, "SoftmaxWithLoss"}, {22, "Split"}, {33, "Slice"}, {23, "TanH"}, {24, "WindowData"},
      {31, "Threshold"}};
  for (const auto& e : kTab)
    if (e.v == t) return e.n;
  return "";
}

// V0LayerParameter (nested in a V1 layer as `layer = 1`): name = 1, type = 2 (string), blobs = 50
bool parse_v0_layer(Reader r, SnapLayer* l) {
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if ((field == 1 || field == 2) && wt == 2) {
      Reader s = r.sub();
      (field == 1 ? l->name : l->type).assign((const char*)s.p, (size_t)(s.end - s.p));
    } else if (field == 50 && wt == 2) {
      SnapBlob b;
      if (!parse_blob(r.sub(), &b)) return false;
      l->blobs.push_back(std::move(b));
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

bool parse_v1_layer(Reader r, SnapLayer* l) {
  SnapLayer v0;
  bool has_v0 = false;
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field == 4 && wt == 2) {
      Reader s = r.sub();
      l->name.assign((const char*)s.p, (size_t)(s.end - s.p));
    } else if (field == 5 && wt == 0) {
      l->type = v1_type_name((int)r.varint());
    } else if (field == 6 && wt == 2) {
      SnapBlob b;
      if (!parse_blob(r.sub(), &b)) return false;
      l->blobs.push_back(std::move(b));
    } else if (field == 1 && wt == 2) {
      if (!parse_v0_layer(r.sub(), &v0)) return false;
      has_v0 = true;
    } else {
      r.skip(wt);
    }
  }
  if (has_v0) {                                     // a V0 net wrapped in V1 connections: the V0 message has it all
    if (l->name.empty()) l->name = v0.name;
    if (l->type.empty()) l->type = v0.type;
    if (l->blobs.empty()) l->blobs = std::move(v0.blobs);
  }
  return r.ok;
}

// ------------------------------- wire-format writer --------------------------
void put_varint(std::string* o, uint64_t v) {
  while (v >= 0x80) { o->push_back((char)(v | 0x80)); v >>= 7; }
  o->push_back((char)v);
}
void put_key(std::string* o, int field, int wt) { put_varint(o, ((uint64_t)field << 3) | (uint64_t)wt); }
void put_bytes(std::string* o, int field, const std::string& payload) {
  put_key(o, field, 2);
  put_varint(o, payload.size());
  o->append(payload);
}
std::string encode_blob(const SnapBlob& b) {
  std::string dims, shape, out;
  for (int d : b.shape) put_varint(&dims, (uint64_t)(int64_t)d);
  put_bytes(&shape, 1, dims);                                   // BlobShape.dim, packed
  put_bytes(&out, 7, shape);                                    // BlobProto.shape
  std::string data((const char*)b.data.data(), b.data.size() * 4);
  put_bytes(&out, 5, data);                                     // BlobProto.data, packed
  return out;
}
std::string encode_layer(const SnapLayer& l) {
  std::string out;
  put_bytes(&out, 1, l.name);
  put_bytes(&out, 2, l.type);
  for (const SnapBlob& b : l.blobs) put_bytes(&out, 7, encode_blob(b));
  return out;
}

}  // namespace

struct mms_snapshot {
  std::string net_name;
  std::vector<SnapLayer> layers;
  bool hdf5 = false;
};

namespace {
// /data/<layer name>/<param index> -> SnapLayer{name, "", blobs in index order}; HDF5 snapshots carry no types
mms_snapshot* open_hdf5_snapshot(const char* path, std::string* err) {
  mms_h5::File f;
  if (!f.Open(path, err)) return nullptr;
  std::unique_ptr<mms_snapshot> s(new mms_snapshot);
  s->hdf5 = true;
  bool has_data = false;
  for (const std::string& g : f.GroupNames()) {
    if (g == "data") has_data = true;
    if (g.compare(0, 5, "data/") == 0 && g.find('/', 5) == std::string::npos) {
      SnapLayer l;
      l.name = g.substr(5);
      s->layers.push_back(std::move(l));
    }
  }
  if (!has_data) { *err = std::string("Error reading weights from ") + path + ": no /data group"; return nullptr; }
  for (SnapLayer& l : s->layers) {
    for (int j = 0;; ++j) {
      const std::string ds = "data/" + l.name + "/" + std::to_string(j);
      if (!f.Find(ds)) break;
      SnapBlob b;
      mms_h5::DatasetInfo info;
      if (!f.ReadFloat(ds, &info, &b.data, err)) return nullptr;
      for (int64_t d : info.dims) b.shape.push_back((int)d);
      l.blobs.push_back(std::move(b));
    }
  }
  return s.release();
}
}  // namespace

extern "C" {

mms_snapshot_t* mms_snapshot_open(const char* path, char* err, int err_len) {
  auto fail = [&](const char* m) -> mms_snapshot_t* {
    if (err && err_len > 0) std::snprintf(err, err_len, "%s: %s", m, path ? path : "(null)");
    return nullptr;
  };
  FILE* f = path ? std::fopen(path, "rb") : nullptr;
  if (!f) return fail("cannot open snapshot");
  std::string buf;
  char chunk[1 << 16];
  size_t n;
  while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, n);
  std::fclose(f);
  static const unsigned char kH5[8] = {0x89, 'H', 'D', 'F', 0x0d, 0x0a, 0x1a, 0x0a};
  bool is_h5 = false;
  for (size_t off = 0; off + 8 <= buf.size(); off = off ? off * 2 : 512)   // the signature may follow a user block
    if (!std::memcmp(buf.data() + off, kH5, 8)) { is_h5 = true; break; }
  if (is_h5) {
    std::string e;
    mms_snapshot* hs = open_hdf5_snapshot(path, &e);
    if (!hs && err && err_len > 0) std::snprintf(err, err_len, "%s", e.c_str());
    return hs;
  }
  std::unique_ptr<mms_snapshot> s(new mms_snapshot);
  Reader r{(const uint8_t*)buf.data(), (const uint8_t*)buf.data() + buf.size()};
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field == 1 && wt == 2) {
      Reader t = r.sub();
      s->net_name.assign((const char*)t.p, (size_t)(t.end - t.p));
    } else if (field == 100 && wt == 2) {
      SnapLayer l;
      if (!parse_layer(r.sub(), &l)) return fail("malformed LayerParameter in");
      s->layers.push_back(std::move(l));
    } else if (field == 2 && wt == 2) {              // V1 `layers`
      SnapLayer l;
      if (!parse_v1_layer(r.sub(), &l)) return fail("malformed V1LayerParameter in");
      s->layers.push_back(std::move(l));
    } else {
      r.skip(wt);
    }
  }
  if (!r.ok) return fail("malformed NetParameter in");
  return s.release();
}
void mms_snapshot_close(mms_snapshot_t* s) { delete s; }
const char* mms_snapshot_net_name(const mms_snapshot_t* s) { return s->net_name.c_str(); }
int mms_snapshot_num_layers(const mms_snapshot_t* s) { return (int)s->layers.size(); }
const char* mms_snapshot_layer_name(const mms_snapshot_t* s, int i) { return s->layers[i].name.c_str(); }
const char* mms_snapshot_layer_type(const mms_snapshot_t* s, int i) { return s->layers[i].type.c_str(); }
int mms_snapshot_num_blobs(const mms_snapshot_t* s, int i) { return (int)s->layers[i].blobs.size(); }
int mms_snapshot_blob_shape(const mms_snapshot_t* s, int i, int j, int* shape, int max_axes) {
  const std::vector<int>& sh = s->layers[i].blobs[j].shape;
  for (int a = 0; a < (int)sh.size() && a < max_axes; ++a) shape[a] = sh[a];
  return (int)sh.size();
}
int mms_snapshot_blob_count(const mms_snapshot_t* s, int i, int j) { return (int)s->layers[i].blobs[j].data.size(); }
const float* mms_snapshot_blob_data(const mms_snapshot_t* s, int i, int j) { return s->layers[i].blobs[j].data.data(); }

// Net::CopyTrainedLayersFrom for one layer: match by name, blob counts and shapes must agree.
// Returns 0 on success, 1 if no layer of that name is in the snapshot (Caffe ignores such
// layers), 2 on a blob-count mismatch and 3 on a shape mismatch (both fatal in Caffe).
int mms_layer_copy_from_snapshot(mms_layer_t* layer, const mms_snapshot_t* s, const char* layer_name) {
  const SnapLayer* src = nullptr;
  for (const SnapLayer& l : s->layers)
    if (l.name == layer_name) { src = &l; break; }
  if (!src) return 1;
  const int n = mms_layer_num_param_blobs(layer);
  if (n != (int)src->blobs.size()) return 2;
  for (int j = 0; j < n; ++j) {
    mms_blob_t* dst = mms_layer_param_blob(layer, j);
    const SnapBlob& b = src->blobs[j];
    if (mms_blob_count(dst) != (int)b.data.size()) return 3;
    // ShapeEquals: identical shapes, or (legacy 4-D source) identical after stripping leading 1s
    std::vector<int> a, c;
    for (int ax = 0; ax < mms_blob_num_axes(dst); ++ax) a.push_back(mms_blob_shape(dst, ax));
    c = b.shape;
    if (a != c) {
      auto strip = [](std::vector<int> v) { while (v.size() > 1 && v.front() == 1) v.erase(v.begin()); return v; };
      if (strip(a) != strip(c)) return 3;
    }
  }
  for (int j = 0; j < n; ++j) {
    mms_blob_t* dst = mms_layer_param_blob(layer, j);
    std::memcpy(mms_blob_mutable_cpu(dst, 0), src->blobs[j].data.data(), src->blobs[j].data.size() * 4);
  }
  return 0;
}

struct mms_snapshot_writer {
  std::string net_name;
  std::vector<SnapLayer> layers;
};
mms_snapshot_writer_t* mms_snapshot_writer_create(const char* net_name) {
  auto* w = new mms_snapshot_writer;
  w->net_name = net_name ? net_name : "";
  return w;
}
void mms_snapshot_writer_destroy(mms_snapshot_writer_t* w) { delete w; }
void mms_snapshot_writer_add_layer(mms_snapshot_writer_t* w, const char* name, const char* type) {
  SnapLayer l;
  l.name = name ? name : "";
  l.type = type ? type : "";
  w->layers.push_back(std::move(l));
}
void mms_snapshot_writer_add_blob(mms_snapshot_writer_t* w, const int* shape, int num_axes, const float* data) {
  SnapBlob b;
  size_t count = 1;
  for (int a = 0; a < num_axes; ++a) { b.shape.push_back(shape[a]); count *= (size_t)shape[a]; }
  b.data.assign(data, data + count);
  w->layers.back().blobs.push_back(std::move(b));
}
// Layer::ToProto (layer.hpp:506-514): name, type and the data of every parameter blob.
void mms_snapshot_writer_add_from_layer(mms_snapshot_writer_t* w, mms_layer_t* layer, const char* name) {
  mms_snapshot_writer_add_layer(w, name, mms_layer_type(layer));
  for (int j = 0; j < mms_layer_num_param_blobs(layer); ++j) {
    mms_blob_t* b = mms_layer_param_blob(layer, j);
    std::vector<int> shape;
    for (int ax = 0; ax < mms_blob_num_axes(b); ++ax) shape.push_back(mms_blob_shape(b, ax));
    mms_snapshot_writer_add_blob(w, shape.data(), (int)shape.size(), mms_blob_cpu(b, 0));
  }
}
// Net::ToHDF5 (net.cpp:893-960), data only: /data/<layer name>/<param index>, float32, the blob's shape.
// Layers without parameters get no group (the reference writes an empty one; nothing reads it).  The in-tree
// HDF5 writer holds at most 8 members per group: 8 parameter layers per file.
int mms_snapshot_writer_save_hdf5(const mms_snapshot_writer_t* w, const char* path, char* err, int err_len) {
  std::vector<mms_h5::WriteDataset> sets;
  for (const SnapLayer& l : w->layers) {
    for (size_t j = 0; j < l.blobs.size(); ++j) {
      mms_h5::WriteDataset d;
      d.name = "data/" + l.name + "/" + std::to_string(j);
      for (int x : l.blobs[j].shape) d.dims.push_back(x);
      d.elem_size = 4;
      d.values.assign(l.blobs[j].data.begin(), l.blobs[j].data.end());
      sets.push_back(std::move(d));
    }
  }
  std::string e;
  if (!mms_h5::WriteContiguous(path, sets, &e)) {
    if (err && err_len > 0) std::snprintf(err, err_len, "%s", e.c_str());
    return 1;
  }
  return 0;
}
int mms_snapshot_writer_save(const mms_snapshot_writer_t* w, const char* path) {
  std::string out;
  put_bytes(&out, 1, w->net_name);
  for (const SnapLayer& l : w->layers) put_bytes(&out, 100, encode_layer(l));
  FILE* f = std::fopen(path, "wb");
  if (!f) return 1;
  const size_t n = std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
  return n == out.size() ? 0 : 1;
}

}  // extern "C"
