/*
 * include/mms_layer.h -- C handle API over the C++ mirror of the Caffe
 * Layer/Blob interface (libmms_caffe.so, csrc/caffe_api.hpp + caffe_layers.cpp).
 *
 * The C++ classes are what a Caffe build would use directly;
 * this handle API exists so that non-C++ hosts -- the pytest suite, a pycaffe
 * style binding -- can drive the same objects:
 *   create a layer from prototxt text through LayerRegistry::CreateLayer
 *   (include/caffe/layer_factory.hpp:56-84), wire Blobs, SetUp, Forward, Backward.
 *
 * Error behaviour is Caffe's: a failed CHECK prints the message and abort()s
 * (glog LOG(FATAL), include/caffe/util/device_alternate.hpp:48-76).  Only
 * mms_layer_create reports a prototxt syntax error by returning NULL.
 */
#ifndef MMS_LAYER_H_
#define MMS_LAYER_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mms_blob mms_blob_t;
typedef struct mms_layer mms_layer_t;

/* Blob<float> (include/caffe/blob.hpp:24-277) */
mms_blob_t* mms_blob_create(const int* shape, int num_axes);
void mms_blob_destroy(mms_blob_t* b);
void mms_blob_reshape(mms_blob_t* b, const int* shape, int num_axes);
int mms_blob_num_axes(const mms_blob_t* b);
int mms_blob_shape(const mms_blob_t* b, int axis);
int mms_blob_count(const mms_blob_t* b);
/* which = 0 data, 1 diff.  The *_cpu_* / *_gpu_* pairs move the SyncedMemory
 * head exactly like Blob::cpu_data()/mutable_gpu_diff() etc. */
const float* mms_blob_cpu(mms_blob_t* b, int which);
float* mms_blob_mutable_cpu(mms_blob_t* b, int which);
const float* mms_blob_gpu(mms_blob_t* b, int which);
float* mms_blob_mutable_gpu(mms_blob_t* b, int which);

/* Layer<float> (include/caffe/layer.hpp:32-445).  `prototxt` is one
 * `layer { ... }` message in protobuf text format, e.g.
 *   layer { name: "sim" type: "SimCross" bottom: "q" bottom: "a" top: "s"
 *           sim_cross_param { dist_mode: 2 mesure_count: 4 } }            */
mms_layer_t* mms_layer_create(const char* prototxt, char* err, int err_len);
void mms_layer_destroy(mms_layer_t* l);
const char* mms_layer_type(const mms_layer_t* l);
void mms_layer_setup(mms_layer_t* l, mms_blob_t* const* bottom, int nbottom,
                     mms_blob_t* const* top, int ntop);
float mms_layer_forward(mms_layer_t* l, mms_blob_t* const* bottom, int nbottom,
                        mms_blob_t* const* top, int ntop);
void mms_layer_backward(mms_layer_t* l, mms_blob_t* const* top, int ntop,
                        const int* propagate_down, mms_blob_t* const* bottom,
                        int nbottom);
int mms_layer_num_param_blobs(mms_layer_t* l);
mms_blob_t* mms_layer_param_blob(mms_layer_t* l, int i); /* borrowed; do not destroy */
void mms_layer_set_param_propagate_down(mms_layer_t* l, int i, int v);
/* Per-layer switches of THIS implementation (the reference's Layer has none).  Returns 0, or 1 for a key the
 * layer does not know.
 *   SimCross  "euclid_backward_mode": MMS_EUCLID_BWD_FP32 (0) / MMS_EUCLID_BWD_REFERENCE (1) for this layer's
 *             Backward; -1 (default) = the calling thread's mode (include/mms.h).
 *   SimMatrix "private_qw": 1 keeps the forward's Q*W in a blob of the layer; 0 (default) leaves it in
 *             bottom[1]'s diff like the reference (sim_matrix_layer.cpp:58) and Backward scales it there. */
int mms_layer_set_option(mms_layer_t* l, const char* key, int value);

/* ---- Whole nets: the generated net files of the driver, unmodified -------------------------------------------
 * `prototxt` is a NetParameter in protobuf text format (src/caffe/proto/caffe.proto:63-110, as
 * python/caffe/net_spec.py:31-79 writes it; examples/trec_qa_w2v_mms/do_trec_qa_clean.py:608-615 generates three
 * per run).  Layers are filtered by `phase` (0 TRAIN, 1 TEST; include / exclude rules, net.cpp:272-330), blobs are
 * wired by name (a top named like its bottom is computed in place), parameters named alike are shared
 * (net.cpp:450-530).  Every layer type this library implements is instantiated; the others (Convolution, Pooling,
 * Dropout, InnerProduct, Softmax, ...) are LISTED and SKIPPED -- their parameter messages are parsed over, and a
 * blob that only they produce is an input of the part that runs: fill it through mms_net_blob() before
 * mms_net_setup().  mms_net_setup() = Layer::SetUp for every supported layer whose bottoms have a shape, in file
 * order (returns how many can run; mms_net_layer_why_not() says what a supported but idle layer is waiting for);
 * mms_net_forward() / mms_net_backward() = Net::ForwardFromTo / BackwardFromTo over those layers (net.cpp:535-591)
 * -- all forwards, then all backwards.  Handles returned by mms_net_blob / mms_net_layer are borrowed. */
typedef struct mms_net mms_net_t;
mms_net_t* mms_net_create(const char* prototxt, int phase, char* err, int err_len);
void mms_net_destroy(mms_net_t* n);
/* "fuse_embed_scoring" = 1 (forward-only use: evaluation): a SimCross layer fed by two Embed layers that read one
 * table (and bias) shared by parameter name, and feed nothing else, is run straight from the word ids -- one
 * launch, the (N,W,D) blobs never written; geometries without a fused kernel run the three layers as usual;
 * mms_net_backward after such a Forward aborts.  Returns 0 if the option is known.  mms_net_num_fused: how many
 * SimCross layers the current plan runs that way (after mms_net_setup). */
int mms_net_set_option(mms_net_t* n, const char* key, int value);
int mms_net_num_fused(const mms_net_t* n);
/* Blobs that feed more than one back-propagating layer get Net::Init's Split treatment (insert_splits.cpp:13-88): each
 * consumer reads the blob through a blob of its own (data shared, diff private) and mms_net_backward sums the diffs in
 * consumer order (split_layer.cpp:38-57) before the producer's Backward.  mms_net_num_splits: how many blobs. */
int mms_net_num_splits(const mms_net_t* n);
const char* mms_net_name(const mms_net_t* n);
int mms_net_num_layers(const mms_net_t* n);
const char* mms_net_layer_name(const mms_net_t* n, int i);
const char* mms_net_layer_type(const mms_net_t* n, int i);
int mms_net_layer_supported(const mms_net_t* n, int i);
int mms_net_layer_runnable(const mms_net_t* n, int i);
const char* mms_net_layer_why_not(const mms_net_t* n, int i);
mms_layer_t* mms_net_layer(mms_net_t* n, int i);
int mms_net_num_blobs(const mms_net_t* n);
const char* mms_net_blob_name(const mms_net_t* n, int i);
mms_blob_t* mms_net_blob(mms_net_t* n, const char* name);
int mms_net_setup(mms_net_t* n);
float mms_net_forward(mms_net_t* n);
void mms_net_backward(mms_net_t* n);

/* Caffe::set_mode (include/caffe/common.hpp): 0 = CPU, 1 = GPU (default).
 * This library is GPU-only: Forward/Backward in CPU mode is a fatal error. */
void mms_caffe_set_mode(int gpu);
void mms_caffe_set_random_seed(unsigned seed);
/* Comma-separated registered layer types ("PairRankLoss,SimCross,SimMatrix"). */
const char* mms_layer_registry_types(void);

/* Layer<double> (the reference instantiates every layer for float and double, common.hpp:41-44):
 * SimCross, SimMatrix and PairRankLoss are registered for double as well.  The handle API above is
 * float; this one call creates a Layer<double> from prototxt, loads the bottoms (and optionally the
 * parameter blobs) from host arrays, runs SetUp / Forward / Backward and copies out top[0], the
 * bottom diffs and the parameter diffs (which start from the values passed in).  top_diff == NULL
 * means a loss layer (diff 1).  Returns 0, or non-zero with a message in err. */
int mms_layer_run_f64(const char* prototxt, int nbottom, const int* bottom_axes, const int* bottom_dims,
                      const double* const* bottom_data, int nparam, const double* const* param_data,
                      const double* top_diff, const int* propagate_down, double* top_out, long long top_capacity,
                      int* top_dims_out, int* top_axes_out, double* const* bottom_diff_out,
                      double* const* param_diff_out, char* err, int err_len);

/* ------------------------------------------------------------------------- *
 * Snapshots.  mms_snapshot_open reads a binary NetParameter (.caffemodel: the current
 * `layer` list, and the V1 `layers` / V0 lists of older files with their enum types mapped to type strings as
 * Caffe's upgrade does) or an HDF5-format snapshot (snapshot_format: HDF5, /data/<layer>/<index>; recognised
 * by its signature; such files carry no layer types).  Host-only except the two *_layer functions, which touch
 * Blob memory.
 * ------------------------------------------------------------------------- */
typedef struct mms_snapshot mms_snapshot_t;
typedef struct mms_snapshot_writer mms_snapshot_writer_t;

mms_snapshot_t* mms_snapshot_open(const char* path, char* err, int err_len);
void mms_snapshot_close(mms_snapshot_t* s);
const char* mms_snapshot_net_name(const mms_snapshot_t* s);
int mms_snapshot_num_layers(const mms_snapshot_t* s);
const char* mms_snapshot_layer_name(const mms_snapshot_t* s, int layer);
const char* mms_snapshot_layer_type(const mms_snapshot_t* s, int layer);
int mms_snapshot_num_blobs(const mms_snapshot_t* s, int layer);
int mms_snapshot_blob_shape(const mms_snapshot_t* s, int layer, int blob, int* shape, int max_axes);
int mms_snapshot_blob_count(const mms_snapshot_t* s, int layer, int blob);
const float* mms_snapshot_blob_data(const mms_snapshot_t* s, int layer, int blob);
/* Net::CopyTrainedLayersFrom for one layer (match by name; 0 ok, 1 name absent,
 * 2 blob-count mismatch, 3 shape mismatch). */
int mms_layer_copy_from_snapshot(mms_layer_t* layer, const mms_snapshot_t* s, const char* layer_name);

mms_snapshot_writer_t* mms_snapshot_writer_create(const char* net_name);
void mms_snapshot_writer_destroy(mms_snapshot_writer_t* w);
void mms_snapshot_writer_add_layer(mms_snapshot_writer_t* w, const char* name, const char* type);
void mms_snapshot_writer_add_blob(mms_snapshot_writer_t* w, const int* shape, int num_axes, const float* data);
/* Layer::ToProto (include/caffe/layer.hpp:506-514). */
void mms_snapshot_writer_add_from_layer(mms_snapshot_writer_t* w, mms_layer_t* layer, const char* name);
int mms_snapshot_writer_save(const mms_snapshot_writer_t* w, const char* path);
/* Net::ToHDF5 (net.cpp:893-960), data only; at most 8 parameter layers per file. */
int mms_snapshot_writer_save_hdf5(const mms_snapshot_writer_t* w, const char* path, char* err, int err_len);

/* ------------------------------------------------------------------------- *
 * HDF5 batch files, host only.  What the "HDF5Data" layer uses
 * in place of libhdf5's H5Fopen / H5LTfind_dataset / H5LTget_dataset_info /
 * H5LTread_dataset_float (src/caffe/util/hdf5.cpp:10-73): root-group datasets of
 * integer or float type, contiguous / compact / chunked (+deflate, shuffle,
 * fletcher32), converted to float.  type_class: 0 H5T_INTEGER, 1 H5T_FLOAT.
 * The writer emits what h5py's `f[name] = ndarray` does for float32 (elem_size 4)
 * and float64 (8) arrays -- the driver's format, do_trec_qa_clean.py:237-246.
 * ------------------------------------------------------------------------- */
typedef struct mms_h5_file mms_h5_file_t;
typedef struct mms_h5_writer mms_h5_writer_t;
mms_h5_file_t* mms_h5_open(const char* path, char* err, int err_len);
void mms_h5_close(mms_h5_file_t* h);
int mms_h5_num_datasets(const mms_h5_file_t* h);
const char* mms_h5_dataset_name(const mms_h5_file_t* h, int i);
/* returns the rank, or -1 with a message in err */
int mms_h5_dataset_info(const mms_h5_file_t* h, const char* name, long long* dims, int max_axes,
                        int* type_class, int* elem_size, char* err, int err_len);
int mms_h5_read_float(const mms_h5_file_t* h, const char* name, float* out, long long capacity,
                      char* err, int err_len);
mms_h5_writer_t* mms_h5_writer_create(void);
void mms_h5_writer_destroy(mms_h5_writer_t* w);
void mms_h5_writer_add(mms_h5_writer_t* w, const char* name, const long long* dims, int num_axes,
                       int elem_size, const double* values);
int mms_h5_writer_save(const mms_h5_writer_t* w, const char* path, char* err, int err_len);

#ifdef __cplusplus
}
#endif
#endif /* MMS_LAYER_H_ */
