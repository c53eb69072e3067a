// csrc/mms_abi.hip -- extern "C" entry points declared in include/mms.h:
// argument validation and dispatch only; kernels live in the sibling files.
#include "mms_common.h"

namespace mms {
// simcross_elementwise.hip
int simcross_elementwise_forward(int mode, int N, int W1, int W2, int D, const float* q,
                                 const float* a, float* top, float* norm0, float* norm1,
                                 hipStream_t s);
int simcross_elementwise_backward(int mode, int N, int W1, int W2, int D, const float* q,
                                  const float* a, const float* top, const float* top_diff,
                                  const float* norm0, const float* norm1, float* dq, float* da,
                                  hipStream_t s);
int simcross_elementwise_forward_backward(int mode, int N, int W1, int W2, int D, const float* q,
                                          const float* a, const float* top_diff, float* top,
                                          float* norm0, float* norm1, float* dq, float* da,
                                          hipStream_t s);
int simcross_euclid_rows_f16(int N, int D, const void* q, const void* a, const float* top_diff,
                             float* top, void* dq, void* da, bool bwd, hipStream_t s);
int simcross_cosine_rows_f16(int N, int D, const void* q, const void* a, const float* top_diff, float* top,
                             float* norm0, float* norm1, void* dq, void* da, bool bwd, hipStream_t s);
// bilinear.hip
size_t bilinear_workspace_bytes(int N, int W1, int W2, int D, int M);
int bilinear_forward(int N, int W1, int W2, int D, int M, const float* q, const float* a,
                     const float* W, const float* bias, float* top, void* ws, size_t ws_bytes,
                     hipStream_t s);
int bilinear_backward(int N, int W1, int W2, int D, int M, const float* q, const float* a,
                      const float* W, int bias_term, const float* top_diff, float* dq, float* da,
                      float* dW, float* dbias, void* ws, size_t ws_bytes, hipStream_t s);
int embed_bilinear_forward(int N, int W1, int W2, int D, int M, int K, const float* index_q,
                           const float* index_a, const float* table, const float* embed_bias, const float* W,
                           const float* bias, float* top, hipStream_t s);
size_t simmatrix_workspace_bytes(int N, int K1, int K2);
int simmatrix_forward(int N, int K1, int K2, const float* q, const float* a, const float* W,
                      float* top, float* qw, hipStream_t s, const float* rd_bias, void* ws = nullptr, size_t ws_bytes = 0);
int set_matrix_mode(int mode);
int get_matrix_mode();
int simmatrix_forward_f16(int N, int K1, int K2, const void* q, const void* a, const float* W, float* top, void* ws,
                          size_t ws_bytes, hipStream_t s);
int simmatrix_forward_train_f16(int N, int K1, int K2, const void* q, const void* a, const float* W, float* top, float* qw,
                                void* ws, size_t ws_bytes, hipStream_t s);
int simmatrix_backward_f16(int N, int K1, int K2, const void* q, const void* a, const float* W, const float* qw,
                           const float* top_diff, void* dq, void* da, float* dW, void* ws, size_t ws_bytes, hipStream_t s);
int embed_simcross_forward(int mode, int N, int W1, int W2, int D, int K, const float* index_q,
                           const float* index_a, const float* weight, const float* embed_bias, float* top,
                           float* norm0, float* norm1, hipStream_t s);
int simmatrix_backward(int N, int K1, int K2, const float* q, const float* a, const float* W,
                       const float* top_diff, int ppd, int pd0, int pd1, float* dq, float* da,
                       float* dW, const float* qw, void* ws, size_t ws_bytes, hipStream_t s);
size_t triplet_simmatrix_workspace_bytes(int N, int K1, int K2);
int triplet_simmatrix_step(int N, int K1, int K2, float margin, float loss_weight, const float* q, const float* ap,
                           const float* an, const float* y, const float* W, float* s_pos, float* s_neg, float* loss,
                           float* dq, float* dap, float* dan, float* dW, void* ws, size_t ws_bytes, hipStream_t s);
// pairrank.hip
size_t pairrank_workspace_bytes(int count);
int pairrank_forward(int count, float margin, const float* a, const float* b, const float* y,
                     float* ordered, float* similar, float* loss, void* ws, size_t ws_bytes,
                     hipStream_t s);
int pairrank_backward(int count, float top_diff, const float* y, const float* ordered,
                      const float* similar, float* da, float* db, hipStream_t s);
size_t triplet_workspace_bytes(int N);
int triplet_workspace_init(void* ws, size_t ws_bytes, hipStream_t s);
int triplet_euclid_step(int N, int D, float margin, float loss_weight, const float* q,
                        const float* ap, const float* an, const float* y, float* s_pos,
                        float* s_neg, float* loss, float* dq, float* dap, float* dan, void* ws,
                        size_t ws_bytes, hipStream_t s);
// ranking.hip
size_t rank_workspace_bytes(int n);
int rank_map_mrr(int n, int fixed_axis, const float* prob, const float* label, const float* group,
                 float* map_out, float* mrr_out, int* effective, void* ws, size_t ws_bytes,
                 hipStream_t s);
int rank_auc(int n, int dim, int fixed_axis, int inner, const float* prob, const float* label, int has_ignore,
             int ignore_label, float* auc_out, void* ws, size_t ws_bytes, hipStream_t s);
int rank_accuracy(int count, const float* a, const float* b, const float* label, float* out,
                  void* ws, size_t ws_bytes, hipStream_t s);
// embed.hip
size_t embed_workspace_bytes(int M, int N);
int embed_forward(int M, int N, int K, const float* index, const float* weight, const float* bias,
                  float* top, hipStream_t s);
int embed_backward(int M, int N, int K, const float* index, const float* top_diff,
                   float* weight_diff, float* bias_diff, void* ws, size_t ws_bytes, hipStream_t s);
int embed_backward_pair(int M0, int M1, int N, int K, const float* index0, const float* top_diff0, const float* index1,
                        const float* top_diff1, float* weight_diff, float* bias_diff, void* ws, size_t ws_bytes,
                        hipStream_t s, int index_ready);
bool embed_pair_index_supported(int M0, int M1, int K);
int embed_forward_pair(int M0, int M1, int N, int K, const float* index0, const float* index1, const float* weight,
                       const float* bias, float* top0, float* top1, void* index_ws, size_t index_ws_bytes, hipStream_t s);
int euclid_backward_mode();
void set_euclid_backward_mode(int m);
int pairrank_hinge_mode();
void set_pairrank_hinge_mode(int m);
int triplet_finish_mode();
void set_triplet_finish_mode(int m);
int loss_sum_mode();
void set_loss_sum_mode(int m);
int rank_tie_mode();
void set_rank_tie_mode(int m);
int f16_distance_mode();
void set_f16_distance_mode(int m);
// f64_paths.hip
size_t simcross_workspace_bytes_f64(int mode, int N, int W1, int W2, int D, int M);
int simcross_forward_f64(int mode, int N, int W1, int W2, int D, int M, const double* q, const double* a,
                         const double* W, const double* bias, double* top, double* norm0, double* norm1,
                         void* ws, size_t ws_bytes, hipStream_t s);
int simcross_backward_f64(int mode, int N, int W1, int W2, int D, int M, const double* q, const double* a,
                          const double* W, int bias_term, const double* top, const double* top_diff,
                          const double* norm0, const double* norm1, int pd0, int pd1, double* dq, double* da,
                          double* dW, double* dbias, void* ws, size_t ws_bytes, hipStream_t s);
int simmatrix_forward_f64(int N, int K1, int K2, const double* q, const double* a, const double* W,
                          double* top, double* scratch, hipStream_t s);
int simmatrix_backward_f64(int N, int K1, int K2, const double* q, const double* a, const double* W,
                           const double* top_diff, int ppd, int pd0, int pd1, double* dq, double* da,
                           double* dW, hipStream_t s);
int pairrank_forward_f64(int count, double margin, const double* a, const double* b, const double* y,
                         double* ordered, double* similar, double* loss, hipStream_t s);
int pairrank_backward_f64(int count, double top_diff, const double* y, const double* ordered,
                          const double* similar, double* da, double* db, hipStream_t s);
int feed_gather_rows(int rows, int row_elems, int src_rows, const float* src, const int* perm, int first,
                     float* dst, hipStream_t s);
}  // namespace mms

using namespace mms;

namespace {
bool dims_ok(int mode, int N, int W1, int W2, int D, int M) {
  if (mode < 0 || mode > 2) return false;
  if (N < 0 || W1 <= 0 || W2 <= 0 || D <= 0 || M <= 0) return false;
  if (mode != 2 && M != 1) return false;
  // element counts must fit the int indexing the reference itself uses
  const long long lim = 0x7fffffffLL;
  if ((long long)N * W1 * D > lim || (long long)N * W2 * D > lim ||
      (long long)N * M * W1 * W2 > lim)
    return false;
  return true;
}
hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }
}  // namespace

// x . y with the result left on the device (fixed-shape tree: deterministic); what Layer::Forward needs for a
// loss top in GPU mode (include/caffe/layer.hpp:469-481 calls caffe_gpu_dot there)
namespace mms {
// measurement aid (include/mms.h: mms_null_launch)
__global__ __launch_bounds__(512) void null_kernel(int) {}

template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(int n, const T* __restrict__ x, const T* __restrict__ y,
                                                  T* __restrict__ out) {
  __shared__ T red[256];
  T s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i] * y[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}
struct SplitPtrs { const float* p[8]; };
// out = ((carry? out : p0) + p1) + ... in index order
__global__ __launch_bounds__(256) void split_sum_kernel(int n, int k, SplitPtrs s, int carry, float* __restrict__ out) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float v = carry ? out[i] : s.p[0][i];
    for (int j = carry ? 0 : 1; j < k; ++j) v = v + s.p[j][i];
    out[i] = v;
  }
}
}  // namespace mms

extern "C" {

int mms_version(void) { return MMS_VERSION; }

const char* mms_error_string(int code) {
  switch (code) {
    case MMS_OK: return "ok";
    case MMS_ERR_INVALID_ARG: return "invalid argument";
    case MMS_ERR_UNSUPPORTED: return "unsupported configuration";
    case MMS_ERR_WORKSPACE: return "workspace missing or too small";
    case MMS_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown error";
  }
}

size_t mms_simcross_workspace_bytes(int dist_mode, int N, int W1, int W2, int D, int M) {
  if (!dims_ok(dist_mode, N, W1, W2, D, M)) return 0;
  if (dist_mode == 2) return bilinear_workspace_bytes(N, W1, W2, D, M);
  return 0;
}

int mms_simcross_forward_f32(int dist_mode, int N, int W1, int W2, int D, int M, const float* q,
                             const float* a, const float* W, const float* bias, float* top,
                             float* norm0, float* norm1, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (!dims_ok(dist_mode, N, W1, W2, D, M)) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !top) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 0 && (!norm0 || !norm1)) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 2) {
    if (!W) return MMS_ERR_INVALID_ARG;
    return bilinear_forward(N, W1, W2, D, M, q, a, W, bias, top, workspace, workspace_bytes,
                            as_stream(stream));
  }
  return simcross_elementwise_forward(dist_mode, N, W1, W2, D, q, a, top, norm0, norm1,
                                      as_stream(stream));
}

int mms_simcross_backward_f32(int dist_mode, int N, int W1, int W2, int D, int M, const float* q,
                              const float* a, const float* W, int bias_term, const float* top,
                              const float* top_diff, const float* norm0, const float* norm1,
                              int propagate_down0, int propagate_down1, float* dq, float* da,
                              float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                              void* stream) {
  if (!dims_ok(dist_mode, N, W1, W2, D, M)) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !top || !top_diff || !dq || !da) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 0 && (!norm0 || !norm1)) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 2 && (!W || !dW || (bias_term && !dbias))) return MMS_ERR_INVALID_ARG;
  hipStream_t s = as_stream(stream);
  if (!(propagate_down0 || propagate_down1)) {
    // sim_cross_layer.cpp:176-177 then nothing else (:201)
    if (hipMemsetAsync(dq, 0, sizeof(float) * (size_t)N * W1 * D, s) != hipSuccess ||
        hipMemsetAsync(da, 0, sizeof(float) * (size_t)N * W2 * D, s) != hipSuccess)
      return MMS_ERR_LAUNCH;
    return MMS_OK;
  }
  if (dist_mode == 2)
    return bilinear_backward(N, W1, W2, D, M, q, a, W, bias_term, top_diff, dq, da, dW, dbias,
                             workspace, workspace_bytes, s);
  return simcross_elementwise_backward(dist_mode, N, W1, W2, D, q, a, top, top_diff, norm0, norm1,
                                       dq, da, s);
}

int mms_simcross_forward_block_f32(const mms_simcross_args_f32* p, void* stream) {
  if (!p) return MMS_ERR_INVALID_ARG;
  return mms_simcross_forward_f32(p->dist_mode, p->N, p->W1, p->W2, p->D, p->M, p->q, p->a, p->W, p->bias, p->top,
                                  p->norm0, p->norm1, p->workspace, p->workspace_bytes, stream);
}
int mms_simcross_backward_block_f32(const mms_simcross_args_f32* p, void* stream) {
  if (!p) return MMS_ERR_INVALID_ARG;
  return mms_simcross_backward_f32(p->dist_mode, p->N, p->W1, p->W2, p->D, p->M, p->q, p->a, p->W, p->bias_term,
                                   p->top, p->top_diff, p->norm0, p->norm1, p->propagate_down0, p->propagate_down1,
                                   p->dq, p->da, p->dW, p->dbias, p->workspace, p->workspace_bytes, stream);
}

int mms_simcross_forward_backward_f32(int dist_mode, int N, int W1, int W2, int D, int M,
                                      const float* q, const float* a, const float* W,
                                      const float* bias, const float* top_diff, float* top,
                                      float* norm0, float* norm1, float* dq, float* da, float* dW,
                                      float* dbias, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (!dims_ok(dist_mode, N, W1, W2, D, M)) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !top || !top_diff || !dq || !da) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 0 && (!norm0 || !norm1)) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 2) {
    int rc = mms_simcross_forward_f32(2, N, W1, W2, D, M, q, a, W, bias, top, norm0, norm1,
                                      workspace, workspace_bytes, stream);
    if (rc != MMS_OK) return rc;
    return mms_simcross_backward_f32(2, N, W1, W2, D, M, q, a, W, bias != nullptr, top, top_diff,
                                     norm0, norm1, 1, 1, dq, da, dW, dbias, workspace,
                                     workspace_bytes, stream);
  }
  return simcross_elementwise_forward_backward(dist_mode, N, W1, W2, D, q, a, top_diff, top, norm0,
                                               norm1, dq, da, as_stream(stream));
}

int mms_simcross_euclid_forward_f16(int N, int D, const void* q_f16, const void* a_f16, float* top,
                                    void* stream) {
  if (N < 0 || D <= 0 || (long long)N * D > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q_f16 || !a_f16 || !top) return MMS_ERR_INVALID_ARG;
  return simcross_euclid_rows_f16(N, D, q_f16, a_f16, nullptr, top, nullptr, nullptr, false,
                                  as_stream(stream));
}

int mms_simcross_cosine_forward_f16(int N, int D, const void* q_f16, const void* a_f16, float* top, float* norm0,
                                    float* norm1, void* stream) {
  if (N < 0 || D <= 0 || (long long)N * D > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q_f16 || !a_f16 || !top) return MMS_ERR_INVALID_ARG;
  return simcross_cosine_rows_f16(N, D, q_f16, a_f16, nullptr, top, norm0, norm1, nullptr, nullptr, false, as_stream(stream));
}

int mms_simcross_cosine_forward_backward_f16(int N, int D, const void* q_f16, const void* a_f16, const float* top_diff,
                                             float* top, float* norm0, float* norm1, void* dq_f16, void* da_f16,
                                             void* stream) {
  if (N < 0 || D <= 0 || (long long)N * D > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q_f16 || !a_f16 || !top_diff || !top || !dq_f16 || !da_f16) return MMS_ERR_INVALID_ARG;
  return simcross_cosine_rows_f16(N, D, q_f16, a_f16, top_diff, top, norm0, norm1, dq_f16, da_f16, true, as_stream(stream));
}

int mms_simcross_euclid_forward_backward_f16(int N, int D, const void* q_f16, const void* a_f16,
                                             const float* top_diff, float* top, void* dq_f16,
                                             void* da_f16, void* stream) {
  if (N < 0 || D <= 0 || (long long)N * D > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q_f16 || !a_f16 || !top_diff || !top || !dq_f16 || !da_f16) return MMS_ERR_INVALID_ARG;
  return simcross_euclid_rows_f16(N, D, q_f16, a_f16, top_diff, top, dq_f16, da_f16, true,
                                  as_stream(stream));
}

int mms_embed_simcross_forward_f32(int dist_mode, int N, int W1, int W2, int D, int K,
                                   const float* index_q, const float* index_a, const float* weight,
                                   const float* embed_bias, float* top, float* norm0, float* norm1, void* stream) {
  if (dist_mode != 0 && dist_mode != 1) return MMS_ERR_UNSUPPORTED;   // bilinear: run Embed, then SimCross
  if (!dims_ok(dist_mode, N, W1, W2, D, 1) || K <= 0 || (long long)K * D > 0x7fffffffLL)
    return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!index_q || !index_a || !weight || !top) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 0 && (!norm0 || !norm1)) return MMS_ERR_INVALID_ARG;
  return embed_simcross_forward(dist_mode, N, W1, W2, D, K, index_q, index_a, weight, embed_bias, top, norm0,
                                norm1, as_stream(stream));
}

int mms_embed_simcross_bilinear_forward_f32(int N, int W1, int W2, int D, int M, int K,
                                            const float* index_q, const float* index_a, const float* weight,
                                            const float* embed_bias, const float* W, const float* bias,
                                            float* top, void* stream) {
  if (!dims_ok(2, N, W1, W2, D, M) || K <= 0 || (long long)K * D > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!index_q || !index_a || !weight || !W || !top) return MMS_ERR_INVALID_ARG;
  return embed_bilinear_forward(N, W1, W2, D, M, K, index_q, index_a, weight, embed_bias, W, bias, top,
                                as_stream(stream));
}

size_t mms_simmatrix_workspace_bytes(int N, int K1, int K2) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return 0;
  return simmatrix_workspace_bytes(N, K1, K2);
}

int mms_simmatrix_forward_f32(int N, int K1, int K2, const float* q, const float* a,
                              const float* W, float* top, float* qw_scratch, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !W || !top || !qw_scratch) return MMS_ERR_INVALID_ARG;
  return simmatrix_forward(N, K1, K2, q, a, W, top, qw_scratch, as_stream(stream), nullptr);
}

int mms_simmatrix_forward_ws_f32(int N, int K1, int K2, const float* q, const float* a, const float* W, float* top,
                                 float* qw_scratch, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !W || !top || !qw_scratch) return MMS_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < simmatrix_workspace_bytes(N, K1, K2)) return MMS_ERR_WORKSPACE;
  return simmatrix_forward(N, K1, K2, q, a, W, top, qw_scratch, as_stream(stream), nullptr, workspace, workspace_bytes);
}

int mms_simmatrix_forward_f16(int N, int K1, int K2, const void* q_f16, const void* a_f16, const float* W, float* top,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q_f16 || !a_f16 || !W || !top) return MMS_ERR_INVALID_ARG;
  return simmatrix_forward_f16(N, K1, K2, q_f16, a_f16, W, top, workspace, workspace_bytes, as_stream(stream));
}

int mms_simmatrix_forward_train_f16(int N, int K1, int K2, const void* q_f16, const void* a_f16, const float* W, float* top,
                                    float* qw_scratch, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q_f16 || !a_f16 || !W || !top || !qw_scratch) return MMS_ERR_INVALID_ARG;
  return simmatrix_forward_train_f16(N, K1, K2, q_f16, a_f16, W, top, qw_scratch, workspace, workspace_bytes, as_stream(stream));
}

int mms_simmatrix_backward_f16(int N, int K1, int K2, const void* q_f16, const void* a_f16, const float* W, const float* qw,
                               const float* top_diff, void* dq_f16, void* da_f16, float* dW, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q_f16 || !a_f16 || !W || !top_diff) return MMS_ERR_INVALID_ARG;
  return simmatrix_backward_f16(N, K1, K2, q_f16, a_f16, W, qw, top_diff, dq_f16, da_f16, dW, workspace, workspace_bytes,
                                as_stream(stream));
}

int mms_set_matrix_mode(int mode) { return set_matrix_mode(mode); }
int mms_get_matrix_mode(void) { return get_matrix_mode(); }

int mms_simmatrix_backward_f32(int N, int K1, int K2, const float* q, const float* a,
                               const float* W, const float* top_diff, int param_propagate_down,
                               int propagate_down0, int propagate_down1, float* dq, float* da,
                               float* dW, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !W || !top_diff) return MMS_ERR_INVALID_ARG;
  if ((param_propagate_down && !dW) || (propagate_down0 && !dq) || (propagate_down1 && !da))
    return MMS_ERR_INVALID_ARG;
  return simmatrix_backward(N, K1, K2, q, a, W, top_diff, param_propagate_down, propagate_down0,
                            propagate_down1, dq, da, dW, nullptr, workspace, workspace_bytes,
                            as_stream(stream));
}

int mms_simmatrix_backward_cached_f32(int N, int K1, int K2, const float* q, const float* a,
                                      const float* W, const float* qw, const float* top_diff,
                                      int param_propagate_down, int propagate_down0,
                                      int propagate_down1, float* dq, float* da, float* dW,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !W || !qw || !top_diff) return MMS_ERR_INVALID_ARG;
  if ((param_propagate_down && !dW) || (propagate_down0 && !dq) || (propagate_down1 && !da))
    return MMS_ERR_INVALID_ARG;
  return simmatrix_backward(N, K1, K2, q, a, W, top_diff, param_propagate_down, propagate_down0,
                            propagate_down1, dq, da, dW, qw, workspace, workspace_bytes,
                            as_stream(stream));
}

size_t mms_pairrank_workspace_bytes(int count) {
  return count > 0 ? pairrank_workspace_bytes(count) : 0;
}

int mms_pairrank_forward_f32(int count, float margin, const float* a, const float* b,
                             const float* y, float* ordered, float* similar, float* loss,
                             void* workspace, size_t workspace_bytes, void* stream) {
  // count == 0 would be 0/0 in the reference (:49); reject instead of writing NaN.
  if (count <= 0) return MMS_ERR_INVALID_ARG;
  if (!a || !b || !y || !ordered || !similar || !loss) return MMS_ERR_INVALID_ARG;
  return pairrank_forward(count, margin, a, b, y, ordered, similar, loss, workspace,
                          workspace_bytes, as_stream(stream));
}

int mms_pairrank_backward_f32(int count, float top_diff, const float* y, const float* ordered,
                              const float* similar, int propagate_down0, int propagate_down1,
                              float* da, float* db, void* stream) {
  if (count <= 0) return MMS_ERR_INVALID_ARG;
  if (!y || !ordered || !similar) return MMS_ERR_INVALID_ARG;
  if ((propagate_down0 && !da) || (propagate_down1 && !db)) return MMS_ERR_INVALID_ARG;
  return pairrank_backward(count, top_diff, y, ordered, similar, propagate_down0 ? da : nullptr,
                           propagate_down1 ? db : nullptr, as_stream(stream));
}

// ---- fused learned-metric triplet step ----------------------------------------------------------------------------
namespace {
struct TripSimSlow {                                  // the layers one by one, inside the call (shapes outside the fast path)
  size_t qwp, qwn, ord, sim, gsp, gsn, dq2, lossf, prws, smws, total;
};
TripSimSlow tripsim_slow_layout(int N, int K1, int K2) {
  TripSimSlow w{};
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += mms::round_up(b, 256); return at; };
  w.qwp = take((size_t)N * K2 * 4); w.qwn = take((size_t)N * K2 * 4);
  w.ord = take((size_t)N * 4); w.sim = take((size_t)N * 4); w.gsp = take((size_t)N * 4); w.gsn = take((size_t)N * 4);
  w.dq2 = take((size_t)N * K1 * 4); w.lossf = take(256);
  w.prws = take(mms::pairrank_workspace_bytes(N)); w.smws = take(mms::simmatrix_workspace_bytes(N, K1, K2));
  w.total = o;
  return w;
}
}  // namespace

size_t mms_triplet_simmatrix_workspace_bytes(int N, int K1, int K2) {
  if (N <= 0 || K1 <= 0 || K2 <= 0) return 0;
  const size_t fast = mms::triplet_simmatrix_workspace_bytes(N, K1, K2), slow = tripsim_slow_layout(N, K1, K2).total;
  return fast > slow ? fast : slow;
}

int mms_triplet_simmatrix_step_f32(int N, int K1, int K2, float margin, float loss_weight, const float* q,
                                   const float* a_pos, const float* a_neg, const float* y, const float* W,
                                   float* s_pos, float* s_neg, float* loss, float* dq, float* da_pos, float* da_neg,
                                   float* dW, void* workspace, size_t workspace_bytes, void* stream) {
  if (N <= 0 || K1 <= 0 || K2 <= 0 || (long long)N * K1 > 0x7fffffffLL || (long long)N * K2 > 0x7fffffffLL)
    return MMS_ERR_INVALID_ARG;
  if (!q || !a_pos || !a_neg || !y || !W || !s_pos || !s_neg || !dq || !da_pos || !da_neg || !dW)   // loss may be NULL
    return MMS_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < mms_triplet_simmatrix_workspace_bytes(N, K1, K2)) return MMS_ERR_WORKSPACE;
  hipStream_t s = as_stream(stream);
  const int rc = mms::triplet_simmatrix_step(N, K1, K2, margin, loss_weight, q, a_pos, a_neg, y, W, s_pos, s_neg, loss,
                                             dq, da_pos, da_neg, dW, workspace, workspace_bytes, s);
  if (rc != MMS_ERR_UNSUPPORTED) return rc;
  // layer by layer: SimMatrix x 2 -> PairRankLoss -> PairRankLoss backward -> SimMatrix backward x 2 -> Split sum
  const TripSimSlow lay = tripsim_slow_layout(N, K1, K2);
  char* base = static_cast<char*>(workspace);
  auto f = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  int r = mms::simmatrix_forward(N, K1, K2, q, a_pos, W, s_pos, f(lay.qwp), s, nullptr);
  if (r == MMS_OK) r = mms::simmatrix_forward(N, K1, K2, q, a_neg, W, s_neg, f(lay.qwn), s, nullptr);
  if (r == MMS_OK) r = mms::pairrank_forward(N, margin, s_pos, s_neg, y, f(lay.ord), f(lay.sim), loss ? loss : f(lay.lossf),
                                            base + lay.prws, mms::pairrank_workspace_bytes(N), s);
  if (r == MMS_OK) r = mms::pairrank_backward(N, loss_weight, y, f(lay.ord), f(lay.sim), f(lay.gsp), f(lay.gsn), s);
  if (r == MMS_OK) r = mms::simmatrix_backward(N, K1, K2, q, a_pos, W, f(lay.gsp), 1, 1, 1, dq, da_pos, dW, f(lay.qwp),
                                              base + lay.smws, mms::simmatrix_workspace_bytes(N, K1, K2), s);
  if (r == MMS_OK) r = mms::simmatrix_backward(N, K1, K2, q, a_neg, W, f(lay.gsn), 1, 1, 1, f(lay.dq2), da_neg, dW, f(lay.qwn),
                                              base + lay.smws, mms::simmatrix_workspace_bytes(N, K1, K2), s);
  if (r != MMS_OK) return r;
  const float* two[2] = {dq, f(lay.dq2)};
  return mms_split_backward_f32(N * K1, 2, two, dq, stream);
}

size_t mms_triplet_workspace_bytes(int N) { return N > 0 ? triplet_workspace_bytes(N) : 0; }

int mms_triplet_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
  return triplet_workspace_init(workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int mms_triplet_euclid_step_f32(int N, int D, float margin, float loss_weight, const float* q,
                                const float* a_pos, const float* a_neg, const float* y,
                                float* s_pos, float* s_neg, float* loss, float* dq, float* da_pos,
                                float* da_neg, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (N <= 0 || D <= 0 || (long long)N * D > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (!q || !a_pos || !a_neg || !y || !s_pos || !s_neg || !dq || !da_pos || !da_neg)   // loss may be NULL
    return MMS_ERR_INVALID_ARG;
  return triplet_euclid_step(N, D, margin, loss_weight, q, a_pos, a_neg, y, s_pos, s_neg, loss,
                             dq, da_pos, da_neg, workspace, workspace_bytes, as_stream(stream));
}

size_t mms_rank_workspace_bytes(int n) { return n > 0 ? rank_workspace_bytes(n) : 0; }

int mms_rank_map_mrr_f32(int n, int fixed_axis, const float* prob, const float* label,
                         const float* group, float* map_out, float* mrr_out, int* effective_out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || fixed_axis < 0 || (long long)n * (fixed_axis + 1) > 0x7fffffffLL)
    return MMS_ERR_INVALID_ARG;
  if (!prob || !label || !group) return MMS_ERR_INVALID_ARG;
  return rank_map_mrr(n, fixed_axis, prob, label, group, map_out, mrr_out, effective_out,
                      workspace, workspace_bytes, as_stream(stream));
}

int mms_rank_auc_f32(int n, int dim, int fixed_axis, const float* prob, const float* label,
                     int has_ignore_label, int ignore_label, float* auc_out, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (n <= 0 || dim <= 0 || fixed_axis < 0 || fixed_axis >= dim ||
      (long long)n * dim > 0x7fffffffLL)
    return MMS_ERR_INVALID_ARG;
  if (!prob || !label || !auc_out) return MMS_ERR_INVALID_ARG;
  return rank_auc(n, dim, fixed_axis, 1, prob, label, has_ignore_label, ignore_label, auc_out,
                  workspace, workspace_bytes, as_stream(stream));
}

int mms_rank_auc_nd_f32(int outer, int channels, int inner, int fixed_axis, const float* prob,
                        const float* label, int has_ignore_label, int ignore_label, float* auc_out,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (outer <= 0 || channels <= 0 || inner <= 0 || fixed_axis < 0 || fixed_axis >= channels ||
      (long long)outer * channels * inner > 0x7fffffffLL)
    return MMS_ERR_INVALID_ARG;
  if (!prob || !label || !auc_out) return MMS_ERR_INVALID_ARG;
  return rank_auc(outer * inner, channels * inner, fixed_axis, inner, prob, label, has_ignore_label, ignore_label,
                  auc_out, workspace, workspace_bytes, as_stream(stream));
}

int mms_rank_accuracy_f32(int count, const float* a, const float* b, const float* label,
                          float* acc_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (count <= 0) return MMS_ERR_INVALID_ARG;
  if (!a || !b || !label || !acc_out) return MMS_ERR_INVALID_ARG;
  return rank_accuracy(count, a, b, label, acc_out, workspace, workspace_bytes, as_stream(stream));
}

int mms_embed_backward_pair_f32(int M0, int M1, int N, int K, const float* index0, const float* top_diff0,
                                const float* index1, const float* top_diff1, float* weight_diff, float* bias_diff,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (M0 <= 0 || M1 <= 0 || N <= 0 || K <= 0 || !index0 || !top_diff0 || !index1 || !top_diff1) return MMS_ERR_INVALID_ARG;
  if ((long long)M0 + M1 > 0x7fffffffLL || ((long long)M0 + M1) * N > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (!weight_diff && !bias_diff) return MMS_OK;
  return embed_backward_pair(M0, M1, N, K, index0, top_diff0, index1, top_diff1, weight_diff, bias_diff, workspace,
                             workspace_bytes, as_stream(stream), 0);
}

int mms_embed_pair_index_supported(int M0, int M1, int K) {
  return M0 > 0 && M1 > 0 && K > 0 && embed_pair_index_supported(M0, M1, K) ? 1 : 0;
}

int mms_embed_forward_pair_f32(int M0, int M1, int N, int K, const float* index0, const float* index1,
                               const float* weight, const float* bias, float* top0, float* top1, void* index_workspace,
                               size_t index_workspace_bytes, void* stream) {
  if (M0 <= 0 || M1 <= 0 || N <= 0 || K <= 0 || !index0 || !index1 || !weight || !top0 || !top1) return MMS_ERR_INVALID_ARG;
  if (((long long)M0 + M1) * N > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  return embed_forward_pair(M0, M1, N, K, index0, index1, weight, bias, top0, top1, index_workspace,
                            index_workspace_bytes, as_stream(stream));
}

int mms_embed_backward_pair_indexed_f32(int M0, int M1, int N, int K, const float* index0, const float* top_diff0,
                                        const float* index1, const float* top_diff1, float* weight_diff, float* bias_diff,
                                        void* index_workspace, size_t index_workspace_bytes, void* stream) {
  if (M0 <= 0 || M1 <= 0 || N <= 0 || K <= 0 || !index0 || !top_diff0 || !index1 || !top_diff1) return MMS_ERR_INVALID_ARG;
  if (((long long)M0 + M1) * N > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (!embed_pair_index_supported(M0, M1, K)) return MMS_ERR_UNSUPPORTED;     // no index was built for these sizes
  if (!weight_diff && !bias_diff) return MMS_OK;
  return embed_backward_pair(M0, M1, N, K, index0, top_diff0, index1, top_diff1, weight_diff, bias_diff, index_workspace,
                             index_workspace_bytes, as_stream(stream), 1);
}

size_t mms_embed_workspace_bytes(int M, int N) { return (M > 0 && N > 0) ? embed_workspace_bytes(M, N) : 0; }

int mms_embed_forward_f32(int M, int N, int K, const float* index, const float* weight,
                          const float* bias, float* top, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || (long long)M * N > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (M == 0) return MMS_OK;
  if (!index || !weight || !top) return MMS_ERR_INVALID_ARG;
  return embed_forward(M, N, K, index, weight, bias, top, as_stream(stream));
}

int mms_embed_backward_f32(int M, int N, int K, const float* index, const float* top_diff,
                           float* weight_diff, float* bias_diff, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || (long long)M * N > 0x7fffffffLL) return MMS_ERR_INVALID_ARG;
  if (M == 0 || (!weight_diff && !bias_diff)) return MMS_OK;
  if (!index || !top_diff) return MMS_ERR_INVALID_ARG;
  return embed_backward(M, N, K, index, top_diff, weight_diff, bias_diff, workspace, workspace_bytes,
                        as_stream(stream));
}

int mms_feed_gather_rows_f32(int rows, int row_elems, int src_rows, const float* src, const int* perm,
                             int first, float* dst, void* stream) {
  if (rows < 0 || row_elems <= 0 || src_rows <= 0 || first < 0) return MMS_ERR_INVALID_ARG;
  if ((long long)rows * row_elems > 0x7fffffffLL || (long long)first + rows > src_rows) return MMS_ERR_INVALID_ARG;
  if (rows == 0) return MMS_OK;
  if (!src || !dst) return MMS_ERR_INVALID_ARG;
  return feed_gather_rows(rows, row_elems, src_rows, src, perm, first, dst, as_stream(stream));
}

int mms_set_euclid_backward_mode(int mode) {
  if (mode != MMS_EUCLID_BWD_FP32 && mode != MMS_EUCLID_BWD_REFERENCE) return MMS_ERR_INVALID_ARG;
  set_euclid_backward_mode(mode);
  return MMS_OK;
}
int mms_get_euclid_backward_mode(void) { return euclid_backward_mode(); }

int mms_null_launch(int workgroups, void* stream) {
  if (workgroups <= 0) return MMS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mms::null_kernel, dim3(workgroups), dim3(512), 0, as_stream(stream), workgroups);
  return launch_status();
}
int mms_dot_f32(int n, const float* x, const float* y, float* out, void* stream) {
  if (n < 0 || !out || (n > 0 && (!x || !y))) return MMS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mms::dot_kernel<float>, dim3(1), dim3(256), 0, as_stream(stream), n, x, y, out);
  return launch_status();
}
int mms_split_backward_f32(int count, int ntop, const float* const* top_diffs, float* bottom_diff, void* stream) {
  if (count < 0 || ntop < 1 || !top_diffs || (count > 0 && !bottom_diff)) return MMS_ERR_INVALID_ARG;
  for (int i = 0; i < ntop; ++i) if (count > 0 && !top_diffs[i]) return MMS_ERR_INVALID_ARG;
  if (count == 0) return MMS_OK;
  const unsigned grid = (unsigned)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  for (int first = 0; first < ntop; first += 8) {
    mms::SplitPtrs s{};
    const int k = ntop - first < 8 ? ntop - first : 8;
    for (int j = 0; j < k; ++j) s.p[j] = top_diffs[first + j];
    hipLaunchKernelGGL(mms::split_sum_kernel, dim3(grid), dim3(256), 0, as_stream(stream), count, k, s, first > 0 ? 1 : 0,
                       bottom_diff);
  }
  return launch_status();
}
int mms_dot_f64(int n, const double* x, const double* y, double* out, void* stream) {
  if (n < 0 || !out || (n > 0 && (!x || !y))) return MMS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mms::dot_kernel<double>, dim3(1), dim3(256), 0, as_stream(stream), n, x, y, out);
  return launch_status();
}

int mms_set_pairrank_hinge_mode(int mode) {
  if (mode != MMS_PAIRRANK_HINGE_CPU && mode != MMS_PAIRRANK_HINGE_GPU) return MMS_ERR_INVALID_ARG;
  set_pairrank_hinge_mode(mode);
  return MMS_OK;
}
int mms_get_pairrank_hinge_mode(void) { return pairrank_hinge_mode(); }
int mms_set_triplet_finish_mode(int mode) {
  if (mode != MMS_TRIPLET_FINISH_LAUNCH && mode != MMS_TRIPLET_FINISH_INLAUNCH) return MMS_ERR_INVALID_ARG;
  set_triplet_finish_mode(mode);
  return MMS_OK;
}
int mms_get_triplet_finish_mode(void) { return triplet_finish_mode(); }

int mms_set_loss_sum_mode(int mode) {
  if (mode != MMS_LOSS_SUM_FAST && mode != MMS_LOSS_SUM_REFERENCE) return MMS_ERR_INVALID_ARG;
  set_loss_sum_mode(mode);
  return MMS_OK;
}
int mms_get_loss_sum_mode(void) { return loss_sum_mode(); }

int mms_set_rank_tie_mode(int mode) {
  if (mode != MMS_RANK_TIES_INPUT_ORDER && mode != MMS_RANK_TIES_LIBSTDCXX) return MMS_ERR_INVALID_ARG;
  set_rank_tie_mode(mode);
  return MMS_OK;
}
int mms_get_rank_tie_mode(void) { return rank_tie_mode(); }
int mms_set_f16_distance_mode(int mode) {
  if (mode != MMS_F16_DISTANCE_ORDERED && mode != MMS_F16_DISTANCE_TREE) return MMS_ERR_INVALID_ARG;
  set_f16_distance_mode(mode);
  return MMS_OK;
}
int mms_get_f16_distance_mode(void) { return f16_distance_mode(); }

// ---- double instantiation (csrc/f64_paths.hip): same contracts as the _f32 entry points ----
size_t mms_simcross_workspace_bytes_f64(int dist_mode, int N, int W1, int W2, int D, int M) {
  if (!dims_ok(dist_mode, N, W1, W2, D, M)) return 0;
  return simcross_workspace_bytes_f64(dist_mode, N, W1, W2, D, M);
}

int mms_simcross_forward_f64(int dist_mode, int N, int W1, int W2, int D, int M, const double* q,
                             const double* a, const double* W, const double* bias, double* top,
                             double* norm0, double* norm1, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (!dims_ok(dist_mode, N, W1, W2, D, M)) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !top) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 0 && (!norm0 || !norm1)) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 2 && !W) return MMS_ERR_INVALID_ARG;
  return simcross_forward_f64(dist_mode, N, W1, W2, D, M, q, a, W, bias, top, norm0, norm1, workspace,
                              workspace_bytes, as_stream(stream));
}

int mms_simcross_backward_f64(int dist_mode, int N, int W1, int W2, int D, int M, const double* q,
                              const double* a, const double* W, int bias_term, const double* top,
                              const double* top_diff, const double* norm0, const double* norm1,
                              int propagate_down0, int propagate_down1, double* dq, double* da,
                              double* dW, double* dbias, void* workspace, size_t workspace_bytes,
                              void* stream) {
  if (!dims_ok(dist_mode, N, W1, W2, D, M)) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !top || !top_diff || !dq || !da) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 0 && (!norm0 || !norm1)) return MMS_ERR_INVALID_ARG;
  if (dist_mode == 2 && (!W || !dW || (bias_term && !dbias))) return MMS_ERR_INVALID_ARG;
  return simcross_backward_f64(dist_mode, N, W1, W2, D, M, q, a, W, bias_term, top, top_diff, norm0,
                               norm1, propagate_down0, propagate_down1, dq, da, dW, dbias, workspace,
                               workspace_bytes, as_stream(stream));
}

int mms_simmatrix_forward_f64(int N, int K1, int K2, const double* q, const double* a,
                              const double* W, double* top, double* qw_scratch, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !W || !top || !qw_scratch) return MMS_ERR_INVALID_ARG;
  return simmatrix_forward_f64(N, K1, K2, q, a, W, top, qw_scratch, as_stream(stream));
}

int mms_simmatrix_backward_f64(int N, int K1, int K2, const double* q, const double* a,
                               const double* W, const double* top_diff, int param_propagate_down,
                               int propagate_down0, int propagate_down1, double* dq, double* da,
                               double* dW, void* stream) {
  if (N < 0 || K1 <= 0 || K2 <= 0) return MMS_ERR_INVALID_ARG;
  if (N == 0) return MMS_OK;
  if (!q || !a || !W || !top_diff) return MMS_ERR_INVALID_ARG;
  if ((param_propagate_down && !dW) || (propagate_down0 && !dq) || (propagate_down1 && !da))
    return MMS_ERR_INVALID_ARG;
  return simmatrix_backward_f64(N, K1, K2, q, a, W, top_diff, param_propagate_down, propagate_down0,
                                propagate_down1, dq, da, dW, as_stream(stream));
}

int mms_pairrank_forward_f64(int count, double margin, const double* a, const double* b,
                             const double* y, double* ordered, double* similar, double* loss,
                             void* stream) {
  if (count <= 0) return MMS_ERR_INVALID_ARG;
  if (!a || !b || !y || !ordered || !similar || !loss) return MMS_ERR_INVALID_ARG;
  return pairrank_forward_f64(count, margin, a, b, y, ordered, similar, loss, as_stream(stream));
}

int mms_pairrank_backward_f64(int count, double top_diff, const double* y, const double* ordered,
                              const double* similar, int propagate_down0, int propagate_down1,
                              double* da, double* db, void* stream) {
  if (count <= 0) return MMS_ERR_INVALID_ARG;
  if (!y || !ordered || !similar) return MMS_ERR_INVALID_ARG;
  if ((propagate_down0 && !da) || (propagate_down1 && !db)) return MMS_ERR_INVALID_ARG;
  return pairrank_backward_f64(count, top_diff, y, ordered, similar, propagate_down0 ? da : nullptr,
                               propagate_down1 ? db : nullptr, as_stream(stream));
}

}  // extern "C"
