// csrc/embed.hip -- the Embed layer that feeds SimCross.
//
// Reference: src/caffe/layers/embed_layer.cpp:135-152 (forward), :155-180
// (backward); embed_layer.cu:11-40 is the CUDA version (atomicAdd scatter,
// nondeterministic).  index is a blob of Dtype holding integers (Caffe feeds
// word ids as floats), weight is (K, N) = (vocabulary, embedding dim).
//   forward : top[n,:] = weight[int(index[n]),:] (+ bias)          -- a row gather
//   backward: weight_diff[int(index[n]),:] += top_diff[n,:], n ASCENDING on the
//             CPU.  fp32 addition is not associative, so the reference's result
//             depends on that order.  Here: a stable radix sort of (index, n)
//             builds the inverted index (rows of each word id in n order) and
//             each destination row is accumulated sequentially in exactly that
//             order, starting from the existing diff -- bit-identical to the
//             CPU code, no atomics.  The zero-pad word id owns thousands of rows
//             in a TREC-QA batch; its chain is long (one dependent add per row
//             per column) but is a single destination among ~10^3 short ones.
#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "mms_common.h"

namespace mms {

__global__ __launch_bounds__(256) void embed_fwd_kernel(int M, int N, int K,
                                                        const float* __restrict__ index,
                                                        const float* __restrict__ weight,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ top) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per output row
  if (n >= M) return;
  const int lane = threadIdx.x & 63;
  int idx = (int)index[n];
  idx = idx < 0 ? 0 : (idx >= K ? K - 1 : idx);        // the reference only DCHECKs; stay in bounds
  const float* w = weight + (size_t)idx * N;
  float* t = top + (size_t)n * N;
  for (int d = lane; d < N; d += 64) {
    float v = w[d];
    if (bias) v = 1.0f * (1.0f * bias[d]) + 1.0f * v;   // gemm(M,N,1): alpha*(1*bias) + beta*top
    t[d] = v;
  }
}

__global__ __launch_bounds__(256) void embed_keys_kernel(int M, int K, const float* __restrict__ index,
                                                         unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ vals) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= M) return;
  int idx = (int)index[n];
  idx = idx < 0 ? 0 : (idx >= K ? K - 1 : idx);
  keys[n] = (unsigned)idx;
  vals[n] = (unsigned)n;
}

// One workgroup per sorted position; only the first position of each word id works:
// it owns that destination row and adds the row's contributions in n order.
__global__ __launch_bounds__(256) void embed_bwd_kernel(int M, int N, const unsigned* __restrict__ keys,
                                                        const unsigned* __restrict__ vals,
                                                        const float* __restrict__ top_diff,
                                                        float* __restrict__ weight_diff) {
  const int p = blockIdx.x;
  const unsigned idx = keys[p];
  if (p > 0 && keys[p - 1] == idx) return;
  float* w = weight_diff + (size_t)idx * N;
  for (int d = threadIdx.x; d < N; d += 256) {
    float acc = w[d];
    for (int r = p; r < M && keys[r] == idx; ++r)
      acc = 1.0f * top_diff[(size_t)vals[r] * N + d] + acc;      // caffe_axpy(alpha = 1)
    w[d] = acc;
  }
}

// bias_diff += column sums of top_diff (gemv in the reference: BLAS order, 1e-5 bar).
constexpr int kBiasChunk = 512;
__global__ __launch_bounds__(256) void embed_bias_partial_kernel(int M, int N,
                                                                 const float* __restrict__ top_diff,
                                                                 float* __restrict__ partial) {
  const int c = blockIdx.x;
  const int n0 = c * kBiasChunk, n1 = min(M, n0 + kBiasChunk);
  for (int d = threadIdx.x; d < N; d += 256) {
    float s = 0.f;
    for (int n = n0; n < n1; ++n) s += top_diff[(size_t)n * N + d];
    partial[(size_t)c * N + d] = s;
  }
}
__global__ __launch_bounds__(256) void embed_bias_finish_kernel(int chunks, int N,
                                                                const float* __restrict__ partial,
                                                                float* __restrict__ bias_diff) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= N) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * N + d];
  bias_diff[d] = 1.0f * s + 1.0f * bias_diff[d];
}

struct EmbedWs { size_t k0, k1, v0, v1, partial, temp, total; int chunks; };
static EmbedWs embed_ws(int M, int N) {
  EmbedWs w{};
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += round_up(b, 256); return at; };
  w.k0 = take((size_t)M * 4); w.k1 = take((size_t)M * 4);
  w.v0 = take((size_t)M * 4); w.v1 = take((size_t)M * 4);
  w.chunks = (M + kBiasChunk - 1) / kBiasChunk;
  w.partial = take((size_t)w.chunks * N * 4);
  w.temp = o;
  w.total = o + (size_t)M * 8 + (4u << 20);
  return w;
}
size_t embed_workspace_bytes(int M, int N) { return embed_ws(M, N).total; }

int embed_forward(int M, int N, int K, const float* index, const float* weight, const float* bias,
                  float* top, hipStream_t s) {
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, M, N, K, index,
                     weight, bias, top);
  return launch_status();
}

int embed_backward(int M, int N, int K, const float* index, const float* top_diff,
                   float* weight_diff, float* bias_diff, void* ws, size_t ws_bytes, hipStream_t s) {
  const EmbedWs lay = embed_ws(M, N);
  if (!ws || ws_bytes < lay.temp) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  auto* k0 = reinterpret_cast<unsigned*>(base + lay.k0);
  auto* k1 = reinterpret_cast<unsigned*>(base + lay.k1);
  auto* v0 = reinterpret_cast<unsigned*>(base + lay.v0);
  auto* v1 = reinterpret_cast<unsigned*>(base + lay.v1);
  if (weight_diff) {
    hipLaunchKernelGGL(embed_keys_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, K, index,
                       k0, v0);
    size_t need = 0;
    if (rocprim::radix_sort_pairs(nullptr, need, k0, k1, v0, v1, (size_t)M, 0u, 32u, s) != hipSuccess)
      return MMS_ERR_LAUNCH;
    if (lay.temp + need > ws_bytes) return MMS_ERR_WORKSPACE;
    if (rocprim::radix_sort_pairs(base + lay.temp, need, k0, k1, v0, v1, (size_t)M, 0u, 32u, s) != hipSuccess)
      return MMS_ERR_LAUNCH;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)M), dim3(256), 0, s, M, N, k1, v1, top_diff,
                       weight_diff);
  }
  if (bias_diff) {
    float* partial = reinterpret_cast<float*>(base + lay.partial);
    hipLaunchKernelGGL(embed_bias_partial_kernel, dim3((unsigned)lay.chunks), dim3(256), 0, s, M, N,
                       top_diff, partial);
    hipLaunchKernelGGL(embed_bias_finish_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s,
                       lay.chunks, N, partial, bias_diff);
  }
  return launch_status();
}

// ---- batch feed: rows of a device-resident dataset -> one top blob -----------------
// Replaces the per-row caffe_copy loop of HDF5DataLayer::Forward_cpu/_gpu
// (src/caffe/layers/hdf5_data_layer.cpp:124-151, hdf5_data_layer.cu:19-51): the whole
// file's dataset lives in HBM, a batch is ONE launch gathering `rows` rows through the
// layer's row permutation (identity unless hdf5_data_param.shuffle).
__global__ __launch_bounds__(256) void feed_gather_kernel(int rows, int row_elems, int src_rows,
                                                          const float* __restrict__ src,
                                                          const int* __restrict__ perm, int first,
                                                          float* __restrict__ dst) {
  const size_t total = (size_t)rows * row_elems;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int i = (int)(e / row_elems), d = (int)(e - (size_t)i * row_elems);
    int r = perm ? perm[first + i] : first + i;
    r = r < 0 ? 0 : (r >= src_rows ? src_rows - 1 : r);
    dst[e] = src[(size_t)r * row_elems + d];
  }
}

int feed_gather_rows(int rows, int row_elems, int src_rows, const float* src, const int* perm, int first,
                     float* dst, hipStream_t s) {
  const size_t total = (size_t)rows * row_elems;
  const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(feed_gather_kernel, dim3(blocks), dim3(256), 0, s, rows, row_elems, src_rows, src, perm,
                     first, dst);
  return launch_status();
}

}  // namespace mms
