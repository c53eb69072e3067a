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

// The rows a backward pass adds, in order: layer 0's M0 rows, then (optionally) layer 1's -- the VIRTUAL concatenation
// of two Embed layers that share one table (network_v4's w2v_q / w2v_a, do_trec_qa_clean.py:452-467), whose Backward
// calls the reference runs one after the other into the same weight diff (embed_layer.cpp:155-180): row n of the
// concatenation is layer 0's row n, or layer 1's row n - M0.  A single layer is M0 = M, second = null.
struct EmbedSrc {
  const float* index0; const float* diff0; int M0;
  const float* index1; const float* diff1;
  __device__ __forceinline__ float index(int n) const { return n < M0 ? index0[n] : index1[n - M0]; }
  __device__ __forceinline__ const float* row(unsigned n, int N) const {
    return (int)n < M0 ? diff0 + (size_t)n * N : diff1 + (size_t)(n - M0) * N;
  }
};

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

__global__ __launch_bounds__(256) void embed_keys_kernel(int M, int K, EmbedSrc src,
                                                         unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ vals) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= M) return;
  int idx = (int)src.index(n);
  idx = idx < 0 ? 0 : (idx >= K ? K - 1 : idx);
  keys[n] = (unsigned)idx;
  vals[n] = (unsigned)n;
}

// heads[s] = first sorted position of the s-th distinct word id (rocprim::select over a
// counting iterator with these flags); segment s is [heads[s], heads[s+1]) in (keys, vals).
__global__ __launch_bounds__(256) void embed_head_flags_kernel(int M, const unsigned* __restrict__ keys,
                                                               unsigned char* __restrict__ flags) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < M) flags[p] = (p == 0 || keys[p - 1] != keys[p]) ? 1 : 0;
}

// One workgroup per (word id, 64-column slice): it owns that destination row and adds the
// id's rows in n order.  The adds are a dependent chain by definition (fp32, reference order),
// so everything else is taken off it: wave 0 only adds, from LDS, eight reads ahead; the other
// waves GATHER the next chunk of CH rows (row numbers staged through LDS one chunk further
// ahead, then CH/8 independent loads per thread, all issued before the first LDS write).  One
// memory round trip per chunk instead of two per row -- the zero-pad word id owns thousands
// of rows of a TREC-QA batch.  Short segments (<= CH rows: nearly every id) are one chunk.
template <int CH, int GW>   // GW gathering waves; block = 64 * (GW + 1) threads
__global__ __launch_bounds__(64 * (GW + 1)) void embed_bwd_seg_kernel(
    int M, int N, const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
    const unsigned* __restrict__ heads, const unsigned* __restrict__ nseg,
    EmbedSrc src, float* __restrict__ weight_diff, int rmin, int rmax) {
  extern __shared__ float seg_buf[];             // [2][CH][64] floats, then [2][CH] row numbers
  constexpr int RPT = CH / GW;                   // rows per gathering thread per chunk
  static_assert(CH % GW == 0, "chunk rows must divide evenly over the gathering waves");
  unsigned* vbuf = reinterpret_cast<unsigned*>(seg_buf + 2 * CH * 64);
  const int seg = blockIdx.x, c0 = blockIdx.y * 64, tid = threadIdx.x;
  const int count = (int)nseg[0];
  if (seg >= count) return;
  const int p = (int)heads[seg];
  const int R = (seg + 1 < count ? (int)heads[seg + 1] : M) - p;
  if (R < rmin || R > rmax) return;
  const unsigned idx = keys[p];
  const int gt = tid - 64;                       // >= 0: a gathering thread
  const int col = gt & 63, rg = gt >> 6, gcol = min(c0 + col, N - 1);
  auto stage_rows = [&](int b, int r0) {         // row numbers of chunk [r0, r0 + CH) -> vbuf[b]
    for (int e = gt; gt >= 0 && e < CH; e += 64 * GW) vbuf[b * CH + e] = vals[p + min(r0 + e, R - 1)];
  };
  auto gather = [&](int b, int r0) {             // rows rg, rg + GW, ... of the chunk, column `col`
    if (gt < 0) return;
    float x[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) x[u] = src.row(vbuf[b * CH + rg + u * GW], N)[gcol];
#pragma unroll
    for (int u = 0; u < RPT; ++u) seg_buf[(b * CH + rg + u * GW) * 64 + col] = x[u];
  };
  stage_rows(0, 0);
  __syncthreads();
  gather(0, 0);
  if (CH < R) stage_rows(1, CH);
  __syncthreads();
  const bool adder = tid < 64 && c0 + tid < N;
  float acc = adder ? weight_diff[(size_t)idx * N + c0 + tid] : 0.f;
  for (int r0 = 0, b = 0; r0 < R; r0 += CH, b ^= 1) {
    if (r0 + 2 * CH < R) stage_rows(b, r0 + 2 * CH);      // vbuf[b] was consumed before the last barrier
    if (r0 + CH < R) gather(b ^ 1, r0 + CH);              // row numbers staged one iteration ago
    if (adder) {
      const int rows = min(CH, R - r0);
      const float* colp = seg_buf + (size_t)b * CH * 64 + tid;
      int row = 0;
      for (; row + 8 <= rows; row += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = colp[(row + u) * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = 1.0f * v[u] + acc;     // caffe_axpy(alpha = 1), n ascending
      }
      for (; row < rows; ++row) acc = 1.0f * colp[row * 64] + acc;
    }
    __syncthreads();                                        // chunk r0 + CH is in seg_buf[b ^ 1], its successor's rows in vbuf[b]
  }
  if (adder) weight_diff[(size_t)idx * N + c0 + tid] = acc;
}

// bias_diff += column sums of top_diff (caffe_cpu_gemv in the reference, embed_layer.cpp:176-178: BLAS order, 1e-5
// bar).  Round 3: the first cut gave each of 50 threads a 512-row dependent load-add loop -- 104 us per Embed layer at
// the driver's batch, 70 % of a whole training step of network_v4 (profiles/r03_v4_step_kernel_stats.txt).  Now a
// workgroup owns 128 rows as LANES row lanes x 64 columns, eight independent loads in flight per thread, the lanes
// folded through LDS in a fixed order.  For small batches the two launches ride along with launches that leave the
// chip idle anyway: the partial sums in the workgroups NEXT to the one-workgroup inverted-index build, the final sum
// in workgroups appended to the short-segment launch (3 launches per backward pass instead of 5).
constexpr int kBiasChunk = 128;
template <int LANES>                                  // blockDim.x == 64 * LANES
__device__ __forceinline__ void embed_bias_partial_body(int c, int M, int N, const EmbedSrc& src,
                                                        float* __restrict__ partial, float (*red)[64]) {
  const int ry = threadIdx.x >> 6, dl = threadIdx.x & 63;
  const int n0 = c * kBiasChunk, n1 = min(M, n0 + kBiasChunk);
  for (int d0 = 0; d0 < N; d0 += 64) {
    const int d = min(d0 + dl, N - 1);
    float s = 0.f;
    for (int nb = n0 + ry; nb < n1; nb += 8 * LANES) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src.row((unsigned)min(nb + LANES * u, M - 1), N)[d];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (nb + LANES * u < n1) ? v[u] : 0.f;
    }
    red[ry][dl] = s;
    __syncthreads();
    if (ry == 0 && d0 + dl < N) {
      float t = red[0][dl];
#pragma unroll
      for (int l = 1; l < LANES; ++l) t += red[l][dl];
      partial[(size_t)c * N + d0 + dl] = t;
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void embed_bias_finish_body(int d, int chunks, int N, const float* __restrict__ partial,
                                                       float* __restrict__ bias_diff) {
  if (d >= N) return;
  float s = 0.f;
  for (int c0 = 0; c0 < chunks; c0 += 8) {       // eight loads in flight, same c-ascending order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)min(c0 + u, chunks - 1) * N + d];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (c0 + u < chunks) ? v[u] : 0.f;
  }
  bias_diff[d] = 1.0f * s + 1.0f * bias_diff[d];
}
__global__ __launch_bounds__(256) void embed_bias_partial_kernel(int M, int N, EmbedSrc src,
                                                                 float* __restrict__ partial) {
  __shared__ float red[4][64];
  embed_bias_partial_body<4>((int)blockIdx.x, M, N, src, partial, red);
}
__global__ __launch_bounds__(256) void embed_bias_finish_kernel(int chunks, int N,
                                                                const float* __restrict__ partial,
                                                                float* __restrict__ bias_diff) {
  embed_bias_finish_body((int)(blockIdx.x * 256 + threadIdx.x), chunks, N, partial, bias_diff);
}

// Small batches (M <= 4096: the driver's 50 x 40-word training batch is 2,000 indices per Embed
// layer): the whole inverted index -- clamp, stable radix sort of (id, n), head flags, compaction
// of the head positions and their count -- in ONE workgroup and one launch instead of the dozen
// launches of the device-wide sort + select (each a few microseconds of pure latency).
constexpr int kPrepThreads = 1024, kPrepItems = 4, kPrepMax = kPrepThreads * kPrepItems;
__device__ __forceinline__ void embed_prep_body(int M, int K, unsigned bits, const EmbedSrc& src,
                                                unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                                unsigned* __restrict__ heads, unsigned* __restrict__ nseg) {
  using sort_t = rocprim::block_radix_sort<unsigned, kPrepThreads, kPrepItems, unsigned>;
  using scan_t = rocprim::block_scan<unsigned, kPrepThreads>;
  __shared__ union { typename sort_t::storage_type sort; typename scan_t::storage_type scan; } tmp;
  __shared__ unsigned skeys[kPrepMax + 1];
  const int t = threadIdx.x;
  unsigned k[kPrepItems], v[kPrepItems];
#pragma unroll
  for (int i = 0; i < kPrepItems; ++i) {
    const int n = t * kPrepItems + i;
    int idx = n < M ? (int)src.index(n) : 0;
    idx = idx < 0 ? 0 : (idx >= K ? K - 1 : idx);
    k[i] = n < M ? (unsigned)idx : 0xffffffffu;  // padding sorts to the end (all `bits` + the pad bit)
    v[i] = (unsigned)n;
  }
  sort_t().sort(k, v, tmp.sort, 0, bits + 1 > 32 ? 32 : bits + 1);   // stable: equal ids keep n ascending
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPrepItems; ++i) {
    const int pos = t * kPrepItems + i;
    skeys[pos + 1] = k[i];
    if (pos < M) { keys[pos] = k[i]; vals[pos] = v[i]; }
  }
  if (t == 0) skeys[0] = 0xfffffffeu;            // differs from every real id and from the padding
  __syncthreads();
  unsigned flag[kPrepItems], mine = 0;
#pragma unroll
  for (int i = 0; i < kPrepItems; ++i) {
    const int pos = t * kPrepItems + i;
    flag[i] = (pos < M && skeys[pos] != skeys[pos + 1]) ? 1u : 0u;
    mine += flag[i];
  }
  unsigned before = 0, total = 0;
  scan_t().exclusive_scan(mine, before, 0u, total, tmp.scan);
#pragma unroll
  for (int i = 0; i < kPrepItems; ++i) {
    if (flag[i]) heads[before++] = (unsigned)(t * kPrepItems + i);
  }
  if (t == 0) nseg[0] = total;
}
__global__ __launch_bounds__(kPrepThreads) void embed_prep_small_kernel(
    int M, int K, unsigned bits, EmbedSrc src, unsigned* __restrict__ keys,
    unsigned* __restrict__ vals, unsigned* __restrict__ heads, unsigned* __restrict__ nseg,
    int N, float* __restrict__ bias_partial) {
  if (blockIdx.x > 0) {                          // riders: the bias gradient's partial sums (bias_partial != null)
    __shared__ float red[kPrepThreads / 64][64];
    embed_bias_partial_body<kPrepThreads / 64>((int)blockIdx.x - 1, M, N, src, bias_partial, red);
    return;
  }
  embed_prep_body(M, K, bits, src, keys, vals, heads, nseg);
}

// Forward of two Embed layers over ONE table in one launch, with the inverted index of their concatenated word ids
// built beside the gathers (workgroup 0): the index depends on the ids alone, the one-workgroup sort that builds it
// takes 14 us at a lone workgroup's clock, and in the forward it costs nothing -- the gathers and the layers behind
// them keep the rest of the chip busy.  The backward pass (embed_backward_pair with index_ready) then starts at the
// segment kernels.
__global__ __launch_bounds__(kPrepThreads) void embed_fwd_pair_kernel(
    int M0, int M1, int N, int K, EmbedSrc src, const float* __restrict__ weight, const float* __restrict__ bias,
    float* __restrict__ top0, float* __restrict__ top1, int build, unsigned bits, unsigned* __restrict__ keys,
    unsigned* __restrict__ vals, unsigned* __restrict__ heads, unsigned* __restrict__ nseg) {
  const int M = M0 + M1;
  if (build && blockIdx.x == 0) {
    embed_prep_body(M, K, bits, src, keys, vals, heads, nseg);
    return;
  }
  const int n = ((int)blockIdx.x - (build ? 1 : 0)) * (kPrepThreads / 64) + (int)(threadIdx.x >> 6);   // one wave per row
  if (n >= M) return;
  const int lane = threadIdx.x & 63;
  int idx = (int)src.index(n);
  idx = idx < 0 ? 0 : (idx >= K ? K - 1 : idx);
  const float* w = weight + (size_t)idx * N;
  float* t = n < M0 ? top0 + (size_t)n * N : top1 + (size_t)(n - M0) * N;
  for (int d = lane; d < N; d += 64) {
    float v = w[d];
    if (bias) v = 1.0f * (1.0f * bias[d]) + 1.0f * v;   // gemm(M,N,1): alpha*(1*bias) + beta*top, as embed_fwd_kernel
    t[d] = v;
  }
}

// Short segments (nearly every word id: a handful of rows): one WAVE per (id, 64-column slice),
// four per workgroup, no LDS and no barrier -- the row numbers and then the rows are requested
// all at once (8 or 32 unconditional loads with clamped addresses), then added in order.
__global__ __launch_bounds__(256) void embed_bwd_short_kernel(
    int M, int N, const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
    const unsigned* __restrict__ heads, const unsigned* __restrict__ nseg,
    EmbedSrc src, float* __restrict__ weight_diff, int seg_blocks, int bias_chunks,
    const float* __restrict__ bias_partial, float* __restrict__ bias_diff) {
  if ((int)blockIdx.x >= seg_blocks) {           // riders: the bias gradient's final sum (the partials are a launch old)
    if (blockIdx.y == 0)
      embed_bias_finish_body(((int)blockIdx.x - seg_blocks) * 256 + (int)threadIdx.x, bias_chunks, N, bias_partial, bias_diff);
    return;
  }
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int c = blockIdx.y * 64 + lane;
  const int count = (int)nseg[0];
  if (seg >= count) return;
  const int p = (int)heads[seg];
  const int R = (seg + 1 < count ? (int)heads[seg + 1] : M) - p;
  if (R > 32) return;                            // embed_bwd_seg_kernel's
  const unsigned idx = keys[p];
  const int gc = min(c, N - 1);
  float acc = weight_diff[(size_t)idx * N + gc];
  if (R <= 8) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = src.row(vals[p + min(u, R - 1)], N)[gc];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < R) acc = 1.0f * x[u] + acc;
  } else {
    float x[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) x[u] = src.row(vals[p + min(u, R - 1)], N)[gc];
#pragma unroll
    for (int u = 0; u < 32; ++u) if (u < R) acc = 1.0f * x[u] + acc;
  }
  if (c < N) weight_diff[(size_t)idx * N + c] = acc;
}

struct EmbedWs { size_t k0, k1, v0, v1, flags, heads, nseg, partial, temp, total; int chunks; };
static EmbedWs embed_ws(int M, int N) {
  EmbedWs w{};
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += round_up(b, 256); return at; };
  w.k0 = take((size_t)M * 4); w.k1 = take((size_t)M * 4);
  w.v0 = take((size_t)M * 4); w.v1 = take((size_t)M * 4);
  w.flags = take((size_t)M); w.heads = take((size_t)M * 4); w.nseg = take(256);
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

static int embed_backward_src(int M, int N, int K, const EmbedSrc& src,
                              float* weight_diff, float* bias_diff, void* ws, size_t ws_bytes, hipStream_t s,
                              bool index_ready = false) {
  const EmbedWs lay = embed_ws(M, N);
  if (!ws || ws_bytes < lay.temp) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  auto* k0 = reinterpret_cast<unsigned*>(base + lay.k0);
  auto* k1 = reinterpret_cast<unsigned*>(base + lay.k1);
  auto* v0 = reinterpret_cast<unsigned*>(base + lay.v0);
  auto* v1 = reinterpret_cast<unsigned*>(base + lay.v1);
  bool bias_rides = false;                          // the bias gradient rode along with the weight gradient's launches
  if (weight_diff) {
    unsigned bits = 1;                              // keys are clamped to [0, K): sort those bits only
    while (bits < 32 && (1ull << bits) < (unsigned long long)K) ++bits;
    auto* heads = reinterpret_cast<unsigned*>(base + lay.heads);
    auto* nseg = reinterpret_cast<unsigned*>(base + lay.nseg);
    if (index_ready && M <= kPrepMax && bits < 32) {
      // keys / vals / heads / nseg of this workspace were written by embed_forward_pair for these very ids
      if (bias_diff) {
        hipLaunchKernelGGL(embed_bias_partial_kernel, dim3((unsigned)lay.chunks), dim3(256), 0, s, M, N, src,
                           reinterpret_cast<float*>(base + lay.partial));
        bias_rides = true;                          // (the final sum still rides with the short-segment launch)
      }
    } else if (M <= kPrepMax && bits < 32) {
      bias_rides = bias_diff != nullptr;
      hipLaunchKernelGGL(embed_prep_small_kernel, dim3(1u + (bias_rides ? (unsigned)lay.chunks : 0u)), dim3(kPrepThreads), 0,
                         s, M, K, bits, src, k1, v1, heads, nseg, N, reinterpret_cast<float*>(base + lay.partial));
    } else {
      hipLaunchKernelGGL(embed_keys_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, K, src,
                         k0, v0);
      size_t need = 0;
      if (rocprim::radix_sort_pairs(nullptr, need, k0, k1, v0, v1, (size_t)M, 0u, bits, s) != hipSuccess)
        return MMS_ERR_LAUNCH;
      if (lay.temp + need > ws_bytes) return MMS_ERR_WORKSPACE;
      if (rocprim::radix_sort_pairs(base + lay.temp, need, k0, k1, v0, v1, (size_t)M, 0u, bits, s) != hipSuccess)
        return MMS_ERR_LAUNCH;
      // distinct ids: head flags -> compacted head positions (+ their count, on the device)
      auto* flags = reinterpret_cast<unsigned char*>(base + lay.flags);
      hipLaunchKernelGGL(embed_head_flags_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, k1, flags);
      size_t need2 = 0;
      rocprim::counting_iterator<unsigned> pos(0);
      if (rocprim::select(nullptr, need2, pos, flags, heads, nseg, (size_t)M, s) != hipSuccess) return MMS_ERR_LAUNCH;
      if (lay.temp + need2 > ws_bytes) return MMS_ERR_WORKSPACE;
      if (rocprim::select(base + lay.temp, need2, pos, flags, heads, nseg, (size_t)M, s) != hipSuccess)
        return MMS_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)std::min(M, K), (unsigned)((N + 63) / 64));   // at most min(M, K) distinct ids
    constexpr size_t kLongLds = 2 * 256 * (64 * sizeof(float) + sizeof(unsigned));   // 130 KB of the CU's 160 KB
    static const hipError_t once = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&embed_bwd_seg_kernel<256, 8>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLongLds);
    (void)once;
    const unsigned seg_blocks = (grid.x + 3) / 4, fin_blocks = bias_rides ? (unsigned)((N + 255) / 256) : 0u;
    hipLaunchKernelGGL(embed_bwd_short_kernel, dim3(seg_blocks + fin_blocks, grid.y), dim3(256), 0, s, M, N, k1, v1,
                       heads, nseg, src, weight_diff, (int)seg_blocks, lay.chunks,
                       reinterpret_cast<const float*>(base + lay.partial), bias_diff);
    hipLaunchKernelGGL((embed_bwd_seg_kernel<256, 8>), grid, dim3(576), kLongLds, s, M, N, k1, v1, heads, nseg,
                       src, weight_diff, 33, 0x7fffffff);
  }
  if (bias_diff && !bias_rides) {
    float* partial = reinterpret_cast<float*>(base + lay.partial);
    hipLaunchKernelGGL(embed_bias_partial_kernel, dim3((unsigned)lay.chunks), dim3(256), 0, s, M, N,
                       src, partial);
    hipLaunchKernelGGL(embed_bias_finish_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s,
                       lay.chunks, N, partial, bias_diff);
  }
  return launch_status();
}

int embed_backward(int M, int N, int K, const float* index, const float* top_diff,
                   float* weight_diff, float* bias_diff, void* ws, size_t ws_bytes, hipStream_t s) {
  const EmbedSrc src{index, top_diff, M, nullptr, nullptr};
  return embed_backward_src(M, N, K, src, weight_diff, bias_diff, ws, ws_bytes, s);
}

// Two Embed layers over ONE table: layer 0's Backward, then layer 1's, as one pass over the concatenation of their
// rows (the inverted index is built once, every table row is read and written once, the bias gradient is one sum).
// Same bits as the two calls: a table row's additions keep the order "layer 0's rows ascending, then layer 1's".
int embed_backward_pair(int M0, int M1, int N, int K, const float* index0, const float* top_diff0, const float* index1,
                        const float* top_diff1, float* weight_diff, float* bias_diff, void* ws, size_t ws_bytes,
                        hipStream_t s, int index_ready) {
  const EmbedSrc src{index0, top_diff0, M0, index1, top_diff1};
  return embed_backward_src(M0 + M1, N, K, src, weight_diff, bias_diff, ws, ws_bytes, s, index_ready != 0);
}

// can embed_forward_pair build the index for these sizes?  (the one-workgroup sort: up to 4,096 ids)
bool embed_pair_index_supported(int M0, int M1, int K) {
  unsigned bits = 1;
  while (bits < 32 && (1ull << bits) < (unsigned long long)K) ++bits;
  return (long long)M0 + M1 <= kPrepMax && bits < 32;
}

// top0 = Embed(index0), top1 = Embed(index1), one launch; index_ws (optional, embed_workspace_bytes(M0 + M1, N)): the
// inverted index of the ids in the order (index0, index1) -- hand the SAME workspace and the same order to
// embed_backward_pair with index_ready = 1
int embed_forward_pair(int M0, int M1, int N, int K, const float* index0, const float* index1, const float* weight,
                       const float* bias, float* top0, float* top1, void* index_ws, size_t index_ws_bytes, hipStream_t s) {
  const int M = M0 + M1;
  const EmbedWs lay = embed_ws(M, N);
  const bool build = index_ws != nullptr && embed_pair_index_supported(M0, M1, K);
  if (build && index_ws_bytes < lay.temp) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(index_ws);
  unsigned bits = 1;
  while (bits < 32 && (1ull << bits) < (unsigned long long)K) ++bits;
  const EmbedSrc src{index0, nullptr, M0, index1, nullptr};
  const unsigned rows_per_wg = kPrepThreads / 64;
  hipLaunchKernelGGL(embed_fwd_pair_kernel, dim3((unsigned)((M + rows_per_wg - 1) / rows_per_wg) + (build ? 1u : 0u)),
                     dim3(kPrepThreads), 0, s, M0, M1, N, K, src, weight, bias, top0, top1, build ? 1 : 0, bits,
                     build ? reinterpret_cast<unsigned*>(base + lay.k1) : nullptr,
                     build ? reinterpret_cast<unsigned*>(base + lay.v1) : nullptr,
                     build ? reinterpret_cast<unsigned*>(base + lay.heads) : nullptr,
                     build ? reinterpret_cast<unsigned*>(base + lay.nseg) : nullptr);
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
