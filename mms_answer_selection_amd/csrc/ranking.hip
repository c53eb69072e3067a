// csrc/ranking.hip -- the forward-only ranking metrics that consume the scores
// of the hot path: MAP, MRR, AUC, RankAccuracy.
//
// Reference:
//   MAPLayer::Forward_cpu            src/caffe/layers/map_layer.cpp:41-100
//   MRRLayer::Forward_cpu            src/caffe/layers/mrr_layer.cpp:38-79
//   AUCLayer::Forward_cpu            src/caffe/layers/auc_layer.cpp:47-136
//   RankAccuracyLayer::Forward_cpu   src/caffe/layers/rank_accuracy_layer.cpp:36-50
// The reference buckets items by int(group) in a std::map (ascending group
// id), std::sort's each bucket by score descending, and walks it
// sequentially.  Here: one 64-bit radix sort (rocPRIM) on
//   key = (group id, biased to unsigned) << 32 | order-reversed score bits
// gives "group ascending, score descending" for all buckets at once; one thread
// per bucket then walks its items and one thread folds the buckets in group
// order, both with the reference's float expressions (`ap += ++rank/(i+1)`,
// `mrr += 1.0/(rank+1)` through double, ...), so MAP/MRR/AUC are bit-identical
// to the CPU code whenever the order is defined.  Ties: std::sort is unstable,
// so the reference's order among EQUAL scores is implementation-defined; the
// radix sort is stable (original index ascending).  Results can differ from a
// particular libstdc++ only when equal scores carry different labels IN A BUCKET OF MORE THAN 16 ITEMS: up to
// 16, std::sort is libstdc++'s insertion sort, which is stable -- the same order as here
// (tests/test_gpu_ranking.py, tools/soak_ranking_embed.py).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "libstdcxx_sort.h"
#include "mms_common.h"

namespace mms {

// How EQUAL scores with different labels are ordered (include/mms.h: mms_set_rank_tie_mode); per calling thread.
static thread_local int t_rank_ties = MMS_RANK_TIES_INPUT_ORDER;
int rank_tie_mode() { return t_rank_ties; }
void set_rank_tie_mode(int m) { t_rank_ties = m; }

__device__ __forceinline__ unsigned desc_bits(float s) {
  unsigned u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending total order
  return ~u;                                         // descending
}

// item i = (outer o, inner j) = (i / inner, i % inner); its score is prob[o*stride + offset*inner + j]
// (auc_layer.cpp:75-76; inner = 1 for MAP / MRR and for prob (N, C))
__device__ __forceinline__ unsigned long long rank_key(int i, int stride, int offset, int inner,
                                                        const float* __restrict__ prob,
                                                        const float* __restrict__ group) {
  const unsigned g = group ? (unsigned)((int)group[i]) + 0x80000000u : 0u;   // map<int,...> key order
  const int o = i / inner, j = i - o * inner;
  return ((unsigned long long)g << 32) | desc_bits(prob[(size_t)o * stride + (size_t)offset * inner + j]);
}
__global__ __launch_bounds__(256) void rank_keys_kernel(int n, int stride, int offset, int inner,
                                                        const float* __restrict__ prob,
                                                        const float* __restrict__ group,
                                                        unsigned long long* __restrict__ keys,
                                                        unsigned* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = rank_key(i, stride, offset, inner, prob, group);
  vals[i] = (unsigned)i;
}

// One thread per sorted position; the thread at a bucket's first position walks the
// bucket (map_layer.cpp:76-96, mrr_layer.cpp:57-75).  flags bit 0: the bucket counts
// for MAP (a label == 1 and a label != 1 present, :80-92); bit 1: it counts for MRR
// (a label == 1 and a label == 0 present, mrr_layer.cpp:61-73).
// BYPOS: `label` is already in sorted order (label[p]); otherwise it is gathered through the permutation
template <bool BYPOS = false>
__device__ __forceinline__ void rank_bucket_at(int i, int n, const unsigned long long* keys,
                                               const unsigned* vals, const float* label,
                                               float* ap_out, int* rank_out, int* flags) {
  const unsigned g = (unsigned)(keys[i] >> 32);
  int fl = 0;
  if (i == 0 || (unsigned)(keys[i - 1] >> 32) != g) {
    float ap = 0.f;
    int map_rank = 0, not_one = 0, zero = 0, mrr_rank = -1;
    for (int p = i; p < n && (unsigned)(keys[p] >> 32) == g; ++p) {
      const int lab = (int)(BYPOS ? label[p] : label[vals[p]]);
      const int pos = p - i;
      if (lab == 1) {
        ap += (++map_rank) / (float)(pos + 1);
        if (mrr_rank < 0) mrr_rank = pos;
      } else {
        not_one = 1;
        if (lab == 0) zero = 1;
      }
    }
    if (map_rank >= 1 && not_one) { fl |= 1; ap_out[i] = ap / map_rank; }
    if (mrr_rank >= 0 && zero) { fl |= 2; rank_out[i] = mrr_rank; }
  }
  flags[i] = fl;
}
__global__ __launch_bounds__(256) void rank_bucket_pos_kernel(int n,
                                                              const unsigned long long* __restrict__ keys,
                                                              const float* __restrict__ lab_pos,
                                                              float* __restrict__ ap_out,
                                                              int* __restrict__ rank_out,
                                                              int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) rank_bucket_at<true>(i, n, keys, nullptr, lab_pos, ap_out, rank_out, flags);
}
__global__ __launch_bounds__(256) void rank_bucket_kernel(int n,
                                                          const unsigned long long* __restrict__ keys,
                                                          const unsigned* __restrict__ vals,
                                                          const float* __restrict__ label,
                                                          float* __restrict__ ap_out,
                                                          int* __restrict__ rank_out,
                                                          int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) rank_bucket_at(i, n, keys, vals, label, ap_out, rank_out, flags);
}

// Folds the buckets in sorted (= ascending group id) order with the reference's running sums.
// The sums are sequential by definition; ONE WAVE runs them so that the operands arrive 64
// positions per coalesced load (four loads in flight) instead of one dependent load per
// position: ballots pick the positions that carry a bucket result, and the wave-uniform
// running sums are advanced in position order with v_readlane.
__device__ __forceinline__ void rank_fold_wave(int lane, int n, const float* ap, const int* rank,
                                               const int* flags, float* __restrict__ map_out,
                                               float* __restrict__ mrr_out, int* __restrict__ effective) {
  float map_ = 0.f, mrr = 0.f;
  int eff_map = 0, eff_mrr = 0;
  int nfl[4], nrk[4];
  float na[4];
  auto fetch = [&](int base) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 64 * u + lane, ic = i < n ? i : n - 1;
      nfl[u] = flags[ic];
      na[u] = ap[ic];
      nrk[u] = rank[ic];
    }
  };
  fetch(0);
  for (int base = 0; base < n; base += 256) {
    int fl[4], rk[4];
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      fl[u] = (base + 64 * u + lane < n) ? nfl[u] : 0;
      a[u] = na[u];
      rk[u] = nrk[u];
    }
    if (base + 256 < n) fetch(base + 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned long long m1 = __ballot(fl[u] & 1), m2 = __ballot(fl[u] & 2);
      eff_map += __popcll(m1);
      eff_mrr += __popcll(m2);
      while (m1) {                                                   // map_layer.cpp:93-94
        const int l = __ffsll((long long)m1) - 1;
        m1 &= m1 - 1;
        map_ += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[u]), l));
      }
      while (m2) {   // mrr += 1.0/(mrr_rank+1): float + double, stored back to float (mrr_layer.cpp:75)
        const int l = __ffsll((long long)m2) - 1;
        m2 &= m2 - 1;
        mrr = (float)((double)mrr + 1.0 / (__builtin_amdgcn_readlane(rk[u], l) + 1));
      }
    }
  }
  if (lane == 0) {
    if (map_out) *map_out = map_ / eff_map;   // NaN when no bucket counts, like the reference (:99)
    if (mrr_out) *mrr_out = mrr / eff_mrr;
    if (effective) *effective = eff_map;
  }
}
__global__ __launch_bounds__(64) void rank_fold_kernel(int n, const float* __restrict__ ap,
                                                       const int* __restrict__ rank,
                                                       const int* __restrict__ flags,
                                                       float* __restrict__ map_out,
                                                       float* __restrict__ mrr_out,
                                                       int* __restrict__ effective) {
  rank_fold_wave(threadIdx.x, n, ap, rank, flags, map_out, mrr_out, effective);
}

// AUC: global descending sort, then the reference's sequential walk (auc_layer.cpp:119-134):
//   high += lab; auc += high * (1 - lab)     (ints; the product is converted to float and added)
// `high` is an integer prefix sum (exact in any order); the float running sum is the only
// sequential part.  One wave: 64 sorted items per step, labels gathered through the sort
// permutation four steps ahead, `high` by a wave scan, and the 64 terms added in item order
// with compile-time-lane v_readlane (two instructions per item, one dependent add).
__device__ __forceinline__ int wave_inclusive_scan_i32(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ void auc_fold_wave(int lane, int n, const unsigned* vals,
                                              const float* __restrict__ label, int has_ignore,
                                              int ignore_label, float* __restrict__ auc_out) {
  float auc = 0.f;
  int high = 0, count = 0;
  float nxt[4];
  auto fetch = [&](int base) {                    // clamped addresses: unconditional loads
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 64 * u + lane;
      nxt[u] = vals ? label[vals[i < n ? i : n - 1]] : label[i < n ? i : n - 1];   // vals == nullptr: labels by position
    }
  };
  fetch(0);
  for (int base = 0; base < n; base += 256) {
    int lab[4];
    bool use[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 64 * u + lane;
      lab[u] = (int)nxt[u];
      use[u] = i < n && !(has_ignore && lab[u] == ignore_label);     // :68-70 (skipped items keep their order)
      if (!use[u]) lab[u] = 0;
    }
    if (base + 256 < n) fetch(base + 256);        // in flight behind this block's 256 dependent adds
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int incl = high + wave_inclusive_scan_i32(lab[u], lane);
      const float term = use[u] ? (float)(incl * (1 - lab[u])) : 0.f;   // skipped items add +0: exact (auc >= 0 ... or any)
      count += __popcll(__ballot(use[u]));
      high = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
      for (int l = 0; l < 64; ++l)
        auc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), l));
    }
  }
  if (lane == 0) *auc_out = high > 0 ? auc / high / (count - high) : 0.f;
}
__global__ __launch_bounds__(64) void auc_fold_kernel(int n, const unsigned* __restrict__ vals,
                                                      const float* __restrict__ label, int has_ignore,
                                                      int ignore_label, float* __restrict__ auc_out) {
  auc_fold_wave(threadIdx.x, n, vals, label, has_ignore, ignore_label, auc_out);
}

// ---- MMS_RANK_TIES_LIBSTDCXX: the order a libstdc++ build of the reference leaves EQUAL scores in -----------------
// std::sort is not stable; where equal scores carry different labels the metrics depend on what that algorithm
// does.  Up to 16 items it is a stable insertion sort (= the stable order above).  For larger buckets with such a
// tie the bucket is re-sorted by ONE lane running libstdcxx_sort.h on the bucket's items in their original
// (push_back) order -- sequential by nature, hence opt-in: a few microseconds for a TREC-QA candidate group, but
// milliseconds for AUC's single bucket of thousands of items.
constexpr int kTieLds = 4096;
__global__ __launch_bounds__(256) void rank_ties_detect_kernel(int n, const unsigned long long* __restrict__ keys,
                                                               const unsigned* __restrict__ vals,
                                                               const float* __restrict__ label,
                                                               float* __restrict__ lab_pos, int* __restrict__ work,
                                                               int* __restrict__ nwork) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  lab_pos[i] = label[vals[i]];
  const unsigned g = (unsigned)(keys[i] >> 32);
  if (i != 0 && (unsigned)(keys[i - 1] >> 32) == g) return;
  int m = 0;
  bool cross = false;
  for (int p = i; p < n && (unsigned)(keys[p] >> 32) == g; ++p) {
    ++m;
    if (p > i && (unsigned)keys[p] == (unsigned)keys[p - 1] && (int)label[vals[p]] != (int)label[vals[p - 1]]) cross = true;
  }
  if (cross && m > 16) {
    const int w = atomicAdd(nwork, 1);
    work[2 * w] = i;
    work[2 * w + 1] = m;
  }
}
__global__ __launch_bounds__(64) void rank_ties_emulate_kernel(int stride, int offset, int inner,
                                                               const float* __restrict__ prob,
                                                               const float* __restrict__ label,
                                                               const unsigned* __restrict__ vals,
                                                               const int* __restrict__ work,
                                                               const int* __restrict__ nwork, int skip_ignored,
                                                               int ignore_label, float* __restrict__ lab_pos,
                                                               SortItem* __restrict__ g_items,
                                                               unsigned* __restrict__ g_idx) {
  __shared__ SortItem s_items[kTieLds];
  __shared__ unsigned s_idx[kTieLds];
  if (threadIdx.x != 0) return;                      // the algorithm is sequential: one lane per bucket
  const int cnt = *nwork;
  for (int w = blockIdx.x; w < cnt; w += gridDim.x) {
    const int start = work[2 * w], m = work[2 * w + 1];
    SortItem* it = m <= kTieLds ? s_items : g_items + start;
    unsigned* ix = m <= kTieLds ? s_idx : g_idx + start;
    // the bucket in push_back order = ascending item index (a heap sort of the indices: they are distinct)
    for (int j = 0; j < m; ++j) ix[j] = vals[start + j];
    for (int root0 = m / 2 - 1; root0 >= 0; --root0) {
      int root = root0;
      const unsigned v = ix[root];
      for (int c = 2 * root + 1; c < m; c = 2 * root + 1) {
        if (c + 1 < m && ix[c + 1] > ix[c]) ++c;
        if (ix[c] <= v) break;
        ix[root] = ix[c]; root = c;
      }
      ix[root] = v;
    }
    for (int end = m - 1; end > 0; --end) {
      const unsigned v = ix[end];
      ix[end] = ix[0];
      int root = 0;
      for (int c = 1; c < end; c = 2 * root + 1) {
        if (c + 1 < end && ix[c + 1] > ix[c]) ++c;
        if (ix[c] <= v) break;
        ix[root] = ix[c]; root = c;
      }
      ix[root] = v;
    }
    int used = 0;
    for (int j = 0; j < m; ++j) {
      const int idx = (int)ix[j], lab = (int)label[idx];
      if (skip_ignored && lab == ignore_label) continue;                        // auc_layer.cpp:68-70: never pushed
      const int o = idx / inner, jj = idx - o * inner;
      it[used].key = prob[(size_t)o * stride + (size_t)offset * inner + jj];
      it[used].lab = lab;
      ++used;
    }
    libstdcxx_sort(it, used);
    for (int j = 0; j < used; ++j) lab_pos[start + j] = (float)it[j].lab;
    for (int j = used; j < m; ++j) lab_pos[start + j] = (float)ignore_label;    // skipped items: behind the rest, still skipped
  }
}

// ---- the whole metric in ONE workgroup for small inputs (a mini-batch's worth of candidates) ----------------------
// keys -> LDS, a bitonic sort of (key, original index) pairs in LDS (the index as the minor key makes it the
// STABLE order the radix sort gives), then the same bucket walks and the same one-wave fold as above, reading
// LDS: one launch instead of keys + radix passes + walks + fold.  Same expressions in the same order: the same
// bits.  Measured with the capacity at 2048: 1,517 items 47 us against 45-51 us for the multi-launch path -- one
// CU's LDS and barriers are no faster than rocPRIM's several launches at that size -- so it serves n <= 512 only.
constexpr int kRankSmall = 512;
template <int MODE>                                  // 0: MAP / MRR, 1: AUC
__global__ __launch_bounds__(1024) void rank_small_kernel(int n, int stride, int offset, int inner,
                                                          const float* __restrict__ prob,
                                                          const float* __restrict__ label,
                                                          const float* __restrict__ group, int has_ignore,
                                                          int ignore_label, float* __restrict__ out0,
                                                          float* __restrict__ out1, int* __restrict__ effective) {
  __shared__ unsigned long long keys[kRankSmall];
  __shared__ unsigned vals[kRankSmall];
  __shared__ float ap[MODE == 0 ? kRankSmall : 1];
  __shared__ int rk[MODE == 0 ? kRankSmall : 1];
  __shared__ int fl[MODE == 0 ? kRankSmall : 1];
  __shared__ float slab[MODE == 0 ? kRankSmall : 1];   // labels in sorted order: the bucket walks read LDS only
  const int t = threadIdx.x;
  int P = 64;
  while (P < n) P <<= 1;                             // sorted size: a power of two, padded with maximal keys
  for (int i = t; i < P; i += 1024) {
    keys[i] = i < n ? rank_key(i, stride, offset, inner, prob, group) : ~0ull;
    vals[i] = (unsigned)i;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = t; p < (P >> 1); p += 1024) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), q = i | j;   // the pair (i, i ^ j) with i < q; j = 2^m
        const bool up = (i & k) == 0;
        const unsigned long long ki = keys[i], kq = keys[q];
        const unsigned vi = vals[i], vq = vals[q];
        const bool gt = ki > kq || (ki == kq && vi > vq);
        if (gt == up) { keys[i] = kq; keys[q] = ki; vals[i] = vq; vals[q] = vi; }
      }
      __syncthreads();
    }
  }
  if (MODE == 0) {
    // a walk that gathers label[vals[p]] itself pays one memory round trip per item (its loop exit depends on
    // the keys, so the loads cannot be issued ahead)
    for (int i = t; i < n; i += 1024) slab[i] = label[vals[i]];
    __syncthreads();
    for (int i = t; i < n; i += 1024) rank_bucket_at<true>(i, n, keys, vals, slab, ap, rk, fl);
    __syncthreads();
    if (t < 64) rank_fold_wave(t, n, ap, rk, fl, out0, out1, effective);
  } else {
    if (t < 64) auc_fold_wave(t, n, vals, label, has_ignore, ignore_label, out0);
  }
}

__global__ __launch_bounds__(256) void rank_accuracy_kernel(int count, const float* __restrict__ a,
                                                            const float* __restrict__ b,
                                                            const float* __restrict__ label,
                                                            unsigned* __restrict__ partial) {
  __shared__ unsigned red[4];
  unsigned c = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256)
    c += (label[i] * (a[i] - b[i])) > 0 ? 1u : 0u;      // :45
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void rank_accuracy_finish_kernel(int blocks, int count, const unsigned* __restrict__ partial,
                                            float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  unsigned long long c = 0;
  for (int i = 0; i < blocks; ++i) c += partial[i];
  // the reference accumulates 0/1 into a float: exact up to 2^24, then it sticks
  float acc = c > 16777216ull ? 16777216.f : (float)c;
  *out = acc / count;
}

// ------------------------------- workspace layout ---------------------------
struct RankWs {
  size_t keys0, keys1, vals0, vals1, ap, rr, flags, labpos, work, nwork, titems, tidx, temp, total;
};
static RankWs rank_ws(int n) {
  RankWs w{};
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += round_up(b, 256); return at; };
  w.keys0 = take((size_t)n * 8); w.keys1 = take((size_t)n * 8);
  w.vals0 = take((size_t)n * 4); w.vals1 = take((size_t)n * 4);
  w.ap = take((size_t)n * 4); w.rr = take((size_t)n * 4); w.flags = take((size_t)n * 4);
  w.labpos = take((size_t)n * 4); w.work = take((size_t)n * 8); w.nwork = take(256);      // MMS_RANK_TIES_LIBSTDCXX
  w.titems = take((size_t)n * sizeof(SortItem)); w.tidx = take((size_t)n * 4);
  w.temp = o;
  w.total = o + (size_t)n * 8 + (4u << 20);   // generous bound for rocPRIM's scratch; checked at run time
  return w;
}
size_t rank_workspace_bytes(int n) { return rank_ws(n).total; }

static int sort_pairs(const RankWs& lay, char* base, size_t ws_bytes, int n, unsigned bits,
                      hipStream_t s) {
  auto* k0 = reinterpret_cast<unsigned long long*>(base + lay.keys0);
  auto* k1 = reinterpret_cast<unsigned long long*>(base + lay.keys1);
  auto* v0 = reinterpret_cast<unsigned*>(base + lay.vals0);
  auto* v1 = reinterpret_cast<unsigned*>(base + lay.vals1);
  size_t need = 0;
  if (rocprim::radix_sort_pairs(nullptr, need, k0, k1, v0, v1, (size_t)n, 0u, bits, s) != hipSuccess)
    return MMS_ERR_LAUNCH;
  if (lay.temp + need > ws_bytes) return MMS_ERR_WORKSPACE;
  if (rocprim::radix_sort_pairs(base + lay.temp, need, k0, k1, v0, v1, (size_t)n, 0u, bits, s) !=
      hipSuccess)
    return MMS_ERR_LAUNCH;
  return MMS_OK;
}

int rank_map_mrr(int n, int fixed_axis, const float* prob, const float* label, const float* group,
                 float* map_out, float* mrr_out, int* effective, void* ws, size_t ws_bytes,
                 hipStream_t s) {
  const bool libstd = rank_tie_mode() == MMS_RANK_TIES_LIBSTDCXX;
  if (n > 0 && n <= kRankSmall && !libstd) {         // evaluation-sized: one workgroup, one launch
    hipLaunchKernelGGL(rank_small_kernel<0>, dim3(1), dim3(1024), 0, s, n, fixed_axis + 1, fixed_axis, 1, prob,
                       label, group, 0, 0, map_out, mrr_out, effective);
    return launch_status();
  }
  const RankWs lay = rank_ws(n);
  if (!ws || ws_bytes < lay.temp) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  const unsigned grid = (unsigned)((n + 255) / 256);
  // score of item i: prob[i*(fixed_axis+1) + fixed_axis]  (map_layer.cpp:50, mrr_layer.cpp:49)
  hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(256), 0, s, n, fixed_axis + 1, fixed_axis, 1,
                     prob, group, reinterpret_cast<unsigned long long*>(base + lay.keys0),
                     reinterpret_cast<unsigned*>(base + lay.vals0));
  int rc = sort_pairs(lay, base, ws_bytes, n, 64u, s);
  if (rc != MMS_OK) return rc;
  auto* keys = reinterpret_cast<unsigned long long*>(base + lay.keys1);
  auto* vals = reinterpret_cast<unsigned*>(base + lay.vals1);
  if (libstd) {
    auto* labpos = reinterpret_cast<float*>(base + lay.labpos);
    auto* work = reinterpret_cast<int*>(base + lay.work);
    auto* nwork = reinterpret_cast<int*>(base + lay.nwork);
    if (hipMemsetAsync(nwork, 0, sizeof(int), s) != hipSuccess) return MMS_ERR_LAUNCH;
    hipLaunchKernelGGL(rank_ties_detect_kernel, dim3(grid), dim3(256), 0, s, n, keys, vals, label, labpos, work, nwork);
    hipLaunchKernelGGL(rank_ties_emulate_kernel, dim3(256), dim3(64), 0, s, fixed_axis + 1, fixed_axis, 1, prob, label,
                       vals, work, nwork, 0, 0, labpos, reinterpret_cast<SortItem*>(base + lay.titems),
                       reinterpret_cast<unsigned*>(base + lay.tidx));
    hipLaunchKernelGGL(rank_bucket_pos_kernel, dim3(grid), dim3(256), 0, s, n, keys, labpos,
                       reinterpret_cast<float*>(base + lay.ap), reinterpret_cast<int*>(base + lay.rr),
                       reinterpret_cast<int*>(base + lay.flags));
  } else
  hipLaunchKernelGGL(rank_bucket_kernel, dim3(grid), dim3(256), 0, s, n, keys, vals, label,
                     reinterpret_cast<float*>(base + lay.ap), reinterpret_cast<int*>(base + lay.rr),
                     reinterpret_cast<int*>(base + lay.flags));
  hipLaunchKernelGGL(rank_fold_kernel, dim3(1), dim3(64), 0, s, n,
                     reinterpret_cast<float*>(base + lay.ap), reinterpret_cast<int*>(base + lay.rr),
                     reinterpret_cast<int*>(base + lay.flags), map_out, mrr_out, effective);
  return launch_status();
}

// n = outer * inner items; dim = channels * inner floats per outer index
int rank_auc(int n, int dim, int fixed_axis, int inner, const float* prob, const float* label, int has_ignore,
             int ignore_label, float* auc_out, void* ws, size_t ws_bytes, hipStream_t s) {
  const bool libstd = rank_tie_mode() == MMS_RANK_TIES_LIBSTDCXX;
  if (n > 0 && n <= kRankSmall && !libstd) {
    hipLaunchKernelGGL(rank_small_kernel<1>, dim3(1), dim3(1024), 0, s, n, dim, fixed_axis, inner, prob, label,
                       static_cast<const float*>(nullptr), has_ignore, ignore_label, auc_out,
                       static_cast<float*>(nullptr), static_cast<int*>(nullptr));
    return launch_status();
  }
  const RankWs lay = rank_ws(n);
  if (!ws || ws_bytes < lay.temp) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  const unsigned grid = (unsigned)((n + 255) / 256);
  // score of item i: prob[i*dim + fixed_axis]  (auc_layer.cpp:75-76 with inner_num = 1)
  hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(256), 0, s, n, dim, fixed_axis, inner, prob,
                     static_cast<const float*>(nullptr),
                     reinterpret_cast<unsigned long long*>(base + lay.keys0),
                     reinterpret_cast<unsigned*>(base + lay.vals0));
  int rc = sort_pairs(lay, base, ws_bytes, n, 32u, s);
  if (rc != MMS_OK) return rc;
  if (libstd) {
    auto* keys = reinterpret_cast<unsigned long long*>(base + lay.keys1);
    auto* vals = reinterpret_cast<unsigned*>(base + lay.vals1);
    auto* labpos = reinterpret_cast<float*>(base + lay.labpos);
    auto* work = reinterpret_cast<int*>(base + lay.work);
    auto* nwork = reinterpret_cast<int*>(base + lay.nwork);
    if (hipMemsetAsync(nwork, 0, sizeof(int), s) != hipSuccess) return MMS_ERR_LAUNCH;
    hipLaunchKernelGGL(rank_ties_detect_kernel, dim3(grid), dim3(256), 0, s, n, keys, vals, label, labpos, work, nwork);
    hipLaunchKernelGGL(rank_ties_emulate_kernel, dim3(1), dim3(64), 0, s, dim, fixed_axis, inner, prob, label, vals,
                       work, nwork, has_ignore, ignore_label, labpos, reinterpret_cast<SortItem*>(base + lay.titems),
                       reinterpret_cast<unsigned*>(base + lay.tidx));
    hipLaunchKernelGGL(auc_fold_kernel, dim3(1), dim3(64), 0, s, n, static_cast<const unsigned*>(nullptr), labpos,
                       has_ignore, ignore_label, auc_out);
    return launch_status();
  }
  hipLaunchKernelGGL(auc_fold_kernel, dim3(1), dim3(64), 0, s, n,
                     reinterpret_cast<unsigned*>(base + lay.vals1), label, has_ignore, ignore_label,
                     auc_out);
  return launch_status();
}

int rank_accuracy(int count, const float* a, const float* b, const float* label, float* out,
                  void* ws, size_t ws_bytes, hipStream_t s) {
  int blocks = (count + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  if (!ws || ws_bytes < (size_t)blocks * sizeof(unsigned)) return MMS_ERR_WORKSPACE;
  unsigned* partial = static_cast<unsigned*>(ws);
  hipLaunchKernelGGL(rank_accuracy_kernel, dim3(blocks), dim3(256), 0, s, count, a, b, label, partial);
  hipLaunchKernelGGL(rank_accuracy_finish_kernel, dim3(1), dim3(64), 0, s, blocks, count, partial, out);
  return launch_status();
}

}  // namespace mms
