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

// ---- 512 < n <= 2048 (round 3): the same one-workgroup metric with a sort that stays in registers -------------------
// The LDS bitonic sort above pays a workgroup barrier and two LDS round trips for each of its log^2 passes: at 2,048
// padded items that is 66 passes and 47 us, no better than rocPRIM's launches -- while 1,517 candidates are exactly
// the TREC-QA test split a TEST net ranks per pass (cfg 4).  Here thread t of 1,024 holds items 2t and 2t + 1 in
// registers; a compare-exchange at distance 1 is inside the thread, at distances 2..64 a wave shuffle (no barrier, no
// LDS allocation), and only distances 128..1024 -- 10 of the 66 passes -- go through LDS with one barrier each
// (ping-pong buffers).  Then the sorted (key, index) pairs are written to LDS once and the SAME bucket walks and the
// SAME one-wave fold run on them: same expressions, same order, same bits.
constexpr int kRankMid = 2048;
template <int MODE>                                  // 0: MAP / MRR, 1: AUC
__global__ __launch_bounds__(1024) void rank_mid_kernel(int n, int stride, int offset, int inner,
                                                        const float* __restrict__ prob,
                                                        const float* __restrict__ label,
                                                        const float* __restrict__ group, int has_ignore,
                                                        int ignore_label, float* __restrict__ out0,
                                                        float* __restrict__ out1, int* __restrict__ effective) {
  __shared__ unsigned long long keys[kRankMid];
  __shared__ unsigned vals[kRankMid];
  __shared__ unsigned long long xk[kRankMid];          // the second exchange buffer (cross-wave passes)
  __shared__ unsigned xv[kRankMid];
  __shared__ float ap[MODE == 0 ? kRankMid : 1];
  __shared__ int rk[MODE == 0 ? kRankMid : 1];
  __shared__ int fl[MODE == 0 ? kRankMid : 1];
  __shared__ float slab[MODE == 0 ? kRankMid : 1];
  const int t = threadIdx.x;
#ifdef MMS_RANK_STAMPS      // dev-only (tools/rank_mid_probe.py): phase times in 10-ns ticks behind the results, out0[4..]
  unsigned long long st0 = __builtin_amdgcn_s_memrealtime(), st1 = 0, st2 = 0, st3 = 0, st4 = 0;
#endif
  unsigned long long k[2];
  unsigned v[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = 2 * t + e;
    k[e] = i < n ? rank_key(i, stride, offset, inner, prob, group) : ~0ull;   // padding sorts behind everything
    v[e] = (unsigned)i;
  }
  // (key, index) in lexicographic order; `lower`: this element sits at the smaller position of its pair
  auto keep_mine = [](unsigned long long km, unsigned vm, unsigned long long ko, unsigned vo, bool lower, bool up) {
    const bool mine_gt = km > ko || (km == ko && vm > vo);
    return (mine_gt == (lower == up)) ? false : true;   // ascending pair keeps the smaller one at the lower position
  };
#ifdef MMS_RANK_STAMPS
  st1 = __builtin_amdgcn_s_memrealtime();
#endif
  // Sort = (1) each wave sorts its 128 items in registers (bitonic network of 28 passes; the partner of a lane at
  // distance 1 / 2 / 4 / 8 / 16 / 32 comes by DPP or v_permlane{16,32}_swap, no LDS), then (2) four MERGE rounds
  // 128 -> 256 -> 512 -> 1024 -> 2048: every item finds by binary search how many items of the sibling run precede it
  // ((key, index) pairs are distinct, so "precede" needs no tie rule) and is written to its merged position in the other
  // LDS buffer.  8 + 9 + 10 + 11 dependent LDS reads and four barriers instead of the 66 passes (56 of them 6
  // ds_bpermute each, 10 through LDS with a barrier) of one flat bitonic network: 19 -> 11.8 us at 1,517 items on a lone
  // workgroup (stamps, tools/rank_mid_probe.py: the 38 search steps are bound by the LDS round trip at that clock).
  const int lane_s = t & 63;
  auto lane_xor = [&](unsigned x, int m) -> unsigned {          // the value lane (lane ^ m) holds
    switch (m) {
      case 1: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
      case 2: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
      case 4: {
        const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0xf, true);   // row_shl:4: lane i <- i + 4
        const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4: lane i <- i - 4
        return (lane_s & 4) ? dn : up;
      }
      case 8: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);     // row_ror:8 (= xor 8 in a row of 16)
      case 16: { const auto sw = __builtin_amdgcn_permlane16_swap(x, x, false, false); return sw[0] ^ sw[1] ^ x; }
      default: { const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false); return sw[0] ^ sw[1] ^ x; }
    }
  };
#pragma unroll
  for (int kk = 2; kk <= 128; kk <<= 1) {
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
      if (j == 1) {                                   // local positions 2 lane and 2 lane + 1: inside the thread
        const bool up = ((2 * lane_s) & kk) == 0;
        const bool gt = k[0] > k[1] || (k[0] == k[1] && v[0] > v[1]);
        if (gt == up) {
          const unsigned long long tk = k[0]; k[0] = k[1]; k[1] = tk;
          const unsigned tv = v[0]; v[0] = v[1]; v[1] = tv;
        }
      } else {
        const int m = j >> 1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int p = 2 * lane_s + e;               // position inside the wave's block of 128
          const unsigned lo = lane_xor((unsigned)k[e], m), hi = lane_xor((unsigned)(k[e] >> 32), m);
          const unsigned vo = lane_xor(v[e], m);
          const unsigned long long ko = ((unsigned long long)hi << 32) | lo;
          const bool lower = (p & j) == 0, up = (p & kk) == 0;
          if (!keep_mine(k[e], v[e], ko, vo, lower, up)) { k[e] = ko; v[e] = vo; }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) { keys[2 * t + e] = k[e]; vals[2 * t + e] = v[e]; }
  int buf = 0;                                        // 0: the runs are in keys / vals, 1: in xk / xv
#pragma unroll 1
  for (int L = 128; L < kRankMid; L <<= 1) {
    __syncthreads();
    const unsigned long long* ck = buf ? xk : keys;
    const unsigned* cv = buf ? xv : vals;
    unsigned long long* nk = buf ? keys : xk;
    unsigned* nv = buf ? vals : xv;
    int cnt[2], sbase[2];
    unsigned long long mk[2];
    unsigned mv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pos = 2 * t + e;
      mk[e] = ck[pos]; mv[e] = cv[pos];
      sbase[e] = (pos & ~(L - 1)) ^ L;                // first position of the sibling run
      cnt[e] = 0;
    }
    for (int sstep = L >> 1; sstep >= 1; sstep >>= 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = sbase[e] + cnt[e] + sstep - 1;
        const unsigned long long ok = ck[q];
        const unsigned ov = cv[q];
        if (ok < mk[e] || (ok == mk[e] && ov < mv[e])) cnt[e] += sstep;
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = sbase[e] + cnt[e];                // (cnt <= L - 1 here: one more look decides cnt == L)
      const unsigned long long ok = ck[q];
      const unsigned ov = cv[q];
      if (ok < mk[e] || (ok == mk[e] && ov < mv[e])) cnt[e] += 1;
      const int pos = 2 * t + e;
      const int np = (pos & ~(2 * L - 1)) + (pos & (L - 1)) + cnt[e];
      nk[np] = mk[e]; nv[np] = mv[e];
    }
    buf ^= 1;
  }
  __syncthreads();
  {
    const unsigned long long* ck = buf ? xk : keys;
    const unsigned* cv = buf ? xv : vals;
#pragma unroll
    for (int e = 0; e < 2; ++e) { k[e] = ck[2 * t + e]; v[e] = cv[2 * t + e]; }
  }
  __syncthreads();                                    // the last readers of `keys` / `vals` as exchange buffers are done
#ifdef MMS_RANK_STAMPS
  st2 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
  for (int e = 0; e < 2; ++e) { keys[2 * t + e] = k[e]; vals[2 * t + e] = v[e]; }
  __syncthreads();
  if (MODE == 0) {
    // A lone workgroup runs at a fraction of the chip's loaded clock: the per-bucket walks of rank_bucket_at (one
    // thread per bucket, a data-dependent exit and a division per item: two LDS round trips + ~40 dependent
    // instructions per iteration) took 15.8 us at 1,517 items and the one-wave fold with its double divisions 7.6 us
    // (stamps, tools/rank_mid_probe.py).  Same expressions and the same summation orders, but everything that is not
    // a running fp sum now happens in parallel: bucket heads, lengths and positive counts by workgroup scans, every
    // positive's term map_rank / (pos + 1) and every bucket's ap / map_rank and 1.0 / (mrr_rank + 1) by its own
    // thread; what is left sequential is adds over loads whose addresses are known up front.
    int* hd = rk;                                    // compacted bucket heads, hd[B] = n  (rk / fl / ap are free until the end)
    float* term = ap;
    __shared__ int wtot[2][16];
    __shared__ int nb_s;
    const int i0 = 2 * t, i1 = 2 * t + 1, lane = t & 63, wv = t >> 6;
    // labels in sorted order (the gather through the permutation: global loads, all independent)
    const int l0 = i0 < n ? (int)label[vals[i0]] : 0, l1 = i1 < n ? (int)label[vals[i1]] : 0;
    const unsigned g0 = (unsigned)(k[0] >> 32), g1 = (unsigned)(k[1] >> 32);
    const unsigned gprev = __shfl_up(g1, 1, 64);
    unsigned gp = gprev;                             // group of position i0 - 1: the previous thread's second item
    if (lane == 0) gp = t ? (unsigned)(keys[i0 - 1] >> 32) : 0u;
    const int h0 = i0 < n && (i0 == 0 || gp != g0), h1 = i1 < n && g1 != g0;
    // two workgroup scans at once over the items in position order: (a) bucket number = inclusive count of heads,
    // (b) inclusive count of label == 1
    int a0 = h0, a1 = h0 + h1, b0 = (l0 == 1), b1 = b0 + (l1 == 1);
    int sa = a1, sb = b1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int oa = __shfl_up(sa, d, 64), ob = __shfl_up(sb, d, 64);
      if (lane >= d) { sa += oa; sb += ob; }
    }
    if (lane == 63) { wtot[0][wv] = sa; wtot[1][wv] = sb; }
    __syncthreads();
    int pa = sa - a1, pb = sb - b1;                  // exclusive prefix inside the wave
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wv) { pa += wtot[0][w]; pb += wtot[1][w]; } }
    a0 += pa; a1 += pa; b0 += pb; b1 += pb;          // inclusive counts at positions i0, i1
    if (t == 1023) nb_s = a1;                        // (padding positions add nothing)
    slab[i0] = __int_as_float(l0); slab[i1] = __int_as_float(l1);     // labels as ints, position order
    xv[i0] = (unsigned)b0; xv[i1] = (unsigned)b1;                       // positive counts (the exchange buffer is free)
    if (h0) hd[a0 - 1] = i0;
    if (h1) hd[a1 - 1] = i1;
    __syncthreads();
    const int B = nb_s;
    if (t == 0) hd[B] = n;
    __syncthreads();
    // every positive's term, in parallel: (++map_rank) / (float)(pos + 1), map_layer.cpp:84-86
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = e ? i1 : i0;
      if (i < n) {
        const int bkt = (e ? a1 : a0) - 1, head = hd[bkt];
        const int before = head ? (int)xv[head - 1] : 0;               // positives in front of the bucket
        const int lab = e ? l1 : l0, mr = (int)(e ? b1 : b0) - before;
        term[i] = lab == 1 ? mr / (float)(i - head + 1) : 0.f;
      }
    }
    __syncthreads();
    // one thread per bucket: the ordered sum of its terms (x + 0.0f == x: the non-positive items add nothing), its
    // flags and its two quotients.  The bucket's extent is known, so the loads run ahead of the adds.
    float b_ap = 0.f;
    double b_inv = 0.0;
    int b_fl = 0;
    if (t < B) {
      const int head = hd[t], end = hd[t + 1];
      float apv = 0.f;
      int not_one = 0, zero = 0, first = -1;
      for (int p = head; p < end; p += 8) {
        float tv[8];
        int lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int q = p + u < end ? p + u : end - 1; tv[u] = term[q]; lv[u] = __float_as_int(slab[q]); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (p + u < end) {
            apv += tv[u];
            if (lv[u] == 1) { if (first < 0) first = p + u - head; }
            else { not_one = 1; if (lv[u] == 0) zero = 1; }
          }
        }
      }
      const int map_rank = (int)xv[end - 1] - (head ? (int)xv[head - 1] : 0);
      if (map_rank >= 1 && not_one) { b_fl |= 1; b_ap = apv / map_rank; }                 // map_layer.cpp:92-94
      if (first >= 0 && zero) { b_fl |= 2; b_inv = 1.0 / (first + 1); }                   // mrr_layer.cpp:75
    }
#ifdef MMS_RANK_STAMPS
    st3 = __builtin_amdgcn_s_memrealtime();
#endif
    // the fold over the buckets in sorted (ascending group) order: running sums only, one wave, 64 buckets per step
    float* bap = reinterpret_cast<float*>(keys);     // (the sorted keys are not needed any more)
    double* binv = reinterpret_cast<double*>(xk);
    int* bfl = fl;
    if (t < B) { bap[t] = b_ap; binv[t] = b_inv; bfl[t] = b_fl; }
    __syncthreads();
    if (t < 64) {
      float map_ = 0.f, mrr = 0.f;
      int eff_map = 0, eff_mrr = 0;
      for (int base = 0; base < B; base += 64) {
        const int bi = base + lane < B ? base + lane : B - 1;
        const float la = bap[bi];
        const long long li = __double_as_longlong(binv[bi]);
        const int lf = base + lane < B ? bfl[bi] : 0;
        unsigned long long m1 = __ballot(lf & 1), m2 = __ballot(lf & 2);
        eff_map += __popcll(m1);
        eff_mrr += __popcll(m2);
        while (m1) {                                                   // map_layer.cpp:93-94
          const int l = __ffsll((long long)m1) - 1;
          m1 &= m1 - 1;
          map_ += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(la), l));
        }
        while (m2) {   // mrr += 1.0/(mrr_rank+1): float + double, stored back to float (mrr_layer.cpp:75)
          const int l = __ffsll((long long)m2) - 1;
          m2 &= m2 - 1;
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(li & 0xffffffffll), l);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(li >> 32), l);
          mrr = (float)((double)mrr + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)));
        }
      }
      if (lane == 0) {
        if (out0) *out0 = map_ / eff_map;             // NaN when no bucket counts, like the reference (:99)
        if (out1) *out1 = mrr / eff_mrr;
        if (effective) *effective = eff_map;
#ifdef MMS_RANK_STAMPS
        st4 = __builtin_amdgcn_s_memrealtime();
        out0[4] = (float)(st1 - st0); out0[5] = (float)(st2 - st1); out0[6] = (float)(st3 - st2); out0[7] = (float)(st4 - st3);
#endif
      }
    }
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
  if (n > kRankSmall && n <= kRankMid && !libstd) {  // a test split's worth (1,517 candidates): still one launch
    hipLaunchKernelGGL(rank_mid_kernel<0>, dim3(1), dim3(1024), 0, s, n, fixed_axis + 1, fixed_axis, 1, prob,
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
  if (n > kRankSmall && n <= kRankMid && !libstd) {
    hipLaunchKernelGGL(rank_mid_kernel<1>, dim3(1), dim3(1024), 0, s, n, dim, fixed_axis, inner, prob, label,
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
