// csrc/libstdcxx_sort.h -- libstdc++'s std::sort, restated step by step for (score, label) items under the
// comparator of the reference's ranking layers, `lhs.first > rhs.first` (map_layer.cpp:33-38, mrr_layer.cpp,
// auc_layer.cpp).  std::sort is not stable: where EQUAL scores carry different labels, MAP / MRR / AUC depend on the
// order this particular algorithm leaves them in.  Followed here (GCC's bits/stl_algo.h and bits/stl_heap.h):
//   __sort -> __introsort_loop (depth limit 2*floor(log2 n), threshold 16, __unguarded_partition_pivot with
//   __move_median_to_first(first, first+1, mid, last-1), heap sort via __partial_sort when the limit is hit)
//   -> __final_insertion_sort (__insertion_sort on the first 16, __unguarded_insertion_sort on the rest).
// Host-and-device code: tests/test_libstdcxx_sort.py compiles it with g++ and checks it against the real
// std::sort (random, tied and median-of-three-adversarial sequences); ranking.hip runs it on one lane per bucket
// in MMS_RANK_TIES_LIBSTDCXX mode.
#ifndef MMS_LIBSTDCXX_SORT_H_
#define MMS_LIBSTDCXX_SORT_H_

#ifndef MMS_HD
#ifdef __HIPCC__
#define MMS_HD __host__ __device__ __forceinline__
#else
#define MMS_HD inline
#endif
#endif

namespace mms {

struct SortItem { float key; int lab; };

MMS_HD bool ls_comp(const SortItem& a, const SortItem& b) { return a.key > b.key; }
MMS_HD void ls_swap(SortItem* v, long a, long b) { const SortItem t = v[a]; v[a] = v[b]; v[b] = t; }

MMS_HD void ls_unguarded_linear_insert(SortItem* v, long last) {
  const SortItem val = v[last];
  long next = last - 1;
  while (ls_comp(val, v[next])) { v[last] = v[next]; last = next; --next; }
  v[last] = val;
}
MMS_HD void ls_insertion_sort(SortItem* v, long first, long last) {
  if (first == last) return;
  for (long i = first + 1; i != last; ++i) {
    if (ls_comp(v[i], v[first])) {
      const SortItem val = v[i];
      for (long k = i; k > first; --k) v[k] = v[k - 1];      // move_backward(first, i, i + 1)
      v[first] = val;
    } else {
      ls_unguarded_linear_insert(v, i);
    }
  }
}
MMS_HD void ls_push_heap(SortItem* v, long first, long hole, long top, SortItem value) {
  long parent = (hole - 1) / 2;
  while (hole > top && ls_comp(v[first + parent], value)) {
    v[first + hole] = v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[first + hole] = value;
}
MMS_HD void ls_adjust_heap(SortItem* v, long first, long hole, long len, SortItem value) {
  const long top = hole;
  long second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (ls_comp(v[first + second], v[first + (second - 1)])) --second;
    v[first + hole] = v[first + second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    v[first + hole] = v[first + (second - 1)];
    hole = second - 1;
  }
  ls_push_heap(v, first, hole, top, value);
}
MMS_HD void ls_heap_sort(SortItem* v, long first, long last) {   // __partial_sort(first, last, last)
  const long len = last - first;
  if (len >= 2) {                                                // __make_heap
    long parent = (len - 2) / 2;
    while (true) {
      const SortItem value = v[first + parent];
      ls_adjust_heap(v, first, parent, len, value);
      if (parent == 0) break;
      --parent;
    }
  }
  while (last - first > 1) {                                     // __sort_heap: __pop_heap(first, last, last)
    --last;
    const SortItem value = v[last];
    v[last] = v[first];
    ls_adjust_heap(v, first, 0, last - first, value);
  }
}
MMS_HD long ls_partition_pivot(SortItem* v, long first, long last) {
  const long mid = first + (last - first) / 2;
  const long a = first + 1, b = mid, c = last - 1;               // __move_median_to_first(first, a, b, c)
  if (ls_comp(v[a], v[b])) {
    if (ls_comp(v[b], v[c])) ls_swap(v, first, b);
    else if (ls_comp(v[a], v[c])) ls_swap(v, first, c);
    else ls_swap(v, first, a);
  } else if (ls_comp(v[a], v[c])) ls_swap(v, first, a);
  else if (ls_comp(v[b], v[c])) ls_swap(v, first, c);
  else ls_swap(v, first, b);
  long lo = first + 1, hi = last;                                // __unguarded_partition(first + 1, last, first)
  while (true) {
    while (ls_comp(v[lo], v[first])) ++lo;
    --hi;
    while (ls_comp(v[first], v[hi])) --hi;
    if (!(lo < hi)) return lo;
    ls_swap(v, lo, hi);
    ++lo;
  }
}
// std::sort(v, v + n, lhs.first > rhs.first).  `stack`: room for 2 * 64 longs (pending right-hand ranges)
MMS_HD void libstdcxx_sort(SortItem* v, long n) {
  if (n <= 0) return;
  long lg = 0;
  while ((1L << (lg + 1)) <= n) ++lg;                            // std::__lg
  // __introsort_loop, its recursion on [cut, last) unrolled on an explicit stack; the recursive call is made
  // BEFORE the loop continues on [first, cut), so right-hand ranges are finished first -- the order matters only
  // through the depth limit each range inherits, which the stack carries
  long sf[128], sl[128], sd[128];
  int sp = 0;
  sf[0] = 0; sl[0] = n; sd[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    long first = sf[sp], last = sl[sp], depth = sd[sp];
    while (last - first > 16) {
      if (depth == 0) { ls_heap_sort(v, first, last); break; }
      --depth;
      const long cut = ls_partition_pivot(v, first, last);
      sf[sp] = cut; sl[sp] = last; sd[sp] = depth; ++sp;         // the recursive call's range
      last = cut;
    }
  }
  if (n > 16) {                                                  // __final_insertion_sort
    ls_insertion_sort(v, 0, 16);
    for (long i = 16; i != n; ++i) ls_unguarded_linear_insert(v, i);
  } else {
    ls_insertion_sort(v, 0, n);
  }
}

}  // namespace mms
#endif  // MMS_LIBSTDCXX_SORT_H_
