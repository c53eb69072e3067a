// tests/libstdcxx_sort_check.cpp -- compiled and run by tests/test_libstdcxx_sort.py: csrc/libstdcxx_sort.h against the
// real std::sort under the ranking layers' comparator (lhs.first > rhs.first), payload = original position, on random,
// heavily tied, ascending, organ-pipe and adversarial sequences (the heap-sort fallback is hit thousands of times).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "libstdcxx_sort.h"
static bool cmp(const std::pair<float,int>& a, const std::pair<float,int>& b) { return a.first > b.first; }
int main() {
  std::mt19937 rng(1);
  long bad = 0, runs = 0;
  for (int rep = 0; rep < 20000; ++rep) {
    int n = rep < 100 ? rep : (int)(rng() % (rep % 7 == 0 ? 5000 : 200)) + 1;
    int levels = 1 + (int)(rng() % (rep % 3 == 0 ? 3 : (rep % 3 == 1 ? 40 : 100000)));
    std::vector<std::pair<float,int>> a(n); std::vector<mms::SortItem> b(n);
    int kind = rep % 5;
    for (int i = 0; i < n; ++i) {
      float k = (float)(rng() % levels) / levels;
      if (kind == 3) k = (float)i / n;          // ascending (worst case for descending order)
      if (kind == 4) k = (float)((i * 7919) % 97) ; 
      a[i] = {k, i}; b[i] = {k, i};
    }
    if (kind == 2 && n > 40) {                   // median-of-three killer-like: organ pipe
      for (int i = 0; i < n; ++i) { float k = (float)(i < n / 2 ? i : n - i); a[i].first = k; b[i].key = k; }
    }
    std::sort(a.begin(), a.end(), cmp);
    mms::libstdcxx_sort(b.data(), n);
    ++runs;
    for (int i = 0; i < n; ++i) if (a[i].second != b[i].lab || a[i].first != b[i].key) { ++bad; break; }
  }
  // force the heap-sort fallback: adversarial input built against THIS implementation (McIlroy's antiquicksort)
  {
    int n = 4000; std::vector<int> val(n, -1); int nsolid = 0, candidate = 0; const int gas = n;
    std::vector<int> idx(n); for (int i = 0; i < n; ++i) idx[i] = i;
    auto key = [&](int i) { return val[i] < 0 ? gas : val[i]; };
    auto acmp = [&](int x, int y) {               // descending comparator on adversarial keys: x > y
      if (val[x] < 0 && val[y] < 0) { if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++; }
      if (val[x] < 0) candidate = x; else if (val[y] < 0) candidate = y;
      return key(x) > key(y);
    };
    std::sort(idx.begin(), idx.end(), acmp);
    std::vector<std::pair<float,int>> a(n); std::vector<mms::SortItem> b(n);
    for (int i = 0; i < n; ++i) { float k = (float)(key(i) / 3); a[i] = {k, i}; b[i] = {k, i}; }   // /3: ties
    std::sort(a.begin(), a.end(), cmp); mms::libstdcxx_sort(b.data(), n); ++runs;
    for (int i = 0; i < n; ++i) if (a[i].second != b[i].lab) { ++bad; break; }
  }
  std::printf("runs %ld mismatches %ld\n", runs, bad);
  return bad != 0;
}
