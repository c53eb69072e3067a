// csrc/f64_paths.hip -- the `double` instantiation of the path.
//
// The reference instantiates every layer for float AND double
// (INSTANTIATE_CLASS, include/caffe/common.hpp:41-44).  fp64 is not where this
// workload lives (the driver trains in float), so these kernels are FUNCTIONAL,
// not tuned: one thread per output element, the reference's loop order inside
// it, grid-stride over the outputs.  What they guarantee is the numerics:
//   * Euclidean forward / backward and PairRankLoss elementwise terms and the
//     bilinear dbias: the reference's operation order => bit-identical to the
//     CPU code (-ffp-contract=off, IEEE f64 sqrt / divide);
//   * cosine, bilinear, SimMatrix: k-ascending dot products (one legal BLAS
//     order), compared at 1e-12 relative in the tests.
// Same entry-point semantics as the _f32 functions of include/mms.h.
#include "mms_common.h"

namespace mms {
namespace {

typedef long long i64;
constexpr int kT = 256;

inline unsigned blocks_for(i64 n) {
  i64 b = (n + kT - 1) / kT;
  if (b < 1) b = 1;
  if (b > 65536) b = 65536;
  return (unsigned)b;
}
#define MMS_GRID_LOOP(i, n) \
  for (i64 i = (i64)blockIdx.x * kT + threadIdx.x; i < (n); i += (i64)gridDim.x * kT)

// ---------------------------------------------------------------- SimCross, modes 0 / 1
__global__ __launch_bounds__(kT) void d_row_norm(const double* __restrict__ x, double* __restrict__ nrm,
                                                 i64 rows, int D) {
  MMS_GRID_LOOP(r, rows) {
    const double* p = x + r * D;
    double s = 0;
    for (int d = 0; d < D; ++d) s += p[d] * p[d];
    nrm[r] = sqrt(s);
  }
}

template <int MODE>
__global__ __launch_bounds__(kT) void d_cross_fwd(int N, int W1, int W2, int D, const double* __restrict__ q,
                                                  const double* __restrict__ a, const double* __restrict__ n0,
                                                  const double* __restrict__ n1, double* __restrict__ top) {
  const i64 total = (i64)N * W1 * W2;
  MMS_GRID_LOOP(t, total) {
    const int k = (int)(t % W2), j = (int)((t / W2) % W1);
    const i64 n = t / ((i64)W1 * W2);
    const double* x = q + (n * W1 + j) * D;
    const double* y = a + (n * W2 + k) * D;
    double s = 0;
    if (MODE == 1) {
      for (int d = 0; d < D; ++d) { const double df = x[d] - y[d]; s += df * df; }   // :100-105
      top[t] = 1 / (1 + sqrt(s));                                                     // :106-107
    } else {
      for (int d = 0; d < D; ++d) s += x[d] * y[d];
      top[t] = s / n0[n * W1 + j] / n1[n * W2 + k];                                   // :135
    }
  }
}

// dq[n,j,d] = sum_k tt (k ascending); SIDE = 1: da[n,k,d] = sum_j (-tt or the cosine term), j ascending
template <int MODE, int SIDE>
__global__ __launch_bounds__(kT) void d_cross_bwd(int N, int W1, int W2, int D, const double* __restrict__ q,
                                                  const double* __restrict__ a, const double* __restrict__ top,
                                                  const double* __restrict__ dT, const double* __restrict__ n0,
                                                  const double* __restrict__ n1, double* __restrict__ out) {
  const int Wo = SIDE ? W2 : W1, Wi = SIDE ? W1 : W2;
  const i64 total = (i64)N * Wo * D;
  MMS_GRID_LOOP(e, total) {
    const int d = (int)(e % D), o = (int)((e / D) % Wo);
    const i64 n = e / ((i64)Wo * D);
    double acc = 0;                                  // :176-177 zero, then += in index order
    for (int i = 0; i < Wi; ++i) {
      const int j = SIDE ? i : o, k = SIDE ? o : i;
      const i64 t = (n * W1 + j) * W2 + k;
      const double qv = q[(n * W1 + j) * D + d], av = a[(n * W2 + k) * D + d];
      const double T = top[t], g = dT[t];
      double tt;
      if (MODE == 1) {
        tt = g * T * T * T * (qv - av) / (T - 1 + 1e-9);          // :214-216
        if (SIDE) tt = -tt;
      } else {
        const double a0 = n0[n * W1 + j], a1 = n1[n * W2 + k];
        tt = SIDE ? g * (qv / a0 / a1 - av * T / (a1 * a1))       // :243-246
                  : g * (av / a0 / a1 - qv * T / (a0 * a0));      // :238-241
      }
      acc += tt;
    }
    out[e] = acc;
  }
}

// ---------------------------------------------------------------- SimCross mode 2 (bilinear)
// tmp[n,m,j,e] = sum_d Q[n,j,d] W[m,d,e]
__global__ __launch_bounds__(kT) void d_qw(int N, int M, int W1, int D, const double* __restrict__ q,
                                           const double* __restrict__ W, double* __restrict__ tmp) {
  const i64 total = (i64)N * M * W1 * D;
  MMS_GRID_LOOP(x, total) {
    const int e = (int)(x % D), j = (int)((x / D) % W1), m = (int)((x / ((i64)D * W1)) % M);
    const i64 n = x / ((i64)D * W1 * M);
    const double* qr = q + (n * W1 + j) * D;
    const double* w = W + (i64)m * D * D + e;
    double s = 0;
    for (int d = 0; d < D; ++d) s += qr[d] * w[(i64)d * D];
    tmp[x] = s;
  }
}
// top[n,m,j,k] = sum_e tmp[n,m,j,e] A[n,k,e] (+ bias[m,j,k])
__global__ __launch_bounds__(kT) void d_bil_top(int N, int M, int W1, int W2, int D, const double* __restrict__ tmp,
                                                const double* __restrict__ a, const double* __restrict__ bias,
                                                double* __restrict__ top) {
  const i64 total = (i64)N * M * W1 * W2;
  MMS_GRID_LOOP(t, total) {
    const int k = (int)(t % W2), j = (int)((t / W2) % W1), m = (int)((t / ((i64)W2 * W1)) % M);
    const i64 n = t / ((i64)W2 * W1 * M);
    const double* r = tmp + ((n * M + m) * W1 + j) * D;
    const double* ar = a + (n * W2 + k) * D;
    double s = 0;
    for (int e = 0; e < D; ++e) s += r[e] * ar[e];
    top[t] = bias ? bias[((i64)m * W1 + j) * W2 + k] + s : s;       // :155-159
  }
}
// t1[n,m,d,k] = sum_j Q[n,j,d] dT[n,m,j,k]
__global__ __launch_bounds__(kT) void d_qt_dt(int N, int M, int W1, int W2, int D, const double* __restrict__ q,
                                              const double* __restrict__ dT, double* __restrict__ t1) {
  const i64 total = (i64)N * M * D * W2;
  MMS_GRID_LOOP(x, total) {
    const int k = (int)(x % W2), d = (int)((x / W2) % D), m = (int)((x / ((i64)W2 * D)) % M);
    const i64 n = x / ((i64)W2 * D * M);
    double s = 0;
    for (int j = 0; j < W1; ++j) s += q[(n * W1 + j) * D + d] * dT[((n * M + m) * W1 + j) * W2 + k];
    t1[x] = s;
  }
}
// dW[m,d,e] = sum_n sum_k t1[n,m,d,k] A[n,k,e]   (W.diff is overwritten: :256 zeroes it first)
__global__ __launch_bounds__(kT) void d_dw(int N, int M, int W2, int D, const double* __restrict__ t1,
                                           const double* __restrict__ a, double* __restrict__ dW) {
  const i64 total = (i64)M * D * D;
  MMS_GRID_LOOP(x, total) {
    const int e = (int)(x % D), d = (int)((x / D) % D), m = (int)(x / ((i64)D * D));
    double acc = 0;
    for (i64 n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < W2; ++k) s += t1[((n * M + m) * D + d) * W2 + k] * a[(n * W2 + k) * D + e];
      acc = s + acc;                                                 // gemm beta = 1, n ascending
    }
    dW[x] = acc;
  }
}
// t2[n,m,d,k] = sum_e W[m,d,e] A[n,k,e]
__global__ __launch_bounds__(kT) void d_w_at(int N, int M, int W2, int D, const double* __restrict__ W,
                                             const double* __restrict__ a, double* __restrict__ t2) {
  const i64 total = (i64)N * M * D * W2;
  MMS_GRID_LOOP(x, total) {
    const int k = (int)(x % W2), d = (int)((x / W2) % D), m = (int)((x / ((i64)W2 * D)) % M);
    const i64 n = x / ((i64)W2 * D * M);
    const double* w = W + ((i64)m * D + d) * D;
    const double* ar = a + (n * W2 + k) * D;
    double s = 0;
    for (int e = 0; e < D; ++e) s += w[e] * ar[e];
    t2[x] = s;
  }
}
// dq[n,j,d] = sum_m sum_k dT[n,m,j,k] t2[n,m,d,k]
__global__ __launch_bounds__(kT) void d_bil_dq(int N, int M, int W1, int W2, int D, const double* __restrict__ dT,
                                               const double* __restrict__ t2, double* __restrict__ dq) {
  const i64 total = (i64)N * W1 * D;
  MMS_GRID_LOOP(x, total) {
    const int d = (int)(x % D), j = (int)((x / D) % W1);
    const i64 n = x / ((i64)D * W1);
    double acc = 0;
    for (int m = 0; m < M; ++m) {
      double s = 0;
      for (int k = 0; k < W2; ++k) s += dT[((n * M + m) * W1 + j) * W2 + k] * t2[((n * M + m) * D + d) * W2 + k];
      acc = s + acc;
    }
    dq[x] = acc;
  }
}
// da[n,k,e] = sum_m sum_j dT[n,m,j,k] tmp[n,m,j,e]
__global__ __launch_bounds__(kT) void d_bil_da(int N, int M, int W1, int W2, int D, const double* __restrict__ dT,
                                               const double* __restrict__ tmp, double* __restrict__ da) {
  const i64 total = (i64)N * W2 * D;
  MMS_GRID_LOOP(x, total) {
    const int e = (int)(x % D), k = (int)((x / D) % W2);
    const i64 n = x / ((i64)D * W2);
    double acc = 0;
    for (int m = 0; m < M; ++m) {
      double s = 0;
      for (int j = 0; j < W1; ++j) s += dT[((n * M + m) * W1 + j) * W2 + k] * tmp[((n * M + m) * W1 + j) * D + e];
      acc = s + acc;
    }
    da[x] = acc;
  }
}
// dbias[m,j,k] = dT[n,m,j,k] + dbias, n ascending (:301-304: accumulated into the existing diff)
__global__ __launch_bounds__(kT) void d_dbias(int N, i64 per, const double* __restrict__ dT, double* __restrict__ dbias) {
  MMS_GRID_LOOP(x, per) {
    double acc = dbias[x];
    for (i64 n = 0; n < N; ++n) acc = dT[n * per + x] + acc;
    dbias[x] = acc;
  }
}
__global__ __launch_bounds__(kT) void d_fill0(double* __restrict__ p, i64 n) {
  MMS_GRID_LOOP(i, n) p[i] = 0;
}

// ---------------------------------------------------------------- SimMatrix
// scratch[i,c] = sum_r Q[i,r] W[r,c]
__global__ __launch_bounds__(kT) void d_sm_qw(int N, int K1, int K2, const double* __restrict__ q,
                                              const double* __restrict__ W, double* __restrict__ scratch) {
  const i64 total = (i64)N * K2;
  MMS_GRID_LOOP(x, total) {
    const int c = (int)(x % K2);
    const i64 i = x / K2;
    double s = 0;
    for (int r = 0; r < K1; ++r) s += q[i * K1 + r] * W[(i64)r * K2 + c];
    scratch[x] = s;
  }
}
__global__ __launch_bounds__(kT) void d_sm_top(int N, int K2, const double* __restrict__ a,
                                               const double* __restrict__ scratch, double* __restrict__ top) {
  MMS_GRID_LOOP(i, (i64)N) {
    double s = 0;
    for (int c = 0; c < K2; ++c) s += a[i * K2 + c] * scratch[i * K2 + c];
    top[i] = s;
  }
}
// dW[r,c] += sum_i (dT_i q[i,r]) a[i,c], i ascending (N sger's, :75-78)
__global__ __launch_bounds__(kT) void d_sm_dw(int N, int K1, int K2, const double* __restrict__ q,
                                              const double* __restrict__ a, const double* __restrict__ dT,
                                              double* __restrict__ dW) {
  const i64 total = (i64)K1 * K2;
  MMS_GRID_LOOP(x, total) {
    const int c = (int)(x % K2), r = (int)(x / K2);
    double acc = dW[x];
    for (i64 i = 0; i < N; ++i) acc += (dT[i] * q[i * K1 + r]) * a[i * K2 + c];
    dW[x] = acc;
  }
}
// SIDE 0: dq[i,r] = dT_i * sum_c W[r,c] a[i,c];  SIDE 1: da[i,c] = dT_i * sum_r W[r,c] q[i,r]
template <int SIDE>
__global__ __launch_bounds__(kT) void d_sm_dx(int N, int K1, int K2, const double* __restrict__ x,
                                              const double* __restrict__ W, const double* __restrict__ dT,
                                              double* __restrict__ out) {
  const int Ko = SIDE ? K2 : K1, Ki = SIDE ? K1 : K2;
  const i64 total = (i64)N * Ko;
  MMS_GRID_LOOP(e, total) {
    const int o = (int)(e % Ko);
    const i64 i = e / Ko;
    double s = 0;
    for (int u = 0; u < Ki; ++u) s += (SIDE ? W[(i64)u * K2 + o] : W[(i64)o * K2 + u]) * x[i * Ki + u];
    out[e] = dT[i] * s;
  }
}

// ---------------------------------------------------------------- PairRankLoss
__global__ __launch_bounds__(kT) void d_pair_fwd(int count, double margin, const double* __restrict__ a,
                                                 const double* __restrict__ b, const double* __restrict__ y,
                                                 double* __restrict__ ordered, double* __restrict__ similar) {
  MMS_GRID_LOOP(i, (i64)count) {
    const double diff = a[i] - b[i];
    similar[i] = diff;
    double o = diff * y[i];
    o = -1.0 * o + 0.0 * o;
    ordered[i] = o + margin;
  }
}
// The reference's loss is a sequential sum over i (:40-50); reproduced by ONE thread so the
// double loss is bit-identical too (count is N*C: thousands of terms).
__global__ void d_pair_loss(int count, const double* __restrict__ y, const double* __restrict__ ordered,
                            const double* __restrict__ similar, double* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double l = 0;
  for (int i0 = 0; i0 < count; i0 += 8) {        // eight elements requested before the first add
    double o[8], yy[8], sm[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u, count - 1);
      o[u] = ordered[i]; yy[u] = y[i]; sm[u] = similar[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < count) l += (o[u] > 0.0 ? o[u] : 0.0) + fabs((1 - yy[u]) * sm[u]);
  }
  *loss = l / (double)count;
}
__global__ __launch_bounds__(kT) void d_pair_bwd(int count, double s0, double s1, const double* __restrict__ y,
                                                 const double* __restrict__ ordered,
                                                 const double* __restrict__ similar, double* __restrict__ da,
                                                 double* __restrict__ db) {
  MMS_GRID_LOOP(e, (i64)count) {
    const double ot = ordered[e] > 0 ? 1.0 : 0.0;
    const double st = (1 - y[e]) * similar[e] > 0 ? 1.0 : -1.0;
    const double inner = ot * y[e] - st * (1 - y[e]);
    if (da) da[e] = s0 * inner;
    if (db) db[e] = s1 * inner;
  }
}

#define L(kernel, n, ...) hipLaunchKernelGGL(kernel, dim3(blocks_for(n)), dim3(kT), 0, s, __VA_ARGS__)

}  // namespace

size_t simcross_workspace_bytes_f64(int mode, int N, int W1, int W2, int D, int M) {
  if (mode != 2) return 0;
  const size_t wmax = (size_t)(W1 > W2 ? W1 : W2);
  return 2 * sizeof(double) * (size_t)N * M * wmax * D;     // tmp (Q W) and t1 / t2
}

int simcross_forward_f64(int mode, int N, int W1, int W2, int D, int M, const double* q, const double* a,
                         const double* W, const double* bias, double* top, double* norm0, double* norm1,
                         void* ws, size_t ws_bytes, hipStream_t s) {
  if (N == 0) return MMS_OK;
  if (mode == 1) {
    L((d_cross_fwd<1>), (i64)N * W1 * W2, N, W1, W2, D, q, a, nullptr, nullptr, top);
  } else if (mode == 0) {
    L(d_row_norm, (i64)N * W1, q, norm0, (i64)N * W1, D);
    L(d_row_norm, (i64)N * W2, a, norm1, (i64)N * W2, D);
    L((d_cross_fwd<0>), (i64)N * W1 * W2, N, W1, W2, D, q, a, norm0, norm1, top);
  } else {
    if (!ws || ws_bytes < simcross_workspace_bytes_f64(2, N, W1, W2, D, M)) return MMS_ERR_WORKSPACE;
    double* tmp = static_cast<double*>(ws);
    L(d_qw, (i64)N * M * W1 * D, N, M, W1, D, q, W, tmp);
    L(d_bil_top, (i64)N * M * W1 * W2, N, M, W1, W2, D, tmp, a, bias, top);
  }
  return launch_status();
}

int simcross_backward_f64(int mode, int N, int W1, int W2, int D, int M, const double* q, const double* a,
                          const double* W, int bias_term, const double* top, const double* top_diff,
                          const double* norm0, const double* norm1, int pd0, int pd1, double* dq, double* da,
                          double* dW, double* dbias, void* ws, size_t ws_bytes, hipStream_t s) {
  if (N == 0) return MMS_OK;
  if (!(pd0 || pd1)) {                      // :176-177: zeroed unconditionally, nothing else happens
    L(d_fill0, (i64)N * W1 * D, dq, (i64)N * W1 * D);
    L(d_fill0, (i64)N * W2 * D, da, (i64)N * W2 * D);
    return launch_status();
  }
  if (mode == 1) {
    L((d_cross_bwd<1, 0>), (i64)N * W1 * D, N, W1, W2, D, q, a, top, top_diff, nullptr, nullptr, dq);
    L((d_cross_bwd<1, 1>), (i64)N * W2 * D, N, W1, W2, D, q, a, top, top_diff, nullptr, nullptr, da);
  } else if (mode == 0) {
    L((d_cross_bwd<0, 0>), (i64)N * W1 * D, N, W1, W2, D, q, a, top, top_diff, norm0, norm1, dq);
    L((d_cross_bwd<0, 1>), (i64)N * W2 * D, N, W1, W2, D, q, a, top, top_diff, norm0, norm1, da);
  } else {
    if (!ws || ws_bytes < simcross_workspace_bytes_f64(2, N, W1, W2, D, M)) return MMS_ERR_WORKSPACE;
    const size_t wmax = (size_t)(W1 > W2 ? W1 : W2);
    double* tmp = static_cast<double*>(ws);
    double* t12 = tmp + (size_t)N * M * wmax * D;
    L(d_qt_dt, (i64)N * M * D * W2, N, M, W1, W2, D, q, top_diff, t12);
    L(d_dw, (i64)M * D * D, N, M, W2, D, t12, a, dW);
    L(d_w_at, (i64)N * M * D * W2, N, M, W2, D, W, a, t12);
    L(d_bil_dq, (i64)N * W1 * D, N, M, W1, W2, D, top_diff, t12, dq);
    L(d_qw, (i64)N * M * W1 * D, N, M, W1, D, q, W, tmp);
    L(d_bil_da, (i64)N * W2 * D, N, M, W1, W2, D, top_diff, tmp, da);
    if (bias_term && dbias) L(d_dbias, (i64)M * W1 * W2, N, (i64)M * W1 * W2, top_diff, dbias);
  }
  return launch_status();
}

int simmatrix_forward_f64(int N, int K1, int K2, const double* q, const double* a, const double* W,
                          double* top, double* scratch, hipStream_t s) {
  if (N == 0) return MMS_OK;
  L(d_sm_qw, (i64)N * K2, N, K1, K2, q, W, scratch);
  L(d_sm_top, (i64)N, N, K2, a, scratch, top);
  return launch_status();
}

int simmatrix_backward_f64(int N, int K1, int K2, const double* q, const double* a, const double* W,
                           const double* top_diff, int ppd, int pd0, int pd1, double* dq, double* da,
                           double* dW, hipStream_t s) {
  if (N == 0) return MMS_OK;
  if (ppd && dW) L(d_sm_dw, (i64)K1 * K2, N, K1, K2, q, a, top_diff, dW);
  if (pd0 && dq) L((d_sm_dx<0>), (i64)N * K1, N, K1, K2, a, W, top_diff, dq);
  if (pd1 && da) L((d_sm_dx<1>), (i64)N * K2, N, K1, K2, q, W, top_diff, da);
  return launch_status();
}

int pairrank_forward_f64(int count, double margin, const double* a, const double* b, const double* y,
                         double* ordered, double* similar, double* loss, hipStream_t s) {
  if (count == 0) return MMS_OK;
  L(d_pair_fwd, (i64)count, count, margin, a, b, y, ordered, similar);
  hipLaunchKernelGGL(d_pair_loss, dim3(1), dim3(64), 0, s, count, y, ordered, similar, loss);
  return launch_status();
}

int pairrank_backward_f64(int count, double top_diff, const double* y, const double* ordered,
                          const double* similar, double* da, double* db, hipStream_t s) {
  if (count == 0 || (!da && !db)) return MMS_OK;
  const double scale = top_diff / count;
  L(d_pair_bwd, (i64)count, count, -1.0 * scale, 1.0 * scale, y, ordered, similar, da, db);
  return launch_status();
}

}  // namespace mms
