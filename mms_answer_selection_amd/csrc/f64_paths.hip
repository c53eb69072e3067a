// csrc/f64_paths.hip -- the `double` instantiation of the path.
//
// The reference instantiates every layer for float AND double
// (INSTANTIATE_CLASS, include/caffe/common.hpp:41-44).  fp64 is not where this
// workload lives (the driver trains in float).  The order-defined paths (Euclid,
// PairRankLoss, dbias) are FUNCTIONAL kernels: one thread per output element, the
// reference's loop order inside it, grid-stride over the outputs.  Round 2: every
// GEMM-shaped product of the bilinear mode and of SimMatrix runs on the fp64 matrix
// pipe (gemm64_kernel below, v_mfma_f64_16x16x4_f64: each MFMA is an fmaf chain in
// double, no reduced precision) -- 16384 x 300 x 300 SimMatrix forward 1394 -> 122 us,
// backward 11.6 ms -> 0.59 ms; the driver's 50 x 40 x 40 x 50 M = 4 bilinear backward 671 -> 133 us.  What they guarantee is the numerics:
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

// Sentence-vector geometry (W1 = W2 = 1), Euclidean: the d-ascending sum is a dependent chain per pair, so a THREAD
// owns a pair -- but it reads its two rows a 128-byte line at a time (sixteen doubles per operand as eight 16-byte
// loads, the next tile requested before the current one is summed) instead of one double per round trip:
// 4096 x 300 forward 69 -> 12 us, same bits (:100-107 order).
__global__ __launch_bounds__(64) void d_euclid_rows_fwd(int N, int D, const double* __restrict__ q,
                                                        const double* __restrict__ a, double* __restrict__ top) {
  const i64 n = (i64)blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const double* x = q + n * D;
  const double* y = a + n * D;
  typedef double d2 __attribute__((ext_vector_type(2)));
  double s = 0;
  int d = 0;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (vec && D >= 16) {
    d2 xa[8], ya[8], xb[8], yb[8];
    auto load = [&](d2 (&xv)[8], d2 (&yv)[8], int at) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xv[u] = *reinterpret_cast<const d2*>(x + at + 2 * u);
        yv[u] = *reinterpret_cast<const d2*>(y + at + 2 * u);
      }
    };
    auto sum = [&](const d2 (&xv)[8], const d2 (&yv)[8]) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double f0 = xv[u].x - yv[u].x, f1 = xv[u].y - yv[u].y;
        s += f0 * f0;
        s += f1 * f1;
      }
    };
    const int nt = D / 16;                           // whole 128-byte tiles; xa always holds tile t
    load(xa, ya, 0);
    int t = 0;
    for (; t + 2 <= nt; t += 2) {
      load(xb, yb, 16 * (t + 1));
      sum(xa, ya);
      if (t + 2 < nt) load(xa, ya, 16 * (t + 2));
      sum(xb, yb);
    }
    if (t < nt) sum(xa, ya);
    d = 16 * nt;
  }
  for (; d < D; ++d) { const double df = x[d] - y[d]; s += df * df; }
  top[n] = 1 / (1 + sqrt(s));
}

// Sentence-vector geometry, cosine (dot and norms are BLAS-ordered in the reference: any order, 1e-12): one WAVE
// per pair, lanes stride over d, three wave sums.
__global__ __launch_bounds__(kT) void d_cosine_rows_fwd(int N, int D, const double* __restrict__ q,
                                                        const double* __restrict__ a, double* __restrict__ n0,
                                                        double* __restrict__ n1, double* __restrict__ top) {
  const i64 n = (i64)blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  const double* x = q + n * D;
  const double* y = a + n * D;
  double sxx = 0, syy = 0, sxy = 0;
  for (int d = lane; d < D; d += 64) { const double u = x[d], v = y[d]; sxx += u * u; syy += v * v; sxy += u * v; }
  for (int o = 32; o; o >>= 1) { sxx += __shfl_xor(sxx, o); syy += __shfl_xor(syy, o); sxy += __shfl_xor(sxy, o); }
  if (lane == 0) {
    const double a0 = sqrt(sxx), a1 = sqrt(syy);
    n0[n] = a0; n1[n] = a1;
    top[n] = sxy / a0 / a1;                                                           // :135
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

// ---------------------------------------------------------------- fp64 GEMM on the matrix pipe
// C[b0,b1] (M x N) (+)= epilogue( sum_{seg, kk} A[m, (seg,kk)] * B[(seg,kk), n] )
//   * any element strides for A and B (row- or column-major operands, transposes are strides); the threads
//     that stage a tile run along whichever stride is 1, so both layouts load coalesced;
//   * K = nseg segments of L: segment `seg` of A / B starts a_seg / b_seg elements further on (the sums over
//     pairs n or measures m of the bilinear backward are ONE product with a segmented K);
//   * two batch indices (b0 = blockIdx.z / nb1, b1 = blockIdx.z % nb1) with separate strides for A, B, C, the
//     bias matrix and the row-scale vector (A varies with the pair only, W with the measure only, ...);
//   * epilogue: acc *= rowscale[m] (SimMatrix's dT_i), acc = bias[m,n] + acc (:155-159), C = acc or C += acc;
//   * kscale[k]: A[m,k] is multiplied by kscale[k] as it is staged ((dT_i q[i,r]) of sim_matrix_layer.cpp:75-78).
// 64 x 64 tile per workgroup of four waves (32 x 32 = 2 x 2 MFMA blocks each), K tiles of 16 through LDS.
struct G64 {
  int M, N, L, nseg;
  const double* A; i64 sam, sak, a_seg, a_b0, a_b1;
  const double* B; i64 sbk, sbn, b_seg, b_b0, b_b1;
  double* C; i64 ldc, c_b0, c_b1;
  int nb1, beta;
  const double* rowscale; i64 rs_b0;
  const double* kscale;
  const double* bias; i64 ldbias, bias_b1;
};
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int G64_T = 64, G64_K = 16, G64_LS = G64_T + 4;

__global__ __launch_bounds__(256) void gemm64_kernel(G64 g) {
  __shared__ double As[G64_K * G64_LS];             // [k][m]
  __shared__ double Bs[G64_K * G64_LS];             // [k][n]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int b0 = blockIdx.z / g.nb1, b1 = blockIdx.z - b0 * g.nb1;
  const int m0 = blockIdx.y * G64_T, n0 = blockIdx.x * G64_T;
  const double* A = g.A + b0 * g.a_b0 + b1 * g.a_b1;
  const double* B = g.B + b0 * g.b_b0 + b1 * g.b_b1;
  // staging maps: the fastest thread index follows the unit stride
  const bool a_kfast = (g.sak == 1), b_nfast = (g.sbn == 1) || (g.sbk != 1);
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  const int li = lane & 15, lg = lane >> 4;
  v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  // K tiles in order (seg major): tile i+1's global loads are issued before tile i's MFMAs
  const int tiles_per_seg = (g.L + G64_K - 1) / G64_K, ntiles = g.nseg * tiles_per_seg;
  double va[4], vb[4];
  auto fetch = [&](int tile) {
    const int seg = tile / tiles_per_seg, k0 = (tile - seg * tiles_per_seg) * G64_K;
    const double* As_g = A + seg * g.a_seg;
    const double* Bs_g = B + seg * g.b_seg;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u;
      const int kk = a_kfast ? (e & 15) : (e >> 6), mm = a_kfast ? (e >> 4) : (e & 63);
      const int k = k0 + kk, m = m0 + mm;
      const bool ok = k < g.L && m < g.M;
      double v = As_g[(i64)(ok ? m : 0) * g.sam + (i64)(ok ? k : 0) * g.sak];
      if (g.kscale) v = g.kscale[seg * (i64)g.L + (ok ? k : 0)] * v;
      va[u] = ok ? v : 0.0;
      const int kb = b_nfast ? (e >> 6) : (e & 15), nn = b_nfast ? (e & 63) : (e >> 4);
      const int k2 = k0 + kb, n = n0 + nn;
      const bool ok2 = k2 < g.L && n < g.N;
      const double w = Bs_g[(i64)(ok2 ? k2 : 0) * g.sbk + (i64)(ok2 ? n : 0) * g.sbn];
      vb[u] = ok2 ? w : 0.0;
    }
  };
  fetch(0);
  for (int tile = 0; tile < ntiles; ++tile) {
    __syncthreads();                                 // the previous tile's MFMAs have read As / Bs
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u;
      As[(a_kfast ? (e & 15) : (e >> 6)) * G64_LS + (a_kfast ? (e >> 4) : (e & 63))] = va[u];
      Bs[(b_nfast ? (e >> 6) : (e & 15)) * G64_LS + (b_nfast ? (e & 63) : (e >> 4))] = vb[u];
    }
    __syncthreads();
    if (tile + 1 < ntiles) fetch(tile + 1);
#pragma unroll
    for (int ks = 0; ks < G64_K; ks += 4) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = As[(ks + lg) * G64_LS + wm + 16 * i + li];
        b[i] = Bs[(ks + lg) * G64_LS + wn + 16 * i + li];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  double* C = g.C + b0 * g.c_b0 + b1 * g.c_b1;
  const double* rs = g.rowscale ? g.rowscale + b0 * g.rs_b0 : nullptr;
  const double* bias = g.bias ? g.bias + b1 * g.bias_b1 : nullptr;
  // what the epilogue reads is in registers before its first store (clamped addresses keep the loads
  // unconditional): element by element, each load waited with vmcnt(0) for the store in front of it
  double rsv[2][4], bsv[2][2][4], cvv[2][2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const i64 mc = min(m0 + wm + 16 * i + 4 * r + lg, g.M - 1);
      rsv[i][r] = rs ? rs[mc] : 1.0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const i64 nc = min(n0 + wn + 16 * j + li, g.N - 1);
        bsv[i][j][r] = bias ? bias[mc * g.ldbias + nc] : 0.0;
        cvv[i][j][r] = g.beta ? C[mc * g.ldc + nc] : 0.0;
      }
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      asm volatile("" : "+v"(rsv[i][r]));
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(bsv[i][j][r]), "+v"(cvv[i][j][r]));
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // D layout of v_mfma_f64_16x16x4_f64: register r of lane l holds row 4*r + l/16, column l%16 (four groups
        // of one row per lane -- unlike the fp32 16x16x4, whose lanes hold four CONSECUTIVE rows)
        const int m = m0 + wm + 16 * i + 4 * r + lg, n = n0 + wn + 16 * j + li;
        if (m < g.M && n < g.N) {
          double v = acc[i][j][r];
          if (rs) v = rsv[i][r] * v;
          if (bias) v = bsv[i][j][r] + v;
          double* c = C + (i64)m * g.ldc + n;
          *c = g.beta ? cvv[i][j][r] + v : v;
        }
      }
}

// The same product for SMALL outputs with a LONG inner dimension (SimMatrix's dW: 300 x 300 from K = 16384 rows;
// the bilinear dW: D x D from K = N*W2): a 64 x 64 tiling gives 25 workgroups that each walk 1024 K tiles one
// memory round trip at a time (3.1 ms).  Here a workgroup of SIXTEEN waves owns a 32 x 32 tile and every wave
// walks its own interleaved share of K, operands straight from global memory into the MFMA registers (no LDS,
// no barrier in the loop, sixteen independent load streams per tile); the sixteen partial tiles are then added
// in wave order through LDS -- a fixed order, so the result is deterministic.
constexpr int G64_TK = 32, G64_TKW = 16;
__global__ __launch_bounds__(64 * G64_TKW) void gemm64_tallk_kernel(G64 g) {
  __shared__ double red[G64_TK * G64_TK];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int b0 = blockIdx.z / g.nb1, b1 = blockIdx.z - b0 * g.nb1;
  const int m0 = blockIdx.y * G64_TK, n0 = blockIdx.x * G64_TK;
  const double* A = g.A + b0 * g.a_b0 + b1 * g.a_b1;
  const double* B = g.B + b0 * g.b_b0 + b1 * g.b_b1;
  const int li = lane & 15, lg = lane >> 4;
  const i64 Ktot = (i64)g.nseg * g.L;
  v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  const int mi[2] = {m0 + li, m0 + 16 + li}, ni[2] = {n0 + li, n0 + 16 + li};
  constexpr int UN = 4;                              // k steps in flight per wave
  for (i64 base = (i64)wave * 4 * UN; base < Ktot; base += (i64)G64_TKW * 4 * UN) {
    double a[UN][2], b[UN][2];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const i64 kf = base + 4 * u + lg;
      const bool kok = kf < Ktot;
      const i64 seg = kok ? kf / g.L : 0, kk = kok ? kf - seg * g.L : 0;
      const double ks = (g.kscale && kok) ? g.kscale[kf] : 1.0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = kok && mi[i] < g.M;
        const double v = A[(i64)(ok ? mi[i] : 0) * g.sam + seg * g.a_seg + kk * g.sak];
        a[u][i] = ok ? (g.kscale ? ks * v : v) : 0.0;
        const bool ok2 = kok && ni[i] < g.N;
        const double w = B[seg * g.b_seg + kk * g.sbk + (i64)(ok2 ? ni[i] : 0) * g.sbn];
        b[u][i] = ok2 ? w : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
  }
  for (int w = 0; w < G64_TKW; ++w) {                // partial tiles added in wave order
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = (16 * i + 4 * r + lg) * G64_TK + 16 * j + li;
            red[e] = (w == 0) ? acc[i][j][r] : red[e] + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  double* C = g.C + b0 * g.c_b0 + b1 * g.c_b1;
  const double* rs = g.rowscale ? g.rowscale + b0 * g.rs_b0 : nullptr;
  const double* bias = g.bias ? g.bias + b1 * g.bias_b1 : nullptr;
  {
    const int e = t;                                 // 1024 threads, 1024 outputs
    const int m = m0 + e / G64_TK, n = n0 + e % G64_TK;
    if (m < g.M && n < g.N) {
      double v = red[e];
      if (rs) v = rs[m] * v;
      if (bias) v = bias[(i64)m * g.ldbias + n] + v;
      double* c = C + (i64)m * g.ldc + n;
      *c = g.beta ? *c + v : v;
    }
  }
}

inline G64 g64(int M, int N, int K, const double* A, i64 sam, i64 sak, const double* B, i64 sbk, i64 sbn,
               double* C, i64 ldc) {
  G64 g{};
  g.M = M; g.N = N; g.L = K; g.nseg = 1;
  g.A = A; g.sam = sam; g.sak = sak;
  g.B = B; g.sbk = sbk; g.sbn = sbn;
  g.C = C; g.ldc = ldc; g.nb1 = 1;
  return g;
}
inline void run_g64(const G64& g, int nb0, hipStream_t s) {
  const i64 tiles64 = (i64)((g.N + G64_T - 1) / G64_T) * ((g.M + G64_T - 1) / G64_T) * nb0 * g.nb1;
  if ((i64)g.nseg * g.L >= 1024 && tiles64 <= 64) {  // small output, long K: sixteen K streams per tile
    const dim3 gr((unsigned)((g.N + G64_TK - 1) / G64_TK), (unsigned)((g.M + G64_TK - 1) / G64_TK),
                  (unsigned)(nb0 * g.nb1));
    hipLaunchKernelGGL(gemm64_tallk_kernel, gr, dim3(64 * G64_TKW), 0, s, g);
    return;
  }
  const dim3 grid((unsigned)((g.N + G64_T - 1) / G64_T), (unsigned)((g.M + G64_T - 1) / G64_T),
                  (unsigned)(nb0 * g.nb1));
  hipLaunchKernelGGL(gemm64_kernel, grid, dim3(256), 0, s, g);
}

// top[i] = sum_c a[i,c] scratch[i,c], c ascending within a lane, lanes summed in a fixed tree: one wave per row
__global__ __launch_bounds__(kT) void d_sm_top_wave(int N, int K2, const double* __restrict__ a,
                                                    const double* __restrict__ scratch, double* __restrict__ top) {
  const i64 row = (i64)blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  double s = 0;
  for (int c = lane; c < K2; c += 64) s += a[row * K2 + c] * scratch[row * K2 + c];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) top[row] = s;
}

// ---------------------------------------------------------------- SimCross mode 2 (bilinear)
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

// tmp[n,m] (W1 x D) = Q_n (W1 x D) . W_m (D x D)
static void bil_qw(int N, int M, int W1, int D, const double* q, const double* W, double* tmp, hipStream_t s) {
  G64 g = g64(W1, D, D, q, D, 1, W, D, 1, tmp, D);
  g.nb1 = M;
  g.a_b0 = (i64)W1 * D;
  g.b_b1 = (i64)D * D;
  g.c_b0 = (i64)M * W1 * D; g.c_b1 = (i64)W1 * D;
  run_g64(g, N, s);
}

size_t simcross_workspace_bytes_f64(int mode, int N, int W1, int W2, int D, int M) {
  if (mode != 2) return 0;
  const size_t wmax = (size_t)(W1 > W2 ? W1 : W2);
  return 2 * sizeof(double) * (size_t)N * M * wmax * D;     // tmp (Q W) and t1 / t2
}

int simcross_forward_f64(int mode, int N, int W1, int W2, int D, int M, const double* q, const double* a,
                         const double* W, const double* bias, double* top, double* norm0, double* norm1,
                         void* ws, size_t ws_bytes, hipStream_t s) {
  if (N == 0) return MMS_OK;
  const bool rows = (W1 == 1 && W2 == 1);
  if (mode == 1 && rows) {
    hipLaunchKernelGGL(d_euclid_rows_fwd, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, N, D, q, a, top);
  } else if (mode == 1) {
    L((d_cross_fwd<1>), (i64)N * W1 * W2, N, W1, W2, D, q, a, nullptr, nullptr, top);
  } else if (mode == 0 && rows) {
    hipLaunchKernelGGL(d_cosine_rows_fwd, dim3((unsigned)((N + 3) / 4)), dim3(kT), 0, s, N, D, q, a, norm0, norm1, top);
  } else if (mode == 0) {
    L(d_row_norm, (i64)N * W1, q, norm0, (i64)N * W1, D);
    L(d_row_norm, (i64)N * W2, a, norm1, (i64)N * W2, D);
    L((d_cross_fwd<0>), (i64)N * W1 * W2, N, W1, W2, D, q, a, norm0, norm1, top);
  } else {
    if (!ws || ws_bytes < simcross_workspace_bytes_f64(2, N, W1, W2, D, M)) return MMS_ERR_WORKSPACE;
    double* tmp = static_cast<double*>(ws);
    bil_qw(N, M, W1, D, q, W, tmp, s);
    // top[n,m] (W1 x W2) = tmp[n,m] (W1 x D) . A_n^T (+ bias[m])
    G64 g = g64(W1, W2, D, tmp, D, 1, a, 1, D, top, W2);
    g.nb1 = M;
    g.a_b0 = (i64)M * W1 * D; g.a_b1 = (i64)W1 * D;
    g.b_b0 = (i64)W2 * D;
    g.c_b0 = (i64)M * W1 * W2; g.c_b1 = (i64)W1 * W2;
    g.bias = bias; g.ldbias = W2; g.bias_b1 = (i64)W1 * W2;
    run_g64(g, N, s);
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
    {  // t1[n,m] (D x W2) = Q_n^T (D x W1) . dT[n,m] (W1 x W2)
      G64 g = g64(D, W2, W1, q, 1, D, top_diff, W2, 1, t12, W2);
      g.nb1 = M;
      g.a_b0 = (i64)W1 * D;
      g.b_b0 = (i64)M * W1 * W2; g.b_b1 = (i64)W1 * W2;
      g.c_b0 = (i64)M * D * W2; g.c_b1 = (i64)D * W2;
      run_g64(g, N, s);
    }
    {  // dW[m] (D x D) = sum_n t1[n,m] (D x W2) . A_n (W2 x D): K = N segments of W2   (:256 zeroes W.diff first)
      G64 g = g64(D, D, W2, t12, W2, 1, a, D, 1, dW, D);
      g.nseg = N; g.a_seg = (i64)M * D * W2; g.b_seg = (i64)W2 * D;
      g.a_b0 = (i64)D * W2;                           // b0 = measure m
      g.c_b0 = (i64)D * D;
      run_g64(g, M, s);
    }
    {  // t2[n,m] (D x W2) = W_m (D x D) . A_n^T
      G64 g = g64(D, W2, D, W, D, 1, a, 1, D, t12, W2);
      g.nb1 = M;
      g.a_b1 = (i64)D * D;
      g.b_b0 = (i64)W2 * D;
      g.c_b0 = (i64)M * D * W2; g.c_b1 = (i64)D * W2;
      run_g64(g, N, s);
    }
    {  // dq[n] (W1 x D) = sum_m dT[n,m] (W1 x W2) . t2[n,m]^T: K = M segments of W2
      G64 g = g64(W1, D, W2, top_diff, W2, 1, t12, 1, W2, dq, D);
      g.nseg = M; g.a_seg = (i64)W1 * W2; g.b_seg = (i64)D * W2;
      g.a_b0 = (i64)M * W1 * W2; g.b_b0 = (i64)M * D * W2; g.c_b0 = (i64)W1 * D;
      run_g64(g, N, s);
    }
    bil_qw(N, M, W1, D, q, W, tmp, s);
    {  // da[n] (W2 x D) = sum_m dT[n,m]^T (W2 x W1) . tmp[n,m] (W1 x D): K = M segments of W1
      G64 g = g64(W2, D, W1, top_diff, 1, W2, tmp, D, 1, da, D);
      g.nseg = M; g.a_seg = (i64)W1 * W2; g.b_seg = (i64)W1 * D;
      g.a_b0 = (i64)M * W1 * W2; g.b_b0 = (i64)M * W1 * D; g.c_b0 = (i64)W2 * D;
      run_g64(g, N, s);
    }
    if (bias_term && dbias) L(d_dbias, (i64)M * W1 * W2, N, (i64)M * W1 * W2, top_diff, dbias);
  }
  return launch_status();
}

int simmatrix_forward_f64(int N, int K1, int K2, const double* q, const double* a, const double* W,
                          double* top, double* scratch, hipStream_t s) {
  if (N == 0) return MMS_OK;
  run_g64(g64(N, K2, K1, q, K1, 1, W, K2, 1, scratch, K2), 1, s);           // scratch = Q W   (:55-58)
  hipLaunchKernelGGL(d_sm_top_wave, dim3((unsigned)((N + 3) / 4)), dim3(kT), 0, s, N, K2, a, scratch, top);
  return launch_status();
}

int simmatrix_backward_f64(int N, int K1, int K2, const double* q, const double* a, const double* W,
                           const double* top_diff, int ppd, int pd0, int pd1, double* dq, double* da,
                           double* dW, hipStream_t s) {
  if (N == 0) return MMS_OK;
  if (ppd && dW) {                                   // dW += (diag(dT) Q)^T A   (:75-78, accumulated)
    G64 g = g64(K1, K2, N, q, 1, K1, a, K2, 1, dW, K2);
    g.kscale = top_diff; g.beta = 1;
    run_g64(g, 1, s);
  }
  if (pd0 && dq) {                                   // dq = diag(dT) (A W^T)   (:81-86)
    G64 g = g64(N, K1, K2, a, K2, 1, W, 1, K2, dq, K1);
    g.rowscale = top_diff;
    run_g64(g, 1, s);
  }
  if (pd1 && da) {                                   // da = diag(dT) (Q W)     (:88-93)
    G64 g = g64(N, K2, K1, q, K1, 1, W, K2, 1, da, K2);
    g.rowscale = top_diff;
    run_g64(g, 1, s);
  }
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
