// csrc/bilinear.hip -- the learned-metric (bilinear W) paths on the gfx950
// matrix cores: SimCross dist_mode 2 and SimMatrix, forward and backward.
//
// Reference:
//   SimCross mode 2 fwd  sim_cross_layer.cpp:140-161   T[n,m] = Q_n W_m A_n^T (+ bias_m)
//   SimCross mode 2 bwd  sim_cross_layer.cpp:251-305   dW_m = sum_n Q_n^T dT_nm A_n (W.diff zeroed),
//                         dQ_n += dT_nm (W_m A_n^T)^T, dA_n += dT_nm^T (Q_n W_m), dbias += dT_n
//   SimMatrix fwd/bwd    sim_matrix_layer.cpp:53-65, 68-95
// The reference issues 2 (fwd) / 6 (bwd) small cblas_sgemm calls per (pair,
// measure) on the host -- even in GPU mode (sim_cross_layer.cu:187-189,
// 240-242).  Here the contraction over the embedding dimension is regrouped so
// that all pairs share ONE large GEMM per weight matrix:
//   fwd:  tmp_m = Q_all W_m            (N*W1 x D x D)    then T = tmp A^T per pair
//   bwd:  U_nm = dT_nm A_n , V_nm = dT_nm^T Q_n           (small, per pair)
//         dQ_all = sum_m U_m W_m^T , dA_all = sum_m V_m W_m   (N*W x D x D)
//         dW_m   = Q_all^T U_m                              (D x D x N*W1, split-K)
// Algebraically identical to the reference's grouping; fp32 rounding differs
// (as it does between BLAS libraries), tests hold it to 1e-5.
//
// All products use v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, each MFMA
// bit-equal to a k-ordered fmaf chain -- no reduced-precision path.
// Deterministic: split-K partial slabs are summed in a fixed order, no atomics.
#include <type_traits>

#include "mms_common.h"
#include "panel_gemm.h"
#include "bx3_gemm.h"

namespace mms {

// Dev-only phase stamps (tools/gemmstamp.hip builds this file with -DMMS_GEMM_STAMPS): thread 0 of
// every workgroup of the fast kernel records s_memtime at its phase boundaries, plus where it ran.
#ifdef MMS_GEMM_STAMPS
__device__ unsigned long long* mms_gemm_stamp_buf = nullptr;
#define MMS_GSTAMP(k)                                                                          \
  do {                                                                                         \
    if (mms_gemm_stamp_buf && threadIdx.x == 0)                                                \
      mms_gemm_stamp_buf[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define MMS_GSTAMP_REAL(k)                                                                     \
  do {                                                                                         \
    if (mms_gemm_stamp_buf && threadIdx.x == 0)                                                \
      mms_gemm_stamp_buf[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define MMS_GSTAMP_WHERE()                                                                     \
  do {                                                                                         \
    if (mms_gemm_stamp_buf && threadIdx.x == 0)                                                \
      mms_gemm_stamp_buf[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + 7] = \
          ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
  } while (0)
#else
#define MMS_GSTAMP(k) do {} while (0)
#define MMS_GSTAMP_WHERE() do {} while (0)
#define MMS_GSTAMP_REAL(k) do {} while (0)
#endif

typedef float v16f __attribute__((ext_vector_type(16)));

struct GemmArgs {
  int M, N, K;
  const float* A; long long a_rs, a_cs;  // A(i,k) = A[i*a_rs + k*a_cs]
  const float* B; long long b_rs, b_cs;  // B(k,j) = B[k*b_rs + j*b_cs]
  float* C; long long ldc;               // C(i,j) = C[i*ldc + j]
  // blockIdx.z = (b0 * nb1 + b1) * ksplit + ks
  int nb1, ksplit, kchunk;
  long long a_b0, a_b1, b_b0, b_b1, c_b0, c_b1, c_ks;
  // "stacked" split-K: chunk ks is a product of its own, A + ks*a_ks times B + ks*b_ks over k in [0, K)
  // (sum over measures of U_m W_m: K is not one contiguous axis); partials land at C + ks*c_ks as usual.
  int ks_stacked; long long a_ks, b_ks;
  const float* rowscale; long long rs_b0;  // optional C(i,j) = rowscale[i] * acc
  const float* addend; long long ad_b1;    // optional C(i,j) += addend[i*ldc + j]
  const float* bkscale;                    // optional B(k,j) *= bkscale[k] on load (fast j-vector path)
  int beta_one;                            // C = result + C
  int stream_c;                            // C is written once and not re-read soon: non-temporal stores
  int a_ifast, b_jfast;                    // which index is contiguous in memory
};

constexpr int BM = 128, BN = 64, BK = 16;
constexpr int LSA = BM + 4, LSB = BN + 4;

// 256 threads = 4 waves stacked along M; wave w owns rows [32w,32w+32) x 64 cols
// = two 32x32 MFMA tiles.  LDS tiles are k-major so a fragment read is 32
// consecutive floats per half-wave (conflict-free).
__global__ __launch_bounds__(256) void gemm32_kernel(GemmArgs g) {
  __shared__ float As[BK * LSA];
  __shared__ float Bs[BK * LSB];

  const int z = blockIdx.z;
  const int ks = z % g.ksplit;
  const int b1 = (z / g.ksplit) % g.nb1;
  const int b0 = (z / g.ksplit) / g.nb1;
  const float* A = g.A + b0 * g.a_b0 + b1 * g.a_b1;
  const float* B = g.B + b0 * g.b_b0 + b1 * g.b_b1;
  float* C = g.C + b0 * g.c_b0 + b1 * g.c_b1 + ks * g.c_ks;
  int kbeg = ks * g.kchunk;
  int kend = min(g.K, kbeg + g.kchunk);
  if (g.ks_stacked) { A += ks * g.a_ks; B += ks * g.b_ks; kbeg = 0; kend = g.K; }
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;

  // staging registers: A tile 128x16 -> 8 per thread, B tile 16x64 -> 4 per thread
  float ra[8], rb[4];
  bool oka[8], okb[4];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int i, k;
      if (g.a_ifast) { i = t & 127; k = (t >> 7) + 2 * p; }
      else { k = t & 15; i = (t >> 4) + 16 * p; }
      const int gi = i0 + i, gk = k0 + k;
      // clamped, unconditional load; zeroed at the LDS store (as `ok ? load : 0` the loads are
      // emitted one by one, each waited for: see the fast kernel below)
      oka[p] = gi < g.M && gk < kend;
      ra[p] = A[(long long)min(gi, g.M - 1) * g.a_rs + (long long)min(gk, g.K - 1) * g.a_cs];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int j, k;
      if (g.b_jfast) { j = t & 63; k = (t >> 6) + 4 * p; }
      else { k = t & 15; j = (t >> 4) + 16 * p; }
      const int gj = j0 + j, gk = k0 + k;
      okb[p] = gj < g.N && gk < kend;
      rb[p] = B[(long long)min(gk, g.K - 1) * g.b_rs + (long long)min(gj, g.N - 1) * g.b_cs];
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int i, k;
      if (g.a_ifast) { i = t & 127; k = (t >> 7) + 2 * p; }
      else { k = t & 15; i = (t >> 4) + 16 * p; }
      As[k * LSA + i] = oka[p] ? ra[p] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int j, k;
      if (g.b_jfast) { j = t & 63; k = (t >> 6) + 4 * p; }
      else { k = t & 15; j = (t >> 4) + 16 * p; }
      Bs[k * LSB + j] = okb[p] ? rb[p] : 0.f;
    }
  };

  v16f acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

  if (kbeg < kend) {
    load_tiles(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      __syncthreads();
      store_tiles();
      __syncthreads();
      if (k0 + BK < kend) load_tiles(k0 + BK);
      const int ar = wave * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        const float av = As[(kk + kh) * LSA + ar];
        const float bv0 = Bs[(kk + kh) * LSB + (lane & 31)];
        const float bv1 = Bs[(kk + kh) * LSB + 32 + (lane & 31)];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv1, acc1, 0, 0, 0);
      }
    }
  }

  // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const float* rs = g.rowscale ? g.rowscale + b0 * g.rs_b0 : nullptr;
  const float* ad = g.addend ? g.addend + b1 * g.ad_b1 : nullptr;
  // everything the epilogue reads is requested before its first store (see gemm32_fast_tile)
  float rsv[16] = {}, adv[2][16] = {}, cv[2][16] = {};
  if (rs || ad || g.beta_one) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gi = i0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const long long gic = gi < g.M ? gi : g.M - 1;
      rsv[r] = rs ? rs[gic] : 1.0f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gj = j0 + 32 * h + (lane & 31);
        const long long gjc = gj < g.N ? gj : g.N - 1;
        adv[h][r] = ad ? ad[gic * g.ldc + gjc] : 0.f;
        cv[h][r] = g.beta_one ? C[gic * g.ldc + gjc] : 0.f;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    asm volatile("" : "+v"(rsv[r]), "+v"(adv[0][r]), "+v"(adv[1][r]), "+v"(cv[0][r]), "+v"(cv[1][r]));   // in registers HERE
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gi = i0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (gi >= g.M) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int gj = j0 + 32 * h + (lane & 31);
      if (gj >= g.N) continue;
      float v = h ? acc1[r] : acc0[r];
      if (rs) v = rsv[r] * v;
      if (ad) v = adv[h][r] + v;
      float* c = C + gi * g.ldc + gj;
      if (g.beta_one) v = v + cv[h][r];
      *c = v;
    }
  }
}

// ---- fast path: 64x64x16 tiles, 16-byte global loads, permuted-k fragments --
// Eligible when each operand is contiguous in one direction with 16-byte
// alignment (A along k or along i, B along j or along k) -- true for every
// large GEMM of this file at D = 300 / 1024.  Differences from the generic
// kernel above:
//   * FK/16 float4 global loads per operand per thread per k-tile, issued for the
//     NEXT tile before the MFMAs of the current one; pointers are bumped, no
//     64-bit multiplies in the loop;
//   * A lives in LDS as [i][k] (k contiguous, row stride 20 floats).  A
//     32x32x2 MFMA consumes two k values per instruction and the pairing is
//     free as long as A and B agree, so instruction t of a tile uses
//     k = t (lanes 0-31) and k = t + FK/2 (lanes 32-63): every lane's FK/2 A
//     values are then CONTIGUOUS -- FK/8 ds_read_b128, conflict-free at stride
//     FK+4 (20 or 36 floats) -- instead of FK/2 ds_read_b32;
//   * 4 waves as 2 x 2, each one 32x32 accumulator: at M = 16384, N = 300 the
//     grid is 256 x 5 = 1280 workgroups = exactly 5 per CU (no tail).
// FK = 16 measured faster than 32 at cfg 3 (43 vs 67 us for the 16384x300x300 product):
// the shallower tile keeps more workgroups' loads in flight per CU.
// The k-tile depth FK is a template parameter: 16 when several workgroups share a CU (cfg 3: their
// MFMA phases cover each other's barriers), 32 when a product is so small that a CU holds one or
// two workgroups and every tile boundary (LDS write -> barrier -> LDS read, ~0.3 us) is exposed --
// half as many boundaries for the driver's 32 x 40 x 40 x 300 bilinear products.
constexpr int FM = 64, FN = 64, LSJ = FN + 4;

// VW = floats per global load: 4 (16-byte-aligned rows, e.g. D = 300 / 1024) or 2 (8-byte-aligned rows:
// the driver's default D = 50, whose rows are 200 bytes).
// One 64x64 output tile of one product.  AKT / BJT: 1 or 0 fix the operand layouts at compile time (the
// single-product kernel below), -1 takes them from rt_ak / rt_bj (the grouped kernel, whose problems differ).
template <int AKT, int BJT, bool KSCALE, int FK, int VW>
__device__ __forceinline__ void gemm32_fast_tile(const GemmArgs& g, int bx, int by, int z, bool rt_ak,
                                                 bool rt_bj, float* As2base, float* Bs2base) {
  const bool A_KVEC = AKT < 0 ? rt_ak : (AKT != 0);
  const bool B_JVEC = BJT < 0 ? rt_bj : (BJT != 0);
  typedef float VT __attribute__((ext_vector_type(VW)));
  constexpr int LSK = FK + 4;
  constexpr int FSL = FK / (4 * VW);   // vector load slots per operand per thread per tile
  constexpr int KV = FK / VW;          // vectors along the k extent of a tile
  constexpr int JV = 64 / VW;          // vectors along the 64-wide extent of a tile
  constexpr int FH = FK / 2;           // k values per half-wave per tile
  const int ks = z % g.ksplit;
  const int b1 = (z / g.ksplit) % g.nb1;
  const int b0 = (z / g.ksplit) / g.nb1;
  const float* A = g.A + b0 * g.a_b0 + b1 * g.a_b1;
  const float* B = g.B + b0 * g.b_b0 + b1 * g.b_b1;
  float* C = g.C + b0 * g.c_b0 + b1 * g.c_b1 + ks * g.c_ks;
  int kbeg = ks * g.kchunk;
  int kend = min(g.K, kbeg + g.kchunk);
  if (g.ks_stacked) { A += ks * g.a_ks; B += ks * g.b_ks; kbeg = 0; kend = g.K; }
  const int i0 = by * FM, j0 = bx * FN;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;

  // this thread's load slots: FSL float4 per operand per tile
  int ai[FSL], ak[FSL], bj[FSL], bk[FSL];  // tile-local coordinates
#pragma unroll
  for (int sl = 0; sl < FSL; ++sl) {
    const int u = t + 256 * sl;
    if (A_KVEC) { ai[sl] = u / KV; ak[sl] = (u % KV) * VW; } else { ak[sl] = u / JV; ai[sl] = (u % JV) * VW; }
    if (B_JVEC) { bk[sl] = u / JV; bj[sl] = (u % JV) * VW; } else { bj[sl] = u / KV; bk[sl] = (u % KV) * VW; }
  }
  const float* pa[FSL];
  const float* pb[FSL];
  bool a_ok[FSL], b_ok[FSL];
#pragma unroll
  for (int sl = 0; sl < FSL; ++sl) {
    a_ok[sl] = i0 + ai[sl] < g.M;              // M % VW == 0 on the i-vector path
    b_ok[sl] = j0 + bj[sl] < g.N;              // N % VW == 0 on the j-vector path
    pa[sl] = A + (long long)(i0 + ai[sl]) * g.a_rs + (long long)(kbeg + ak[sl]) * g.a_cs;
    pb[sl] = B + (long long)(kbeg + bk[sl]) * g.b_rs + (long long)(j0 + bj[sl]) * g.b_cs;
  }
  const long long a_step = (long long)FK * g.a_cs, b_step = (long long)FK * g.b_rs;
  const float* ksc = g.bkscale;

  VT ra[FSL], rb[FSL];
  float sc[FSL];
  bool la[FSL], lb[FSL];                       // was the slot inside the matrix?
#pragma unroll
  for (int sl = 0; sl < FSL; ++sl) sc[sl] = 1.f;
  // Out-of-range slots load from a valid address (the operand's base) and are zeroed when they
  // are WRITTEN TO LDS.  `cond ? *p : zero` instead makes the compiler select between p and the
  // address of a private zero: flat loads through scratch, each followed by vmcnt(0) -- nothing
  // stays in flight behind the MFMAs.
  auto load = [&](int k0) {
#pragma unroll
    for (int sl = 0; sl < FSL; ++sl) {
      la[sl] = a_ok[sl] && k0 + ak[sl] < kend;
      lb[sl] = b_ok[sl] && k0 + bk[sl] < kend;
      ra[sl] = *reinterpret_cast<const VT*>(la[sl] ? pa[sl] : A);
      rb[sl] = *reinterpret_cast<const VT*>(lb[sl] ? pb[sl] : B);
      // the scale is only FETCHED here (clamped index, no dependent use): multiplying now
      // would put a vmcnt(0) wait in front of the MFMAs and drain the prefetch
      if (KSCALE) sc[sl] = ksc[min(k0 + bk[sl], g.K - 1)];
      pa[sl] += a_step;
      pb[sl] += b_step;
    }
  };
  auto store = [&](int buf) {
    float* As = As2base + buf * (FM * LSK);
    float* Bs = Bs2base + buf * (FK * LSJ);
    const VT zv = 0.f;
#pragma unroll
    for (int sl = 0; sl < FSL; ++sl) {
      if (!la[sl]) ra[sl] = zv;
      if (!lb[sl]) rb[sl] = zv;
      if (A_KVEC) {
        *reinterpret_cast<VT*>(&As[ai[sl] * LSK + ak[sl]]) = ra[sl];
      } else {
#pragma unroll
        for (int c = 0; c < VW; ++c) As[(ai[sl] + c) * LSK + ak[sl]] = ra[sl][c];
      }
      if (B_JVEC) {
        VT v = rb[sl];
        if (KSCALE) v *= sc[sl];
        *reinterpret_cast<VT*>(&Bs[bk[sl] * LSJ + bj[sl]]) = v;
      } else {
#pragma unroll
        for (int c = 0; c < VW; ++c) Bs[(bk[sl] + c) * LSJ + bj[sl]] = rb[sl][c];
      }
    }
  };

  v16f acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;

  if (kbeg < kend) {
    load(kbeg);
    store(0);
    __syncthreads();
    MMS_GSTAMP(1);
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += FK, cur ^= 1) {
      const bool more = k0 + FK < kend;
      if (more) load(k0 + FK);                    // global -> registers, in flight behind the MFMAs
      const float* As = As2base + cur * (FM * LSK);
      const float* Bs = Bs2base + cur * (FK * LSJ);
      const float4* arow = reinterpret_cast<const float4*>(&As[(wm * 32 + r) * LSK + FH * h]);
      float av[FH], bv[FH];
#pragma unroll
      for (int u4 = 0; u4 < FH / 4; ++u4) {
        const float4 v = arow[u4];
        av[4 * u4] = v.x; av[4 * u4 + 1] = v.y; av[4 * u4 + 2] = v.z; av[4 * u4 + 3] = v.w;
      }
#pragma unroll
      for (int u = 0; u < FH; ++u) bv[u] = Bs[(u + FH * h) * LSJ + wn * 32 + r];
#pragma unroll
      for (int u = 0; u < FH; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      if (more) store(cur ^ 1);                   // the other buffer: nobody reads it this tile
      __syncthreads();
    }
  }

  MMS_GSTAMP(2);
  const float* rs = g.rowscale ? g.rowscale + b0 * g.rs_b0 : nullptr;
  const float* ad = g.addend ? g.addend + b1 * g.ad_b1 : nullptr;
  const int gj = j0 + wn * 32 + r;
  // Everything the epilogue reads is requested before its first store (clamped addresses keep the loads
  // unconditional).  Element by element -- load, use, store, next load -- every load waited with vmcnt(0) for
  // the acknowledgement of the store before it (C may alias what is read, so the compiler cannot hoist):
  // sixteen dependent memory round trips per thread whenever a row scale, an addend or C += was asked for.
  float rsv[16] = {}, adv[16] = {}, cv[16] = {};
  if (rs || ad || g.beta_one) {
    const long long gjc = gj < g.N ? gj : g.N - 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int gi = i0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      const long long gic = gi < g.M ? gi : g.M - 1;
      rsv[q] = rs ? rs[gic] : 1.0f;
      adv[q] = ad ? ad[gic * g.ldc + gjc] : 0.f;
      cv[q] = g.beta_one ? C[gic * g.ldc + gjc] : 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(rsv[q]), "+v"(adv[q]), "+v"(cv[q]));   // in registers HERE
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int gi = i0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (gi >= g.M || gj >= g.N) continue;
    float v = acc[q];
    if (rs) v = rsv[q] * v;
    if (ad) v = adv[q] + v;
    float* c = C + gi * g.ldc + gj;
    if (g.beta_one) *c = v + cv[q];
    else if (g.stream_c) __builtin_nontemporal_store(v, c);
    else *c = v;
  }
#ifdef MMS_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MMS_GSTAMP(3);
  MMS_GSTAMP_REAL(5);
}


template <bool A_KVEC, bool B_JVEC, bool KSCALE, int FK, int VW>
__global__ __launch_bounds__(256) void gemm32_fast_kernel(GemmArgs g) {
  __shared__ float As2[2 * FM * (FK + 4)];      // double-buffered: one barrier per k-tile
  __shared__ float Bs2[2 * FK * LSJ];
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own
  // L2) in linear-id order (x fastest, then y, then z), so linear ids that differ by 8 share an L2.
  // Remap so that CONSECUTIVE logical tiles land on one XCD: the column tiles of one row panel
  // (they re-read the same A rows), and -- for a split-K product -- all tiles of one k-chunk (each
  // re-reads the chunk's A and B slabs; dealt over 8 L2s those slabs came from HBM 6 times over).
  MMS_GSTAMP(0);
  MMS_GSTAMP_REAL(4);
  MMS_GSTAMP_WHERE();
  int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
  {
    const int plane = gridDim.x * gridDim.y, total = plane * gridDim.z;
    if ((total & 7) == 0) {
      const int id = z * plane + by * gridDim.x + bx;
      int tl = (id & 7) * (total >> 3) + (id >> 3);
      z = tl / plane;
      tl -= z * plane;
      by = tl / gridDim.x;
      bx = tl - by * gridDim.x;
    }
  }
  gemm32_fast_tile<A_KVEC ? 1 : 0, B_JVEC ? 1 : 0, KSCALE, FK, VW>(g, bx, by, z, false, false, As2, Bs2);
}

// Several SMALL products in one launch (the driver's batch of 50 pairs makes every product of the bilinear
// backward a 5-8 us launch at the latency floor: U and V, then dQ, dA and dW, are independent of each
// other).  Workgroup w of the 1-D grid belongs to the problem whose [first, first + count) holds w; operand
// layouts are run-time flags.  No XCD remapping: the problems are small by construction.
constexpr int kGroupMax = 4;
struct GemmGroup {
  GemmArgs g[kGroupMax];
  int first[kGroupMax + 1];      // first workgroup of each problem; first[n] = total
  int gx[kGroupMax], gy[kGroupMax];
  int ak[kGroupMax], bj[kGroupMax];
  int n;
};

template <int FK, int VW>
__global__ __launch_bounds__(256) void gemm32_group_kernel(GemmGroup grp) {
  __shared__ float As2[2 * FM * (FK + 4)];
  __shared__ float Bs2[2 * FK * LSJ];
  const int w = blockIdx.x;
  int p = 0;
#pragma unroll
  for (int i = 1; i < kGroupMax; ++i)
    if (i < grp.n && w >= grp.first[i]) p = i;
  int l = w - grp.first[p];
  const int plane = grp.gx[p] * grp.gy[p];
  const int z = l / plane;
  l -= z * plane;
  const int by = l / grp.gx[p], bx = l - by * grp.gx[p];
  // p is uniform: the struct members come from the kernarg segment with scalar loads at a computed offset
  gemm32_fast_tile<-1, -1, false, FK, VW>(grp.g[p], bx, by, z, grp.ak[p] != 0, grp.bj[p] != 0, As2, Bs2);
}

static bool multv(long long x, int vw) { return x % vw == 0; }
// which fast variant (if any) can run these arguments: 0 none, else 1 + 2*A_KVEC + B_JVEC; *vw = floats
// per global load (4, else 2)
static int gemm_fast_variant(const GemmArgs& g, int* vw_out = nullptr) {
  if (g.kchunk % 32 != 0 && g.ksplit > 1 && !g.ks_stacked) return 0;   // split boundaries must fall on k-tile boundaries (16 or 32)
  if (g.bkscale && !(g.b_cs == 1)) return 0;
  for (int vw = 4; vw >= 2; vw -= 2) {
    const uintptr_t am = (uintptr_t)(4 * vw - 1);
    const bool bases = (reinterpret_cast<uintptr_t>(g.A) & am) == 0 && (reinterpret_cast<uintptr_t>(g.B) & am) == 0 &&
                       multv(g.a_b0, vw) && multv(g.a_b1, vw) && multv(g.b_b0, vw) && multv(g.b_b1, vw) &&
                       (!g.ks_stacked || (multv(g.a_ks, vw) && multv(g.b_ks, vw)));
    if (!bases || !multv(g.K, vw)) continue;
    int a_kvec;
    if (g.a_cs == 1 && multv(g.a_rs, vw)) a_kvec = 1;
    else if (g.a_rs == 1 && multv(g.a_cs, vw) && multv(g.M, vw)) a_kvec = 0;
    else continue;
    int b_jvec;
    if (g.b_cs == 1 && multv(g.b_rs, vw) && multv(g.N, vw)) b_jvec = 1;
    else if (g.b_rs == 1 && multv(g.b_cs, vw)) b_jvec = 0;
    else continue;
    if (vw_out) *vw_out = vw;
    return 1 + 2 * a_kvec + b_jvec;
  }
  return 0;
}

static GemmArgs gemm_args(int M, int N, int K, const float* A, long long a_rs, long long a_cs,
                          const float* B, long long b_rs, long long b_cs, float* C,
                          long long ldc) {
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.a_rs = a_rs; g.a_cs = a_cs;
  g.B = B; g.b_rs = b_rs; g.b_cs = b_cs;
  g.C = C; g.ldc = ldc;
  g.nb1 = 1; g.ksplit = 1; g.kchunk = K;
  g.a_ifast = (a_rs == 1 && a_cs != 1);
  g.b_jfast = (b_cs == 1);
  return g;
}

static void gemm_launch(const GemmArgs& g0, int nb0, hipStream_t s) {
  // gridDim.z <= 65535: slice the outer batch when (pairs x measures x splits) is larger.
  const int per_b0 = g0.nb1 * g0.ksplit;
  const int max_b0 = per_b0 > 65535 ? 1 : 65535 / per_b0;
  for (int b = 0; b < nb0; b += max_b0) {
    const int nb = (nb0 - b) < max_b0 ? (nb0 - b) : max_b0;
    GemmArgs g = g0;
    g.A += (long long)b * g.a_b0;
    g.B += (long long)b * g.b_b0;
    g.C += (long long)b * g.c_b0;
    if (g.rowscale) g.rowscale += (long long)b * g.rs_b0;
    int vw = 4;
    const int fv = gemm_fast_variant(g, &vw);
    if (fv) {
      dim3 grid((g.N + FN - 1) / FN, (g.M + FM - 1) / FM, nb * per_b0);
      const bool ksc = g.bkscale != nullptr;   // only with B_JVEC (gemm_fast_variant)
      const bool deep = (long long)grid.x * grid.y * grid.z <= 2 * 256;   // at most two workgroups per CU
#define MMS_FAST(a, b, c)                                                                              \
  do {                                                                                                 \
    if (deep && vw == 4) hipLaunchKernelGGL((gemm32_fast_kernel<a, b, c, 32, 4>), grid, dim3(256), 0, s, g);  \
    else if (vw == 4) hipLaunchKernelGGL((gemm32_fast_kernel<a, b, c, 16, 4>), grid, dim3(256), 0, s, g);     \
    else if (deep) hipLaunchKernelGGL((gemm32_fast_kernel<a, b, c, 32, 2>), grid, dim3(256), 0, s, g);        \
    else hipLaunchKernelGGL((gemm32_fast_kernel<a, b, c, 16, 2>), grid, dim3(256), 0, s, g);                  \
  } while (0)
      switch (fv - 1) {
        case 0: MMS_FAST(false, false, false); break;
        case 1: if (ksc) MMS_FAST(false, true, true); else MMS_FAST(false, true, false); break;
        case 2: MMS_FAST(true, false, false); break;
        default: if (ksc) MMS_FAST(true, true, true); else MMS_FAST(true, true, false); break;
      }
#undef MMS_FAST
      continue;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, nb * per_b0);
    hipLaunchKernelGGL(gemm32_kernel, grid, dim3(256), 0, s, g);
  }
}

// Launch up to kGroupMax independent small products as one grid (gemm32_group_kernel).  Returns false --
// nothing launched -- when a problem needs the stride-generic kernel or an epilogue the group kernel
// does not carry, or when the products are big enough to deserve their own tuned launches.
static bool gemm_launch_group(const GemmArgs* gs, const int* nb0s, int n, hipStream_t s) {
  if (n < 2 || n > kGroupMax) return false;
  GemmGroup grp{};
  int vw = 4, total = 0;
  for (int i = 0; i < n; ++i) {
    const GemmArgs& g = gs[i];
    if (g.bkscale || g.rowscale || g.addend) return false;
    int v = 4;
    const int fv = gemm_fast_variant(g, &v);
    if (!fv) return false;
    vw = v < vw ? v : vw;
    grp.g[i] = g;
    grp.ak[i] = ((fv - 1) >> 1) & 1;
    grp.bj[i] = (fv - 1) & 1;
    grp.gx[i] = (g.N + FN - 1) / FN;
    grp.gy[i] = (g.M + FM - 1) / FM;
    const long long cnt = (long long)grp.gx[i] * grp.gy[i] * nb0s[i] * g.nb1 * g.ksplit;
    if (cnt > 1536) return false;                // a product this large keeps its own XCD-ordered launch
    grp.first[i] = total;
    total += (int)cnt;
  }
  if (total > 3072) return false;
  for (int i = n; i <= kGroupMax; ++i) grp.first[i] = total;
  grp.n = n;
  const bool deep = total <= 2 * 256;
  if (deep && vw == 4) hipLaunchKernelGGL((gemm32_group_kernel<32, 4>), dim3(total), dim3(256), 0, s, grp);
  else if (vw == 4) hipLaunchKernelGGL((gemm32_group_kernel<16, 4>), dim3(total), dim3(256), 0, s, grp);
  else if (deep) hipLaunchKernelGGL((gemm32_group_kernel<32, 2>), dim3(total), dim3(256), 0, s, grp);
  else hipLaunchKernelGGL((gemm32_group_kernel<16, 2>), dim3(total), dim3(256), 0, s, grp);
  return true;
}

// s + part[0*n + e] + part[1*n + e] + ... in s-ascending order (the reference accumulates over pairs / measures in
// that order).  All requests of a batch are in flight before the first add -- a load-add-load loop costs one
// memory round trip per slab; batches of 32 above eight slabs (a training batch of 50 pairs: two round trips
// instead of seven), of 8 below (dQ / dA over four measures: no wasted requests).
__device__ __forceinline__ float ordered_slab_sum(const float* __restrict__ part, long long n, long long e, int splits,
                                                  float s) {
  if (splits > 8) {
    for (int k0 = 0; k0 < splits; k0 += 32) {
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = part[(long long)min(k0 + u, splits - 1) * n + e];
#pragma unroll
      for (int u = 0; u < 32; ++u) s += (k0 + u < splits) ? v[u] : 0.f;
    }
    return s;
  }
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = part[(long long)min(u, splits - 1) * n + e];
#pragma unroll
  for (int u = 0; u < 8; ++u) s += (u < splits) ? v[u] : 0.f;
  return s;
}

// Several split-K reductions in one launch: problem p owns blocks [first[p], first[p+1]).
struct ReduceGroup {
  const float* part[kGroupMax];
  float* out[kGroupMax];
  long long n[kGroupMax];
  int splits[kGroupMax];
  int accumulate[kGroupMax];     // the sum starts from out[e] (bias.diff += ..., sim_cross_layer.cpp:301-304) instead of 0
  int first[kGroupMax + 1];
  int cnt;
};
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(ReduceGroup rg) {
  int p = 0;
#pragma unroll
  for (int i = 1; i < kGroupMax; ++i)
    if (i < rg.cnt && (int)blockIdx.x >= rg.first[i]) p = i;
  const float* __restrict__ part = rg.part[p];
  float* __restrict__ out = rg.out[p];
  const long long n = rg.n[p];
  const int splits = rg.splits[p];
  const long long stride = (long long)(rg.first[p + 1] - rg.first[p]) * 256;
  const bool accumulate = rg.accumulate[p] != 0;
  for (long long e = (long long)(blockIdx.x - rg.first[p]) * 256 + threadIdx.x; e < n; e += stride) {
    out[e] = ordered_slab_sum(part, n, e, splits, accumulate ? out[e] : 0.f);
  }
}

// out[e] (= or +=) sum_s part[s*n + e], s ascending.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part,
                                                            int splits, long long n,
                                                            float* __restrict__ out,
                                                            int accumulate) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
    const float s = ordered_slab_sum(part, n, e, splits, 0.f);
    out[e] = accumulate ? out[e] + s : s;
  }
}

// SimMatrix backward on the bf16 pipe: the split-K reduction of dW and the operand image of W^T for the dq product are two
// independent small launches in a row; here they are ONE -- workgroups [0, red_blocks) reduce, the rest split.
__global__ __launch_bounds__(256) void splitk_reduce_split_kernel(const float* __restrict__ part, int splits, long long n,
                                                                  float* __restrict__ out, int accumulate, int red_blocks,
                                                                  const Bx3SplitArgs sp) {
  if ((int)blockIdx.x < red_blocks) {
    const long long stride = (long long)red_blocks * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
      const float s = ordered_slab_sum(part, n, e, splits, 0.f);
      out[e] = accumulate ? out[e] + s : s;
    }
    return;
  }
  bx3_split_b_body(sp, ((int)blockIdx.x - red_blocks) * 256 + threadIdx.x, ((int)gridDim.x - red_blocks) * 256);
}

// SimMatrix backward: the split-K reduction of dW and the transpose of W (the dq product's k-major B operand) are
// two independent ~5-us launches in a row; here they are ONE -- workgroups [0, red_blocks) reduce, the rest
// transpose 32 x 32 tiles -- which takes a launch (1.6 us of floor + the shorter kernel) off a cfg 3 step.
__global__ __launch_bounds__(256) void splitk_reduce_transpose_kernel(const float* __restrict__ part, int splits,
                                                                      long long n, float* __restrict__ out,
                                                                      int accumulate, int red_blocks,
                                                                      const float* __restrict__ tin,
                                                                      float* __restrict__ tout, int rows, int cols) {
  if ((int)blockIdx.x < red_blocks) {
    const long long stride = (long long)red_blocks * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
      const float s = ordered_slab_sum(part, n, e, splits, 0.f);
      out[e] = accumulate ? out[e] + s : s;
    }
    return;
  }
  __shared__ float tile[32][33];
  const int tb = (int)blockIdx.x - red_blocks, tiles_x = (cols + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = (tb % tiles_x) * 32, r0 = (tb / tiles_x) * 32;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int rr = r0 + ty + 8 * u, cc = c0 + tx;
    if (rr < rows && cc < cols) tile[ty + 8 * u][tx] = tin[(long long)rr * cols + cc];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int oc = c0 + ty + 8 * u, orr = r0 + tx;          // tout[oc][orr] = tin[orr][oc]
    if (oc < cols && orr < rows) tout[(long long)oc * rows + orr] = tile[tx][ty + 8 * u];
  }
}

// out[r][c] = scale[r] * x[r][c]
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ scale,
                                                       float* __restrict__ out, long long rows,
                                                       int cols) {
  const long long n = rows * cols;
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride)
    out[e] = scale[e / cols] * x[e];
}

// out[r][c] = scale[r] * x[r][c]; out may BE x.
__global__ __launch_bounds__(256) void rowscale_inplace_ok_kernel(const float* x, const float* __restrict__ scale,
                                                                  float* out, long long rows, int cols) {
  const long long n = rows * cols;
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride)
    out[e] = scale[e / cols] * x[e];
}

// out[r][c] = scale[r] * x[r][c] for 16-byte-aligned rows (cols % 4 == 0); out may BE x (each thread
// reads the float4 it overwrites).  Streaming stores: the result is read next by another layer.
__global__ __launch_bounds__(256) void rowscale4_kernel(const float4* x, const float* __restrict__ scale,
                                                        float4* out, long long rows, int cols4) {
  const long long n = rows * cols4;
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
    const float sc = scale[e / cols4];
    const float4 v = x[e];
    stream_store(out + e, make_float4(sc * v.x, sc * v.y, sc * v.z, sc * v.w));
  }
}

// top[r] = dot(x[r], y[r]) (+ bias)  -- one wave per row, fixed butterfly.
// top index = r*top_stride ; bias is a single value (may be null).
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ top, long long rows,
                                                     int cols, long long top_stride) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * cols;
  const float* yr = y + r * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c] * yr[c];
  s = wave_sum(s);
  if (lane == 0) top[r * top_stride] = bias ? (*bias + s) : s;
}

// dbias[e] = dT[n][e] + dbias[e] for n ascending (sim_cross_layer.cpp:301-304: same order, bit-exact).
// The sum of one output is a dependent chain over n; what can be hidden is memory latency.  A workgroup
// owns 64 consecutive outputs: its four waves each fetch 16 of the next 64 rows (coalesced 256-byte
// segments) into LDS while wave 0 adds the previous 64 rows in order.  (One thread per output with
// eight loads per round trip took 66 us at the 1517-candidate test split; this takes ~12.)
// per_n == 1 (one scalar bias: SimCross bilinear at W1 = W2 = 1, one measure): the chain kernel above runs on ONE
// lane that fetches its own operands, a memory round trip per 64 terms (300 us at 16384 pairs).  Here the whole
// wave fetches -- 64 consecutive terms per load, four loads ahead -- and the running sum, wave-uniform, takes them in
// n order through v_readlane: the dependent add is all that is left (~3 ns per term).
__global__ __launch_bounds__(64) void dbias_scalar_kernel(const float* __restrict__ top_diff, int N,
                                                          float* __restrict__ dbias) {
  const int lane = threadIdx.x;
  float s = dbias[0];
  float nx[4];
  auto fetch = [&](int base) {
#pragma unroll
    for (int u = 0; u < 4; ++u) nx[u] = top_diff[min(base + 64 * u + lane, N - 1)];
  };
  fetch(0);
  for (int base = 0; base < N; base += 256) {
    float cur[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = nx[u];
    if (base + 256 < N) fetch(base + 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int left = N - (base + 64 * u);
      if (left >= 64) {
#pragma unroll
        for (int l = 0; l < 64; ++l) s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[u]), l)) + s;
      } else {
        for (int l = 0; l < left; ++l) s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[u]), l)) + s;
      }
    }
  }
  if (lane == 0) dbias[0] = s;
}
__global__ __launch_bounds__(256) void dbias_kernel(const float* __restrict__ top_diff, int N,
                                                    int per_n, float* __restrict__ dbias) {
  constexpr int CH = 64, RPW = CH / 4;
  __shared__ float buf[2][CH][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const bool ok = e < per_n;
  const int ec = ok ? e : per_n - 1;
  float s = dbias[ec];
  float r[RPW];
  auto fetch = [&](int c) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int n = c * CH + wave * RPW + u;
      r[u] = top_diff[(size_t)min(n, N - 1) * per_n + ec];
    }
  };
  const int nchunks = (N + CH - 1) / CH;
  fetch(0);
  for (int c = 0; c < nchunks; ++c) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) buf[c & 1][wave * RPW + u][lane] = r[u];
    __syncthreads();                             // chunk c complete in LDS; adds of chunk c-1 finished
    if (c + 1 < nchunks) fetch(c + 1);
    if (wave == 0) {
      const int cnt = min(CH, N - c * CH);
      for (int u = 0; u < cnt; ++u) s = buf[c & 1][u][lane] + s;
    }
  }
  if (wave == 0 && ok) dbias[e] = s;
}

// Few outputs (sentence-vector geometry: per_n = M): the work IS the N-long dependent add chain of each
// output, and rows of consecutive n share cache lines.  One wave, 64 loads in flight while the previous 64
// values are added; no LDS, no barrier in the chain's way.
__global__ __launch_bounds__(64) void dbias_chain_kernel(const float* __restrict__ top_diff, int N,
                                                         int per_n, float* __restrict__ dbias) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= per_n) return;
  float s = dbias[e];
  constexpr int B = 64;
  float cur[B], nxt[B];
  const int full = N / B;
  if (full > 0) {
#pragma unroll
    for (int u = 0; u < B; ++u) cur[u] = top_diff[(size_t)u * per_n + e];
  }
  for (int b = 0; b < full; ++b) {
    if (b + 1 < full) {
#pragma unroll
      for (int u = 0; u < B; ++u) nxt[u] = top_diff[(size_t)((b + 1) * B + u) * per_n + e];
    }
#pragma unroll
    for (int u = 0; u < B; ++u) s = cur[u] + s;
#pragma unroll
    for (int u = 0; u < B; ++u) cur[u] = nxt[u];
  }
  for (int n = full * B; n < N; ++n) s = top_diff[(size_t)n * per_n + e] + s;
  dbias[e] = s;
}

static unsigned ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// Split count for a product with a long K (the dW products: K = pairs).  Workgroups = tiles x batch x
// splits; the chip takes them 256 x (workgroups per CU) at a time, so the count should (a) reach ~3 per CU
// and (b) nearly FILL its last round: 25 tiles x 32 splits = 800 is 3.1 per CU -- a fourth round for 12 %
// of the CUs, 78 % efficient -- while 25 x 40 = 1000 fills 97.6 % of four rounds (cfg 3's dW product:
// 52 -> 44 us).  Splits of 8 or more come in multiples of 8 so that the XCD-aware order applies.
static int pick_ksplit(int Mt, int Nt, int K, int* kchunk, int batch = 1) {
  const long long tiles = (long long)((Mt + FM - 1) / FM) * ((Nt + FN - 1) / FN) * (batch > 0 ? batch : 1);
  const long long maxs = (K + 63) / 64;           // at least 64 of K (two deep k-tiles) per split
  long long best = 1;
  double best_score = -1.0;
  for (long long sp = 1; sp <= maxs && sp <= 256; ++sp) {
    if (sp >= 8 && (sp & 7)) continue;
    const long long x = tiles * sp;
    if (x > 1280 && sp > 1) break;
    const double rounds = (double)((x + 255) / 256);
    const double eff = (double)x / 256.0 / rounds;              // how full the last round is
    const double fill = x >= 768 ? 1.0 : (double)x / 768.0;     // ~3 workgroups per CU hide latency
    const double score = eff * fill;
    if (score > best_score + 1e-9) { best_score = score; best = sp; }
  }
  int chunk = (int)((K + best - 1) / best);
  chunk = (chunk + 31) / 32 * 32;               // a multiple of either k-tile depth (16, 32)
  *kchunk = chunk;
  return (K + chunk - 1) / chunk;
}

// ------------------------------ workspace layout ----------------------------
struct BilinearWs {
  size_t u_off, v_off, part_off, mpart_off, mpart2_off, total;
  int ksplit, kchunk;
};
static bool pair_bwd_fits(int N, int W1, int W2, int D, int M) {      // = pair_bwd_eligible (defined with the kernel)
  return W1 <= 48 && W2 <= 48 && D <= 64 && W1 * W2 > 1 && N <= 256 && (long long)N * M <= 65535;
}
// The sentence-vector geometry with ONE measure (W1 = W2 = 1, M = 1: BASELINE cfg 3 written as a SimCross layer) IS
// SimMatrix's arithmetic -- T_n = q_n^T W a_n (+ bias), dW = sum_n dT_n q_n a_n^T, dq_n = dT_n W a_n,
// da_n = dT_n W^T q_n -- so it takes SimMatrix's panel-GEMM launches (row dot and row scale as epilogues) instead
// of the generic GEMM + rowdot / rowscale launches: 45 + 136 us -> the SimMatrix figures (recomputing Q.W).
size_t simmatrix_workspace_bytes(int N, int K1, int K2);
int simmatrix_forward(int N, int K1, int K2, const float* q, const float* a, const float* W, float* top, float* qw,
                      hipStream_t s, const float* rd_bias, void* ws = nullptr, size_t ws_bytes = 0);
int simmatrix_backward(int N, int K1, int K2, const float* q, const float* a, const float* W,
                       const float* top_diff, int ppd, int pd0, int pd1, float* dq, float* da,
                       float* dW, const float* qw, void* ws, size_t ws_bytes, hipStream_t s);
static bool bilinear_as_simmatrix(int W1, int W2, int M) { return W1 == 1 && W2 == 1 && M == 1; }

static BilinearWs bilinear_ws(int N, int W1, int W2, int D, int M) {
  BilinearWs w{};
  const size_t u = (size_t)M * N * W1 * D, v = (size_t)M * N * W2 * D;
  w.ksplit = pick_ksplit(D, D, N * W1, &w.kchunk, M);
  w.u_off = 0;
  w.v_off = round_up(u * sizeof(float), 256);
  w.part_off = w.v_off + round_up(v * sizeof(float), 256);
  // split-K slabs of dW -- or, when the fused per-pair backward runs, one D x D partial per (pair, measure)
  const size_t slabs = pair_bwd_fits(N, W1, W2, D, M) && N > w.ksplit ? (size_t)N : (size_t)w.ksplit;
  w.mpart_off = w.part_off + round_up(slabs * M * D * D * sizeof(float), 256);
  // per-measure partial products of dQ and of dA (M > 1 only): [M][N*W1][D], [M][N*W2][D]
  w.mpart2_off = w.mpart_off + (M > 1 ? round_up(u * sizeof(float), 256) : 0);
  w.total = w.mpart2_off + (M > 1 ? round_up(v * sizeof(float), 256) : 0);
  if (bilinear_as_simmatrix(W1, W2, M)) {            // [Q.W, N x D][SimMatrix's own workspace]
    const size_t sm = round_up((size_t)N * D * sizeof(float), 256) + simmatrix_workspace_bytes(N, D, D);
    if (sm > w.total) w.total = sm;
  }
  return w;
}
size_t bilinear_workspace_bytes(int N, int W1, int W2, int D, int M) {
  return bilinear_ws(N, W1, W2, D, M).total;
}

// LDS writes of this wave visible to its own later reads (no other wave touches the slice)
__device__ __forceinline__ void wave_lds_sync_local() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- fused forward for word grids (the driver's 40 x 40 x Dw geometry) ---------------------------
// One workgroup per pair n: T[n,m] = (Q_n W_m) A_n^T + bias_m for every measure m, with Q_n W_m kept in
// LDS -- the (M, N*W1, D) intermediate of the two-GEMM formulation (written to and read back from HBM:
// 2 x 48 MB at the 1517-candidate test split) never exists, and the forward is ONE launch.
//   * q_n and a_n are staged once as zero-padded images, row stride 68 floats (rows 4 banks apart: the
//     16 rows x 4 k of an MFMA operand read hit 64 distinct banks);
//   * work items (measure m, 16-row tile of Q) are dealt to the four waves.  An item runs
//     stage 1  tmp (16 x D)  = Q rows x W_m : ceil(D/16) accumulators, B operand W_m[k][j] read straight
//              from global memory (M*D*D floats: L1/L2-resident), one 4-byte load per MFMA;
//     stage 2  T   (16 x W2) = tmp x A_n^T  : tmp goes through the wave's own LDS slice to become an A
//              operand (k-major per lane), B operand from the a image;
//   * v_mfma_f32_16x16x4_f32: W = 40 fills 40/48 of the tiles (32x32 tiles: 40/64).
// Eligible for W1, W2 <= 48 and D <= 64; anything else takes the two batched GEMMs below.
constexpr int PF_LS = 68, PF_ROWS = 48, PF_TD = 4;
typedef float v4f __attribute__((ext_vector_type(4)));

// Embed fused into the staging loads (the mode network_v4 scores with): with g.iq != nullptr,
// q and a are both the embedding TABLE (K x D) and row r of pair n is table row g.iq[n*W1 + r] (g.ia likewise):
// the (N, W, D) blobs the Embed layers would write and SimCross read back never exist.
struct PairGather {
  const float* iq;
  const float* ia;
  int K;
  const float* bias;     // the Embed layer's bias (D floats) or nullptr: row value = bias[d] + table[id][d]
};
__device__ __forceinline__ int pair_gather_id(float v, int K) {   // as mms_embed_forward_f32 clamps
  const int i = (int)v;
  return i < 0 ? 0 : (i >= K ? K - 1 : i);
}

template <int KS>                                  // k steps of 4: 13 covers D <= 52 (the driver's 50), 16 D <= 64
__global__ __launch_bounds__(256) void bilinear_pair_fwd_kernel(
    int N, int W1, int W2, int D, int M, const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ top,
    PairGather g = PairGather{nullptr, nullptr, 0, nullptr}) {
  __shared__ float qs[PF_ROWS * PF_LS];
  __shared__ float as[PF_ROWS * PF_LS];
  __shared__ float ts[4][16 * PF_LS];
  const int n = blockIdx.x;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const float* qn = q + (size_t)n * W1 * D;
  const float* an = a + (size_t)n * W2 * D;
  const int li = lane & 15, lk = lane >> 4;
  const int ti_n = (W1 + 15) / 16;
  // Items are dealt in contiguous runs (m-major), so a wave mostly stays on one measure and keeps that
  // measure's B operands -- W_m[k][j] for its lane, all k steps -- in registers: they are fetched once,
  // all loads in flight together (a load per MFMA inside the k loop costs a memory round trip per step).
  const int items = M * ti_n, per = (items + 3) / 4;
  float wf[KS][PF_TD];
  auto fetch_w = [&](int m) {
    const float* Wm = W + (size_t)m * D * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + lk;
#pragma unroll
      for (int d = 0; d < PF_TD; ++d) wf[ks][d] = Wm[(size_t)min(k, D - 1) * D + min(16 * d + li, D - 1)];
    }
  };
  auto mask_w = [&]() {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int d = 0; d < PF_TD; ++d)
        if (!(4 * ks + lk < D && 16 * d + li < D)) wf[ks][d] = 0.f;
  };
  int have_m = -1;
  if (wave * per < items) {                        // the first measure's operands: in flight behind the staging
    have_m = (wave * per) / ti_n;
    fetch_w(have_m);
  }
  // zero-padded images: every load issued (clamped, unconditional) before the first LDS write
  constexpr int NE = (PF_ROWS * PF_LS + 255) / 256;
  float vq[NE], va[NE];
  if (g.iq) {                                      // ids first (all in flight), then the table rows
    float fq[NE], fa[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int r = (256 * u + t) / PF_LS;
      fq[u] = g.iq[(size_t)n * W1 + min(r, W1 - 1)];
      fa[u] = g.ia[(size_t)n * W2 + min(r, W2 - 1)];
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = 256 * u + t;
      const int c = e - (e / PF_LS) * PF_LS;
      vq[u] = q[(size_t)pair_gather_id(fq[u], g.K) * D + min(c, D - 1)];
      va[u] = a[(size_t)pair_gather_id(fa[u], g.K) * D + min(c, D - 1)];
      if (g.bias) { const float bv = g.bias[min(c, D - 1)]; vq[u] = bv + vq[u]; va[u] = bv + va[u]; }
    }
  } else {
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = 256 * u + t;
      const int r = e / PF_LS, c = e - r * PF_LS;
      vq[u] = qn[(size_t)min(r, W1 - 1) * D + min(c, D - 1)];
      va[u] = an[(size_t)min(r, W2 - 1) * D + min(c, D - 1)];
    }
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = 256 * u + t;
    const int r = e / PF_LS, c = e - r * PF_LS;
    if (e < PF_ROWS * PF_LS) {
      qs[e] = (r < W1 && c < D) ? vq[u] : 0.f;
      as[e] = (r < W2 && c < D) ? va[u] : 0.f;
    }
  }
  __syncthreads();
  if (have_m >= 0) mask_w();
  float* tw = ts[wave];
  for (int item = wave * per; item < min(items, (wave + 1) * per); ++item) {
    const int m = item / ti_n, ti = item - m * ti_n;
    if (m != have_m) {
      fetch_w(m);
      mask_w();
      have_m = m;
    }
    // this item's bias values: requested now, used after the two stages (a load in the epilogue would
    // expose a memory round trip per item)
    const float* bm = bias ? bias + (size_t)m * W1 * W2 : nullptr;
    float bv[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(16 * ti + 4 * lk + r, W1 - 1), col = min(16 * c + li, W2 - 1);
        bv[c][r] = bm ? bm[row * W2 + col] : 0.f;
      }
    // stage 1: tmp[16 x D] = Q[16 rows of tile ti] . W_m
    v4f acc1[PF_TD];
#pragma unroll
    for (int d = 0; d < PF_TD; ++d) acc1[d] = (v4f){0.f, 0.f, 0.f, 0.f};
    // padded tiles are computed too (their operands are zero): no branch between MFMAs
    float a1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a1[ks] = qs[(16 * ti + li) * PF_LS + 4 * ks + lk];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int d = 0; d < PF_TD; ++d)
        acc1[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], wf[ks][d], acc1[d], 0, 0, 0);
    }
    // C layout: col = lane & 15, row = 4 * (lane >> 4) + reg  ->  the wave's LDS slice, row-major
#pragma unroll
    for (int d = 0; d < PF_TD; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) tw[(4 * lk + r) * PF_LS + 16 * d + li] = acc1[d][r];
    wave_lds_sync_local();
    // stage 2: T[16 x W2] = tmp . A_n^T   (B[k][j] = a[j][k])
    v4f acc2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc2[c] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + lk;
      const float av = tw[li * PF_LS + k];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        acc2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, as[(16 * c + li) * PF_LS + k], acc2[c], 0, 0, 0);
    }
    wave_lds_sync_local();                         // tw is rewritten by this wave's next item
    float* tn = top + ((size_t)n * M + m) * W1 * W2;
#pragma unroll
    for (int c = 0; c < 3; ++c)                    // in registers before the first store (else: vmcnt(0) behind each)
      asm volatile("" : "+v"(bv[c][0]), "+v"(bv[c][1]), "+v"(bv[c][2]), "+v"(bv[c][3]));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int col = 16 * c + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * lk + r;
        if (row < W1 && col < W2) {
          float v = acc2[c][r];
          if (bm) v = bv[c][r] + v;                // the addend form of the GEMM epilogue (:156-158)
          tn[row * W2 + col] = v;
        }
      }
    }
  }
}


// ---- fused backward for word grids at TRAINING batch sizes (the driver's 50 x 40 x 40 x Dw) -----------------
// At a batch of 50 pairs every product of the bilinear backward is a 4-8 us launch at the latency floor (six
// launches, 27 us).  Here ONE launch runs all five products of a (pair, measure): a workgroup stages q_n, a_n,
// dT_nm and W_m as zero-padded LDS images (row stride 68, as in bilinear_pair_fwd_kernel), then
//   phase 1   U = dT A (W1 x D), V = dT^T Q (W2 x D)              -> LDS
//   phase 2   dQ_nm = U W_m^T, dA_nm = V W_m, dW_nm = Q^T U        -> per-(n, m) partials in the workspace
// on v_mfma_f32_16x16x4_f32, the 16 x 16 output tiles of a phase dealt round-robin to the four waves.  The sums
// the reference takes in place -- dQ_n over m (sim_cross_layer.cpp:291-294), dA_n over m (:296-299), dW_m over n
// (:286-289) -- are taken afterwards by ONE grouped reduction launch in the same ascending orders.
constexpr int FB_LS = 68, FB_W = 48, FB_D = 64;
// KSW / KSD: k-steps of 4 over a word axis / the embedding axis, fixed at compile time so that a tile's operand
// reads are ALL issued before its MFMAs (a rolled read-read-MFMA loop paid an LDS round trip per k-step: 17.6 us);
// the images are zero beyond W and D, so steps past the real extent add exact zeros.
template <int KSW, int KSD>
__global__ __launch_bounds__(512) void bilinear_pair_bwd_kernel(
    int N, int W1, int W2, int D, int M, const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ W, const float* __restrict__ top_diff, float* __restrict__ mq,
    float* __restrict__ ma, float* __restrict__ wpart) {
  __shared__ float qs[FB_W * FB_LS], as[FB_W * FB_LS], ts[FB_W * FB_LS], ws[FB_D * FB_LS];
  __shared__ float us[FB_W * FB_LS], vs[FB_W * FB_LS];
  const int n = blockIdx.x / M, m = blockIdx.x - n * M;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, g = lane >> 4;
  const float* qn = q + (size_t)n * W1 * D;
  const float* an = a + (size_t)n * W2 * D;
  const float* Wm = W + (size_t)m * D * D;
  const float* dT = top_diff + ((size_t)n * M + m) * W1 * W2;
  // zero-padded images: every load issued (clamped, unconditional) before the first LDS write
  constexpr int NT = 512, NWV = NT / 64;
  constexpr int NE = (FB_W * FB_LS + NT - 1) / NT, NEW = (FB_D * FB_LS + NT - 1) / NT;
  {
    float vq[NE], va[NE], vt[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
      vq[u] = qn[(size_t)min(row, W1 - 1) * D + min(c, D - 1)];
      va[u] = an[(size_t)min(row, W2 - 1) * D + min(c, D - 1)];
      vt[u] = dT[(size_t)min(row, W1 - 1) * W2 + min(c, W2 - 1)];
    }
    // W_m's image is requested HERE, with the other three: behind the first LDS writes it was a second, exposed
    // memory round trip in a workgroup whose whole life is ~10 us
    float vw[NEW];
#pragma unroll
    for (int u = 0; u < NEW; ++u) {
      const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
      vw[u] = Wm[(size_t)min(row, D - 1) * D + min(c, D - 1)];
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
      if (e < FB_W * FB_LS) {
        qs[e] = (row < W1 && c < D) ? vq[u] : 0.f;
        as[e] = (row < W2 && c < D) ? va[u] : 0.f;
        ts[e] = (row < W1 && c < W2) ? vt[u] : 0.f;
        us[e] = 0.f;                                   // rows / columns no tile writes must read as zero
        vs[e] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < NEW; ++u) {
      const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
      if (e < FB_D * FB_LS) ws[e] = (row < D && c < D) ? vw[u] : 0.f;
    }
  }
  __syncthreads();
  const int tw1 = (W1 + 15) >> 4, tw2 = (W2 + 15) >> 4, td = (D + 15) >> 4;
  // one 16 x 16 tile: C(i0 + 4g + j, j0 + r) = sum_k A(i0 + r', k) B(k, j0 + r); A / B given as (base, row stride,
  // k stride): element (x, k) of an operand lives at base[x * xs + k * ks]
  auto tile = [&](auto nks_tag, const float* A, int axs, int aks, const float* B, int bxs, int bks, int i0, int j0) {
    constexpr int NKS = decltype(nks_tag)::value;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A + (i0 + r) * axs + g * aks;
    const float* bp = B + (j0 + r) * bxs + g * bks;
    float av[NKS], bv[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { av[ks] = ap[4 * ks * aks]; bv[ks] = bp[4 * ks * bks]; }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
    return acc;
  };
  const std::integral_constant<int, KSW> kw{};
  const std::integral_constant<int, KSD> kd{};
  auto put_lds = [&](float* dst, int i0, int j0, const v4f& acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(i0 + 4 * g + j) * FB_LS + j0 + r] = acc[j];
  };
  // phase 1: U (tw1 x td tiles) then V (tw2 x td tiles)
  const int nU = tw1 * td, nV = tw2 * td;
  for (int it = wave; it < nU + nV; it += NWV) {
    if (it < nU) {
      const int ti = it / td, tj = it - ti * td;
      // U[j][d] = sum_k dT[j][k] A[k][d]:  A-operand (row j, k) = ts[j*LS + k];  B-operand (k, col d) = as[k*LS + d]
      put_lds(us, 16 * ti, 16 * tj, tile(kw, ts, FB_LS, 1, as, 1, FB_LS, 16 * ti, 16 * tj));
    } else {
      const int e = it - nU, ti = e / td, tj = e - ti * td;
      // V[k][d] = sum_j dT[j][k] Q[j][d]:  A-operand (row k, kk = j) = ts[j*LS + k];  B-operand (j, col d) = qs[j*LS + d]
      put_lds(vs, 16 * ti, 16 * tj, tile(kw, ts, 1, FB_LS, qs, 1, FB_LS, 16 * ti, 16 * tj));
    }
  }
  __syncthreads();
  // phase 2
  const int nQ = tw1 * td, nA = tw2 * td, nW = td * td;
  float* mqn = mq + ((size_t)m * N + n) * W1 * D;       // [M][N*W1][D]
  float* man = ma + ((size_t)m * N + n) * W2 * D;       // [M][N*W2][D]
  float* wpn = wpart + ((size_t)n * M + m) * D * D;     // [N][M][D][D]
  auto put_global = [&](float* dst, int ld, int rows, int cols, int i0, int j0, const v4f& acc) {
    const int col = j0 + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = i0 + 4 * g + j;
      if (row < rows && col < cols) dst[(size_t)row * ld + col] = acc[j];
    }
  };
  for (int it = wave; it < nQ + nA + nW; it += NWV) {
    if (it < nQ) {
      const int ti = it / td, tj = it - ti * td;
      // dQ[j][d'] = sum_d U[j][d] W[d'][d]:  A (row j, k = d) = us[j*LS + d];  B (k = d, col d') = ws[d'*LS + d]
      put_global(mqn, D, W1, D, 16 * ti, 16 * tj, tile(kd, us, FB_LS, 1, ws, FB_LS, 1, 16 * ti, 16 * tj));
    } else if (it < nQ + nA) {
      const int e = it - nQ, ti = e / td, tj = e - ti * td;
      // dA[k][d'] = sum_d V[k][d] W[d][d']:  A (row k, kk = d) = vs[k*LS + d];  B (d, col d') = ws[d*LS + d']
      put_global(man, D, W2, D, 16 * ti, 16 * tj, tile(kd, vs, FB_LS, 1, ws, 1, FB_LS, 16 * ti, 16 * tj));
    } else {
      const int e = it - nQ - nA, ti = e / td, tj = e - ti * td;
      // dW[d][d'] = sum_j Q[j][d] U[j][d']:  A (row d, k = j) = qs[j*LS + d];  B (j, col d') = us[j*LS + d']
      put_global(wpn, D, D, D, 16 * ti, 16 * tj, tile(kw, qs, 1, FB_LS, us, 1, FB_LS, 16 * ti, 16 * tj));
    }
  }
}

// The forward twin of bilinear_pair_bwd_kernel for training batches: a workgroup per (pair, measure) stages q_n,
// a_n and W_m, forms tmp = Q_n W_m in LDS and T_nm = tmp A_n^T (+ bias_m) straight to `top` -- one launch instead
// of two batched GEMMs with a (M, N*W1, D) intermediate in HBM.  (bilinear_pair_fwd_kernel, one workgroup per
// PAIR with W_m operands held in registers, stays the choice for evaluation batches of hundreds of pairs.)
template <int KSD>
__global__ __launch_bounds__(512) void bilinear_pairm_fwd_kernel(
    int N, int W1, int W2, int D, int M, const float* __restrict__ q, const float* __restrict__ a,
    const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ top,
    PairGather gth = PairGather{nullptr, nullptr, 0, nullptr}) {
  __shared__ float qs[FB_W * FB_LS], as[FB_W * FB_LS], ws[FB_D * FB_LS], ps[FB_W * FB_LS];
  constexpr int NT = 512, NWV = NT / 64;
  const int n = blockIdx.x / M, m = blockIdx.x - n * M;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, g = lane >> 4;
  const float* qn = q + (size_t)n * W1 * D;
  const float* an = a + (size_t)n * W2 * D;
  const float* Wm = W + (size_t)m * D * D;
  constexpr int NE = (FB_W * FB_LS + NT - 1) / NT, NEW = (FB_D * FB_LS + NT - 1) / NT;
  {
    float vq[NE], va[NE], vw[NEW];
    if (gth.iq) {                                  // Embed fused in: ids first, then the table rows
      float fq[NE], fa[NE];
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const int row = (NT * u + t) / FB_LS;
        fq[u] = gth.iq[(size_t)n * W1 + min(row, W1 - 1)];
        fa[u] = gth.ia[(size_t)n * W2 + min(row, W2 - 1)];
      }
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const int e = NT * u + t, c = e - (e / FB_LS) * FB_LS;
        vq[u] = q[(size_t)pair_gather_id(fq[u], gth.K) * D + min(c, D - 1)];
        va[u] = a[(size_t)pair_gather_id(fa[u], gth.K) * D + min(c, D - 1)];
        if (gth.bias) { const float bv = gth.bias[min(c, D - 1)]; vq[u] = bv + vq[u]; va[u] = bv + va[u]; }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
        vq[u] = qn[(size_t)min(row, W1 - 1) * D + min(c, D - 1)];
        va[u] = an[(size_t)min(row, W2 - 1) * D + min(c, D - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < NEW; ++u) {
      const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
      vw[u] = Wm[(size_t)min(row, D - 1) * D + min(c, D - 1)];
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
      if (e < FB_W * FB_LS) {
        qs[e] = (row < W1 && c < D) ? vq[u] : 0.f;
        as[e] = (row < W2 && c < D) ? va[u] : 0.f;
        ps[e] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < NEW; ++u) {
      const int e = NT * u + t, row = e / FB_LS, c = e - row * FB_LS;
      if (e < FB_D * FB_LS) ws[e] = (row < D && c < D) ? vw[u] : 0.f;
    }
  }
  __syncthreads();
  const int tw1 = (W1 + 15) >> 4, tw2 = (W2 + 15) >> 4, td = (D + 15) >> 4;
  auto tile = [&](const float* A, int axs, int aks, const float* B, int bxs, int bks, int i0, int j0) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A + (i0 + r) * axs + g * aks;
    const float* bp = B + (j0 + r) * bxs + g * bks;
    float av[KSD], bv[KSD];
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) { av[ks] = ap[4 * ks * aks]; bv[ks] = bp[4 * ks * bks]; }
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
    return acc;
  };
  // tmp[j][d'] = sum_d Q[j][d] W[d][d']:  A (row j, k = d) = qs[j*LS + d];  B (d, col d') = ws[d*LS + d']
  for (int it = wave; it < tw1 * td; it += NWV) {
    const int ti = it / td, tj = it - ti * td;
    const v4f acc = tile(qs, FB_LS, 1, ws, 1, FB_LS, 16 * ti, 16 * tj);
#pragma unroll
    for (int j = 0; j < 4; ++j) ps[(16 * ti + 4 * g + j) * FB_LS + 16 * tj + r] = acc[j];
  }
  __syncthreads();
  // T[j][k] = sum_d' tmp[j][d'] A[k][d'] (+ bias[j][k]):  A (row j, k = d') = ps[j*LS + d'];  B (d', col k) = as[k*LS + d']
  float* tn = top + ((size_t)n * M + m) * W1 * W2;
  const float* bm = bias ? bias + (size_t)m * W1 * W2 : nullptr;
  for (int it = wave; it < tw1 * tw2; it += NWV) {
    const int ti = it / tw2, tj = it - ti * tw2;
    const int col = 16 * tj + r;
    float bvv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bvv[j] = bm ? bm[min(16 * ti + 4 * g + j, W1 - 1) * W2 + min(col, W2 - 1)] : 0.f;
    const v4f acc = tile(ps, FB_LS, 1, as, FB_LS, 1, 16 * ti, 16 * tj);
    asm volatile("" : "+v"(bvv[0]), "+v"(bvv[1]), "+v"(bvv[2]), "+v"(bvv[3]));   // in registers before the first store (else: vmcnt(0) behind it)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 16 * ti + 4 * g + j;
      if (row < W1 && col < W2) tn[row * W2 + col] = bm ? bvv[j] + acc[j] : acc[j];   // the addend form (:156-158)
    }
  }
}
static bool pair_bwd_eligible(int N, int W1, int W2, int D, int M) {
  return W1 <= FB_W && W2 <= FB_W && D <= FB_D && W1 * W2 > 1 && N <= 256 && (long long)N * M <= 65535;
}

// top = SimCross_bilinear(Embed(index_q), Embed(index_a)) in ONE launch, for the word-grid geometries the two
// fused forward kernels cover (W1, W2 <= 48, D <= 64): embed_layer.cpp:135-152 (embed_bias: the Embed layers' bias blob or null) followed by
// sim_cross_layer.cpp:140-161, the gather done by the staging loads.  Same kernels, same operand values: the
// bits of mms_embed_forward_f32 x2 followed by mms_simcross_forward_f32.  Other geometries: MMS_ERR_UNSUPPORTED.
int embed_bilinear_forward(int N, int W1, int W2, int D, int M, int K, const float* index_q,
                           const float* index_a, const float* table, const float* embed_bias, const float* W,
                           const float* bias, float* top, hipStream_t s) {
  if (N == 0) return MMS_OK;
  const PairGather g{index_q, index_a, K, embed_bias};
  if (W1 <= PF_ROWS && W2 <= PF_ROWS && D <= 16 * PF_TD && W1 * W2 > 1 && N >= 512) {
    if (D <= 52)
      hipLaunchKernelGGL(bilinear_pair_fwd_kernel<13>, dim3(N), dim3(256), 0, s, N, W1, W2, D, M, table, table, W,
                         bias, top, g);
    else
      hipLaunchKernelGGL(bilinear_pair_fwd_kernel<16>, dim3(N), dim3(256), 0, s, N, W1, W2, D, M, table, table, W,
                         bias, top, g);
    return launch_status();
  }
  if (pair_bwd_eligible(N, W1, W2, D, M)) {
    if (D <= 52)
      hipLaunchKernelGGL((bilinear_pairm_fwd_kernel<13>), dim3(N * M), dim3(512), 0, s, N, W1, W2, D, M, table, table, W, bias, top, g);
    else
      hipLaunchKernelGGL((bilinear_pairm_fwd_kernel<16>), dim3(N * M), dim3(512), 0, s, N, W1, W2, D, M, table, table, W, bias, top, g);
    return launch_status();
  }
  return MMS_ERR_UNSUPPORTED;
}

int bilinear_forward(int N, int W1, int W2, int D, int M, const float* q, const float* a,
                     const float* W, const float* bias, float* top, void* ws, size_t ws_bytes,
                     hipStream_t s) {
  const BilinearWs lay = bilinear_ws(N, W1, W2, D, M);
  if (!ws || ws_bytes < lay.total) return MMS_ERR_WORKSPACE;
  if (bilinear_as_simmatrix(W1, W2, M))
  {
    const size_t off = round_up((size_t)N * D * sizeof(float), 256);      // [Q.W][SimMatrix's own workspace]
    return simmatrix_forward(N, D, D, q, a, W, top, static_cast<float*>(ws), s, bias, static_cast<char*>(ws) + off,
                             ws_bytes - off);
  }
  // large batches only (evaluation: the 1517 TREC-QA test candidates, 89 -> 59 us): at the training batch of
  // 50 pairs both forms sit at the launch floor and the two small GEMMs are marginally quicker
  if (W1 <= PF_ROWS && W2 <= PF_ROWS && D <= 16 * PF_TD && W1 * W2 > 1 && N >= 512) {
    if (D <= 52)
      hipLaunchKernelGGL(bilinear_pair_fwd_kernel<13>, dim3(N), dim3(256), 0, s, N, W1, W2, D, M, q, a, W,
                         bias, top);
    else
      hipLaunchKernelGGL(bilinear_pair_fwd_kernel<16>, dim3(N), dim3(256), 0, s, N, W1, W2, D, M, q, a, W,
                         bias, top);
    return launch_status();
  }
  if (pair_bwd_eligible(N, W1, W2, D, M)) {          // training batches: one launch, one workgroup per (pair, measure)
    if (D <= 52)
      hipLaunchKernelGGL((bilinear_pairm_fwd_kernel<13>), dim3(N * M), dim3(512), 0, s, N, W1, W2, D, M, q, a, W, bias, top);
    else
      hipLaunchKernelGGL((bilinear_pairm_fwd_kernel<16>), dim3(N * M), dim3(512), 0, s, N, W1, W2, D, M, q, a, W, bias, top);
    return launch_status();
  }
  float* tmp = reinterpret_cast<float*>(static_cast<char*>(ws) + lay.u_off);
  const long long R = (long long)N * W1;
  // tmp[m] = Q_all W_m   (:148-149, batched over all pairs)
  {
    GemmArgs g = gemm_args((int)R, D, D, q, D, 1, W, D, 1, tmp, D);
    g.nb1 = M; g.b_b1 = (long long)D * D; g.c_b1 = R * D;
    gemm_launch(g, 1, s);
  }
  if (W1 == 1 && W2 == 1) {
    // T[n,m] = tmp[m][n] . a[n] (+ bias[m])   (:151-158)
    for (int m = 0; m < M; ++m)
      hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s,
                         tmp + (size_t)m * R * D, a, bias ? bias + m : nullptr, top + m,
                         (long long)N, D, (long long)M);
  } else {
    // T[n,m] = tmp[m][n] A_n^T (+ bias_m), batched over (n, m)
    GemmArgs g = gemm_args(W1, W2, D, tmp, D, 1, a, 1, D, top, W2);
    g.nb1 = M;
    g.a_b0 = (long long)W1 * D; g.a_b1 = R * D;
    g.b_b0 = (long long)W2 * D; g.b_b1 = 0;
    g.c_b0 = (long long)M * W1 * W2; g.c_b1 = (long long)W1 * W2;
    g.addend = bias; g.ad_b1 = (long long)W1 * W2;
    gemm_launch(g, N, s);
  }
  return launch_status();
}

int bilinear_backward(int N, int W1, int W2, int D, int M, const float* q, const float* a,
                      const float* W, int bias_term, const float* top_diff, float* dq, float* da,
                      float* dW, float* dbias, void* ws, size_t ws_bytes, hipStream_t s) {
  const BilinearWs lay = bilinear_ws(N, W1, W2, D, M);
  if (!ws || ws_bytes < lay.total) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  float* U = reinterpret_cast<float*>(base + lay.u_off);     // [M][N*W1][D]
  float* V = reinterpret_cast<float*>(base + lay.v_off);     // [M][N*W2][D]
  float* part = reinterpret_cast<float*>(base + lay.part_off);
  const long long R1 = (long long)N * W1, R2 = (long long)N * W2;

  if (pair_bwd_eligible(N, W1, W2, D, M)) {
    // one launch for the five products of every (pair, measure), one grouped launch for the three sums
    float* mq = M > 1 ? reinterpret_cast<float*>(base + lay.mpart_off) : dq;
    float* ma = M > 1 ? reinterpret_cast<float*>(base + lay.mpart2_off) : da;
    if (W1 <= 40 && W2 <= 40 && D <= 52)           // the driver's geometry: 10 / 13 k-steps
      hipLaunchKernelGGL((bilinear_pair_bwd_kernel<10, 13>), dim3(N * M), dim3(512), 0, s, N, W1, W2, D, M, q, a, W,
                         top_diff, mq, ma, part);
    else
      hipLaunchKernelGGL((bilinear_pair_bwd_kernel<12, 16>), dim3(N * M), dim3(512), 0, s, N, W1, W2, D, M, q, a, W,
                         top_diff, mq, ma, part);
    ReduceGroup rg{};
    const long long nWt = (long long)M * D * D;
    int first = 0, cnt = 0;
    auto add = [&](const float* p, float* o, long long n, int splits) {
      rg.part[cnt] = p; rg.out[cnt] = o; rg.n[cnt] = n; rg.splits[cnt] = splits; rg.first[cnt] = first;
      first += (int)ew_blocks(n);
      ++cnt;
    };
    if (M > 1) { add(mq, dq, R1 * D, M); add(ma, da, R2 * D, M); }
    add(part, dW, nWt, N);                           // W.diff is overwritten (:256), pairs summed ascending
    if (bias_term) {
      // bias.diff += dT_n, n ascending (:301-304): top_diff IS the [pair][M*W1*W2] stack of addends
      add(top_diff, dbias, (long long)M * W1 * W2, N);
      rg.accumulate[cnt - 1] = 1;
    }
    for (int i = cnt; i <= kGroupMax; ++i) rg.first[i] = first;
    rg.cnt = cnt;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(first), dim3(256), 0, s, rg);
    return launch_status();
  }
  // dbias first: it depends on nothing the products write
  if (bias_term) {
    const int per_n = M * W1 * W2;
    if (per_n == 1)
      hipLaunchKernelGGL(dbias_scalar_kernel, dim3(1), dim3(64), 0, s, top_diff, N, dbias);
    else if (per_n <= 256)
      hipLaunchKernelGGL(dbias_chain_kernel, dim3((per_n + 63) / 64), dim3(64), 0, s, top_diff, N, per_n,
                         dbias);
    else
      hipLaunchKernelGGL(dbias_kernel, dim3((per_n + 63) / 64), dim3(256), 0, s, top_diff, N, per_n,
                         dbias);
  }
  if (bilinear_as_simmatrix(W1, W2, M)) {
    // W.diff is OVERWRITTEN by SimCross (:256) where SimMatrix accumulates: start from zero (0 + x = x exactly)
    if (hipMemsetAsync(dW, 0, sizeof(float) * (size_t)D * D, s) != hipSuccess) return MMS_ERR_LAUNCH;
    const size_t off = round_up((size_t)N * D * sizeof(float), 256);
    return simmatrix_backward(N, D, D, q, a, W, top_diff, 1, 1, 1, dq, da, dW, nullptr, base + off,
                              ws_bytes - off, s);
  }
  // U_nm = dT_nm A_n  (W1 x D x W2) ;  V_nm = dT_nm^T Q_n  (W2 x D x W1)
  if (W1 == 1 && W2 == 1 && M == 1) {
    hipLaunchKernelGGL(rowscale_kernel, dim3(ew_blocks(R2 * D)), dim3(256), 0, s, a, top_diff, U,
                       R2, D);
    hipLaunchKernelGGL(rowscale_kernel, dim3(ew_blocks(R1 * D)), dim3(256), 0, s, q, top_diff, V,
                       R1, D);
  } else {
    GemmArgs guv[2];
    guv[0] = gemm_args(W1, D, W2, top_diff, W2, 1, a, D, 1, U, D);
    guv[0].nb1 = M;
    guv[0].a_b0 = (long long)M * W1 * W2; guv[0].a_b1 = (long long)W1 * W2;
    guv[0].b_b0 = (long long)W2 * D; guv[0].b_b1 = 0;
    guv[0].c_b0 = (long long)W1 * D; guv[0].c_b1 = R1 * D;
    guv[1] = gemm_args(W2, D, W1, top_diff, 1, W2, q, D, 1, V, D);
    guv[1].nb1 = M;
    guv[1].a_b0 = (long long)M * W1 * W2; guv[1].a_b1 = (long long)W1 * W2;
    guv[1].b_b0 = (long long)W1 * D; guv[1].b_b1 = 0;
    guv[1].c_b0 = (long long)W2 * D; guv[1].c_b1 = R2 * D;
    const int nb[2] = {N, N};
    if (!gemm_launch_group(guv, nb, 2, s)) {      // small batches: U and V in one launch
      gemm_launch(guv[0], N, s);
      gemm_launch(guv[1], N, s);
    }
  }
  // dQ_all = sum_m U_m W_m^T ; dA_all = sum_m V_m W_m   (:291-299; m = 0 overwrites, which also realises
  // the unconditional zeroing of :176-177) ; dW_m = Q_all^T U_m  (:286-289), K = N*W1 split across
  // workgroups; W.diff is overwritten because the reference zeroes it first (:256).  The three products
  // are independent: one grouped launch when they are small, then one grouped reduction.
  GemmArgs g3[3];
  const int one3[3] = {1, 1, 1};
  const long long nW = (long long)M * D * D;
  g3[2] = gemm_args(D, D, (int)R1, q, 1, D, U, D, 1, part, D);
  g3[2].nb1 = M; g3[2].b_b1 = R1 * D; g3[2].c_b1 = (long long)D * D;
  g3[2].ksplit = lay.ksplit; g3[2].kchunk = lay.kchunk; g3[2].c_ks = nW;
  if (M == 1) {
    g3[0] = gemm_args((int)R1, D, D, U, D, 1, W, 1, D, dq, D);
    g3[1] = gemm_args((int)R2, D, D, V, D, 1, W, D, 1, da, D);
    if (!gemm_launch_group(g3, one3, 3, s)) {
      gemm_launch(g3[0], 1, s);
      gemm_launch(g3[1], 1, s);
      gemm_launch(g3[2], 1, s);
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ew_blocks(nW)), dim3(256), 0, s, part, lay.ksplit, nW,
                       dW, 0);
  } else {
    // the M products of one operand run as ONE "stacked" split-K product (chunk m = measure m: M x the
    // workgroups of a single product, which alone covers a fraction of the chip at the driver's sizes);
    // their sum over m, ascending -- the order the reference accumulates in -- is taken by the reduction.
    float* mq = reinterpret_cast<float*>(base + lay.mpart_off);
    float* ma = reinterpret_cast<float*>(base + lay.mpart2_off);
    g3[0] = gemm_args((int)R1, D, D, U, D, 1, W, 1, D, mq, D);
    g3[0].ksplit = M; g3[0].ks_stacked = 1; g3[0].a_ks = R1 * D; g3[0].b_ks = (long long)D * D; g3[0].c_ks = R1 * D;
    g3[1] = gemm_args((int)R2, D, D, V, D, 1, W, D, 1, ma, D);
    g3[1].ksplit = M; g3[1].ks_stacked = 1; g3[1].a_ks = R2 * D; g3[1].b_ks = (long long)D * D; g3[1].c_ks = R2 * D;
    if (!gemm_launch_group(g3, one3, 3, s)) {
      gemm_launch(g3[0], 1, s);
      gemm_launch(g3[1], 1, s);
      gemm_launch(g3[2], 1, s);
    }
    ReduceGroup rg{};
    const float* parts[3] = {mq, ma, part};
    float* outs[3] = {dq, da, dW};
    const long long ns[3] = {R1 * D, R2 * D, nW};
    const int sp[3] = {M, M, lay.ksplit};
    int first = 0;
    for (int i = 0; i < 3; ++i) {
      rg.part[i] = parts[i]; rg.out[i] = outs[i]; rg.n[i] = ns[i]; rg.splits[i] = sp[i];
      rg.first[i] = first;
      first += (int)ew_blocks(ns[i]);
    }
    rg.first[3] = first;
    rg.cnt = 3;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(first), dim3(256), 0, s, rg);
  }
  return launch_status();
}

// ---------------------------------- SimMatrix -------------------------------
struct SimMatrixWs {
  size_t u_off, part_off, wt_off, img_off, total;
  int ksplit, kchunk;
};

// Which matrix pipe the tall-times-weight products of the learned-metric paths run on (mms_set_matrix_mode):
// 0 (default) = the bf16 pipe on exact three-way splits of the fp32 operands (bx3_gemm.h), 1 = fp32 MFMA (panel_gemm.h).
static int g_matrix_mode = 0;
int set_matrix_mode(int mode) {
  if (mode != 0 && mode != 1) return MMS_ERR_INVALID_ARG;
  g_matrix_mode = mode;
  return MMS_OK;
}
int get_matrix_mode() { return g_matrix_mode; }
// below this many rows the fp32 kernel's 64-row panels fill the chip better and the split launch is not worth its 3 us
static bool bx3_rows_worth(int M) { return M >= 2048; }
static SimMatrixWs simmatrix_ws(int N, int K1, int K2) {
  SimMatrixWs w{};
  w.ksplit = pick_ksplit(K1, K2, N, &w.kchunk);
  int pchunk = 0;
  const int psplit = panel_pick_ksplit((K1 + 63) / 64, 1, N, &pchunk);   // the panel kernel's split (if it runs)
  w.u_off = 0;
  w.part_off = round_up((size_t)N * K2 * sizeof(float), 256);
  int tchunk = 0;
  const int tsplit = bx3_tn_pick_chunks(N, bx3_tn_quads(K1, K2), &tchunk);      // the split-bf16 dW kernel's split (if it runs)
  int slabs = psplit > w.ksplit ? psplit : w.ksplit;
  if (tsplit > slabs) slabs = tsplit;
  w.wt_off = w.part_off + round_up((size_t)slabs * K1 * K2 * sizeof(float), 256);
  w.img_off = w.wt_off + round_up((size_t)K1 * K2 * sizeof(float), 256);    // W^T for the dq product (fp32 MFMA mode)
  const size_t ia = bx3_image_bytes(K2, K1), ib = bx3_image_bytes(K1, K2);  // the split image of W (forward) or W^T (dq)
  w.total = w.img_off + round_up(ia > ib ? ia : ib, 256);
  return w;
}
size_t simmatrix_workspace_bytes(int N, int K1, int K2) { return simmatrix_ws(N, K1, K2).total; }

int simmatrix_forward(int N, int K1, int K2, const float* q, const float* a, const float* W,
                      float* top, float* qw, hipStream_t s, const float* rd_bias, void* ws, size_t ws_bytes) {
  // qw = Q W  (:60-61) ; top_i = a_i . qw_i  (:62-64)
  if (g_matrix_mode == 0 && ws && bx3_rows_worth(N)) {
    const SimMatrixWs lay = simmatrix_ws(N, K1, K2);
    Bx3Args b{};
    b.M = N; b.N = K2; b.K = K1; b.A = q; b.lda = K1; b.C = qw; b.ldc = K2;
    b.Y = a; b.ldy = K2; b.rowdot = top; b.rd_stride = 1; b.rd_bias = rd_bias;
    if (ws_bytes >= lay.total && bx3_eligible(b)) {
      bx3_u4* img = reinterpret_cast<bx3_u4*>(static_cast<char*>(ws) + lay.img_off);
      b.img = img;
      // the image of W; its launch also zeroes the scores when two column groups add their halves into them
      bx3_split_b(W, K2, 1, K1, K2, img, s, bx3_groups(K2) == 2 ? top : nullptr, 1, N);
      bx3_launch(b, s);
      return launch_status();
    }
  }
  {
    // one launch: the row dot is the product's epilogue
    PanelArgs p = panel_args(N, K2, K1, q, K1, W, K2, qw, K2);
    p.Y = a; p.ldy = K2; p.rowdot = top; p.rd_stride = 1;
    p.rd_bias = rd_bias;                        // SimCross bilinear's bias (one scalar at W1 = W2 = 1), else null
    if (panel_eligible(p, true)) {
      panel_launch(p, true, s);
      return launch_status();
    }
  }
  GemmArgs g = gemm_args(N, K2, K1, q, K1, 1, W, K2, 1, qw, K2);
  gemm_launch(g, 1, s);
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, a, qw, rd_bias,
                     top, (long long)N, K2, 1LL);
  return launch_status();
}

// fp16-STORAGE scoring (round 3): q (N, K1) and a (N, K2) are IEEE halves in HBM, W (K1, K2) and the scores fp32.
// top_i = a_i . (q_i W) on the bf16 pipe: a half is the exact sum of two bf16 values, the weight of three, so five of
// the six partial products exist and each is exact in fp32 -- the result is the fp32 layer's on the widened inputs
// to fp32 rounding (1e-5 bar as everywhere BLAS-ordered).  No Q.W output: scoring does not need it.  No fp32 fallback:
// shapes outside the kernel are MMS_ERR_UNSUPPORTED.
int simmatrix_forward_f16(int N, int K1, int K2, const void* q, const void* a, const float* W, float* top, void* ws,
                          size_t ws_bytes, hipStream_t s) {
  const SimMatrixWs lay = simmatrix_ws(N, K1, K2);
  if (!ws || ws_bytes < lay.total) return MMS_ERR_WORKSPACE;
  Bx3Args b{};
  b.M = N; b.N = K2; b.K = K1; b.A = static_cast<const float*>(q); b.lda = K1; b.a_half = 1;
  b.Y = static_cast<const float*>(a); b.ldy = K2; b.rowdot = top; b.rd_stride = 1;
  if (!bx3_eligible(b)) return MMS_ERR_UNSUPPORTED;
  bx3_u4* img = reinterpret_cast<bx3_u4*>(static_cast<char*>(ws) + lay.img_off);
  b.img = img;
  bx3_split_b(W, K2, 1, K1, K2, img, s, bx3_groups(K2) == 2 ? top : nullptr, 1, N);
  bx3_launch(b, s);
  return launch_status();
}

// da (halves) = diag(dT) . P  (P fp32: the training forward's Q.W), RNE at the store
__global__ __launch_bounds__(256) void rowscale_to_half_kernel(const float4* __restrict__ P, const float* __restrict__ dT,
                                                               void* __restrict__ out, long long rows, int cols4) {
  typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
  const long long n = rows * cols4, stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
    const float4 v = P[e];
    const float sc = dT[e / cols4];
    const hf4 o = {(_Float16)(0.f + sc * v.x), (_Float16)(0.f + sc * v.y), (_Float16)(0.f + sc * v.z), (_Float16)(0.f + sc * v.w)};
    __builtin_nontemporal_store(o, reinterpret_cast<hf4*>(out) + e);
  }
}

// fp16-STORAGE training forward / backward of SimMatrix (round 3): q, a and the bottom gradients dq, da are halves in
// HBM; W, dW, the scores, top_diff and the forward's Q.W (qw, (N, K2), the scratch the backward scales into da) fp32.
// All three products run on the bf16 pipe with the half operands split exactly into two planes (bx3_gemm.h);
// gradients are rounded to half (RNE) at the store.  No fp32 fallback: MMS_ERR_UNSUPPORTED outside the kernels' shapes.
int simmatrix_forward_train_f16(int N, int K1, int K2, const void* q, const void* a, const float* W, float* top, float* qw,
                                void* ws, size_t ws_bytes, hipStream_t s) {
  const SimMatrixWs lay = simmatrix_ws(N, K1, K2);
  if (!ws || ws_bytes < lay.total) return MMS_ERR_WORKSPACE;
  Bx3Args b{};
  b.M = N; b.N = K2; b.K = K1; b.A = static_cast<const float*>(q); b.lda = K1; b.a_half = 1; b.C = qw; b.ldc = K2;
  b.Y = static_cast<const float*>(a); b.ldy = K2; b.rowdot = top; b.rd_stride = 1;
  if (!bx3_eligible(b)) return MMS_ERR_UNSUPPORTED;
  bx3_u4* img = reinterpret_cast<bx3_u4*>(static_cast<char*>(ws) + lay.img_off);
  b.img = img;
  bx3_split_b(W, K2, 1, K1, K2, img, s, bx3_groups(K2) == 2 ? top : nullptr, 1, N);
  bx3_launch(b, s);
  return launch_status();
}
int simmatrix_backward_f16(int N, int K1, int K2, const void* q, const void* a, const float* W, const float* qw,
                           const float* top_diff, void* dq, void* da, float* dW, void* ws, size_t ws_bytes, hipStream_t s) {
  const SimMatrixWs lay = simmatrix_ws(N, K1, K2);
  if (!ws || ws_bytes < lay.total) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  float* part = reinterpret_cast<float*>(base + lay.part_off);
  bx3_u4* img = reinterpret_cast<bx3_u4*>(base + lay.img_off);
  Bx3TnArgs t{};
  t.M = K1; t.N = K2; t.K = N; t.A = static_cast<const float*>(q); t.lda = K1; t.B = static_cast<const float*>(a); t.ldb = K2;
  t.kscale = top_diff; t.C = part; t.c_ks = (long long)K1 * K2; t.ab_half = 1;
  t.nchunks = bx3_tn_pick_chunks(N, bx3_tn_quads(K1, K2), &t.kchunk);
  Bx3Args bq{};
  bq.M = N; bq.N = K1; bq.K = K2; bq.A = static_cast<const float*>(a); bq.lda = K2; bq.a_half = 1;
  bq.C = static_cast<float*>(dq); bq.ldc = K1; bq.c_half = 1; bq.rowscale = top_diff; bq.stream_c = 1; bq.img = img;
  bool da_side = false;
  if (dq && da && qw && K2 >= 8) {                 // da rides in the dq launch's loader waves (as in the fp32 path)
    bq.side_in = qw; bq.side_out = static_cast<float*>(da); bq.side_scale = top_diff; bq.side_ld = K2; bq.side_cols = K2;
    bq.side_half = 1;
    da_side = bx3_eligible(bq);
    if (!da_side) { bq.side_in = nullptr; bq.side_out = nullptr; bq.side_scale = nullptr; bq.side_half = 0; }
  }
  if ((dW && !bx3_tn_eligible(t)) || (dq && !bx3_eligible(bq)) || (da && (!qw || (K2 & 3) != 0 || !aligned16(qw) ||
                                                                         (reinterpret_cast<uintptr_t>(da) & 7u) != 0)))
    return MMS_ERR_UNSUPPORTED;
  if (dW) {
    // dW += Q^T diag(dT) A   (:73-80), both operands widened and split on the fly
    bx3_tn_launch(t, s);
    const unsigned rb = ew_blocks((long long)K1 * K2);
    if (dq) {
      const Bx3SplitArgs sp = bx3_split_args(W, 1, K2, K2, K1, img);
      hipLaunchKernelGGL(splitk_reduce_split_kernel, dim3(rb + bx3_split_blocks(sp)), dim3(256), 0, s, part, t.nchunks,
                         (long long)K1 * K2, dW, 1, (int)rb, sp);
    } else {
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, s, part, t.nchunks, (long long)K1 * K2, dW, 1);
    }
  } else if (dq) {
    bx3_split_b(W, 1, K2, K2, K1, img, s);
  }
  if (dq) bx3_launch(bq, s);                       // dq_j = dT_j * (W a_j)   (:88, NoTrans)
  if (da && !da_side)                              // da_j = dT_j * (W^T q_j) (:88, Trans): the forward's product, scaled
    hipLaunchKernelGGL(rowscale_to_half_kernel, dim3(ew_blocks((long long)N * (K2 / 4))), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(qw), top_diff, da, (long long)N, K2 / 4);
  return launch_status();
}

// qw (optional): the forward's Q.W, unchanged since; may alias da.  da_j = dT_j * (W^T q_j) is row j of
// Q.W scaled by dT_j -- the product the forward already made with the same kernel and k order, so
// reusing it returns the same bits as recomputing it and saves one of the four GEMMs of a step.
int simmatrix_backward(int N, int K1, int K2, const float* q, const float* a, const float* W,
                       const float* top_diff, int ppd, int pd0, int pd1, float* dq, float* da,
                       float* dW, const float* qw, void* ws, size_t ws_bytes, hipStream_t s) {
  const SimMatrixWs lay = simmatrix_ws(N, K1, K2);
  // will the dq product take the panel kernel (and need W^T)?  Then its transpose rides in the dW reduction's launch.
  float* const Wt_ws = (ws && ws_bytes >= lay.total) ? reinterpret_cast<float*>(static_cast<char*>(ws) + lay.wt_off) : nullptr;
  bool dq_panel = false, wt_done = false;
  Bx3Args bq{};                                 // the dq product on the bf16 pipe (matrix mode 0), if it can run there
  bool dq_bx3 = false;
  if (pd0 && g_matrix_mode == 0 && ws && ws_bytes >= lay.total && bx3_rows_worth(N)) {
    bq.M = N; bq.N = K1; bq.K = K2; bq.A = a; bq.lda = K2; bq.C = dq; bq.ldc = K1; bq.rowscale = top_diff; bq.stream_c = 1;
    if (pd1 && qw && (K2 & 3) == 0 && K2 >= 8) {
      bq.side_in = qw; bq.side_out = da; bq.side_scale = top_diff; bq.side_ld = K2; bq.side_cols = K2;
    }
    dq_bx3 = bx3_eligible(bq);
    if (!dq_bx3 && bq.side_in) { bq.side_in = nullptr; bq.side_out = nullptr; bq.side_scale = nullptr; dq_bx3 = bx3_eligible(bq); }
  }
  if (pd0 && Wt_ws && !dq_bx3) {
    PanelArgs pq = panel_args(N, K1, K2, a, K2, Wt_ws, K1, dq, K1);
    pq.rowscale = top_diff;
    dq_panel = panel_eligible(pq, true);
  }
  if (ppd) {
    if (!ws || ws_bytes < lay.total) return MMS_ERR_WORKSPACE;
    char* base = static_cast<char*>(ws);
    float* U = reinterpret_cast<float*>(base + lay.u_off);
    float* part = reinterpret_cast<float*>(base + lay.part_off);
    // dW += sum_i dT_i q_i a_i^T = Q^T (diag(dT) A)   (:73-80, accumulating)
    bool dw_done = false;
    if (g_matrix_mode == 0 && bx3_rows_worth(N)) {
      // on the bf16 pipe: both operands split on the fly (bx3_gemm.h, bx3_tn_kernel), slabs summed in chunk order
      Bx3TnArgs t{};
      t.M = K1; t.N = K2; t.K = N; t.A = q; t.lda = K1; t.B = a; t.ldb = K2; t.kscale = top_diff; t.C = part;
      t.c_ks = (long long)K1 * K2;
      t.nchunks = bx3_tn_pick_chunks(N, bx3_tn_quads(K1, K2), &t.kchunk);
      if (bx3_tn_eligible(t)) {
        bx3_tn_launch(t, s);
        const unsigned rb = ew_blocks((long long)K1 * K2);
        if (dq_bx3) {                            // the image of W^T for the dq product rides in the reduction's launch
          bx3_u4* img = reinterpret_cast<bx3_u4*>(static_cast<char*>(ws) + lay.img_off);
          const Bx3SplitArgs sp = bx3_split_args(W, 1, K2, K2, K1, img);
          hipLaunchKernelGGL(splitk_reduce_split_kernel, dim3(rb + bx3_split_blocks(sp)), dim3(256), 0, s, part, t.nchunks,
                             (long long)K1 * K2, dW, 1, (int)rb, sp);
          wt_done = true;
        } else {
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, s, part, t.nchunks, (long long)K1 * K2, dW, 1);
        }
        dw_done = true;
      }
    }
    if (!dw_done) {
      PanelArgs p = panel_args(K1, K2, N, q, K1, a, K2, part, K2);
      p.kscale = top_diff;                      // A(i, k = pair) = q_k[i] * dT_k
      p.ksplit = panel_pick_ksplit(p.row_blocks, 1, N, &p.kchunk);
      p.c_ks = (long long)K1 * K2;
      if (p.ksplit > 1 && panel_eligible(p, false)) {
        panel_launch(p, false, s);
        const unsigned rb = ew_blocks((long long)K1 * K2);
        if (dq_panel) {
          const unsigned tb = (unsigned)(((K2 + 31) / 32) * ((K1 + 31) / 32));
          hipLaunchKernelGGL(splitk_reduce_transpose_kernel, dim3(rb + tb), dim3(256), 0, s, part, p.ksplit,
                             (long long)K1 * K2, dW, 1, (int)rb, W, Wt_ws, K1, K2);
          wt_done = true;
        } else {
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, s, part, p.ksplit, (long long)K1 * K2, dW, 1);
        }
        dw_done = true;
      }
    }
    if (!dw_done) {
    GemmArgs g = gemm_args(K1, K2, N, q, 1, K1, a, K2, 1, part, K2);
    g.ksplit = lay.ksplit; g.kchunk = lay.kchunk; g.c_ks = (long long)K1 * K2;
    g.bkscale = top_diff;                       // B(k = pair, j) = dT_k * a_k[j], scaled on load
    if (!gemm_fast_variant(g)) {                // generic kernel: materialise U = diag(dT) A first
      hipLaunchKernelGGL(rowscale_kernel, dim3(ew_blocks((long long)N * K2)), dim3(256), 0, s, a,
                         top_diff, U, (long long)N, K2);
      g.B = U;
      g.bkscale = nullptr;
    }
    gemm_launch(g, 1, s);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ew_blocks((long long)K1 * K2)), dim3(256), 0, s,
                       part, lay.ksplit, (long long)K1 * K2, dW, 1);
    }
  }
  bool da_done = false;
  if (pd0 && dq_bx3) {
    // dq_j = dT_j * (W a_j)   (:88, NoTrans, beta 0): B(k, n) = W[n][k], split straight from W's rows
    bx3_u4* img = reinterpret_cast<bx3_u4*>(static_cast<char*>(ws) + lay.img_off);
    bq.img = img;
    if (!wt_done) bx3_split_b(W, 1, K2, K2, K1, img, s);
    bx3_launch(bq, s);
    da_done = bq.side_in != nullptr;
  } else if (pd0) {
    // dq_j = dT_j * (W a_j)   (:88, NoTrans, beta 0)
    float* Wt = Wt_ws;
    PanelArgs p = panel_args(N, K1, K2, a, K2, Wt, K1, dq, K1);     // B(k, n) = W[n][k] = Wt[k][n]
    p.rowscale = top_diff;
    p.stream_c = 1;                             // read next by another layer, not by this call
    if (pd1 && qw && K2 <= 304) {
      // da_j = dT_j * (row j of the forward's Q.W): a streaming pass with no arithmetic to speak of, carried
      // by this product's loader waves while its compute waves keep the matrix pipe busy
      p.side_in = qw; p.side_out = da; p.side_scale = top_diff; p.side_ld = K2; p.side_cols = K2;
    }
    if (Wt && panel_eligible(p, true)) {
      if (!wt_done)
        hipLaunchKernelGGL(pg_transpose_kernel, dim3((K2 + 31) / 32, (K1 + 31) / 32), dim3(256), 0, s, W, Wt, K1, K2);
      panel_launch(p, true, s);
      da_done = p.side_in != nullptr;
    } else {
      GemmArgs g = gemm_args(N, K1, K2, a, K2, 1, W, 1, K2, dq, K1);
      g.rowscale = top_diff;
      g.stream_c = 1;
      gemm_launch(g, 1, s);
    }
  }
  if (da_done) {
    // written by the dq launch
  } else if (pd1 && qw) {
    if ((K2 & 3) == 0 && aligned16(qw) && aligned16(da)) {
      hipLaunchKernelGGL(rowscale4_kernel, dim3(ew_blocks((long long)N * (K2 / 4))), dim3(256), 0, s,
                         reinterpret_cast<const float4*>(qw), top_diff, reinterpret_cast<float4*>(da),
                         (long long)N, K2 / 4);
    } else {
      const float* x = qw;
      hipLaunchKernelGGL(rowscale_inplace_ok_kernel, dim3(ew_blocks((long long)N * K2)), dim3(256), 0, s, x,
                         top_diff, da, (long long)N, K2);
    }
  } else if (pd1 && g_matrix_mode == 0 && ws && ws_bytes >= lay.total && bx3_rows_worth(N) && [&] {
               // da_j = dT_j * (W^T q_j)   (:88, Trans, beta 0): the forward's product (same kernel, same image, same
               // k order: the bits of the cached form above), scaled in its epilogue
               Bx3Args b{};
               b.M = N; b.N = K2; b.K = K1; b.A = q; b.lda = K1; b.C = da; b.ldc = K2; b.rowscale = top_diff; b.stream_c = 1;
               if (!bx3_eligible(b)) return false;
               bx3_u4* img = reinterpret_cast<bx3_u4*>(static_cast<char*>(ws) + lay.img_off);
               b.img = img;
               bx3_split_b(W, K2, 1, K1, K2, img, s);
               bx3_launch(b, s);
               return true;
             }()) {
    // written by the launch above
  } else if (pd1) {
    // da_j = dT_j * (W^T q_j)   (:88, Trans, beta 0)
    PanelArgs p = panel_args(N, K2, K1, q, K1, W, K2, da, K2);
    p.rowscale = top_diff;
    p.stream_c = 1;                             // read next by another layer, not by this call
    if (panel_eligible(p, true)) {
      panel_launch(p, true, s);
    } else {
      GemmArgs g = gemm_args(N, K2, K1, q, K1, 1, W, K2, 1, da, K2);
      g.rowscale = top_diff;
      g.stream_c = 1;
      gemm_launch(g, 1, s);
    }
  }
  return launch_status();
}

// ------------------------- fused learned-metric triplet step (round 3) -------------------------
// The net  SimMatrix(q, a+) , SimMatrix(q, a-)  (W shared by parameter name) -> PairRankLoss, forward and backward, as
// THREE products instead of the layers' six (sim_matrix_layer.cpp:53-95 twice, pair_rank_loss_layer.cpp:26-84):
//   P = Q W is the same for both branches: one product, whose epilogue takes both row dots s+ = P_i . a+_i and
//   s- = P_i . a-_i, PairRankLoss's term and gradients g+, g- for the row, and writes da+ = g+ P_i, da- = g- P_i
//   (sim_matrix_layer.cpp:88, Trans) and B_i = g+ a+_i + g- a-_i;
//   dq = B W^T   (the Split sum of the two branches' dq_i = g W a_i, :88 NoTrans, as one product);
//   dW += Q^T B  (the two branches' sum_i g_i q_i a_i^T, :73-80, as one split-K product).
// Neither P nor the (N, 1) score gradients reach HBM.
int triplet_loss_from_terms(const float* terms, int N, float* loss, hipStream_t s);
int pairrank_hinge_mode();

struct TripSimWs {
  size_t b_off, terms_off, ones_off, wt_off, part_off, img_off, total;
  int ksplit, kchunk;
};
static TripSimWs tripsim_ws(int N, int K1, int K2) {
  TripSimWs w{};
  w.ksplit = panel_pick_ksplit((K1 + 63) / 64, 1, N, &w.kchunk);
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += round_up(b, 256); return at; };
  w.b_off = take((size_t)N * K2 * sizeof(float));
  w.terms_off = take((size_t)N * sizeof(float));
  w.ones_off = take((size_t)N * sizeof(float));
  w.wt_off = take((size_t)K1 * K2 * sizeof(float));
  int tchunk = 0;
  const int tsplit = bx3_tn_pick_chunks(N, bx3_tn_quads(K1, K2), &tchunk);   // the bf16-pipe dW kernel's split (if it runs)
  const int slabs = tsplit > w.ksplit ? tsplit : (w.ksplit > 0 ? w.ksplit : 1);
  w.part_off = take((size_t)slabs * K1 * K2 * sizeof(float));
  w.img_off = take(bx3_image_bytes(K1, K2));                                  // the split image of W^T (dq on the bf16 pipe)
  w.total = o;
  return w;
}
size_t triplet_simmatrix_workspace_bytes(int N, int K1, int K2) { return tripsim_ws(N, K1, K2).total; }

// MMS_ERR_UNSUPPORTED when the shapes are outside the panel kernel (the caller then runs the layers one by one)
int triplet_simmatrix_step(int N, int K1, int K2, float margin, float loss_weight, const float* q, const float* ap,
                           const float* an, const float* y, const float* W, float* s_pos, float* s_neg, float* loss,
                           float* dq, float* dap, float* dan, float* dW, void* ws, size_t ws_bytes, hipStream_t s) {
  const TripSimWs lay = tripsim_ws(N, K1, K2);
  if (!ws || ws_bytes < lay.total) return MMS_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  float* B = reinterpret_cast<float*>(base + lay.b_off);
  float* terms = reinterpret_cast<float*>(base + lay.terms_off);
  float* ones = reinterpret_cast<float*>(base + lay.ones_off);
  float* Wt = reinterpret_cast<float*>(base + lay.wt_off);
  float* part = reinterpret_cast<float*>(base + lay.part_off);
  const float scale = loss_weight / (float)N;                       // pair_rank_loss_layer.cpp:64, count = N * 1
  // P = Q W with the triplet epilogue
  PanelArgs p1 = panel_args(N, K2, K1, q, K1, W, K2, nullptr, K2);
  p1.Y = ap; p1.Y2 = an; p1.ldy = K2; p1.rowdot = s_pos; p1.rd_stride = 1;
  p1.trip_y = y; p1.trip_margin = margin; p1.trip_s0 = -1.0f * scale; p1.trip_s1 = 1.0f * scale;
  p1.trip_hinge_ge = pairrank_hinge_mode() == MMS_PAIRRANK_HINGE_GPU ? 1 : 0;
  p1.trip_sneg = s_neg; p1.trip_terms = terms; p1.trip_dapos = dap; p1.trip_daneg = dan; p1.trip_b = B;
  // dq = B W^T  (B(k, n) = W[n][k] = Wt[k][n])
  PanelArgs p2 = panel_args(N, K1, K2, B, K2, Wt, K1, dq, K1);
  p2.stream_c = 1;
  // dW += Q^T B, split over the pairs
  PanelArgs p3 = panel_args(K1, K2, N, q, K1, B, K2, part, K2);
  p3.kscale = ones;
  p3.ksplit = lay.ksplit; p3.kchunk = lay.kchunk; p3.c_ks = (long long)K1 * K2;
  if (!panel_eligible(p1, true) || !panel_eligible(p2, true) || p3.ksplit <= 1 || !panel_eligible(p3, false))
    return MMS_ERR_UNSUPPORTED;
  // (the ones are the fp32 split-K kernel's k-scale; the bf16-pipe dW kernel takes "no scale" as such)
  Bx3TnArgs t{};
  Bx3Args bq{};
  bool back_bx3 = false;
  if (g_matrix_mode == 0 && bx3_rows_worth(N)) {
    t.M = K1; t.N = K2; t.K = N; t.A = q; t.lda = K1; t.B = B; t.ldb = K2; t.kscale = nullptr; t.C = part;
    t.c_ks = (long long)K1 * K2;
    t.nchunks = bx3_tn_pick_chunks(N, bx3_tn_quads(K1, K2), &t.kchunk);
    bq.M = N; bq.N = K1; bq.K = K2; bq.A = B; bq.lda = K2; bq.C = dq; bq.ldc = K1; bq.stream_c = 1;
    bq.img = reinterpret_cast<bx3_u4*>(base + lay.img_off);
    back_bx3 = bx3_tn_eligible(t) && bx3_eligible(bq);
  }
  if (!back_bx3 &&
      hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ones), 0x3f800000, (size_t)N, s) != hipSuccess) return MMS_ERR_LAUNCH;
  panel_launch(p1, true, s);
  if (loss) {
    const int rc = triplet_loss_from_terms(terms, N, loss, s);
    if (rc != MMS_OK) return rc;
  }
  // The two backward products on the bf16 pipe (matrix mode 0; bx3_gemm.h): dW += Q^T B from both operands split on the
  // fly, dq = B W^T with the image of W^T built in the reduction's launch.  (The forward stays on the fp32 pipe: its
  // epilogue needs whole rows of Q W in one workgroup, the bf16 kernel's workgroups own half a row each.)
  if (back_bx3) {
    bx3_tn_launch(t, s);
    const unsigned rb = ew_blocks((long long)K1 * K2);
    const Bx3SplitArgs sp = bx3_split_args(W, 1, K2, K2, K1, const_cast<bx3_u4*>(bq.img));
    hipLaunchKernelGGL(splitk_reduce_split_kernel, dim3(rb + bx3_split_blocks(sp)), dim3(256), 0, s, part, t.nchunks,
                       (long long)K1 * K2, dW, 1, (int)rb, sp);
    bx3_launch(bq, s);
    return launch_status();
  }
  panel_launch(p3, false, s);
  {
    const unsigned rb = ew_blocks((long long)K1 * K2), tb = (unsigned)(((K2 + 31) / 32) * ((K1 + 31) / 32));
    hipLaunchKernelGGL(splitk_reduce_transpose_kernel, dim3(rb + tb), dim3(256), 0, s, part, p3.ksplit,
                       (long long)K1 * K2, dW, 1, (int)rb, W, Wt, K1, K2);
  }
  panel_launch(p2, true, s);
  return launch_status();
}

}  // namespace mms
