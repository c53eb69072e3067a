// csrc/bx3_gemm.h -- the tall-times-small-weight products of the learned-metric paths (C[M x N] = A[M x K] . B[K x N],
// M = all pairs, N, K <= 320: SimMatrix's Q W and (dT A) W^T, sim_matrix_layer.cpp:60-61, :88) on the BF16 matrix
// pipe at fp32 accuracy (round 3).  Included by bilinear.hip.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (157 TF); round 2/3 measured the fp32 panel kernel at
// 52-56 % of it with every remedy tried.  The bf16 pipe is 16x faster per instruction-cycle, and an fp32
// value is EXACTLY the sum of three bf16 values:  x = h + m + l,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
// (8 + 8 + 8 significant bits, RNE at each level; the two subtractions are exact in fp32).  A product of two fp32
// values is then nine bf16 products, each EXACT in fp32 (16 significant bits); the six of weight >= 2^-16 relative
// -- hh, hm, mh, mm, hl, lh -- are kept, the three of weight <= 2^-24 (ml, lm, ll) dropped: per-product relative error
// < 2^-22, fp32 accumulate in the MFMA.  Six v_mfma_f32_32x32x16_bf16 do the work of sixteen fp32 16x16x4 steps in
// 6/16 of the pipe time; results agree with an fp32 BLAS product inside these layers' 1e-5 contract (the reference
// calls cblas_sgemm, no defined order) -- tests/test_gpu_parity.py holds them to fp64.  Inputs holding an infinity
// come out as NaN (inf - inf in the split), where an fp32 product gives inf or NaN; so do magnitudes above bf16's largest
// finite value (3.39e38: the first plane rounds to infinity).
//
// Shape of a launch:
//   * the weight B is split ONCE per call by bx3_split_b_kernel into an operand IMAGE: for k-step s (16 deep), column
//     tile t (32 wide), plane p in {h, m, l}: a 1-KB block = lane l's 8 bf16 (B(16s + 8(l/32) + j, 32t + l%32)), i.e.
//     exactly what ds_read_b128 hands the MFMA; k >= K and n >= N are zeros.  Strided source: W^T costs nothing extra;
//   * what bounds the product is no longer the matrix pipe (6 of 16 fp32-steps' time) but the bytes a CU takes in from
//     L2: every workgroup streams the weight image (6 bytes per element).  Measured (tools/bx3bench.hip, round 3): a
//     64-row x 320-column workgroup takes in 660 KB and its loaders alone need 20 us (36 GB/s per CU, 9.7 TB/s chip-wide;
//     4 or 8 loader waves, rotated k order: the same).  Bytes per workgroup are least for a near-square output tile, so
//     a workgroup owns 128 ROWS x ONE column group of 32 NTW <= 160 columns (N = 300: two groups; the two workgroups of
//     a row panel run on the same XCD and share the panel's rows in its L2): 437 KB;
//   * 4 compute waves (32 rows x NTW accumulator tiles of 32x32 each) + 4 loader waves, one per SIMD beside a compute
//     wave (an LDS-DMA instruction holds its wave's issue for 60-180 cycles: panel_gemm.h);
//   * per k-step loaders 0-2 bring plane h / m / l of the NTW image blocks and loader 3 the panel's 128 x 16 raw fp32
//     A values by LDS-DMA (global_load_lds_dwordx4) into a SIX-slot ring, three steps in flight beyond the two that
//     must have landed, one counted vmcnt + one barrier per step; the A lanes fetch (row l%32, 4 k) so that the compute
//     wave's two ds_read_b128 are linear in the lane id;
//   * a compute wave splits its A fragment in registers (44 VALU per k-step, issued in the shadow of 6 NTW MFMAs of
//     32 cycles each: unlike the fp32 MFMA, the bf16 MFMA holds vector issue for 8 of its 32 cycles only -- measured
//     cost 2.6-2.9 cycles per VALU all the same, tools/bx3struct.hip); the next step's A fragment and B fragments are
//     read while the current step's MFMAs run;
//   * epilogue as panel_gemm.h: accumulators through LDS, 16-byte row-segment stores, the per-row scale and the
//     row dot of the SimMatrix forward folded in (two column groups: each adds its half of the row dot to a zeroed
//     output -- two addends, so the sum does not depend on who arrives first);
//   * the loader waves also carry the backward's streaming side job (Bx3Side) and request the row dot's operand
//     ahead of the epilogue; AH = true is the fp16-STORAGE form (A, Y and optionally C as halves: two planes, five
//     products).  Further down: bx3_tn_kernel, the weight gradient (both operands activations, split-K).
#ifndef MMS_BX3_GEMM_H_
#define MMS_BX3_GEMM_H_

#include <type_traits>

#include "mms_common.h"
#include "panel_gemm.h"

namespace mms {

typedef __bf16 bx3_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx3_h2 __attribute__((ext_vector_type(2)));
typedef float bx3_f2 __attribute__((ext_vector_type(2)));
typedef float bx3_f16 __attribute__((ext_vector_type(16)));
typedef unsigned bx3_u4 __attribute__((ext_vector_type(4)));

struct Bx3Args {
  int M, N, K;
  const float* A; long long lda;          // A(i,k) = A[i*lda + k]   (lda % 4 == 0, K % 4 == 0, 16-byte aligned)
  const bx3_u4* img;                      // operand image of B (bx3_split_b)
  float* C; long long ldc;                // may be null (row dot only)
  const float* rowscale;                  // v(i,:) = rowscale[i] * acc(i,:)
  const float* Y; long long ldy;          // rowdot[i*rd_stride] = (rd_bias[0] +) sum_n v(i,n) * Y(i,n)
  float* rowdot; long long rd_stride; const float* rd_bias;
  int stream_c;
  int a_half;                             // A and Y are IEEE half in memory (lda, ldy in halves; K % 8 == 0): fp16-STORAGE scoring
  int c_half;                             // C is IEEE half in memory (ldc in halves; values rounded RNE at the store)
  int groups;                             // set by bx3_launch
  // side job of the loader waves: side_out(i,:) = side_scale[i] * side_in(i,:), side_cols (% 4 == 0) floats per row
  const float* side_in; float* side_out; const float* side_scale; long long side_ld; int side_cols;
  int side_half;                          // side_out is IEEE half in memory (side_ld = its row stride in halves too; side_in stays fp32)
};

template <int NTW>
struct Bx3Geom {
  static constexpr int ROWS = 128;                    // rows per workgroup: 4 compute waves x 32
  static constexpr int NTHR = 512;                    // 4 compute + 4 loader waves
  static constexpr int BLK = 3 * NTW;                 // 1-KB image blocks per k-step (one column group)
  static constexpr int A_OFF = BLK * 1024;
  static constexpr int SLOT = A_OFF + 8192;           // + 128 rows x 16 k of raw fp32 A
  static constexpr int NS = 6;
  static constexpr int LDC = 32 * NTW + 8;            // staging stride (4 rows apart = 32 banks apart)
  static constexpr int SIDE_OFF = NS * SLOT + 256;    // behind the ring and the 256-byte sink of the Y prefetch
  static constexpr size_t kLdsBytes =                 // (the ring is reused by the epilogue's staging)
      (size_t)SIDE_OFF + 20480 > (size_t)ROWS * LDC * 4 ? (size_t)SIDE_OFF + 20480 : (size_t)ROWS * LDC * 4;
  static_assert(kLdsBytes <= 160 * 1024, "LDS");
};


__host__ __device__ inline int bx3_ksteps(int K) { return (K + 15) / 16; }
inline int bx3_groups(int N) { return N > 160 ? 2 : 1; }
inline int bx3_ntw(int N) { const int g = bx3_groups(N); return ((N + 31) / 32 + g - 1) / g; }
inline size_t bx3_image_bytes(int N, int K) { return (size_t)bx3_ksteps(K) * 3 * bx3_groups(N) * bx3_ntw(N) * 1024; }

// x = h + m + l exactly; two values per call, packed as the MFMA wants them (low half = first)
__device__ __forceinline__ void bx3_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  bx3_f2 v = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx3_h2));
  v.x = x0 - __builtin_bit_cast(float, h << 16);
  v.y = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx3_h2));
  v.x = v.x - __builtin_bit_cast(float, m << 16);
  v.y = v.y - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx3_h2));
}

struct Bx3Frag { bx3_h8 h, m, l; };
__device__ __forceinline__ Bx3Frag bx3_split8(const pg_v4f r0, const pg_v4f r1) {
  unsigned hh[4], mm[4], ll[4];
  bx3_split2(r0[0], r0[1], hh[0], mm[0], ll[0]);
  bx3_split2(r0[2], r0[3], hh[1], mm[1], ll[1]);
  bx3_split2(r1[0], r1[1], hh[2], mm[2], ll[2]);
  bx3_split2(r1[2], r1[3], hh[3], mm[3], ll[3]);
  const bx3_u4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
  Bx3Frag f;
  f.h = __builtin_bit_cast(bx3_h8, h); f.m = __builtin_bit_cast(bx3_h8, m); f.l = __builtin_bit_cast(bx3_h8, l);
  return f;
}

// Eight halves (one 16-byte read) -> fragment: a half is h + m exactly, the third plane is zero (never multiplied)
typedef _Float16 bx3_hf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ Bx3Frag bx3_split8_half(const bx3_u4 raw) {
  const bx3_hf8 x = __builtin_bit_cast(bx3_hf8, raw);
  unsigned hh[4], mm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = (float)x[2 * i], x1 = (float)x[2 * i + 1];
    bx3_f2 v = {x0, x1};
    hh[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx3_h2));
    v.x = x0 - __builtin_bit_cast(float, hh[i] << 16);
    v.y = x1 - __builtin_bit_cast(float, hh[i] & 0xffff0000u);
    mm[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx3_h2));
  }
  const bx3_u4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, z = {0u, 0u, 0u, 0u};
  Bx3Frag f;
  f.h = __builtin_bit_cast(bx3_h8, h); f.m = __builtin_bit_cast(bx3_h8, m); f.l = __builtin_bit_cast(bx3_h8, z);
  return f;
}

// Eight floats that are widened halves -> the two planes that hold them exactly (no third level: 8 instead of 11 VALU a pair)
__device__ __forceinline__ void bx3_split8_two(const float* x, bx3_u4& h, bx3_u4& m) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bx3_f2 v = {x[2 * i], x[2 * i + 1]};
    const unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx3_h2));
    v.x = x[2 * i] - __builtin_bit_cast(float, hh << 16);
    v.y = x[2 * i + 1] - __builtin_bit_cast(float, hh & 0xffff0000u);
    h[i] = hh;
    m[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx3_h2));
  }
}

// B(k, n) = B[k*b_k + n*b_n]  ->  image [k-step][column tile][plane][lane]; one thread per (k-step, column tile, lane).
// Rider: zero[0 .. zero_n) = 0 (the row-dot output two column groups add into).
struct Bx3SplitArgs {
  const float* B; long long b_k, b_n; int K, N, ksteps, ntc; bx3_u4* img;
  float* zero; long long zero_stride; int zero_n;
};
__device__ __forceinline__ void bx3_split_b_body(const Bx3SplitArgs& a, int idx, int nthreads) {
  for (int z = idx; z < a.zero_n; z += nthreads) a.zero[(long long)z * a.zero_stride] = 0.f;
  if (idx >= a.ksteps * a.ntc * 64) return;
  const int lane = idx & 63, t = (idx >> 6) % a.ntc, s = (idx >> 6) / a.ntc;
  const int col = 32 * t + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {       // (clamped address + select: no divergent branch around the load)
    const bool ok = col < a.N && k0 + j < a.K;
    const float v = a.B[(long long)(ok ? k0 + j : 0) * a.b_k + (long long)(ok ? col : 0) * a.b_n];
    x[j] = ok ? v : 0.f;
  }
  const pg_v4f r0 = {x[0], x[1], x[2], x[3]}, r1 = {x[4], x[5], x[6], x[7]};
  const Bx3Frag f = bx3_split8(r0, r1);
  bx3_u4* o = a.img + ((size_t)(s * a.ntc + t) * 3) * 64 + lane;
  o[0] = __builtin_bit_cast(bx3_u4, f.h); o[64] = __builtin_bit_cast(bx3_u4, f.m); o[128] = __builtin_bit_cast(bx3_u4, f.l);
}
__global__ __launch_bounds__(256) void bx3_split_b_kernel(const Bx3SplitArgs a) {
  bx3_split_b_body(a, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

#ifdef MMS_BX3_STAMPS     // dev-only (tools/bx3bench.hip): per-workgroup wall-clock (100 MHz) / shader-clock stamps of wave 0
__device__ unsigned long long* bx3_stamp_buf = nullptr;
#define BX3_STAMP(k, v)                                                                        \
  do {                                                                                         \
    if (bx3_stamp_buf && threadIdx.x == 0) bx3_stamp_buf[(size_t)blockIdx.x * 16 + (k)] = (v); \
  } while (0)
#else
#define BX3_STAMP(k, v) do {} while (0)
#endif

// LDS-DMA with a scalar base and a 32-bit per-lane offset: no VALU per request (the loaders share their SIMD's vector
// issue with a compute wave; per-lane 64-bit addresses cost it some 130 VALU per k-step)
__device__ __forceinline__ void bx3_dma16s(const void* sbase, unsigned voff, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void bx3_dma4s(const void* sbase, unsigned voff, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

template <int N>
__device__ __forceinline__ void bx3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the loader's loop: PER DMAs per k-step, issued by issue(step)
template <int PER, int NS, class Issue, class Side, class SideOps>
__device__ __forceinline__ void bx3_loader_loop(int ksteps, Issue&& issue, Side&& side, SideOps&& side_ops) {
#if defined(MMS_BX3_ABLATE) && MMS_BX3_ABLATE == 4       // MMS_BX3_ABLATE / MMS_BX3TN_ABLATE: dev-only timing ablations (tools/bx3bench.hip)
  __builtin_amdgcn_s_barrier();
  return;
#endif
#if defined(MMS_BX3_ABLATE) && (MMS_BX3_ABLATE == 1 || MMS_BX3_ABLATE == 3)
  for (int j = 0; j <= ksteps; ++j) __builtin_amdgcn_s_barrier();
  return;
#endif
  for (int s = 0; s < NS - 1 && s < ksteps; ++s) issue(s);
  for (int j = 0; j <= ksteps; ++j) {
    // barrier j: steps j and j+1 have landed; issued so far: up to step j + NS - 2
    const int last = ksteps - 1;
    const int hi = j + NS - 2 < last ? j + NS - 2 : last, need = j + 1 < last ? j + 1 : last;
    const int out = hi - need;                       // steps allowed to be in flight: 0 .. NS - 3
    if (out >= 3) {
      const int so = side_ops();                      // side operations among the youngest 3 PER + so
      if (so >= 9) bx3_wait_vm<3 * PER + 9>();
      else if (so >= 6) bx3_wait_vm<3 * PER + 6>();
      else bx3_wait_vm<3 * PER>();
    }
    else if (out == 2) bx3_wait_vm<2 * PER>();
    else if (out == 1) bx3_wait_vm<PER>();
    else bx3_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    side(j);                                          // (its requests are older than this step's DMAs: see Bx3Side)
    if (j + NS - 1 < ksteps) issue(j + NS - 1);       // into the slot of step j - 1, which the barrier just released
  }
}

// The loader waves' side job (the backward's da = diag(dT) . (Q W), sim_matrix_layer.cpp:88 Trans, a streaming pass
// that used to be a launch of its own): the workgroup's 128 rows x its group's share of the side_cols / 4 float4
// columns, ONE float4 (and its row's scale) per loader lane and step, requested by LDS-DMA into a four-deep ring
// behind the operand ring and read back, scaled and stored three steps later.  (Through LDS, not registers: a load
// issued by hand lands whenever it lands, in a register the compiler may by then have copied or given away -- seen as a
// write fault.)  Three steps on, 3 PER younger DMAs have been issued behind a request, so the loop's counted wait
// covers it; the loop in turn is told how many side operations sit among the youngest (ops()), or its waits would reach
// into the DMAs it means to leave in flight (measured: +4 us on the main loop).  The stores are the compiler's.
struct Bx3Side {
  static constexpr int SLOT = 4096 + 1024;          // 256 lanes x (16 + 4) bytes per step
  static constexpr int BYTES = 4 * SLOT;
  const float* in; float* out; const float* scale; long long ld;
  int out_half;
  int row0, total, c0, nc;         // rows from row0, total = rows * nc float4, float4 columns [c0, c0 + nc)
  int lane, w;                     // loader wave w
  int cnt;                         // float4 per lane: indices 0 .. cnt-1 (index i is requested at step i)
  int n1, n2, n3;                  // side operations issued one, two and three steps ago
  unsigned lds;                    // LDS byte address of the side ring
  const unsigned char* ldsp;

  __device__ __forceinline__ void init(const Bx3Args& p, int r0, int grp, int ngroups, int rows_per_wg, int w_, int lane_,
                                       unsigned lds_byte, const unsigned char* lds_ptr) {
    in = p.side_in; out = p.side_out; scale = p.side_scale; ld = p.side_ld; out_half = p.side_half;
    const int SC = p.side_cols >> 2, half = (SC + 1) >> 1;
    c0 = ngroups == 2 ? grp * half : 0;
    nc = ngroups == 2 ? (grp ? SC - half : half) : SC;
    row0 = r0;
    const int rows = p.M - r0 < rows_per_wg ? p.M - r0 : rows_per_wg;
    total = rows * nc;
    w = w_; lane = lane_;
    cnt = in ? (total + 255) >> 8 : 0;
    n1 = n2 = n3 = 0;
    lds = lds_byte; ldsp = lds_ptr;
  }
  __device__ __forceinline__ void request(int i) {
    int f = i * 256 + w * 64 + lane;
    f = f < total ? f : total - 1;
    const int r = f / nc, cc = f - r * nc;
    const unsigned slot = lds + (unsigned)((i & 3) * SLOT);
    bx3_dma16s(in, (unsigned)(((long long)(row0 + r) * ld + 4 * (c0 + cc)) * 4), slot + (unsigned)(w * 1024));
    bx3_dma4s(scale, (unsigned)((row0 + r) * 4), slot + 4096u + (unsigned)(w * 256));
  }
  __device__ __forceinline__ void store(int i) {      // only behind a wait that covers request i
    const int f = i * 256 + w * 64 + lane;
    const unsigned char* slot = ldsp + (i & 3) * SLOT;
    const pg_v4f x = *reinterpret_cast<const pg_v4f*>(slot + w * 1024 + lane * 16);
    const float sc = *reinterpret_cast<const float*>(slot + 4096 + w * 256 + lane * 4);
    if (f < total) {
      const int r = f / nc, cc = f - r * nc;
      const pg_v4f v = sc * x;
      if (out_half) {
        typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
        const hf4 o = {(_Float16)(0.f + v[0]), (_Float16)(0.f + v[1]), (_Float16)(0.f + v[2]), (_Float16)(0.f + v[3])};
        __builtin_nontemporal_store(o, reinterpret_cast<hf4*>(reinterpret_cast<_Float16*>(out) + (long long)(row0 + r) * ld + 4 * (c0 + cc)));
      } else {
        __builtin_nontemporal_store(v, reinterpret_cast<pg_v4f*>(out + (long long)(row0 + r) * ld + 4 * (c0 + cc)));
      }
    }
  }
  __device__ __forceinline__ int ops() const { return n1 + n2 + n3; }     // among the youngest at the next wait
  __device__ __forceinline__ void step(int j) {
    int n = 0;
    asm volatile("" ::: "memory");
    if (j >= 3 && j - 3 < cnt) { store(j - 3); n += 1; }
    if (j < cnt) { request(j); n += 2; }
    n3 = n2; n2 = n1; n1 = n;
  }
  __device__ __forceinline__ void finish(int steps_done) {   // what the loop was too short for
    if (!cnt) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int loaded = steps_done < cnt ? steps_done : cnt;           // indices < loaded were requested
    for (int i = steps_done - 3 > 0 ? steps_done - 3 : 0; i < loaded; ++i) store(i);
    for (int i = loaded; i < cnt; ++i) {
      request(i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store(i);
    }
  }
};

// AH: the A operand (and the row dot's Y) are fp16 in memory.  A half is h + m exactly (8 + 3 significant bits), so the
// fragment has two planes and a product five partial products; its rows are half the bytes (4 A blocks per k-step).
template <int NTW, bool AH = false>
__global__ __launch_bounds__(512, 1) void bx3_kernel(const Bx3Args p) {
  using G = Bx3Geom<NTW>;
  static_assert(G::NS == 6, "bx3_loader_loop's waits assume three steps in flight");
  static_assert(Bx3Side::BYTES == 20480, "Bx3Geom::kLdsBytes reserves 20480 bytes for the side ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char bx3_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroups b and b + 8 (same XCD, dispatched together) are the two column groups of one row panel
  const int ngroups = p.groups;
  const int grp = ngroups == 2 ? (blockIdx.x >> 3) & 1 : 0;
  const int panel = ngroups == 2 ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : blockIdx.x;
  const int row0 = panel * G::ROWS;
  if (row0 >= p.M) return;                                // grid rounded up to a multiple of 16 (whole workgroups only)
  const int ksteps = bx3_ksteps(p.K);
  BX3_STAMP(0, __builtin_amdgcn_s_memrealtime());
  const unsigned lds0 = (unsigned)(uintptr_t)bx3_lds;     // LDS byte address of the ring

  bx3_f16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // the epilogue's per-row scales (two rows per lane: wave w writes rows 16 w .. 16 w + 15), requested now: in the
  // epilogue the load would be a memory round trip in front of the stores
  float scv[2] = {1.f, 1.f};
  if (p.rowscale) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int gi = row0 + 16 * wave + 8 * pass + (lane >> 3);
      scv[pass] = p.rowscale[gi < p.M ? gi : p.M - 1];
    }
  }

  if (wave >= 4) {
    // ------------------------------- loader waves -------------------------------
    const int w = wave - 4;
    Bx3Side sd;
    sd.init(p, row0, grp, ngroups, G::ROWS, w, lane, lds0 + (unsigned)G::SIDE_OFF, bx3_lds + G::SIDE_OFF);
    if (w < 3) {
      // plane w of the group's NTW image blocks of the step
      const char* const src0 = reinterpret_cast<const char*>(p.img) + ((size_t)grp * NTW * 3 + w) * 1024;
      const size_t step_bytes = (size_t)ngroups * NTW * 3 * 1024;
      const unsigned voff = (unsigned)lane * 16u;
      bx3_loader_loop<NTW, G::NS>(ksteps, [&](int s) {
        const unsigned slot = lds0 + (unsigned)((s % G::NS) * G::SLOT + w * 1024);
        const char* src = src0 + (size_t)s * step_bytes;
#pragma unroll
        for (int t = 0; t < NTW; ++t) bx3_dma16s(src + t * 3072, voff, slot + (unsigned)(t * 3072));
      }, [&](int j) { sd.step(j); }, [&]() { return sd.ops(); });
    } else if (AH) {
      // the panel's raw A as halves: 4 blocks = row tiles; lane fetches row 32 wr + l % 32, the 8 halves k = 16 s + 8 (l / 32)
      const _Float16* const abase_h = reinterpret_cast<const _Float16*>(p.A) + (long long)row0 * p.lda;
      unsigned aoffh[4], aoffh_t[4];
      const int akoff = 8 * (lane >> 5);
      const int klast = 16 * (ksteps - 1) + akoff;
#pragma unroll
      for (int wr = 0; wr < 4; ++wr) {
        int r = 32 * wr + (lane & 31);
        r = row0 + r < p.M ? r : p.M - 1 - row0;
        aoffh[wr] = (unsigned)(((long long)r * p.lda + akoff) * 2);
        const int k0 = klast <= p.K - 8 ? klast : p.K - 8;            // (K % 8 == 0: a chunk is whole or past the end)
        aoffh_t[wr] = (unsigned)(((long long)r * p.lda + k0 - 16 * (ksteps - 1)) * 2);
      }
      constexpr int LPR = (32 * NTW * 2 + 127) / 128 + 1;
      const char* const ybase = reinterpret_cast<const char*>(p.Y ? (const void*)p.Y : (const void*)p.A);
      const long long ystride = p.Y ? p.ldy * 2 : 0;
      const long long ymax = p.Y ? ((long long)(p.M - 1) * p.ldy + p.N) * 2 - 4 : 0;
      bx3_loader_loop<5, G::NS>(ksteps, [&](int s) {
        {
          int idx = (s - 3) * 64 + lane;
          idx = idx > 0 ? idx : 0;
          idx = idx < G::ROWS * LPR ? idx : G::ROWS * LPR - 1;
          long long off = (long long)(row0 + idx / LPR) * ystride + (long long)(grp * 32 * NTW) * 2 + (idx % LPR) * 128;
          off = off < ymax ? off : ymax;
          off &= ~3LL;
          bx3_dma4s(ybase, (unsigned)off, lds0 + (unsigned)(G::NS * G::SLOT));
        }
        const unsigned slot = lds0 + (unsigned)((s % G::NS) * G::SLOT + G::A_OFF);
        const _Float16* sb = abase_h + 16 * s;
#pragma unroll
        for (int wr = 0; wr < 4; ++wr)
          bx3_dma16s(sb, s < ksteps - 1 ? aoffh[wr] : aoffh_t[wr], slot + (unsigned)(wr * 1024));
      }, [&](int j) { sd.step(j); }, [&]() { return sd.ops(); });
    } else {
      // the panel's raw A: 8 blocks = (row tile wr, k quad q): lane fetches row 32 wr + l % 32, k = 16 s + 8 (l / 32) + 4 q.
      // Per-lane byte offsets from the panel's first row, fixed for the whole loop (rows past M: the last row); the
      // last step's are clamped to the row's end (any finite values: the image holds zeros there).
      const float* const abase_g = p.A + (long long)row0 * p.lda;
      unsigned aoff[4], aoff_t0[4], aoff_t1[4];
      const int akoff = 8 * (lane >> 5);
      const int klast = 16 * (ksteps - 1) + akoff;
#pragma unroll
      for (int wr = 0; wr < 4; ++wr) {
        int r = 32 * wr + (lane & 31);
        r = row0 + r < p.M ? r : p.M - 1 - row0;
        aoff[wr] = (unsigned)(((long long)r * p.lda + akoff) * 4);
        const int k0 = klast <= p.K - 4 ? klast : p.K - 4, k1 = klast + 4 <= p.K - 4 ? klast + 4 : p.K - 4;
        aoff_t0[wr] = (unsigned)(((long long)r * p.lda + k0 - 16 * (ksteps - 1)) * 4);
        aoff_t1[wr] = (unsigned)(((long long)r * p.lda + k1 - 16 * (ksteps - 1)) * 4);
      }
      // The row dot's operand Y (this group's 128 x 32 NTW tile) is read by the epilogue, when nothing else is in flight;
      // one dword per 128-byte line requested here, beside the first steps' DMAs, has it waiting in L2 / the Infinity
      // Cache by then.  One request per step and lane whatever the step (a fixed count for the loop's counted waits).
      constexpr int LPR = (32 * NTW * 4 + 127) / 128 + 1;     // lines a row's segment can touch
      const char* const ybase = reinterpret_cast<const char*>(p.Y ? p.Y : p.A);
      const long long ystride = p.Y ? p.ldy * 4 : 0;
      const long long ymax = p.Y ? ((long long)(p.M - 1) * p.ldy + p.N) * 4 - 4 : 0;
      bx3_loader_loop<9, G::NS>(ksteps, [&](int s) {
        {
          int idx = (s - 3) * 64 + lane;            // (from step 3 on: in front of the first steps they delay barrier 0)
          idx = idx > 0 ? idx : 0;
          idx = idx < G::ROWS * LPR ? idx : G::ROWS * LPR - 1;
          long long off = (long long)(row0 + idx / LPR) * ystride + (long long)(grp * 32 * NTW) * 4 + (idx % LPR) * 128;
          off = off < ymax ? off : ymax;
          // (an LDS-DMA into a 256-byte sink behind the ring: a load into a VGPR would land, later, in a register the
          //  compiler has long since given to something else)
          bx3_dma4s(ybase, (unsigned)off, lds0 + (unsigned)(G::NS * G::SLOT));
        }
        const unsigned slot = lds0 + (unsigned)((s % G::NS) * G::SLOT + G::A_OFF);
        const float* sb = abase_g + 16 * s;
        if (s < ksteps - 1) {
#pragma unroll
          for (int wr = 0; wr < 4; ++wr) {
            bx3_dma16s(sb, aoff[wr], slot + (unsigned)(wr * 2048));
            bx3_dma16s(sb + 4, aoff[wr], slot + (unsigned)(wr * 2048 + 1024));
          }
        } else {
#pragma unroll
          for (int wr = 0; wr < 4; ++wr) {
            bx3_dma16s(sb, aoff_t0[wr], slot + (unsigned)(wr * 2048));
            bx3_dma16s(sb, aoff_t1[wr], slot + (unsigned)(wr * 2048 + 1024));
          }
        }
      }, [&](int j) { sd.step(j); }, [&]() { return sd.ops(); });
    }
    sd.finish(ksteps + 1);
  } else {
    // ------------------------------- compute waves ------------------------------
    const unsigned char* const abase = bx3_lds + G::A_OFF + wave * (AH ? 1024 : 2048) + lane * 16;
    auto read_a = [&](int slot_, pg_v4f& x0, pg_v4f& x1) {       // the raw A fragment of a slot (AH: one read, 8 halves)
      x0 = *reinterpret_cast<const pg_v4f*>(abase + slot_ * G::SLOT);
      if (!AH) x1 = *reinterpret_cast<const pg_v4f*>(abase + slot_ * G::SLOT + 1024);
    };
    auto split_a = [&](const pg_v4f& x0, const pg_v4f& x1) {
      return AH ? bx3_split8_half(__builtin_bit_cast(bx3_u4, x0)) : bx3_split8(x0, x1);
    };
    const unsigned char* const bbase = bx3_lds + lane * 16;
    auto read_b = [&](int slot, int t, bx3_h8& h, bx3_h8& m, bx3_h8& l) {
      const bx3_u4* q = reinterpret_cast<const bx3_u4*>(bbase + slot * G::SLOT + t * 3072);
      h = __builtin_bit_cast(bx3_h8, q[0]); m = __builtin_bit_cast(bx3_h8, q[64]); l = __builtin_bit_cast(bx3_h8, q[128]);
    };
    __builtin_amdgcn_s_barrier();                    // barrier 0: steps 0 and 1 have landed
    asm volatile("" ::: "memory");
    BX3_STAMP(1, __builtin_amdgcn_s_memrealtime());
    BX3_STAMP(4, __builtin_amdgcn_s_memtime());
#if defined(MMS_BX3_ABLATE) && (MMS_BX3_ABLATE == 2 || MMS_BX3_ABLATE == 3)
    for (int s = 0; s < ksteps; ++s) __builtin_amdgcn_s_barrier();
    if (false)
#endif
    {
    pg_v4f r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
    read_a(0, r0, r1);
    Bx3Frag a = split_a(r0, r1);
    bx3_h8 bh[NTW], bm[NTW], bl[NTW];
    // (read in tile order, pinned: the counted wait at the top of the loop is the stricter of the two ways in, and with
    //  tile 0 read last here it would be lgkmcnt(0) on every step -- the LDS latency of the step's last reads, exposed)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      __builtin_amdgcn_sched_barrier(0);
      read_b(0, t, bh[t], bm[t], bl[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
    for (int s = 0; s < ksteps; ++s) {
      const int nslot = slot + 1 < G::NS ? slot + 1 : 0;
      // Tile t: its six MFMAs, then the NEXT step's fragments of the same tile are requested into the registers the
      // MFMAs have just read -- a whole step (about 960 cycles) ahead of their use, three reads per tile-time instead
      // of the burst of fifteen per wave at the top of the step the scheduler would make of it (four waves' 60 KB
      // queueing on the LDS while the first MFMAs wait).  The next step's raw A fragment is requested behind tile 0 and
      // split in tile 1's region, eight VALU behind each MFMA.  (Past the last step the reads fetch a landed slot's
      // stale bytes that nothing uses.)
      Bx3Frag an;
      if (NTW == 1) read_a(nslot, r0, r1);              // one tile: there is no "tile 1" to split behind
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (!AH) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, bh[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, bl[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, bm[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, bh[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, bm[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, bh[t], acc[t], 0, 0, 0);
        if (t == (NTW > 1 ? 1 : 0)) {
          an = split_a(r0, r1);
#pragma unroll
          for (int g = 0; g < (AH ? 5 : 6); ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        read_b(nslot, t, bh[t], bm[t], bl[t]);
        if (t == 0 && NTW > 1) read_a(nslot, r0, r1);
      }
      __builtin_amdgcn_sched_barrier(0);
      a = an;
      slot = nslot;
      asm volatile("" ::: "memory");
#if !defined(MMS_BX3_ABLATE) || MMS_BX3_ABLATE != 4
      __builtin_amdgcn_s_barrier();                  // barrier s + 1: this step's slot is free; steps s + 1, s + 2 have landed
#endif
      asm volatile("" ::: "memory");
    }
    }
  }

  // ---------------------------------- epilogue ----------------------------------
  // (the loop's last barrier: every ring read is done and every DMA has landed)
  BX3_STAMP(2, __builtin_amdgcn_s_memrealtime());
  BX3_STAMP(5, __builtin_amdgcn_s_memtime());
  // Accumulators through LDS (the ring is free: the loop's last barrier), then all eight waves write 16-byte row
  // segments.  Register i of a 32x32 tile holds row 8 (i / 4) + 4 (lane / 32) + i % 4 at column lane % 32.
  float* stage = reinterpret_cast<float*>(bx3_lds);
  if (wave < 4) {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = 32 * wave + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        stage[r * G::LDC + 32 * t + (lane & 31)] = acc[t][i];
      }
  }
  __syncthreads();
  // wave w: rows 16 w .. 16 w + 15, eight rows per pass: lane -> (row lane / 8, float4 columns lane % 8 + 8 j)
  const int col0 = grp * 32 * NTW;                       // the group's first column
  const float bias = (p.rd_bias && grp == 0) ? p.rd_bias[0] : 0.f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = 16 * wave + 8 * pass + (lane >> 3), gi = row0 + r;
    const bool rok = gi < p.M;
    const float sc = scv[pass];
    pg_v4f y[NTW];
    if (p.Y) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int c = 4 * ((lane & 7) + 8 * j);
        const bool ok = rok && col0 + c < p.N;
        if (AH) {
          typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
          const hf4 yh = ok ? *reinterpret_cast<const hf4*>(reinterpret_cast<const _Float16*>(p.Y) + (long long)gi * p.ldy + col0 + c)
                            : hf4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          y[j] = pg_v4f{(float)yh[0], (float)yh[1], (float)yh[2], (float)yh[3]};
        } else {
          y[j] = ok ? *reinterpret_cast<const pg_v4f*>(p.Y + (long long)gi * p.ldy + col0 + c) : pg_v4f{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int c = 4 * ((lane & 7) + 8 * j);            // column inside the group
      if (rok && col0 + c < p.N) {
        pg_v4f v = *reinterpret_cast<const pg_v4f*>(stage + r * G::LDC + c);
        if (p.rowscale) v *= sc;
        if (p.Y) { part += v[0] * y[j][0]; part += v[1] * y[j][1]; part += v[2] * y[j][2]; part += v[3] * y[j][3]; }
        if (p.C) {
          if (AH && p.c_half) {                            // (fp16-storage family only: the flag is not read otherwise)
            typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
            const hf4 o = {(_Float16)(0.f + v[0]), (_Float16)(0.f + v[1]), (_Float16)(0.f + v[2]), (_Float16)(0.f + v[3])};
            hf4* dst = reinterpret_cast<hf4*>(reinterpret_cast<_Float16*>(p.C) + (long long)gi * p.ldc + col0 + c);
            if (p.stream_c) __builtin_nontemporal_store(o, dst);
            else *dst = o;
          } else {
            pg_v4f* dst = reinterpret_cast<pg_v4f*>(p.C + (long long)gi * p.ldc + col0 + c);
            if (p.stream_c) __builtin_nontemporal_store(v, dst);
            else *dst = v;
          }
        }
      }
    }
    if (p.rowdot) {
      part = dpp_add<0xB1, 0xf>(part);                   // the row's eight lanes: quad_perm [1,0,3,2], [2,3,0,1],
      part = dpp_add<0x4E, 0xf>(part);
      part = dpp_add<0x141, 0xf>(part);                  // row_half_mirror
      if (rok && (lane & 7) == 0) {
        float* dst = p.rowdot + (long long)gi * p.rd_stride;
        if (ngroups == 2) atomicAdd(dst, bias + part);   // onto the zero bx3_split_b wrote: two addends, order-free
        else *dst = p.rd_bias ? bias + part : part;
      }
    }
  }
  BX3_STAMP(3, __builtin_amdgcn_s_memrealtime());
}

inline bool bx3_eligible(const Bx3Args& p) {
  if (p.a_half)                                        // fp16-storage scoring: A and Y halves, 16-byte chunks of 8 k
    return p.M >= 1 && p.N >= 4 && p.N <= 320 && p.K >= 8 && (p.N & 3) == 0 && (p.K & 7) == 0 && (p.lda & 7) == 0 &&
           aligned16(p.A) && (!p.C || ((p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & (p.c_half ? 7u : 15u)) == 0)) &&
           (!p.side_in || ((p.side_cols & 3) == 0 && p.side_cols >= 8 && (p.side_ld & 3) == 0 && aligned16(p.side_in) && p.side_scale &&
                           (reinterpret_cast<uintptr_t>(p.side_out) & (p.side_half ? 7u : 15u)) == 0 &&
                           (long long)p.M * p.side_ld < (1LL << 30))) &&
           (!p.Y || ((p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Y) & 7u) == 0 && (long long)p.M * p.ldy < (1LL << 30))) &&
           (long long)p.M * p.lda < (1LL << 30);
  return p.M >= 1 && p.N >= 4 && p.N <= 320 && p.K >= 4 && (p.N & 3) == 0 && (p.K & 3) == 0 && (p.lda & 3) == 0 &&
         aligned16(p.A) && (!p.C || ((p.ldc & 3) == 0 && aligned16(p.C))) &&
         (!p.Y || ((p.ldy & 3) == 0 && aligned16(p.Y))) && (long long)p.M * p.lda < (1LL << 30) &&
         (!p.Y || (long long)p.M * p.ldy < (1LL << 30)) &&
         (!p.side_in || ((p.side_cols & 3) == 0 && p.side_cols >= 8 && (p.side_ld & 3) == 0 && aligned16(p.side_in) &&
                         aligned16(p.side_out) && p.side_scale && (long long)p.M * p.side_ld < (1LL << 30)));
}

// zero / zero_stride / zero_n: the row-dot output of a two-group product (null / 0 otherwise)
inline Bx3SplitArgs bx3_split_args(const float* B, long long b_k, long long b_n, int K, int N, bx3_u4* img,
                                   float* zero = nullptr, long long zero_stride = 1, int zero_n = 0) {
  Bx3SplitArgs a{};
  a.B = B; a.b_k = b_k; a.b_n = b_n; a.K = K; a.N = N; a.ksteps = bx3_ksteps(K); a.ntc = bx3_groups(N) * bx3_ntw(N); a.img = img;
  a.zero = zero; a.zero_stride = zero_stride; a.zero_n = zero ? zero_n : 0;
  return a;
}
inline unsigned bx3_split_blocks(const Bx3SplitArgs& a) { return (unsigned)((a.ksteps * a.ntc * 64 + 255) / 256); }
inline void bx3_split_b(const float* B, long long b_k, long long b_n, int K, int N, bx3_u4* img, hipStream_t s,
                        float* zero = nullptr, long long zero_stride = 1, int zero_n = 0) {
  const Bx3SplitArgs a = bx3_split_args(B, b_k, b_n, K, N, img, zero, zero_stride, zero_n);
  hipLaunchKernelGGL(bx3_split_b_kernel, dim3(bx3_split_blocks(a)), dim3(256), 0, s, a);
}

template <int NTW, bool AH = false>
inline void bx3_launch_t(const Bx3Args& p, hipStream_t s) {
  using G = Bx3Geom<NTW>;
  static bool once = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&bx3_kernel<NTW, AH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)G::kLdsBytes) == hipSuccess;
  }();
  (void)once;
  const unsigned panels = (unsigned)((p.M + G::ROWS - 1) / G::ROWS);
  const unsigned grid = p.groups == 2 ? ((panels + 7) / 8) * 16 : panels;
  hipLaunchKernelGGL((bx3_kernel<NTW, AH>), dim3(grid), dim3(G::NTHR), G::kLdsBytes, s, p);
}

inline void bx3_launch(Bx3Args p, hipStream_t s) {
  p.groups = bx3_groups(p.N);
  if (p.a_half) {
    switch (bx3_ntw(p.N)) {
      case 1: bx3_launch_t<1, true>(p, s); break;
      case 2: bx3_launch_t<2, true>(p, s); break;
      case 3: bx3_launch_t<3, true>(p, s); break;
      case 4: bx3_launch_t<4, true>(p, s); break;
      default: bx3_launch_t<5, true>(p, s); break;
    }
    return;
  }
  switch (bx3_ntw(p.N)) {
    case 1: bx3_launch_t<1>(p, s); break;
    case 2: bx3_launch_t<2>(p, s); break;
    case 3: bx3_launch_t<3>(p, s); break;
    case 4: bx3_launch_t<4>(p, s); break;
    default: bx3_launch_t<5>(p, s); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The weight gradient dW = Q^T diag(dT) A (sim_matrix_layer.cpp:73-80): a SMALL output (K1 x K2 <= 320 x 320) summed
// over ALL pairs -- both operands are activations, each read once, and the contraction index (the pair) is the slow
// axis of both.  Split-K over pairs; a workgroup owns one 160 x 160 quadrant of the output for one chunk of pairs:
//   * 4 compute waves, each an 80 x 80 block = 5 x 5 accumulator tiles of v_mfma_f32_16x16x32_bf16 (k-step: 32 pairs);
//   * 4 splitter waves build the operand image in LDS: a tile is 16 columns x 32 pairs; lane (c = l % 16, g = l / 16)
//     loads the 8 values X[n0 + 8 g + j][col0 + c] (each load: four 64-byte row segments), multiplies the A side by
//     dT[n] (the reference forms U = diag(dT) A in fp32 first: same rounding), splits them (x = h + m + l) and writes
//     its three 16-byte fragments -- exactly the MFMA's operand layout, so the transposition costs nothing;
//   * two image buffers of 20 tiles x 3 planes x 1 KB: the splitters fill one while the compute waves read the other,
//     one workgroup barrier per step; the splitters' global loads run one step further ahead in registers;
//   * the quadrant's partial sums go to slab `chunk` of the split-K buffer; splitk_reduce_kernel adds the slabs in
//     chunk order (deterministic) into dW.
// The four quadrants of a chunk run on one XCD (they share the chunk's rows in its L2).
struct Bx3TnArgs {
  int M, N, K;                           // C (M x N) = sum over k < K of A[k][i] * (kscale[k] * B[k][j])
  const float* A; long long lda;
  const float* B; long long ldb;
  const float* kscale;                   // K values, or null
  int ab_half;                           // A and B are IEEE half in memory (lda, ldb in halves): the fp16-storage family
  float* C; long long c_ks;              // slab ks at C + ks * c_ks, row stride N
  int kchunk, nchunks;                   // pairs per chunk (a multiple of 32), chunks
};

constexpr int BX3TN_Q = 160;             // quadrant edge
constexpr int BX3TN_BUF = 20 * 3 * 1024; // one image buffer
constexpr size_t kBx3TnLds = 2 * (size_t)BX3TN_BUF;

inline int bx3_tn_pick_chunks(int K, int quads, int* kchunk) {
  int want = 256 / (quads > 0 ? quads : 1);
  const int maxs = (K + 63) / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  int chunk = (K + want - 1) / want;
  chunk = (chunk + 31) / 32 * 32;
  *kchunk = chunk;
  return (K + chunk - 1) / chunk;
}
inline int bx3_tn_quads(int M, int N) { return ((M + BX3TN_Q - 1) / BX3TN_Q) * ((N + BX3TN_Q - 1) / BX3TN_Q); }

// HALF: A and B are IEEE half in memory (a template parameter: as a run-time flag the second load form cost the fp32 kernel
// 72 VGPRs and 1 KB of scratch -- cfg 3's dW 34 -> 48 us)
template <bool HALF>
__global__ __launch_bounds__(512, 1) void bx3_tn_kernel(const Bx3TnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qcols = (p.N + BX3TN_Q - 1) / BX3TN_Q, quads = qcols * ((p.M + BX3TN_Q - 1) / BX3TN_Q);
  // workgroup b: XCD b % 8 takes whole chunks: its workgroups [slot * quads, ..) are the quadrants of one chunk
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = (slot / quads) * 8 + xcd, quad = slot % quads;
  if (chunk >= p.nchunks) return;
  const int i0 = (quad / qcols) * BX3TN_Q, j0 = (quad % qcols) * BX3TN_Q;
  const int n0 = chunk * p.kchunk;
  const int nend = n0 + p.kchunk < p.K ? n0 + p.kchunk : p.K;
  const int steps = (nend - n0 + 31) >> 5;

  if (wave >= 4) {
    // ------------------------------ splitter waves ------------------------------
    // Units of 32 columns x 16 pairs (so that a load covers 128 contiguous bytes of a row: with 16-column units the
    // 64-byte segments of 1,200-byte rows straddled two sectors each): unit U = 2 P + h, P < 5: columns [i0 + 32 P, +32)
    // of A, else [j0 + 32 (P - 5), +32) of B; h = which 16 pairs of the step.  Lane (c32 = l % 32, g2 = l / 32) holds
    // column c32's pairs 16 h + 8 g2 + j -- the fragment of 16-column tile 2 P + c32 / 16, lane slot (c32 % 16, k group
    // 2 h + g2).  Wave w takes units w, w + 4, ..., w + 16: all of k half h = w & 1.
    const int w = wave - 4, c32 = lane & 31, g2 = lane >> 5, hk = w & 1;
    const int g = 2 * hk + g2;          // the lane's k group of the 32-pair step
    const float* xb[5];                 // the unit's operand (wave-uniform)
    long long ld[5];
    int colc[5];
    unsigned ldsoff[5];                 // byte offset of the lane's 16 bytes inside an image buffer (plane h; m, l: + 1024, 2048)
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int P = (w + 4 * u) >> 1;
      const bool isb = P >= 5;
      int col = isb ? j0 + 32 * (P - 5) + c32 : i0 + 32 * P + c32;
      const int lim = isb ? p.N : p.M;
      col = col < lim ? col : lim - 1;                 // columns past the edge: a valid column's values (outputs never stored)
      xb[u] = isb ? p.B : p.A;
      ld[u] = isb ? p.ldb : p.lda;
      colc[u] = col;
      const int t = 2 * P + (c32 >> 4);                // (P >= 5: tiles 10..19 are the B side, as before)
      ldsoff[u] = (unsigned)(t * 3072 + ((c32 & 15) + 16 * g) * 16);
    }
    // two register stages: step s + 2 is requested BEFORE step s + 1 is split, a whole step ahead of its use
    float raw[2][5][8], ksc[2][8];
    const float* const ksp = p.kscale;                        // null: no scale
    auto fetch = [&](auto stage, int s) {
      constexpr int R = decltype(stage)::value;
#if defined(MMS_BX3TN_ABLATE) && MMS_BX3TN_ABLATE == 3
      return;
#endif
      const int nb = __builtin_amdgcn_readfirstlane(n0 + 32 * s);   // wave-uniform by construction; said so, the row bases are scalar
      if (nb + 32 <= nend) {
        // a whole step: one 64-bit address per tile, then row to row by one add -- the splitters share the vector pipe
        // with a compute wave's MFMAs, and a full address per load (64-bit multiply-adds) cost more than the split itself
        if (ksp) {
          const float* kp = ksp + (nb + 8 * g);
#pragma unroll
          for (int j = 0; j < 8; ++j) ksc[R][j] = kp[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) ksc[R][j] = 1.f;
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const float* pu = xb[u] + (long long)(nb + 8 * g) * ld[u] + colc[u];
          if (HALF) {
            const _Float16* ph = reinterpret_cast<const _Float16*>(xb[u]) + (long long)(nb + 8 * g) * ld[u] + colc[u];
#pragma unroll
            for (int j = 0; j < 8; ++j) { raw[R][u][j] = (float)*ph; ph += ld[u]; }
            continue;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) { raw[R][u][j] = *pu; pu += ld[u]; }
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // (branch-free: a load under a per-lane condition becomes a divergent branch with its own wait)
        int n = nb + 8 * g + j;
        const bool ok = n < nend;
        n = ok ? n : p.K - 1;
        const float kv = ksp ? ksp[n] : 1.f;                    // (wave-uniform choice: no divergent branch)
        ksc[R][j] = ok ? kv : 0.f;                              // pairs past the chunk's end add zero
#pragma unroll
        for (int u = 0; u < 5; ++u)
          raw[R][u][j] = HALF ? (float)reinterpret_cast<const _Float16*>(xb[u])[(long long)n * ld[u] + colc[u]]
                                   : xb[u][(long long)n * ld[u] + colc[u]];
      }
    };
    auto emit = [&](auto stage, int buf) {
      constexpr int R = decltype(stage)::value;
#if defined(MMS_BX3TN_ABLATE) && MMS_BX3TN_ABLATE == 1
      return;
#endif
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const bool isb = ((w + 4 * u) >> 1) >= 5;           // wave-uniform
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = isb ? raw[R][u][j] * ksc[R][j] : raw[R][u][j];
        bx3_u4* o = reinterpret_cast<bx3_u4*>(tn_lds + buf * BX3TN_BUF + ldsoff[u]);
        if (HALF && !isb) {                               // the A side of the half form: widened halves, two planes (the third is
          bx3_u4 h2, m2;                                    // never read: the compute waves skip its product)
          bx3_split8_two(x, h2, m2);
          o[0] = h2; o[64] = m2;
          continue;
        }
        const pg_v4f r0 = {x[0], x[1], x[2], x[3]}, r1 = {x[4], x[5], x[6], x[7]};
        const Bx3Frag f = bx3_split8(r0, r1);
        o[0] = __builtin_bit_cast(bx3_u4, f.h); o[64] = __builtin_bit_cast(bx3_u4, f.m); o[128] = __builtin_bit_cast(bx3_u4, f.l);
      }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    fetch(S0{}, 0);
    if (steps > 1) fetch(S1{}, 1);
    emit(S0{}, 0);
    // step s: request step s + 2 into stage s & 1 (split one step ago), then split stage (s + 1) & 1 into buffer (s + 1) & 1
    for (int s = 0; s < steps; ++s) {
      __syncthreads();                                      // buffer s & 1 is complete; the other one is free
      if (s & 1) {
        if (s + 2 < steps) fetch(S1{}, s + 2);
        if (s + 1 < steps) emit(S0{}, 0);
      } else {
        if (s + 2 < steps) fetch(S0{}, s + 2);
        if (s + 1 < steps) emit(S1{}, 1);
      }
    }
    return;
  }
  // ------------------------------- compute waves -------------------------------
  const int wr = wave >> 1, wc = wave & 1;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 acc[5][5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < steps; ++s) {
    __syncthreads();
    const bx3_u4* img = reinterpret_cast<const bx3_u4*>(tn_lds + (s & 1) * BX3TN_BUF) + lane;
#if defined(MMS_BX3TN_ABLATE) && MMS_BX3TN_ABLATE == 2
    continue;
#endif
    bx3_h8 ah[5], am[5], al[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const bx3_u4* q = img + (5 * wr + a) * 192;
      ah[a] = __builtin_bit_cast(bx3_h8, q[0]); am[a] = __builtin_bit_cast(bx3_h8, q[64]);
      if (!HALF) al[a] = __builtin_bit_cast(bx3_h8, q[128]);
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const bx3_u4* q = img + (10 + 5 * wc + b) * 192;
      const bx3_h8 bh = __builtin_bit_cast(bx3_h8, q[0]), bm = __builtin_bit_cast(bx3_h8, q[64]), bl = __builtin_bit_cast(bx3_h8, q[128]);
#pragma unroll
      for (int a = 0; a < 5; ++a) {
        if (!HALF) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[a], bh, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bl, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[a], bm, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[a], bh, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bm, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
      }
    }
  }
  // register r of a 16x16 tile: row 4 (lane / 16) + r, column lane % 16
  float* slab = p.C + (long long)chunk * p.c_ks;
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * (5 * wr + a) + 4 * (lane >> 4) + r, j = j0 + 16 * (5 * wc + b) + (lane & 15);
        if (i < p.M && j < p.N) slab[(long long)i * p.N + j] = acc[a][b][r];
      }
}

inline bool bx3_tn_eligible(const Bx3TnArgs& p) {
  return p.M >= 1 && p.N >= 1 && p.M <= 2 * BX3TN_Q && p.N <= 2 * BX3TN_Q && p.K >= 1 && p.kchunk > 0 && (p.kchunk & 31) == 0 &&
         (long long)p.K * p.lda < (1LL << 31) && (long long)p.K * p.ldb < (1LL << 31) && 32 * p.lda < (1LL << 28) &&
         32 * p.ldb < (1LL << 28);
}

template <bool HALF>
inline void bx3_tn_launch_t(const Bx3TnArgs& p, hipStream_t s) {
  static bool once = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&bx3_tn_kernel<HALF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)kBx3TnLds) == hipSuccess;
  }();
  (void)once;
  const int quads = bx3_tn_quads(p.M, p.N);
  const unsigned grid = (unsigned)(((p.nchunks + 7) / 8) * 8 * quads);
  hipLaunchKernelGGL(bx3_tn_kernel<HALF>, dim3(grid), dim3(512), kBx3TnLds, s, p);
}
inline void bx3_tn_launch(const Bx3TnArgs& p, hipStream_t s) {
  if (p.ab_half) bx3_tn_launch_t<true>(p, s);
  else bx3_tn_launch_t<false>(p, s);
}

}  // namespace mms
#endif  // MMS_BX3_GEMM_H_
