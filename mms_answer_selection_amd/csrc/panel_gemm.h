// csrc/panel_gemm.h -- "panel" fp32-MFMA GEMM for the shapes the learned-metric paths actually have:
// a TALL operand (all pairs: tens of thousands of rows) times a SMALL square weight (D x D, D <= 304),
// and the transposed-tall x tall product of the weight gradient.  Included by bilinear.hip.
//
//   C[M x N] = A[M x K] . B[K x N],   B(k,n) = B[k*ldb + n],   N <= 16 * NT  (NT = 19: the 300-d width)
//
// A 64 x 64 output tiling of these products (gemm32_fast_kernel) re-stages every A row panel once per
// column tile, gives each wave ONE accumulator tile (one MFMA chain) and needs 1,280 workgroups whose
// prologues, barriers and store epilogues the MFMA pipe sits out: 47 % MFMA-busy at cfg 3.  Here a
// workgroup owns a 64-row PANEL and ALL N columns, one workgroup per CU:
//   * wave w owns rows [16w, 16w+16) x NT column tiles of 16: NT independent accumulators of
//     v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate, bit-equal to an fmaf chain);
//   * MFMA step (h, i) of a 32-deep k-tile multiplies the k values {k0 + 16h + 4g + i : g = lane / 16}:
//     a permutation of the k order inside a tile, applied to both operands alike, chosen so that a lane's
//     four consecutive steps read four CONSECUTIVE k of its row (one 16-byte LDS read for A);
//   * both operand tiles reach LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write pass) into a
//     THREE-stage ring: the tile computed now, one landed or landing, one just issued.  Every wave issues the
//     same number of DMA instructions per tile (slots past the tile's end fetch a clamped address into
//     padding), so ONE counted `s_waitcnt vmcnt(per-tile count)` + a raw s_barrier per k-tile of 152 MFMAs
//     per wave (~2.5 us) retires exactly the tile that is about to be read and leaves the next one in
//     flight -- with two stages and a full drain at each barrier a fifth of the kernel was spent waiting
//     for memory (profiles/r02_panel_gemm.txt).  The DMAs are inline asm: hipcc neither counts nor drains them;
//   * the DMAs are issued by four LOADER waves (waves 4-7, one per SIMD beside a compute wave): an LDS-DMA
//     instruction holds its wave's issue for 60-180 cycles, and 14 of them per tile in front of the compute
//     waves' MFMAs left the matrix pipe idle for a ninth of the kernel.  Loader and compute waves meet at
//     the per-tile barrier only.  The loaders' spare time and registers carry the memory-bound side jobs of
//     the callers: prefetching the row-dot operand Y of the SimMatrix forward, and the streaming pass
//     da = diag(dT) . (Q W) of its backward (`side_*`), which used to be a launch of its own;
//   * LDS images: B [32][16NT+4] (row stride = 4 mod 8 floats: the 4-row spread of a ds_read_b32 operand
//     read hits 32 distinct banks); A_KC: A [64][40] (stride 40: each 16-lane group of a ds_read_b128 covers
//     64 distinct banks); !A_KC: A^T [32][68] plus the 32 kscale values of the tile;
//   * the epilogue passes the accumulators through LDS so that global accesses are 256-byte row segments,
//     and carries what used to be separate launches: the per-row scale (rowscale_kernel) and the row dot
//     product of the SimMatrix forward (rowdot_kernel);
//   * A_KC = false reads A "transposed" (A(i,k) = A[k*lda + i], scaled by kscale[k]): the weight
//     gradient dW = Q^T diag(dT) A as a split-K product whose chunks are placed so that the row blocks
//     of one k-chunk run on ONE XCD (they stream the same rows of B through that XCD's L2).
// Reference semantics are those of the callers (bilinear.hip header); fp32 rounding differs from the
// 64 x 64 kernel's (another k order), inside the 1e-5 contract of these BLAS-backed products.
#ifndef MMS_PANEL_GEMM_H_
#define MMS_PANEL_GEMM_H_

#include <type_traits>

#include "mms_common.h"
#include "pairrank_math.h"

namespace mms {

typedef float pg_v4f __attribute__((ext_vector_type(4)));

struct PanelArgs {
  int M, N, K;
  const float* A; long long lda;     // A_KC: A(i,k) = A[i*lda + k]; else A(i,k) = A[k*lda + i]
  const float* B; long long ldb;     // B(k,n) = B[k*ldb + n]
  float* C; long long ldc;           // may be null (row dot only)
  const float* kscale;               // !A_KC only: A(i,k) *= kscale[k]
  const float* rowscale;             // v(i,:) = rowscale[i] * acc(i,:)
  const float* Y; long long ldy;     // rowdot[i*rd_stride] = (rd_bias[b] +) sum_n v(i,n) * Y(i,n)
  float* rowdot; long long rd_stride; const float* rd_bias;
  int nseg; long long a_seg, b_seg;  // sum over segments s of A_s . B_s  (A += s*a_seg, B += s*b_seg)
  int nb; long long b_b, c_b, rd_b;  // batch (blockIdx.y): B += b*b_b, C += b*c_b, rowdot += b*rd_b
  int ksplit, kchunk; long long c_ks;  // split-K (nseg == 1): chunk ks covers [ks*kchunk, ..), slab at C + ks*c_ks
  int row_blocks;
  int stream_c;                      // C is written once and read by a later launch: non-temporal stores
  // side job of the loader waves (ksplit == 1, nb == 1): side_out(i,:) = side_scale[i] * side_in(i,:) for the
  // M rows of the product, side_cols (% 4 == 0, <= 16*NT) floats per row; side_out may be side_in
  const float* side_in; float* side_out; const float* side_scale; long long side_ld; int side_cols;
  // fused learned-metric triplet step (A_KC, ksplit == 1, nb == 1; round 3): the product is P = Q W, Y = A+ and Y2 = A-.
  // The epilogue forms both row dots s+ = P_i . a+_i (-> rowdot) and s- = P_i . a-_i (-> trip_sneg), PairRankLoss's term
  // and its two gradients g+, g- for the row (pair_rank_loss_layer.cpp:28-37, 72-79), and writes what the backward
  // needs -- da+ = g+ P_i, da- = g- P_i, B_i = g+ a+_i + g- a-_i -- so that neither P nor the (N, 1) score diffs ever
  // reach HBM; trip_terms[i] = the row's loss term.  C must be null.
  const float* Y2; const float* trip_y; float trip_margin, trip_s0, trip_s1; int trip_hinge_ge;
  float* trip_sneg; float* trip_terms; float* trip_dapos; float* trip_daneg; float* trip_b;
};

constexpr int PG_LDA = 40;           // A_KC image row stride (32 k values + 8 pad floats)
constexpr int PG_LDT = 68;           // !A_KC image row stride (64 rows of the panel + 4 pad floats)

template <int NT, bool A_KC>
struct PanelGeom {
  static constexpr int LD = 16 * NT + 4;                 // B row stride; epilogue staging stride
  static constexpr int B_CPR = LD / 4;                   // 16-byte chunks per B row
  static constexpr int B_NCH = 32 * B_CPR;
  static constexpr int NBW = ((B_NCH + 63) / 64 + 3) / 4;   // B DMA instructions per wave per tile
  static constexpr int A_CPR = A_KC ? PG_LDA / 4 : PG_LDT / 4;
  static constexpr int A_NCH = (A_KC ? 64 : 32) * A_CPR;
  static constexpr int NAW = ((A_NCH + 63) / 64 + 3) / 4;   // A DMA instructions per wave per tile
  static constexpr int NPT = NBW + NAW + (A_KC ? 0 : 1);    // DMA instructions per wave per tile (all kinds)
  static constexpr int B_F = NBW * 4 * 256;              // floats
  static constexpr int A_F = NAW * 4 * 256;
  static constexpr int S_F = 64;                         // kscale slot (one dword DMA: 64 lanes x 4 bytes)
  static constexpr int STAGE_F = B_F + A_F + S_F;
  static constexpr size_t kLdsBytes =
      (3 * (size_t)STAGE_F > 64 * (size_t)LD ? 3 * (size_t)STAGE_F : 64 * (size_t)LD) * sizeof(float);
  // the hand-issued tile (round 3): operands are read DIST ahead of their MFMA; NOPS MFMAs per tile
  static constexpr int DIST = 2 * NT - 1 < 16 ? 2 * NT - 1 : 16;
  static constexpr int NOPS = 8 * NT;
  static_assert(NPT < 60, "vmcnt is a 6-bit counter");
};

// One LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds_byte, lds_byte + 1024).
// M0 is compiler-reserved and not preserved around an asm statement: saved and restored inside it.
__device__ __forceinline__ void pg_dma16(const float* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void pg_dma4(const float* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

#ifdef MMS_PG_STAMPS      // dev-only (tools/panelbench.hip): per-workgroup shader-clock / wall-clock stamps around the main loop
__device__ unsigned long long* pg_stamp_buf = nullptr;
#define PG_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (pg_stamp_buf && threadIdx.x == 0) {                                                           \
      pg_stamp_buf[(size_t)blockIdx.x * 8 + 2 * (k)] = __builtin_amdgcn_s_memtime();                  \
      pg_stamp_buf[(size_t)blockIdx.x * 8 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime();          \
    }                                                                                                 \
  } while (0)
#else
#define PG_STAMP(k) do {} while (0)
#endif

// f(integral_constant<int, I>) for I = 0 .. N-1: the hand-issued tile needs its operand index as a compile-time
// constant (instruction immediates); a `#pragma unroll` loop only promises that if the unroller agrees
template <int I, int N, typename F>
__device__ __forceinline__ void pg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pg_static_for<I + 1, N>(f);
  }
}

template <int NT, bool A_KC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void panel_gemm_kernel(PanelArgs p) {
  using G = PanelGeom<NT, A_KC>;
  extern __shared__ float4 pg_lds4[];
  float* lds = reinterpret_cast<float*>(pg_lds4);
  constexpr int LD = G::LD, NBW = G::NBW, NAW = G::NAW, NPT = G::NPT, STAGE_F = G::STAGE_F;
  // LDS byte address of the ring (dynamic LDS starts at the workgroup's LDS base: offset 0 with no static LDS)
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;

  const int t = threadIdx.x, lane = t & 63, r = lane & 15, g = lane >> 4;
  const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave8 >= 4;                    // waves 4-7 move data, waves 0-3 multiply
  const int wave = wave8 & 3;                        // the 16-row block of the panel / the DMA slot group

  // ---- which panel / k-chunk / batch entry --------------------------------------------------------
  int rb, ks = 0;
  if (p.ksplit > 1) {
    // workgroups are dealt round-robin over the 8 XCDs: ids equal mod 8 share an L2.  All row blocks of
    // one k-chunk get ids equal mod 8.
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
    ks = (local / p.row_blocks) * 8 + xcd;
    rb = local % p.row_blocks;
    if (ks >= p.ksplit) return;                       // whole workgroup, before any barrier
  } else {
    rb = blockIdx.x;
  }
  const int bt = blockIdx.y;
  const float* Bb = p.B + (long long)bt * p.b_b;
  const int i0 = rb * 64 + wave * 16;
  const int kbeg = ks * p.kchunk;
  const int kend = p.ksplit > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const int tiles_seg = (kend - kbeg + 31) >> 5;
  const int ntiles = tiles_seg * p.nseg;

  // ---- DMA slots: this wave's instructions j = wave + 4*jj; per-lane source offsets are tile-invariant ----
  // offsets are relative to the tile's origin (B: row k0 of the segment; A_KC: column k0; !A_KC: row k0);
  // -1 = the slot holds nothing (pad chunk, row or column outside the matrix): fetch a clamped address.
  int boff[NBW], bkk[NBW], aoff[NAW], akk[NAW];
#pragma unroll
  for (int jj = 0; jj < NBW; ++jj) {
    const int c = (wave + 4 * jj) * 64 + lane;
    const int kk = c / G::B_CPR, cc = c - kk * G::B_CPR;
    const bool ok = kk < 32 && 4 * cc < p.N;
    boff[jj] = ok ? (int)(kk * p.ldb) + 4 * cc : -1;
    bkk[jj] = kk;
  }
#pragma unroll
  for (int jj = 0; jj < NAW; ++jj) {
    const int c = (wave + 4 * jj) * 64 + lane;
    if (A_KC) {
      const int row = c / G::A_CPR, cc = c - row * G::A_CPR;
      const bool ok = row < 64 && cc < 8;
      aoff[jj] = ok ? (int)(min(rb * 64 + row, p.M - 1) * p.lda) + 4 * cc : -1;
      akk[jj] = 4 * cc;
    } else {
      const int kk = c / G::A_CPR, cc = c - kk * G::A_CPR;
      const bool ok = kk < 32 && cc < 16 && rb * 64 + 4 * cc < p.M;   // M % 4 == 0: a chunk is all-in or all-out
      aoff[jj] = ok ? (int)(kk * p.lda) + rb * 64 + 4 * cc : -1;
      akk[jj] = kk;
    }
  }
  // Round 3 (tools/mfmastruct.hip, profiles/r03_mfmastruct.txt): a VALU instruction issued by a loader wave costs the
  // compute wave on its SIMD ~8.5 cycles of matrix-pipe time, and the per-slot address arithmetic below (selects and
  // 64-bit adds: ~6 VALU instructions per DMA, 84 per tile) is what held the main loop at ~40 cycles per MFMA instead
  // of the 33 the same DMAs cost when issued without any: 19 MFMAs + 19 operand reads per k-step run at 32.1 cycles
  // per MFMA alone, 33.3 beside 14 DMAs per tile, 36.8 beside 84 VALU instructions, 41.6 beside both.  A FULL tile
  // therefore issues with none: the tile's origin is a scalar base (SALU), the per-lane byte offsets are
  // tile-invariant registers (slots that hold nothing fetch the tile's first bytes), M0 moves are SALU.  Only a
  // partial tile (the last of a segment) takes the selects.
  unsigned bvo[NBW], avo[NAW];
#pragma unroll
  for (int jj = 0; jj < NBW; ++jj) bvo[jj] = boff[jj] >= 0 ? (unsigned)boff[jj] * 4u : 0u;
#pragma unroll
  for (int jj = 0; jj < NAW; ++jj) avo[jj] = aoff[jj] >= 0 ? (unsigned)aoff[jj] * 4u : 0u;
  const unsigned svo = (unsigned)(lane & 31) * 4u;
  auto dma16s = [&](const float* sbase, unsigned voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
  };
  auto dma4s = [&](const float* sbase, unsigned voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
  };
  auto issue_tile = [&](int stage, int seg, int k0) {
    seg = __builtin_amdgcn_readfirstlane(seg);          // wave-uniform by construction; said so, the tile origins below
    k0 = __builtin_amdgcn_readfirstlane(k0);            // are scalar registers (the DMAs take them as their SGPR base)
    const unsigned sb = lds_base + (unsigned)(stage * STAGE_F) * 4u;
    const float* Bs = Bb + (long long)seg * p.b_seg + (long long)k0 * p.ldb;
    const float* As = p.A + (long long)seg * p.a_seg + (A_KC ? (long long)k0 : (long long)k0 * p.lda);
    if (k0 + 32 <= kend) {                              // a full tile (wave-uniform)
#pragma unroll
      for (int jj = 0; jj < NBW; ++jj) dma16s(Bs, bvo[jj], sb + (unsigned)(wave + 4 * jj) * 1024u);
#pragma unroll
      for (int jj = 0; jj < NAW; ++jj) dma16s(As, avo[jj], sb + (unsigned)G::B_F * 4u + (unsigned)(wave + 4 * jj) * 1024u);
      if (!A_KC) dma4s(p.kscale + k0, svo, sb + (unsigned)(G::B_F + G::A_F) * 4u);
      return;
    }
#pragma unroll
    for (int jj = 0; jj < NBW; ++jj) {
      const bool ok = boff[jj] >= 0 && k0 + bkk[jj] < kend;
      pg_dma16(ok ? Bs + boff[jj] : Bb, sb + (unsigned)(wave + 4 * jj) * 1024u);
    }
#pragma unroll
    for (int jj = 0; jj < NAW; ++jj) {
      const bool ok = aoff[jj] >= 0 && k0 + akk[jj] < kend;
      pg_dma16(ok ? As + aoff[jj] : p.A, sb + (unsigned)G::B_F * 4u + (unsigned)(wave + 4 * jj) * 1024u);
    }
    if (!A_KC) {                                      // the tile's 32 kscale values (every wave: same bytes, same place)
      const int kc = min(k0 + (lane & 31), kend - 1);
      pg_dma4(p.kscale + kc, sb + (unsigned)(G::B_F + G::A_F) * 4u);
    }
  };

  pg_v4f acc[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) acc[tt] = (pg_v4f){0.f, 0.f, 0.f, 0.f};

  // One k-tile = 8 MFMA steps (h, i) x NT column tiles.  The compute waves' stream is software-pipelined by
  // hand across steps AND across tiles: while the NT MFMAs of step s issue, the B operands of step s+1 are
  // read from LDS into a second register set, one LDS read in the shadow of each MFMA; the barrier that
  // publishes tile T+1 sits in front of step 7 of tile T (every LDS read of tile T has completed by then: its
  // step-7 operands are already in registers), so the first operands of tile T+1 -- its A fragments and its
  // step-0 B row -- are fetched beside the last MFMAs of tile T.  (Left to itself the compiler emits
  // read -> wait -> two MFMAs; all reads first, then all MFMAs, idles the matrix pipe while the reads issue.)
  pg_v4f av[2];
  float bv[2][NT];
#ifdef MMS_PG_STAMPS
  unsigned long long pg_wait_lgkm = 0, pg_wait_bar = 0;
#endif
  auto read_a = [&](pg_v4f (&dst)[2], int stage) {
    const float* At = lds + stage * STAGE_F + G::B_F;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (A_KC) {
        dst[h] = *reinterpret_cast<const pg_v4f*>(At + (wave * 16 + r) * PG_LDA + 16 * h + 4 * g);
      } else {
        const float* sc = At + G::A_F;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          dst[h][i] = At[(16 * h + 4 * g + i) * PG_LDT + wave * 16 + r] * sc[16 * h + 4 * g + i];
      }
    }
  };
  auto read_b = [&](float (&dst)[NT], int stage, int st) {
    const float* bn = lds + stage * STAGE_F + (16 * (st >> 2) + 4 * g + (st & 3)) * LD + r;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) dst[tt] = bn[16 * tt];
  };
  auto interleave = [&]() {                          // issue order: one LDS read in the shadow of each MFMA
#pragma unroll
    for (int u = 0; u < (NT + 1) / 2; ++u) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NT - (NT + 1) / 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);     // what is left of the reads (A fragments)
    __builtin_amdgcn_sched_barrier(0);
  };
  // A full tile whose operands of step 0 (av, bv[0]) are already in registers.  BAR: tile T+1 exists (the
  // barrier that publishes it is taken in front of step 7); PRE: tile T+1 is a full tile, prefetch its first
  // operands beside step 7.
  auto full_tile = [&](auto bar_tag, auto pre_tag, int stage, int nstage) {
    constexpr bool BAR = decltype(bar_tag)::value, PRE = decltype(pre_tag)::value;
#if defined(MMS_PG_ABLATE) && MMS_PG_ABLATE >= 2      // dev-only timing ablation: MFMAs alone, operands from registers
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (st == 7 && BAR) asm volatile("s_barrier" ::: "memory");
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st >> 2][st & 3], av[(st >> 2) ^ 1][st & 3], acc[tt], 0, 0, 0);
    }
    return;
#endif
#pragma unroll
    for (int st = 0; st < 7; ++st) {
      read_b(bv[(st + 1) & 1], stage, st + 1);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st >> 2][st & 3], bv[st & 1][tt], acc[tt], 0, 0, 0);
      interleave();
    }
#ifdef MMS_PG_STAMPS
    const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
    if (BAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
    if (BAR) asm volatile("s_barrier" ::: "memory");
    const unsigned long long tb2 = __builtin_amdgcn_s_memtime();
    pg_wait_lgkm += tb1 - tb0; pg_wait_bar += tb2 - tb1;
#else
    if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // B_{T+1}
#endif
    pg_v4f an[2];
    if (PRE) {
      read_a(an, nstage);
      read_b(bv[0], nstage, 0);
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
      acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][3], bv[1][tt], acc[tt], 0, 0, 0);
    interleave();
    if (PRE) {
      av[0] = an[0]; av[1] = an[1];
      // pin the prefetch in THIS tile: hipcc otherwise sinks these loads across the loop's back edge, next to
      // their uses in step 0 of the next tile (read -> wait -> MFMA again)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) asm volatile("" : "+v"(bv[0][tt]));
      asm volatile("" : "+v"(av[0]), "+v"(av[1]));
    }
  };
  // The partial tile that ends a segment (K % 32 != 0): k values past kend are zeroed in BOTH operands (the
  // LDS rows behind them hold whatever the clamped DMA fetched).  Not pipelined: one per segment.
  auto tail_tile = [&](int stage, int k0) {
    const float* Bs = lds + stage * STAGE_F;
    pg_v4f at[2];
    read_a(at, stage);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (k0 + 16 * h >= kend) continue;              // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = k0 + 16 * h + 4 * g + i < kend;
        const float a1 = ok ? at[h][i] : 0.f;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          float b1 = Bs[(16 * h + 4 * g + i) * LD + r + 16 * tt];
          if (!ok) b1 = 0.f;
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[tt], 0, 0, 0);
        }
      }
    }
  };

  // ---- round 3: the compute waves' tile, HAND-ISSUED -------------------------------------------------------------
  // tools/mfmastruct.hip (profiles/r03_mfmastruct.txt): every VALU instruction that shares a SIMD with the MFMA
  // stream -- the loaders' address arithmetic, but just as much the compute wave's OWN LDS address updates, 3-6 per
  // k-step in the compiler-scheduled tile above -- costs the matrix pipe 8-11 cycles: 32.1 cycles per MFMA for the
  // bare stream, 34.4 with four VALU instructions per k-step, 39.3 measured in the product loop.  Here the tile has
  // none: each stage has ONE per-lane base register for B and one for A, every operand is `ds_read_b32 v, base
  // offset:IMM` (all offsets are compile-time constants below 64 KB), issued exactly DIST = 16 operands ahead of its
  // MFMA so that `s_waitcnt lgkmcnt(15)` -- the counter has four bits -- in front of an MFMA means "its operand has
  // landed" (LDS operations return in order), and __builtin_amdgcn_sched_barrier pins the order read, wait, MFMA.
  // The microbenchmark's loop of exactly this shape runs at 32.1 cycles per MFMA, 34.6 with the ring, the barrier and
  // the loaders' DMAs beside it.  Operand n = NT st + tt of a tile lives in slot n mod 2 NT.
#ifndef MMS_PG_HAND
#define MMS_PG_HAND 1
#endif
  float bb[2 * NT];
  pg_v4f an[2];                                       // A_KC: the next tile's fragments as read
  float araw[8], sraw[8];                             // !A_KC: the next tile's A values and their k scales as read
  unsigned bbase[3], abase[3], scbase[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    bbase[s3] = lds_base + (unsigned)(s3 * STAGE_F + 4 * g * LD + r) * 4u;
    abase[s3] = lds_base + (unsigned)(s3 * STAGE_F + G::B_F + (A_KC ? (wave * 16 + r) * PG_LDA + 4 * g : 4 * g * PG_LDT + wave * 16 + r)) * 4u;
    scbase[s3] = lds_base + (unsigned)(s3 * STAGE_F + G::B_F + G::A_F + 4 * g) * 4u;
  }
  // the three bases ROTATE with the tiles ([0]: the tile being multiplied, [1]: the next one): three register moves per
  // array per tile instead of a selection by T % 3, which the compiler turned into branches around the tile bodies
  // and, at their merges, into copies of the whole accumulator set
  auto hand_rotate = [&]() {
    unsigned t0 = bbase[0]; bbase[0] = bbase[1]; bbase[1] = bbase[2]; bbase[2] = t0;
    t0 = abase[0]; abase[0] = abase[1]; abase[1] = abase[2]; abase[2] = t0;
    if (!A_KC) { t0 = scbase[0]; scbase[0] = scbase[1]; scbase[1] = scbase[2]; scbase[2] = t0; }
  };
  auto hand_issue_a = [&](int which) {                // tile [which]'s A operands: asked for, not waited for
    const unsigned ab = which ? abase[1] : abase[0];
    if (A_KC) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(an[0]) : "v"(ab) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(an[1]) : "v"(ab) : "memory");
    } else {
      const unsigned sb3 = which ? scbase[1] : scbase[0];
      pg_static_for<0, 8>([&araw, &sraw, ab, sb3](auto hi_tag) {
        constexpr int hi = decltype(hi_tag)::value;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(araw[hi]) : "v"(ab), "n"((16 * (hi >> 2) + (hi & 3)) * PG_LDT * 4) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(sraw[hi]) : "v"(sb3), "n"((16 * (hi >> 2) + (hi & 3)) * 4) : "memory");
      });
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto hand_finish_a = [&]() {                        // behind the A reads AND at most 15 younger LDS operations
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(G::DIST < 16 ? G::DIST : 15) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (A_KC) {
      av[0] = an[0]; av[1] = an[1];
    } else {
#pragma unroll
      for (int hi = 0; hi < 8; ++hi) av[hi >> 2][hi & 3] = araw[hi] * sraw[hi];
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#define MMS_PG_BREAD(base, m)                                                                                          \
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bb[(m) % (2 * NT)]) : "v"(base),                                 \
               "n"(((16 * (((m) / NT) >> 2) + (((m) / NT) & 3)) * G::LD + 16 * ((m) % NT)) * 4) : "memory")
  auto hand_prime = [&]() {                           // tile [0] has landed: its A operands and its first G::DIST B operands
    hand_issue_a(0);
    const unsigned bc = bbase[0];
    pg_static_for<0, G::DIST>([&bb, bc](auto m_tag) {
      constexpr int m = decltype(m_tag)::value;
      MMS_PG_BREAD(bc, m);
    });
    __builtin_amdgcn_sched_barrier(0);
    hand_finish_a();
  };
  // A full tile whose A operands and first G::DIST B operands have been asked for.  BAR: tile T+1 exists -- its barrier
  // sits in front of MFMA G::NOPS - G::DIST, where every read of this tile has been issued (and is awaited: the loaders
  // overwrite this stage next).  PRE: tile T+1 is a full tile: its A operands and first G::DIST B operands are asked for
  // beside the last G::DIST MFMAs of this one.
  auto hand_tile = [&](auto bar_tag, auto pre_tag) {
    constexpr bool BAR = decltype(bar_tag)::value, PRE = decltype(pre_tag)::value;
    const unsigned bc = bbase[0], bn = bbase[1];
    __builtin_amdgcn_sched_barrier(0);
    pg_static_for<0, G::NOPS>([&acc, &av, &bb, &hand_issue_a, bc, bn](auto n_tag) {
      constexpr int n = decltype(n_tag)::value;
      if (BAR && n == G::NOPS - G::DIST) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // B_{T+1}
        __builtin_amdgcn_sched_barrier(0);
        if (PRE) hand_issue_a(1);
      }
      if (!(BAR && n >= G::NOPS - G::DIST)) {               // (behind the barrier's wait every operand of this tile is there)
        // younger reads in front of MFMA n: G::DIST - 1 while the stream runs, fewer at the end of a tile without successor
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(G::NOPS - 1 - n < G::DIST - 1 ? G::NOPS - 1 - n : G::DIST - 1) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[n % NT] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(n / NT) >> 2][(n / NT) & 3], bb[n % (2 * NT)], acc[n % NT], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (n + G::DIST < G::NOPS) MMS_PG_BREAD(bc, n + G::DIST);
      else if (PRE) MMS_PG_BREAD(bn, n + G::DIST - G::NOPS);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (PRE) hand_finish_a();
    hand_rotate();
  };
#undef MMS_PG_BREAD

  // ---- main loop: three-stage ring, one barrier per 32-deep k-tile -----------------------------------
  // Both roles execute EXACTLY ntiles + 2 workgroup barriers (B_0 .. B_{ntiles-1}, B_end, B_stage).
  // Loader invariant at the top of iteration T: tiles T and T+1 have been issued (NPT DMAs per wave each, in
  // that order; past the last tile the slots fetch clamped addresses into a stage nobody reads), and side-job
  // loads / stores are always issued BEFORE the tile's DMAs of the same iteration.  vmcnt(NPT) therefore means
  // "tile T has landed for this wave"; barrier B_T extends that to every wave and also tells the loaders that
  // every compute wave has finished tile T-1, whose stage the DMAs of tile T+2 may now overwrite.
  const int nfull = (kend - kbeg) >> 5;
  const bool has_tail = ((kend - kbeg) & 31) != 0;
  const int NC = p.N >> 2;                            // float4 per row of C
  constexpr int NCC = (4 * NT + 15) / 16;
  pg_v4f y4[4][NCC];                                  // loaders, row dot: Y rows g + 4 rr of this wave's block
  pg_v4f y4b[4][NCC];                                 // ... and Y2's (fused triplet step)
  if (loader) {
    // A loader issues ~100 instructions per tile, its SIMD partner 250: at equal priority the younger loader
    // waited ~450 cycles per DMA for an issue slot (the loaders, not the matrix pipe, then paced the kernel)
    __builtin_amdgcn_s_setprio(3);
    if (p.Y) {                                        // issued first, consumed in the epilogue
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int growc = min(i0 + g + 4 * rr, p.M - 1);
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
          const int c4 = r + 16 * cc;
          y4[rr][cc] = *reinterpret_cast<const pg_v4f*>(p.Y + (long long)growc * p.ldy + 4 * (c4 < NC ? c4 : 0));
        }
      }
    }
    if (p.Y2) {                                       // the negatives' rows, like Y
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int growc = min(i0 + g + 4 * rr, p.M - 1);
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
          const int c4 = r + 16 * cc;
          y4b[rr][cc] = *reinterpret_cast<const pg_v4f*>(p.Y2 + (long long)growc * p.ldy + 4 * (c4 < NC ? c4 : 0));
        }
      }
    }
    int iseg = 0, ik0 = kbeg;                          // the next tile to issue
    auto issue_next = [&](int stage) {
      issue_tile(stage, iseg < p.nseg ? iseg : 0, iseg < p.nseg ? ik0 : kbeg);
      ik0 += 32;
      if (ik0 >= kend) { ik0 = kbeg; ++iseg; }
    };
    // side job: rows g + 4 rr of this wave's 16-row block, one rr per iteration: requested in iteration rr,
    // scaled and stored in iteration rr + 1.
    // Its loads are issued by hand (asm), like the DMAs: the compiler does not see the DMAs, so for a load it
    // does see it counts no younger operation and waits with vmcnt(<= 4) at the first use -- i.e. until the
    // tile requested a moment ago has landed.  That cost one tile of prefetch distance in each of the first
    // four iterations of every workgroup.  Requested by hand, side-job data is covered by the loop's own counted
    // wait (it is issued BEFORE the DMAs of its iteration: vmcnt(NPT) at the top of the next one means it has
    // landed); side_pin() then hands the registers to the compiler, and nothing may read them before it.
    const int SC = p.side_in ? p.side_cols >> 2 : 0;
    pg_v4f sx[NCC];
    float ssc[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.side_in) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float* src = p.side_scale + min(i0 + g + 4 * rr, p.M - 1);
        asm volatile("global_load_dword %0, %1, off" : "=v"(ssc[rr]) : "v"(src) : "memory");
      }
    }
    auto side_load = [&](int rr) {
      const int growc = min(i0 + g + 4 * rr, p.M - 1);
#pragma unroll
      for (int cc = 0; cc < NCC; ++cc) {
        const int c4 = r + 16 * cc;
        const float* src = p.side_in + (long long)growc * p.side_ld + 4 * (c4 < SC ? c4 : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sx[cc]) : "v"(src) : "memory");
      }
    };
    auto side_pin = [&]() {                            // only behind a wait that covers the requests above
#pragma unroll
      for (int cc = 0; cc < NCC; ++cc) asm volatile("" : "+v"(sx[cc]));
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) asm volatile("" : "+v"(ssc[rr]));
    };
    auto side_store = [&](int rr) {
      const int grow = i0 + g + 4 * rr;
      const float sc = ssc[rr];
#pragma unroll
      for (int cc = 0; cc < NCC; ++cc) {
        const int c4 = r + 16 * cc;
        if (c4 < SC && grow < p.M)
          __builtin_nontemporal_store(sc * sx[cc], reinterpret_cast<pg_v4f*>(p.side_out + (long long)grow * p.side_ld + 4 * c4));
      }
    };
    if (ntiles > 0) {
      issue_next(0);
      issue_next(1);
    }
#ifdef MMS_PG_STAMPS
    unsigned long long ld_wait = 0, ld_bar = 0, ld_issue = 0;
#endif
    const int SP = ntiles >= 8 ? 2 : 1;               // side-job passes: requested in iteration SP rr, stored one later
    int s_loaded = 0, s_stored = 0;
    for (int T = 0; T < ntiles; ++T) {
#ifdef MMS_PG_STAMPS
      const unsigned long long l0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
      const unsigned long long l1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_barrier" ::: "memory");
      const unsigned long long l2 = __builtin_amdgcn_s_memtime();
      if (T > 0) { ld_wait += l1 - l0; ld_bar += l2 - l1; }
#else
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPT) : "memory");       // B_T
#endif
      if (p.side_in) {
        // round 3: one row pass every SPth iteration instead of the first four -- 256 workgroups streaming their
        // 2 x 77 KB at the same moment, in front of their own first tiles, cost the launch 3-4 us
        if (s_stored < s_loaded && T >= SP * s_stored + 1) { side_pin(); side_store(s_stored); ++s_stored; }
        if (s_loaded < 4 && s_loaded == s_stored && T >= SP * s_loaded) { side_load(s_loaded); ++s_loaded; }
      }
#if defined(MMS_PG_ABLATE) && MMS_PG_ABLATE >= 1      // dev-only timing ablation (tools/panelbench.hip): no loads after the prologue
      if (false)
#endif
      issue_next((T + 2) % 3);
#ifdef MMS_PG_STAMPS
      if (T > 0) ld_issue += __builtin_amdgcn_s_memtime() - l2;
#endif
    }
#ifdef MMS_PG_STAMPS
    if (pg_stamp_buf && threadIdx.x == 256) {
      pg_stamp_buf[(size_t)(gridDim.x + blockIdx.x) * 8 + 0] = ld_wait;
      pg_stamp_buf[(size_t)(gridDim.x + blockIdx.x) * 8 + 1] = ld_bar;
      pg_stamp_buf[(size_t)(gridDim.x + blockIdx.x) * 8 + 2] = ld_issue;
    }
#endif
    if (p.side_in) {                                  // what the loop was too short for
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      side_pin();
      if (s_stored < s_loaded) { side_store(s_stored); ++s_stored; }
      for (int rr = s_loaded; rr < 4; ++rr) {
        side_load(rr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        side_pin();
        side_store(rr);
      }
    }
  } else {
    // Barriers: B_0 here, B_{T+1} inside tile T (full tiles: in front of step 7; a partial tile: behind it).
    int T = 0;
    PG_STAMP(0);
    if (ntiles > 0) asm volatile("s_barrier" ::: "memory");                              // B_0
#if MMS_PG_HAND && !defined(MMS_PG_ABLATE)
    // Same-box A/B (gpurun_out/r3/ab_hand.txt, three alternations): the hand-issued tile is worth 2.4 us on the
    // split-K dW product (38.7 -> 36.3 us), nothing on Q.W (40.1 / 40.4) and COSTS 1.8 us where the loaders carry
    // the da side job (36.6 -> 38.4) -- the launch is then power-bound: the shader clock falls from 2.18 to 2.10 GHz
    // as the stream tightens (stamps).  So it serves the !A_KC product only; MMS_PG_HAND=2 forces it everywhere.
    if (p.nseg == 1 && (!A_KC || MMS_PG_HAND >= 2)) {
      // one segment (every product of cfg 3): all full tiles that are followed by a full tile run in ONE loop body
      if (nfull > 0) {
        hand_prime();
        for (int f = 0; f + 1 < nfull; ++f) hand_tile(std::true_type{}, std::true_type{});
        if (has_tail) hand_tile(std::true_type{}, std::false_type{});
        else hand_tile(std::false_type{}, std::false_type{});
      }
      if (has_tail) tail_tile(nfull % 3, kbeg + 32 * nfull);
      T = ntiles;
    } else
#endif
    {
    bool primed = false;                               // are tile T's first operands in registers?
    for (int seg = 0; seg < p.nseg; ++seg) {
      int k0 = kbeg;
      for (int f = 0; f < nfull; ++f, k0 += 32, ++T) {
        if (!primed) {
          read_a(av, T % 3);
          read_b(bv[0], T % 3, 0);
        }
        const bool next_exists = T + 1 < ntiles;
        const bool next_full = next_exists && !(has_tail && f == nfull - 1);
        if (next_full) full_tile(std::true_type{}, std::true_type{}, T % 3, (T + 1) % 3);
        else if (next_exists) full_tile(std::true_type{}, std::false_type{}, T % 3, (T + 1) % 3);
        else full_tile(std::false_type{}, std::false_type{}, T % 3, (T + 1) % 3);
        primed = next_full;
      }
      if (has_tail) {
        tail_tile(T % 3, k0);
        ++T;
        if (T < ntiles) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B_T of the next segment's first tile
        primed = false;
      }
    }
    }
    PG_STAMP(1);
  }
  // ---- epilogue: accumulators -> LDS (16 x LD slice per compute wave) -> 256-byte row segments ----------
  if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMAs (and side job) have landed
  asm volatile("s_barrier" ::: "memory");                            // B_end: ring drained, nobody reads it any more
#if defined(MMS_PG_ABLATE) && MMS_PG_ABLATE >= 3      // dev-only: no epilogue (one store keeps the accumulators live)
  {
    float keep = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) keep += acc[tt][0] + acc[tt][1] + acc[tt][2] + acc[tt][3];
    if (keep == 12345.678f && p.C) p.C[t] = keep;
    return;
  }
#endif
  float* Cs = lds + wave * 16 * LD;
  if (!loader) {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int j = 0; j < 4; ++j) Cs[(4 * g + j) * LD + 16 * tt + r] = acc[tt][j];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // B_stage: the staged panel is visible to all 8 waves

  if (p.Y2) {
    // fused triplet step: the loader wave holds both answers' rows; it takes the two dots, PairRankLoss for the row
    // and the three scaled row stores (the compute waves have nothing to add: P itself is not stored)
    if (!loader) return;
    float yv[4], diffv[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) yv[rr] = p.trip_y[min(i0 + g + 4 * rr, p.M - 1)];
    asm volatile("" : "+v"(yv[0]), "+v"(yv[1]), "+v"(yv[2]), "+v"(yv[3]));
    (void)diffv;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row_l = g + 4 * rr, grow = i0 + row_l;
      const bool valid = grow < p.M;
      pg_v4f v[NCC];
      float dp = 0.f, dn = 0.f;
#pragma unroll
      for (int cc = 0; cc < NCC; ++cc) {
        const int c4 = r + 16 * cc;
        if (c4 < NC) {
          v[cc] = *reinterpret_cast<const pg_v4f*>(Cs + row_l * LD + 4 * c4);
          dp += (v[cc][0] * y4[rr][cc][0] + v[cc][1] * y4[rr][cc][1]) + (v[cc][2] * y4[rr][cc][2] + v[cc][3] * y4[rr][cc][3]);
          dn += (v[cc][0] * y4b[rr][cc][0] + v[cc][1] * y4b[rr][cc][1]) + (v[cc][2] * y4b[rr][cc][2] + v[cc][3] * y4b[rr][cc][3]);
        }
      }
      dp = dpp_add<0xB1, 0xf>(dp); dn = dpp_add<0xB1, 0xf>(dn);      // every lane of the row's 16 ends with the totals
      dp = dpp_add<0x4E, 0xf>(dp); dn = dpp_add<0x4E, 0xf>(dn);
      dp = dpp_add<0x141, 0xf>(dp); dn = dpp_add<0x141, 0xf>(dn);
      dp = dpp_add<0x140, 0xf>(dp); dn = dpp_add<0x140, 0xf>(dn);
      const PairTerm pt = pair_term(dp, dn, yv[rr], p.trip_margin);   // pair_rank_loss_layer.cpp:28-37
      float gp, gn;
      pair_grad(yv[rr], pt.ordered, pt.similar, p.trip_s0, p.trip_s1, gp, gn, p.trip_hinge_ge != 0);   // :72-79
      if (r == 0 && valid) {
        p.rowdot[grow] = dp;
        p.trip_sneg[grow] = dn;
        p.trip_terms[grow] = pt.term;
      }
#pragma unroll
      for (int cc = 0; cc < NCC; ++cc) {
        const int c4 = r + 16 * cc;
        if (c4 < NC && valid) {
          const long long at = (long long)grow * p.ldy + 4 * c4;
          __builtin_nontemporal_store(gp * v[cc], reinterpret_cast<pg_v4f*>(p.trip_dapos + at));     // da+ = g+ (W^T q)
          __builtin_nontemporal_store(gn * v[cc], reinterpret_cast<pg_v4f*>(p.trip_daneg + at));     // da- = g- (W^T q)
          *reinterpret_cast<pg_v4f*>(p.trip_b + at) = gp * y4[rr][cc] + gn * y4b[rr][cc];            // read next by two products
        }
      }
    }
    return;
  }
  // Output pass over block `wave`'s 16 rows, lane (g, r): rows g + 4 rr, float4 r + 16 cc.  With a row dot the
  // loader wave (which holds Y) takes the dots and the compute wave the stores; otherwise they split the rows.
  float* Cg = p.C ? p.C + (long long)bt * p.c_b + (long long)ks * p.c_ks : nullptr;
  const bool do_dot = loader && p.Y;
  const bool do_store = Cg && (p.Y ? !loader : true);
  const int rr0 = p.Y ? 0 : (loader ? 2 : 0), rr1 = p.Y ? 4 : (loader ? 4 : 2);
  // every scalar the pass needs is in a register before its first store: the stores sit under lane masks, so a
  // load consumed after one became s_waitcnt vmcnt(0) -- a wait for the ACKNOWLEDGEMENT of the row group just
  // stored, four times per workgroup
  float rsv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  if (p.rowscale) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) rsv[rr] = p.rowscale[min(i0 + g + 4 * rr, p.M - 1)];
  }
  float rdb = (do_dot && p.rd_bias) ? p.rd_bias[bt] : 0.f;
  asm volatile("" : "+v"(rsv[0]), "+v"(rsv[1]), "+v"(rsv[2]), "+v"(rsv[3]), "+v"(rdb));
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    if (rr < rr0 || rr >= rr1) continue;
    if (!do_dot && !do_store) continue;
    const int row_l = g + 4 * rr, grow = i0 + row_l;
    const bool valid = grow < p.M;
    const float rs = rsv[rr];
    float dot = 0.f;
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
      const int c4 = r + 16 * cc;
      if (c4 < NC) {
        pg_v4f v = *reinterpret_cast<const pg_v4f*>(Cs + row_l * LD + 4 * c4);
        if (p.rowscale) v = rs * v;
        if (do_dot) dot += (v[0] * y4[rr][cc][0] + v[1] * y4[rr][cc][1]) + (v[2] * y4[rr][cc][2] + v[3] * y4[rr][cc][3]);
#if defined(MMS_PG_ABLATE) && MMS_PG_ABLATE == -4    // dev-only: epilogue without its global stores
        if (do_store && valid && v[0] == 12345.678f) {
#else
        if (do_store && valid) {
#endif
          pg_v4f* dst = reinterpret_cast<pg_v4f*>(Cg + (long long)grow * p.ldc + 4 * c4);
          if (p.stream_c) __builtin_nontemporal_store(v, dst);
          else *dst = v;
        }
      }
    }
    if (do_dot) {
      dot = dpp_add<0xB1, 0xf>(dot);                  // the four steps that sum a row of 16 lanes
      dot = dpp_add<0x4E, 0xf>(dot);
      dot = dpp_add<0x141, 0xf>(dot);
      dot = dpp_add<0x140, 0xf>(dot);
      if (r == 0 && valid) {
        float* out = p.rowdot + (long long)bt * p.rd_b + (long long)grow * p.rd_stride;
        *out = p.rd_bias ? (rdb + dot) : dot;
      }
    }
  }
#ifdef MMS_PG_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (pg_stamp_buf && threadIdx.x == 0) {
    pg_stamp_buf[(size_t)blockIdx.x * 8 + 6] = pg_wait_lgkm;
    pg_stamp_buf[(size_t)blockIdx.x * 8 + 7] = pg_wait_bar;
  }
#endif
  PG_STAMP(2);
}

// ------------------------------------------------ host side -----------------------------------------
inline PanelArgs panel_args(int M, int N, int K, const float* A, long long lda, const float* B, long long ldb,
                            float* C, long long ldc) {
  PanelArgs p{};
  p.M = M; p.N = N; p.K = K; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.nseg = 1; p.nb = 1; p.ksplit = 1; p.kchunk = K; p.rd_stride = 1;
  p.row_blocks = (M + 63) / 64;
  return p;
}

inline bool pg_mult4(long long x) { return (x & 3) == 0; }

// Can the panel kernel run this product, and is the product big enough to prefer it?
inline bool panel_eligible(const PanelArgs& p, bool a_kc) {
  if (p.N > 304 || p.N < 4 || !pg_mult4(p.N) || !pg_mult4(p.ldb) || !aligned16(p.B) || !pg_mult4(p.b_b) ||
      !pg_mult4(p.b_seg))
    return false;
  if (!pg_mult4(p.lda) || !aligned16(p.A) || !pg_mult4(p.K) || !pg_mult4(p.a_seg)) return false;
  if (!a_kc && (!pg_mult4(p.M) || !p.kscale)) return false;
  if (p.C && (!pg_mult4(p.ldc) || !aligned16(p.C) || !pg_mult4(p.c_b) || !pg_mult4(p.c_ks))) return false;
  if (p.Y && (!pg_mult4(p.ldy) || !aligned16(p.Y))) return false;
  if (p.Y2 && (!a_kc || !p.Y || p.C || p.ksplit != 1 || p.nb != 1 || p.nseg != 1 || !aligned16(p.Y2) || !p.trip_y || !p.rowdot ||
               !p.trip_sneg || !p.trip_terms || !aligned16(p.trip_dapos) || !aligned16(p.trip_daneg) || !aligned16(p.trip_b)))
    return false;
  if (p.ksplit > 1 && (p.nseg != 1 || (p.kchunk & 31))) return false;
  const int nt = p.N <= 64 ? 4 : p.N <= 112 ? 7 : p.N <= 208 ? 13 : 19;       // panel_launch's choice
  if (p.side_in && (p.ksplit != 1 || p.nb != 1 || !pg_mult4(p.side_cols) || p.side_cols > 16 * nt || !pg_mult4(p.side_ld) ||
                    !aligned16(p.side_in) || !aligned16(p.side_out) || !p.side_scale))
    return false;
  if ((long long)p.M * p.lda >= (1LL << 30) || 64LL * p.ldb >= (1LL << 30)) return false;   // 32-bit slot BYTE offsets
  // a panel kernel launch has row_blocks x ksplit x nb workgroups of 4 waves: below ~a third of the chip the
  // 64 x 64 tiling's extra parallelism wins
  return (long long)p.row_blocks * p.ksplit * p.nb >= 96;
}

template <int NT, bool A_KC>
inline void panel_launch_t(const PanelArgs& p, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = PanelGeom<NT, A_KC>::kLdsBytes;
  auto kern = panel_gemm_kernel<NT, A_KC>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const unsigned gx = p.ksplit > 1 ? 8u * ((p.ksplit + 7) / 8) * p.row_blocks : (unsigned)p.row_blocks;
  hipLaunchKernelGGL(kern, dim3(gx, p.nb), dim3(512), lds, s, p);
}

inline void panel_launch(const PanelArgs& p, bool a_kc, hipStream_t s) {
#define MMS_PG(NT)                                                \
  do {                                                            \
    if (a_kc) panel_launch_t<NT, true>(p, s);                     \
    else panel_launch_t<NT, false>(p, s);                         \
  } while (0)
  if (p.N <= 64) MMS_PG(4);
  else if (p.N <= 112) MMS_PG(7);
  else if (p.N <= 208) MMS_PG(13);
  else MMS_PG(19);
#undef MMS_PG
}

// split count / chunk for a split-K panel product: all row blocks of a chunk on one XCD, at most one
// workgroup per CU (32 per XCD), chunks a multiple of the 32-deep k-tile
inline int panel_pick_ksplit(int row_blocks, int nb, int K, int* kchunk) {
  int per_xcd = 32 / (row_blocks * nb > 0 ? row_blocks * nb : 1);
  if (per_xcd < 1) per_xcd = 1;
  int want = 8 * per_xcd;
  const int maxs = (K + 63) / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  int chunk = (K + want - 1) / want;
  chunk = (chunk + 31) / 32 * 32;
  *kchunk = chunk;
  return (K + chunk - 1) / chunk;
}

// out (cols x rows) = in^T (in: rows x cols, row-major): the weight handed to the panel kernel as a k-major B
// when the product needs W^T (SimMatrix dq = diag(dT) A W^T).  W is D x D <= 360 KB: one small launch.
__global__ __launch_bounds__(256) void pg_transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int rr = r0 + ty + 8 * u, cc = c0 + tx;
    if (rr < rows && cc < cols) tile[ty + 8 * u][tx] = in[(long long)rr * cols + cc];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int oc = c0 + ty + 8 * u, orr = r0 + tx;          // out[oc][orr] = in[orr][oc]
    if (oc < cols && orr < rows) out[(long long)oc * rows + orr] = tile[tx][ty + 8 * u];
  }
}

}  // namespace mms
#endif  // MMS_PANEL_GEMM_H_
