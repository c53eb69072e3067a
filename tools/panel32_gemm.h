// tools/panel32_gemm.h -- round 3 EXPERIMENT (not part of the product; kept for tools/panel32bench.hip): the panel fp32-MFMA GEMM re-cut for TWO co-resident workgroups per CU.
//
//   C[M x N] = A[M x K] . B[K x N]      (same PanelArgs, same semantics and k order class as panel_gemm.h)
//
// What round 2's kernel (one 64-row panel per CU, ONE compute wave per SIMD beside a loader wave) left on the
// table, measured (gpurun_out/r2/panelbench_*, profiles/r02_panel_gemm_pmc.txt): its main loop ran 29.5 us for
// 21.5 us of MFMA issue (at the 2.15 GHz the chip holds under this load) -- a single wave per SIMD stops the matrix
// pipe at every tile transition, barrier and LDS wait it meets -- while tools/mfmarate.hip shows two INDEPENDENT
// waves per SIMD sharing the pipe at 147 TFLOP/s.  Here:
//   * a workgroup is FOUR waves, one per SIMD, and owns a 32-row panel and all N columns: 2 row blocks of 16 x 2
//     column halves (10 + 9 column tiles of 16 at N = 300).  Two workgroups are resident per CU (75 KB of LDS each,
//     two waves per SIMD at up to 256 registers), each with its own barriers, and their MFMA streams interleave on
//     every SIMD; the co-resident workgroup takes the halves the other way round, so every SIMD carries 19 tiles.
//     (Six-wave workgroups -- four compute + two loader waves -- were built first and did NOT co-reside: the
//     dispatcher puts a workgroup's waves on the SIMDs 2,2,1,1, and twice that is 4 x 136 registers on SIMD 0.)
//   * there are no loader waves: every wave issues its quarter of each tile's LDS-DMAs (global_load_lds_dwordx4)
//     right behind the tile barrier.  Round 2 could not afford that -- its DMA address arithmetic was VALU work,
//     and VALU instructions queue behind the MFMA stream (~32 cycles each, ~300 cycles per DMA measured) -- so here
//     a FULL tile issues with none: the tile origin is a scalar base (SALU), the per-lane byte offsets are
//     tile-invariant registers, M0 moves are SALU;
//   * 16-deep k-tiles in a three-stage LDS ring of 25 KB stages, one counted `s_waitcnt vmcnt` + one raw barrier
//     per tile.  The barrier of tile T+1 sits in the MIDDLE of tile T (between its steps 1 and 2): behind it the
//     wave issues tile T+3's DMAs and fetches the first operands of tile T+1 beside the MFMAs of steps 2-3, in front
//     of it those of steps 2-3 beside the MFMAs of steps 0-1;
//   * B reaches LDS either k-major ([16 k][N + 4], the forward's W) or N-MAJOR ([N][16 k], 16-byte chunks
//     XOR-swizzled by (n >> 2) & 3 so that the 8-byte operand reads of 16 lanes cover 32 distinct banks): the
//     backward's dq = diag(dT) A W^T reads W^T's k-tiles straight out of W -- the per-step transpose of W is gone;
//   * the epilogue leaves from the accumulator registers (64-byte row segments per 16-column tile), with the row
//     scale and the SimMatrix row dot applied on the way: the panel's rows of Y arrive by DMA in the two tile slots
//     behind the last tile (slots the ring protocol fills anyway), and the two column halves of a row meet in LDS in
//     a fixed order.  The side job (da = diag(dT) QW) is spread over the main loop, one row pass at a time.
// Reference semantics: those of the callers (bilinear.hip); results are inside the 1e-5 contract of these
// BLAS-backed products (sim_matrix_layer.cpp:53-95, sim_cross_layer.cpp:140-161, 251-305).
#ifndef MMS_PANEL32_GEMM_H_
#define MMS_PANEL32_GEMM_H_

#include "panel_gemm.h"

namespace mms {

typedef float p32_v2f __attribute__((ext_vector_type(2)));

template <int NT, bool A_KC, bool B_NM>
struct P32Geom {
  static constexpr int LD = 16 * NT + 4;                  // k-major B row stride (floats)
  static constexpr int B_CPR = LD / 4;                    // 16-byte chunks per k-major B row
  static constexpr int B_NCH = B_NM ? 16 * NT * 4 : 16 * B_CPR;
  static constexpr int NBW = ((B_NCH + 63) / 64 + 3) / 4;   // B DMA instructions per wave per tile
  static constexpr int NAW = 1;                            // A DMA instructions per wave per tile (image: 4 KB)
  static constexpr int NPT = NBW + NAW + (A_KC ? 0 : 1);   // DMA instructions per wave per tile, all kinds
  static constexpr int B_F = NBW * 4 * 256;               // floats
  static constexpr int A_F = NAW * 4 * 256;
  static constexpr int S_F = 256;                          // kscale slot (!A_KC); keeps stages 1 KB multiples
  static constexpr int STAGE_F = B_F + A_F + S_F;
  static constexpr int X_F = 64;                           // row-dot exchange: 2 row blocks x 16 rows (+ pad)
  static constexpr size_t kLdsBytes = (3 * (size_t)STAGE_F + X_F) * sizeof(float);
  static constexpr int NT0 = (NT + 1) / 2, NT1 = NT / 2;   // column tiles of the two halves
  static constexpr int YC = LD / 4, NYI = (32 * YC + 63) / 64, YSLOT = 4 * NPT;   // the Y panel as 1-KB DMAs
  static_assert(NPT < 60, "vmcnt is a 6-bit counter");
  static_assert(2 * kLdsBytes <= 160 * 1024, "two workgroups must fit one CU's LDS");
  static_assert(!A_KC || NYI <= 2 * YSLOT, "the Y panel must fit the two tile slots behind the last tile");
};

constexpr int P32_LDT = 36;            // !A_KC image: A^T [16 k][32 rows + 4]

#ifdef MMS_P32_STAMPS     // dev-only (tools/panel32bench.hip): per-workgroup wall-clock / shader-clock stamps
__device__ unsigned long long* p32_stamp_buf = nullptr;
#define P32_STAMP(k, v)                                                                       \
  do {                                                                                        \
    if (p32_stamp_buf && threadIdx.x == 0) p32_stamp_buf[(size_t)blockIdx.x * 16 + (k)] = (v); \
  } while (0)
#else
#define P32_STAMP(k, v) do {} while (0)
#endif

// One workgroup's work: workgroup number `wg` (of its product's grid) and batch entry `bt`.
template <int NT, bool A_KC, bool B_NM>
__device__ __forceinline__ void panel32_body(const PanelArgs& p, const int wg, const int bt) {
  using G = P32Geom<NT, A_KC, B_NM>;
  extern __shared__ float4 p32_lds4[];
  float* lds = reinterpret_cast<float*>(p32_lds4);
  constexpr int LD = G::LD, NBW = G::NBW, NPT = G::NPT, STAGE_F = G::STAGE_F;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;

  const int t = threadIdx.x, lane = t & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

  // ---- which panel / k-chunk / batch entry (as panel_gemm.h, 32-row panels) ------------------------------------
  int rb, ks = 0;
  if (p.ksplit > 1) {
    const int id = wg, xcd = id & 7, local = id >> 3;
    ks = (local / p.row_blocks) * 8 + xcd;
    rb = local % p.row_blocks;
    if (ks >= p.ksplit) return;                       // whole workgroup, before any barrier
  } else {
    rb = wg;
  }
  const float* Bb = p.B + (long long)bt * p.b_b;
  const int R0 = rb * 32;
  const int kbeg = ks * p.kchunk;
  const int kend = p.ksplit > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const int ntiles = (kend - kbeg + 15) >> 4;           // nseg == 1 (panel32_eligible)
  float* xch = lds + 3 * STAGE_F;                     // row-dot exchange
  P32_STAMP(0, __builtin_amdgcn_s_memrealtime());

  // ====================================== this wave's quarter of the DMAs ==========================================
  // DMA instruction j = wave + 4 jj of a tile covers chunks 64 j .. 64 j + 63 of its image.  Per-lane source
  // offsets are tile-invariant; -1 = the slot holds nothing (pad chunk, row or column outside the matrix).
  int boff[NBW], bkk[NBW], aoff, akk;
#pragma unroll
  for (int jj = 0; jj < NBW; ++jj) {
    const int c = (wave + 4 * jj) * 64 + lane;
    if (B_NM) {
      const int n = c >> 2, koff = 4 * ((c & 3) ^ ((n >> 2) & 3));
      const bool ok = n < p.N;
      boff[jj] = ok ? (int)(n * p.ldb) + koff : -1;
      bkk[jj] = koff;
    } else {
      const int kk = c / G::B_CPR, cc = c - kk * G::B_CPR;
      const bool ok = kk < 16 && 4 * cc < p.N;
      boff[jj] = ok ? (int)(kk * p.ldb) + 4 * cc : -1;
      bkk[jj] = kk;
    }
  }
  {
    const int c = wave * 64 + lane;
    if (A_KC) {
      const int row = c >> 2, koff = 4 * ((c & 3) ^ ((row >> 2) & 3));
      const bool ok = row < 32;
      aoff = ok ? (int)(min(R0 + row, p.M - 1) * p.lda) + koff : -1;
      akk = koff;
    } else {
      const int kk = c / 9, cc = c - kk * 9;
      const bool ok = kk < 16 && cc < 8 && R0 + 4 * cc < p.M;       // M % 4 == 0: a chunk is all-in or all-out
      aoff = ok ? (int)(kk * p.lda) + R0 + 4 * cc : -1;
      akk = kk;
    }
  }
  // A FULL tile issues without a VALU instruction (header): scalar tile origin + tile-invariant byte offsets (slots
  // that hold nothing fetch the tile's first bytes).  Only a partial tile (the last one) takes the per-slot selects.
  unsigned bvo[NBW];
#pragma unroll
  for (int jj = 0; jj < NBW; ++jj) bvo[jj] = boff[jj] >= 0 ? (unsigned)boff[jj] * 4u : 0u;
  const unsigned avo = aoff >= 0 ? (unsigned)aoff * 4u : 0u;
  const unsigned svo = (unsigned)(lane & 15) * 4u;
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
  auto issue_tile = [&](int stage, int k0) {
    const unsigned sb = lds_base + (unsigned)(stage * STAGE_F) * 4u;
    const float* Bs = Bb + (B_NM ? (long long)k0 : (long long)k0 * p.ldb);
    const float* As = p.A + (A_KC ? (long long)k0 : (long long)k0 * p.lda);
    if (k0 + 16 <= kend) {                              // wave-uniform
#pragma unroll
      for (int jj = 0; jj < NBW; ++jj) dma16s(Bs, bvo[jj], sb + (unsigned)(wave + 4 * jj) * 1024u);
      dma16s(As, avo, sb + (unsigned)G::B_F * 4u + (unsigned)wave * 1024u);
      if (!A_KC) dma4s(p.kscale + k0, svo, sb + (unsigned)(G::B_F + G::A_F) * 4u);
      return;
    }
#pragma unroll
    for (int jj = 0; jj < NBW; ++jj) {
      const bool ok = boff[jj] >= 0 && k0 + bkk[jj] < kend;
      pg_dma16(ok ? Bs + boff[jj] : Bb, sb + (unsigned)(wave + 4 * jj) * 1024u);
    }
    {
      const bool ok = aoff >= 0 && k0 + akk < kend;
      pg_dma16(ok ? As + aoff : p.A, sb + (unsigned)G::B_F * 4u + (unsigned)wave * 1024u);
    }
    if (!A_KC) {                                        // the tile's 16 kscale values (every wave: same bytes, same place)
      const int kc = min(k0 + (lane & 15), kend - 1);
      pg_dma4(p.kscale + kc, sb + (unsigned)(G::B_F + G::A_F) * 4u);
    }
  };
  // Row dot (SimMatrix forward): the panel's 32 rows of Y reach LDS in the two tile slots BEHIND the last tile as a
  // chunk-linear image [32 rows][LD / 4 chunks] cut into 1-KB instructions: instruction i of the image is DMA
  // u = i mod YSLOT of slot i / YSLOT, at byte u * 1024 of that slot's stage.  (ntiles >= 2: panel32_eligible.)
  constexpr int YC = G::YC, NYI = G::NYI, YSLOT = G::YSLOT;
  const bool with_y = A_KC && p.Y != nullptr;
  unsigned yvo[2][NPT];
  if (with_y) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int jj = 0; jj < NPT; ++jj) {
        const int i = sl * YSLOT + jj * 4 + wave, c = i * 64 + lane;
        const int row = c / YC, cc = c - row * YC;
        const bool ok = i < NYI && row < 32 && 4 * cc < p.N;
        yvo[sl][jj] = ok ? (unsigned)(min(R0 + row, p.M - 1) * (int)p.ldy + 4 * cc) * 4u : 0u;
      }
  }
  // tile number i into its stage: a real tile, a slot of the Y panel, or (to keep the counted waits uniform) the
  // first tile again into a stage nobody reads any more
  auto issue = [&](int i) {
#if defined(MMS_P32_ABLATE) && MMS_P32_ABLATE == 1     // dev-only timing ablation (tools/panel32bench.hip): no DMAs behind the prologue
    if (i >= 3) return;
#endif
    const int stage = i % 3;
    if (i < ntiles) {
      issue_tile(stage, kbeg + 16 * i);
    } else if (with_y && i < ntiles + 2) {
      const unsigned sb = lds_base + (unsigned)(stage * STAGE_F) * 4u;
      const bool second = i > ntiles;
#pragma unroll
      for (int jj = 0; jj < NPT; ++jj) dma16s(p.Y, second ? yvo[1][jj] : yvo[0][jj], sb + (unsigned)(jj * 4 + wave) * 1024u);
    } else {
      issue_tile(stage, kbeg);
    }
  };
  // side job (panel_gemm.h): side_out(i,:) = side_scale[i] * side_in(i,:) for the panel's rows; this wave's eight
  // rows R0 + 8 wave + g + 4 rr, one pass rr in flight at a time, requested by hand in FRONT of an iteration's DMAs
  // (the loop's counted wait then covers it) and stored a few iterations later: spread over the main loop.
  constexpr int NCC = (4 * NT + 15) / 16, NRR = 2;
  const int SC = p.side_in ? p.side_cols >> 2 : 0;
  const int srow0 = R0 + 8 * wave + g;
  pg_v4f sx[NCC];
  float ssc = 0.f;
  auto side_load = [&](int rr) {
    const int growc = min(srow0 + 4 * rr, p.M - 1);
    asm volatile("global_load_dword %0, %1, off" : "=v"(ssc) : "v"(p.side_scale + growc) : "memory");
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
      const int c4 = r + 16 * cc;
      const float* src = p.side_in + (long long)growc * p.side_ld + 4 * (c4 < SC ? c4 : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sx[cc]) : "v"(src) : "memory");
    }
  };
  auto side_pin = [&]() {                              // only behind a wait that covers the requests above
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) asm volatile("" : "+v"(sx[cc]));
    asm volatile("" : "+v"(ssc));
  };
  auto side_store = [&](int rr) {
    const int grow = srow0 + 4 * rr;
    const float sc = ssc;
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
      const int c4 = r + 16 * cc;
      if (c4 < SC && grow < p.M)
        __builtin_nontemporal_store(sc * sx[cc], reinterpret_cast<pg_v4f*>(p.side_out + (long long)grow * p.side_ld + 4 * c4));
    }
  };
  const int SP = ntiles / (NRR + 1) > 0 ? ntiles / (NRR + 1) : 1, SH = (SP + 1) / 2;
  int s_loaded = 0, s_stored = 0;
  auto side_step = [&](int T) {                       // iteration T of the main loop, in front of its DMAs
    if (s_stored < s_loaded && T >= SP * s_stored + SH) { side_pin(); side_store(s_stored); ++s_stored; }
    if (s_loaded < NRR && s_loaded == s_stored && T >= SP * s_loaded) { side_load(s_loaded); ++s_loaded; }
  };

  // =========================================== the products ========================================================
  const int flip = (wg >> 8) & 1;                     // the co-resident workgroup takes the halves the other way round
  const int rbw = wave & 1, ch = (wave >> 1) ^ flip;
  const int i0 = R0 + 16 * rbw;                       // (a row block past the matrix multiplies padding and stores nothing)
  auto body = [&](auto ntw_tag, auto tt0_tag) {
    constexpr int NTW = decltype(ntw_tag)::value, TT0 = decltype(tt0_tag)::value;
    pg_v4f acc[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) acc[tt] = (pg_v4f){0.f, 0.f, 0.f, 0.f};
    float bh[2][2][NTW];                              // [half][step of the half][column tile]: B operands of steps
                                                      // (0,1) and (2,3); scalars, so that the pairs a k-major
                                                      // ds_read2_b32 returns (two tiles of ONE step) and the pairs an
                                                      // n-major ds_read_b64 returns (two steps of ONE tile) both land
                                                      // where they are used, without copies
    pg_v4f av, an;
    const int sw = (r >> 2) & 3;                      // the image swizzle of this lane's row / column
    auto read_a = [&](pg_v4f& dst, int stage) {
      const float* At = lds + stage * STAGE_F + G::B_F;
      if (A_KC) {
        dst = *reinterpret_cast<const pg_v4f*>(At + (rbw * 16 + r) * 16 + 4 * (g ^ sw));
      } else {
        const float* sc = At + G::A_F;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = At[(4 * g + i) * P32_LDT + rbw * 16 + r] * sc[4 * g + i];
      }
    };
    auto read_half = [&](float (&dst)[2][NTW], int stage, int h) {
      const float* Bs = lds + stage * STAGE_F;
      if (B_NM) {
        const float* bn = Bs + (16 * TT0 + r) * 16 + 4 * (g ^ sw) + 2 * h;
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const p32_v2f x = *reinterpret_cast<const p32_v2f*>(bn + 256 * tt);
          dst[0][tt] = x[0];
          dst[1][tt] = x[1];
        }
      } else {
        const float* bn = Bs + (4 * g + 2 * h) * LD + 16 * TT0 + r;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int tt = 0; tt < NTW; ++tt) dst[i][tt] = bn[i * LD + 16 * tt];
      }
    };
    auto mfma_half = [&](const pg_v4f& a, const float (&b)[2][NTW], int h) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * h + i], b[i][tt], acc[tt], 0, 0, 0);
    };
    auto interleave = [&]() {                          // one LDS read in the shadow of each of the first MFMAs
#pragma unroll
      for (int u = 0; u < NTW + 4; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTW, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NTW + 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // k values of a partial tile past kend are zeroed in BOTH operands (the LDS rows behind them hold whatever the
    // clamped DMA fetched); valid k of a tile are 4 g + i < rem, rem a multiple of 4: a per-lane property of g.
    auto mask_a = [&](pg_v4f& a, int rem) {
      const bool ok = 4 * g < rem;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = ok ? a[i] : 0.f;
    };
    auto mask_b = [&](float (&b)[2][NTW], int rem) {
      const bool ok = 4 * g < rem;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) b[i][tt] = ok ? b[i][tt] : 0.f;
    };
    // Tile T, whose first operands (av, bh[0]) are already in registers.  MASK: a partial tile (rem < 16 valid k).
    // !LAST: tile T+1 exists.  Its barrier B_{T+1} is taken between this tile's halves: in front of it this wave's
    // DMAs of tile T+1 have landed (counted wait: only tile T+2's may still be in flight) and all its LDS reads of
    // tile T are complete; behind it every wave's are, so tile T's stage takes the DMAs of tile T+3, and tile T+1's
    // first operands are fetched beside the MFMAs of the second half.
    auto tile = [&](auto mask_tag, auto last_tag, int T, int rem) {
      constexpr bool MASK = decltype(mask_tag)::value, LAST = decltype(last_tag)::value;
      const int st = T % 3, nst = (T + 1) % 3;
      if (MASK) { mask_a(av, rem); mask_b(bh[0], rem); }
      read_half(bh[1], st, 1);
      mfma_half(av, bh[0], 0);
      interleave();
      if (MASK) mask_b(bh[1], rem);
      if (!LAST) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPT) : "memory");   // B_{T+1}
        if (A_KC) {
          // issued by hand, FIRST behind the barrier, into registers of its own: left to the compiler, `an` is
          // coalesced with `av` and its read can only issue behind the last MFMA of the tile (a full LDS round trip
          // in front of the next tile's first MFMA).  LDS operations return in order and this one is the oldest of
          // the phase, so every wait the compiler counts for its own reads covers it too.
          const unsigned addr = lds_base + (unsigned)(nst * STAGE_F + G::B_F + (rbw * 16 + r) * 16 + 4 * (g ^ sw)) * 4u;
          asm volatile("ds_read_b128 %0, %1" : "=&v"(an) : "v"(addr) : "memory");
        } else {
          read_a(an, nst);
        }
        read_half(bh[0], nst, 0);
        if (p.side_in) side_step(T);
        issue(T + 3);
      }
      mfma_half(av, bh[1], 1);
      interleave();
      if (!LAST) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an) :: "memory");
        av = an;
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) asm volatile("" : "+v"(bh[0][0][tt]), "+v"(bh[0][1][tt]));   // keep the prefetch in THIS tile
        asm volatile("" : "+v"(av));
      }
    };

    if (ntiles > 0) {
      issue(0);
      issue(1);
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPT) : "memory");         // B_0: tile 0 has landed
      issue(2);
      P32_STAMP(2, __builtin_amdgcn_s_memrealtime());
      P32_STAMP(4, __builtin_amdgcn_s_memtime());
      read_a(av, 0);
      read_half(bh[0], 0, 0);
    }
    // nseg == 1 (panel32_eligible): the tiles are nfull full ones and at most one partial tile at the end.  ONE loop
    // body (a full tile that is followed by another tile); the last tile -- full or partial -- is peeled.
    {
      using TT_ = std::true_type; using FF_ = std::false_type;
      const int nfull = (kend - kbeg) >> 4;
      const int rem = (kend - kbeg) & 15;
      const int nloop = rem ? nfull : nfull - 1;
      int T = 0;
      for (; T < nloop; ++T) tile(FF_{}, FF_{}, T, 16);
      if (rem) tile(TT_{}, TT_{}, T, rem);
      else if (nfull > 0) tile(FF_{}, TT_{}, T, 16);
    }
    P32_STAMP(3, __builtin_amdgcn_s_memrealtime());
    P32_STAMP(5, __builtin_amdgcn_s_memtime());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing of this wave is in flight towards LDS any more
    if (p.side_in) {                                  // what the loop was too short for
      side_pin();
      if (s_stored < s_loaded) { side_store(s_stored); ++s_stored; }
      for (int rr = s_loaded; rr < NRR; ++rr) {
        side_load(rr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        side_pin();
        side_store(rr);
      }
    }

    // ---- epilogue: straight from the accumulators -----------------------------------------------------------------
    float* Cg = p.C ? p.C + (long long)bt * p.c_b + (long long)ks * p.c_ks : nullptr;
    float rs[4] = {1.f, 1.f, 1.f, 1.f};
    if (p.rowscale) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rs[j] = p.rowscale[min(i0 + 4 * g + j, p.M - 1)];
    }
    float yv[NTW][4];
    if (with_y) {
      asm volatile("s_barrier" ::: "memory");                                        // B_Y: every wave's share of Y has landed
      const float* ys[2] = {lds + (ntiles % 3) * STAGE_F, lds + ((ntiles + 1) % 3) * STAGE_F};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cb = (rbw * 16 + 4 * g + j) * YC + 4 * TT0 + (r >> 2);
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const int c = cb + 4 * tt, i = c >> 6;
          const float* img = i >= YSLOT ? ys[1] + (i - YSLOT) * 256 : ys[0] + i * 256;
          yv[tt][j] = 16 * (TT0 + tt) + r < p.N ? img[(c & 63) * 4 + (r & 3)] : 0.f;   // columns past N: neither the image
        }                                                                              // nor the accumulator holds anything
      }
    }
    if (p.rowscale) {
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tt][j] = rs[j] * acc[tt][j];
    }
    if (Cg) {
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        const int col = 16 * (TT0 + tt) + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = i0 + 4 * g + j;
          if (row < p.M && col < p.N) {
            float* dst = Cg + (long long)row * p.ldc + col;
            if (p.stream_c) __builtin_nontemporal_store(acc[tt][j], dst);
            else *dst = acc[tt][j];
          }
        }
      }
    }
    if (with_y) {
      float dot[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float d = 0.f;
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) d += acc[tt][j] * yv[tt][j];      // this half's columns 16 tt + r, tt ascending
        d = dpp_add<0xB1, 0xf>(d);                    // ... and the four steps that sum a row of 16 lanes
        d = dpp_add<0x4E, 0xf>(d);
        d = dpp_add<0x141, 0xf>(d);
        d = dpp_add<0x140, 0xf>(d);
        dot[j] = d;
      }
      if (ch == 1 && r == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xch[rbw * 16 + 4 * g + j] = dot[j];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                // B_dot
      if (ch == 0 && r == 0) {
        const float rdb = p.rd_bias ? p.rd_bias[bt] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = i0 + 4 * g + j;
          if (row < p.M) {
            const float d = dot[j] + xch[rbw * 16 + 4 * g + j];                       // columns ascending: half 0, then half 1
            p.rowdot[(long long)bt * p.rd_b + (long long)row * p.rd_stride] = p.rd_bias ? (rdb + d) : d;
          }
        }
      }
    }
    P32_STAMP(6, __builtin_amdgcn_s_memrealtime());
  };
  if (ch == 0) body(std::integral_constant<int, G::NT0>{}, std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, G::NT1>{}, std::integral_constant<int, G::NT0>{});
}

template <int NT, bool A_KC, bool B_NM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void panel32_kernel(PanelArgs p) {
  panel32_body<NT, A_KC, B_NM>(p, (int)blockIdx.x, (int)blockIdx.y);
}

// TWO independent products in ONE launch -- the SimMatrix backward's dq = diag(dT) A W^T (n-major B, with the da side
// job) and its split-K dW = Q^T diag(dT) A.  Launched one after the other, each product's workgroups start together,
// run in phase and end together: prologue (first tile from HBM), DMA bursts and epilogue (19.7 MB of stores) of the
// two resident workgroups of a CU coincide.  Here blocks of 256 consecutive workgroup ids alternate between the
// products, so a CU holds one workgroup of EACH; they differ in length and drift apart, and whatever one of them is
// waiting for, the other's MFMAs fill the matrix pipe.  nwg1 / nwg2: the products' own grid sizes.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void panel32_pair_kernel(PanelArgs p1, PanelArgs p2,
                                                                                                   int nwg1, int nwg2) {
  const int b = (int)blockIdx.x, c = b >> 8, local = ((c >> 1) << 8) | (b & 255);
  if ((c & 1) == 0) {
    if (local < nwg1) panel32_body<NT, true, true>(p1, local, 0);
  } else {
    if (local < nwg2) panel32_body<NT, false, false>(p2, local, 0);
  }
}

// ------------------------------------------------ host side -----------------------------------------
// Eligibility on top of panel_eligible(): the 32-row kernel serves the big single-segment products (cfg 3 and its
// like); everything else keeps panel_gemm.h's kernel.  b_nm: B is handed over as B(k,n) = B[n*ldb + k].
inline bool panel32_eligible(const PanelArgs& p, bool a_kc, bool b_nm) {
  if (p.N > 304 || p.N < 4 || !pg_mult4(p.N) || !pg_mult4(p.ldb) || !aligned16(p.B) || !pg_mult4(p.b_b) || !pg_mult4(p.b_seg))
    return false;
  if (!pg_mult4(p.lda) || !aligned16(p.A) || !pg_mult4(p.K) || !pg_mult4(p.a_seg)) return false;
  if (!a_kc && (!pg_mult4(p.M) || !p.kscale || b_nm)) return false;
  if (p.nseg != 1 || (p.ksplit > 1 && (p.kchunk & 15))) return false;
  if (p.Y && (!a_kc || p.ksplit != 1 || p.K < 32 || !pg_mult4(p.ldy) || !aligned16(p.Y) || (long long)p.M * p.ldy >= (1LL << 29)))
    return false;                                     // the Y panel rides in the two tile slots behind the last tile
  if (p.side_in && (p.ksplit != 1 || p.nb != 1 || !pg_mult4(p.side_cols) || p.side_cols > 304 || !pg_mult4(p.side_ld) ||
                    !aligned16(p.side_in) || !aligned16(p.side_out) || !p.side_scale))
    return false;
  if ((long long)p.M * p.lda >= (1LL << 29) || 304LL * p.ldb >= (1LL << 29)) return false;   // 32-bit byte offsets
  const long long rb32 = (p.M + 31) / 32;
  return rb32 * p.ksplit * p.nb >= 192;               // below that the chip is not filled twice over
}

template <int NT, bool A_KC, bool B_NM>
inline void panel32_launch_t(PanelArgs p, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds = P32Geom<NT, A_KC, B_NM>::kLdsBytes;
  auto kern = panel32_kernel<NT, A_KC, B_NM>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  p.row_blocks = (p.M + 31) / 32;
  const unsigned gx = p.ksplit > 1 ? 8u * ((p.ksplit + 7) / 8) * p.row_blocks : (unsigned)p.row_blocks;
  hipLaunchKernelGGL(kern, dim3(gx, p.nb), dim3(256), lds, s, p);
}

inline void panel32_launch(const PanelArgs& p, bool a_kc, bool b_nm, hipStream_t s) {
#define MMS_P32(NT)                                                       \
  do {                                                                    \
    if (!a_kc) panel32_launch_t<NT, false, false>(p, s);                  \
    else if (b_nm) panel32_launch_t<NT, true, true>(p, s);                \
    else panel32_launch_t<NT, true, false>(p, s);                         \
  } while (0)
  if (p.N <= 112) MMS_P32(7);
  else if (p.N <= 208) MMS_P32(13);
  else MMS_P32(19);
#undef MMS_P32
}

// dq product (A_KC, n-major B; nb == 1, ksplit == 1) + dW product (!A_KC, split-K) in one launch
inline void panel32_pair_launch(PanelArgs p1, PanelArgs p2, hipStream_t s) {
  p1.row_blocks = (p1.M + 31) / 32;
  p2.row_blocks = (p2.M + 31) / 32;
  const int nwg1 = p1.row_blocks;
  const int nwg2 = p2.ksplit > 1 ? 8 * ((p2.ksplit + 7) / 8) * p2.row_blocks : p2.row_blocks;
  const int ch1 = (nwg1 + 255) / 256, ch2 = (nwg2 + 255) / 256;
  const unsigned grid = 256u * 2u * (unsigned)(ch1 > ch2 ? ch1 : ch2);
#define MMS_P32P(NT)                                                                                              \
  do {                                                                                                            \
    static bool attr_set = false;                                                                                 \
    auto kern = panel32_pair_kernel<NT>;                                                                          \
    const size_t lds = P32Geom<NT, true, true>::kLdsBytes > P32Geom<NT, false, false>::kLdsBytes                  \
                           ? P32Geom<NT, true, true>::kLdsBytes : P32Geom<NT, false, false>::kLdsBytes;           \
    if (!attr_set) {                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p1, p2, nwg1, nwg2);                                  \
  } while (0)
  const int n = p1.N > p2.N ? p1.N : p2.N;
  if (n <= 112) MMS_P32P(7);
  else if (n <= 208) MMS_P32P(13);
  else MMS_P32P(19);
#undef MMS_P32P
}

// split-K for the 32-row kernel: all row blocks of a chunk on one XCD (they stream the same rows of B through that
// XCD's L2), at most TWO workgroups per CU = 64 per XCD, chunks a multiple of the 16-deep k-tile.
inline int panel32_pick_ksplit(int M, int nb, int K, int* kchunk) {
  const int rb32 = (M + 31) / 32;
  int per_xcd = 64 / (rb32 * nb > 0 ? rb32 * nb : 1);
  if (per_xcd < 1) per_xcd = 1;
  int want = 8 * per_xcd;
  const int maxs = (K + 63) / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  int chunk = (K + want - 1) / want;
  chunk = (chunk + 15) / 16 * 16;
  *kchunk = chunk;
  return (K + chunk - 1) / chunk;
}

}  // namespace mms
#endif  // MMS_PANEL32_GEMM_H_
