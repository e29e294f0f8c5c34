// Batched-path GEMM (several songs per handle, M = passes x songs x T >= 2048 rows): 128-row tiles with TWO blocks
// resident per CU.
//
// Why.  The phase stamps of the batched step (tools/diag/phase_times.py, profiles/r03p_phase_times_b8.txt) show what
// a 128 x 128 tile of gemm_h16.h is made of when every CU owns four of them in a row: 2.4 us until the first K-tile
// has landed, 12.1 us of main loop, 6.5 us of epilogue (accumulators -> slab -> row scale, bias, gated GELU, hi / lo
// split, stores) -- 21 us, of which the loop that feeds the MFMA pipe is 58 %, and nothing overlaps: the two-plane
// K-tile of 64 is 64 KB, the two-deep ring 128 KB, so ONE block fits a CU and its prologue, loop and epilogue run
// back to back, four times per launch.  Here a K-tile is 32 deep (64-byte rows: 32 KB per stage at 128 x 128), the
// ring two deep, the epilogue runs in two 64-row halves through a 34 KB slab and the bias / row-statistics region is
// sized at launch for the D / 32 partials there are: 77 KB per block at D = 768, so TWO blocks (8 waves, 2 per SIMD,
// <= 256 registers each) share a CU, and one block's epilogue and prologue run under the other's main loop.  The loop
// itself is the plain one of the K = 32 experiment (tools/ubench/gemm_h16_k32.h: fragments of tile kt+1 read into a
// second register set under the MFMAs of tile kt, one barrier per tile); what the second block hides is exactly what
// that experiment lacked.
//
// Layout.  A stage holds, per plane, [BM rows | BN rows] of 64 bytes.  LDS-DMA (global_load_lds, 1 KiB = 16 rows per
// wave-instruction) writes rows linearly, so the bank swizzle sits on the SOURCE side as in gemm_h16.h: 16-byte chunk
// c of row r is stored at position c ^ G[(r >> 2) & 3], G = {0, 3, 2, 1}.  A fragment read (ds_read_b128: lane = (row
// l & 15, chunk l >> 4)) then touches, in each of the instruction's four 16-lane groups, sixteen different
// (row mod 4, position) pairs = all 64 banks once (MI355X_MICROARCH.md, LDS table).
//
// Epilogues are those of gemm_h16.h, called per 64-row half exactly as a 64-row tile would call them (prefetch of the
// aux rows, row statistics, run), tile map and range flag likewise.  No weight prefetch from these launches: with
// M >= 2048 a weight tile is re-read by 16+ row tiles and the second block covers the first touch.
#pragma once
#include "gemm_h16.h"

namespace msd {

constexpr int kPairBK = 32;   // K elements per tile: one MFMA K-step, 64-byte rows

// byte offset of chunk `c` (0..3) of `row` in a [rows][32] 16-bit LDS tile
__device__ __forceinline__ int lds_pair_off(int row, int c) { return row * 64 + ((c ^ ((0 - (row >> 2)) & 3)) << 4); }

// ssq partials per row of an epilogue that carries a RowScale (0 for the others): sizes the aux region at launch
template <class Epi>
inline auto epi_rowscale_tiles(const Epi& e, int) -> decltype(e.rsc.tiles) { return e.rsc.ssq ? e.rsc.tiles : 0; }
template <class Epi>
inline int epi_rowscale_tiles(const Epi&, long) { return -1; }   // no RowScale member: the epilogue's static aux size

template <int NP, int BM, int BN, class Epi>
__global__ void __launch_bounds__(256, 2) gemm_h16_pair_kernel(GemmParams p, Epi epi) {
  static_assert(NP == 2 && BM == 128, "the batched tiles exist for the two-plane modes, 128 rows");
  constexpr int NS = 2;
  constexpr int WM = BM / 2, WN = BN / 2;           // 2 x 2 waves: wave (wm, wn) owns rows wm * 64.., columns wn * WN..
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  constexpr int STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  constexpr int ROWBLK = (BM + BN) / 16;            // 16-row DMA pieces per plane
  constexpr int T = NP * ROWBLK;                    // DMA instructions per K-tile, all waves
  static_assert(T % 4 == 0, "a whole number of DMA instructions per wave");
  constexpr int PW = T / 4;                         // per wave
  constexpr int HM = BM / 2;                        // rows per epilogue half
  constexpr int LDS_LD = BN + kSlabPad;
  static_assert((HM * LDS_LD + HM) * 4 <= NS * STAGE_BYTES, "half-tile slab must fit the operand LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware tile map of gemm_h16_dma_kernel
  const int nbm = p.M / BM, nbn = p.N / BN;
  const int RX = p.xcd_rows, CX = 8 / RX;
  const int xcd = blockIdx.x & 7, tt = blockIdx.x >> 3;
  const int nbm_x = (nbm + RX - 1) / RX, nbn_x = (nbn + CX - 1) / CX;
  int bm, bn;
  if (p.xcd_walk_n) {
    bm = (tt / nbn_x) * RX + xcd / CX; bn = (tt % nbn_x) * CX + xcd % CX;
  } else {
    bm = (tt % nbm_x) * RX + xcd / CX; bn = (tt / nbm_x) * CX + xcd % CX;
  }
  if (bn >= nbn || bm >= nbm) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = bm * BM, n0 = bn * BN;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // A DMA piece is one 16-row block of one plane of one operand; this lane fetches row r = lane >> 2 of the block,
  // source chunk (lane & 3) ^ G[(r >> 2) & 3]
  const int r16 = lane >> 2, csrc = (lane & 3) ^ ((0 - (r16 >> 2)) & 3);
  // Addresses as wave-uniform 64-bit bases (scalar registers) plus ONE 32-bit lane offset per operand: per-piece
  // 64-bit lane pointers cost 2 PW vector registers, which this kernel (accumulators + two fragment sets at 256
  // registers, two waves per SIMD) does not have.
  const unsigned offA = (unsigned)(r16 * p.lda + csrc * 8) * 2u, offB = (unsigned)(r16 * p.ldb + csrc * 8) * 2u;
  // pieces 0 .. PA-1 of a wave are A pieces (a = wave + 4 i of the NP * BM / 16), the rest B pieces: which operand a
  // piece belongs to is a compile-time property of its index
  constexpr int PA = NP * BM / 64, PB = NP * BN / 64;
  static_assert(PA + PB == PW, "pieces per wave");
  const char* gbase[PW];
  int ldst[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    if (i < PA) {
      const int a = wave + 4 * i, pl = a / (BM / 16), blk = a % (BM / 16);
      gbase[i] = reinterpret_cast<const char*>(p.A[pl] + (size_t)(m0 + blk * 16) * p.lda);
      ldst[i] = pl * A_BYTES + blk * 1024;
    } else {
      const int b = wave + 4 * (i - PA), pl = b / (BN / 16), blk = b % (BN / 16);
      gbase[i] = reinterpret_cast<const char*>(p.B[pl] + (size_t)(n0 + blk * 16) * p.ldb);
      ldst[i] = NP * A_BYTES + pl * B_BYTES + blk * 1024;
    }
  }
#define MSD_P_ISSUE1(KT, BUF, I)                                                                          \
  __builtin_amdgcn_global_load_lds(                                                                       \
      (gptr_t)(gbase[I] + (size_t)((KT) * (kPairBK * 2)) + (size_t)((I) < PA ? offA : offB)),             \
      (lptr_t)(smem + (BUF) * STAGE_BYTES + ldst[I]), 16, 0, 0);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / kPairBK;
  // ---- prologue: both ring slots in flight, then the epilogue's aux rows of the two halves -----------------------
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < nk) {
#pragma unroll
      for (int i = 0; i < PW; ++i) MSD_P_ISSUE1(s, s, i)
    }
  char* const aux = smem + NS * STAGE_BYTES;
  epi.template prefetch<HM, BN, 0>(aux, m0, n0, wave, lane);
  epi.template prefetch<HM, BN, 0>(aux + p.aux_half, m0 + HM, n0, wave, lane);
  __builtin_amdgcn_sched_barrier(0);

  constexpr int RD = NP * (FM + FN);                     // ds_read_b128 per K-tile
  constexpr int MQ = 3 * FM * FN;                        // MFMAs per K-tile
  constexpr int MPR = MQ / RD;
#define MSD_P_READ1(FA, FB, BUF, Q)                                                                   \
  {                                                                                                   \
    const int pl_ = (Q) / (FM + FN), r_ = (Q) % (FM + FN);                                            \
    const char* base_ = smem + (BUF) * STAGE_BYTES;                                                   \
    if (r_ < FM)                                                                                      \
      FA[pl_][r_ < FM ? r_ : 0] = *reinterpret_cast<const mfma_h16x8*>(                               \
          base_ + pl_ * A_BYTES + lds_pair_off(wm * WM + r_ * 16 + (lane & 15), lane >> 4));          \
    else                                                                                              \
      FB[pl_][r_ < FM ? 0 : r_ - FM] = *reinterpret_cast<const mfma_h16x8*>(                          \
          base_ + NP * A_BYTES + pl_ * B_BYTES + lds_pair_off(wn * WN + (r_ - FM) * 16 + (lane & 15), lane >> 4)); \
  }
#define MSD_P_MFMA1(FA, FB, E)                                                                        \
  {                                                                                                   \
    const int pr_ = (E) / (FM * FN), t_ = (E) % (FM * FN), i_ = t_ / FN, j_ = t_ % FN;                \
    const int pb_ = (pr_ == 1) ? NP - 1 : 0, pa_ = (pr_ == 2) ? NP - 1 : 0;                           \
    acc[i_][j_] = MSD_MFMA_16X16X32(FB[pb_][j_], FA[pa_][i_], acc[i_][j_], 0, 0, 0);                  \
  }
  // one K-tile: [DMA piece q of tile KT_ISSUE into BUF_I | read q of the tile in BUF_R -> FAn/FBn | MFMAs on FAc/FBc]
#define MSD_P_TILE(DO_ISSUE, KT_ISSUE, BUF_I, DO_READ, FAn, FBn, BUF_R, FAc, FBc)                     \
  {                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < RD; ++q_) {                                               \
      if (DO_ISSUE && q_ < PW) MSD_P_ISSUE1(KT_ISSUE, BUF_I, q_)                                      \
      if (DO_READ) MSD_P_READ1(FAn, FBn, BUF_R, q_)                                                   \
      _Pragma("unroll") for (int e_ = q_ * MPR; e_ < (q_ + 1 == RD ? MQ : (q_ + 1) * MPR); ++e_)      \
        MSD_P_MFMA1(FAc, FBc, e_)                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                 \
  }

  mfma_h16x8 fa0[NP][FM], fb0[NP][FN], fa1[NP][FM], fb1[NP][FN];
  if (nk >= NS) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PW) : "memory");   // tile 0 landed (the aux rows may still fly)
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < RD; ++q) MSD_P_READ1(fa0, fb0, 0, q)
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)

  // Tile kt's fragments are in registers at the top of an iteration.  The wait + barrier publish tile kt+1 and tell
  // every wave that slot kt % 2 has been read by all, so the DMA of tile kt+2 goes into it while tile kt is multiplied
  // (the other block of this CU multiplies meanwhile whenever this one waits).  Two iterations per loop trip so that
  // the register sets alternate without copies.
  int kt = 0, buf = 0;
#define MSD_P_STEP(DO_ISSUE, FAc, FBc, FAn, FBn)                                                      \
  {                                                                                                   \
    const int nb = buf ^ 1;                                                                           \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
    __builtin_amdgcn_s_barrier();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    MSD_P_TILE(DO_ISSUE, kt + NS, buf, 1, FAn, FBn, nb, FAc, FBc)                                     \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                               \
    buf = nb;                                                                                         \
    ++kt;                                                                                             \
  }
  // (the issue predicate is a wave-uniform run-time branch around each DMA piece: one loop body instead of a steady
  // state and a drain variant of it, which cost registers this kernel does not have)
  while (kt + 2 < nk) {
    MSD_P_STEP(kt + NS < nk, fa0, fb0, fa1, fb1)
    MSD_P_STEP(kt + NS < nk, fa1, fb1, fa0, fb0)
  }
  // kt is even here and the current fragments are in set 0; one or two tiles remain
  if (kt + 1 < nk) {
    MSD_P_STEP(0, fa0, fb0, fa1, fb1)
    MSD_P_TILE(0, 0, 0, 0, fa0, fb0, 0, fa1, fb1)
  } else {
    MSD_P_TILE(0, 0, 0, 0, fa1, fb1, 0, fa0, fb0)
  }
#undef MSD_P_STEP
#undef MSD_P_TILE
#undef MSD_P_MFMA1
#undef MSD_P_READ1
#undef MSD_P_ISSUE1

  // ---- epilogue, one 64-row half at a time: the waves that own the half fill the slab, all four run the epilogue ---
  float* slab = reinterpret_cast<float*>(smem);
  const int lm = lane & 15, ln = (lane >> 4) * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();   // h = 0: all fragment reads of the ring are done; h = 1: the first half's slab has been read
    if (wm == h) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *reinterpret_cast<float4*>(slab + (size_t)(i * 16 + lm) * LDS_LD + wn * WN + j * 16 + ln) =
              make_float4(acc[i][j][0] * kWScaleInv, acc[i][j][1] * kWScaleInv, acc[i][j][2] * kWScaleInv,
                          acc[i][j][3] * kWScaleInv);
    }
    const char* auxh = aux + h * p.aux_half;
    epi.template stats<HM, LDS_LD>(slab, m0 + h * HM, tid, auxh);
    __syncthreads();
    epi.template run<HM, BN, LDS_LD>(slab, m0 + h * HM, n0, tid, auxh, /*stats_done=*/true, SatFlag{p.sat, p.sat_tag});
  }
}

// LDS of one block: the ring + the aux rows of both halves, sized for the partials this launch has
template <int NP, int BM, int BN, class Epi>
inline int gemm_h16_pair_aux_half(const Epi& epi) {
  const int tiles = epi_rowscale_tiles(epi, 0);
  return tiles < 0 ? Epi::template aux_bytes<BM / 2, BN>() : (tiles == 0 ? 0 : rowscale_ssq_bytes<BM / 2>(tiles) + 1024);
}
template <int NP, int BM, int BN>
constexpr int gemm_h16_pair_ring() { return 2 * NP * (BM + BN) * 64; }

template <int NP, int BM, int BN, class Epi>
inline hipError_t gemm_h16_pair_prepare() {
  // up to 80 KB so that two blocks share the CU's 160 KB; a larger aux region (D > 768) still runs, one block per CU
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_pair_kernel<NP, BM, BN, Epi>),
                             hipFuncAttributeMaxDynamicSharedMemorySize,
                             gemm_h16_pair_ring<NP, BM, BN>() + 2 * Epi::template aux_bytes<BM / 2, BN>());
}

template <int NP, int BM, int BN, class Epi>
inline hipError_t launch_gemm_h16_pair(GemmParams p, const Epi& epi, hipStream_t stream) {
  static const hipError_t attr = gemm_h16_pair_prepare<NP, BM, BN, Epi>();
  if (attr != hipSuccess) return attr;
  p.aux_half = gemm_h16_pair_aux_half<NP, BM, BN>(epi);
  const int smem = gemm_h16_pair_ring<NP, BM, BN>() + 2 * p.aux_half;
  const int rx = p.xcd_rows, cx = 8 / rx;
  const int grid = 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / BM + rx - 1) / rx);
  hipLaunchKernelGGL((gemm_h16_pair_kernel<NP, BM, BN, Epi>), dim3(grid), dim3(256), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
