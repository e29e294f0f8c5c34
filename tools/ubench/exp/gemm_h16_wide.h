// Batched-path GEMM, gated-MLP input projection (M = passes x songs x T >= 2048 rows, N a multiple of 128): 256 x 128
// tiles on EIGHT waves.
//
// Why.  Counter passes over the batched step (tools/diag/pmc_diag.sh, profiles/r03s_diag_b8_*.csv) put the 128 x 128
// tile of gemm_h16.h at 42 % MFMA-pipe occupancy with the waves 21 % parked (SQ_WAIT_ANY), 40 % in issue stalls
// (SQ_WAIT_INST_ANY) and an L2 read latency of ~415 clocks at 79 % hit rate: not latency, not LDS (8 % bank-conflict
// cycles) -- the CU's vector-memory path accepts an LDS-DMA instruction (1 KiB) about every 32 clocks (31 - 35 B/clk,
// the same ceiling the one-song tiles sit on), every one of them is issued from the in-order stream of a
// wave that also has to issue the MFMAs, and a 128 x 128 two-plane tile needs 64 KB of them per 1 536 MFMA clocks:
// 2 000 clocks of ingest per K-tile.  The only lever is fewer operand bytes per MFMA: a 256 x 128 tile moves
// (256 + 128) rows where two 128 x 128 tiles move 512 -- 25 % less -- and puts two waves on every SIMD, so one wave's
// DMA issue slots and barrier waits sit under the other's MFMAs.  The K-tile is 32 deep (64-byte rows, the layout
// and source-side swizzle of gemm_h16_pair.h): 48 KB per stage, a THREE-deep ring in 144 KB.
//
// The aux rows of the epilogue (row-statistics partials, bias row) do not fit beside that ring; they are fetched into
// the freed ring after the main loop (one exposed L2 round trip per tile, two tiles per CU and launch).  The epilogue
// is the unchanged 256-thread functor of gemm_h16.h, run by the two halves of the block side by side: waves 0-3 take
// rows 0..127 of the tile, waves 4-7 rows 128..255, each half in two 64-row passes through its own 34 KB slab.
#pragma once
#include "gemm_h16_pair.h"

namespace msd {

template <int NP, int BN, class Epi>
__global__ void __launch_bounds__(512, 2) gemm_h16_wide_kernel(GemmParams p, Epi epi) {
  static_assert(NP == 2, "the batched tiles exist for the two-plane modes");
  constexpr int BM = 256, NS = 3, NW = 8;
  constexpr int WM = 64, WN = BN / 2;               // 4 x 2 waves: wave (wm, wn) owns rows wm * 64.., columns wn * WN..
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  constexpr int STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  constexpr int PA = NP * BM / 16 / NW, PB = NP * BN / 16 / NW;   // DMA pieces per wave and K-tile
  static_assert(PB * NW * 16 == NP * BN, "a whole number of B pieces per wave");
  constexpr int PW = PA + PB;
  constexpr int HM = 64;                            // rows per epilogue pass
  constexpr int LDS_LD = BN + kSlabPad;
  constexpr int SLAB_BYTES = ((HM * LDS_LD + HM) * 4 + 1023) / 1024 * 1024;
  constexpr int AUX_OFF = 2 * SLAB_BYTES;           // aux rows of the four passes, behind the two slabs
  static_assert(AUX_OFF + 4 * Epi::template aux_bytes<HM, BN>() <= NS * STAGE_BYTES, "slabs + aux rows must fit the ring");
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
  // source chunk (lane & 3) ^ G[(r >> 2) & 3].  Wave-uniform 64-bit bases (scalar registers) + one 32-bit lane offset
  // per operand (gemm_h16_pair.h).
  const int r16 = lane >> 2, csrc = (lane & 3) ^ ((0 - (r16 >> 2)) & 3);
  const unsigned offA = (unsigned)(r16 * p.lda + csrc * 8) * 2u, offB = (unsigned)(r16 * p.ldb + csrc * 8) * 2u;
  const char* gbase[PW];
  int ldst[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    if (i < PA) {
      const int a = wave + NW * i, pl = a / (BM / 16), blk = a % (BM / 16);
      gbase[i] = reinterpret_cast<const char*>(p.A[pl] + (size_t)(m0 + blk * 16) * p.lda);
      ldst[i] = pl * A_BYTES + blk * 1024;
    } else {
      const int b = wave + NW * (i - PA), pl = b / (BN / 16), blk = b % (BN / 16);
      gbase[i] = reinterpret_cast<const char*>(p.B[pl] + (size_t)(n0 + blk * 16) * p.ldb);
      ldst[i] = NP * A_BYTES + pl * B_BYTES + blk * 1024;
    }
  }
#define MSD_W_ISSUE1(KT, BUF, I)                                                                          \
  __builtin_amdgcn_global_load_lds(                                                                       \
      (gptr_t)(gbase[I] + (size_t)((KT) * (kPairBK * 2)) + (size_t)((I) < PA ? offA : offB)),             \
      (lptr_t)(smem + (BUF) * STAGE_BYTES + ldst[I]), 16, 0, 0);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / kPairBK;   // >= NS (launcher)
  // ---- prologue: the three ring slots in flight ---------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int i = 0; i < PW; ++i) MSD_W_ISSUE1(s, s, i)
  }
  __builtin_amdgcn_sched_barrier(0);

  constexpr int RD = NP * (FM + FN);                     // ds_read_b128 per K-tile
  constexpr int MQ = 3 * FM * FN;                        // MFMAs per K-tile
  constexpr int MPR = MQ / RD;
#define MSD_W_READ1(FA, FB, BUF, Q)                                                                   \
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
#define MSD_W_MFMA1(FA, FB, E)                                                                        \
  {                                                                                                   \
    const int pr_ = (E) / (FM * FN), t_ = (E) % (FM * FN), i_ = t_ / FN, j_ = t_ % FN;                \
    const int pb_ = (pr_ == 1) ? NP - 1 : 0, pa_ = (pr_ == 2) ? NP - 1 : 0;                           \
    acc[i_][j_] = MSD_MFMA_16X16X32(FB[pb_][j_], FA[pa_][i_], acc[i_][j_], 0, 0, 0);                  \
  }
  // one K-tile: [DMA piece q of tile KT_ISSUE into BUF_I | read q of the tile in BUF_R -> FAn/FBn | MFMAs on FAc/FBc]
#define MSD_W_TILE(DO_ISSUE, KT_ISSUE, BUF_I, DO_READ, FAn, FBn, BUF_R, FAc, FBc)                     \
  {                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < RD; ++q_) {                                               \
      if (DO_ISSUE && q_ < PW) MSD_W_ISSUE1(KT_ISSUE, BUF_I, q_)                                      \
      if (DO_READ) MSD_W_READ1(FAn, FBn, BUF_R, q_)                                                   \
      _Pragma("unroll") for (int e_ = q_ * MPR; e_ < (q_ + 1 == RD ? MQ : (q_ + 1) * MPR); ++e_)      \
        MSD_W_MFMA1(FAc, FBc, e_)                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                 \
  }

  mfma_h16x8 fa0[NP][FM], fb0[NP][FN], fa1[NP][FM], fb1[NP][FN];
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PW) : "memory");   // tile 0 landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < RD; ++q) MSD_W_READ1(fa0, fb0, 0, q)
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)

  // Tile kt's fragments are in registers at the top of a step.  The counted wait + barrier publish tile kt+1 (tile
  // kt+2 may still fly) and tell every wave that slot kt % 3 has been read by all, so the DMA of tile kt+3 goes into
  // it while tile kt is multiplied.  Two steps per loop trip so that the register sets alternate without copies; the
  // issue predicate and the choice between the counted and the complete wait are wave-uniform run-time branches.
  int kt = 0, buf = 0;
#define MSD_W_STEP(FAc, FBc, FAn, FBn)                                                                \
  {                                                                                                   \
    const int nb = buf + 1 == NS ? 0 : buf + 1;                                                       \
    if (kt + 2 < nk) {                                                                                \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");                                       \
    } else {                                                                                          \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
    }                                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    MSD_W_TILE(kt + NS < nk, kt + NS, buf, 1, FAn, FBn, nb, FAc, FBc)                                 \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                               \
    buf = nb;                                                                                         \
    ++kt;                                                                                             \
  }
  while (kt + 2 < nk) {
    MSD_W_STEP(fa0, fb0, fa1, fb1)
    MSD_W_STEP(fa1, fb1, fa0, fb0)
  }
  // kt is even here and the current fragments are in set 0; one or two tiles remain
  if (kt + 1 < nk) {
    MSD_W_STEP(fa0, fb0, fa1, fb1)
    MSD_W_TILE(false, 0, 0, 0, fa0, fb0, 0, fa1, fb1)
  } else {
    MSD_W_TILE(false, 0, 0, 0, fa1, fb1, 0, fa0, fb0)
  }
#undef MSD_W_STEP
#undef MSD_W_TILE
#undef MSD_W_MFMA1
#undef MSD_W_READ1
#undef MSD_W_ISSUE1

  // ---- epilogue: the two 256-thread halves side by side, each 128 rows in two 64-row passes -------------------------
  const int grp = wave >> 2, wl = wave & 3, tidl = tid & 255;
  float* slab = reinterpret_cast<float*>(smem + grp * SLAB_BYTES);
  char* const aux = smem + AUX_OFF;
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  __syncthreads();   // all fragment reads of the ring are done: it becomes slabs + aux rows
#pragma unroll
  for (int h = 0; h < 2; ++h)
    epi.template prefetch<HM, BN, 0>(aux + (2 * grp + h) * p.aux_half, m0 + (2 * grp + h) * HM, n0, wl, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();   // h = 0: aux rows visible; h = 1: the first pass's slab has been read
    const int qd = 2 * grp + h;   // 64-row quarter of the tile = wm of the waves that own it
    if (wm == qd) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *reinterpret_cast<float4*>(slab + (size_t)(i * 16 + lm) * LDS_LD + wn * WN + j * 16 + ln) =
              make_float4(acc[i][j][0] * kWScaleInv, acc[i][j][1] * kWScaleInv, acc[i][j][2] * kWScaleInv,
                          acc[i][j][3] * kWScaleInv);
    }
    const char* auxq = aux + qd * p.aux_half;
    epi.template stats<HM, LDS_LD>(slab, m0 + qd * HM, tidl, auxq);
    __syncthreads();
    epi.template run<HM, BN, LDS_LD>(slab, m0 + qd * HM, n0, tidl, auxq, /*stats_done=*/true, SatFlag{p.sat, p.sat_tag});
  }
}

template <int NP, int BN>
constexpr int gemm_h16_wide_smem() { return 3 * NP * (256 + BN) * 64; }

template <int NP, int BN, class Epi>
inline hipError_t gemm_h16_wide_prepare() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_wide_kernel<NP, BN, Epi>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, gemm_h16_wide_smem<NP, BN>());
}

// does this problem fit the wide tiles?  (whole tiles, at least a ring of K-tiles)
template <int BN>
inline bool gemm_h16_wide_fits(int M, int N, int K) { return M % 256 == 0 && N % BN == 0 && K % kPairBK == 0 && K / kPairBK >= 3; }

template <int NP, int BN, class Epi>
inline hipError_t launch_gemm_h16_wide(GemmParams p, const Epi& epi, hipStream_t stream) {
  static const hipError_t attr = gemm_h16_wide_prepare<NP, BN, Epi>();
  if (attr != hipSuccess) return attr;
  p.aux_half = gemm_h16_pair_aux_half<NP, 128, BN>(epi);   // per 64-row pass
  const int rx = p.xcd_rows, cx = 8 / rx;
  const int grid = 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / 256 + rx - 1) / rx);
  constexpr int smem = gemm_h16_wide_smem<NP, BN>();
  hipLaunchKernelGGL((gemm_h16_wide_kernel<NP, BN, Epi>), dim3(grid), dim3(512), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
