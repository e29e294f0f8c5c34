// EXPERIMENT RECORD, not part of the product build (round 3).  Parity-green when it was wired
// into csrc/msd_api.hip (tests/test_gpu_model.py::test_batched_songs_use_big_tiles_and_match_oracle, profiles/
// r03m_batched_test.log) and 10 % SLOWER than the 2-deep K = 64 tiles at 8 and 16 songs per GPU (profiles/r03m_k32_ab.log).
// To re-run: include it next to csrc/gemm_h16.h, route the 128-row tiles of msd_api.hip's gemm<> to launch_gemm_h16_k32.
//
// A batched-path GEMM main loop (M = passes x songs x T >= 2048 rows): 128-row tiles on K-tiles of 32.
//
// Why another loop.  With several songs per handle every CU owns several 128 x 128 (128 x 96) tiles, and the
// ablation of the 128-row tiles of gemm_h16.h (profiles/r02_gemm_ablation.log) showed the loop bound by
// the L2 -> LDS ingest at ~23 B/clk per CU -- two thirds of what the same CU sustains on the 64-row tiles: a two-plane
// K-tile of 64 is 64 KB there, so the 160 KB of LDS hold a ring of TWO slots, and while one tile is multiplied at most
// ONE is in flight.  Latency, not bandwidth.  Here a K-tile is 32 deep (rows of 64 bytes): a slot is 32 KB (28 KB),
// the ring is FOUR deep, two to three tiles are in flight behind the one being multiplied, and the loop structure is
// the plain one that goes with it -- the fragments of tile kt+1 are read into a second register set under the MFMAs
// of tile kt, one barrier per K-tile.
//
// Layout.  A stage holds, per plane, [BM rows | BN rows] of 64 bytes.  LDS-DMA (global_load_lds, 1 KiB = 16 rows per
// wave-instruction) writes rows linearly, so the bank swizzle sits on the SOURCE side as in gemm_h16.h: 16-byte chunk
// c of row r is stored at position c ^ G[(r >> 2) & 3], G = {0, 3, 2, 1}.  A fragment read (ds_read_b128: lane = (row
// l & 15, chunk l >> 4)) then touches, in each of the instruction's four 16-lane groups, sixteen different
// (row mod 4, position) pairs = all 64 banks once (MI355X_MICROARCH.md, LDS table).  The NP x (BM + BN) / 16 DMA
// instructions of a K-tile are dealt round-robin over the four waves (32 or 28: a whole number each).
//
// Epilogues, tile map and launch conventions are those of gemm_h16.h (the weight prefetch: see the end of gemm_tile_k32).
#pragma once
#include "../../music-spectrogram-diffusion_amd/csrc/gemm_h16.h"

namespace msd {

constexpr int kK32 = 32;   // K elements per tile: one MFMA K-step, 64-byte rows

// byte offset of chunk `c` (0..3) of `row` in a [rows][32] 16-bit LDS tile
__device__ __forceinline__ int lds_k32_off(int row, int c) { return row * 64 + ((c ^ ((0 - (row >> 2)) & 3)) << 4); }

template <int NP, int BM, int BN, int NS, class Epi, int PF>
__device__ __forceinline__ void gemm_tile_k32(const GemmParams& p, const Epi& epi, int bm, int bn, char* smem) {
  static_assert(NP == 2, "the batched tiles exist for the two-plane modes only");
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  constexpr int STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  constexpr int ROWBLK = (BM + BN) / 16;            // 16-row DMA pieces per plane
  constexpr int T = NP * ROWBLK;                    // DMA instructions per K-tile, all waves
  static_assert(T % 4 == 0, "a whole number of DMA instructions per wave");
  constexpr int PW = T / 4;                         // per wave
  constexpr int LDS_LD = BN + kSlabPad;
  static_assert((BM * LDS_LD + BM) * 4 <= NS * STAGE_BYTES, "epilogue slab must fit the operand LDS");
  static_assert((NS - 2) * PW <= 63, "vmcnt immediate");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = bm * BM, n0 = bn * BN;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // DMA piece j = wave + 4 i of a K-tile: plane j / ROWBLK, 16-row block j % ROWBLK of [A rows | B rows]; this lane
  // fetches row r = lane >> 2 of the block, source chunk (lane & 3) ^ G[(r >> 2) & 3]
  const int r16 = lane >> 2, csrc = (lane & 3) ^ ((0 - (r16 >> 2)) & 3);
  const h16_t* gsrc[PW];
  int ldst[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int j = wave + 4 * i;
    const int pl = j / ROWBLK, blk = j % ROWBLK;
    if (blk < BM / 16) {
      gsrc[i] = p.A[pl] + (size_t)(m0 + blk * 16 + r16) * p.lda + csrc * 8;
      ldst[i] = pl * A_BYTES + blk * 1024;
    } else {
      gsrc[i] = p.B[pl] + (size_t)(n0 + (blk - BM / 16) * 16 + r16) * p.ldb + csrc * 8;
      ldst[i] = NP * A_BYTES + pl * B_BYTES + (blk - BM / 16) * 1024;
    }
  }
#define MSD_K_ISSUE1(KT, BUF, I) \
  __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[I] + (KT) * kK32), (lptr_t)(smem + (BUF) * STAGE_BYTES + ldst[I]), 16, 0, 0);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / kK32;
  // ---- prologue: NS tiles in flight ------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < nk) {
#pragma unroll
      for (int i = 0; i < PW; ++i) MSD_K_ISSUE1(s, s, i)
    }
  char* const aux = smem + NS * STAGE_BYTES;
  epi.template prefetch<BM, BN, 0>(aux, m0, n0, wave, lane);
  __builtin_amdgcn_sched_barrier(0);

  // fragment read number Q of the tile in slot BUF into FA / FB
  constexpr int RD = NP * (FM + FN);                     // ds_read_b128 per K-tile
  constexpr int MQ = 3 * FM * FN;                        // MFMAs per K-tile
  constexpr int MPR = MQ / RD;
#define MSD_K_READ1(FA, FB, BUF, Q)                                                                   \
  {                                                                                                   \
    const int pl_ = (Q) / (FM + FN), r_ = (Q) % (FM + FN);                                            \
    const char* base_ = smem + (BUF) * STAGE_BYTES;                                                   \
    if (r_ < FM)                                                                                      \
      FA[pl_][r_ < FM ? r_ : 0] = *reinterpret_cast<const mfma_h16x8*>(                               \
          base_ + pl_ * A_BYTES + lds_k32_off(wm * WM + r_ * 16 + (lane & 15), lane >> 4));           \
    else                                                                                              \
      FB[pl_][r_ < FM ? 0 : r_ - FM] = *reinterpret_cast<const mfma_h16x8*>(                          \
          base_ + NP * A_BYTES + pl_ * B_BYTES + lds_k32_off(wn * WN + (r_ - FM) * 16 + (lane & 15), lane >> 4)); \
  }
#define MSD_K_MFMA1(FA, FB, E)                                                                        \
  {                                                                                                   \
    const int pr_ = (E) / (FM * FN), t_ = (E) % (FM * FN), i_ = t_ / FN, j_ = t_ % FN;                \
    const int pb_ = (pr_ == 1) ? NP - 1 : 0, pa_ = (pr_ == 2) ? NP - 1 : 0;                           \
    acc[i_][j_] = MSD_MFMA_16X16X32(FB[pb_][j_], FA[pa_][i_], acc[i_][j_], 0, 0, 0);                  \
  }
  // one K-tile: [DMA piece q of tile KT_ISSUE into BUF_I | read q of the tile in BUF_R -> FAn/FBn | MFMAs on FAc/FBc]
#define MSD_K_TILE(DO_ISSUE, KT_ISSUE, BUF_I, DO_READ, FAn, FBn, BUF_R, FAc, FBc)                     \
  {                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < RD; ++q_) {                                               \
      if (DO_ISSUE && q_ < PW) MSD_K_ISSUE1(KT_ISSUE, BUF_I, q_)                                      \
      if (DO_READ) MSD_K_READ1(FAn, FBn, BUF_R, q_)                                                   \
      _Pragma("unroll") for (int e_ = q_ * MPR; e_ < (q_ + 1 == RD ? MQ : (q_ + 1) * MPR); ++e_)      \
        MSD_K_MFMA1(FAc, FBc, e_)                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                 \
  }

  mfma_h16x8 fa0[NP][FM], fb0[NP][FN], fa1[NP][FM], fb1[NP][FN];
  if (nk >= NS) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PW) : "memory");   // tile 0 landed
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < RD; ++q) MSD_K_READ1(fa0, fb0, 0, q)
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)

  // Tile kt's fragments are in registers at the top of an iteration.  The wait + barrier publish tile kt+1 and
  // tell every wave that slot kt % NS has been read by all (those reads completed before the previous
  // iteration ended), so the DMA of tile kt+NS can go into it while tile kt is multiplied.  Two iterations per
  // loop trip so that the register sets alternate without copies.
  int kt = 0, buf = 0;
#define MSD_K_STEP(DO_ISSUE, VMWAIT, FAc, FBc, FAn, FBn)                                              \
  {                                                                                                   \
    int nb = buf + 1;                                                                                 \
    if (nb == NS) nb = 0;                                                                             \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMWAIT) : "memory");                                     \
    __builtin_amdgcn_s_barrier();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    MSD_K_TILE(DO_ISSUE, kt + NS, buf, 1, FAn, FBn, nb, FAc, FBc)                                     \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                               \
    buf = nb;                                                                                         \
    ++kt;                                                                                             \
  }
  // steady state: tiles kt+1 .. kt+NS-1 are in flight at the wait, tile kt+1 must have landed
  while (kt + NS + 1 < nk) {
    MSD_K_STEP(1, (NS - 2) * PW, fa0, fb0, fa1, fb1)
    MSD_K_STEP(1, (NS - 2) * PW, fa1, fb1, fa0, fb0)
  }
  // drain: nothing (or one last tile) left to issue; waits are complete ones
  while (kt + 2 < nk) {
    if (kt + NS < nk) {
      MSD_K_STEP(1, 0, fa0, fb0, fa1, fb1)
    } else {
      MSD_K_STEP(0, 0, fa0, fb0, fa1, fb1)
    }
    if (kt + NS < nk) {
      MSD_K_STEP(1, 0, fa1, fb1, fa0, fb0)
    } else {
      MSD_K_STEP(0, 0, fa1, fb1, fa0, fb0)
    }
  }
  // kt is even here and the current fragments are in set 0; one or two tiles remain
  if (kt + 1 < nk) {
    MSD_K_STEP(0, 0, fa0, fb0, fa1, fb1)
    MSD_K_TILE(0, 0, 0, 0, fa0, fb0, 0, fa1, fb1)
  } else {
    MSD_K_TILE(0, 0, 0, 0, fa1, fb1, 0, fa0, fb0)
  }
#undef MSD_K_STEP
#undef MSD_K_TILE
#undef MSD_K_MFMA1
#undef MSD_K_READ1
#undef MSD_K_ISSUE1

  __syncthreads();  // all fragment reads done before the slab overwrites the ring
  // The later launch's weights are touched HERE, by the compute waves behind their main loop (the in-epilogue form
  // of gemm_h16.h), whatever the build's prefetch mechanism: this loop needs ~340 registers, and a fifth (prefetch)
  // wave in the block would halve the budget to 256 and spill (172 VGPRs to scratch when it was tried).  On 50 - 90 us
  // kernels the wait for the touches at the end is noise.
  PrefetchRegsT<PF> pf_keep;
  prefetch_weights<PF>(p.pf, blockIdx.x, gridDim.x, p.B[0], pf_keep);
  float* slab = reinterpret_cast<float*>(smem);
  const int lm = lane & 15, ln = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
      *reinterpret_cast<float4*>(slab + (size_t)(wm * WM + i * 16 + lm) * LDS_LD + wn * WN + j * 16 + ln) =
          make_float4(acc[i][j][0] * kWScaleInv, acc[i][j][1] * kWScaleInv, acc[i][j][2] * kWScaleInv,
                      acc[i][j][3] * kWScaleInv);
  epi.template stats<BM, LDS_LD>(slab, m0, tid, aux);
  __syncthreads();
  epi.template run<BM, BN, LDS_LD>(slab, m0, n0, tid, aux, /*stats_done=*/true, SatFlag{p.sat, p.sat_tag});
  prefetch_done(pf_keep);
}

template <int NP, int BM, int BN, int NS, class Epi, int PF = kPfNone>
__global__ void __launch_bounds__(256) gemm_h16_k32_kernel(GemmParams p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  gemm_tile_k32<NP, BM, BN, NS, Epi, PF>(p, epi, bm, bn, smem);
}

template <int NP, int BM, int BN, int NS, class Epi>
constexpr int gemm_h16_k32_smem() { return NS * NP * (BM + BN) * 64 + Epi::template aux_bytes<BM, BN>(); }

template <int NP, int BM, int BN, int NS, class Epi>
inline hipError_t gemm_h16_k32_prepare() {
  constexpr int smem = gemm_h16_k32_smem<NP, BM, BN, NS, Epi>();
  hipError_t e = hipSuccess, r;
#define MSD_K32_ATTR(PF_)                                                                                           \
  if ((r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_k32_kernel<NP, BM, BN, NS, Epi, PF_>),        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem)) != hipSuccess) e = r;
  MSD_K32_ATTR(0) MSD_K32_ATTR(1) MSD_K32_ATTR(2)
#undef MSD_K32_ATTR
  return e;
}

template <int NP, int BM, int BN, int NS, class Epi>
inline hipError_t launch_gemm_h16_k32(const GemmParams& p, const Epi& epi, hipStream_t stream) {
  constexpr int smem = gemm_h16_k32_smem<NP, BM, BN, NS, Epi>();
  static const hipError_t attr = gemm_h16_k32_prepare<NP, BM, BN, NS, Epi>();
  if (attr != hipSuccess) return attr;
  const int rx = p.xcd_rows, cx = 8 / rx;
  const int grid = 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / BM + rx - 1) / rx);
  const int npf = prefetch_kind(p.pf);
  if (npf >= 2) hipLaunchKernelGGL((gemm_h16_k32_kernel<NP, BM, BN, NS, Epi, 2>), dim3(grid), dim3(256), smem, stream, p, epi);
  else if (npf == 1) hipLaunchKernelGGL((gemm_h16_k32_kernel<NP, BM, BN, NS, Epi, 1>), dim3(grid), dim3(256), smem, stream, p, epi);
  else hipLaunchKernelGGL((gemm_h16_k32_kernel<NP, BM, BN, NS, Epi, 0>), dim3(grid), dim3(256), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
