// Batched-path GEMM, gated-MLP input projection (M >= 2048 rows, N a multiple of 128): 256 x 128 tiles with LOADER
// waves -- four waves multiply, four waves do nothing but issue the LDS-DMA.
//
// Why.  The 256 x 128 eight-wave tile (gemm_h16_wide.h) moves 25 % fewer operand bytes per MFMA than two 128 x 128
// tiles and still needed ~3 400 clocks per K-tile of 32 where the MFMA pipe needs 1 536 and the CU's vector-memory
// path ~1 550 (48 KB at 31 B/clk): the two do not overlap, they ADD UP (profiles/r03t_wide_ab.log; the counter passes
// profiles/r03s_diag_b8_*.csv show the same on the 128 x 128 tile: 40 % of the wave cycles are issue stalls).  An
// LDS-DMA instruction holds the wave that issues it until the address unit takes it -- 50 - 250 clocks once that unit
// is the bottleneck -- and in every kernel so far the waves that issue the DMA are the waves that issue the MFMAs, in
// order.  Here they are not: waves 4 - 7 (one per SIMD) issue all 48 DMA pieces of a K-tile and wait for them; waves
// 0 - 3 (one per SIMD, 128 x 64 of the tile each) read fragments and multiply, and never touch the vector-memory
// path inside the loop.  A SIMD whose loader wave is held by the address unit keeps issuing its consumer's MFMAs.
//
// Ring: three stages of 48 KB (K-tile of 32, 64-byte rows, the layout and source-side swizzle of gemm_h16_pair.h).
// One s_barrier per K-tile: a loader arrives when tile kt+1 has landed, a consumer when its fragment reads of tile
// kt are complete; behind it the loaders refill slot kt % 3 with tile kt+3 and the consumers finish tile kt.
// Registers: one consumer wave holds 128 accumulator registers + one fragment set (96) -- single-buffered, the hi
// planes are read first and the first of the three products starts on them while the lo planes arrive.
// Epilogue: the unchanged 256-thread functors of gemm_h16.h on the four consumer waves, four 64-row passes through a
// 34 KB slab; the aux rows (row statistics, bias) are fetched into the freed ring behind the loop.  The loader waves
// have ended by then; loader 0 first issues the weight touches for the next launch (gemm_h16.h prefetch_wave).
#pragma once
#include "gemm_h16_pair.h"

namespace msd {

template <int NP, int BN, class Epi, int PF = kPfNone>
__global__ void __launch_bounds__(512, 2) gemm_h16_ls_kernel(GemmParams p, Epi epi) {
  static_assert(NP == 2, "the batched tiles exist for the two-plane modes");
  constexpr int BM = 256, NS = 3, NL = 4;           // NL loader waves
  constexpr int WM = 128, WN = BN / 2;              // 2 x 2 consumer waves: wave (wm, wn) owns rows wm * 128.., columns wn * WN..
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  constexpr int STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  constexpr int PA = NP * BM / 16 / NL, PB = NP * BN / 16 / NL;   // DMA pieces per loader wave and K-tile
  static_assert(PB * NL * 16 == NP * BN, "a whole number of B pieces per loader");
  constexpr int PW = PA + PB;
  static_assert((NS - 1) * PW <= 63, "vmcnt immediate");
  constexpr int HM = 64;                            // rows per epilogue pass
  constexpr int LDS_LD = BN + kSlabPad;
  constexpr int SLAB_BYTES = ((HM * LDS_LD + HM) * 4 + 1023) / 1024 * 1024;
  constexpr int AUX_OFF = SLAB_BYTES;               // aux rows of the four passes, behind the slab
  static_assert(AUX_OFF + 4 * Epi::template aux_bytes<HM, BN>() <= NS * STAGE_BYTES, "slab + aux rows must fit the ring");
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
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk = p.K / kPairBK;   // >= NS (launcher)

  MSD_TS_BEGIN(7, blockIdx.x)
  if (wave >= 4) {
    // ================================================ loader waves ===================================================
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int lw = wave - 4;
    // a DMA piece = one 16-row block of one plane of one operand; this lane fetches row r = lane >> 2 of the block,
    // source chunk (lane & 3) ^ G[(r >> 2) & 3]; wave-uniform 64-bit bases + one 32-bit lane offset per operand
    const int r16 = lane >> 2, csrc = (lane & 3) ^ ((0 - (r16 >> 2)) & 3);
    const unsigned offA = (unsigned)(r16 * p.lda + csrc * 8) * 2u, offB = (unsigned)(r16 * p.ldb + csrc * 8) * 2u;
    const char* gbase[PW];
    int ldst[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      if (i < PA) {
        const int a = lw + NL * i, pl = a / (BM / 16), blk = a % (BM / 16);
        gbase[i] = reinterpret_cast<const char*>(p.A[pl] + (size_t)(m0 + blk * 16) * p.lda);
        ldst[i] = pl * A_BYTES + blk * 1024;
      } else {
        const int b = lw + NL * (i - PA), pl = b / (BN / 16), blk = b % (BN / 16);
        gbase[i] = reinterpret_cast<const char*>(p.B[pl] + (size_t)(n0 + blk * 16) * p.ldb);
        ldst[i] = NP * A_BYTES + pl * B_BYTES + blk * 1024;
      }
    }
#define MSD_L_ISSUE(KT, BUF)                                                                                   \
  {                                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < PW; ++i_)                                                          \
      __builtin_amdgcn_global_load_lds(                                                                        \
          (gptr_t)(gbase[i_] + (size_t)((KT) * (kPairBK * 2)) + (size_t)(i_ < PA ? offA : offB)),              \
          (lptr_t)(smem + (BUF) * STAGE_BYTES + ldst[i_]), 16, 0, 0);                                          \
  }
#pragma unroll
    for (int s = 0; s < NS; ++s) MSD_L_ISSUE(s, s)
#if MSD_TIMESTAMPS
    if (threadIdx.x == 256 && blockIdx.x < kTsBlocks) g_msd_ts[7][blockIdx.x][7] = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PW) : "memory");   // tile 0 landed
    __builtin_amdgcn_s_barrier();                                          // B(-1): tile 0 visible
    int buf = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
      // tile kt+1 must have landed; tile kt+2 (the newest PW instructions, if it exists) may still fly
      if (kt + 2 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // B(kt): the consumers have read tile kt (slot `buf` is free); tile kt+1 visible
      if (kt + NS < nk) MSD_L_ISSUE(kt + NS, buf)
      buf = buf + 1 == NS ? 0 : buf + 1;
    }
#undef MSD_L_ISSUE
    // warm the next launch's weights (one wave per block, as the prefetch wave of gemm_h16_dma_kernel does)
    if (lw == 0) prefetch_wave<PF>(p.pf, blockIdx.x, gridDim.x, p.B[0]);
    return;
  }

  // ================================================== consumer waves ===================================================
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  mfma_h16x8 fa[NP][FM], fb[NP][FN];
  // fragment reads of plane PL of the tile in slot BUF
#define MSD_C_READ(PL, BUF)                                                                                     \
  {                                                                                                             \
    const char* base_ = smem + (BUF) * STAGE_BYTES;                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < FM; ++i_)                                                           \
      fa[PL][i_] = *reinterpret_cast<const mfma_h16x8*>(                                                        \
          base_ + (PL) * A_BYTES + lds_pair_off(wm * WM + i_ * 16 + (lane & 15), lane >> 4));                   \
    _Pragma("unroll") for (int j_ = 0; j_ < FN; ++j_)                                                           \
      fb[PL][j_] = *reinterpret_cast<const mfma_h16x8*>(                                                        \
          base_ + NP * A_BYTES + (PL) * B_BYTES + lds_pair_off(wn * WN + j_ * 16 + (lane & 15), lane >> 4));    \
  }
  // one of the three products of a tile: planes PA_ of A and PB_ of B
#define MSD_C_PRODUCT(PA_, PB_)                                                                                 \
  {                                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < FM; ++i_)                                                           \
      _Pragma("unroll") for (int j_ = 0; j_ < FN; ++j_)                                                         \
        acc[i_][j_] = MSD_MFMA_16X16X32(fb[PB_][j_], fa[PA_][i_], acc[i_][j_], 0, 0, 0);                        \
  }
  __builtin_amdgcn_s_barrier();   // B(-1): tile 0 visible
  MSD_TS_AT(7, blockIdx.x, 1)
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // hi planes first; when they are here (the lgkm counter is four bits wide: with all 24 reads outstanding the
    // compiler can only wait for all of them) the lo-plane reads go out and hi x hi runs under them
    MSD_C_READ(0, buf)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    MSD_C_READ(1, buf)
    __builtin_amdgcn_sched_barrier(0);
    MSD_C_PRODUCT(0, 0)
    __builtin_amdgcn_sched_barrier(0);    // (pins the wait behind the 32 MFMAs: the scheduler hoists it otherwise)
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): all fragment reads of this tile are complete
    if (kt + 1 < nk) __builtin_amdgcn_s_barrier();   // B(kt): slot `buf` may be refilled; tile kt+1 visible
    __builtin_amdgcn_sched_barrier(0);
    MSD_C_PRODUCT(0, 1)           // A hi x B lo
    MSD_C_PRODUCT(1, 0)           // A lo x B hi
    __builtin_amdgcn_sched_barrier(0);
    buf = buf + 1 == NS ? 0 : buf + 1;
  }
#undef MSD_C_PRODUCT
#undef MSD_C_READ

  // ---- epilogue on the four consumer waves: four 64-row passes through one slab ------------------------------------
  float* slab = reinterpret_cast<float*>(smem);
  char* const aux = smem + AUX_OFF;
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  MSD_TS_AT(7, blockIdx.x, 2)
  __syncthreads();   // every consumer is done with the ring (the loaders' last DMA landed before B(nk-2)): slab + aux
#pragma unroll
  for (int q = 0; q < 4; ++q) epi.template prefetch<HM, BN, 0>(aux + q * p.aux_half, m0 + q * HM, n0, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MSD_TS_AT(7, blockIdx.x, 3)
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    __syncthreads();   // h = 0: aux rows visible; later: the previous pass's slab has been read
    if (wm == (h >> 1)) {
#pragma unroll
      for (int i = 0; i < FM / 2; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const f32x4 a = acc[(h & 1) * (FM / 2) + i][j];
          *reinterpret_cast<float4*>(slab + (size_t)(i * 16 + lm) * LDS_LD + wn * WN + j * 16 + ln) =
              make_float4(a[0] * kWScaleInv, a[1] * kWScaleInv, a[2] * kWScaleInv, a[3] * kWScaleInv);
        }
    }
    const char* auxq = aux + h * p.aux_half;
    epi.template stats<HM, LDS_LD>(slab, m0 + h * HM, tid, auxq);
    __syncthreads();
    epi.template run<HM, BN, LDS_LD>(slab, m0 + h * HM, n0, tid, auxq, /*stats_done=*/true, SatFlag{p.sat, p.sat_tag});
    if (h == 0) { MSD_TS_AT(7, blockIdx.x, 4) }
  }
  MSD_TS_AT(7, blockIdx.x, 5)
  MSD_TS_END(7, blockIdx.x, gridDim.x)
}

template <int NP, int BN>
constexpr int gemm_h16_ls_smem() { return 3 * NP * (256 + BN) * 64; }

template <int NP, int BN, class Epi>
inline hipError_t gemm_h16_ls_prepare() {
  constexpr int smem = gemm_h16_ls_smem<NP, BN>();
  const hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_ls_kernel<NP, BN, Epi, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_ls_kernel<NP, BN, Epi, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  return a != hipSuccess ? a : b;
}

// does this problem fit?  (whole tiles, at least a ring of K-tiles)
template <int BN>
inline bool gemm_h16_ls_fits(int M, int N, int K) { return M % 256 == 0 && N % BN == 0 && K % kPairBK == 0 && K / kPairBK >= 3; }

template <int NP, int BN, class Epi>
inline hipError_t launch_gemm_h16_ls(GemmParams p, const Epi& epi, hipStream_t stream) {
  static const hipError_t attr = gemm_h16_ls_prepare<NP, BN, Epi>();
  if (attr != hipSuccess) return attr;
  p.aux_half = gemm_h16_pair_aux_half<NP, 128, BN>(epi);   // per 64-row pass
  constexpr int smem = gemm_h16_ls_smem<NP, BN>();
  const int rx = p.xcd_rows, cx = 8 / rx;
  const int grid = 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / 256 + rx - 1) / rx);
  if (prefetch_kind(p.pf) >= 1) hipLaunchKernelGGL((gemm_h16_ls_kernel<NP, BN, Epi, 1>), dim3(grid), dim3(512), smem, stream, p, epi);
  else hipLaunchKernelGGL((gemm_h16_ls_kernel<NP, BN, Epi, 0>), dim3(grid), dim3(512), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
