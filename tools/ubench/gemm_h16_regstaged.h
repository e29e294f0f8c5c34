// Register-staged predecessor of the product GEMM (global -> register ring -> ds_write -> LDS), kept ONLY for
// the ablation micro-benchmarks (tools/ubench/gemm_bench.hip): the product path uses the LDS-DMA kernel of
// music-spectrogram-diffusion_amd/csrc/gemm_h16.h, which this header includes for the shared pieces.
#pragma once
#define MSD_EPI_AUX_OPTIONAL 1   // this kernel has no aux LDS region: the epilogues read their operands from global memory
#include "../../music-spectrogram-diffusion_amd/csrc/gemm_h16.h"

namespace msd {

#ifndef MSD_ABL
#define MSD_ABL 0  // ablation switch for tools/ubench/gemm_bench.hip; 0 = the product kernel
#endif

template <int NP, int BM, int BN, int R, class Epi>
__global__ void __launch_bounds__(256) gemm_h16_kernel(GemmParams p, Epi epi) {
  constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile (2 x 2 waves)
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  constexpr int A_LD = BM / 32, B_LD = BN / 32;  // 16-byte loads per thread per plane
  constexpr int LDS_LD = BN + kSlabPad;
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile rows must be a multiple of 32");
  static_assert((BM * LDS_LD + BM) * 4 <= 2 * STAGE_BYTES, "epilogue slab must fit the operand LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile mapping -------------------------------------------------
  const int nbm = p.M / BM, nbn = p.N / BN;
  int bm, bn;
  {
    const int b = blockIdx.x;
    if ((nbn & 7) == 0) {
      const int xcd = b & 7, t = b >> 3;  // t-th block of this XCD
      bm = t % nbm;
      bn = (t / nbm) * 8 + xcd;
    } else {
      bm = b % nbm;
      bn = b / nbm;
    }
  }
  const int m0 = bm * BM, n0 = bn * BN;

  // global source of this thread: rows (tid>>3) + 32*i, 16-byte chunk tid&7
  const int ld_row = tid >> 3, ld_chunk = tid & 7;
  const h16_t* ga[NP];
  const h16_t* gb[NP];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    ga[pl] = p.A[pl] + (size_t)(m0 + ld_row) * p.lda + ld_chunk * 8;
    gb[pl] = p.B[pl] + (size_t)(n0 + ld_row) * p.ldb + ld_chunk * 8;
  }
  const size_t a_step = (size_t)32 * p.lda, b_step = (size_t)32 * p.ldb;

  u32x4 ra[R][NP][A_LD], rb[R][NP][B_LD];
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define MSD_G_LOAD(S, KT)                                                                  \
  if (MSD_ABL != 1 || (KT) <= R) {                                                         \
    const int k0_ = (KT) * kGemmBK;                                                        \
    _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                    \
      _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                     \
          ra[S][pl][i] = *reinterpret_cast<const u32x4*>(ga[pl] + i * a_step + k0_);       \
      _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                     \
          rb[S][pl][i] = *reinterpret_cast<const u32x4*>(gb[pl] + i * b_step + k0_);       \
    }                                                                                      \
  }
#define MSD_G_STORE(S, BUF)                                                                \
  if (MSD_ABL != 4) {                                                                      \
    char* base_ = smem + (BUF) * STAGE_BYTES;                                              \
    _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                    \
      _Pragma("unroll") for (int i = 0; i < A_LD; ++i)                                     \
          *reinterpret_cast<u32x4*>(base_ + pl * A_BYTES + lds_tile_off(ld_row + 32 * i, ld_chunk)) = ra[S][pl][i]; \
      _Pragma("unroll") for (int i = 0; i < B_LD; ++i)                                     \
          *reinterpret_cast<u32x4*>(base_ + NP * A_BYTES + pl * B_BYTES + lds_tile_off(ld_row + 32 * i, ld_chunk)) = rb[S][pl][i]; \
    }                                                                                      \
  }
  // D[n][m] orientation (first operand = W^T fragment): lane holds C[m = l&15][n = (l>>4)*4 + r]
#define MSD_G_COMPUTE(BUF)                                                                 \
  {                                                                                        \
    const char* base_ = smem + (BUF) * STAGE_BYTES;                                        \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                     \
      mfma_h16x8 fa[NP][FM], fb[NP][FN];                                                  \
      const int c_ = kk * 4 + (lane >> 4);                                                 \
      if (MSD_ABL != 3 || kt == 0)                                                         \
      _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                  \
        _Pragma("unroll") for (int i = 0; i < FM; ++i)                                     \
            fa[pl][i] = *reinterpret_cast<const mfma_h16x8*>(                             \
                base_ + pl * A_BYTES + lds_tile_off(wm * WM + i * 16 + (lane & 15), c_));  \
        _Pragma("unroll") for (int j = 0; j < FN; ++j)                                     \
            fb[pl][j] = *reinterpret_cast<const mfma_h16x8*>(                             \
                base_ + NP * A_BYTES + pl * B_BYTES + lds_tile_off(wn * WN + j * 16 + (lane & 15), c_)); \
      }                                                                                    \
      if (MSD_ABL == 2) {                                                                  \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int j = 0; j < FN; ++j) \
          _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                              \
            asm volatile("" ::"v"(fa[pl][i]), "v"(fb[pl][j]));                             \
          }                                                                                \
      } else                                                                               \
      _Pragma("unroll") for (int i = 0; i < FM; ++i)                                       \
      _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                     \
        acc[i][j] = MSD_MFMA_16X16X32(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0); \
        if (NP == 2) {                                                                     \
          acc[i][j] = MSD_MFMA_16X16X32(fb[NP - 1][j], fa[0][i], acc[i][j], 0, 0, 0); \
          acc[i][j] = MSD_MFMA_16X16X32(fb[0][j], fa[NP - 1][i], acc[i][j], 0, 0, 0); \
        }                                                                                  \
      }                                                                                    \
    }                                                                                      \
  }
#define MSD_PIN() __builtin_amdgcn_sched_barrier(0)

  // Tile kt lives in LDS buffer kt & 1; register stage kt % R holds tile kt while in
  // flight.  Invariant at the top of iteration kt: LDS[kt&1] = tile kt (visible to
  // all waves), register stages hold tiles kt+1 .. kt+R (loads issued, maybe in flight).
  const int nk = p.K / kGemmBK;
  int kt = 0;
  if (nk > 2 * R) {
    // prologue: tile 0 through registers into LDS, then fill the ring with tiles 1..R
    MSD_G_LOAD(0, 0)
    MSD_PIN();
    MSD_G_STORE(0, 0)
    MSD_PIN();
#pragma unroll
    for (int s = 0; s < R; ++s) {
      MSD_G_LOAD(((s + 1) % R), s + 1)
      MSD_PIN();
    }
    __syncthreads();
    // steady state, unrolled by 2R so that ring stage AND LDS buffer are compile-time
    for (; kt + 2 * R + R < nk; kt += 2 * R) {
#pragma unroll
      for (int u = 0; u < 2 * R; ++u) {
        // tile kt+u is in LDS[(u)&1] (kt is a multiple of 2R, hence even); tile kt+u+1 is
        // in ring stage (u+1)%R: move it to the other LDS buffer, refill the stage
        MSD_G_STORE(((u + 1) % R), ((u + 1) & 1))
        MSD_PIN();
        MSD_G_LOAD(((u + 1) % R), kt + u + 1 + R)
        MSD_PIN();
        MSD_G_COMPUTE((u & 1))
        MSD_PIN();
        __syncthreads();
      }
    }
    // drain with conditional refills
    for (; kt < nk; kt += 2 * R) {
#pragma unroll
      for (int u = 0; u < 2 * R; ++u) {
        if (kt + u < nk) {
          if (kt + u + 1 < nk) MSD_G_STORE(((u + 1) % R), ((u + 1) & 1))
          if (kt + u + 1 + R < nk) MSD_G_LOAD(((u + 1) % R), kt + u + 1 + R)
          MSD_G_COMPUTE((u & 1))
          __syncthreads();
        }
      }
    }
  } else {
    // short K: plain double-buffered loop through ring stage 0
    MSD_G_LOAD(0, 0)
    MSD_G_STORE(0, 0)
    __syncthreads();
    for (; kt < nk; ++kt) {
      if (kt + 1 < nk) MSD_G_LOAD(0, kt + 1)
      if (kt & 1) { MSD_G_COMPUTE(1) } else { MSD_G_COMPUTE(0) }
      if (kt + 1 < nk) {
        if (kt & 1) { MSD_G_STORE(0, 0) } else { MSD_G_STORE(0, 1) }
      }
      __syncthreads();
    }
  }
#undef MSD_G_LOAD
#undef MSD_G_STORE
#undef MSD_G_COMPUTE
#undef MSD_PIN

  // ---- accumulators -> LDS slab (operand buffers are dead after the last barrier) ---
  float* slab = reinterpret_cast<float*>(smem);
  const int lm = lane & 15, ln = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
      *reinterpret_cast<float4*>(slab + (size_t)(wm * WM + i * 16 + lm) * LDS_LD + wn * WN + j * 16 + ln) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  __syncthreads();
  epi.template run<BM, BN, LDS_LD>(slab, m0, n0, tid);
}

template <int NP, int BM, int BN, int R, class Epi>
constexpr int gemm_h16_smem() { return 2 * NP * (BM + BN) * 128; }

// one-time opt-in to > 64 KiB dynamic LDS; call for every instantiation OUTSIDE stream capture
template <int NP, int BM, int BN, int R, class Epi>
inline hipError_t gemm_h16_prepare() {
  constexpr int smem = gemm_h16_smem<NP, BM, BN, R, Epi>();
  if (smem < 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_kernel<NP, BM, BN, R, Epi>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

template <int NP, int BM, int BN, int R, class Epi>
inline hipError_t launch_gemm_h16(const GemmParams& p, const Epi& epi, hipStream_t stream) {
  // (> 64 KiB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize, set once in
  //  msd_api.hip:set_func_attrs -- never during stream capture)
  constexpr int smem = gemm_h16_smem<NP, BM, BN, R, Epi>();
  auto kern = gemm_h16_kernel<NP, BM, BN, R, Epi>;
  static const hipError_t attr = gemm_h16_prepare<NP, BM, BN, R, Epi>();
  if (attr != hipSuccess) return attr;
  const int grid = (p.M / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
