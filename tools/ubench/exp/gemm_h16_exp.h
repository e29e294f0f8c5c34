// Experiment kernels of the 16-bit-plane GEMM that were measured and rejected (docs/history.md); compiled only into
// the MSD_EXPERIMENTS build of the library (tools/ubench/exp/build_exp.py), never into the product:
//   * gemm_h16_splitk_kernel  -- 4-way split-K MLP output projection with an XCD-local reduce-scatter (+2.5 % step)
//   * gemm_h16_dual_kernel    -- two independent GEMMs in one grid (hoisted cross-attention query projection, 0..+1 %)
//   * EpiAddStoreH16          -- the hoist's second half
// Included at the end of csrc/gemm_h16.h when MSD_EXPERIMENTS is set.
#pragma once

namespace msd {

// Split-K variant for the one GEMM of the step whose K is long and whose N is short (the MLP output projection,
// M x D x F: 64 x 32 tiles over 32 K-tiles spent 9 of their 15.5 us streaming (64 + 32) rows per K-tile; SK blocks of
// a 64 x 128 tile stream (64 + 128) rows over K / SK).  Grid = 8 XCDs x (tiles per XCD x SK) blocks, all resident
// at once (host checks tiles * SK <= CUs: the blocks of a tile wait for each other).  Block b runs on XCD b % 8
// (observed placement; verified in the kernel): XCD (xr, xc) of the xcd_rows x (8 / xcd_rows) grid owns the row
// tiles [xr nbm / RX, +nbm / RX) x column tiles [xc nbn / CX, +nbn / CX); slot = b / 8 = (tile of that XCD, split).
template <int NP, int BM, int BN, int NS, int SK, class Epi, int PF = kPfNone>
__global__ void __launch_bounds__(256 + pf_threads(PF)) gemm_h16_splitk_kernel(GemmParams p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (kPfWave && PF != kPfNone) {
    if (threadIdx.x >= 256) {
      prefetch_wave<PF>(p.pf, blockIdx.x, gridDim.x, p.B[0]);
      return;
    }
  }
  const int nbm = p.M / BM, nbn = p.N / BN;
  const int RX = p.xcd_rows, CX = 8 / RX;
  const int nbm_x = nbm / RX, nbn_x = nbn / CX;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int g = slot / SK, ks = slot % SK;
  if (g >= nbm_x * nbn_x) return;
  const int bm = (xcd / CX) * nbm_x + g / nbn_x, bn = (xcd % CX) * nbn_x + g % nbn_x;
  gemm_tile<NP, BM, BN, NS, Epi, 0, PF, SK>(p, epi, bm, bn, smem, ks, bm * nbn + bn);
}

// Two independent GEMMs of one tile shape in ONE launch (no data flows between them): blocks [0, n2) run problem 2,
// the rest problem 1.  Used by the HOISTED cross-attention query
// projection (msd_api.hip decoder_layers): its first half rides on the QKV launch's idle CUs, its second half on the
// launch of the self-attention output projection -- a launch boundary less per layer.  Each problem keeps its own
// XCD-aware tile map (n2 is a multiple of 8, so a block's XCD is the same in the launch-wide and in the
// problem-local numbering).
template <int NP, int BM, int BN, int NS, class Epi1, class Epi2, int PF = kPfNone>
__global__ void __launch_bounds__(256 + pf_threads(PF)) gemm_h16_dual_kernel(GemmParams p1, Epi1 e1, GemmParams p2, Epi2 e2, int n2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (kPfWave && PF != kPfNone) {
    if (threadIdx.x >= 256) {   // prefetch wave: every block of the launch takes part
      prefetch_wave<PF>(p2.pf, blockIdx.x, gridDim.x, p1.B[0]);
      return;
    }
  }
  const bool second = (int)blockIdx.x < n2;
  const GemmParams& p = second ? p2 : p1;
  const int b = second ? (int)blockIdx.x : (int)blockIdx.x - n2;
  const int nbm = p.M / BM, nbn = p.N / BN;
  const int RX = p.xcd_rows, CX = 8 / RX;
  const int xcd = b & 7, tt = b >> 3;
  const int nbm_x = (nbm + RX - 1) / RX, nbn_x = (nbn + CX - 1) / CX;
  int bm, bn;
  if (p.xcd_walk_n) {
    bm = (tt / nbn_x) * RX + xcd / CX; bn = (tt % nbn_x) * CX + xcd % CX;
  } else {
    bm = (tt % nbm_x) * RX + xcd / CX; bn = (tt / nbm_x) * CX + xcd % CX;
  }
  if (bn >= nbn || bm >= nbm) return;
  // In-epilogue prefetch builds (MSD_PF_WAVE=0): the touches ride on problem 2's blocks only (p2.pf_nblk = n2): ONE
  // prefetch site in the kernel, in the arm behind which nothing but its own epilogue runs --
  // tools/check_prefetch_regs.py follows the control flow, and the structurised two-arm layout of this kernel
  // re-tests its condition after the first arm, which no text tool can see through.
  if (!second) {
    gemm_tile<NP, BM, BN, NS, Epi1, 0, kPfNone>(p1, e1, bm, bn, smem);
    return;
  }
  gemm_tile<NP, BM, BN, NS, Epi2, 0, PF>(p2, e2, bm, bn, smem);
}

// C (row-major 16-bit planes) = acc + addend[m][n] (fp32): the second half of the hoisted cross-attention query
// projection adds the first half, which an earlier launch left in float32.  The addend tile is prefetched into the
// aux LDS region like the residual tile of EpiResidualNorm (BN == 32), so the epilogue issues no global load.
template <int NP>
struct EpiAddStoreH16 {
  h16_t* out[2];
  int ldc;
  const float* addend;
  int ld_add;
  template <int BM, int BN> static constexpr int aux_bytes() { return BN == 32 ? BM * 128 : 0; }
  template <int BM, int BN, int CP = 0>
  __device__ void prefetch(char* aux, int m0, int n0, int wave, int lane) const {
    if (BN != 32) return;
    for (int i = wave; i < BM / 8; i += 4)
      __builtin_amdgcn_global_load_lds(
          (aux_gptr_t)(addend + (size_t)(m0 + 8 * i + (lane >> 3)) * ld_add + n0 + (lane & 7) * 4),
          lds_ptr_of(aux + i * 1024), 16, 0, CP);
  }
  template <int BM, int LD>
  __device__ void stats(float*, int, int, const char*) const {}
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    const bool pre = BN == 32 && aux_present(aux);
    typedef const __attribute__((address_space(3))) f32x4* lds_cf32x4;
    RangeCheck rc;
    for (int item = tid; item < BM * BN / 8; item += 256) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
      float v[8];
      tile_row8<LD>(s0, m, n, v);
      f32x4 a, b;
      if (pre) {
        lds_cf32x4 xs = (lds_cf32x4)(aux);
        a = xs[(m * BN + n) / 4]; b = xs[(m * BN + n) / 4 + 1];
      } else {
        const float* pa = addend + (size_t)(m0 + m) * ld_add + n0 + n;
        a = *reinterpret_cast<const f32x4*>(pa); b = *reinterpret_cast<const f32x4*>(pa + 4);
      }
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
      v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
      store_h16x8<NP>(out, (size_t)(m0 + m) * ldc + n0 + n, v, rc);
    }
    rc.commit(sf.p, sf.tag);
  }
};

// ---- dual launch ----------------------------------------------------------------------------------------------
template <int NP, int BM, int BN, int NS, class Epi1, class Epi2>
constexpr int gemm_h16_dual_smem() {
  constexpr int a = gemm_h16_dma_smem<NP, BM, BN, NS, Epi1>(), b = gemm_h16_dma_smem<NP, BM, BN, NS, Epi2>();
  return a > b ? a : b;
}
template <int NP, int BM, int BN, int NS, class Epi1, class Epi2>
inline hipError_t gemm_h16_dual_prepare() {
  constexpr int smem = gemm_h16_dual_smem<NP, BM, BN, NS, Epi1, Epi2>();
  if (smem < 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_dual_kernel<NP, BM, BN, NS, Epi1, Epi2, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_dual_kernel<NP, BM, BN, NS, Epi1, Epi2, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  return e != hipSuccess ? e : r;
}
inline int gemm_grid_blocks(const GemmParams& p, int BM, int BN) {
  const int rx = p.xcd_rows, cx = 8 / rx;
  return 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / BM + rx - 1) / rx);
}
// p1 / e1: the first problem; p2 / e2: the second.  The weight prefetch target (at most one) is taken from p1.pf.
template <int NP, int BM, int BN, int NS, class Epi1, class Epi2>
inline hipError_t launch_gemm_h16_dual(const GemmParams& p1, const Epi1& e1, GemmParams p2, const Epi2& e2, hipStream_t stream) {
  constexpr int smem = gemm_h16_dual_smem<NP, BM, BN, NS, Epi1, Epi2>();
  static const hipError_t attr = gemm_h16_dual_prepare<NP, BM, BN, NS, Epi1, Epi2>();
  if (attr != hipSuccess) return attr;
  const int n1 = gemm_grid_blocks(p1, BM, BN), n2 = gemm_grid_blocks(p2, BM, BN);
  p2.pf = p1.pf;
  p2.pf_nblk = n2;
  if (NP == 2 && prefetch_kind(p1.pf) >= 1)
    hipLaunchKernelGGL((gemm_h16_dual_kernel<NP, BM, BN, NS, Epi1, Epi2, 1>), dim3(n1 + n2), dim3(256 + pf_threads(1)), smem, stream, p1, e1, p2, e2, n2);
  else
    hipLaunchKernelGGL((gemm_h16_dual_kernel<NP, BM, BN, NS, Epi1, Epi2, 0>), dim3(n1 + n2), dim3(256), smem, stream, p1, e1, p2, e2, n2);
  return hipGetLastError();
}

// ---- split-K launch -------------------------------------------------------------------------------------------
template <int NP, int BM, int BN, int NS, int SK, class Epi>
constexpr int gemm_h16_splitk_smem() { return NS * NP * (BM + BN) * 128 + Epi::template aux_bytes<BM, BN / SK>(); }

// (rows, columns) of the XCD grid for a split-K launch: as many column groups as divide the column tiles
inline int splitk_xcd_rows(int nbm, int nbn) {
  for (int cx = 4; cx >= 1; cx >>= 1)
    if (nbn % cx == 0 && nbm % (8 / cx) == 0) return 8 / cx;
  return 0;   // no grid fits: the caller keeps the plain kernel
}

// floats / words of workspace a split-K GEMM of M x N needs
template <int BM, int BN, int SK>
inline size_t splitk_part_floats(int M, int N) { return (size_t)(M / BM) * (N / BN) * SK * BM * BN; }

template <int NP, int BM, int BN, int NS, int SK, class Epi>
inline hipError_t gemm_h16_splitk_prepare() {
  constexpr int smem = gemm_h16_splitk_smem<NP, BM, BN, NS, SK, Epi>();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_splitk_kernel<NP, BM, BN, NS, SK, Epi, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_splitk_kernel<NP, BM, BN, NS, SK, Epi, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  return e != hipSuccess ? e : r;
}

// p.xcd_rows must come from splitk_xcd_rows(); p.sk_* must be set; (M/BM) * (N/BN) * SK blocks must be co-resident
template <int NP, int BM, int BN, int NS, int SK, class Epi>
inline hipError_t launch_gemm_h16_splitk(const GemmParams& p, const Epi& epi, hipStream_t stream) {
  constexpr int smem = gemm_h16_splitk_smem<NP, BM, BN, NS, SK, Epi>();
  static const hipError_t attr = gemm_h16_splitk_prepare<NP, BM, BN, NS, SK, Epi>();
  if (attr != hipSuccess) return attr;
  const int grid = (p.M / BM) * (p.N / BN) * SK;   // = 8 XCDs x tiles per XCD x SK
  if (prefetch_kind(p.pf) >= 1)
    hipLaunchKernelGGL((gemm_h16_splitk_kernel<NP, BM, BN, NS, SK, Epi, 1>), dim3(grid), dim3(256 + pf_threads(1)), smem, stream, p, epi);
  else
    hipLaunchKernelGGL((gemm_h16_splitk_kernel<NP, BM, BN, NS, SK, Epi, 0>), dim3(grid), dim3(256), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
