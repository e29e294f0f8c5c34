// XCD-resident kernel chains: several dependent phases of a decoder layer in ONE launch.
//
// Why.  At B = 1 a DDPM step is 111 short launches; each kernel boundary costs ~3.8 us (2.4 us of bare
// dispatch + the ramp of a cold kernel), ~0.4 ms of a 1.17 ms step.  A device-wide barrier inside a kernel is
// no cheaper (3.5 us measured, and data written on one XCD is NOT visible to loads from another XCD without an
// L2 write-back: tools/ubench/xcd_sync.hip, profiles/r02_xcd_sync.log) -- but the 32 CUs of ONE XCD share
// one L2, and among them a barrier costs 0.85 us and plain stores -> (L1-bypassing) loads are coherent.
//
// How.  Every phase of a layer between two self-attentions is ROW-LOCAL (GEMMs, norms, cross-attention over the
// cached K/V, MLP): give each XCD whole 64-row tiles of the activations (block b runs on XCD b % 8 -- checked
// by the probe above) and let its 32 blocks walk the column tiles of phase after phase, separated by the
// XCD-local barrier below.  Rows never leave their XCD inside the chain, so no cross-XCD visibility is ever
// needed; the kernel boundary stays only where rows mix (self-attention needs the K/V of every row).
// Operands another block of the same launch produced are loaded with sc1 (gemm_tile<.., CP = 16>).
#pragma once
#include "gemm_h16.h"

namespace msd {

#ifndef MSD_CHAIN_CP
#define MSD_CHAIN_CP 16
#endif
constexpr int kChainCP = MSD_CHAIN_CP;   // 16 = sc1: bypass the CU's L1, hit the XCD's L2
constexpr int kChainSpinLimit = 4000000;
constexpr int kBarStride = 64;        // one counter per XCD, 256 B apart; word 1 of a slot = the XCC_ID its blocks reported

// Barrier among the `n` blocks of this XCD.  `cnt` is monotonic (never reset): the arrival index tells the
// round.  Every wave first waits for its own stores to reach L2.  The spin is bounded: a lost block (e.g. fewer
// resident CUs than the grid assumes) raises *err instead of hanging the device.
__device__ __forceinline__ void xcd_barrier(unsigned* cnt, unsigned n, int* err) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old / n + 1) * n;
    int spins = 0;
    // wrap-safe compare (the counters are also zeroed at the start of every msd_* call: msd_api.hip reset_sync_words)
    while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      if (++spins > kChainSpinLimit) { atomicAdd(err, 1); break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// One GEMM phase: this XCD's row tiles x all column tiles, dealt to its blocks.
template <int NP, int BM, int BN, int NS, class Epi>
__device__ __forceinline__ void chain_gemm_phase(const GemmParams& p, const Epi& epi, int xcd, int slot, int nslot,
                                                 char* smem) {
  const int nbm = p.M / BM, nbn = p.N / BN;
  for (int bm = xcd; bm < nbm; bm += 8)
    for (int bn = slot; bn < nbn; bn += nslot) {
      gemm_tile<NP, BM, BN, NS, Epi, kChainCP>(p, epi, bm, bn, smem);
      __syncthreads();   // the epilogue slab aliases the operand ring of the next tile
    }
}

// MLP block + the next layer's fused QKV projection (network.py:241-256 -> next layer's :174-189):
//   g = gelu(y.wi_0) * (y.wi_1)  ->  x += g.wo ; y', ssq  ->  q|k|v = rstd (y'.Wqkv) + bW
template <int NP>
struct MlpChainParams {
  GemmParams g_in;  EpiGeglu<NP> e_in;
  GemmParams g_out; EpiResidualNorm<NP> e_out;
  GemmParams g_qkv; EpiQKV<NP> e_qkv;
  int has_qkv;       // 0 after the last layer (the final projection follows)
  WeightPrefetch pf; // MSD_CHAIN=2: the weights of phases 1 / 2 (and nothing else), touched by the launch's prefetch wave
  unsigned* bar;     // [8][kBarStride]
  int* err;
};

template <int NP, int QKV_BN>
constexpr int mlp_chain_smem() {
  constexpr int a = gemm_h16_dma_smem<NP, 64, 128, 3, EpiGeglu<NP>>();
  constexpr int b = gemm_h16_dma_smem<NP, 64, 32, 4, EpiResidualNorm<NP>>();
  constexpr int c = gemm_h16_dma_smem<NP, 64, QKV_BN, 3, EpiQKV<NP>>();
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

template <int NP, int QKV_BN>
__global__ void __launch_bounds__(256) mlp_chain_kernel(MlpChainParams<NP> P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  unsigned* bar = P.bar + xcd * kBarStride;
  // The chain's correctness rests on an OBSERVED placement (block b runs on XCD b % 8): every block reports its
  // XCC_ID to its slot's placement word; a second value in one slot raises *err (bit 16 up), which fails the
  // msd_* call that launched the chain (msd_api.hip check_sync_words).
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 0xf) + 1u;
    const unsigned seen = atomicCAS(bar + 1, 0u, xcc);
    if (seen != 0u && seen != xcc) atomicAdd(P.err, 1 << 16);
  }
  chain_gemm_phase<NP, 64, 128, 3>(P.g_in, P.e_in, xcd, slot, nslot, smem);
  xcd_barrier(bar, nslot, P.err);
  chain_gemm_phase<NP, 64, 32, 4>(P.g_out, P.e_out, xcd, slot, nslot, smem);
  if (!P.has_qkv) return;
  xcd_barrier(bar, nslot, P.err);
  chain_gemm_phase<NP, 64, QKV_BN, 3>(P.g_qkv, P.e_qkv, xcd, slot, nslot, smem);
}

// ---- round 4: the same chain with the next phase's WEIGHT tiles pre-staged (MSD_CHAIN=2) ---------------------------
// "issue phase N+1's weight tiles into a slab that does not alias the ring BEFORE the
// xcd_barrier and while the consumer waves run phase N's epilogue".  Weights depend on nothing computed in the launch,
// so behind its main loop a block starts the LDS-DMA of the weight half of its NEXT tile's first ring stages
// (gemm_prestage_b, uncounted by the compiler), into a ring placed where neither this tile's epilogue slab nor its aux
// rows live; after the barrier only the activation half (L2-resident: the same XCD just wrote it) is fetched.
// LDS map (bytes; NP = 2, one block per CU, 160 KiB):
//   phase 0  gated MLP-in  64 x 128, 3 stages of 48 KiB   ring [0, 147456)          aux [147456, 156672)   slab [0, 34048)
//   phase 1  MLP-out       64 x 32,  4 stages of 24 KiB   ring [36864, 135168)      aux [135168, 146432)   slab [36864, 46336)
//   phase 2  QKV           64 x BN,  3 stages             ring [30720, ...)         aux behind it          (BN = 96: ends at 162816)
// Phase 1's whole ring sits between phase 0's slab and aux, so all four weight stages are pre-staged; phase 2's stage-0
// ACTIVATION half covers phase 1's slab (it is fetched after the barrier) and its weight halves avoid slab and aux of
// phase 1 for the first kChainPs2 stages (static_asserts below).
// The weights of phases 1 / 2 are additionally warmed into the memory-side cache by a prefetch wave at the start of the
// launch, as the stand-alone launches' predecessors do for them (round 2's chain started every phase HBM-cold).
constexpr int kChainOff1 = 36864, kChainOff2 = 30720;
template <int QKV_BN> constexpr int chain_ps2() { return QKV_BN == 96 ? 2 : 3; }

template <int NP, int BM, int BN, int NS>
struct PrestageHook {
  const GemmParams& p;   // the NEXT phase's problem
  bool on;               // false: nothing to stage (no tile in the next phase)
  int bn;
  unsigned ring;   // LDS byte address
  int nst;
  __device__ __forceinline__ void after_loop() const {
    if (on) gemm_prestage_b<NP, BM, BN, NS>(p, bn, ring, nst);
  }
};

template <int NP, int QKV_BN>
constexpr int mlp_chain_ps_smem() {
  constexpr int c_end = kChainOff2 + gemm_h16_dma_smem<NP, 64, QKV_BN, 3, EpiQKV<NP>>();
  constexpr int a_end = gemm_h16_dma_smem<NP, 64, 128, 3, EpiGeglu<NP>>();
  constexpr int b_end = kChainOff1 + gemm_h16_dma_smem<NP, 64, 32, 4, EpiResidualNorm<NP>>();
  return c_end > a_end ? (c_end > b_end ? c_end : b_end) : (a_end > b_end ? a_end : b_end);
}

template <int NP, int QKV_BN>
__global__ void __launch_bounds__(256 + 64) mlp_chain_ps_kernel(MlpChainParams<NP> P) {
  static_assert(NP == 2, "two-plane modes only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (threadIdx.x >= 256) {   // prefetch wave: phases 1 / 2 weights -> Infinity Cache, then it ends (no barrier counts it)
    prefetch_wave<2>(P.pf, blockIdx.x, gridDim.x, P.g_in.B[0]);
    return;
  }
  // ---- compile-time LDS map checks --------------------------------------------------------------------------------
  constexpr int S0 = 2 * (64 + 128) * 128, S1 = 2 * (64 + 32) * 128, S2 = 2 * (64 + QKV_BN) * 128;
  constexpr int SLAB0 = (64 * (128 + kSlabPad) + 64) * 4, AUX0 = 3 * S0;
  constexpr int SLAB1_LO = kChainOff1, SLAB1_HI = kChainOff1 + (64 * (32 + kSlabPad) + 64) * 4;
  constexpr int AUX1_LO = kChainOff1 + 4 * S1, AUX1_HI = AUX1_LO + EpiResidualNorm<NP>::template aux_bytes<64, 32>();
  constexpr int PS2 = chain_ps2<QKV_BN>();
  static_assert(kChainOff1 >= SLAB0 && AUX1_HI <= AUX0, "phase 1 (ring + aux) must sit between phase 0's slab and aux");
  static_assert(kChainOff2 <= SLAB1_LO && kChainOff2 + 2 * 64 * 128 >= SLAB1_HI, "phase 2's stage-0 activation half must cover phase 1's slab");
  static_assert(kChainOff2 + (PS2 - 1) * S2 + S2 <= AUX1_LO, "phase 2's pre-staged weight halves must end below phase 1's aux rows");
  static_assert(mlp_chain_ps_smem<NP, QKV_BN>() <= 160 * 1024, "LDS");

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  unsigned* bar = P.bar + xcd * kBarStride;
  if (threadIdx.x == 0) {   // placement check, as in mlp_chain_kernel
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 0xf) + 1u;
    const unsigned seen = atomicCAS(bar + 1, 0u, xcc);
    if (seen != 0u && seen != xcc) atomicAdd(P.err, 1 << 16);
  }
  // tiles of a phase: row tiles bm = xcd, xcd + 8, ...; column tiles bn = slot, slot + nslot, ...  Only the FIRST tile
  // of a block's next phase is pre-staged (at one song every block has at most one tile per phase anyway).
  const int nbm = P.g_in.M / 64;
  const int nbn0 = P.g_in.N / 128, nbn1 = P.g_out.N / 32, nbn2 = P.g_qkv.N / QKV_BN;
  const bool own_rows = xcd < nbm;
  char* const ring0 = smem;
  char* const ring1 = smem + kChainOff1;
  char* const ring2 = smem + kChainOff2;
  const unsigned lds_base = (unsigned)(size_t)smem;   // low 32 bits of a generic LDS pointer = its LDS address
  // ---- phase 0: gated MLP input -------------------------------------------------------------------------------------
  {
    const int nk1 = P.g_out.K / kGemmBK;
    PrestageHook<NP, 64, 32, 4> h1{P.g_out, own_rows && slot < nbn1, slot, lds_base + kChainOff1, nk1 < 4 ? nk1 : 4};
    bool first = true;
    for (int bm = xcd; bm < nbm; bm += 8)
      for (int bn = slot; bn < nbn0; bn += nslot) {
        // the hook fires behind the LAST tile's loop: an earlier tile's slab / ring would overwrite the staged bytes
        const bool last = bm + 8 >= nbm && bn + nslot >= nbn0;
        if (last) gemm_tile<NP, 64, 128, 3, EpiGeglu<NP>, kChainCP, kPfNone, 1, 0>(P.g_in, P.e_in, bm, bn, ring0, 0, 0, h1);
        else gemm_tile<NP, 64, 128, 3, EpiGeglu<NP>, kChainCP>(P.g_in, P.e_in, bm, bn, ring0);
        __syncthreads();
        first = false;
      }
    if (first) h1.after_loop();   // a block without a phase-0 tile still stages its phase-1 weights
  }
  xcd_barrier(bar, nslot, P.err);
  // ---- phase 1: MLP output projection (+ residual, next norm's planes) -----------------------------------------------
  {
    const int nk2 = P.g_qkv.K / kGemmBK;
    PrestageHook<NP, 64, QKV_BN, 3> h2{P.g_qkv, P.has_qkv && own_rows && slot < nbn2, slot, lds_base + kChainOff2, nk2 < PS2 ? nk2 : PS2};
    bool first = true;
    for (int bm = xcd; bm < nbm; bm += 8)
      for (int bn = slot; bn < nbn1; bn += nslot) {
        const bool last = bm + 8 >= nbm && bn + nslot >= nbn1;
        const bool staged = first && bm == xcd && bn == slot;
        if (staged && last) gemm_tile<NP, 64, 32, 4, EpiResidualNorm<NP>, kChainCP, kPfNone, 1, 4>(P.g_out, P.e_out, bm, bn, ring1, 0, 0, h2);
        else if (staged) gemm_tile<NP, 64, 32, 4, EpiResidualNorm<NP>, kChainCP, kPfNone, 1, 4>(P.g_out, P.e_out, bm, bn, ring1);
        else if (last) gemm_tile<NP, 64, 32, 4, EpiResidualNorm<NP>, kChainCP, kPfNone, 1, 0>(P.g_out, P.e_out, bm, bn, ring1, 0, 0, h2);
        else gemm_tile<NP, 64, 32, 4, EpiResidualNorm<NP>, kChainCP>(P.g_out, P.e_out, bm, bn, ring1);
        __syncthreads();
        first = false;
      }
    if (first) h2.after_loop();
  }
  if (!P.has_qkv) return;
  xcd_barrier(bar, nslot, P.err);
  // ---- phase 2: the next layer's fused QKV projection ---------------------------------------------------------------
  {
    bool first = true;
    for (int bm = xcd; bm < nbm; bm += 8)
      for (int bn = slot; bn < nbn2; bn += nslot) {
        if (first && bm == xcd && bn == slot) gemm_tile<NP, 64, QKV_BN, 3, EpiQKV<NP>, kChainCP, kPfNone, 1, PS2>(P.g_qkv, P.e_qkv, bm, bn, ring2);
        else gemm_tile<NP, 64, QKV_BN, 3, EpiQKV<NP>, kChainCP>(P.g_qkv, P.e_qkv, bm, bn, ring2);
        __syncthreads();
        first = false;
      }
  }
}

template <int NP, int QKV_BN>
inline hipError_t mlp_chain_ps_prepare() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_ps_kernel<NP, QKV_BN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, mlp_chain_ps_smem<NP, QKV_BN>());
}

template <int NP, int QKV_BN>
inline hipError_t launch_mlp_chain_ps(const MlpChainParams<NP>& P, int cus, hipStream_t stream) {
  static const hipError_t attr = mlp_chain_ps_prepare<NP, QKV_BN>();
  if (attr != hipSuccess) return attr;
  constexpr int smem = mlp_chain_ps_smem<NP, QKV_BN>();
  hipLaunchKernelGGL((mlp_chain_ps_kernel<NP, QKV_BN>), dim3(cus), dim3(256 + 64), smem, stream, P);
  return hipGetLastError();
}

template <int NP, int QKV_BN>
inline hipError_t mlp_chain_prepare() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_kernel<NP, QKV_BN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, mlp_chain_smem<NP, QKV_BN>());
}

// grid = one block per CU (the LDS footprint keeps it at one): `cus` must be a multiple of 8
template <int NP, int QKV_BN>
inline hipError_t launch_mlp_chain(const MlpChainParams<NP>& P, int cus, hipStream_t stream) {
  static const hipError_t attr = mlp_chain_prepare<NP, QKV_BN>();
  if (attr != hipSuccess) return attr;
  constexpr int smem = mlp_chain_smem<NP, QKV_BN>();
  hipLaunchKernelGGL((mlp_chain_kernel<NP, QKV_BN>), dim3(cus), dim3(256), smem, stream, P);
  return hipGetLastError();
}

}  // namespace msd
