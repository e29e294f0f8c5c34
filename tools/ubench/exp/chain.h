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
