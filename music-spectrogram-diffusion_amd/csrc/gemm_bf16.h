// bf16 / bf16x3 MFMA GEMM for the small-M (M = 256..2304) transformer projections.
//
//   C[M,N] (+)= A[M,K] . W[K,N]        A: bf16 planes [M,K] row-major
//                                      W: packed at load time as W^T planes [N,K]
// NP = 1: plain bf16 operands.  NP = 2 ("bf16x3"): A = Ah+Al, W = Wh+Wl and the
// product is Ah.Wh + Ah.Wl + Al.Wh (three v_mfma_f32_16x16x32_bf16 per tile),
// fp32 accumulate: fp32-class accuracy at the bf16 MFMA rate / 3.
//
// Regime: M is 256-512, so a launch has few output tiles and each tile's K loop is
// a serial chain of memory latencies.  Structure chosen for that regime:
//   * block = 4 waves that ALL own the same BM x BN output tile and split K between
//     them (wave w takes the 32-wide K-steps w, w+4, ...): 4x shorter dependent
//     chains, and no operand byte is fetched twice inside a block;
//   * both operands are K-contiguous, so A and W^T fragments are plain 16-byte
//     global loads (lane: row l&15, k-chunk l>>4) straight into the MFMA operand
//     registers through a DEPTH-deep, statically indexed register ring -- no LDS,
//     no barrier and no s_waitcnt vmcnt(0) inside the K loop;
//   * the four partial tiles are summed through LDS (two rounds, 2 slabs) and the
//     epilogue runs on the summed tile with a row-of-8 (or column-of-8) item per
//     thread, i.e. fully coalesced 16/32-byte stores whatever the MFMA layout was.
// blockIdx -> tile is XCD-aware (block b runs on XCD b % 8): the BM-blocks of one
// column slice are consecutive on ONE XCD, so a weight slice is fetched from HBM
// once into that XCD's L2.
#pragma once
#include "common.h"

namespace msd {

struct GemmParams {
  const bf16_t* A[2];
  const bf16_t* B[2];
  int lda, ldb;
  int M, N, K;
};

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

__device__ __forceinline__ mfma_bf16x8 as_frag(uint4 v) {
  union { uint4 u; mfma_bf16x8 f; } c;
  c.u = v;
  return c.f;
}

__device__ __forceinline__ mfma_bf16x8 ld_frag16(const bf16_t* p) {
  return as_frag(*reinterpret_cast<const uint4*>(p));
}

// key position permutation inside each group of 16 keys of a V^T row, so that the
// attention kernel's P^T fragment (32x32 MFMA C layout) lines up with 16-byte
// V^T loads: offset o -> 8*((o>>2)&1) + (o&3) + 4*(o>>3)   (see attention.h)
__device__ __forceinline__ int vt_perm16(int o) { return 8 * ((o >> 2) & 1) + (o & 3) + 4 * (o >> 3); }

constexpr int kSlabPad = 4;

template <int NP, int BM, int BN, int DEPTH, bool PIN, class Epi>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(GemmParams p, Epi epi) {
  constexpr int FM = BM / 16, FN = BN / 16;
  constexpr int LDS_LD = BN + kSlabPad;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* slab = reinterpret_cast<float*>(smem_raw);  // [2][BM][LDS_LD]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware tile mapping -------------------------------------------------
  const int nbm = p.M / BM, nbn = p.N / BN;
  int bm, bn;
  {
    const int b = blockIdx.x, nblk = nbm * nbn;
    if ((nbn & 7) == 0) {
      const int xcd = b & 7, t = b >> 3;   // t-th block of this XCD
      bm = t % nbm;
      bn = (t / nbm) * 8 + xcd;
    } else {
      bm = b % nbm;
      bn = b / nbm;
    }
    (void)nblk;
  }
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- K loop: this wave's steps are wave, wave+4, ... -------------------------
  const int nsteps = p.K / 32;
  const int cnt = (nsteps - wave + 3) >> 2;
  const bf16_t* ap[NP];
  const bf16_t* bp[NP];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    ap[pl] = p.A[pl] + (size_t)(m0 + (lane & 15)) * p.lda + (lane >> 4) * 8 + wave * 32;
    bp[pl] = p.B[pl] + (size_t)(n0 + (lane & 15)) * p.ldb + (lane >> 4) * 8 + wave * 32;
  }
  const size_t a_frag_stride = (size_t)16 * p.lda, b_frag_stride = (size_t)16 * p.ldb;

  mfma_bf16x8 ra[DEPTH][NP][FM], rb[DEPTH][NP][FN];
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define MSD_GEMM_LOAD(STAGE, STEP)                                                        \
  {                                                                                       \
    const int koff_ = (STEP) * 128; /* 4 waves x 32 elements per round */                 \
    _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                   \
      _Pragma("unroll") for (int i = 0; i < FM; ++i)                                      \
          ra[STAGE][pl][i] = ld_frag16(ap[pl] + i * a_frag_stride + koff_);               \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                      \
          rb[STAGE][pl][j] = ld_frag16(bp[pl] + j * b_frag_stride + koff_);               \
    }                                                                                     \
  }
  // D[n][m] orientation (first operand = W^T fragment): lane holds C[m = l&15][n = (l>>4)*4 + r]
#define MSD_GEMM_COMPUTE(STAGE)                                                           \
  {                                                                                       \
    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                        \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[STAGE][0][j], ra[STAGE][0][i], acc[i][j], 0, 0, 0); \
      if (NP == 2) {                                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[STAGE][NP - 1][j], ra[STAGE][0][i], acc[i][j], 0, 0, 0); \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[STAGE][0][j], ra[STAGE][NP - 1][i], acc[i][j], 0, 0, 0); \
      }                                                                                   \
    }                                                                                     \
  }

  int it = 0;
  if (cnt >= 2 * DEPTH) {
    // steady state: the ring is filled and every stage is computed and refilled
    // UNCONDITIONALLY, so hipcc can emit exact counted s_waitcnt vmcnt(N); any
    // conditional load on the way into this loop makes its waitcnt pass assume the
    // shortest path and drain the younger prefetches at every round.
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      MSD_GEMM_LOAD(d, d)
      // the ring must be filled in stage order too: the loop-header wait is the merge
      // of this path and the back edge
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    for (; it + 2 * DEPTH <= cnt; it += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        MSD_GEMM_COMPUTE(d)
        if (PIN) __builtin_amdgcn_sched_barrier(0);  // keep stage d's MFMAs ahead of its refill
        MSD_GEMM_LOAD(d, it + d + DEPTH)
        if (PIN) __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < cnt) MSD_GEMM_LOAD(d, d)
  }
  // drain: at most 2*DEPTH - 1 steps left
  for (; it < cnt; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (it + d < cnt) {
        MSD_GEMM_COMPUTE(d)
        if (it + d + DEPTH < cnt) MSD_GEMM_LOAD(d, it + d + DEPTH)
      }
    }
  }
#undef MSD_GEMM_LOAD
#undef MSD_GEMM_COMPUTE

  // ---- cross-wave reduction (2 rounds through 2 LDS slabs) -----------------------
  auto slab_at = [&](int s, int m, int n) -> float* { return slab + ((size_t)s * BM + m) * LDS_LD + n; };
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<float4*>(slab_at(wave - 2, i * 16 + lm, j * 16 + ln)) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float4* q = reinterpret_cast<float4*>(slab_at(wave, i * 16 + lm, j * 16 + ln));
        const float4 o = *q;
        *q = make_float4(acc[i][j][0] + o.x, acc[i][j][1] + o.y, acc[i][j][2] + o.z, acc[i][j][3] + o.w);
      }
  }
  __syncthreads();

  // ---- epilogue on the summed tile: items of 8 outputs per thread ---------------
  const float* s0 = slab;
  const float* s1 = slab + (size_t)BM * LDS_LD;
  epi.template run<BM, BN, LDS_LD>(s0, s1, m0, n0, tid);
}

// ----------------------------------------------------------------------------
// Epilogues.  run<BM,BN,LD>(s0, s1, m0, n0, tid): tile value (m,n) = s0[m*LD+n] + s1[m*LD+n].
// ----------------------------------------------------------------------------
template <int LD>
__device__ __forceinline__ void tile_row8(const float* s0, const float* s1, int m, int n, float v[8]) {
  const float4 a0 = *reinterpret_cast<const float4*>(s0 + m * LD + n);
  const float4 a1 = *reinterpret_cast<const float4*>(s0 + m * LD + n + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(s1 + m * LD + n);
  const float4 b1 = *reinterpret_cast<const float4*>(s1 + m * LD + n + 4);
  v[0] = a0.x + b0.x; v[1] = a0.y + b0.y; v[2] = a0.z + b0.z; v[3] = a0.w + b0.w;
  v[4] = a1.x + b1.x; v[5] = a1.y + b1.y; v[6] = a1.z + b1.z; v[7] = a1.w + b1.w;
}

template <int NP>
__device__ __forceinline__ void store_bf16x8(bf16_t* const* planes, size_t off, const float v[8]) {
  uint32_t wh[4], wl[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bf16_t h0, l0, h1, l1;
    if (NP == 2) {
      split_bf16(v[2 * e], h0, l0);
      split_bf16(v[2 * e + 1], h1, l1);
      wl[e] = pack2(l0, l1);
    } else {
      h0 = f2bf(v[2 * e]);
      h1 = f2bf(v[2 * e + 1]);
    }
    wh[e] = pack2(h0, h1);
  }
  *reinterpret_cast<uint4*>(planes[0] + off) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
  if (NP == 2) *reinterpret_cast<uint4*>(planes[1] + off) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
}

// C (row-major bf16 planes) = acc
template <int NP>
struct EpiStoreBf16 {
  bf16_t* out[2];
  int ldc;
  template <int BM, int BN, int LD>
  __device__ void run(const float* s0, const float* s1, int m0, int n0, int tid) const {
    for (int item = tid; item < BM * BN / 8; item += 256) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
      float v[8];
      tile_row8<LD>(s0, s1, m, n, v);
      store_bf16x8<NP>(out, (size_t)(m0 + m) * ldc + n0 + n, v);
    }
  }
};

// Fused QKV (or K|V) projection: columns [0, v_start) -> row-major bf16 `qk`
// [M, ld_qk]; columns [v_start, N) -> V^T planes [seg][N - v_start][vt_ld] with the
// key axis permuted per 16 (seg = m / seg_len, key = m % seg_len).
template <int NP>
struct EpiQKV {
  bf16_t* qk[2];
  bf16_t* vt[2];
  int ld_qk, v_start, seg_len, vt_ld, vt_rows;
  template <int BM, int BN, int LD>
  __device__ void run(const float* s0, const float* s1, int m0, int n0, int tid) const {
    if (n0 < v_start) {
      for (int item = tid; item < BM * BN / 8; item += 256) {
        const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
        float v[8];
        tile_row8<LD>(s0, s1, m, n, v);
        store_bf16x8<NP>(qk, (size_t)(m0 + m) * ld_qk + n0 + n, v);
      }
    } else {
      // transposed items: (column n, 8 consecutive rows = keys).  Keys o..o+7 of a
      // 16-group land on two runs of 4 consecutive permuted positions.
      for (int item = tid; item < BM * BN / 8; item += 256) {
        const int n = item / (BM / 8), mm = (item % (BM / 8)) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = s0[(mm + e) * LD + n] + s1[(mm + e) * LD + n];
        const int mg = m0 + mm, seg = mg / seg_len, key = mg % seg_len;
        bf16_t* base[2];
        const size_t row = ((size_t)seg * vt_rows + (n0 + n - v_start)) * vt_ld + (key & ~15);
        base[0] = vt[0] + row;
        base[1] = vt[NP - 1] + row;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int kp = vt_perm16((key & 15) + 4 * hh);
          bf16_t h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (NP == 2) split_bf16(v[4 * hh + e], h[e], l[e]);
            else h[e] = f2bf(v[4 * hh + e]);
          }
          *reinterpret_cast<uint2*>(base[0] + kp) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
          if (NP == 2)
            *reinterpret_cast<uint2*>(base[1] + kp) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
        }
      }
    }
  }
};

// x[M, ldx] (fp32 residual stream) += acc
struct EpiResidual {
  float* x;
  int ldx;
  template <int BM, int BN, int LD>
  __device__ void run(const float* s0, const float* s1, int m0, int n0, int tid) const {
    for (int item = tid; item < BM * BN / 8; item += 256) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
      float v[8];
      tile_row8<LD>(s0, s1, m, n, v);
      float4* px = reinterpret_cast<float4*>(x + (size_t)(m0 + m) * ldx + n0 + n);
      float4 a = px[0], b = px[1];
      a.x += v[0]; a.y += v[1]; a.z += v[2]; a.w += v[3];
      b.x += v[4]; b.y += v[5]; b.z += v[6]; b.w += v[7];
      px[0] = a; px[1] = b;
    }
  }
};

// out fp32 [M, ldc] = acc
struct EpiStoreF32 {
  float* out;
  int ldc;
  template <int BM, int BN, int LD>
  __device__ void run(const float* s0, const float* s1, int m0, int n0, int tid) const {
    for (int item = tid; item < BM * BN / 8; item += 256) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
      float v[8];
      tile_row8<LD>(s0, s1, m, n, v);
      float4* po = reinterpret_cast<float4*>(out + (size_t)(m0 + m) * ldc + n0 + n);
      po[0] = make_float4(v[0], v[1], v[2], v[3]);
      po[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
};

// Gated GELU (layers.py:483-497 with activations ('gelu','linear')).  The packed
// weight interleaves wi_0 / wi_1 in blocks of 16 output columns: packed columns
// [32g, 32g+16) are wi_0 columns [16g, 16g+16), [32g+16, 32g+32) the matching wi_1.
//   g_out[m][n0/2 + j] = gelu(tile[m][pc(j)]) * tile[m][pc(j) + 16]
template <int NP>
struct EpiGeglu {
  bf16_t* out[2];
  int ldc;  // = F
  template <int BM, int BN, int LD>
  __device__ void run(const float* s0, const float* s1, int m0, int n0, int tid) const {
    static_assert(BN % 32 == 0, "gated epilogue needs whole wi_0/wi_1 groups");
    constexpr int OUT_N = BN / 2;  // output columns per tile
    for (int item = tid; item < BM * OUT_N / 8; item += 256) {
      const int m = item / (OUT_N / 8), j = (item % (OUT_N / 8)) * 8;  // 8 output cols j..j+7
      const int pc = (j / 16) * 32 + (j % 16);
      float a[8], b[8], v[8];
      tile_row8<LD>(s0, s1, m, pc, a);
      tile_row8<LD>(s0, s1, m, pc + 16, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(a[e]) * b[e];
      store_bf16x8<NP>(out, (size_t)(m0 + m) * ldc + n0 / 2 + j, v);
    }
  }
};

template <int NP, int BM, int BN, int DEPTH, bool PIN, class Epi>
inline hipError_t launch_gemm_bf16(const GemmParams& p, const Epi& epi, hipStream_t stream) {
  constexpr int smem = 2 * BM * (BN + kSlabPad) * 4;
  static_assert(smem <= 64 * 1024, "tile needs the large-LDS attribute");
  auto kern = gemm_bf16_kernel<NP, BM, BN, DEPTH, PIN, Epi>;
  const int grid = (p.M / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
