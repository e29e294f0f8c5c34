// bf16 / bf16x3 MFMA GEMM for the small-M (M = 256..2304) transformer projections.
//
//   C[M,N] (+)= A[M,K] . W[K,N]        A: bf16 planes [M,K] row-major
//                                      W: packed at load time as W^T planes [N,K]
// NP = 1: plain bf16 operands.  NP = 2 ("bf16x3"): A = Ah+Al, W = Wh+Wl and the
// product is Ah.Wh + Ah.Wl + Al.Wh (three v_mfma_f32_16x16x32_bf16 per tile),
// fp32 accumulate: fp32-class accuracy at the bf16 MFMA rate / 3.
//
// Block = 4 waves (2x2), tile BM x BN x BK, both operands staged through LDS by
// registers (16 B/lane global loads -> XOR-swizzled ds_write_b128), double
// buffered with one barrier per K-step; fragments by ds_read_b128.  Both
// operands are K-contiguous, so A and B fragments are the same 16-byte reads.
// The epilogue functor owns the output orientation:
//   swapped (D = W^T-frag x A-frag): lane holds C[m][n..n+3] -> row-major stores
//   direct  (D = A-frag x W^T-frag): lane holds C[m..m+3][n] -> transposed stores
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

template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  // byte offset of 16-byte chunk `chunk` of `row` in a [rows][BK] bf16 tile
  constexpr int CH = BK / 8;
  const int swz = (CH == 8) ? (row & 7) : ((row >> 1) & 3);
  return row * (BK * 2) + (((chunk ^ swz) & (CH - 1)) << 4);
}

template <int NP, int BM, int BN, int BK, class Epi>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(GemmParams p, Epi epi) {
  constexpr int CH = BK / 8;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  constexpr int A_IT = BM * CH / 256, B_IT = BN * CH / 256;
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for 256 threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = p.N / BN;
  const int m0 = (blockIdx.x / nbn) * BM, n0 = (blockIdx.x % nbn) * BN;
  const bool swapped = epi.swapped(n0);

  uint4 ra[NP][A_IT], rb[NP][B_IT];
  auto load_global = [&](int k0) {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + i * 256, row = idx / CH, c = idx % CH;
        ra[pl][i] = *reinterpret_cast<const uint4*>(p.A[pl] + (size_t)(m0 + row) * p.lda + k0 + c * 8);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + i * 256, row = idx / CH, c = idx % CH;
        rb[pl][i] = *reinterpret_cast<const uint4*>(p.B[pl] + (size_t)(n0 + row) * p.ldb + k0 + c * 8);
      }
    }
  };
  auto store_lds = [&](int stage) {
    char* base = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + i * 256, row = idx / CH, c = idx % CH;
        *reinterpret_cast<uint4*>(base + pl * A_BYTES + lds_off<BK>(row, c)) = ra[pl][i];
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + i * 256, row = idx / CH, c = idx % CH;
        *reinterpret_cast<uint4*>(base + NP * A_BYTES + pl * B_BYTES + lds_off<BK>(row, c)) = rb[pl][i];
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const char* base = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      mfma_bf16x8 a[NP][FM], b[NP][FN];
      const int c = kk * 4 + (lane >> 4);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * WM + i * 16 + (lane & 15);
          a[pl][i] = as_frag(*reinterpret_cast<const uint4*>(base + pl * A_BYTES + lds_off<BK>(row, c)));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn * WN + j * 16 + (lane & 15);
          b[pl][j] = as_frag(*reinterpret_cast<const uint4*>(base + NP * A_BYTES + pl * B_BYTES + lds_off<BK>(row, c)));
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (swapped) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
            if (NP == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[NP - 1][j], a[0][i], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][j], a[NP - 1][i], acc[i][j], 0, 0, 0);
            }
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
            if (NP == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[NP - 1][j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NP - 1][i], b[0][j], acc[i][j], 0, 0, 0);
            }
          }
        }
    }
  };

  const int nk = p.K / BK;
  load_global(0);
  store_lds(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_global((kt + 1) * BK);
    compute(kt & 1);
    if (kt + 1 < nk) store_lds((kt + 1) & 1);
    __syncthreads();
  }

  // epilogue: element (i, j, r) of this lane is
  //   swapped: C[mw + i*16 + (lane&15)][nw + j*16 + (lane>>4)*4 + r]
  //   direct : C[mw + i*16 + (lane>>4)*4 + r][nw + j*16 + (lane&15)]
  const int mw = m0 + wm * WM, nw = n0 + wn * WN;
  epi.template store<FM, FN>(acc, mw, nw, lane, swapped);
}

// ----------------------------------------------------------------------------
// Epilogues
// ----------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void store_bf16x4(bf16_t* const* planes, size_t off, const f32x4& v) {
  bf16_t h[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (NP == 2) split_bf16(v[r], h[r], l[r]);
    else h[r] = f2bf(v[r]);
  }
  *reinterpret_cast<uint2*>(planes[0] + off) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
  if (NP == 2)
    *reinterpret_cast<uint2*>(planes[1] + off) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}

// key position permutation inside each group of 16 keys of a V^T row, so that the
// attention kernel's P^T fragment (32x32 MFMA C layout) lines up with 16-byte
// V^T loads: offset o -> 8*((o>>2)&1) + (o&3) + 4*(o>>3)   (see attention.h)
__device__ __forceinline__ int vt_perm16(int o) { return 8 * ((o >> 2) & 1) + (o & 3) + 4 * (o >> 3); }

// C (row-major bf16 planes) = acc
template <int NP>
struct EpiStoreBf16 {
  bf16_t* out[2];
  int ldc;
  __device__ bool swapped(int) const { return true; }
  template <int FM, int FN>
  __device__ void store(f32x4 (&acc)[FM][FN], int mw, int nw, int lane, bool) const {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int m = mw + i * 16 + (lane & 15), n = nw + j * 16 + (lane >> 4) * 4;
        store_bf16x4<NP>(out, (size_t)m * ldc + n, acc[i][j]);
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
  __device__ bool swapped(int n0) const { return n0 < v_start; }
  template <int FM, int FN>
  __device__ void store(f32x4 (&acc)[FM][FN], int mw, int nw, int lane, bool sw) const {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (sw) {
          const int m = mw + i * 16 + (lane & 15), n = nw + j * 16 + (lane >> 4) * 4;
          store_bf16x4<NP>(qk, (size_t)m * ld_qk + n, acc[i][j]);
        } else {
          const int m = mw + i * 16 + (lane >> 4) * 4, n = nw + j * 16 + (lane & 15) - v_start;
          const int seg = m / seg_len, key = m % seg_len;
          const int kp = (key & ~15) + vt_perm16(key & 15);  // 4 consecutive keys stay consecutive
          store_bf16x4<NP>(vt, ((size_t)seg * vt_rows + n) * vt_ld + kp, acc[i][j]);
        }
      }
  }
};

// x[M, ldx] (fp32 residual stream) += acc
struct EpiResidual {
  float* x;
  int ldx;
  __device__ bool swapped(int) const { return true; }
  template <int FM, int FN>
  __device__ void store(f32x4 (&acc)[FM][FN], int mw, int nw, int lane, bool) const {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int m = mw + i * 16 + (lane & 15), n = nw + j * 16 + (lane >> 4) * 4;
        float4* px = reinterpret_cast<float4*>(x + (size_t)m * ldx + n);
        float4 v = *px;
        v.x += acc[i][j][0]; v.y += acc[i][j][1]; v.z += acc[i][j][2]; v.w += acc[i][j][3];
        *px = v;
      }
  }
};

// out fp32 [M, ldc] = acc
struct EpiStoreF32 {
  float* out;
  int ldc;
  __device__ bool swapped(int) const { return true; }
  template <int FM, int FN>
  __device__ void store(f32x4 (&acc)[FM][FN], int mw, int nw, int lane, bool) const {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int m = mw + i * 16 + (lane & 15), n = nw + j * 16 + (lane >> 4) * 4;
        *reinterpret_cast<float4*>(out + (size_t)m * ldc + n) =
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
  }
};

// Gated GELU (layers.py:483-497 with activations ('gelu','linear')).  The packed
// weight interleaves wi_0 / wi_1 in blocks of 16 output columns, so fragment
// pair (2j, 2j+1) holds gelu-input and linear-input of the same 16 columns:
//   g[m][(nw/2) + j*16 + ..] = gelu(acc[i][2j]) * acc[i][2j+1]
template <int NP>
struct EpiGeglu {
  bf16_t* out[2];
  int ldc;  // = F
  __device__ bool swapped(int) const { return true; }
  template <int FM, int FN>
  __device__ void store(f32x4 (&acc)[FM][FN], int mw, int nw, int lane, bool) const {
    static_assert(FN % 2 == 0, "gated epilogue needs fragment pairs");
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN / 2; ++j) {
        const int m = mw + i * 16 + (lane & 15), n = nw / 2 + j * 16 + (lane >> 4) * 4;
        f32x4 g;
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = gelu_tanh(acc[i][2 * j][r]) * acc[i][2 * j + 1][r];
        store_bf16x4<NP>(out, (size_t)m * ldc + n, g);
      }
  }
};

template <int NP, int BM, int BN, int BK, class Epi>
inline hipError_t launch_gemm_bf16(const GemmParams& p, const Epi& epi, hipStream_t stream) {
  constexpr int smem = 2 * NP * (BM + BN) * BK * 2;
  auto kern = gemm_bf16_kernel<NP, BM, BN, BK, Epi>;
  static_assert(smem <= 64 * 1024, "tile needs the large-LDS attribute");
  const int grid = (p.M / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
