// Exact-fp32 GEMM on v_mfma_f32_16x16x4_f32 for the places the reference keeps in
// float32 on purpose or where K is tiny:
//   * spec_out_dense 768->128 "float32 for stability"           (network.py:452-456)
//   * continuous_inputs_projection / input_proj 128->D          (network.py:420-425, 321-325)
//   * load-time tables: time-embedding MLP and the 2*Ld FiLM projections for all
//     N steps at once                                            (network.py:377-392; layers.py:660-663)
// C[M,N] = A[M,K] . B[K,N], all row-major fp32.  The f32 MFMA is bit-for-bit a
// k-ordered fmaf chain (MI355X guide), i.e. plain fp32 arithmetic.
// Block = 4 waves (2x2) on a 64x64 tile, BK = 16, LDS-staged; rows >= M are
// guarded (M need not be a tile multiple); N % 64 == 0 and K % 16 == 0.
#pragma once
#include "common.h"

namespace msd {

struct GemmF32Params {
  const float* A;
  const float* B;
  int lda, ldb;
  int M, N, K;
};

struct EpiF32Store {
  float* out;
  int ldc;
  __device__ void operator()(int m, int n, float v) const { out[(size_t)m * ldc + n] = v; }
};
struct EpiF32Swish {  // nn.swish (network.py:385,391)
  float* out;
  int ldc;
  __device__ void operator()(int m, int n, float v) const {
    out[(size_t)m * ldc + n] = v / (1.0f + expf(-v));
  }
};
// out[m][packed(n)] = v, packed(n) = (n/16)*32 + which*16 + n%16: the wi_0/wi_1 column
// interleave of the gated-MLP weight (gemm_h16.h EpiGeglu), for the folded FiLM-bias table
struct EpiF32StoreGated {
  float* out;
  int ldc, which;
  __device__ void operator()(int m, int n, float v) const {
    out[(size_t)m * ldc + (n / 16) * 32 + which * 16 + (n % 16)] = v;
  }
};
// decoder input: x[pass][m][:] = z[m] . W_in + pos[m % T]   (network.py:420-427);
// the same rows feed the conditional and the unconditional pass.
struct EpiF32InProj {
  float* x;
  const float* pos;
  int ldx, T, pass_stride_rows, passes;
  __device__ void operator()(int m, int n, float v) const {
    const float r = v + pos[(size_t)(m % T) * ldx + n];
    for (int ps = 0; ps < passes; ++ps) x[((size_t)ps * pass_stride_rows + m) * ldx + n] = r;
  }
};
// context encoder input: x[m] = ctx_scaled[m] . W + pos[pos_idx[m]]   (network.py:321-339)
struct EpiF32AddRows {
  float* out;
  const float* table;
  const int* row_idx;
  int ldc;
  __device__ void operator()(int m, int n, float v) const {
    out[(size_t)m * ldc + n] = v + table[(size_t)row_idx[m] * ldc + n];
  }
};

template <class Epi>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmF32Params p, Epi epi) {
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ __attribute__((aligned(16))) float As[BM][BK + 1];
  __shared__ __attribute__((aligned(16))) float Bs[BK][BN + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = p.N / BN;
  const int m0 = (blockIdx.x / nbn) * BM, n0 = (blockIdx.x % nbn) * BN;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.K; k0 += BK) {
    {  // A tile: 64 x 16 floats = 256 float4
      const int row = tid >> 2, c4 = (tid & 3) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + row < p.M) v = *reinterpret_cast<const float4*>(p.A + (size_t)(m0 + row) * p.lda + k0 + c4);
      As[row][c4] = v.x; As[row][c4 + 1] = v.y; As[row][c4 + 2] = v.z; As[row][c4 + 3] = v.w;
    }
    {  // B tile: 16 x 64 floats = 256 float4
      const int row = tid >> 4, c4 = (tid & 15) * 4;
      const float4 v = *reinterpret_cast<const float4*>(p.B + (size_t)(k0 + row) * p.ldb + n0 + c4);
      *reinterpret_cast<float4*>(&Bs[row][c4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[wm * 32 + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk + (lane >> 4)][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // C layout: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        const int n = n0 + wn * 32 + j * 16 + (lane & 15);
        if (m < p.M) epi(m, n, acc[i][j][r]);
      }
}

// ----------------------------------------------------------------------------
// Final projection of the decoder in exact fp32 (network.py:445-456), folded with the
// decoder_norm:  eps[m][n] = rstd[m] * sum_k x[m][k] * (gamma[k] W[k][n]).
// M = 512, N = 128, K = 768 is 0.1 GFLOP: tiny, so the 4 waves of a block split K
// (wave w takes the 16-wide K groups w, w+4, ...) on one 32 x 32 tile, 64 blocks.
// v_mfma_f32_16x16x4_f32 operand k = lane>>4; a lane loads float4 x[row][k0 + 4g .. +3]
// (g = lane>>4) and uses component c in MFMA c, so MFMA c contracts k = k0 + 4g + c; the
// B operand of MFMA c is the matching row Wg[k0 + 4g + c][n].
// ----------------------------------------------------------------------------
struct FinalProjParams {
  const float* x;      // [M, K] fp32 residual stream
  const float* wg;     // [K, N] = diag(gamma) . W
  const float* ssq;    // [M][tiles] partial sums of squares of x
  float* out;          // [M, N]
  int M, N, K, tiles;
  float inv_d;
};

constexpr int kFinalProjWaves = 8;   // K is split over the waves of a block (latency-bound: M*N is tiny)

// RT: 16-row MFMA tiles per block (block tile = 16*RT rows x 32 columns)
template <int RT>
__global__ void __launch_bounds__(64 * kFinalProjWaves) final_proj_f32_kernel(FinalProjParams p) {
  warm_kernargs<kernarg_lines<FinalProjParams>()>();
  constexpr int BMR = 16 * RT;
  __shared__ __attribute__((aligned(16))) float red[kFinalProjWaves][BMR][33];
  __shared__ float rstd[BMR];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.y * BMR, n0 = blockIdx.x * 32;   // grid (N / 32, M / BMR): no integer division at the entry
  const int g = lane >> 4, r = lane & 15;
  // row statistics once per row: the 64 lanes of wave 0 take a quarter of a row's partial sums each (lane = quarter x
  // row), all loads in flight together, two shuffles.  Round 3's form -- one thread per row, a run-time loop -- was 24
  // DEPENDENT global loads (vmcnt(0) behind each) that the whole block then waited for at its barrier: about 5 of this
  // kernel's 10.6 us.  (BMR <= 16 rows here; tiles <= 32.)
  float ss = 0.f;
  float sv[8];   // issued here, reduced BEHIND the K loop: wave 0's operand loads must not queue behind a wait for these
  static_assert(BMR <= 16, "one wave covers the rows of the tile in quarters");
  if (wave == 0) {
    const int row = lane & 15, part = lane >> 4;
    const int per = (p.tiles + 3) >> 2;                        // partial sums per quarter (<= 8)
    const float* q = p.ssq + (size_t)(m0 + (row < BMR ? row : 0)) * p.tiles;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = part * per + i;
      sv[i] = q[(i < per && t < p.tiles) ? t : 0];
    }
  }
  f32x4 acc[RT][2];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // one MFMA K-step is 4 wide: lane (g, r) holds A[row r][k0 + 4g .. +3] and B[k0 + 4g + c][col r];
  // a wave's slice is 16 wide (4 lane groups x 4), slices go round-robin over the waves; the
  // (HBM-cold) operand loads of U slices of a wave are issued together.
  const float* xa = p.x + (size_t)(m0 + r) * p.K + 4 * g;
  const float* wb = p.wg + (size_t)(4 * g) * p.N + n0 + r;
  constexpr int U = 6;   // K slices in flight per wave (768 / (16 * 8) = 6: one round at D = 768)
  for (int kb = wave * 16; kb < p.K; kb += 16 * kFinalProjWaves * U) {
    float4 a[U][RT];
    float b[U][2][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {          // all operand loads first (tail slices re-read slice kb, unused)
      const int k0 = kb + u * 16 * kFinalProjWaves;
      const int kk = k0 < p.K ? k0 : kb;
#pragma unroll
      for (int i = 0; i < RT; ++i) a[u][i] = *reinterpret_cast<const float4*>(xa + (size_t)(i * 16) * p.K + kk);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) b[u][j][c] = wb[(size_t)(kk + c) * p.N + j * 16];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kb + u * 16 * kFinalProjWaves < p.K) {   // wave-uniform
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            const float av = c == 0 ? a[u][i].x : (c == 1 ? a[u][i].y : (c == 2 ? a[u][i].z : a[u][i].w));
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][j][c], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  }
  if (wave == 0) {
    const int part = lane >> 4, per = (p.tiles + 3) >> 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += (i < per && part * per + i < p.tiles) ? sv[i] : 0.f;
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
  }
  if (threadIdx.x < BMR) rstd[threadIdx.x] = 1.0f / sqrtf(ss * p.inv_d + 1e-6f);
  // C layout: col = lane & 15, row = (lane >> 4) * 4 + e
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][i * 16 + g * 4 + e][j * 16 + r] = acc[i][j][e];
  __syncthreads();
  for (int item = threadIdx.x; item < BMR * 32; item += 64 * kFinalProjWaves) {
    const int m = item >> 5, n = item & 31;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kFinalProjWaves; ++w) v += red[w][m][n];
    p.out[(size_t)(m0 + m) * p.N + n0 + n] = v * rstd[m];
  }
}

// wg[k][n] = gamma[k] * w[k][n]
__global__ void scale_rows_kernel(const float* w, const float* gamma, float* wg, int K, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < K * N) wg[i] = gamma[i / N] * w[i];
}

template <class Epi>
inline hipError_t launch_gemm_f32(const GemmF32Params& p, const Epi& epi, hipStream_t stream) {
  const int grid = ((p.M + 63) / 64) * (p.N / 64);
  hipLaunchKernelGGL(gemm_f32_kernel<Epi>, dim3(grid), dim3(256), 0, stream, p, epi);
  return hipGetLastError();
}

}  // namespace msd
