// The product GEMMs with their REAL epilogues (row statistics, step-indexed bias, V^T transposed
// stores, gated GELU, residual + folded-norm outputs), cold weights: what each launch of the DDPM
// step costs in isolation.  Compare with profiles/r01_bench_kernel_stats.csv (the same kernels in situ).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_epi_0 gemm_epi.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "exp/src_r04/gemm_h16.h"   // round-4 sources: the ablation switches live there, not in the product
using namespace msd;

template <class T> T* dmalloc(size_t n, int fill = 0) { T* p; (void)hipMalloc(&p, n * sizeof(T)); (void)hipMemset(p, fill, n * sizeof(T)); return p; }

template <int BM, int BN, int NS, class Epi>
double timeit(GemmParams p, Epi epi, h16_t* b0, h16_t* b1, size_t bstride, int copies, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto go = [&](int it) { p.B[0] = b0 + (size_t)(it % copies) * bstride; p.B[1] = b1 + (size_t)(it % copies) * bstride;
                          (void)launch_gemm_h16_dma<2, BM, BN, NS>(p, epi, 0); };
  for (int i = 0; i < 5; ++i) go(i);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < iters; ++i) go(i + 5); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / iters;
}

int main() {
  const int M = 512, D = 768, J = 768, F = 2048, T = 256, COPIES = 24, tiles = D / 32;
  int* step = dmalloc<int>(2);
  float* ssq = dmalloc<float>((size_t)M * tiles, 0x3c);
  float* bias = dmalloc<float>((size_t)8192);
  float* g = dmalloc<float>((size_t)D, 0x3c);
  float* x = dmalloc<float>((size_t)M * D);
  RowScale rs; rs.ssq = ssq; rs.tiles = tiles; rs.inv_d = 1.0f / D; rs.bias = bias; rs.bias_step_stride = 0; rs.step_ptr = step;
  auto planes = [&](size_t n, h16_t** p) { p[0] = dmalloc<h16_t>(n, 0x3c); p[1] = dmalloc<h16_t>(n, 0x3b); };
  h16_t *y[2], *gact[2], *qk[2], *vt[2], *gout[2], *wq[2], *wi[2], *wo[2];
  planes((size_t)M * D, y); planes((size_t)M * F, gact); planes((size_t)M * 2 * J + 4096, qk); planes((size_t)M * J + 4096, vt); planes((size_t)M * 2 * F + 4096, gout);
  planes((size_t)COPIES * 3 * J * D, wq); planes((size_t)COPIES * 2 * F * D, wi); planes((size_t)COPIES * D * F, wo);
  GemmParams p; p.A[0] = y[0]; p.A[1] = y[1]; p.lda = D; p.ldb = D; p.M = M; p.K = D;
  {
    EpiQKV<2> e; e.qk[0] = qk[0]; e.qk[1] = qk[1]; e.vt[0] = vt[0]; e.vt[1] = vt[1]; e.ld_qk = 2 * J; e.v_start = 2 * J; e.seg_len = T; e.vt_ld = T; e.vt_rows = J; e.rsc = rs;
    p.N = 3 * J;
    printf("qkv     64x96  EpiQKV        : %.1f us\n", timeit<64, 96, 3>(p, e, wq[0], wq[1], (size_t)3 * J * D, COPIES, 96));
    EpiStoreH16<2> s; s.out[0] = gout[0]; s.out[1] = gout[1]; s.ldc = 3 * J;
    printf("qkv     64x96  plain store   : %.1f us\n", timeit<64, 96, 3>(p, s, wq[0], wq[1], (size_t)3 * J * D, COPIES, 96));
    s.rsc = rs;
    printf("qkv     64x96  store+rowscale: %.1f us\n", timeit<64, 96, 3>(p, s, wq[0], wq[1], (size_t)3 * J * D, COPIES, 96));
    s.rsc.bias = nullptr;
    printf("qkv     64x96  store+rstd only (no bias row, no step load): %.1f us\n", timeit<64, 96, 3>(p, s, wq[0], wq[1], (size_t)3 * J * D, COPIES, 96));
  }
  {
    EpiGeglu<2> e; e.out[0] = gact[0]; e.out[1] = gact[1]; e.ldc = F; e.rsc = rs;
    p.N = 2 * F;
    printf("mlp_in  64x128 EpiGeglu      : %.1f us\n", timeit<64, 128, 3>(p, e, wi[0], wi[1], (size_t)2 * F * D, COPIES, 96));
    EpiStoreH16<2> s; s.out[0] = gout[0]; s.out[1] = gout[1]; s.ldc = 2 * F;   // (gout is large enough: M * 2F / 2 planes.. use half)
    p.N = F;
    printf("mlp_in/2 64x128 plain store  : %.1f us (N = F only)\n", timeit<64, 128, 3>(p, s, wi[0], wi[1], (size_t)2 * F * D, COPIES, 96));
  }
  {
    EpiResidualNorm<2> e; e.x = x; e.ldx = D; e.y[0] = y[0]; e.y[1] = y[1]; e.ssq = ssq; e.tiles = tiles; e.g_lo = g; e.g_lo_stride = 0; e.g_hi = g; e.g_hi_stride = 0; e.split_row = 0; e.step_ptr = step;
    GemmParams q = p; q.A[0] = gact[0]; q.A[1] = gact[1]; q.lda = F; q.ldb = F; q.N = D; q.K = F;
    printf("mlp_out 64x32  EpiResidualNorm: %.1f us\n", timeit<64, 32, 4>(q, e, wo[0], wo[1], (size_t)D * F, COPIES, 96));
    EpiResidual r{x, D};
    printf("mlp_out 64x32  EpiResidual    : %.1f us\n", timeit<64, 32, 4>(q, r, wo[0], wo[1], (size_t)D * F, COPIES, 96));
    GemmParams a = p; a.A[0] = qk[0]; a.A[1] = qk[1]; a.lda = J; a.ldb = J; a.N = D; a.K = J;
    printf("attn_out 32x32 EpiResidualNorm: %.1f us\n", timeit<32, 32, 4>(a, e, wq[0], wq[1], (size_t)D * J, COPIES, 96));
  }
  return 0;
}
