// Standalone timing + ablation harness for csrc/gemm_h16.h on the DDPM step's GEMM shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMSD_ABL=n] -o gemm_bench gemm_bench.hip
// MSD_ABL: 0 full kernel; 1 no global loads in the loop; 2 no MFMA; 3 no ds_read; 4 no ds_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_h16_regstaged.h"
using namespace msd;

template <int NP, int BM, int BN, int R, bool DMA = false>
double run(int M, int N, int K, int iters, bool resid) {
  h16_t *a[2], *b[2]; float* c; h16_t* o[2];
  for (int i = 0; i < 2; ++i) { hipMalloc(&a[i], (size_t)M * K * 2); hipMalloc(&b[i], (size_t)N * K * 2); hipMalloc(&o[i], (size_t)M * N * 2);
    hipMemset(a[i], 0x3c, (size_t)M * K * 2); hipMemset(b[i], 0x3b, (size_t)N * K * 2); }
  hipMalloc(&c, (size_t)M * N * 4); hipMemset(c, 0, (size_t)M * N * 4);
  GemmParams p; for (int i = 0; i < 2; ++i) { p.A[i] = a[i]; p.B[i] = b[i]; } p.lda = K; p.ldb = K; p.M = M; p.N = N; p.K = K;
  EpiResidual er{c, N}; EpiStoreH16<NP> es; es.out[0] = o[0]; es.out[1] = o[1]; es.ldc = N;

  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&]() {
    if constexpr (DMA) { if (resid) launch_gemm_h16_dma<NP, BM, BN, R>(p, er, 0); else launch_gemm_h16_dma<NP, BM, BN, R>(p, es, 0); }
    else { if (resid) launch_gemm_h16<NP, BM, BN, R>(p, er, 0); else launch_gemm_h16<NP, BM, BN, R>(p, es, 0); }
  };
  // a second, unrelated kernel between launches so that weights are not L2-hot is NOT done here:
  // this measures the back-to-back (L2/MALL-warm) cost; the step re-reads each weight once per 2 ms.
  for (int i = 0; i < 5; ++i) go();
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < iters; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  for (int i = 0; i < 2; ++i) { hipFree(a[i]); hipFree(b[i]); hipFree(o[i]); } hipFree(c);
  return ms * 1e3 / iters;
}

// same GEMM, but every launch reads a DIFFERENT copy of the weights (COPIES * bytes > L2 + MALL):
// what a step sees, where each weight matrix is touched once per ~1.4 ms.
template <int NP, int BM, int BN, int NS>
double run_cold(int M, int N, int K, int iters, bool resid) {
  const int COPIES = 48;
  h16_t *a[2], *b[2]; float* c; h16_t* o[2];
  for (int i = 0; i < 2; ++i) { hipMalloc(&a[i], (size_t)M * K * 2); hipMalloc(&b[i], (size_t)COPIES * N * K * 2); hipMalloc(&o[i], (size_t)M * N * 2);
    hipMemset(a[i], 0x3c, (size_t)M * K * 2); hipMemset(b[i], 0x3b, (size_t)COPIES * N * K * 2); }
  hipMalloc(&c, (size_t)M * N * 4); hipMemset(c, 0, (size_t)M * N * 4);
  GemmParams p; for (int i = 0; i < 2; ++i) p.A[i] = a[i]; p.lda = K; p.ldb = K; p.M = M; p.N = N; p.K = K;
  EpiResidual er{c, N}; EpiStoreH16<NP> es; es.out[0] = o[0]; es.out[1] = o[1]; es.ldc = N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&](int it) {
    for (int i = 0; i < 2; ++i) p.B[i] = b[i] + (size_t)(it % COPIES) * N * K;
    if (resid) launch_gemm_h16_dma<NP, BM, BN, NS>(p, er, 0); else launch_gemm_h16_dma<NP, BM, BN, NS>(p, es, 0);
  };
  for (int i = 0; i < 5; ++i) go(i);
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < iters; ++i) go(i + 5); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  for (int i = 0; i < 2; ++i) { hipFree(a[i]); hipFree(b[i]); hipFree(o[i]); } hipFree(c);
  return ms * 1e3 / iters;
}

int main() {
  struct S { const char* name; int M, N, K; bool resid; } shapes[] = {
      {"qkv      512x2304x768 ", 512, 2304, 768, false}, {"attn_out 512x768x768  ", 512, 768, 768, true},
      {"cross_q  256x768x768  ", 256, 768, 768, false},  {"mlp_in   512x4096x768 ", 512, 4096, 768, false},
      {"mlp_out  512x768x2048 ", 512, 768, 2048, true}};
  printf("MSD_ABL=%d\n", MSD_ABL);
  for (auto& s : shapes) {
    const double gf = 2.0 * s.M * s.N * s.K / 1e9;
    double t1 = run<2, 64, 64, 2>(s.M, s.N, s.K, 200, s.resid);
    double t2 = run<2, 128, 64, 2>(s.M, s.N, s.K, 200, s.resid);
    double t3 = run<1, 64, 64, 2>(s.M, s.N, s.K, 200, s.resid);
    double t4 = run<1, 128, 64, 2>(s.M, s.N, s.K, 200, s.resid);
    printf("%s bf16x3: 64x64 %6.1f us (%5.1f TF) | 128x64 %6.1f us (%5.1f TF) || bf16: 64x64 %6.1f us | 128x64 %6.1f us\n", s.name, t1,
           gf / t1 * 1e-3, t2, gf / t2 * 1e-3, t3, t4);
    if (MSD_ABL == 0) {
      double d1 = run<2, 64, 64, 3, true>(s.M, s.N, s.K, 200, s.resid);
      double d2 = run<2, 64, 64, 2, true>(s.M, s.N, s.K, 200, s.resid);
      double d3 = run<2, 128, 64, 3, true>(s.M, s.N, s.K, 200, s.resid);
      double d4 = run<1, 64, 64, 4, true>(s.M, s.N, s.K, 200, s.resid);
      double d5 = run<1, 128, 128, 3, true>(s.M, s.N, s.K, 200, s.resid);
      printf("   DMA     bf16x3: 64x64 NS3 %6.1f us (%5.1f TF) NS2 %6.1f | 128x64 NS3 %6.1f us || bf16: 64x64 NS4 %6.1f us | 128x128 NS3 %6.1f\n",
             d1, gf / d1 * 1e-3, d2, d3, d4, d5);
      double e1 = run<2, 32, 64, 3, true>(s.M, s.N, s.K, 200, s.resid);
      double e2 = run<2, 32, 64, 4, true>(s.M, s.N, s.K, 200, s.resid);
      double e3 = run<2, 64, 32, 3, true>(s.M, s.N, s.K, 200, s.resid);
      double e4 = run<2, 32, 32, 4, true>(s.M, s.N, s.K, 200, s.resid);
      printf("   DMA     bf16x3: 32x64 NS3 %6.1f NS4 %6.1f | 64x32 NS3 %6.1f | 32x32 NS4 %6.1f\n", e1, e2, e3, e4);
      if (s.N % 96 == 0) printf("   DMA     bf16x3: 64x96 NS2 %6.1f | 128x96 NS2 %6.1f | cold 64x96 NS2 %6.1f\n", run<2, 64, 96, 2, true>(s.M, s.N, s.K, 200, s.resid),
             run<2, 128, 96, 2, true>(s.M, s.N, s.K, 200, s.resid), run_cold<2, 64, 96, 2>(s.M, s.N, s.K, 192, s.resid));
      if (s.N % 96 == 0) printf("   COLD 64x96 NS3 %6.1f | 32x96 NS3 %6.1f NS4 %6.1f\n", run_cold<2, 64, 96, 3>(s.M, s.N, s.K, 192, s.resid),
             run_cold<2, 32, 96, 3>(s.M, s.N, s.K, 192, s.resid), run_cold<2, 32, 96, 4>(s.M, s.N, s.K, 192, s.resid));
      if (s.N % 128 == 0) printf("   COLD 64x128 NS2 %6.1f NS3 %6.1f | 32x128 NS3 %6.1f NS4 %6.1f\n", run_cold<2, 64, 128, 2>(s.M, s.N, s.K, 192, s.resid), run_cold<2, 64, 128, 3>(s.M, s.N, s.K, 192, s.resid),
             run_cold<2, 32, 128, 3>(s.M, s.N, s.K, 192, s.resid), run_cold<2, 32, 128, 4>(s.M, s.N, s.K, 192, s.resid));
      printf("   COLD 64x64 NS3 %6.1f NS4 %6.1f | 32x64 NS4 %6.1f | 64x32 NS4 %6.1f | 32x32 NS6 %6.1f\n", run_cold<2, 64, 64, 3>(s.M, s.N, s.K, 192, s.resid), run_cold<2, 64, 64, 4>(s.M, s.N, s.K, 192, s.resid),
             run_cold<2, 32, 64, 4>(s.M, s.N, s.K, 192, s.resid), run_cold<2, 64, 32, 4>(s.M, s.N, s.K, 192, s.resid), run_cold<2, 32, 32, 6>(s.M, s.N, s.K, 192, s.resid));
      printf("   COLD weights (48 copies): 64x64 NS2 %6.1f | 32x32 NS4 %6.1f\n", run_cold<2, 64, 64, 2>(s.M, s.N, s.K, 192, s.resid),
             run_cold<2, 32, 32, 4>(s.M, s.N, s.K, 192, s.resid));
    }
  }
  return 0;
}
