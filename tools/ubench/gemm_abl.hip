// Main-loop ablation of the product GEMM: build with -DMSD_DMA_ABL=0..3
//   0 full | 1 no MFMA | 2 no LDS fragment reads | 3 no DMA inside the loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include "exp/src_r04/gemm_h16.h"   // round-4 sources: the ablation switches live there, not in the product
using namespace msd;
template <int NP, int BM, int BN, int NS>
double run(int M, int N, int K, int iters) {
  h16_t *a[2], *b[2]; h16_t* o[2];
  for (int i = 0; i < 2; ++i) { hipMalloc(&a[i], (size_t)M * K * 2); hipMalloc(&b[i], (size_t)N * K * 2); hipMalloc(&o[i], (size_t)M * N * 2);
    hipMemset(a[i], 0x3c, (size_t)M * K * 2); hipMemset(b[i], 0x3b, (size_t)N * K * 2); }
  GemmParams p; for (int i = 0; i < 2; ++i) { p.A[i] = a[i]; p.B[i] = b[i]; } p.lda = K; p.ldb = K; p.M = M; p.N = N; p.K = K;
  EpiStoreH16<NP> es; es.out[0] = o[0]; es.out[1] = o[1]; es.ldc = N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch_gemm_h16_dma<NP, BM, BN, NS>(p, es, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < iters; ++i) launch_gemm_h16_dma<NP, BM, BN, NS>(p, es, 0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  for (int i = 0; i < 2; ++i) { hipFree(a[i]); hipFree(b[i]); hipFree(o[i]); }
  return ms * 1e3 / iters;
}
int main() {
  printf("ABL=%d  (warm weights, M=512; us at K=768 / K=3072; per-tile = diff/36)\n", MSD_DMA_ABL);
  { double t1 = run<2, 64, 96, 3>(512, 2304, 768, 100), t2 = run<2, 64, 96, 3>(512, 2304, 3072, 100); printf("64x96 NS3  N=2304: %6.1f %6.1f  per-tile %.3f us\n", t1, t2, (t2 - t1) / 36); }
  { double t1 = run<2, 64, 128, 3>(512, 4096, 768, 100), t2 = run<2, 64, 128, 3>(512, 4096, 3072, 100); printf("64x128 NS3 N=4096: %6.1f %6.1f  per-tile %.3f us\n", t1, t2, (t2 - t1) / 36); }
  { double t1 = run<2, 32, 32, 4>(512, 768, 768, 100), t2 = run<2, 32, 32, 4>(512, 768, 3072, 100); printf("32x32 NS4  N=768 : %6.1f %6.1f  per-tile %.3f us\n", t1, t2, (t2 - t1) / 36); }
  { double t1 = run<2, 128, 128, 2>(4096, 4096, 768, 20), t2 = run<2, 128, 128, 2>(4096, 4096, 3072, 20); printf("128x128 NS2 M=4096 N=4096: %6.1f %6.1f  per-tile %.3f us\n", t1, t2, (t2 - t1) / 36); }
  return 0;
}
