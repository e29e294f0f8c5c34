// Fixed cost vs per-K-tile cost of the product GEMM (csrc/gemm_h16.h) at the DDPM step's M = 512.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_ksweep_0 gemm_ksweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "exp/src_r04/gemm_h16.h"   // round-4 sources: the ablation switches live there, not in the product
using namespace msd;

template <int NP, int BM, int BN, int NS>
double run(int M, int N, int K, int iters, bool cold_a = false) {
  const int COPIES = 24;  // rotate weight copies: cold like the step
  h16_t *a[2], *b[2]; h16_t* o[2];
  for (int i = 0; i < 2; ++i) { hipMalloc(&a[i], (size_t)COPIES * M * K * 2); hipMalloc(&b[i], (size_t)COPIES * N * K * 2); hipMalloc(&o[i], (size_t)M * N * 2);
    hipMemset(a[i], 0x3c, (size_t)COPIES * M * K * 2); hipMemset(b[i], 0x3b, (size_t)COPIES * N * K * 2); }
  GemmParams p; for (int i = 0; i < 2; ++i) p.A[i] = a[i]; p.lda = K; p.ldb = K; p.M = M; p.N = N; p.K = K;
  EpiStoreH16<NP> es; es.out[0] = o[0]; es.out[1] = o[1]; es.ldc = N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&](int it) { for (int i = 0; i < 2; ++i) { p.B[i] = b[i] + (size_t)(it % COPIES) * N * K; if (cold_a) p.A[i] = a[i] + (size_t)(it % COPIES) * M * K; } launch_gemm_h16_dma<NP, BM, BN, NS>(p, es, 0); };
  for (int i = 0; i < 5; ++i) go(i);
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < iters; ++i) go(i + 5); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  for (int i = 0; i < 2; ++i) { hipFree(a[i]); hipFree(b[i]); hipFree(o[i]); }
  return ms * 1e3 / iters;
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
  {  // launch floor: back-to-back dependent empty kernels on one stream
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel, 256 blocks, stream back-to-back: %.2f us / launch\n", ms);
  }
  const int Ks[] = {128, 256, 512, 768, 1536, 3072};
  printf("cold weights, bf16x3, M = 512; us per launch by K (tiles of 64)\n");
  printf("%-22s", "tile");
  for (int K : Ks) printf(" K=%-6d", K);
  printf("\n%-22s", "64x96 NS3  N=2304");
  for (int K : Ks) printf(" %7.1f ", run<2, 64, 96, 3>(512, 2304, K, 96));
  printf("\n%-22s", "64x128 NS3 N=4096");
  for (int K : Ks) printf(" %7.1f ", run<2, 64, 128, 3>(512, 4096, K, 96));
  printf("\n%-22s", "64x32 NS4  N=768");
  for (int K : Ks) printf(" %7.1f ", run<2, 64, 32, 4>(512, 768, K, 96));
  printf("\n%-22s", "32x32 NS4  N=768");
  for (int K : Ks) printf(" %7.1f ", run<2, 32, 32, 4>(512, 768, K, 96));
  printf("\ncold A as well (activations written by the previous kernel are not in this XCD's L2):\n");
  printf("%-22s", "64x96 NS3  N=2304");
  for (int K : Ks) printf(" %7.1f ", run<2, 64, 96, 3>(512, 2304, K, 96, true));
  printf("\n%-22s", "64x128 NS3 N=4096");
  for (int K : Ks) printf(" %7.1f ", run<2, 64, 128, 3>(512, 4096, K, 96, true));
  printf("\n%-22s", "64x32 NS4  N=768");
  for (int K : Ks) printf(" %7.1f ", run<2, 64, 32, 4>(512, 768, K, 96, true));
  printf("\n");
  return 0;
}
