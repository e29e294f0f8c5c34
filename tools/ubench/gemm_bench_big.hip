// Large-M (batched songs) tile study for csrc/gemm_h16.h: M = 2 * B * 256 rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_bench_big_0 gemm_bench_big.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "exp/src_r04/gemm_h16.h"   // round-4 sources: the ablation switches live there, not in the product
using namespace msd;

template <int NP, int BM, int BN, int NS>
double run(int M, int N, int K, int iters, bool resid) {
  h16_t *a[2], *b[2]; float* c; h16_t* o[2];
  for (int i = 0; i < 2; ++i) { hipMalloc(&a[i], (size_t)M * K * 2); hipMalloc(&b[i], (size_t)N * K * 2); hipMalloc(&o[i], (size_t)M * N * 2);
    hipMemset(a[i], 0x3c, (size_t)M * K * 2); hipMemset(b[i], 0x3b, (size_t)N * K * 2); }
  hipMalloc(&c, (size_t)M * N * 4); hipMemset(c, 0, (size_t)M * N * 4);
  GemmParams p; for (int i = 0; i < 2; ++i) { p.A[i] = a[i]; p.B[i] = b[i]; } p.lda = K; p.ldb = K; p.M = M; p.N = N; p.K = K;
  EpiResidual er{c, N}; EpiStoreH16<NP> es; es.out[0] = o[0]; es.out[1] = o[1]; es.ldc = N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&]() { if (resid) launch_gemm_h16_dma<NP, BM, BN, NS>(p, er, 0); else launch_gemm_h16_dma<NP, BM, BN, NS>(p, es, 0); };
  for (int i = 0; i < 3; ++i) go();
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < iters; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  for (int i = 0; i < 2; ++i) { hipFree(a[i]); hipFree(b[i]); hipFree(o[i]); } hipFree(c);
  return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8;
  const int M = 2 * B * 256;
  struct S { const char* name; int N, K; bool resid; } shapes[] = {
      {"qkv     ", 2304, 768, false}, {"attn_out", 768, 768, true}, {"mlp_in  ", 4096, 768, false}, {"mlp_out ", 768, 2048, true}};
  printf("M = %d\n", M);
  for (auto& s : shapes) {
    const double gf = 2.0 * M * s.N * s.K / 1e9;
    double t[6];
    t[0] = run<2, 64, 64, 2>(M, s.N, s.K, 20, s.resid);
    t[1] = run<2, 64, 128, 3>(M, s.N, s.K, 20, s.resid);
    t[2] = run<2, 128, 64, 3>(M, s.N, s.K, 20, s.resid);
    t[3] = run<2, 128, 128, 2>(M, s.N, s.K, 20, s.resid);
    t[4] = s.N % 96 == 0 ? run<2, 128, 96, 2>(M, s.N, s.K, 20, s.resid) : 0;
    t[5] = run<1, 128, 128, 3>(M, s.N, s.K, 20, s.resid);
    printf("%s N=%d K=%d (%.1f GF): 64x64/2 %.0f us %.0f TF | 64x128/3 %.0f us %.0f TF | 128x64/3 %.0f us %.0f TF | 128x128/2 %.0f us %.0f TF | 128x96/2 %.0f us | bf16 128x128/3 %.0f us %.0f TF\n",
           s.name, s.N, s.K, gf, t[0], gf / t[0] * 1e-3, t[1], gf / t[1] * 1e-3, t[2], gf / t[2] * 1e-3, t[3], gf / t[3] * 1e-3, t[4], t[5], gf / t[5] * 1e-3);
  }
  return 0;
}
