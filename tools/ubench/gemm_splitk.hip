// What would split-K buy the M = 512 GEMMs of one DDPM step?  (next-round question)
// A block's L2 -> LDS ingest is (BM + BN) * K per tile; with one tile per CU the tile area is fixed by
// M * N / 256, so the only way to cut the ingest is a larger tile over a SHORTER K: s-way split-K lets
// tiles be s times larger.  This harness times the main loop + a plain two-plane store of such a split
// WITHOUT its reduction, by running the product kernel on (M, s * N, K / s): same block count, same
// per-block ingest, same output bytes per block as the split would have -- an upper bound on the gain,
// to be compared with the <= 1-2 us a same-XCD reduction hand-off costs (tools/ubench/xcd_sync.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/ubench/exp/src_r04 -o tools/ubench/gemm_splitk_0 tools/ubench/gemm_splitk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "gemm_h16.h"
using namespace msd;

// every launch reads a different copy of the weights (COPIES * bytes > L2 + MALL), as in the step
template <int NP, int BM, int BN, int NS>
double run_cold(int M, int N, int K, int iters) {
  const size_t wbytes = (size_t)N * K * 2;
  int COPIES = (int)((600ull << 20) / (2 * wbytes)) + 1;
  if (COPIES > 64) COPIES = 64;
  h16_t *a[2], *b[2], *o[2];
  for (int i = 0; i < 2; ++i) {
    hipMalloc(&a[i], (size_t)M * K * 2); hipMalloc(&b[i], (size_t)COPIES * wbytes); hipMalloc(&o[i], (size_t)M * N * 2);
    hipMemset(a[i], 0x3c, (size_t)M * K * 2); hipMemset(b[i], 0x3b, (size_t)COPIES * wbytes);
  }
  GemmParams p; for (int i = 0; i < 2; ++i) p.A[i] = a[i]; p.lda = K; p.ldb = K; p.M = M; p.N = N; p.K = K;
  EpiStoreH16<NP> es; es.out[0] = o[0]; es.out[1] = o[1]; es.ldc = N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&](int it) {
    for (int i = 0; i < 2; ++i) p.B[i] = b[i] + (size_t)(it % COPIES) * N * K;
    launch_gemm_h16_dma<NP, BM, BN, NS>(p, es, 0);
  };
  for (int i = 0; i < 5; ++i) go(i);
  hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < iters; ++i) go(i + 5); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  for (int i = 0; i < 2; ++i) { hipFree(a[i]); hipFree(b[i]); hipFree(o[i]); }
  return ms * 1e3 / iters;
}

#define T(NP, BM, BN, NS, M, N, K) printf("   %3dx%-3d NS%d on %4d x %5d x %4d (%3d blocks): %6.1f us\n", BM, BN, NS, M, N, K, \
                                          ((M) / (BM)) * ((N) / (BN)), run_cold<NP, BM, BN, NS>(M, N, K, 192))

int main() {
  printf("mlp_in 512 x 4096 x 768 (product: 64x128 NS3)\n");
  T(2, 64, 128, 3, 512, 4096, 768);
  T(2, 128, 128, 2, 512, 8192, 384);    // 2-way split
  T(2, 128, 128, 3, 512, 8192, 384);
  printf("mlp_out 512 x 768 x 2048 (product: 64x32 NS4)\n");
  T(2, 64, 32, 4, 512, 768, 2048);
  T(2, 64, 64, 3, 512, 1536, 1024);     // 2-way
  T(2, 64, 128, 3, 512, 3072, 512);     // 4-way
  T(2, 128, 64, 3, 512, 3072, 512);     // 4-way, tall
  T(2, 128, 128, 2, 512, 6144, 256);    // 8-way
  printf("qkv 512 x 2304 x 768 (product: 64x96 NS3)\n");
  T(2, 64, 96, 3, 512, 2304, 768);
  T(2, 128, 96, 2, 512, 4608, 384);     // 2-way
  printf("attn_out / cross_out 512 x 768 x 768 (product: 32x32 NS4)\n");
  T(2, 32, 32, 4, 512, 768, 768);
  T(2, 64, 32, 4, 512, 1536, 384);      // 2-way
  T(2, 64, 64, 3, 512, 3072, 192);      // 4-way
  T(2, 64, 96, 3, 512, 2304, 256);      // 3-way
  printf("cross_q 256 x 768 x 768 (product: 32x32 NS4)\n");
  T(2, 32, 32, 4, 256, 768, 768);
  T(2, 64, 32, 4, 256, 1536, 384);      // 2-way
  T(2, 64, 64, 3, 256, 3072, 192);      // 4-way
  return 0;
}
