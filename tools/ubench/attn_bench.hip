// Standalone timing + ablation harness for csrc/attention.h.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMSD_ATT_ABL=n] -o attn_bench attn_bench.hip
// MSD_ATT_ABL: 0 full; 1 fast __expf; 2 no exp at all; 3 no DMA after the prologue; 4 no PV MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include "exp/src_r04/attention.h"   // round-4 sources: the ablation switches live there, not in the product
using namespace msd;

template <int NP, int KS = 1>
double run(int segs, int heads, int nq, int nkeys, int kpad, int iters) {
  const int J = heads * 64;
  AttnParams p; int* nk; hipMalloc(&nk, segs * 4);
  std::vector<int> h(segs, nkeys); hipMemcpy(nk, h.data(), segs * 4, hipMemcpyHostToDevice);
  h16_t *q[2], *k[2], *v[2], *o[2];
  for (int i = 0; i < 2; ++i) {
    hipMalloc(&q[i], (size_t)segs * nq * J * 2); hipMalloc(&k[i], (size_t)segs * kpad * J * 2);
    hipMalloc(&v[i], (size_t)segs * kpad * J * 2); hipMalloc(&o[i], (size_t)segs * nq * J * 2);
    hipMemset(q[i], 0x3c, (size_t)segs * nq * J * 2); hipMemset(k[i], 0x3b, (size_t)segs * kpad * J * 2);
    hipMemset(v[i], 0x3c, (size_t)segs * kpad * J * 2);
    p.q[i] = q[i]; p.k[i] = k[i]; p.vt[i] = v[i]; p.o[i] = o[i];
  }
  p.n_keys = nk; p.ldq = J; p.ldk = J; p.ldo = J; p.vt_ld = kpad; p.q_rows_per_seg = nq;
  p.k_seg_stride = (size_t)kpad * J; p.vt_seg_stride = (size_t)J * kpad; p.k_rows = kpad;
  p.ksplit = KS; p.total_rows = segs * nq;
  hipMalloc(&p.part_o, (size_t)KS * segs * nq * J * 4); hipMalloc(&p.part_ml, (size_t)KS * segs * nq * heads * 8);
  for (int i = 0; i < 3; ++i) launch_attention<NP>(p, heads, segs, 0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < iters; ++i) launch_attention<NP>(p, heads, segs, 0); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / iters;
}

int main() {
  printf("MSD_ATT_ABL=%d\n", MSD_ATT_ABL);
  printf("self  2x12 heads 256x256     : bf16x3 %6.1f us | bf16 %6.1f us\n", run<2>(2, 12, 256, 256, 256, 200), run<1>(2, 12, 256, 256, 256, 200));
  printf("cross 1x12 heads 256x1344    : bf16x3 %6.1f us | bf16 %6.1f us\n", run<2>(1, 12, 256, 1344, 2304, 200), run<1>(1, 12, 256, 1344, 2304, 200));
  printf("cross 1x12 heads 256x2304    : bf16x3 %6.1f us | bf16 %6.1f us\n", run<2>(1, 12, 256, 2304, 2304, 200), run<1>(1, 12, 256, 2304, 2304, 200));
  printf("cross 256x1344 ksplit 2/3/4/6: bf16x3 %6.1f %6.1f %6.1f %6.1f us\n", run<2, 2>(1, 12, 256, 1344, 2304, 200), run<2, 3>(1, 12, 256, 1344, 2304, 200), run<2, 4>(1, 12, 256, 1344, 2304, 200), run<2, 6>(1, 12, 256, 1344, 2304, 200));
  printf("cross 256x2304 ksplit 2/3/4/6: bf16x3 %6.1f %6.1f %6.1f %6.1f us\n", run<2, 2>(1, 12, 256, 2304, 2304, 200), run<2, 3>(1, 12, 256, 2304, 2304, 200), run<2, 4>(1, 12, 256, 2304, 2304, 200), run<2, 6>(1, 12, 256, 2304, 2304, 200));
  printf("self  256x256  ksplit 2      : bf16x3 %6.1f us\n", run<2, 2>(2, 12, 256, 256, 256, 200));
  return 0;
}
