// How much of the decoder's cross-attention launch is the HBM-cold K / V^T cache?  (round 6)
// The launch as the product runs it at one song (12 heads x 256 query rows, key split 4 merged in the launch, 64-row blocks,
// both planes), back to back in one stream:  WARM = the same K / V^T every launch (14 MB: they stay in the L2s / the
// memory-side cache)  vs  COLD = COPIES rotating caches (COPIES x 14 MB >> the 256 MB Infinity Cache: every launch finds
// its keys in HBM, which is what a DDPM step sees -- each layer's cache is read once per ~1 ms between 0.5 GB of other
// traffic).  The difference is the most a perfect K / V warm-up could give.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/attn_cold_0 tools/ubench/attn_cold.hip && tools/ubench/attn_cold_0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../music-spectrogram-diffusion_amd/csrc/attention.h"
using namespace msd;

int main() {
  const int heads = 12, nq = 256, kpad = 2304, J = heads * 64, KS = 4, COPIES = 40, iters = 400;
  int* tickets; hipMalloc(&tickets, 4096 * 4); hipMemset(tickets, 0, 4096 * 4);
  float *part_o, *part_ml;
  hipMalloc(&part_o, (size_t)8 * nq * J * 4); hipMalloc(&part_ml, (size_t)8 * nq * heads * 8);
  h16_t *q[2], *o[2];
  std::vector<h16_t*> k[2], v[2];
  for (int i = 0; i < 2; ++i) {
    hipMalloc(&q[i], (size_t)nq * J * 2); hipMalloc(&o[i], (size_t)nq * J * 2);
    hipMemset(q[i], 0x2c, (size_t)nq * J * 2);
    for (int c = 0; c < COPIES; ++c) {
      h16_t *kk, *vv;
      hipMalloc(&kk, (size_t)kpad * J * 2); hipMalloc(&vv, (size_t)kpad * J * 2);
      hipMemset(kk, 0x2b, (size_t)kpad * J * 2); hipMemset(vv, 0x3c, (size_t)kpad * J * 2);
      k[i].push_back(kk); v[i].push_back(vv);
    }
  }
  h16_t* wbuf[8];
  for (int i = 0; i < 8; ++i) { hipMalloc(&wbuf[i], (size_t)2 * 4096 * 768 * 2 * 4); hipMemset(wbuf[i], 0x11, (size_t)2 * 4096 * 768 * 2 * 4); }   // (8 x 50 MB: never cache-resident)
  for (int nkeys : {557, 1136, 1701}) {
    int* nk; hipMalloc(&nk, 4); hipMemcpy(nk, &nkeys, 4, hipMemcpyHostToDevice);
    for (int touch = -1; touch < 3; ++touch) {   // -1: no prefetch wave at all; 0: prefetch wave with a 6.3 MB weight target (what the product's launch carries: MLP-in's planes), no K / V touches; 1: + touch-ahead 2
      double us[2];
      for (int cold = 0; cold < 2; ++cold) {
        auto launch = [&](int it) {
          AttnParams p;
          const int c = cold ? it % COPIES : 0;
          for (int i = 0; i < 2; ++i) { p.q[i] = q[i]; p.k[i] = k[i][c]; p.vt[i] = v[i][c]; p.o[i] = o[i]; }
          p.n_keys = nk; p.ldq = J; p.ldk = J; p.ldo = J; p.vt_ld = kpad; p.q_rows_per_seg = nq;
          p.k_seg_stride = (size_t)kpad * J; p.vt_seg_stride = (size_t)J * kpad; p.k_rows = kpad; p.vt_cols = kpad;
          p.ksplit = nkeys > 768 ? KS : 2; p.total_rows = nq; p.part_o = part_o; p.part_ml = part_ml; p.tickets = tickets;
          p.touch_ahead = touch > 0 ? 2 : 0; p.pf_late = touch == 2 ? 1 : 0;
          if (touch >= 0) { PrefetchTarget t; t.set(wbuf[it % 8], wbuf[it % 8] + (size_t)4096 * 768, 4096, 768 * 2, 768 * 2); p.pf.add(t); }   // (rotating: cold like a step's weights)
          launch_attention<2>(p, heads, 1, 0);
        };
        for (int i = 0; i < 50; ++i) launch(i);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); for (int i = 0; i < iters; ++i) launch(i); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        us[cold] = ms * 1e3 / iters;
      }
      printf("cross-attention 12 x 256 x %4d keys, %-34s: warm K/V %6.2f us   cold K/V %6.2f us   (cold - warm %5.2f)\n", nkeys,
             touch < 0 ? "no prefetch wave" : (touch == 0 ? "prefetch wave (12.6 MB target)" : (touch == 1 ? "prefetch wave + K/V touch-ahead 2" : "... weights behind barrier 0")), us[0], us[1], us[1] - us[0]);
    }
  }
  return 0;
}
