// Microbenchmark: L2-resident global_load_dwordx4 throughput per CU for the access
// shapes a small-M GEMM can use to fetch a [rows][K] bf16 operand on gfx950:
//   0: MFMA-fragment shape  -- 16 rows x 64 B per wave-instruction (lane: row l&15, 16B chunk l>>4)
//   1: half-row shape       --  8 rows x 128 B (lane: row l>>3, chunk l&7)      [LDS-staging shape, BK=64]
//   2: full 256 B rows      --  4 rows x 256 B (lane: row l>>4, chunk l&15)     [BK=128]
//   3: contiguous 1 KiB     --  one 1 KiB run per instruction
// Build: hipcc --offload-arch=gfx950 -O3 -o load_patterns load_patterns.hip ; run: ./load_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PAT>
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ buf, int row_stride16, int nrows, int iters, uint4* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int row, chunk, rows_per_inst;
  if (PAT == 0) { row = lane & 15; chunk = lane >> 4; rows_per_inst = 16; }
  else if (PAT == 1) { row = lane >> 3; chunk = lane & 7; rows_per_inst = 8; }
  else if (PAT == 2) { row = lane >> 4; chunk = lane & 15; rows_per_inst = 4; }
  else { row = 0; chunk = lane; rows_per_inst = 1; }
  uint4 acc = make_uint4(0, 0, 0, 0);
  // each wave walks its own row window; the whole buffer (2 MiB) stays in L2
  int r0 = ((blockIdx.x * 4 + wave) * 64) % nrows;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = (r0 + u * rows_per_inst + row) % nrows;
      const int kc = (PAT == 3) ? 0 : ((it * 4) % (row_stride16 / 16)) * ((PAT == 0) ? 4 : (PAT == 1 ? 8 : 16));
      const uint4 v = buf[(size_t)r * row_stride16 + (kc % row_stride16) + chunk];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    r0 = (r0 + 8 * rows_per_inst) % nrows;
  }
  if (acc.x == 0x12345678u) out[threadIdx.x] = acc;
}

int main() {
  const int row_bytes = 4096;             // K = 2048 bf16
  const int row_stride16 = row_bytes / 16;
  const int nrows = 512;                  // 2 MiB
  uint4* buf; uint4* out;
  hipMalloc(&buf, (size_t)nrows * row_bytes); hipMalloc(&out, 4096);
  hipMemset(buf, 1, (size_t)nrows * row_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu)
    for (int pat = 0; pat < 4; ++pat) {
      const int grid = 256 * blocks_per_cu;
      auto launch = [&]() {
        if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, buf, row_stride16, nrows, iters, out);
        if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, buf, row_stride16, nrows, iters, out);
        if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, buf, row_stride16, nrows, iters, out);
        if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, buf, row_stride16, nrows, iters, out);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)grid * 256 * 16.0 * 8 * iters;
      printf("pattern %d blocks/CU %d: %.1f GB/s total, %.1f B/clk/CU @2.1GHz (%.3f ms)\n", pat, blocks_per_cu,
             bytes / ms / 1e6, bytes / (ms * 1e-3) / 256 / 2.1e9, ms);
    }
  return 0;
}
