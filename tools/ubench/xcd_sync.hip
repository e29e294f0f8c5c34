// Can the blocks of ONE XCD (32 CUs behind one L2) hand data to each other inside a kernel cheaply?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o xcd_sync_0 xcd_sync.hip
// 1. block -> XCD map: XCC_ID of block b against b % 8 (what the XCD-aware tile maps assume)
// 2. barrier among the 32 blocks of an XCD (one L2 atomic + a polling load that bypasses the CU's L1),
//    against a barrier among all 256 blocks, us per barrier
// 3. producer -> consumer hand-off through the XCD's L2 across such a barrier, same addresses rewritten
//    every round: plain loads (may hit stale L1 lines), L1-bypassing loads (sc1), LDS-DMA with sc1
// Every spin is bounded: a broken protocol reports an error instead of hanging the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kSpinLimit = 2000000;

__global__ void probe_kernel(unsigned* xcc, unsigned* hwid) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    hwid[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  }
}

// barrier over `n` blocks on counter `cnt` (monotonic); returns false on timeout
__device__ __forceinline__ bool group_barrier(unsigned* cnt, unsigned n, int* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this block's stores have reached L2
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old / n + 1) * n;
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > kSpinLimit) { ok = false; atomicAdd(err, 1); break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

// mode 0: barrier per XCD (32 blocks), mode 1: one barrier over all blocks
__global__ void barrier_bench_kernel(unsigned* cnt, int iters, int mode, long long* clocks, int* err) {
  extern __shared__ char smem[];
  const int xcd = blockIdx.x & 7;
  unsigned* c = mode == 0 ? cnt + xcd * 64 : cnt + 8 * 64;
  const unsigned n = mode == 0 ? gridDim.x / 8 : gridDim.x;
  group_barrier(c, n, err);
  const long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i)
    if (!group_barrier(c, n, err)) break;
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Round k: block (xcd, slot) writes 1024 floats f(k, xcd, slot, i) to its slice of buf (same addresses every
// round); barrier over the XCD (mode 0) or the grid (mode 1); it then reads the slice of its neighbour
// (slot+1 of the same XCD in mode 0; block+1 -- another XCD -- in mode 1) in three ways and counts mismatches.
__global__ void handoff_kernel(float* buf, unsigned* cnt, int rounds, int mode, int* bad_plain, int* bad_sc1,
                               int* bad_dma, int* err) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x / 8;
  unsigned* c = mode == 0 ? cnt + xcd * 64 : cnt + 8 * 64;
  const unsigned n = mode == 0 ? nslot : gridDim.x;
  const int tid = threadIdx.x;
  int nb_block;
  if (mode == 0) nb_block = ((slot + 1) % nslot) * 8 + xcd; else nb_block = (blockIdx.x + 1) % gridDim.x;
  float* mine = buf + (size_t)blockIdx.x * 1024;
  const float* theirs = buf + (size_t)nb_block * 1024;
  int p = 0, s = 0, d = 0;
  for (int k = 0; k < rounds; ++k) {
    for (int i = tid; i < 1024; i += blockDim.x) mine[i] = (float)(k * 7 + blockIdx.x) + (float)i * 0.001f;
    if (!group_barrier(c, n, err)) break;
    float* lds = reinterpret_cast<float*>(smem);
    // LDS-DMA with sc1: wave w moves 1 KiB chunk w (4 waves x 256 floats)
    {
      const int wave = tid >> 6, lane = tid & 63;
      __builtin_amdgcn_global_load_lds((gptr_t)(theirs + wave * 256 + lane * 4), (lptr_t)(lds + wave * 256), 16, 0, 16);
    }
    for (int i = tid; i < 1024; i += blockDim.x) {
      const float want = (float)(k * 7 + nb_block) + (float)i * 0.001f;
      const float a = theirs[i];
      const float b = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      p += (a != want);
      s += (b != want);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += blockDim.x) {
      const float want = (float)(k * 7 + nb_block) + (float)i * 0.001f;
      d += (lds[i] != want);
    }
    // nobody may overwrite its slice before every reader is done with it
    if (!group_barrier(c, n, err)) break;
  }
  if (p) atomicAdd(bad_plain, p);
  if (s) atomicAdd(bad_sc1, s);
  if (d) atomicAdd(bad_dma, d);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus;
  printf("device %s, %d CUs\n", prop.name, cus);
  const int lds = 100 * 1024;   // one block per CU
  CHECK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHECK(hipFuncSetAttribute((const void*)barrier_bench_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHECK(hipFuncSetAttribute((const void*)handoff_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  unsigned *xcc, *hwid, *cnt;
  long long* clocks;
  int* flags;
  float* buf;
  CHECK(hipMalloc(&xcc, blocks * 4)); CHECK(hipMalloc(&hwid, blocks * 4)); CHECK(hipMalloc(&cnt, 9 * 64 * 4));
  CHECK(hipMalloc(&clocks, blocks * 8)); CHECK(hipMalloc(&flags, 16 * 4)); CHECK(hipMalloc(&buf, (size_t)blocks * 1024 * 4));
  // 1. block -> XCD
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), lds, 0, xcc, hwid);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> hx(blocks), hh(blocks);
    CHECK(hipMemcpy(hx.data(), xcc, blocks * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hh.data(), hwid, blocks * 4, hipMemcpyDeviceToHost));
    int mism = 0, per[16] = {0};
    for (int b = 0; b < blocks; ++b) { mism += ((hx[b] & 15) != (unsigned)(b & 7)); per[hx[b] & 15]++; }
    printf("probe %d: XCC_ID != block %% 8 for %d of %d blocks; blocks per XCC:", rep, mism, blocks);
    for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
    printf("  (raw xcc[0..3] = %08x %08x %08x %08x)\n", hx[0], hx[1], hx[2], hx[3]);
  }
  // 2. barrier cost
  for (int mode = 0; mode < 2; ++mode) {
    CHECK(hipMemset(cnt, 0, 9 * 64 * 4)); CHECK(hipMemset(flags, 0, 64));
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(barrier_bench_kernel, dim3(blocks), dim3(256), lds, 0, cnt, iters, mode, clocks, flags);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int herr = 0;
    CHECK(hipMemcpy(&herr, flags, 4, hipMemcpyDeviceToHost));
    std::vector<long long> hc(blocks);
    CHECK(hipMemcpy(hc.data(), clocks, blocks * 8, hipMemcpyDeviceToHost));
    printf("%s barrier: %.3f us each (kernel %.3f ms / %d), wall_clock ticks/barrier %.1f, timeouts %d\n",
           mode == 0 ? "per-XCD (32 blocks)" : "whole-grid", ms * 1e3 / iters, ms, iters, (double)hc[0] / iters, herr);
  }
  // 3. hand-off correctness
  for (int mode = 0; mode < 2; ++mode) {
    CHECK(hipMemset(cnt, 0, 9 * 64 * 4)); CHECK(hipMemset(flags, 0, 64));
    const int rounds = 500;
    hipLaunchKernelGGL(handoff_kernel, dim3(blocks), dim3(256), lds, 0, buf, cnt, rounds, mode, flags + 1, flags + 2, flags + 3, flags);
    CHECK(hipDeviceSynchronize());
    int h[4];
    CHECK(hipMemcpy(h, flags, 16, hipMemcpyDeviceToHost));
    printf("hand-off %s, %d rounds x %d blocks x 1024 floats: mismatches plain %d, sc1 load %d, LDS-DMA sc1 %d, timeouts %d\n",
           mode == 0 ? "inside an XCD" : "across XCDs", rounds, blocks, h[1], h[2], h[3], h[0]);
  }
  return 0;
}
