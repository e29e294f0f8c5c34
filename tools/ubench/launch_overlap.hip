// What does a KERNEL BOUNDARY cost against a DEVICE-FLAG hand-off between two co-scheduled launches?
// ("overlap launches instead of fusing them": launch the successor without a graph edge, let it
// issue its weight stages, gate its activation loads on a completion counter of the predecessor.)
//
// A chain of K dependent launches of G blocks x 256 threads.  Launch k: [W] reads `wbytes` per block from a read-only
// "weight" buffer (independent of the chain), [A] reads its 16 KiB slice of the activation buffer the predecessor wrote
// (slice of ANOTHER block: cross-CU, cross-XCD traffic), spins `work` iterations of dependent FMAs, writes value + 1 to
// its slice of the other buffer.  After K launches every element must equal K (stale reads = wrong answer).
//
//   mode 0  one stream, plain graph edges between the launches (what the product's step graph is)
//   mode 1  two graph branches (even / odd launches), NO edge between consecutive launches: launch k+1 does [W], then
//           waits for launch k's completion counter (agent-scope acquire), then [A] ...; launch k's blocks release +
//           count at their end
//   mode 2  as 1, but [W] AFTER the wait (only dispatch / entry overlap)
//   mode 3  one stream WITH edges and with the counter protocol as well (the protocol's own cost)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/launch_overlap_0 tools/ubench/launch_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Params {
  const float* w;       // weights: [G][wbytes/4] floats, rotated per launch
  size_t w_stride;      // floats between launches' weight sets
  const float* in;      // activations in  [G][4096]
  float* out;           // activations out [G][4096]
  unsigned* cnt;        // [K] completion counters
  unsigned* err;        // timeouts
  float* sink;
  int k, G, wfloats, work, mode, shift;
};

__global__ void __launch_bounds__(256) reset_kernel(unsigned* cnt, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) cnt[i] = 0;
}

__global__ void __launch_bounds__(256) chain_kernel(Params p) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, b = blockIdx.x;
  float acc = 0.f;
  auto weights = [&]() {
    const float4* w = reinterpret_cast<const float4*>(p.w + (size_t)p.k * p.w_stride + (size_t)b * p.wfloats);
    float4 s = make_float4(0, 0, 0, 0);
    for (int i = tid; i < p.wfloats / 4; i += 256) { const float4 v = w[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    acc += (s.x + s.y) + (s.z + s.w);
  };
  const bool flags = p.mode != 0;
  if (p.mode == 1 || p.mode == 3) weights();
  if (flags && p.k > 0) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(p.cnt + p.k - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.G) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 100000) { atomicAdd(p.err, 1u); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave: invalidate what its CU may hold of the old buffer
  }
  if (p.mode == 0 || p.mode == 2) weights();
  // activations: the slice written by block (b + shift) % G of the predecessor
  const int src = (b + p.shift) % p.G;
  const float4* a = reinterpret_cast<const float4*>(p.in + (size_t)src * 4096);
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = a[tid + 256 * i];
  float t = acc * 1e-30f;
  for (int i = 0; i < p.work; ++i) t = t * 0.999f + 1e-9f;   // dependent chain: ~4 clocks per iteration
  float4* o = reinterpret_cast<float4*>(p.out + (size_t)b * 4096);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float d = 1.0f + t * 1e-30f;
    o[tid + 256 * i] = make_float4(v[i].x + d, v[i].y + d, v[i].z + d, v[i].w + d);
  }
  if (flags) {
    __syncthreads();   // (includes the wait for this block's stores)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(p.cnt + p.k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 12345.678f) p.sink[0] = acc;
}

int main(int argc, char** argv) {
  const int K = 64, G = argc > 1 ? atoi(argv[1]) : 256;
  const int lds_kb = argc > 2 ? atoi(argv[2]) : 64;
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
  hipEvent_t ef, ej, t0, t1;
  CHECK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  const int max_wfloats = 64 * 1024 / 4;
  float *w, *buf[2], *sink; unsigned *cnt, *err;
  const size_t w_stride = (size_t)G * max_wfloats;
  CHECK(hipMalloc(&w, w_stride * K * sizeof(float)));              // 64 launches x 16 MB = 1 GB: HBM-cold every time
  CHECK(hipMemset(w, 0, w_stride * K * sizeof(float)));
  for (int i = 0; i < 2; ++i) CHECK(hipMalloc(&buf[i], (size_t)G * 4096 * sizeof(float)));
  CHECK(hipMalloc(&cnt, K * sizeof(unsigned))); CHECK(hipMalloc(&err, sizeof(unsigned))); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(err, 0, sizeof(unsigned)));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("chain of %d launches, %d blocks x 256 threads, %d KiB LDS per block\n", K, G, lds_kb);
  printf("%-6s %-8s %-8s %12s %12s %8s\n", "mode", "wKiB", "work", "us/launch", "vs mode 0", "check");
  for (int wkb : {0, 16, 64}) {
    for (int work : {0, 2000, 6000}) {
      double base = 0;
      for (int mode : {0, 3, 2, 1}) {
        if (wkb == 0 && mode == 1) continue;   // (== mode 2 without weights)
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(256), 0, sa, cnt, K);
        const bool two = mode == 1 || mode == 2;
        if (two) { CHECK(hipEventRecord(ef, sa)); CHECK(hipStreamWaitEvent(sb, ef, 0)); }
        for (int k = 0; k < K; ++k) {
          Params p;
          p.w = w; p.w_stride = w_stride; p.in = buf[k & 1]; p.out = buf[(k + 1) & 1]; p.cnt = cnt; p.err = err; p.sink = sink;
          p.k = k; p.G = G; p.wfloats = wkb * 256; p.work = work; p.mode = mode; p.shift = 37;
          hipLaunchKernelGGL(chain_kernel, dim3(G), dim3(256), lds_kb * 1024, (two && (k & 1)) ? sb : sa, p);
        }
        if (two) { CHECK(hipEventRecord(ej, sb)); CHECK(hipStreamWaitEvent(sa, ej, 0)); }
        CHECK(hipStreamEndCapture(sa, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CHECK(hipMemsetAsync(buf[0], 0, (size_t)G * 4096 * sizeof(float), sa));
        const int reps = 20;
        for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(exec, sa));   // warm-up (3 x 64 launches: buf[0] holds 192)
        CHECK(hipStreamSynchronize(sa));
        CHECK(hipEventRecord(t0, sa));
        for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(exec, sa));
        CHECK(hipEventRecord(t1, sa));
        CHECK(hipStreamSynchronize(sa));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, t0, t1));
        std::vector<float> h((size_t)G * 4096);
        CHECK(hipMemcpy(h.data(), buf[0], h.size() * sizeof(float), hipMemcpyDeviceToHost));
        const float want = (float)(K * (reps + 3));
        size_t bad = 0;
        for (float x : h) bad += (x != want);
        unsigned herr = 0; CHECK(hipMemcpy(&herr, err, sizeof(unsigned), hipMemcpyDeviceToHost));
        const double us = ms * 1e3 / (reps * K);
        if (mode == 0) base = us;
        printf("%-6d %-8d %-8d %12.3f %+11.3f  %s%s\n", mode, wkb, work, us, us - base, bad ? "WRONG" : "ok", herr ? " TIMEOUTS" : "");
        fflush(stdout);
        if (herr) CHECK(hipMemset(err, 0, sizeof(unsigned)));
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
      }
    }
  }
  return 0;
}
