// Does the cache policy of a kernel's LAST stores change what the kernel boundary behind it costs?  (Round 5.)
// The L2s of the eight XCDs are private: at the end of a kernel the runtime's release writes back what is dirty in
// them before a dependent kernel may start.  A chain of 64 dependent launches (256 blocks x 256 threads; each block
// reads the 16 KiB slice another block of its predecessor wrote, works ~`work` iterations, writes its own 16 KiB):
// stores plain / nt / sc1 (write-through to the memory side) / sc0 sc1 -- us per launch, and the sum is checked.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/store_policy_0 tools/ubench/store_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int POLICY>
__device__ __forceinline__ void store4(float4* p, float4 vv) {
  const f32x4 v = {vv.x, vv.y, vv.z, vv.w};
  if constexpr (POLICY == 0) *p = vv;
  else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}

template <int POLICY>
__global__ void __launch_bounds__(256) chain_kernel(const float* in, float* out, int G, int work, int kb) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, b = blockIdx.x, src = (b + 37) % G;
  float t = 0.f;
  for (int rep = 0; rep < kb; ++rep) {   // kb x 16 KiB per block
    const float4* a = reinterpret_cast<const float4*>(in + ((size_t)rep * G + src) * 4096);   // (kb = 1: 16 KiB per block)
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[tid + 256 * i];
    for (int i = 0; i < work; ++i) t = t * 0.999f + 1e-9f;
    float4* o = reinterpret_cast<float4*>(out + ((size_t)rep * G + b) * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = 1.0f + t * 1e-30f;
      store4<POLICY>(o + tid + 256 * i, make_float4(v[i].x + d, v[i].y + d, v[i].z + d, v[i].w + d));
    }
  }
}

template <int POLICY>
double run(float* buf[2], int G, int work, int kb, hipStream_t s, bool* ok) {
  const int K = 64, reps = 20;
  hipGraph_t graph; hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < K; ++k) hipLaunchKernelGGL(chain_kernel<POLICY>, dim3(G), dim3(256), 64 * 1024, s, buf[k & 1], buf[(k + 1) & 1], G, work, kb);
  CHECK(hipStreamEndCapture(s, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CHECK(hipMemsetAsync(buf[0], 0, (size_t)kb * G * 4096 * sizeof(float), s));
  for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(exec, s));
  CHECK(hipStreamSynchronize(s));
  hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  CHECK(hipEventRecord(t0, s));
  for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(exec, s));
  CHECK(hipEventRecord(t1, s));
  CHECK(hipStreamSynchronize(s));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, t0, t1));
  std::vector<float> h((size_t)kb * G * 4096);
  CHECK(hipMemcpy(h.data(), buf[0], h.size() * sizeof(float), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (float x : h) bad += (x != (float)(K * (reps + 3)));
  *ok = bad == 0;
  if (bad) printf("   [policy %d: %zu of %zu wrong, first values %g %g %g, expected %d]\n", POLICY, bad, h.size(), h[0], h[1], h[h.size() - 1], K * (reps + 3));
  CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
  return ms * 1e3 / (reps * K);
}

int main() {
  const int G = 256;
  hipStream_t s; CHECK(hipStreamCreate(&s));
  float* buf[2];
  for (int i = 0; i < 2; ++i) CHECK(hipMalloc(&buf[i], (size_t)8 * G * 4096 * sizeof(float)));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-10s %-8s %10s %10s %10s %10s   (us per launch; 64 dependent launches of 256 blocks)\n", "KiB/block", "work", "plain", "nt", "sc1", "sc0 sc1");
  for (int kb : {1, 2, 4}) for (int work : {0, 500}) {
    bool ok[4];
    const double a = run<0>(buf, G, work, kb, s, &ok[0]), b = run<1>(buf, G, work, kb, s, &ok[1]);
    const double c = run<2>(buf, G, work, kb, s, &ok[2]), d = run<3>(buf, G, work, kb, s, &ok[3]);
    printf("%-10d %-8d %10.3f %10.3f %10.3f %10.3f   %s\n", kb * 16, work, a, b, c, d, (ok[0] && ok[1] && ok[2] && ok[3]) ? "ok" : "WRONG");
    fflush(stdout);
  }
  return 0;
}
