// Common device helpers for the gfx950 kernels (wave64, MFMA, bf16 split).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msd {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator

constexpr int kWave = 64;

// round-to-nearest-even float -> bf16 bits (finite inputs; NaN kept quiet)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// hi/lo split: x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi)  (error ~2^-17 |x|)
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

__device__ __forceinline__ uint32_t pack2(bf16_t a, bf16_t b) {
  return (uint32_t)a | ((uint32_t)b << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // flax.linen.gelu(approximate=True): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float c = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * (x * x * x))));
}

__device__ __forceinline__ float swishf(float x) { return x / (1.0f + __expf(-x)); }

}  // namespace msd
