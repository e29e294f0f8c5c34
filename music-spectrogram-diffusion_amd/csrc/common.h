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

// round-to-nearest-even float -> bf16 bits.  The native cast lowers to
// v_cvt_pk_bf16_f32 (two values per instruction); hand-written bit arithmetic costs
// ~8 VALU ops per value and made the attention kernel VALU-bound.
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
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
  // flax.linen.gelu(approximate=True): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)
  //   = x - x / (e^(2u) + 1)            (v_exp_f32 + v_rcp_f32: ~1e-7 relative, vs ~50 instructions
  // of ocml tanhf in front of every gated-MLP store); e -> inf gives x, e -> 0 gives 0.
  const float u = 0.7978845608028654f * (x + 0.044715f * (x * x * x));
  const float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);   // e^(2u) = 2^(2u log2 e)
  return x - x * __builtin_amdgcn_rcpf(e + 1.0f);
}

// e^x as one v_exp_f32 (2^(x log2 e)); relative error ~|x| 2^-24, used for softmax where x <= 0
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ float swishf(float x) { return x / (1.0f + __expf(-x)); }

}  // namespace msd
