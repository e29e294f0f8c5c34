// Common device helpers for the gfx950 kernels (wave64, MFMA, half-plane operand split).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// MSD_EXPERIMENTS=1 builds the A/B library of tools/ubench/exp (rejected kernels + environment switches: XCD-resident
// chains, split-K, dual launches, batched tile variants).  The PRODUCT build (0) contains none of them and reads no
// environment variable: what a caller may choose is in msd_config (include/msd_amd.h, ABI 4).
#ifndef MSD_EXPERIMENTS
#define MSD_EXPERIMENTS 0
#endif

namespace msd {

constexpr bool kExperiments = MSD_EXPERIMENTS != 0;

typedef uint16_t h16_t;  // raw bits of one element of an operand plane (IEEE half, or bfloat16: see below)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator

constexpr int kWave = 64;

// Operand planes: 16-bit elements, one plane (hi) or two (hi + lo) per GEMM / attention operand.
//
// Default build (libmsd_amd.so): IEEE half (v_mfma_f32_*_f16 -- the rate of the bf16 MFMAs).  hi + lo then carry 22
// significand bits instead of bfloat16's 16, which puts the split-operand mode on the float32 oracle's error
// (6.8e-5 against 1.4e-4 on the 1000-step segment; same bytes, same MFMA count).  Half has 5 exponent
// bits, so
//   * an activation with |x| > 65504 does not fit; it is DETECTED (RangeCheck below, one flag word per handle) and
//     msd_encode / msd_sample then fail with MSD_ERR_RANGE, pointing at the bfloat16-plane build;
//   * weights are packed multiplied by kWScale and every GEMM multiplies its fp32 accumulators by kWScaleInv (powers
//     of two: exact), so that the lo plane of |w| ~ 0.03 weights is a NORMAL half (7e-6 would be subnormal: 2
//     significant bits).  |w| < 128 is representable (checked at load time: msd_finalize_weights).
// -DMSD_PLANE_BF16=1 (libmsd_amd_bf16.so): bfloat16 planes -- 8 exponent bits, no saturation, no weight scale, 16
// significand bits in hi + lo: the range-safe alternative for weights / activations beyond the half range.
#ifndef MSD_PLANE_BF16
#define MSD_PLANE_BF16 0
#endif

#if MSD_PLANE_BF16
typedef __bf16 plane_elem;
#define MSD_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MSD_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
constexpr float kWScale = 1.0f, kWScaleInv = 1.0f, kPlaneMax = 3.0e38f;
constexpr bool kPlaneSaturates = false;
constexpr const char* kPlaneName = "bfloat16 planes";
#else
typedef _Float16 plane_elem;
#define MSD_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MSD_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
constexpr float kWScale = 512.0f, kWScaleInv = 1.0f / 512.0f, kPlaneMax = 65504.0f;
constexpr bool kPlaneSaturates = true;
constexpr const char* kPlaneName = "half planes";
#endif

typedef plane_elem plane2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// round-to-nearest-even float -> plane bits / back.  NOT saturating: a half-plane conversion of |x| > 65504 gives
// inf -- every conversion site of an ACTIVATION therefore feeds a RangeCheck (below), and a flagged msd_sample /
// msd_encode fails with MSD_ERR_RANGE instead of returning a wrong spectrogram.  (Round 2 clamped silently, with two
// v_med3_f32 per split; the reference is float32 and has no such failure mode: gin/.../t5_base.gin:72.)
__device__ __forceinline__ h16_t f2h(float f) { return __builtin_bit_cast(h16_t, (plane_elem)f); }
__device__ __forceinline__ float h2f(h16_t b) { return (float)__builtin_bit_cast(plane_elem, b); }

// two values -> one packed dword of the hi plane (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: one instruction)
__device__ __forceinline__ uint32_t cvt2_h16(float a, float b) {
  const f32x2 x = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, plane2));
}

// hi/lo split of two values at once: x ~= hi + lo with hi = plane(x), lo = plane(x - hi); both planes come out as
// packed dwords (2 packed conversions + 2 widening conversions + one packed subtract for the pair, against ~15
// scalar instructions for two split_h16 calls: the plane-writing epilogues were VALU-heavy)
__device__ __forceinline__ void split2_h16(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f32x2 x = {a, b};
  const plane2 h = __builtin_convertvector(x, plane2);
  const f32x2 r = x - __builtin_convertvector(h, f32x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, plane2));
}

// hi/lo split of one value
__device__ __forceinline__ void split_h16(float x, h16_t& hi, h16_t& lo) {
  hi = f2h(x);
  lo = f2h(x - h2f(hi));
}

__device__ __forceinline__ uint32_t pack2(h16_t a, h16_t b) {
  return (uint32_t)a | ((uint32_t)b << 16);
}

// Range check of the values a thread converts to half planes: running max of |x| (one v_max3_f32 per two values),
// ONE compare at the end; a thread that saw |x| > 65504 stores `tag` (which kernel class: msd_api.hip) to the
// handle's flag word.  Compiles to nothing in the bfloat16-plane build (float32's exponent range).
struct RangeCheck {
  float m = 0.f;
  __device__ __forceinline__ void see(float a, float b) {
    if constexpr (kPlaneSaturates) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
  }
  __device__ __forceinline__ void see(float a) {
    if constexpr (kPlaneSaturates) asm("v_max_f32 %0, %0, |%1|" : "+v"(m) : "v"(a));
  }
  __device__ __forceinline__ void commit(unsigned* flag, unsigned tag) const {
    if constexpr (kPlaneSaturates) {
      if (flag != nullptr && !(m <= kPlaneMax)) *flag = tag;
    }
  }
};

// Kernel arguments live in a per-dispatch memory segment that is cold in every cache when a step's kernel starts (a
// graph replay touches it once per millisecond, between 2 GB of other traffic).  The compiler loads arguments lazily,
// near their first use, so a kernel with a 300 - 400 byte argument block (GemmParams + epilogue) walked through it in
// three or four DEPENDENT scalar-load rounds, each a miss to memory, before its first LDS-DMA went out.  warm_kernargs
// requests one word of each of the block's first LINES cache lines in one burst at the kernel's first instruction: one
// miss latency instead of three or four; the lazy loads behind it hit the scalar cache.  (Round 4; measured: DESIGN 6.)
template <int LINES>
__device__ __forceinline__ void warm_kernargs() {
  typedef const __attribute__((address_space(4))) uint32_t* kptr_t;
  kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t t[LINES];
#pragma unroll
  for (int k = 0; k < LINES; ++k) t[k] = ka[16 * k];
#pragma unroll
  for (int k = 0; k < LINES; ++k) asm volatile("" ::"s"(t[k]));
}
template <class... Args>
constexpr int kernarg_lines() {   // cache lines covered by the explicit arguments (each aligned to 8: pointers inside)
  int bytes = 0;
  const int sz[] = {(int)sizeof(Args)...};
  for (int s : sz) bytes = (bytes + 7) / 8 * 8 + s;
  return (bytes + 63) / 64;
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
