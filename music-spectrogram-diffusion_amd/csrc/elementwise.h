// Non-GEMM kernels of the path: fused RMSNorm(+FiLM) -> bf16 planes, the fused
// DDPM/DDIM sampler update, weight packing, embeddings, Philox normal fill.
#pragma once
#include "common.h"

namespace msd {

// ---------------------------------------------------------------------------
// T5 LayerNorm (RMSNorm, eps 1e-6, fp32 statistics; layers.py:632-649) fused
// with the FiLM modulation x*(scale+1)+bias (layers.py:664-665) whose scale/bias
// come from the step-indexed table built at load time, and with the cast to the
// GEMM operand format.  One wave per row, float4 per lane.
//   film == nullptr     -> plain RMSNorm (cross-attention / encoder / final norm)
//   film[(step*slots + slot) * 2D + {0..D-1: scale, D..2D-1: bias}]
// OUT: 0 = bf16 plane, 1 = bf16 hi+lo planes, 2 = fp32
// ---------------------------------------------------------------------------
struct NormParams {
  const float* x;       // [rows, D]
  const float* gamma;   // [D]
  const float* film;    // table or nullptr
  const int* step_ptr;  // device scalar: current scan index i
  int film_slots, film_slot;
  int rows, D;
  h16_t* out[2];
  float* out_f32;
  unsigned* sat = nullptr;   // half-plane range flag (common.h RangeCheck)
  unsigned sat_tag = 1;
};

template <int OUT, int VPL>  // VPL = float4 loads per lane: D <= 256*VPL
__global__ void __launch_bounds__(256) rmsnorm_film_kernel(NormParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float* xr = p.x + (size_t)row * p.D;
  float4 v[VPL];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < p.D) {
      v[i] = *reinterpret_cast<const float4*>(xr + c);
      ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
  }
  ss = wave_sum(ss);
  const float rstd = 1.0f / sqrtf(ss / (float)p.D + 1e-6f);
  const float* fs = nullptr;
  if (p.film) fs = p.film + ((size_t)(*p.step_ptr) * p.film_slots + p.film_slot) * (2 * p.D);
  RangeCheck rc;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < p.D) {
      const float4 g = *reinterpret_cast<const float4*>(p.gamma + c);
      float y[4] = {v[i].x * rstd * g.x, v[i].y * rstd * g.y, v[i].z * rstd * g.z, v[i].w * rstd * g.w};
      if (fs) {
        const float4 sc = *reinterpret_cast<const float4*>(fs + c);
        const float4 bi = *reinterpret_cast<const float4*>(fs + p.D + c);
        y[0] = y[0] * (sc.x + 1.0f) + bi.x;
        y[1] = y[1] * (sc.y + 1.0f) + bi.y;
        y[2] = y[2] * (sc.z + 1.0f) + bi.z;
        y[3] = y[3] * (sc.w + 1.0f) + bi.w;
      }
      const size_t off = (size_t)row * p.D + c;
      if (OUT == 2) {
        *reinterpret_cast<float4*>(p.out_f32 + off) = make_float4(y[0], y[1], y[2], y[3]);
      } else {
        uint32_t h[2], l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          rc.see(y[2 * e], y[2 * e + 1]);
          if (OUT == 1) split2_h16(y[2 * e], y[2 * e + 1], h[e], l[e]);
          else h[e] = cvt2_h16(y[2 * e], y[2 * e + 1]);
        }
        *reinterpret_cast<uint2*>(p.out[0] + off) = make_uint2(h[0], h[1]);
        if (OUT == 1) *reinterpret_cast<uint2*>(p.out[1] + off) = make_uint2(l[0], l[1]);
      }
    }
  }
  rc.commit(p.sat, p.sat_tag);
}

// ---------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller standard normals (documented generator shared with
// oracle/philox.py).  key = (seed_lo, seed_hi); counter = (block, subseq,
// stream_lo, stream_hi); element e = 4*block + j, j in 0..3;
//   u = ((x >> 8) + 0.5) * 2^-24 ; (z0, z1) = sqrt(-2 ln u0) * (cos, sin)(2 pi u1)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += W0; k1 += W1;
  }
}

// The four normals of counter block `blk` of sub-sequence `subseq`: ONE function for the fill kernel below and for the
// sampler's own draw (sampler_step_kernel with no noise buffer), so that both write the same bits.
__device__ __forceinline__ void philox_normal4(uint32_t blk, uint32_t subseq, uint32_t seed_lo, uint32_t seed_hi,
                                               uint32_t stream_lo, uint32_t stream_hi, float (&z)[4]) {
  uint32_t c[4] = {blk, subseq, stream_lo, stream_hi};
  philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u0 = ((float)(c[2 * h] >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float u1 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float rad = sqrtf(-2.0f * logf(u0));
    const float ang = 6.283185307179586f * u1;
    z[2 * h] = rad * cosf(ang);
    z[2 * h + 1] = rad * sinf(ang);
  }
}

// grid.y = number of consecutive sub-sequences; row y gets subseq0 + y, offset y * n
__global__ void philox_normal_kernel(float* out, int64_t n, uint32_t seed_lo, uint32_t seed_hi,
                                     uint32_t stream_lo, uint32_t stream_hi, uint32_t subseq0) {
  const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blk * 4 >= n) return;
  out += (int64_t)blockIdx.y * n;
  float z[4];
  philox_normal4((uint32_t)blk, subseq0 + blockIdx.y, seed_lo, seed_hi, stream_lo, stream_hi, z);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (blk * 4 + j < n) out[blk * 4 + j] = z[j];
}

// ---------------------------------------------------------------------------
// Fused sampler update = everything in eval_step.body after the decoder calls
// (diffusion_utils.py:416-452): CFG combine, x0-from-eps, clip, posterior mean,
// + std * noise (DDPM, diffusion_utils.py:120-163,382-395) or the DDIM update
// (369-379); at scan index 0 returns x0.  Coefficients are per-step scalars from
// the load-time table (row i of `coef`, see msd_api: kCoef*).  Also advances the
// device-side scan index so a captured graph can be replayed N times.
// ---------------------------------------------------------------------------
// Row i of the table (msd_api.hip build_coef_table).  "M*" = the same conversions evaluated at the
// TRAIN schedule's log-SNR, which is where the reference converts the model output
// (diffusion_utils.py:294), while CFG / clip / the sampler use the SAMPLER schedule (:411-412).
enum { kCoefLogsnrT = 0, kCoefLogsnrS, kCoefX0Scale, kCoefX0Eps, kCoefMeanZ, kCoefMeanX0,
       kCoefStd, kCoefEpsScale, kCoefEpsX0, kCoefAlphaS, kCoefSigmaS,
       kCoefMLogsnr, kCoefMX0Scale, kCoefMX0Eps, kCoefMEpsScale, kCoefMEpsX0, kCoefMAlpha, kCoefMSigma,
       kCoefPad0, kCoefPad1, kCoefCount };
enum { kOutEps = 0, kOutX0 = 1, kOutV = 2 };   // = msd_model_output

struct SamplerParams {
  const float* eps;     // [passes][n] decoder outputs (pass 0 = conditional)
  float* z;             // [n] in/out
  const float* const* noise_slot;  // device slot holding the [N][n] per-step draws pointer; a NULL pointer there = draw here
  const uint32_t* rng_key = nullptr;   // device words {seed_lo, seed_hi, stream_lo, stream_hi} of the in-kernel draw (round 6):
                                       // step i's noise = sub-sequence 1 + i of the caller's Philox stream, element block =
                                       // this thread -- exactly the row philox_normal_kernel used to write for it (the
                                       // [N][n] buffer: 131 MB x songs at base, a hipMalloc inside msd_sample)
  const float* coef;    // [N][kCoefCount]
  int* step_ptr;        // scan index i (device); see step_from_slot1
  int step_from_slot1 = 0;
  int n;                // batch*T*n_dims
  int passes;           // 2 with CFG
  float cond_wt;        // eval_condition_weight
  int clip_x0, ddim;
  int model_output = kOutEps;   // what the network predicts (diffusion_utils.py:288-322)
  h16_t* z_hi;         // optional 16-bit planes of the new z (A operand of the next input projection)
  h16_t* z_lo;
  unsigned* sat = nullptr;   // half-plane range flag (common.h RangeCheck)
  unsigned sat_tag = 1;
};

// model output -> (eps, x0) at the TRAIN schedule's log-SNR (diffusion_utils.py:288-322)
template <int mode>
__device__ __forceinline__ void convert_model_output(const float (&c)[kCoefCount], float z, float o,
                                                     float& eps, float& x0) {
  if constexpr (mode == kOutEps) {
    eps = o;
    x0 = c[kCoefMX0Scale] * (z - o * c[kCoefMX0Eps]);
  } else if constexpr (mode == kOutX0) {
    x0 = o;
    eps = c[kCoefMEpsScale] * (z - o * c[kCoefMEpsX0]);
  } else {  // v: x0 = alpha z - sigma v
    x0 = c[kCoefMAlpha] * z - c[kCoefMSigma] * o;
    eps = c[kCoefMEpsScale] * (z - x0 * c[kCoefMEpsX0]);
  }
}

// one element of eval_step.body after the decoder calls (diffusion_utils.py:416-452):
// o_c / o_u = conditional / unconditional model output, nz = the step's normal draw
template <int MODE>
__device__ __forceinline__ float sampler_update(const SamplerParams& p, const float (&c)[kCoefCount], int i, float z,
                                                float o_c, float o_u, float nz) {
  float eps, x0;
  convert_model_output<MODE>(c, z, o_c, eps, x0);
  if (p.passes == 2) {
    float eps_u, x0_u;
    convert_model_output<MODE>(c, z, o_u, eps_u, x0_u);
    eps = p.cond_wt * eps + (1.0f - p.cond_wt) * eps_u;
    x0 = c[kCoefX0Scale] * (z - eps * c[kCoefX0Eps]);
  }
  if (p.clip_x0) {
    // jnp.clip semantics: a NaN stays a NaN (fminf / fmaxf would turn it into a bound and hide a poisoned input --
    // the half-plane range flag sees Inf but not NaN: v_max3_f32 drops NaN operands)
    x0 = x0 < -1.0f ? -1.0f : (x0 > 1.0f ? 1.0f : x0);
    eps = c[kCoefEpsScale] * (z - x0 * c[kCoefEpsX0]);
  }
  float zs;
  if (p.ddim) zs = c[kCoefAlphaS] * x0 + c[kCoefSigmaS] * eps;
  else zs = c[kCoefMeanZ] * z + c[kCoefMeanX0] * x0 + c[kCoefStd] * nz;
  return (i == 0) ? x0 : zs;
}

// MODE = p.model_output as a compile-time constant: with the run-time switch the conversions' two results travelled
// through scratch memory and a maze of scalar branches (885 lines of ISA for an elementwise kernel)
template <int MODE>
__global__ void __launch_bounds__(256) sampler_step_kernel(SamplerParams p) {
  warm_kernargs<kernarg_lines<SamplerParams>()>();
  // step_from_slot1: the step's first kernel (in_proj) copied the index to slot 1 and nobody else
  // reads slot 0 any more in this step, so this launch may decrement slot 0 itself
  const int i = p.step_from_slot1 ? p.step_ptr[1] : p.step_ptr[0];
  const int idx = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx < p.n) {
    // Loads that do not depend on the scan index go out first; the coefficient row (20 floats = five 16-byte loads,
    // rows are 80 bytes apart) and the step's noise follow as soon as the index is here.  The coefficients live in
    // REGISTERS (a pointer into the table made every c[k] of sampler_update a scalar load inside a branch, with a wait
    // each, and pushed the kernel into scratch); native vectors instead of float4 structs copied into arrays.
    const f32x4 zz = *reinterpret_cast<const f32x4*>(p.z + idx);
    const f32x4 ec = *reinterpret_cast<const f32x4*>(p.eps + idx);
    f32x4 eu = {0.f, 0.f, 0.f, 0.f}, nz = {0.f, 0.f, 0.f, 0.f};
    if (p.passes == 2) eu = *reinterpret_cast<const f32x4*>(p.eps + p.n + idx);
    static_assert(kCoefCount == 20, "five 16-byte loads per row");
    f32x4 cr[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) cr[k] = reinterpret_cast<const f32x4*>(p.coef + (size_t)i * kCoefCount)[k];
    if (!p.ddim && i != 0) {
      const float* const noise = *p.noise_slot;   // (wave-uniform: one branch)
      if (noise != nullptr) {
        nz = *reinterpret_cast<const f32x4*>(noise + (size_t)i * p.n + idx);
      } else {   // no buffer: this thread's four draws of sub-sequence 1 + i (idx / 4 = the counter block)
        const uint32_t k0 = p.rng_key[0], k1 = p.rng_key[1], s0 = p.rng_key[2], s1 = p.rng_key[3];
        float d[4];
        philox_normal4((uint32_t)(idx >> 2), 1u + (uint32_t)i, k0, k1, s0, s1, d);
        nz = f32x4{d[0], d[1], d[2], d[3]};
      }
    }
    float c[kCoefCount];
#pragma unroll
    for (int k = 0; k < kCoefCount; ++k) c[k] = cr[k >> 2][k & 3];
    f32x4 out;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = sampler_update<MODE>(p, c, i, zz[k], ec[k], eu[k], nz[k]);
    *reinterpret_cast<f32x4*>(p.z + idx) = out;
    if (p.z_hi) {
      uint32_t h[2], l[2];
      RangeCheck rc;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        rc.see(out[2 * k], out[2 * k + 1]);
        split2_h16(out[2 * k], out[2 * k + 1], h[k], l[k]);
      }
      *reinterpret_cast<uint2*>(p.z_hi + idx) = make_uint2(h[0], h[1]);
      if (p.z_lo) *reinterpret_cast<uint2*>(p.z_lo + idx) = make_uint2(l[0], l[1]);
      rc.commit(p.sat, p.sat_tag);
    }
  }
  // every block has read *step_ptr before any kernel of the next step can start
  // (kernel boundary); the decrement is ordered by the same boundary.
  __syncthreads();
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) p.step_ptr[p.step_from_slot1 ? 0 : 1] = i - 1;
}

inline void launch_sampler_step(const SamplerParams& sp, hipStream_t s) {
  const dim3 grid((sp.n / 4 + 255) / 256), block(256);
  if (sp.model_output == kOutX0) hipLaunchKernelGGL(sampler_step_kernel<kOutX0>, grid, block, 0, s, sp);
  else if (sp.model_output == kOutV) hipLaunchKernelGGL(sampler_step_kernel<kOutV>, grid, block, 0, s, sp);
  else hipLaunchKernelGGL(sampler_step_kernel<kOutEps>, grid, block, 0, s, sp);
}

// g[step][slot][k] = gamma[k] * (film_scale[step][slot][k] + 1): the column multiplier of a
// FiLM-modulated RMSNorm, tabulated for every step (folded-norm GEMM epilogues, gemm_h16.h)
__global__ void build_g_kernel(const float* film, const float* gamma, float* g, int n_steps, int slots,
                               int slot, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_steps * D) return;
  const int step = i / D, k = i % D;
  g[((size_t)step * slots + slot) * D + k] = gamma[k] * (film[((size_t)step * slots + slot) * 2 * D + k] + 1.0f);
}

// step_ptr[0] <- step_ptr[1]  (double-buffered scan index: the sampler writes the
// next index to slot 1 while other blocks of the same launch may still read slot 0)

// scale_to_features (audio_codecs.py:176-183) on the final x0
__global__ void unscale_kernel(const float* x0, float* out, int n, float fmin, float fmax) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (x0[i] + 1.0f) / 2.0f * (fmax - fmin) + fmin;
}

// scale_features(clip=True) (audio_codecs.py:166-174) on the context spectrogram
__global__ void scale_clip_kernel(const float* in, float* out, int n, float fmin, float fmax) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float f = fminf(fmaxf(in[i], fmin), fmax);
    out[i] = (f - fmin) / (fmax - fmin) * 2.0f + (-1.0f);
  }
}

// ---------------------------------------------------------------------------
// Weight packing: W fp32 [K, N] (reference layout, layers.py:430-431) ->
// W^T bf16 planes [N_out_rows, K]; dst row r takes source column src_col[r]
// (identity, QKV concatenation, or the 16-column wi_0/wi_1 interleave).
// ---------------------------------------------------------------------------
// `ldk` (0 = K) is the row length of the destination and `k0` the first column written: K-concatenated weights
// (the hoisted cross-attention query projection stacks two matrices along K).
__global__ void pack_wt_kernel(const float* w, int K, int N, h16_t* hi, h16_t* lo,
                               int dst_row0, int col_map_mode, int ldk, unsigned* absmax, int k0 = 0) {
  // grid: (ceil(K/64), N) ; block 64: thread = k within chunk
  const int n = blockIdx.y;
  const int k = blockIdx.x * 64 + threadIdx.x;
  int dst = dst_row0 + n;
  if (col_map_mode == 1) dst = dst_row0 + (n / 16) * 32 + (n % 16);        // wi_0 of the gated pair
  else if (col_map_mode == 2) dst = dst_row0 + (n / 16) * 32 + 16 + (n % 16);  // wi_1
  const float w0 = k < K ? w[(size_t)k * N + n] : 0.f;
  if (absmax) {   // largest |w| seen (bits of a non-negative float order like unsigned ints): range check at load
    float mx = fabsf(w0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (threadIdx.x == 0) atomicMax(absmax, __float_as_uint(mx));
  }
  if (k >= K) return;
  const float v = w0 * kWScale;   // undone on the accumulators (gemm_tile)
  h16_t h, l;
  split_h16(v, h, l);
  const int ld = ldk > 0 ? ldk : K;
  hi[(size_t)dst * ld + k0 + k] = h;
  if (lo) lo[(size_t)dst * ld + k0 + k] = l;
}

// out[k][n] = sum_d a[k][d] g[d] b[d][n]   (a: [K, D], b: [D, N], all fp32 row-major), accumulated in float64 and rounded
// once: Wo_self . diag(gamma_cross) . Wq of the folded cross-attention query projection (msd_api.hip, load time).
// block (16, 16): thread = (n, k)
__global__ void fold_wq_kernel(const float* a, const float* g, const float* b, float* out, int K, int D, int N) {
  const int n = blockIdx.x * 16 + threadIdx.x, k = blockIdx.y * 16 + threadIdx.y;
  if (n >= N || k >= K) return;
  double acc = 0.0;
  for (int d = 0; d < D; ++d) acc += (double)a[(size_t)k * D + d] * (double)g[d] * (double)b[(size_t)d * N + n];
  out[(size_t)k * N + n] = (float)acc;
}

// token embedding (one-hot contraction == row gather, layers.py:556-559) + position
// table rows (network.py:278-287) for the valid positions of one sequence.
//   x[r] = tok_emb[tokens[pos[r]]] + pos_emb[pos[r]]   r < n_valid ; 0 for padding rows
__global__ void embed_tokens_kernel(const int* tokens, const int* pos, int n_valid, int rows,
                                    const float* tok_emb, const float* pos_emb, float* x, int D) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float v = 0.f;
    if (r < n_valid) {
      const int ps = pos[r];
      v = tok_emb[(size_t)tokens[ps] * D + c] + pos_emb[(size_t)ps * D + c];
    }
    x[(size_t)r * D + c] = v;
  }
}

// dst[r] = (r < n_valid) ? src[idx[r]] : 0   (row gather with zero padding)
__global__ void gather_rows_kernel(const float* src, const int* idx, int n_valid, float* dst, int D) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x)
    dst[(size_t)r * D + c] = (r < n_valid) ? src[(size_t)idx[r] * D + c] : 0.f;
}

// fp32 -> bf16 planes (test helper for the standalone ops)
__global__ void split_planes_kernel(const float* in, h16_t* hi, h16_t* lo, int64_t n, unsigned* sat = nullptr,
                                    unsigned sat_tag = 1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    h16_t h, l;
    RangeCheck rc;
    rc.see(in[i]);
    split_h16(in[i], h, l);
    hi[i] = h;
    if (lo) lo[i] = l;
    rc.commit(sat, sat_tag);
  }
}
__global__ void merge_planes_kernel(const h16_t* hi, const h16_t* lo, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = h2f(hi[i]) + (lo ? h2f(lo[i]) : 0.f);
}

}  // namespace msd
