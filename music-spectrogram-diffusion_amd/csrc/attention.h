// Unscaled dot-product attention, head_dim = 64, for the three attention shapes
// of the path (layers.py:109-181: softmax(q.k^T + bias) v, NO 1/sqrt(d) scaling,
// layers.py:254-258):
//   decoder self-attention   256 x 256, unmasked            (network.py:181-189)
//   decoder cross-attention  256 x S_valid (<= 2304), key-padding mask realised by
//                            dropping padded keys             (network.py:217-235)
//   encoder self-attention   L_valid x L_valid                (network.py:131-137)
// Keys beyond `n_keys` (padding up to a multiple of 32) get -1e30; a masked key has
// weight exactly 0 in the reference too (exp(-1e10 + s - max) == 0 in fp32,
// layers.py:341-346).  n_keys == 0 -> output 0 (layers.py:882-902).
//
// Mapping: block = (32 query rows, one head, one segment), NW waves split the
// key blocks round-robin; each wave runs online softmax over its 32-key blocks:
//   S^T[key][q]  = K_blk . Q^T        v_mfma_f32_32x32x16_bf16, A = K rows, B = Q rows
//   O^T[d][q]   += V^T_blk . P^T      A = V^T rows (key axis pre-permuted), B = P^T
// In the 32x32 C layout a lane owns ONE query column (q = lane&31) and 16 of the
// 32 keys, so max/sum/rescale are lane-local plus one lane^32 exchange, and the
// P^T registers are already the B fragment of the second MFMA -- no LDS, no
// cross-lane shuffles of P.  K, Q, V^T fragments are 16-byte global loads
// (L2-resident: K/V of a layer are <= 7 MB).  Partial (m, l, O) of the NW waves
// are merged through LDS.  NP = 2 runs every product as hi.hi + hi.lo + lo.hi.
#pragma once
#include "common.h"
#include "gemm_bf16.h"

namespace msd {

struct AttnParams {
  const bf16_t* q[2];   // [rows, ldq] row-major, head h at column h*64
  const bf16_t* k[2];   // [seg][keys, ldk]
  const bf16_t* vt[2];  // [seg][heads*64][vt_ld], key axis permuted per 16
  bf16_t* o[2];         // [rows, ldo]
  const int* n_keys;    // [n_segs] valid keys per segment (device)
  int ldq, ldk, ldo, vt_ld;
  int q_rows_per_seg;   // query rows per segment
  size_t k_seg_stride;  // elements between segments of k
  size_t vt_seg_stride; // elements between segments of vt
};

typedef __attribute__((ext_vector_type(8))) __bf16 frag8;

__device__ __forceinline__ frag8 ld_frag(const bf16_t* p) {
  return as_frag(*reinterpret_cast<const uint4*>(p));
}

template <int NP, int NW>
__global__ void __launch_bounds__(NW * 64) attention_kernel(AttnParams p) {
  constexpr float NEG = -1e30f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qb = blockIdx.x, head = blockIdx.y, seg = blockIdx.z;
  const int q_lane = lane & 31, hi = lane >> 5;
  const int nkeys = p.n_keys[seg];
  const int nkb = (nkeys + 31) >> 5;

  const size_t qrow = (size_t)seg * p.q_rows_per_seg + qb * 32 + q_lane;
  const bf16_t* kbase[NP];
  const bf16_t* vbase[NP];
  frag8 qf[NP][4];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    const bf16_t* qp = p.q[pl] + qrow * p.ldq + head * 64 + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[pl][s] = ld_frag(qp + s * 16);
    kbase[pl] = p.k[pl] + (size_t)seg * p.k_seg_stride + head * 64 + hi * 8;
    vbase[pl] = p.vt[pl] + (size_t)seg * p.vt_seg_stride + (size_t)(head * 64 + q_lane) * p.vt_ld + hi * 8;
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = NEG, l_run = 0.f;

  for (int kb = wave; kb < nkb; kb += NW) {
    // ---- S^T = K . Q^T ----------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    frag8 kf[NP][4];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const bf16_t* kp = kbase[pl] + (size_t)(kb * 32 + q_lane) * p.ldk;
#pragma unroll
      for (int st = 0; st < 4; ++st) kf[pl][st] = ld_frag(kp + st * 16);
    }
    // V^T fragments issued early so their latency hides under QK^T + softmax
    frag8 vf[NP][2][2];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          vf[pl][db][ks] = ld_frag(vbase[pl] + (size_t)db * 32 * p.vt_ld + kb * 32 + ks * 16);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][st], qf[0][st], s, 0, 0, 0);
      if (NP == 2) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][st], qf[NP - 1][st], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[NP - 1][st], qf[0][st], s, 0, 0, 0);
      }
    }
    // lane owns keys kb*32 + (r&3) + 8*(r>>2) + 4*hi for r = 0..15
    float bmax = NEG;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= nkeys) s[r] = NEG;
      bmax = fmaxf(bmax, s[r]);
    }
    bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
    const float m_new = fmaxf(m_run, bmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __expf(s[r] - m_new);
      psum += pv[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // ---- P^T fragments (k-slot j of step ks <-> register 8*ks + j) ----------
    frag8 pf[NP][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint32_t wh[4], wl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16_t h0, l0, h1, l1;
        if (NP == 2) {
          split_bf16(pv[8 * ks + 2 * j], h0, l0);
          split_bf16(pv[8 * ks + 2 * j + 1], h1, l1);
          wl[j] = pack2(l0, l1);
        } else {
          h0 = f2bf(pv[8 * ks + 2 * j]);
          h1 = f2bf(pv[8 * ks + 2 * j + 1]);
        }
        wh[j] = pack2(h0, h1);
      }
      pf[0][ks] = as_frag(make_uint4(wh[0], wh[1], wh[2], wh[3]));
      if (NP == 2) pf[NP - 1][ks] = as_frag(make_uint4(wl[0], wl[1], wl[2], wl[3]));
    }
    // ---- O^T += V^T . P^T ---------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0][ks], pf[0][ks], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][1][ks], pf[0][ks], o1, 0, 0, 0);
      if (NP == 2) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0][ks], pf[NP - 1][ks], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][1][ks], pf[NP - 1][ks], o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[NP - 1][0][ks], pf[0][ks], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[NP - 1][1][ks], pf[0][ks], o1, 0, 0, 0);
      }
    }
  }
  // full-row sum: combine the two half-lanes that share a query
  l_run += __shfl_xor(l_run, 32, 64);

  // ---- merge the NW partial results through LDS ------------------------------
  // layout per wave: O [32 q][64 d + 4 pad] fp32, then m[32], l[32]
  constexpr int OLD = 68;
  constexpr int WSTRIDE = 32 * OLD + 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm = reinterpret_cast<float*>(smem_raw);
  float* mine = sm + wave * WSTRIDE;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    // registers 4g..4g+3 are d = 8g + 4hi + 0..3 (rows of the 32x32 C tile)
    const int d = 8 * g + 4 * hi;
    *reinterpret_cast<float4*>(mine + q_lane * OLD + d) =
        make_float4(o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
    *reinterpret_cast<float4*>(mine + q_lane * OLD + 32 + d) =
        make_float4(o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
  }
  if (hi == 0) {
    mine[32 * OLD + q_lane] = m_run;
    mine[32 * OLD + 32 + q_lane] = l_run;
  }
  __syncthreads();
  // 32 q x 8 groups of 8 d = 256 work items
  for (int item = threadIdx.x; item < 256; item += NW * 64) {
    const int q = item >> 3, d0 = (item & 7) * 8;
    float mt = NEG;
#pragma unroll
    for (int w = 0; w < NW; ++w) mt = fmaxf(mt, sm[w * WSTRIDE + 32 * OLD + q]);
    float lt = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* ws = sm + w * WSTRIDE;
      const float f = __expf(ws[32 * OLD + q] - mt);
      lt += ws[32 * OLD + 32 + q] * f;
      const float4 a = *reinterpret_cast<const float4*>(ws + q * OLD + d0);
      const float4 b = *reinterpret_cast<const float4*>(ws + q * OLD + d0 + 4);
      acc[0] += a.x * f; acc[1] += a.y * f; acc[2] += a.z * f; acc[3] += a.w * f;
      acc[4] += b.x * f; acc[5] += b.y * f; acc[6] += b.z * f; acc[7] += b.w * f;
    }
    const float inv = (lt > 0.f) ? 1.0f / lt : 0.f;
    const size_t off = ((size_t)seg * p.q_rows_per_seg + qb * 32 + q) * p.ldo + head * 64 + d0;
    uint32_t wh[4], wl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf16_t h0, l0, h1, l1;
      const float v0 = acc[2 * e] * inv, v1 = acc[2 * e + 1] * inv;
      if (NP == 2) {
        split_bf16(v0, h0, l0);
        split_bf16(v1, h1, l1);
        wl[e] = pack2(l0, l1);
      } else {
        h0 = f2bf(v0);
        h1 = f2bf(v1);
      }
      wh[e] = pack2(h0, h1);
    }
    *reinterpret_cast<uint4*>(p.o[0] + off) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
    if (NP == 2) *reinterpret_cast<uint4*>(p.o[NP - 1] + off) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
  }
}

template <int NP, int NW>
inline hipError_t launch_attention(const AttnParams& p, int q_blocks, int heads, int segs,
                                   hipStream_t stream) {
  // (> 64 KiB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize, set once in
  //  msd_api.hip:set_func_attrs -- never during stream capture)
  constexpr int smem = NW * (32 * 68 + 64) * 4;
  auto kern = attention_kernel<NP, NW>;
  hipLaunchKernelGGL(kern, dim3(q_blocks, heads, segs), dim3(NW * 64), smem, stream, p);
  return hipGetLastError();
}

}  // namespace msd
