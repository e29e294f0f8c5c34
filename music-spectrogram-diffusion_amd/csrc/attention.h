// Unscaled dot-product attention, head_dim = 64, for the three attention shapes
// of the path (layers.py:109-181: softmax(q.k^T + bias) v, NO 1/sqrt(d) scaling,
// layers.py:254-258):
//   decoder self-attention   256 x 256, unmasked            (network.py:181-189)
//   decoder cross-attention  256 x S_valid (<= 2304), key-padding mask realised by
//                            dropping padded keys             (network.py:217-235)
//   encoder self-attention   L_valid x L_valid                (network.py:131-137)
// Keys beyond `n_keys` get -1e30; a masked key has weight exactly 0 in the reference
// too (exp(-1e10 + s - max) == 0 in fp32, layers.py:341-346).  n_keys == 0 -> output 0
// (layers.py:882-902).
//
// Mapping.  Block = (64 query rows, one head, one segment) = 8 waves = 2 query
// blocks x 4 key groups.  K and V^T are streamed through an LDS ring in stages of
// 128 keys by LDS-DMA (global_load_lds, full 128/256-byte rows: the per-CU vector
// memory path moves 43 B/clk for whole lines but ~17 B/clk for MFMA-fragment-shaped
// gathers, tools/ubench/load_patterns.hip), one raw barrier per stage, counted
// s_waitcnt vmcnt(N).  Wave (qb, kg) runs online softmax over key block kg of every
// stage for query block qb:
//   S^T[key][q]  = K_blk . Q^T        v_mfma_f32_32x32x16_bf16, A = K rows, B = Q rows
//   O^T[d][q]   += V^T_blk . P^T      A = V^T rows (key axis pre-permuted), B = P^T
// In the 32x32 C layout a lane owns ONE query column (q = lane&31) and 16 of the
// 32 keys, so max/sum/rescale are lane-local plus one lane^32 exchange, and the
// P^T registers are already the B fragment of the second MFMA (that is what the
// per-16 key permutation of V^T buys) -- no shuffles of P.  The (m, l, O) partials
// of the 4 key groups are merged through LDS.  NP = 2 runs every product as
// hi.hi + hi.lo + lo.hi.  Both LDS tiles are XOR-swizzled on the 16-byte chunk
// index; since the DMA destination is lane-linear, the swizzle is applied to the
// per-lane SOURCE address.
#pragma once
#include "common.h"
#include "gemm_h16.h"

namespace msd {

struct AttnParams {
  const h16_t* q[2];   // [rows, ldq] row-major, head h at column h*64
  const h16_t* k[2];   // [seg][keys, ldk]
  const h16_t* vt[2];  // [seg][heads*64][vt_ld], key axis permuted per 16
  h16_t* o[2];         // [rows, ldo]
  const int* n_keys;    // [n_segs] valid keys per segment (device)
  int ldq, ldk, ldo, vt_ld;
  int q_rows_per_seg;   // query rows per segment (multiple of 64)
  size_t k_seg_stride;  // elements between segments of k
  size_t vt_seg_stride; // elements between segments of vt
  int k_rows;           // allocated key rows per segment (DMA source rows are clamped to it)
  int vt_cols = 0;      // readable key columns of a V^T row from `vt` (0 = vt_ld); smaller when `vt` points into a row
  // key split across blocks (cross-attention: more CUs for the long key axis).  ksplit > 1:
  // block (q-group, ks) takes stages ks, ks+ksplit, ... and writes its un-normalised partial
  // (O relative to its own max m, and m, l) to the workspace; attention_merge_kernel finishes.
  int ksplit;
  int ksplit_log2 = 0;  // log2(ksplit); ksplit is a power of two (launch_attention rounds it down): shifts instead of the
                        // three software integer divisions (~25 dependent instructions each) the kernel ran at its entry
  float* part_o;        // [ksplit][rows][heads*64]
  float* part_ml;       // [ksplit][rows][heads][2]
  int total_rows;       // rows of q over all segments
  WeightPrefetch pf;    // optional: warm a later GEMM's weights in this XCD's L2 (gemm_h16.h)
  unsigned* sat = nullptr;   // half-plane range flag of the handle (common.h RangeCheck)
  unsigned sat_tag = 1;
  int qp = 0;                // query-side single-plane switches (attention_kernel QP bits 0 / 1), NP = 2 only
  int allow_qb4 = 0;         // host only: this launch may run on 128-row blocks (attention_query_blocks)
  int touch_ahead = 0;       // > 0 (launches with a prefetch wave only): that wave also touches the K / V^T lines of the
                             // block's ring stages this many stages AHEAD of their LDS-DMA (kv_touch_ahead below)
  int pf_late = 0;           // touch-ahead launches only (round 6): the prefetch wave issues NOTHING at entry -- its touches of
                             // the launch's weight target and of the ring stages NS .. NS + ahead - 1 go out behind the
                             // first key-loop barrier (stage 0 has landed).  At entry they competed with every block's
                             // first two K / V^T stages (192 x 128 KiB) for the same HBM burst: the launch 14.2 -> 13.5 us
                             // at 1136 cold keys for the weights alone (tools/ubench/attn_cold.hip); step, same process:
                             // weights late -0.9 ... -1.2 %, the stage touches late another -0.5 ... -1.1 % from ~1150 keys,
                             // 0 at 557 (profiles/r06q_pf_place_ab*.log).  The weights still arrive a launch ahead of
                             // their GEMM.  Bit-identical (a touch has no reader).
  int* tickets = nullptr;    // ksplit > 1, in-launch merge (round 6, attention_inlaunch_merge below): one arrival counter per
                             // (segment, query block of this launch, head), all zero between launches; nullptr = the
                             // separate attention_merge_kernel launch finishes the split
  // UN-NORMALISED queries (the folded cross-attention query projection, gemm_h16.h): `q` holds (x (.) gamma) . Wq without
  // the 1/rms of the pre-attention RMSNorm; the kernel scales the logits of query row r by
  // rstd[r] = rsqrt(sum_t q_ssq[r][t] * q_inv_d + 1e-6) -- the algebra of the GEMMs' folded norms (QP bit 2)
  const float* q_ssq = nullptr;   // [rows][q_tiles] partial sums of squares of the residual stream
  int q_tiles = 0;                // <= kAuxMaxTiles
  float q_inv_d = 0.f;
};

typedef __attribute__((ext_vector_type(8))) plane_elem frag8;

__device__ __forceinline__ frag8 ld_frag(const h16_t* p) {
  return as_frag(*reinterpret_cast<const uint4*>(p));
}

constexpr int kAttKG = 4;   // key groups (waves along the 128 keys of a stage); QB query blocks of 32 rows -> QB * 4 waves
constexpr int kAttStageKeys = kAttKG * 32;             // 128 keys per LDS stage
constexpr int kAttKBytes = kAttStageKeys * 128;        // K tile  [128 keys][64 d] bf16
constexpr int kAttVBytes = 64 * kAttStageKeys * 2;     // V^T tile [64 d][128 keys] bf16
constexpr int kAttOLD = 68;                            // merge slab row stride (floats)
constexpr int kAttWStride = 32 * kAttOLD + 64;         // per-wave merge slab (floats)

constexpr int kAttQBytes = 32 * 128;                   // Q tile of one query block and plane: [32 rows][64 d], behind the ring

// ring + Q tiles (reused as the merge slab), then 256 bytes nobody reads: where the touch-ahead's LDS-DMA lands
constexpr int kAttTouchSink = 256;
template <int NP, int NS, int QB>
constexpr int attention_work_smem() {
  return (NS * NP * (kAttKBytes + kAttVBytes) + QB * NP * kAttQBytes > QB * kAttKG * kAttWStride * 4)
             ? NS * NP * (kAttKBytes + kAttVBytes) + QB * NP * kAttQBytes
             : QB * kAttKG * kAttWStride * 4;
}
// (QS: the launches with un-normalised queries also stage the block's rows of q_ssq -- QB x 32 rows x kAuxMaxTiles floats)
constexpr int kAttSsqBytes = 32 * kAuxMaxTiles * 4;   // per query block
template <int NP, int NS, int QB, bool QS = false>
constexpr int attention_smem() { return attention_work_smem<NP, NS, QB>() + (QB < 4 ? kAttTouchSink : 0) + (QS ? QB * kAttSsqBytes : 0); }   // (QB = 4: the 160 KiB are full; no prefetch wave there)

// Query blocks per workgroup for a launch of `blocks64` 64-row blocks (heads x query tiles x key splits x segments):
//   < 128 blocks           -> 32-row blocks (QB = 1): twice as many, when most of the 256 CUs would be idle
//   one round of the chip  -> 64-row blocks (QB = 2): K / V traffic halves
//   more than one round    -> 128-row blocks (QB = 4, 16 waves; round 5): the batched cross-attention is latency-bound on
//                             its two-stage ring (2.5 - 3 us per stage, whatever the rows), so 12 x 4 x songs blocks in 1.5
//                             rounds (8 songs: 384) cost 1.5 block lifetimes where 192 blocks of 128 rows cost one; no
//                             prefetch wave fits beside 16 waves -- the host gives the launch's weight target to a neighbour
__host__ __device__ inline int attention_query_blocks(int blocks64, int q_rows_per_seg, int np) {
  if (blocks64 < 128) return 1;
  if (np == 2 && blocks64 > 256 && q_rows_per_seg % 128 == 0) return 4;
  return 2;
}

// K / V^T touch-ahead (round 5), run by the block's PREFETCH WAVE.  The cached cross-attention K / V^T are HBM-cold at every
// step and the ring holds NS = 2 stages of 64 KiB: a block waits one full HBM latency (3.5 us: the
// "issued -> stage 0 landed" of the phase stamps) for its first stage and AGAIN for every stage it issues later -- its
// key loop runs at 2.5 us per stage where the MFMAs need 0.65 (profiles/r04z_phase_times*.txt; at 8 songs per handle a
// block walks 11 stages: 62 of a layer's 320 us).  More ring does not fit the LDS; the L2 does hold a few stages per
// block.  So the prefetch wave -- whose vmcnt nobody waits for -- touches one dword per 128-byte line of the stages
// `ahead` (> 0) stages in front of the LDS-DMA front: stages NS .. NS + ahead - 1 first (at entry in round 5; behind the
// first key-loop barrier since round 6, AttnParams::pf_late), then one more stage per key-loop barrier, which it takes part
// in (an alive wave counts at s_barrier: it executes exactly the compute waves' barriers --
// one per stage of this block + the two of the merge -- and ends).  The DMA of such a stage then finds its lines in the
// XCD's L2 / the memory-side cache.  Same bytes from HBM, earlier; results are untouched (nobody reads a touch).
// A touch is ONE DWORD PER LANE BY LDS-DMA into 256 bytes of LDS nobody reads (`sink_lds`, behind the block's working
// set): no destination register, so none of the register hazards of the weight touches (gemm_h16.h prefetch_wave: a
// touch into a VGPR inside this run-time loop had its register reused by the compiler -- check_prefetch_regs.py caught
// it on the listing).
template <int NP, int NS, int PF>
__device__ __forceinline__ void kv_touch_ahead(const AttnParams& p, char* sink_lds) {
  const int lane = (int)threadIdx.x & 63;
  const int kl2 = p.ksplit_log2;
  const int ks = (int)(blockIdx.y & (p.ksplit - 1)), head = blockIdx.x, seg = blockIdx.z;
  int nkeys;
  {
    const int* nkp = p.n_keys + seg;
    asm volatile("s_nop 4\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(nkeys) : "s"(nkp) : "memory");
  }
  const int nst_all = (nkeys + kAttStageKeys - 1) / kAttStageKeys;
  const int nst = nst_all > ks ? (nst_all - ks + p.ksplit - 1) >> kl2 : 0;   // (the compute waves' count, to the letter)
  const int ahead = p.touch_ahead;
  const int last_row = p.k_rows - 1, last_kcol = (p.vt_cols > 0 ? p.vt_cols : p.vt_ld) - 8;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // one stage = per plane 128 K rows of 128 bytes + 64 V^T rows of 256 bytes = 256 lines: 4 wave-wide touches per plane
  auto touch_stage = [&](int st) {
    const int kb = (ks + st * p.ksplit) * kAttStageKeys;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tt = u * 64 + lane;
        const h16_t* src;
        if (u < 2) {           // K rows kb + tt
          int row = kb + tt;
          row = row < last_row ? row : last_row;
          src = p.k[pl] + (size_t)seg * p.k_seg_stride + (size_t)row * p.ldk + head * 64;
        } else {               // V^T rows d = (tt - 128) / 2, half = tt & 1 (64 keys = 128 bytes each)
          const int d = (tt - 128) >> 1;
          int col = kb + (tt & 1) * 64;
          col = col < last_kcol ? col : last_kcol;
          src = p.vt[pl] + (size_t)seg * p.vt_seg_stride + (size_t)(head * 64 + d) * p.vt_ld + col;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)sink_lds, 4, 0, 0);
      }
    }
  };
  if (!p.pf_late) for (int st = NS; st < NS + ahead && st < nst; ++st) touch_stage(st);
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  bool weights_done = false;
  if (!p.pf_late || nst == 0) {   // the launch's weight target (a later GEMM's planes) at entry, as before
    prefetch_wave<PF, true>(p.pf, lin, gridDim.x * gridDim.y * gridDim.z, p.q[0], sink_lds);
    weights_done = true;
  }
  {
    for (int st = 0; st < nst; ++st) {
      __builtin_amdgcn_s_barrier();                 // the compute waves' barrier of stage st
      if (st == 0 && p.pf_late) for (int s2 = NS; s2 < NS + ahead && s2 < nst; ++s2) touch_stage(s2);
      if (!weights_done) {                          // (pf_late) stage 0 is in LDS: the first K / V^T burst is over
        prefetch_wave<PF, true>(p.pf, lin, gridDim.x * gridDim.y * gridDim.z, p.q[0], sink_lds);
        weights_done = true;
      }
      if (st + NS + ahead < nst) touch_stage(st + NS + ahead);
    }
    __builtin_amdgcn_s_barrier();                   // ... and the two of their partial merge
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA of this wave outlives it (the LDS goes back with the block)
  }
}

// Finish one (row, head, 8-wide d group) of a key-split attention from the KS partials (m, l) and O (relative to its own
// m): out = sum_ks O_ks e^(m_ks - m) / sum_ks l_ks e^(m_ks - m), sums in split order.  ONE function for the separate merge
// launch and for the in-launch merge below, so that both write the same bits.
template <int KS>
__device__ __forceinline__ void merge_split_partials(const f32x2 (&ml)[KS], const f32x4 (&oa)[KS], const f32x4 (&ob)[KS],
                                                     float (&v)[8]) {
#pragma clang fp contract(off)   // (see attention_kernel)
  float mt = -1e30f, lt = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) mt = fmaxf(mt, ml[ks][0]);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float f = fast_exp(ml[ks][0] - mt);
    lt = __builtin_fmaf(ml[ks][1], f, lt);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e] = __builtin_fmaf(oa[ks][e], f, acc[e]); acc[4 + e] = __builtin_fmaf(ob[ks][e], f, acc[4 + e]); }
  }
  const float inv = (lt > 0.f) ? 1.0f / lt : 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = acc[e] * inv;
}

// In-launch merge of a key-split attention (round 6; VERDICT r05 next #1b): the separate merge launch cost 4.8 us per layer
// for 1.2 us of work -- a kernel boundary.  Here every block of a (segment, query block, head) group publishes its partial
// WRITE-THROUGH (sc1 stores: the bytes are at the memory side, visible to every XCD, once the storing wave's vmcnt has
// drained -- no release fence, which would write back the XCD's whole L2), takes a ticket on the group's counter
// (relaxed, agent scope), and the block that draws the LAST ticket reads all KS partials back with sc1 loads (they bypass
// this CU's L1 and are served coherently: no acquire fence), merges them with the function above and stores the planes.
// MI355X guide, Guideline 16 recipe R1 / the split-K reducer: {sc1 payload -> every storing wave vmcnt(0) -> barrier ->
// one lane's ticket}, reducer: sc1 loads.  Placement-independent: nothing here depends on which XCD a block runs on.
// Round 3's attempt (+10 ... 50 us per launch, docs/history.md) used __threadfence() on both sides: buffer_wbl2 +
// buffer_inv in every block.  The counter is zero between launches: the reducer resets it (msd_sample zeroes the array
// once at its start, in case an aborted launch left a count behind).  `slot` = 4 bytes of LDS nobody else uses by now.
template <int NP, int KS>
__device__ __forceinline__ void attention_inlaunch_merge(const AttnParams& p, int ks, size_t row, int head, int heads, int d0,
                                                         const float (&acc)[8], float mt, float lt, int group, int tid,
                                                         int* slot) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  const size_t rows = (size_t)p.total_rows;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.part_o, 0, (int)((size_t)p.ksplit * rows * heads * 64 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(p.part_ml, 0, (int)((size_t)p.ksplit * rows * heads * 2 * 4), 0x00020000);
  const int item = (int)(row * heads + head);          // (row, head) of this thread, as the merge launch indexes them
  const int stride = (int)(rows * heads);              // items per split
  {
    const int base = ks * stride + item;
    const u4 a = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
    const u4 b = {__float_as_uint(acc[4]), __float_as_uint(acc[5]), __float_as_uint(acc[6]), __float_as_uint(acc[7])};
    __builtin_amdgcn_raw_buffer_store_b128(a, ro, (base * 64 + d0) * 4, 0, /*sc1*/ 16);
    __builtin_amdgcn_raw_buffer_store_b128(b, ro, (base * 64 + d0 + 4) * 4, 0, 16);
    if (d0 == 0) {
      const u2 m2 = {__float_as_uint(mt), __float_as_uint(lt)};
      __builtin_amdgcn_raw_buffer_store_b64(m2, rm, base * 8, 0, 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) ; msd publish drain" ::: "memory");   // EVERY storing wave (guide G16 pitfall 14)
  __syncthreads();
  if (tid == 0) *slot = __hip_atomic_fetch_add(p.tickets + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*slot != KS - 1) return;                         // (block-uniform) not the last arriver: done
  f32x2 ml[KS];
  f32x4 oa[KS], ob[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int base = k * stride + item;
    const u2 m2 = __builtin_amdgcn_raw_buffer_load_b64(rm, base * 8, 0, 16);
    const u4 a = __builtin_amdgcn_raw_buffer_load_b128(ro, (base * 64 + d0) * 4, 0, 16);
    const u4 b = __builtin_amdgcn_raw_buffer_load_b128(ro, (base * 64 + d0 + 4) * 4, 0, 16);
    ml[k] = f32x2{__uint_as_float(m2[0]), __uint_as_float(m2[1])};
    oa[k] = f32x4{__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3])};
    ob[k] = f32x4{__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
  }
  float v[8];
  merge_split_partials<KS>(ml, oa, ob, v);
  RangeCheck rc;
  store_h16x8<NP>(p.o, row * p.ldo + head * 64 + d0, v, rc);
  rc.commit(p.sat, p.sat_tag);
  if (tid == 0) __hip_atomic_store(p.tickets + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next launch
}

// QB query blocks of 32 rows per workgroup (attention_query_blocks above): QB = 2: 64 query rows share each K/V stage;
// QB = 4: 128 rows, 16 waves, no prefetch wave (batched decoder attentions);
// QB = 1: 32 query rows, twice the blocks -- for launches that would otherwise leave most CUs idle
// (decoder self-attention at B = 1: 12 heads x 4 x 2 passes = 96 blocks of 64 rows).
// QP: query-side single-plane switches of the NP = 2 modes (bit 0: Q enters S = K.Q^T as ONE plane, bit 1: P enters
// O += V^T.P^T as one plane): 2 instead of 3 MFMAs for that product, and no hi / lo split of P.  The memory side
// (K, V) always keeps both planes (dropping those costs 20 - 50x the error).  0 = all three products.
// Bit 2 (QS): the queries are un-normalised, see AttnParams::q_ssq.
template <int NP, int NS, int QB, int PF = kPfNone, int QP = 0>
__global__ void __launch_bounds__(QB * kAttKG * 64 + (kPfWaveAttn && PF != kPfNone ? 64 : 0)) attention_kernel(AttnParams p) {
  // Every multiply-add of this kernel that is MEANT to be fused is an explicit fma, and nothing else may be: left to
  // -ffp-contract=fast the instantiations contracted differently (round 6: <QB = 2, PF = 1> fused four packed
  // multiply-adds fewer than <QB = 1, PF = 1> and the PF = 0 pair -- the last bit of a segment then depended on which
  // block shape the launcher picked, e.g. on dedup_layer0 at two songs: tools/diag/bitwise_matrix.py)
#pragma clang fp contract(off)
  static_assert(QB < 4 || PF == kPfNone, "16 compute waves fill the workgroup: no prefetch wave");
  warm_kernargs<kernarg_lines<AttnParams>()>();
  if constexpr (kPfWaveAttn && PF != kPfNone) {
    if (threadIdx.x >= QB * kAttKG * 64) {   // the prefetch wave: a later GEMM's weights (+ this block's later K / V^T stages)
      if (p.touch_ahead <= 0) {   // weights only, and the wave ends (an ended wave does not count at s_barrier)
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        prefetch_wave<PF>(p.pf, lin, gridDim.x * gridDim.y * gridDim.z, p.q[0]);
        return;
      }
      extern __shared__ __attribute__((aligned(16))) char smem_pf[];
      kv_touch_ahead<NP, NS, PF>(p, smem_pf + attention_work_smem<NP, NS, QB>());
      return;
    }
  }
  constexpr bool Q1 = NP == 2 && (QP & 1), P1 = NP == 2 && (QP & 2), QS = (QP & 4) != 0;
  static_assert(!QS || QB < 4, "un-normalised queries: 32- and 64-row blocks only (the 128-row block fills the LDS)");
#if MSD_TIMESTAMPS
  const int ts_blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  constexpr int ts_cls = QB == 1 ? 4 : 5;
#endif
  MSD_TS_BEGIN(ts_cls, ts_blk)
  constexpr int kRows = 32 * QB;                 // query rows per block
  constexpr int JPW = 16 / (QB * kAttKG);        // K (and V^T) DMA instructions per wave, plane and stage
  constexpr float NEG = -1e30f;
  constexpr int STAGE = NP * (kAttKBytes + kAttVBytes);
  constexpr int PW = 2 * JPW * NP;  // DMA instructions per wave per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wave / kAttKG, kg = wave % kAttKG;
  // head is the fastest grid axis: the blocks that share one head's K/V land on few XCDs
  const int kl2 = p.ksplit_log2;
  const int blk = (int)(blockIdx.y >> kl2), ks = (int)(blockIdx.y & (p.ksplit - 1));
  const int head = blockIdx.x, seg = blockIdx.z;
  const int q_lane = lane & 31, hi = lane >> 5;

  // ---- DMA source addressing -------------------------------------------------------
  // K tile: instruction j (0..15) moves keys 8j..8j+7 of the stage, lane = (r = lane>>3, c' = lane&7),
  //         source chunk c' ^ r (rows are 128 B = 8 chunks).  This wave issues j = 2*wave, 2*wave+1.
  // V^T   : instruction j (0..15) moves rows d = 4j..4j+3, lane = (r = lane>>4, c' = lane&15),
  //         source chunk c' ^ (d & 15) (rows are 256 B = 16 chunks).
  const int kr = lane >> 3, kc = (lane & 7) ^ kr;
  const int vr = lane >> 4;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const h16_t* kseg[NP];
  const h16_t* vseg[NP];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    kseg[pl] = p.k[pl] + (size_t)seg * p.k_seg_stride + head * 64 + kc * 8;
    vseg[pl] = p.vt[pl] + (size_t)seg * p.vt_seg_stride + (size_t)(head * 64) * p.vt_ld;
  }
  const int last_row = p.k_rows - 1, last_kcol = (p.vt_cols > 0 ? p.vt_cols : p.vt_ld) - 8;

#define MSD_A_ISSUE(ST, BUF)                                                                      \
  {                                                                                               \
    char* base_ = smem + (BUF) * STAGE;                                                           \
    const int kb_ = (ks + (ST) * p.ksplit) * kAttStageKeys;                                       \
    _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                           \
      _Pragma("unroll") for (int jj = 0; jj < JPW; ++jj) {                                        \
        const int j_ = JPW * wave + jj;                                                           \
        int row_ = kb_ + 8 * j_ + kr;                                                             \
        row_ = row_ < last_row ? row_ : last_row;                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t)(kseg[pl] + (size_t)row_ * p.ldk),               \
            (lptr_t)(base_ + pl * kAttKBytes + j_ * 1024), 16, 0, 0);                             \
        const int d_ = 4 * j_ + vr;                                                               \
        int col_ = kb_ + (((lane & 15) ^ (d_ & 15)) << 3);                                        \
        col_ = col_ < last_kcol ? col_ : last_kcol;                                               \
        __builtin_amdgcn_global_load_lds((gptr_t)(vseg[pl] + (size_t)d_ * p.vt_ld + col_),        \
            (lptr_t)(base_ + NP * kAttKBytes + pl * kAttVBytes + j_ * 1024), 16, 0, 0);           \
      }                                                                                           \
    }                                                                                             \
  }

  // ---- Q tile (B operand of S^T = K.Q^T): LDS-DMA into a region behind the ring, issued BEFORE the ring stages so that
  // the counted vmcnt waits below see it as the oldest operation.  (Round 3 loaded the Q fragments straight into
  // registers: ordinary loads beside LDS-DMA make hipcc wait with vmcnt(0) at their first use -- the first S^T MFMA --
  // i.e. for EVERY prologue stage instead of stage 0; with self-attention's two stages that serialised landing and
  // compute.)  Layout and source swizzle of the K tile: row = query, 8 chunks of 16 bytes, chunk ^ (row & 7).
  const size_t qrow0 = (size_t)seg * p.q_rows_per_seg + blk * kRows + qb * 32;
  const size_t qrow = qrow0 + q_lane;
  constexpr int NPQ = Q1 ? 1 : NP;                     // planes of Q the products use
  constexpr int QOFF = NS * STAGE;
  // QS: the partial sums of squares of the block's query rows (kRows x q_tiles floats, contiguous) -> LDS behind the
  // touch sink, by LDS-DMA like the Q tile and in front of it: the oldest operations of every wave's vmcnt queue, so
  // the counted wait of stage 0 covers them.  (Round 3's hoist read them with ordinary loads into registers: beside
  // LDS-DMA that is a vmcnt(0) at their first use -- the cross-attention launch grew by 1.7 us, docs/history.md.)
  constexpr int SSQOFF = attention_work_smem<NP, NS, QB>() + kAttTouchSink;
  if constexpr (QS) {
    const int bytes = kRows * p.q_tiles * 4;
    if (wave * 1024 < bytes) {   // at most one 1 KiB instruction per wave (q_tiles <= kAuxMaxTiles)
      int off = wave * 1024 + lane * 16;
      off = off < bytes - 16 ? off : bytes - 16;
      const char* src = reinterpret_cast<const char*>(p.q_ssq + ((size_t)seg * p.q_rows_per_seg + blk * kRows) * p.q_tiles) + off;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + SSQOFF + wave * 1024), 16, 0, 0);
    }
  }
#pragma unroll
  for (int pl = 0; pl < NPQ; ++pl)   // wave (qb, kg) moves rows 8 kg .. 8 kg + 7 of its query block, every plane
    __builtin_amdgcn_global_load_lds(
        (gptr_t)(p.q[pl] + (qrow0 + 8 * kg + kr) * p.ldq + head * 64 + kc * 8),
        (lptr_t)(smem + QOFF + (qb * NP + pl) * kAttQBytes + kg * 1024), 16, 0, 0);
  frag8 qf[NP][4];   // read from the Q tile behind the first barrier of the key loop
  __builtin_amdgcn_sched_barrier(0);
  // all NS ring slots are free at the start: NS stages go in flight at once -- UNCONDITIONALLY (round 4): whether a
  // stage exists depends on n_keys, a device word this block has only just requested, and a conditional issue made the
  // whole ring wait for that round trip (the "entry -> DMAs issued 2.3 - 2.5 us" of the stamps).  A stage beyond the
  // key count reads rows clamped into the allocation (MSD_A_ISSUE) and is never looked at; the wait below covers it.
#pragma unroll
  for (int s = 0; s < NS; ++s) MSD_A_ISSUE(s, s)
  __builtin_amdgcn_sched_barrier(0);
  // n_keys is read HERE, behind the issue block: as a scalar load at the top of the kernel it shared lgkmcnt with the kernel-argument loads, whose first use
  // waits for everything scalar -- one L2 round trip in front of the whole prologue.  Now it overlaps the ring's landing.
  // (An explicit scalar load: behind the LDS-DMA builtins the compiler no longer proves the word unclobbered and would
  // fall back to a VECTOR load -- whose use then waits with vmcnt(0) for the whole ring.  Load and wait in one statement,
  // SGPR destination: MI355X guide 5.7 form (i); lgkmcnt does not count the LDS-DMA, which is on vmcnt.)
  int nkeys;
  {
    const int* nkp = p.n_keys + seg;
    asm volatile("s_nop 4\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(nkeys) : "s"(nkp) : "memory");
  }
  const int nst_all = (nkeys + kAttStageKeys - 1) / kAttStageKeys;
  // this block's stages: global stage index = ks + i * ksplit, i = 0..nst-1
  const int nst = nst_all > ks ? (nst_all - ks + p.ksplit - 1) >> kl2 : 0;
  MSD_TS_AT(ts_cls, ts_blk, 1)

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = NEG, l_run = 0.f;
  float rq = 1.f;   // QS: 1/rms of this lane's query row

  int buf = 0;
  // one stage of the key loop.  A lambda so that the QS form can PEEL stage 0 in the source (below); the plain form calls
  // it from the loop it always was.
  auto stage = [&](const int st) __attribute__((always_inline)) {
    // stage st must have landed.  Issued so far: stages 0 .. st+NS-2 (0 .. NS-1 at st = 0).
    if (st == 0) {   // NS stages were issued, whatever nst is
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PW) : "memory");
    } else if (st > 0 && st + NS - 2 < nst && NS > 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // stage st visible everywhere; compute(st-1) done everywhere
    __builtin_amdgcn_sched_barrier(0);
    if (st == 0) {   // the Q tile landed with stage 0 (it was issued first): fragments of this lane's query row
#pragma unroll
      for (int pl = 0; pl < NPQ; ++pl)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          qf[pl][s4] = *reinterpret_cast<const frag8*>(smem + QOFF + (qb * NP + pl) * kAttQBytes + q_lane * 128 +
                                                       (((2 * s4 + hi) ^ (q_lane & 7)) << 4));
      if constexpr (QS) {   // all reads issued together (fixed trip count, clamped index, zero for the tail); q_tiles % 4 == 0
        typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_t;
        lds_f32x4_t sp = (lds_f32x4_t)(size_t)(unsigned)(size_t)(smem + SSQOFF + (qb * 32 + q_lane) * p.q_tiles * 4);
        const int n4 = p.q_tiles >> 2;
        f32x4 sv[kAuxMaxTiles / 4];
#pragma unroll
        for (int t = 0; t < kAuxMaxTiles / 4; ++t) sv[t] = sp[t < n4 ? t : 0];
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < kAuxMaxTiles / 4; ++t) {
          const float g4 = (sv[t][0] + sv[t][1]) + (sv[t][2] + sv[t][3]);
          acc += t < n4 ? g4 : 0.f;
        }
        rq = 1.0f / sqrtf(__builtin_fmaf(acc, p.q_inv_d, 1e-6f));
      }
    }
#if MSD_TIMESTAMPS
    if (st == 0) { MSD_TS_AT(ts_cls, ts_blk, 2) }
#endif
    // The DMA of stage st+NS-1 goes into the slot compute(st-1) just released.  It is issued
    // AFTER this wave's S^T MFMAs (below): as the first thing after the barrier the eight
    // waves' DMA instructions queue up at the CU's address unit and hold back the MFMAs
    // behind them in the in-order instruction stream.
    const bool do_issue = st > 0 && st + NS - 1 < nst;
    int nb = buf + NS - 1;
    if (nb >= NS) nb -= NS;
    const int kb0 = (ks + st * p.ksplit) * kAttStageKeys + kg * 32;  // first key of this wave's block
    if (kb0 >= nkeys) {
      if (do_issue) MSD_A_ISSUE(st + NS - 1, nb)
    } else {
      const char* kt = smem + buf * STAGE;
      const char* vt = kt + NP * kAttKBytes;
      // ---- S^T = K . Q^T ------------------------------------------------------------
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const int krow = kg * 32 + q_lane;
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        frag8 kf[NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          kf[pl] = *reinterpret_cast<const frag8*>(kt + pl * kAttKBytes + krow * 128 +
                                                   (((2 * sx + hi) ^ (krow & 7)) << 4));
        s = MSD_MFMA_32X32X16(kf[0], qf[0][sx], s, 0, 0, 0);
        if (NP == 2) {
          if (!Q1) s = MSD_MFMA_32X32X16(kf[0], qf[NP - 1][sx], s, 0, 0, 0);
          s = MSD_MFMA_32X32X16(kf[NP - 1], qf[0][sx], s, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_issue) MSD_A_ISSUE(st + NS - 1, nb)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (QS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= rq;
      }
      // lane owns keys kb0 + (r&3) + 8*(r>>2) + 4*hi for r = 0..15
      float bmax = NEG;
      if (kb0 + 32 > nkeys) {  // only the last, ragged key block needs the bound
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= nkeys) s[r] = NEG;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) bmax = fmaxf(bmax, s[r]);
      bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
      const float m_new = fmaxf(m_run, bmax);
      const float alpha = fast_exp(m_run - m_new);
      float psum = 0.f;
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = fast_exp(s[r] - m_new);
        psum += pv[r];
      }
      l_run = __builtin_fmaf(l_run, alpha, psum);
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      // ---- P^T fragments (k-slot j of step ks <-> register 8*ks + j) ----------------
      frag8 pf[NP][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint32_t wh[4], wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // softmax weights are in [0, 1]: no range check
          if (NP == 2 && !P1) split2_h16(pv[8 * ks + 2 * j], pv[8 * ks + 2 * j + 1], wh[j], wl[j]);
          else wh[j] = cvt2_h16(pv[8 * ks + 2 * j], pv[8 * ks + 2 * j + 1]);
        }
        pf[0][ks] = as_frag(make_uint4(wh[0], wh[1], wh[2], wh[3]));
        if (NP == 2 && !P1) pf[NP - 1][ks] = as_frag(make_uint4(wl[0], wl[1], wl[2], wl[3]));
      }
      // ---- O^T += V^T . P^T -----------------------------------------------------------
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        frag8 vf[NP][2];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int d = db * 32 + q_lane;
            vf[pl][db] = *reinterpret_cast<const frag8*>(
                vt + pl * kAttVBytes + d * 256 + (((kg * 4 + ks * 2 + hi) ^ (d & 15)) << 4));
          }
        o0 = MSD_MFMA_32X32X16(vf[0][0], pf[0][ks], o0, 0, 0, 0);
        o1 = MSD_MFMA_32X32X16(vf[0][1], pf[0][ks], o1, 0, 0, 0);
        if (NP == 2) {
          if (!P1) {
            o0 = MSD_MFMA_32X32X16(vf[0][0], pf[NP - 1][ks], o0, 0, 0, 0);
            o1 = MSD_MFMA_32X32X16(vf[0][1], pf[NP - 1][ks], o1, 0, 0, 0);
          }
          o0 = MSD_MFMA_32X32X16(vf[NP - 1][0], pf[0][ks], o0, 0, 0, 0);
          o1 = MSD_MFMA_32X32X16(vf[NP - 1][1], pf[0][ks], o1, 0, 0, 0);
        }
      }
    }
    buf = (buf + 1 == NS) ? 0 : buf + 1;
  };
  if constexpr (QS) {
    // stage 0 peeled in the source: it alone reads the Q fragments and the row statistics, and left inside the loop those
    // reads keep the compiler from peeling it by itself as it does for the plain form -- the QS launch then ran 2.4 us
    // longer than the plain one on the same keys (profiles/r06c_*: 17.35 against 14.93 us)
    if (nst > 0) stage(0);
    for (int st = 1; st < nst; ++st) stage(st);
  } else {
    for (int st = 0; st < nst; ++st) stage(st);
  }
#undef MSD_A_ISSUE
  // full-row sum: combine the two half-lanes that share a query
  l_run += __shfl_xor(l_run, 32, 64);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a prologue stage beyond this block's key range may still be landing)
  __syncthreads();  // every wave is done with the K/V ring before it becomes the merge slab
  MSD_TS_AT(ts_cls, ts_blk, 3)
  PrefetchRegsT<PF> pf_keep;   // warm a later GEMM's weights behind the merge below (gemm_h16.h WeightPrefetch)
  {
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if constexpr (!kPfWaveAttn) prefetch_weights<PF>(p.pf, lin, gridDim.x * gridDim.y * gridDim.z, p.q[0], pf_keep);
  }

  // ---- merge the 4 key-group partials of each query block through LDS --------------
  // per wave: O [32 q][64 d + 4 pad] fp32, then m[32], l[32]
  float* sm = reinterpret_cast<float*>(smem);
  float* mine = sm + wave * kAttWStride;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    // registers 4g..4g+3 are d = 8g + 4hi + 0..3 (rows of the 32x32 C tile)
    const int d = 8 * g + 4 * hi;
    *reinterpret_cast<float4*>(mine + q_lane * kAttOLD + d) =
        make_float4(o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
    *reinterpret_cast<float4*>(mine + q_lane * kAttOLD + 32 + d) =
        make_float4(o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
  }
  if (hi == 0) {
    mine[32 * kAttOLD + q_lane] = m_run;
    mine[32 * kAttOLD + 32 + q_lane] = l_run;
  }
  __syncthreads();
  MSD_TS_AT(ts_cls, ts_blk, 4)
  // QB query blocks x 32 q x 8 groups of 8 d = QB * 256 work items = one per thread
  {
    const int item = tid;
    const int mqb = item >> 8, q = (item >> 3) & 31, d0 = (item & 7) * 8;
    const float* wbase = sm + (size_t)(mqb * kAttKG) * kAttWStride;
    float mt = NEG;
#pragma unroll
    for (int w = 0; w < kAttKG; ++w) mt = fmaxf(mt, wbase[w * kAttWStride + 32 * kAttOLD + q]);
    float lt = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int w = 0; w < kAttKG; ++w) {
      const float* ws = wbase + w * kAttWStride;
      const float f = fast_exp(ws[32 * kAttOLD + q] - mt);
      lt = __builtin_fmaf(ws[32 * kAttOLD + 32 + q], f, lt);
      const float4 a = *reinterpret_cast<const float4*>(ws + q * kAttOLD + d0);
      const float4 b = *reinterpret_cast<const float4*>(ws + q * kAttOLD + d0 + 4);
      acc[0] = __builtin_fmaf(a.x, f, acc[0]); acc[1] = __builtin_fmaf(a.y, f, acc[1]);
      acc[2] = __builtin_fmaf(a.z, f, acc[2]); acc[3] = __builtin_fmaf(a.w, f, acc[3]);
      acc[4] = __builtin_fmaf(b.x, f, acc[4]); acc[5] = __builtin_fmaf(b.y, f, acc[5]);
      acc[6] = __builtin_fmaf(b.z, f, acc[6]); acc[7] = __builtin_fmaf(b.w, f, acc[7]);
    }
    const size_t row = (size_t)seg * p.q_rows_per_seg + blk * kRows + mqb * 32 + q;
    if (p.ksplit == 1) {
      const float inv = (lt > 0.f) ? 1.0f / lt : 0.f;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[e] * inv;
      RangeCheck rc;
      store_h16x8<NP>(p.o, row * p.ldo + head * 64 + d0, v, rc);
      rc.commit(p.sat, p.sat_tag);
    } else if (p.tickets != nullptr) {
      // in-launch merge: group = (segment, query block of the launch, head); the LDS word behind the merge slab is free
      const int heads = gridDim.x, nblk = (int)(gridDim.y >> kl2);
      const int group = ((int)seg * nblk + blk) * heads + head;
      int* slot = reinterpret_cast<int*>(smem + QB * kAttKG * kAttWStride * 4);
      if (p.ksplit == 4) attention_inlaunch_merge<NP, 4>(p, ks, row, head, heads, d0, acc, mt, lt, group, tid, slot);
      else if (p.ksplit == 2) attention_inlaunch_merge<NP, 2>(p, ks, row, head, heads, d0, acc, mt, lt, group, tid, slot);
      else attention_inlaunch_merge<NP, 8>(p, ks, row, head, heads, d0, acc, mt, lt, group, tid, slot);
    } else {
      const int heads = gridDim.x;
      float* po = p.part_o + (((size_t)ks * p.total_rows + row) * heads + head) * 64 + d0;
      *reinterpret_cast<float4*>(po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(po + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
      if (d0 == 0) {
        float* pm = p.part_ml + (((size_t)ks * p.total_rows + row) * heads + head) * 2;
        pm[0] = mt;
        pm[1] = lt;
      }
    }
  }
  prefetch_done(pf_keep);
  MSD_TS_AT(ts_cls, ts_blk, 5)
  MSD_TS_END(ts_cls, ts_blk, gridDim.x * gridDim.y * gridDim.z)
}

// Finish a key-split attention: out[row][head*64 + d] = sum_ks O_ks e^(m_ks - m) / sum_ks l_ks e^(m_ks - m)
// KS = the split count as a compile-time constant (2, 4, 8; 0 = run-time loop).  With a run-time count the two loops
// over the splits stayed rolled, each iteration waiting for its own loads: six dependent L2 round trips in a kernel
// that moves 4 MB -- 2.7 of its 5.0 us (profiles/r03p_phase_times_b1.txt).  Unrolled, the (m, l) pairs and the O rows
// of every split are in flight together before anything is computed; the sums still run in split order (same bits).
template <int NP, int KS = 0>
__global__ void __launch_bounds__(256) attention_merge_kernel(AttnParams p, int heads, unsigned inv_heads) {
#pragma clang fp contract(off)   // (see attention_kernel)
  warm_kernargs<kernarg_lines<AttnParams, int, unsigned>()>();
  const int item = blockIdx.x * 256 + threadIdx.x;   // (row, head, 8-wide d group)
  // row = (item / 8) / heads through the host's ceil(2^32 / heads) (exact while (item / 8) * heads < 2^32), no software division
  const int rh = item >> 3, row = heads > 1 ? (int)__umulhi((unsigned)rh, inv_heads) : rh;
  const int d0 = (item & 7) * 8, head = rh - row * heads;
  if (row >= p.total_rows) return;
  MSD_TS_BEGIN(6, blockIdx.x)
  float mt = -1e30f;
  float lt = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  float v[8];
  if constexpr (KS > 0) {
    f32x2 ml[KS];
    f32x4 oa[KS], ob[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const size_t base = ((size_t)ks * p.total_rows + row) * heads + head;
      ml[ks] = *reinterpret_cast<const f32x2*>(p.part_ml + base * 2);
      oa[ks] = *reinterpret_cast<const f32x4*>(p.part_o + base * 64 + d0);
      ob[ks] = *reinterpret_cast<const f32x4*>(p.part_o + base * 64 + d0 + 4);
    }
    merge_split_partials<KS>(ml, oa, ob, v);
  } else {
    for (int ks = 0; ks < p.ksplit; ++ks)
      mt = fmaxf(mt, p.part_ml[(((size_t)ks * p.total_rows + row) * heads + head) * 2]);
    for (int ks = 0; ks < p.ksplit; ++ks) {
      const size_t base = ((size_t)ks * p.total_rows + row) * heads + head;
      const float f = fast_exp(p.part_ml[base * 2] - mt);
      lt = __builtin_fmaf(p.part_ml[base * 2 + 1], f, lt);
      const float4 a = *reinterpret_cast<const float4*>(p.part_o + base * 64 + d0);
      const float4 b = *reinterpret_cast<const float4*>(p.part_o + base * 64 + d0 + 4);
      acc[0] = __builtin_fmaf(a.x, f, acc[0]); acc[1] = __builtin_fmaf(a.y, f, acc[1]);
      acc[2] = __builtin_fmaf(a.z, f, acc[2]); acc[3] = __builtin_fmaf(a.w, f, acc[3]);
      acc[4] = __builtin_fmaf(b.x, f, acc[4]); acc[5] = __builtin_fmaf(b.y, f, acc[5]);
      acc[6] = __builtin_fmaf(b.z, f, acc[6]); acc[7] = __builtin_fmaf(b.w, f, acc[7]);
    }
    const float inv = (lt > 0.f) ? 1.0f / lt : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = acc[e] * inv;
  }
  RangeCheck rc;
  store_h16x8<NP>(p.o, (size_t)row * p.ldo + head * 64 + d0, v, rc);
  rc.commit(p.sat, p.sat_tag);
  MSD_TS_AT(6, blockIdx.x, 5)
  MSD_TS_END(6, blockIdx.x, gridDim.x)
}

template <int NP, int NS, int QB, int QP>
inline hipError_t attention_prepare_one() {
  constexpr int smem = attention_smem<NP, NS, QB, (QP & 4) != 0>();
  if (smem < 64 * 1024) return hipSuccess;
  const hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<NP, NS, QB, kPfNone, QP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipError_t b = hipSuccess;
  if constexpr (QB < 4)
    b = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<NP, NS, QB, NP == 2 ? 1 : 0, QP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  return a != hipSuccess ? a : b;
}

// NS: LDS ring depth (NP = 2: one stage is 64 KiB -> NS = 2; NP = 1: 32 KiB -> NS = 3)
template <int NP>
constexpr int attention_ns() { return (NP == 2) ? 2 : 3; }

template <int NP, int NS = attention_ns<NP>()>
inline hipError_t attention_prepare() {
  hipError_t e = hipSuccess, r;
#define MSD_ATT_PREP(QB_, QP_) if ((r = attention_prepare_one<NP, NS, QB_, QP_>()) != hipSuccess) e = r;
  MSD_ATT_PREP(1, 0) MSD_ATT_PREP(2, 0)
  if constexpr (NP == 2) {
    MSD_ATT_PREP(1, 1) MSD_ATT_PREP(2, 1) MSD_ATT_PREP(1, 2) MSD_ATT_PREP(2, 2) MSD_ATT_PREP(1, 3) MSD_ATT_PREP(2, 3)
    MSD_ATT_PREP(4, 0) MSD_ATT_PREP(4, 1) MSD_ATT_PREP(4, 2) MSD_ATT_PREP(4, 3)
    MSD_ATT_PREP(1, 4) MSD_ATT_PREP(2, 4) MSD_ATT_PREP(1, 5) MSD_ATT_PREP(2, 5) MSD_ATT_PREP(1, 6) MSD_ATT_PREP(2, 6)
    MSD_ATT_PREP(1, 7) MSD_ATT_PREP(2, 7)
  }
#undef MSD_ATT_PREP
  return e;
}

template <int NP, int QP>
inline void launch_attention_qp(const AttnParams& p, int heads, int segs, hipStream_t stream) {
  constexpr int NS = attention_ns<NP>();
  // 64-row blocks share K/V between two query blocks; when that leaves most of the 256 CUs without
  // a block, 32-row blocks (twice as many) finish sooner
  const int blocks64 = heads * (p.q_rows_per_seg / 64) * p.ksplit * segs;
  constexpr bool QS = (QP & 4) != 0;
  constexpr int smem1 = attention_smem<NP, NS, 1, QS>(), smem2 = attention_smem<NP, NS, 2, QS>();
  // one instantiation per prefetch kind (gemm_h16.h); the single-plane mode never prefetches
  constexpr int PFW = NP == 2 ? 1 : 0;   // attention launches carry at most one target
  const bool pfw = NP == 2 && prefetch_kind(p.pf) >= 1;
  const dim3 g1(heads, (p.q_rows_per_seg / 32) * p.ksplit, segs), g2(heads, (p.q_rows_per_seg / 64) * p.ksplit, segs);
  const int qb = attention_query_blocks((p.allow_qb4 && !QS) ? blocks64 : (blocks64 > 256 ? 256 : blocks64), p.q_rows_per_seg, NP);
  if constexpr (NP == 2 && !QS) {
    if (qb == 4) {   // (no prefetch wave: the caller moved the launch's weight target elsewhere -- msd_api.hip)
      const dim3 g4(heads, (p.q_rows_per_seg / 128) * p.ksplit, segs);
      hipLaunchKernelGGL((attention_kernel<NP, NS, 4, kPfNone, QP>), g4, dim3(4 * kAttKG * 64), (attention_smem<NP, NS, 4>()), stream, p);
      return;
    }
  }
  if (qb == 1) {
    if (pfw) hipLaunchKernelGGL((attention_kernel<NP, NS, 1, PFW, QP>), g1, dim3(kAttKG * 64 + (kPfWaveAttn && PFW ? 64 : 0)), smem1, stream, p);
    else hipLaunchKernelGGL((attention_kernel<NP, NS, 1, kPfNone, QP>), g1, dim3(kAttKG * 64), smem1, stream, p);
  } else {
    if (pfw) hipLaunchKernelGGL((attention_kernel<NP, NS, 2, PFW, QP>), g2, dim3(2 * kAttKG * 64 + (kPfWaveAttn && PFW ? 64 : 0)), smem2, stream, p);
    else hipLaunchKernelGGL((attention_kernel<NP, NS, 2, kPfNone, QP>), g2, dim3(2 * kAttKG * 64), smem2, stream, p);
  }
}

template <int NP>
inline hipError_t launch_attention(const AttnParams& p_in, int heads, int segs, hipStream_t stream) {
  constexpr int NS = attention_ns<NP>();
  static const hipError_t attr = attention_prepare<NP, NS>();
  if (attr != hipSuccess) return attr;
  AttnParams p = p_in;
  p.ksplit_log2 = 0;
  while ((2 << p.ksplit_log2) <= p.ksplit) ++p.ksplit_log2;
  p.ksplit = 1 << p.ksplit_log2;   // a power of two (a request for 3 runs as 2): the kernel shifts, it never divides
  if constexpr (NP == 2) {
    if (p.q_ssq != nullptr && (p.q_tiles <= 0 || p.q_tiles > kAuxMaxTiles || p.q_tiles % 4)) return hipErrorInvalidValue;
    switch ((p.qp & 3) | (p.q_ssq != nullptr ? 4 : 0)) {
      case 1: launch_attention_qp<NP, 1>(p, heads, segs, stream); break;
      case 2: launch_attention_qp<NP, 2>(p, heads, segs, stream); break;
      case 3: launch_attention_qp<NP, 3>(p, heads, segs, stream); break;
      case 4: launch_attention_qp<NP, 4>(p, heads, segs, stream); break;
      case 5: launch_attention_qp<NP, 5>(p, heads, segs, stream); break;
      case 6: launch_attention_qp<NP, 6>(p, heads, segs, stream); break;
      case 7: launch_attention_qp<NP, 7>(p, heads, segs, stream); break;
      default: launch_attention_qp<NP, 0>(p, heads, segs, stream); break;
    }
  } else {
    launch_attention_qp<NP, 0>(p, heads, segs, stream);
  }
  if (p.ksplit > 1 && p.tickets == nullptr) {
    const int items = p.total_rows * heads * 8;
    const dim3 mg((items + 255) / 256), mb(256);
    const unsigned inv_heads = heads > 1 ? (unsigned)((0x100000000ull + (unsigned)heads - 1) / (unsigned)heads) : 0u;
    if (p.ksplit == 4) hipLaunchKernelGGL((attention_merge_kernel<NP, 4>), mg, mb, 0, stream, p, heads, inv_heads);
    else if (p.ksplit == 2) hipLaunchKernelGGL((attention_merge_kernel<NP, 2>), mg, mb, 0, stream, p, heads, inv_heads);
    else if (p.ksplit == 8) hipLaunchKernelGGL((attention_merge_kernel<NP, 8>), mg, mb, 0, stream, p, heads, inv_heads);
    else hipLaunchKernelGGL((attention_merge_kernel<NP, 0>), mg, mb, 0, stream, p, heads, inv_heads);
  }
  return hipGetLastError();
}

}  // namespace msd
