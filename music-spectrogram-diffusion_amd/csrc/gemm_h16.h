// 16-bit-plane MFMA GEMM ("h16": IEEE-half planes in libmsd_amd.so, bfloat16 planes in libmsd_amd_bf16.so --
// common.h) for the small-M (M = 256..2304) transformer projections.
//
//   C[M,N] (+)= A[M,K] . W[K,N]        A: 16-bit planes [M,K] row-major
//                                      W: packed at load time as W^T planes [N,K]
// NP = 1: one plane per operand.  NP = 2 ("f16x3" / "bf16x3"): A = Ah+Al, W = Wh+Wl and the
// product is Ah.Wh + Ah.Wl + Al.Wh (three v_mfma_f32_16x16x32_f16 per tile),
// fp32 accumulate: fp32-class accuracy at the 16-bit MFMA rate / 3.
//
// What bounds this kernel on MI355X (profiles/r01_*): with M = 256..512 every
// operand is L2-resident or streamed once, and the limiter is the per-CU vector
// memory path (TA): tools/ubench/load_patterns.hip measures 17 B/clk/CU for
// MFMA-fragment-shaped loads (16 rows x 64 B per instruction) but 43 B/clk/CU for
// full 128-byte lines.  Hence:
//   * operands enter the CU ONCE per block, as full 128 B rows (BK = 64 bf16,
//     8 lanes x 16 B per row, 8 rows per wave-instruction), and are shared by the
//     2x2 waves through LDS; bigger block tiles (BM = 128) for the wide-N GEMMs cut
//     bytes per FLOP further;
//   * global -> LDS by LDS-DMA (global_load_lds) into an NS-deep ring of XOR-swizzled
//     tiles -> ds_read_b128 fragments; the steady-state loop is branch-free with pinned
//     issue order and counted s_waitcnt vmcnt(N), so younger prefetches are never drained;
//   * one barrier per K-tile; the accumulators go through an LDS slab so that the
//     epilogue stores whole 16/32-byte row segments whatever the MFMA layout was.
// blockIdx -> tile is XCD-aware (block b runs on XCD b % 8): the BM-blocks of one
// column slice are consecutive on ONE XCD, so a weight slice is fetched from HBM
// once into that XCD's L2.
#pragma once
#include <type_traits>
#include "common.h"
#include "phase_stamps.h"

namespace msd {

// Warm what a LATER launch will read.  A DDPM step touches every weight matrix (and each layer's cached
// cross-attention K / V^T) once per 1.1 ms, so every launch started on HBM-cold operands (+1.5 .. 2 us against
// warm ones, tools/ubench/gemm_bench.hip).  None of that data depends on anything computed in the step: every
// block of an EARLIER launch, once its main loop is done, touches its share of the later launch's operands --
// one 4-byte load per 128-byte line, nobody reads the data, the latency hides behind the producer's epilogue.
// The lines wait in the memory-side Infinity Cache: touching each line ONCE, from whatever XCD the block
// happens to run on, measured faster than touching it from the XCDs whose blocks will read it
// (profiles/r02_prefetch_ab.log), so a target is just a 2-D byte range shared by all blocks of the launch.
// (Warming each layer's cached cross-attention K / V^T the same way -- from the attention-out GEMM, or from the
// QKV GEMM three launches ahead -- was measured twice and dropped: the producer loses more than the
// cross-attention gains.)
struct PrefetchTarget {
  const char* base[2] = {nullptr, nullptr};   // up to two planes with the same geometry
  int rows = 0;            // rows per plane; 0 = unused slot
  int row_stride = 0;      // bytes between rows
  int lpr = 0;             // 128-byte lines to touch per row (<= 64)
  int lg = 0;              // log2 of the line slots per row (host: smallest power of two >= lpr)
  void set(const void* p0, const void* p1, int rows_, int row_stride_, int row_bytes) {
    base[0] = static_cast<const char*>(p0); base[1] = static_cast<const char*>(p1);
    rows = rows_; row_stride = row_stride_; lpr = (row_bytes + 127) >> 7;
    lg = 0;
    while ((1 << lg) < lpr) ++lg;
  }
};
constexpr int kMaxPrefetchTargets = 1;   // a launch carries at most one target (more were measured: docs/history.md)
struct WeightPrefetch {
  PrefetchTarget t[kMaxPrefetchTargets];
  int n = 0;   // used slots
  void add(const PrefetchTarget& x) { if (x.rows > 0 && n < kMaxPrefetchTargets) t[n++] = x; }
};

// The touches are plain 4-byte loads issued from inline asm into registers nobody reads: the compiler does not
// know they are loads, so it never waits for them (an LDS-DMA touch would be waited for in front of the
// epilogue's first LDS read); `keep` pins the destination registers until prefetch_done() at the end of the
// kernel, and s_endpgm waits for outstanding loads by itself.  kPrefetchPerThread touches per wave and target.
// PF = number of targets, a template parameter: every kernel instantiation has ONE straight-line path through
// here, so the destination registers reach prefetch_done() without a copy or a merge of values --
// tools/check_prefetch_regs.py verifies that on the compiled listing (tests/test_prefetch_static.py).
constexpr int kPrefetchPerThread = 3;
template <int PF>
struct PrefetchRegsT { uint32_t r[PF > 0 ? PF * kPrefetchPerThread : 1]; };
enum { kPfNone = 0 };
__host__ __device__ inline int prefetch_kind(const WeightPrefetch& pf) { return pf.n; }

// blk / nblk: launch-wide index of this block and number of blocks; `valid`: any mapped device address (touched
// by lanes that have nothing to do, so that there is no branch around a load).
// One wave-wide touch covers 64 >> lg whole rows (lane = (row offset, line slot)): everything up to the first
// row of a touch is wave-uniform SCALAR arithmetic and a lane adds its row offset and line.  The index math
// sits between the main loop and the epilogue of every launch and must stay a few VALU instructions (a first
// version with per-lane divisions cost +1 us per launch, more than the prefetch wins).  Waves are dealt
// block-cyclically: touch u of wave w of block `blk` starts at row ((blk + nblk * u) * nwave + w) * rows-per-touch.
template <int PF>
__device__ __forceinline__ void prefetch_weights(const WeightPrefetch& pf, int blk, int nblk, const void* valid,
                                                 PrefetchRegsT<PF>& keep) {
  if constexpr (PF == kPfNone) {
    keep.r[0] = 0;
    return;
  } else {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), nwave = (int)blockDim.x >> 6;
    const int lane = (int)threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const PrefetchTarget& t = pf.t[k];
      const int rows = t.rows, lpr = t.lpr;
      const int planes = t.base[1] && t.base[1] != t.base[0] ? 2 : 1;
      const int rpt = 64 >> t.lg;                                      // rows per touch
      const int sub = lane >> t.lg, line = lane & ((1 << t.lg) - 1);
      const uint32_t lane_off = (uint32_t)sub * (uint32_t)t.row_stride + (uint32_t)line * 128u;
#pragma unroll
      for (int u = 0; u < kPrefetchPerThread; ++u) {
        const int rp0 = ((blk + nblk * u) * nwave + wave) * rpt;       // scalar from here ...
        const int pl = (planes == 2 && rp0 >= rows) ? 1 : 0, row0 = rp0 - pl * rows;
        const char* row_ptr = t.base[pl] + (size_t)row0 * t.row_stride;   // ... to here
        const bool in = rp0 < rows * planes && row0 + sub < rows && line < lpr;
        const char* src = in ? row_ptr + lane_off : reinterpret_cast<const char*>(valid);
        asm volatile("global_load_dword %0, %1, off ; msd_prefetch" : "=&v"(keep.r[k * kPrefetchPerThread + u]) : "v"(src) : "memory");
      }
    }
  }
}
template <int PF>
__device__ __forceinline__ void prefetch_done(const PrefetchRegsT<PF>& keep) {
#pragma unroll
  for (int u = 0; u < (PF > 0 ? PF * kPrefetchPerThread : 1); ++u) asm volatile("" ::"v"(keep.r[u]));
}

// The same touches from a wave of their own: the block is launched with 64 extra threads, and that wave does
// nothing but touch the later launch's operands and end (`pf_wave` in the launch parameters).  Its loads are on
// ITS vmcnt counter: the compute waves never wait for them -- the in-epilogue form above makes every producer wait
// at s_endpgm for touches it issued a microsecond earlier (cross-attention 11.6 -> 13.3 us) -- and the
// touches leave at the start of the launch instead of behind its main loop, which gives the lines that much more
// time to arrive before their consumer starts.  An ended wave no longer counts at s_barrier, so the compute
// waves' barriers are unaffected once it has gone.  One wave issues what the four compute waves issued together.
// (-1.0 % step time same-box against the in-epilogue touches, profiles/r03f_env_ab.log)
constexpr bool kPfWave = true;       // GEMM launches
constexpr bool kPfWaveAttn = true;   // attention launches
// LDS_SINK: the touches are LDS-DMA dwords into 256 bytes of LDS nobody reads instead of loads into a
// register -- for a prefetch wave that goes on living behind them (attention.h kv_touch_ahead): there the register form's
// hazard (the compiler reusing the destination once it believes the value dead) cannot be excluded by construction.
template <int PF, bool LDS_SINK = false>
__device__ __forceinline__ void prefetch_wave(const WeightPrefetch& pf, int blk, int nblk, const void* valid,
                                              char* sink_lds = nullptr) {
  if constexpr (PF != kPfNone) {
    constexpr int TOUCHES = 4 * kPrefetchPerThread;
    const int lane = (int)threadIdx.x & 63;
    // ONE destination register for every touch, read-write in each asm statement and kept live to the end of the
    // wave: the compiler believes an asm output is complete when the statement ends, so a register it were free to
    // reuse (e.g. for the next address -- it did: `global_load_dword v2, v[2:3]` followed by a new address in
    // v[2:3], overwritten by the returning load: a memory fault on the MI355X) must not exist.  Several loads
    // in flight to the same register are harmless: nobody reads it.
    uint32_t sink = 0;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const PrefetchTarget& t = pf.t[k];
      const int rows = t.rows, lpr = t.lpr;
      const int planes = t.base[1] && t.base[1] != t.base[0] ? 2 : 1;
      const int rpt = 64 >> t.lg;
      const int sub = lane >> t.lg, line = lane & ((1 << t.lg) - 1);
      const uint32_t lane_off = (uint32_t)sub * (uint32_t)t.row_stride + (uint32_t)line * 128u;
#pragma unroll
      for (int u = 0; u < TOUCHES; ++u) {
        const int rp0 = (blk + nblk * u) * rpt;
        const int pl = (planes == 2 && rp0 >= rows) ? 1 : 0, row0 = rp0 - pl * rows;
        const char* row_ptr = t.base[pl] + (size_t)row0 * t.row_stride;
        const bool in = rp0 < rows * planes && row0 + sub < rows && line < lpr;
        const char* src = in ? row_ptr + lane_off : reinterpret_cast<const char*>(valid);
        if constexpr (LDS_SINK) {
          typedef const __attribute__((address_space(1))) void* pf_gptr_t;
          typedef __attribute__((address_space(3))) void* pf_lptr_t;
          __builtin_amdgcn_global_load_lds((pf_gptr_t)src, (pf_lptr_t)(size_t)(unsigned)(size_t)sink_lds, 4, 0, 0);
        } else {
          asm volatile("global_load_dword %0, %1, off ; msd_prefetch" : "+v"(sink) : "v"(src) : "memory");
        }
      }
    }
    asm volatile("" ::"v"(sink));
  }
}

struct GemmParams {
  const h16_t* A[2];
  const h16_t* B[2];
  int lda, ldb;
  int M, N, K;
  WeightPrefetch pf;  // optional: warm a later launch's weights (see WeightPrefetch)
  int pf_nblk = 0;    // blocks that take part in the prefetch (0 = the whole grid); they are blockIdx.x < pf_nblk
  int xcd_rows = 1;   // LDS-DMA kernel: the 8 XCDs form an xcd_rows x (8 / xcd_rows) grid over (M, N) tiles
  int xcd_walk_n = 0; // order in which an XCD's blocks walk its tiles
  // block -> tile map, precomputed by the launcher (TileMap::fill): the kernel used to derive it with FIVE software
  // integer divisions (v_rcp_iflag + a dozen dependent SALU instructions each: ~230 instructions, half a microsecond) in
  // front of its first LDS-DMA -- in every one of a step's 72 GEMM launches
  struct TileMap {
    int nbm = 0, nbn = 0;        // tiles
    int nbm_x = 1, nbn_x = 1;    // tiles per XCD row / column group
    int cx_log2 = 3;             // log2 of the column groups (8 / xcd_rows)
    unsigned inv_nbm_x = 0, inv_nbn_x = 0;   // ceil(2^32 / d): floor(t / d) == umulhi(t, inv) while t * d < 2^32 (d >= 2)
    void fill(int M, int N, int BM, int BN, int rx) {
      const int cx = 8 / rx;
      nbm = M / BM; nbn = N / BN;
      nbm_x = (nbm + rx - 1) / rx; nbn_x = (nbn + cx - 1) / cx;
      cx_log2 = cx == 8 ? 3 : (cx == 4 ? 2 : (cx == 2 ? 1 : 0));
      inv_nbm_x = nbm_x > 1 ? (unsigned)((0x100000000ull + (unsigned)nbm_x - 1) / (unsigned)nbm_x) : 0u;
      inv_nbn_x = nbn_x > 1 ? (unsigned)((0x100000000ull + (unsigned)nbn_x - 1) / (unsigned)nbn_x) : 0u;
    }
  } map;
  unsigned* sat = nullptr;   // half-plane range flag of the handle (common.h RangeCheck); nullptr = unchecked
  unsigned sat_tag = 1;      // what a flagged conversion stores there: kernel class + 1 (msd_api.hip)
};

// where an epilogue reports an activation that left the half-plane range
struct SatFlag {
  unsigned* p = nullptr;
  unsigned tag = 1;
};

typedef __attribute__((ext_vector_type(8))) plane_elem mfma_h16x8;
// native vector (NOT HIP's uint4 struct): arrays of it stay in registers under SROA
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ mfma_h16x8 as_frag(uint4 v) {
  union { uint4 u; mfma_h16x8 f; } c;
  c.u = v;
  return c.f;
}

__device__ __forceinline__ mfma_h16x8 ld_frag16(const h16_t* p) {
  return as_frag(*reinterpret_cast<const uint4*>(p));
}

// key position permutation inside each group of 16 keys of a V^T row, so that the
// attention kernel's P^T fragment (32x32 MFMA C layout) lines up with 16-byte
// V^T loads: offset o -> 8*((o>>2)&1) + (o&3) + 4*(o>>3)   (see attention.h)
__device__ __forceinline__ int vt_perm16(int o) { return 8 * ((o >> 2) & 1) + (o & 3) + 4 * (o >> 3); }

constexpr int kSlabPad = 4;
constexpr int kGemmBK = 64;  // bf16 elements per K-tile = one 128-byte line per row

// byte offset of 16-byte chunk `chunk` (0..7) of `row` in a [rows][64] bf16 LDS tile;
// chunk ^ (row & 7) makes every ds_read_b128 lane group (16 rows x one chunk) and every
// ds_write_b128 group (8 lanes of one row) bank-conflict free.
__device__ __forceinline__ int lds_tile_off(int row, int chunk) {
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}

// ----------------------------------------------------------------------------
// LDS-DMA ("global_load_lds") staging.  Ablation of an earlier register-staged kernel
// (kept for the micro-benchmarks only: tools/ubench/gemm_h16_regstaged.h, MSD_ABL=4)
// showed ~45 % of its time in the ds_write pass (ds_write_b128 sustains only ~79 B/clk/CU and sits in front of the
// MFMAs in every wave).  Here the memory pipeline writes the tiles into LDS itself:
// no staging VGPRs, no ds_write, NS-deep LDS ring, one raw s_barrier per K-tile and a
// COUNTED s_waitcnt vmcnt(N) so the younger tiles' DMAs stay in flight across it.
// The DMA destination is wave-uniform base + lane*16, i.e. 8 rows x 128 B land
// row-major; the XOR swizzle that makes ds_read_b128 conflict-free is therefore
// applied to the per-lane SOURCE address (lane (r, c') fetches global chunk c' ^ r).
// ----------------------------------------------------------------------------
// One output tile (bm, bn) by the 256 threads of the calling block; `smem` = the block's dynamic LDS
// (gemm_h16_dma_smem bytes).
// accumulators of one wave's share of a BM x BN tile in the 2 x 2 wave layout (what gemm_tile hands to a register epilogue)
template <int BM, int BN>
struct GemmAcc {
  static constexpr int FM = BM / 32, FN = BN / 32;
  f32x4 a[FM][FN];
};

// PROLOGUE = false / REG_EPI = true (the persistent gated-MLP-in kernel below): the caller has already issued the tile's
// first NS K-tiles and the epilogue operands (gemm_tile_issue + Epi::prefetch), and takes the accumulators back in
// registers right behind the loop -- the ring is free from that point on (no slab), e.g. for the NEXT tile's K-tiles.
template <int NP, int BM, int BN, int NS, class Epi, int PF = kPfNone, bool PROLOGUE = true, bool REG_EPI = false>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const Epi& epi, int bm, int bn, char* smem,
                                          GemmAcc<BM, BN>* acc_out = nullptr, int ts_blk = -1, bool fresh = false) {
#if MSD_TIMESTAMPS
  const int msd_ts_blk = ts_blk >= 0 ? ts_blk : (int)blockIdx.x;   // which record the phase stamps of this tile go to
#endif
  // Wave layout.  2 x 2 waves of (BM/2) x (BN/2) by default.  W13 (BN == 48: the 32 x 48 tile, 256 tiles of a
  // 512 x 768 output = one per CU): three compute waves side by side, each the full 32 rows x 16 columns; wave 3
  // SHADOWS wave 2 (same fragment reads, same MFMAs, no slab store): it is there for its quarter of the DMA issue, and
  // a shadow costs no branch in the pinned instruction stream.  (Measured: a wave 3 that skips the reads and MFMAs
  // behind a wave-uniform branch per slot -- 12 taken branches per K-tile in every wave -- made the K = 2048 launch
  // 12.5 -> 16.2 us, profiles/r04k_*.)  The K order of every output element is the same in both layouts (and in every
  // tile shape): results do not depend on the tile a GEMM runs on.
  constexpr bool W13 = BN == 48;
  static_assert(!W13 || (NP == 2 && BM == 32), "1 x 3 wave layout: 32 x 48 tiles, two planes");
  constexpr int WM = W13 ? BM : BM / 2, WN = W13 ? 16 : BN / 2;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  constexpr int A_LD = BM / 32, B_LD = BN / 32;   // DMA instructions per wave per plane (B_LD: 2 x 2 layout only)
  // B-operand DMA instructions per wave per K-tile.  2 x 2: wave w owns rows [w BN/4, +BN/4) of BOTH planes (B_LD
  // instructions each).  W13: 48 rows are not 4 x 8k, so wave w owns rows [(w & 1) 24, +24) of plane w >> 1.
  constexpr int B_Q = W13 ? BN / 16 : NP * B_LD;
  constexpr int PW = NP * A_LD + B_Q;             // DMA instructions per wave per K-tile
  constexpr int LDS_LD = BN + kSlabPad;
  static_assert((BM * LDS_LD + BM) * 4 <= NS * STAGE_BYTES, "epilogue slab must fit the operand LDS");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = W13 ? 0 : wave >> 1, wn = W13 ? (wave < 2 ? wave : 2) : wave & 1;
  const int m0 = bm * BM, n0 = bn * BN;

  // this wave DMAs rows [wave*BM/4, +BM/4) of every A plane and [wave*BN/4, +BN/4) of every
  // B plane; lane (r = lane>>3, c' = lane&7) fetches global chunk c' ^ r of row 8i + r.
  const int r8 = lane >> 3, csrc = (lane & 7) ^ r8;
  const h16_t* ga[NP];
  const h16_t* gb[NP];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    ga[pl] = p.A[pl] + (size_t)(m0 + wave * (BM / 4) + r8) * p.lda + csrc * 8;
    gb[pl] = W13 ? p.B[wave >> 1] + (size_t)(n0 + (wave & 1) * (BN / 2) + r8) * p.ldb + csrc * 8   // (one plane per wave)
                 : p.B[pl] + (size_t)(n0 + wave * (BN / 4) + r8) * p.ldb + csrc * 8;
  }
  const size_t a_step = (size_t)8 * p.lda, b_step = (size_t)8 * p.ldb;
#define MSD_A_SRC(PL, I, K0) (ga[PL] + (I) * a_step + (K0))
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // DMA number Q (0 .. PW) of this wave for K-tile KT into ring slot BUF.  2 x 2 layout: plane-major, A rows then B
  // rows of a plane.  W13: the A instruction of each plane, then this wave's B_Q instructions of its one B plane.
#define MSD_D_ISSUE1(KT, BUF, Q)                                                             \
  {                                                                                          \
    char* base_ = smem + (BUF) * STAGE_BYTES;                                                \
    const int k0_ = (KT) * kGemmBK;                                                          \
    if constexpr (W13) {                                                                     \
      if ((Q) < NP * A_LD)                                                                   \
        __builtin_amdgcn_global_load_lds((gptr_t)(MSD_A_SRC((Q) / A_LD, (Q) % A_LD, k0_)),   \
            (lptr_t)(base_ + ((Q) / A_LD) * A_BYTES + (wave * (BM / 4) + 8 * ((Q) % A_LD)) * 128), 16, 0, 0); \
      else                                                                                   \
        __builtin_amdgcn_global_load_lds((gptr_t)(gb[0] + ((Q) - NP * A_LD) * b_step + k0_), \
            (lptr_t)(base_ + NP * A_BYTES + (wave >> 1) * B_BYTES +                          \
                     ((wave & 1) * (BN / 2) + 8 * ((Q) - NP * A_LD)) * 128), 16, 0, 0);      \
    } else {                                                                                 \
      constexpr int AB_ = A_LD + (B_LD ? B_LD : 1);                                          \
      const int     pl_ = (Q) / AB_, r_ = (Q) % AB_;                                         \
      if (r_ < A_LD)                                                                         \
        __builtin_amdgcn_global_load_lds((gptr_t)(MSD_A_SRC(pl_, r_, k0_)),                  \
            (lptr_t)(base_ + pl_ * A_BYTES + (wave * (BM / 4) + 8 * r_) * 128), 16, 0, 0);   \
      else                                                                                   \
        __builtin_amdgcn_global_load_lds((gptr_t)(gb[pl_] + (r_ - A_LD) * b_step + k0_),     \
            (lptr_t)(base_ + NP * A_BYTES + pl_ * B_BYTES + (wave * (BN / 4) + 8 * (r_ - A_LD)) * 128), 16, 0, 0); \
    }                                                                                        \
  }
#define MSD_D_ISSUE(KT, BUF)                                                                \
  { _Pragma("unroll") for (int q0_ = 0; q0_ < PW; ++q0_) MSD_D_ISSUE1(KT, BUF, q0_) }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / kGemmBK;
  MSD_TS_BEGIN((ts_class<BM, BN>()), msd_ts_blk)
  // ---- prologue: all NS ring slots are free, so NS K-tiles go in flight at once -------
  if constexpr (PROLOGUE) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < nk) MSD_D_ISSUE(s, s)
  }
  // Epilogue operands (row statistics, step-indexed bias / gain rows, the residual tile) are
  // HBM-cold and used to be read by dependent global loads AFTER the K loop (+2..4 us per
  // launch).  They are DMAed into an aux LDS region behind the ring now, queued behind the
  // first tiles: vmcnt retires in order, so the loop's counted waits stay valid (they can only
  // over-wait by these few instructions) and the final vmcnt(0) covers them.
  char* const aux = smem + NS * STAGE_BYTES;
  if constexpr (PROLOGUE) epi.template prefetch<BM, BN>(aux, m0, n0, wave, lane);
  __builtin_amdgcn_sched_barrier(0);
  MSD_TS_STAMP(BM, BN, 1)

  // Fragment reads of one 32-wide half (kk) of the K-tile in ring slot BUF (prologue only; the
  // loop uses the per-slot forms below)
#define MSD_D_READ(FA, FB, BUF, KK)                                                          \
  {                                                                                          \
    const char* base_ = smem + (BUF) * STAGE_BYTES;                                          \
    const int c_ = (KK) * 4 + (lane >> 4);                                                   \
    _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                      \
      _Pragma("unroll") for (int i = 0; i < FM; ++i)                                         \
        FA[pl][i] = *reinterpret_cast<const mfma_h16x8*>(                                   \
            base_ + pl * A_BYTES + lds_tile_off(wm * WM + i * 16 + (lane & 15), c_));        \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                         \
        FB[pl][j] = *reinterpret_cast<const mfma_h16x8*>(                                   \
            base_ + NP * A_BYTES + pl * B_BYTES + lds_tile_off(wn * WN + j * 16 + (lane & 15), c_)); \
    }                                                                                        \
  }

  // ---- main loop, software-pipelined by half K-tiles ----------------------------------
  // The LDS fragment reads of one half always run under the MFMAs of the previous half, also
  // across the tile boundary: the per-tile barrier sits in the MIDDLE of tile kt (after its
  // last reads were issued), where it frees slot kt for the DMA of tile kt+NS and publishes
  // tile kt+1.  Before, every wave read a whole tile and then multiplied: LDS and MFMA pipes
  // alternated (64x96 tile: ~640 + ~580 clocks per K-tile) instead of overlapping.
  mfma_h16x8 fa0[NP][FM], fb0[NP][FN], fa1[NP][FM], fb1[NP][FN];
  // Fine-grained issue order inside one half step: the wave's DMA and ds_read instructions
  // are spread between its MFMAs, one group per slot q: [DMA q | ds_read q | MFMAs].  Issued
  // as a block they sit in front of the MFMAs in the in-order instruction stream while the
  // CU's address unit (64 B/clk, shared by the four waves) drains them -- the ablation in
  // tools/ubench/gemm_abl.hip showed DMA, LDS-read and MFMA time ADDING UP (0.28 + 0.19 +
  // 0.27 us per 64x96 K-tile) instead of overlapping.  Each slot is pinned by a sched_barrier.
  constexpr int RD = NP * (FM + FN);                    // ds_read_b128 per half step (== PW in the 2 x 2 layout)
  constexpr int MQ = (NP == 2 ? 3 : 1) * FM * FN;       // MFMAs per half step
  constexpr int SLOTS = RD > PW ? RD : PW;              // slots per half step: slot q = [DMA q | read q | MFMAs]
  constexpr int MPR = MQ / SLOTS;                       // MFMAs per slot (remainder in the last)
  static_assert(W13 || PW == RD, "one DMA and one fragment read per slot");
  // fragment read number Q of half KK of the tile in slot BUF
#define MSD_D_READ1(FA, FB, BUF, KK, Q)                                                      \
  {                                                                                          \
    const int     pl_ = (Q) / (FM + FN), r_ = (Q) % (FM + FN);                               \
    const char* base_ = smem + (BUF) * STAGE_BYTES;                                          \
    const int c_ = (KK) * 4 + (lane >> 4);                                                   \
    if (r_ < FM)                                                                             \
      FA[pl_][r_ < FM ? r_ : 0] = *reinterpret_cast<const mfma_h16x8*>(                     \
          base_ + pl_ * A_BYTES + lds_tile_off(wm * WM + r_ * 16 + (lane & 15), c_));        \
    else                                                                                     \
      FB[pl_][r_ < FM ? 0 : r_ - FM] = *reinterpret_cast<const mfma_h16x8*>(                \
          base_ + NP * A_BYTES + pl_ * B_BYTES + lds_tile_off(wn * WN + (r_ - FM) * 16 + (lane & 15), c_)); \
  }
  // MFMA number E: products outermost, so one accumulator recurs every FM*FN MFMAs
#define MSD_D_MFMA1(FA, FB, E)                                                               \
  {                                                                                          \
    const int     pr_ = (E) / (FM * FN), t_ = (E) % (FM * FN), i_ = t_ / FN, j_ = t_ % FN;    \
    const int     pb_ = (pr_ == 1) ? NP - 1 : 0, pa_ = (pr_ == 2) ? NP - 1 : 0;              \
    acc[i_][j_] = MSD_MFMA_16X16X32(FB[pb_][j_], FA[pa_][i_], acc[i_][j_], 0, 0, 0); \
  }
  // one half step: DMA of tile KT (if DO_ISSUE) into BUF_I, reads of (BUF_R, KK) into FAn/FBn,
  // MFMAs on FAc/FBc
#define MSD_D_HALF(DO_ISSUE, KT, BUF_I, FAn, FBn, BUF_R, KK, DO_READ, FAc, FBc)              \
  {                                                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < SLOTS; ++q_) {                                   \
      if (DO_ISSUE && q_ < PW) MSD_D_ISSUE1(KT, BUF_I, q_)                                   \
      if (DO_READ && q_ < RD) MSD_D_READ1(FAn, FBn, BUF_R, KK, q_)                           \
      _Pragma("unroll") for (int e_ = q_ * MPR; e_ < (q_ + 1 == SLOTS ? MQ : (q_ + 1) * MPR); ++e_) \
        MSD_D_MFMA1(FAc, FBc, e_)                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
  }
#define MSD_D_BARRIER __builtin_amdgcn_s_barrier();
#define MSD_D_DOISSUE 1
  // one K-tile: [reads of half 1 | MFMAs of half 0] wait+barrier [DMA of tile kt+NS, reads of
  // the next tile's half 0 | MFMAs of half 1]
#define MSD_D_STEP(DO_ISSUE, VMWAIT)                                                         \
  {                                                                                          \
    MSD_D_HALF(0, 0, 0, fa1, fb1, buf, 1, 1, fa0, fb0)                                       \
    int nb = buf + 1;                                                                        \
    if (nb == NS) nb = 0;                                                                    \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMWAIT) : "memory"); /* tile kt+1 landed */     \
    __builtin_amdgcn_s_waitcnt(0xC07F);  /* lgkmcnt(0): my reads of slot buf are complete */ \
    MSD_D_BARRIER  /* slot buf free everywhere; tile kt+1 visible */                         \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    MSD_D_HALF(DO_ISSUE, kt + NS, buf, fa0, fb0, nb, 0, 1, fa1, fb1)                         \
    /* The next half's fragments landed long ago.  Saying so with a compiler-visible wait */ \
    /* keeps hipcc from putting lgkmcnt(0) between the reads and the MFMAs at the loop   */ \
    /* head (its back-edge merge is conservative), which serialised them.                */ \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                      \
    buf = nb;                                                                                \
  }

  // (`fresh`, !PROLOGUE only: the caller issued exactly what the prologue issues and nothing since -- a block's first tile)
  if ((PROLOGUE || fresh) && nk >= NS) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PW) : "memory");  // tile 0 landed; NS-1 tiles in flight
  } else {   // (!PROLOGUE: the previous tile's epilogue STORES sit between the caller's DMAs and here -- vmcnt counts them too)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  MSD_TS_STAMP(BM, BN, 2)
  MSD_D_READ(fa0, fb0, 0, 0)
  __builtin_amdgcn_s_waitcnt(0xC07F);
  int buf = 0;  // LDS ring slot of tile kt
  int kt = 0;
  // steady state: unconditional DMA of tile kt+NS, tiles kt+2 .. kt+NS-1 stay in flight
  for (; kt + NS < nk; ++kt) MSD_D_STEP(MSD_D_DOISSUE, (NS - 2) * PW)
  // drain: the last NS tiles are all issued.  Counted waits here too (round 4: the drain waited with vmcnt(0), i.e. at
  // its first step for ALL remaining tiles instead of the next one): entered at kt = nk - NS, step j needs tile kt + 1
  // and may leave NS - 2 - j younger tiles in flight.  (nk < NS -- a K shorter than the ring -- keeps vmcnt(0).)
  if (nk >= NS) {
    if constexpr (NS >= 4) { if (kt + 1 < nk) { MSD_D_STEP(0, (NS - 2) * PW) ++kt; } }
    if constexpr (NS >= 3) { if (kt + 1 < nk) { MSD_D_STEP(0, (NS >= 4 ? NS - 3 : NS - 2) * PW) ++kt; } }
    if constexpr (NS >= 5) { static_assert(NS <= 4, "write the drain ladder for this ring depth"); }
  }
  for (; kt + 1 < nk; ++kt) MSD_D_STEP(0, 0)
  // last tile
  MSD_D_HALF(0, 0, 0, fa1, fb1, buf, 1, 1, fa0, fb0)
  MSD_D_HALF(0, 0, 0, fa0, fb0, buf, 0, 0, fa1, fb1)
#undef MSD_D_STEP
#undef MSD_D_HALF
#undef MSD_D_ISSUE1
#undef MSD_D_READ1
#undef MSD_D_MFMA1
#undef MSD_D_DOISSUE
#undef MSD_D_BARRIER
#undef MSD_D_READ
#undef MSD_D_ISSUE
#undef MSD_A_SRC
  __syncthreads();  // all fragment reads done before the slab overwrites the ring
  MSD_TS_STAMP(BM, BN, 3)
  if constexpr (REG_EPI) {
    static_assert(!W13, "register epilogues: 2 x 2 wave layout");
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc_out->a[i][j] = acc[i][j];
    return;
  } else {
  float* slab = reinterpret_cast<float*>(smem);
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  auto store_slab = [&]() {
    if (W13 && wave == 3) return;   // (the shadow wave: wave 2 stores these values)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<float4*>(slab + (size_t)(wm * WM + i * 16 + lm) * LDS_LD + wn * WN + j * 16 + ln) =
            make_float4(acc[i][j][0] * kWScaleInv, acc[i][j][1] * kWScaleInv, acc[i][j][2] * kWScaleInv,
                        acc[i][j][3] * kWScaleInv);   // weights are packed times kWScale (common.h)
  };
  PrefetchRegsT<PF> pf_keep;
  // (the in-epilogue form of the weight prefetch; the product's launches carry a prefetch WAVE instead: kPfWave)
  if constexpr (!kPfWave) prefetch_weights<PF>(p.pf, blockIdx.x, p.pf_nblk > 0 ? p.pf_nblk : (int)gridDim.x, p.B[0], pf_keep);
  else prefetch_weights<kPfNone>(p.pf, 0, 1, nullptr, reinterpret_cast<PrefetchRegsT<kPfNone>&>(pf_keep));
  store_slab();
  epi.template stats<BM, LDS_LD>(slab, m0, tid, aux);   // row statistics next to the slab stores: one barrier
  __syncthreads();
  MSD_TS_STAMP(BM, BN, 4)
  epi.template run<BM, BN, LDS_LD>(slab, m0, n0, tid, aux, /*stats_done=*/true, SatFlag{p.sat, p.sat_tag});
  MSD_TS_STAMP(BM, BN, 5)
  MSD_TS_END((ts_class<BM, BN>()), msd_ts_blk, gridDim.x)
  prefetch_done(pf_keep);
  }
}

// XCD-aware tile map (block b runs on XCD b % 8): XCD x owns the column tiles bn = x, x+8, ...
// and walks their BM-blocks consecutively, so each weight slice is filled into ONE L2.
// The grid is 8 * ceil(nbn / 8) * nbm; blocks past the last column tile exit.
// With xcd_rows = RX > 1 the XCDs also split the M-blocks (XCD (xr, xc) owns bm = xr mod RX,
// bn = xc mod 8/RX): every L2 then fetches A/RX + B*RX/8 instead of A + B/8 -- less fabric
// traffic when the activations are as large as the weights (N = D projections).
// (the divisions are the launcher's: GemmParams::TileMap).  `b`: index of the block among its problem's blocks.
__device__ __forceinline__ bool gemm_block_tile(const GemmParams& p, int b, int& bm, int& bn) {
  const GemmParams::TileMap& tm = p.map;
  const int RX = p.xcd_rows, CX = 1 << tm.cx_log2;
  const int xcd = b & 7, tt = b >> 3;
  const int xr = xcd >> tm.cx_log2, xc = xcd & (CX - 1);
  if (p.xcd_walk_n) {   // column tiles fastest inside an XCD (activation rows stay hot)
    const int q = tm.nbn_x > 1 ? (int)__umulhi((unsigned)tt, tm.inv_nbn_x) : tt;   // tt / nbn_x
    bm = q * RX + xr; bn = (tt - q * tm.nbn_x) * CX + xc;
  } else {              // row tiles fastest (a weight slice stays hot)
    const int q = tm.nbm_x > 1 ? (int)__umulhi((unsigned)tt, tm.inv_nbm_x) : tt;   // tt / nbm_x
    bm = (tt - q * tm.nbm_x) * RX + xr; bn = q * CX + xc;
  }
  return bn < tm.nbn && bm < tm.nbm;
}

constexpr int pf_threads(int pf) { return kPfWave && pf != kPfNone ? 64 : 0; }   // the prefetch wave of a launch, if any

template <int NP, int BM, int BN, int NS, class Epi, int PF = kPfNone>
__global__ void __launch_bounds__(256 + pf_threads(PF)) gemm_h16_dma_kernel(GemmParams p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  warm_kernargs<kernarg_lines<GemmParams, Epi>()>();
  if constexpr (kPfWave && PF != kPfNone) {
    if (threadIdx.x >= 256) {   // the prefetch wave (only launched with pf_wave)
      prefetch_wave<PF>(p.pf, blockIdx.x, gridDim.x, p.B[0]);
      return;
    }
  }
  int bm, bn;
  if (!gemm_block_tile(p, (int)blockIdx.x, bm, bn)) return;
  gemm_tile<NP, BM, BN, NS, Epi, PF>(p, epi, bm, bn, smem);
}

// ----------------------------------------------------------------------------
// Epilogues.  run<BM,BN,LD>(slab, m0, n0, tid): tile value (m,n) = slab[m*LD+n].
// ----------------------------------------------------------------------------
template <int LD>
__device__ __forceinline__ void tile_row8(const float* s0, int m, int n, float v[8]) {
  const float4 a0 = *reinterpret_cast<const float4*>(s0 + m * LD + n);
  const float4 a1 = *reinterpret_cast<const float4*>(s0 + m * LD + n + 4);
  v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w;
  v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
}

template <int NP>
__device__ __forceinline__ void store_h16x8(h16_t* const* planes, size_t off, const float v[8], RangeCheck& rc) {
  uint32_t wh[4], wl[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    rc.see(v[2 * e], v[2 * e + 1]);
    if (NP == 2) split2_h16(v[2 * e], v[2 * e + 1], wh[e], wl[e]);
    else wh[e] = cvt2_h16(v[2 * e], v[2 * e + 1]);
  }
  *reinterpret_cast<uint4*>(planes[0] + off) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
  if (NP == 2) *reinterpret_cast<uint4*>(planes[1] + off) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
}

// ----------------------------------------------------------------------------
// Folded RMSNorm + FiLM (layers.py:632-666).  h = x * rstd[m] * g[k] + b[k] feeds a
// bias-free Dense, so  h.W = rstd[m] * ((x (.) g).W) + (b.W):
//   * the PRODUCER of x (residual epilogue below) also writes y = x (.) g as bf16
//     planes -- g = gamma (.) (film_scale(step) + 1) of the NEXT norm -- and the
//     per-64-column partial sums of squares of x;
//   * the CONSUMER GEMM runs on y and its epilogue applies rstd[m] (from the partial
//     sums) and the step-indexed vector b.W, tabulated at load time for every step.
// This removes the separate norm kernel (and its round trip through HBM) in front of
// every decoder projection.
// ----------------------------------------------------------------------------
struct RowScale {
  const float* ssq = nullptr;   // [M][tiles] partial sums of squares of x; nullptr = no folding
  int tiles = 0;
  float inv_d = 0.f;
  const float* bias = nullptr;  // base of the step-indexed bias table, or nullptr
  int bias_step_stride = 0;     // elements between steps
  const int* step_ptr = nullptr;
};

constexpr int kAuxMaxTiles = 32;  // ssq partials per row the aux region is sized for (D <= 1024)

// Does this tile have an aux LDS region?  Every kernel of the library does (gemm_tile and the batched variants pass
// one), so the answer is a compile-time `true` -- NOT a test of the pointer: a null test of a generic pointer that
// points into LDS is what this ROCm's backend turns into an illegal v_cmp on src_shared_base once a tile loop keeps it
// from folding the test away (which instantiation fails moves with every unrelated edit).  tools/ubench/gemm_h16_regstaged.h, the one caller without aux rows, defines MSD_EPI_AUX_OPTIONAL.
#ifndef MSD_EPI_AUX_OPTIONAL
#define MSD_EPI_AUX_OPTIONAL 0
#endif
__device__ __forceinline__ bool aux_present(const char* aux) {
#if MSD_EPI_AUX_OPTIONAL
  return aux != nullptr;
#else
  (void)aux;
  return true;
#endif
}

typedef const __attribute__((address_space(1))) void* aux_gptr_t;
typedef __attribute__((address_space(3))) void* aux_lptr_t;

// LDS pointer of a generic pointer that is KNOWN to point into LDS: its low 32 bits.  (A generic -> LDS pointer CAST
// carries a null check unless the compiler can fold it, and both that check and a `__builtin_assume(p != nullptr)`
// meant to remove it end up, in kernels with tile loops, as a constant compare against the LDS null that this ROCm's
// backend lowers to an illegal v_cmp on src_shared_base.  An integer round trip has no null semantics.)
__device__ __forceinline__ aux_lptr_t lds_ptr_of(const void* generic) {
  return (aux_lptr_t)(size_t)(unsigned)(size_t)generic;
}

// The scan index (one device word, written by the previous step's sampler launch) as an EXPLICIT scalar load, waited
// for inside the statement (SGPR destination, MI355X guide 5.7 form (i)).  A plain `*step_ptr` behind the ring's LDS-DMA
// builtins compiles to a VECTOR load -- the compiler no longer proves the word unclobbered -- and a vector load beside
// LDS-DMA is waited for with vmcnt(0): the wave stood there until its whole share of the ring had landed, and the
// block's first barrier with it, before the step-indexed gain / bias rows could even be requested (every launch with
// such rows: 60 of a step's 111).  lgkmcnt does not count LDS-DMA.  `p` must be wave-uniform (a kernel argument).
__device__ __forceinline__ int scan_index(const int* p) {
  int v;
  asm volatile("s_nop 4\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// LDS-DMA of `bytes` contiguous, 16-byte aligned global bytes to dst (linear), one 1 KiB
// instruction per wave round-robin.  Lanes past the end re-fetch the last chunk; their LDS
// writes land in the padding (dst needs round_up(bytes, 1024) bytes).
__device__ __forceinline__ void aux_dma_linear(const void* g, char* dst, int bytes, int wave, int lane) {
  const int n_instr = (bytes + 1023) >> 10;
  for (int i = wave; i < n_instr; i += 4) {
    int off = i * 1024 + lane * 16;
    off = off < bytes - 16 ? off : bytes - 16;
    __builtin_amdgcn_global_load_lds((aux_gptr_t)((const char*)g + off), lds_ptr_of(dst + i * 1024), 16, 0, 0);
  }
}

// one instruction: `bytes` (<= 1024) contiguous global bytes to dst
__device__ __forceinline__ void aux_dma_row(const void* g, char* dst, int bytes, int lane) {
  int off = lane * 16;
  off = off < bytes - 16 ? off : bytes - 16;
  __builtin_amdgcn_global_load_lds((aux_gptr_t)((const char*)g + off), lds_ptr_of(dst), 16, 0, 0);
}

// RowScale aux layout: [BM * tiles ssq partials, rounded up to whole KiB (the DMA's granule)][bias row, 1 KiB].  The
// bias row sits right behind the partials this launch has (run-time `tiles`), so a kernel that sizes its LDS at launch
// time (gemm_h16_pair.h) pays for D / 32 tiles, not for kAuxMaxTiles; rowscale_aux_bytes() is the compile-time bound.
template <int BM>
constexpr int rowscale_aux_bytes() { return BM * kAuxMaxTiles * 4 + 1024; }
template <int BM>
__host__ __device__ __forceinline__ int rowscale_ssq_bytes(int tiles) { return (BM * tiles * 4 + 1023) & ~1023; }

template <int BM, int BN>
__device__ __forceinline__ void rowscale_prefetch(const RowScale& r, char* aux, int m0, int n0, int wave, int lane) {
  if (!r.ssq) return;
  aux_dma_linear(r.ssq + (size_t)m0 * r.tiles, aux, BM * r.tiles * 4, wave, lane);
  if (r.bias && wave == 3)
    aux_dma_row(r.bias + (size_t)scan_index(r.step_ptr) * r.bias_step_stride + n0, aux + rowscale_ssq_bytes<BM>(r.tiles), BN * 4, lane);
}

typedef const __attribute__((address_space(3))) float* lds_cf32;   // explicit LDS pointer: ds_read, not flat_load

// The bias row of a tile (index 0 = column n0), in the aux LDS region (rowscale_prefetch put it there).  Always LDS:
// with a second, global-memory form behind a run-time flag every element of the epilogue loops went through two
// scalar branches and -- at the join of the two forms -- an s_waitcnt vmcnt(0), i.e. the second pass of a thread
// over its items waited for the STORES of the first (phase stamps, profiles/r03p_phase_times_b1.txt: 2.5 us between
// slab and stores on the gated-MLP-in tile, 1.9 us on the QKV tile).
typedef const __attribute__((address_space(3))) f32x4* lds_cf32x4;
struct BiasRow {
  lds_cf32 l = nullptr;
  bool present = false;
  __device__ __forceinline__ float at(int i) const { return l[i]; }
  // entries i .. i + 7 (i a multiple of 4): two ds_read_b128
  __device__ __forceinline__ void at8(int i, float (&b)[8]) const {
    const f32x4 b0 = reinterpret_cast<lds_cf32x4>(l + i)[0], b1 = reinterpret_cast<lds_cf32x4>(l + i)[1];
    b[0] = b0[0]; b[1] = b0[1]; b[2] = b0[2]; b[3] = b0[3];
    b[4] = b1[0]; b[5] = b1[1]; b[6] = b1[2]; b[7] = b1[3];
  }
};

// The three forms of an epilogue's element loop -- 0: plain, 1: times rstd[m], 2: times rstd[m] plus bias[n] --
// as compile-time variants chosen ONCE per launch, so the loops themselves are branch-free.
template <class F>
__device__ __forceinline__ void rowscale_variants(bool folded, bool has_bias, F&& f) {
  if (!folded) f(std::integral_constant<int, 0>{});
  else if (!has_bias) f(std::integral_constant<int, 1>{});
  else f(std::integral_constant<int, 2>{});
}

// v = v * rs + bias[i .. i + 7] in form MODE (see rowscale_variants)
template <int MODE>
__device__ __forceinline__ void rowscale8(float (&v)[8], float rs, const BiasRow& bias, int i) {
  if constexpr (MODE == 2) {
    float b[8];
    bias.at8(i, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * rs + b[e];
  } else if constexpr (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * rs + 0.f;
  }
}

// element loops of the epilogues: item = tid, tid + 256, ... < ITEMS with a compile-time trip count (fully unrolled:
// the LDS reads of all passes are issued up front)
#define MSD_EPI_ITEMS(ITEMS, item)                                             \
  _Pragma("unroll") for (int it_ = 0; it_ < ((ITEMS) + 255) / 256; ++it_)      \
    if (const int item = tid + it_ * 256; ((ITEMS) % 256 == 0) || item < (ITEMS))

// rstd of the BM rows of this tile into LDS (rs[0..BM)); block-wide, ends with a barrier.  All 256
// threads take part (256 / BM per row, then a shuffle tree): the serial 24-term sum by BM threads
// over bank-conflicting LDS rows cost ~1 us per launch.  With `aux` the operands were prefetched by
// rowscale_prefetch; without, they are read from global memory.
template <int BM>
__device__ __forceinline__ void tile_rstd_compute(const RowScale& r, float* rs, int m0, int tid, const char* aux) {
  constexpr int TPR = 256 / BM;
  const int row = tid / TPR, part = tid % TPR;
  float acc = 0.f;
  if (aux_present(aux)) {
    // all reads issued together (fixed trip count, clamped index, zero for the tail: same sum order, same bits): the
    // run-time loop was 6 dependent LDS round trips in front of every consumer epilogue
    lds_cf32 q = (lds_cf32)(aux) + row * r.tiles;
    constexpr int MAXIT = kAuxMaxTiles / TPR;
    float v[MAXIT];
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
      const int t = part + i * TPR;
      v[i] = q[t < r.tiles ? t : part];
    }
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) acc += (part + i * TPR < r.tiles) ? v[i] : 0.f;
  } else {
    const float* q = r.ssq + (size_t)(m0 + row) * r.tiles;
    for (int t = part; t < r.tiles; t += TPR) acc += q[t];
  }
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (part == 0) rs[row] = 1.0f / sqrtf(acc * r.inv_d + 1e-6f);
}

// `stats_done`: the caller already ran tile_rstd_compute ahead of its own barrier (the LDS-DMA kernel
// does, next to its accumulator -> slab stores: one block barrier instead of two).
template <int BM>
__device__ __forceinline__ BiasRow tile_rstd(const RowScale& r, float* rs, int m0, int n0, int tid,
                                             const char* aux, bool stats_done = false) {
  if (!stats_done) {
    tile_rstd_compute<BM>(r, rs, m0, tid, aux);
    __syncthreads();
  }
  BiasRow b;
  if (r.bias) {
    if (!aux_present(aux)) __builtin_trap();   // the bias row lives in the aux LDS region (every product kernel has one)
    b.present = true;
    b.l = (lds_cf32)(aux + rowscale_ssq_bytes<BM>(r.tiles));
  }
  return b;
}

// C (row-major bf16 planes) = acc [* rstd[m] + bias[n]]
template <int NP>
struct EpiStoreH16 {
  h16_t* out[2];
  int ldc;
  RowScale rsc;
  template <int BM, int BN> static constexpr int aux_bytes() { return rowscale_aux_bytes<BM>(); }
  template <int BM, int BN>
  __device__ void prefetch(char* aux, int m0, int n0, int wave, int lane) const {
    rowscale_prefetch<BM, BN>(rsc, aux, m0, n0, wave, lane);
  }
  template <int BM, int LD>
  __device__ void stats(float* s0, int m0, int tid, const char* aux) const {
    if (rsc.ssq) tile_rstd_compute<BM>(rsc, s0 + BM * LD, m0, tid, aux);
  }
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    float* rs = s0 + BM * LD;
    BiasRow bias;
    if (rsc.ssq) bias = tile_rstd<BM>(rsc, rs, m0, n0, tid, aux, stats_done);
    RangeCheck rc;
    rowscale_variants(rsc.ssq != nullptr, bias.present, [&](auto mode) {
      MSD_EPI_ITEMS(BM * BN / 8, item) {
        const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
        float v[8];
        tile_row8<LD>(s0, m, n, v);
        rowscale8<decltype(mode)::value>(v, decltype(mode)::value ? rs[m] : 1.f, bias, n);
        store_h16x8<NP>(out, (size_t)(m0 + m) * ldc + n0 + n, v, rc);
      }
    });
    rc.commit(sf.p, sf.tag);
  }
};

// Fused QKV (or K|V) projection: columns [0, v_start) -> row-major bf16 `qk`
// [M, ld_qk]; columns [v_start, N) -> V^T planes [seg][N - v_start][vt_ld] with the
// key axis permuted per 16 (seg = m / seg_len, key = m % seg_len).
template <int NP>
struct EpiQKV {
  h16_t* qk[2];
  h16_t* vt[2];
  int ld_qk, v_start, seg_len, vt_ld, vt_rows;
  RowScale rsc;
  template <int BM, int BN> static constexpr int aux_bytes() { return rowscale_aux_bytes<BM>(); }
  template <int BM, int BN>
  __device__ void prefetch(char* aux, int m0, int n0, int wave, int lane) const {
    rowscale_prefetch<BM, BN>(rsc, aux, m0, n0, wave, lane);
  }
  template <int BM, int LD>
  __device__ void stats(float* s0, int m0, int tid, const char* aux) const {
    if (rsc.ssq) tile_rstd_compute<BM>(rsc, s0 + BM * LD, m0, tid, aux);
  }
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    float* rs = s0 + BM * LD;
    BiasRow bias;
    if (rsc.ssq) bias = tile_rstd<BM>(rsc, rs, m0, n0, tid, aux, stats_done);
    RangeCheck rc;
    if (n0 < v_start) {
      rowscale_variants(rsc.ssq != nullptr, bias.present, [&](auto mode) {
        MSD_EPI_ITEMS(BM * BN / 8, item) {
          const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
          float v[8];
          tile_row8<LD>(s0, m, n, v);
          rowscale8<decltype(mode)::value>(v, decltype(mode)::value ? rs[m] : 1.f, bias, n);
          store_h16x8<NP>(qk, (size_t)(m0 + m) * ld_qk + n0 + n, v, rc);
        }
      });
    } else {
      // transposed items: (column n, 8 rows = keys of one 16-key group).  The key permutation inside a group puts keys
      // {0-3, 8-11} on positions 0 .. 7 and keys {4-7, 12-15} on positions 8 .. 15 (vt_perm16), so an item that takes
      // rows 4h + {0..3} and 8 + 4h + {0..3} of its group writes ONE 16-byte run per plane (round 4; before: 8
      // consecutive rows = two 8-byte runs per plane, and the blocks of the V columns were the launch's last to leave:
      // slab -> stores issued 1.46 us against 0.71 on the Q / K columns, profiles/r04z_phase_times.txt)
      const bool folded = rsc.ssq != nullptr;
      static_assert(BM % 16 == 0, "whole 16-key groups");
      MSD_EPI_ITEMS(BM * BN / 8, item) {
        const int n = item / (BM / 8), q = item % (BM / 8), kg = q >> 1, hh = q & 1;
        const int r0 = kg * 16 + 4 * hh;            // rows r0 .. r0+3 and r0+8 .. r0+11
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = s0[(r0 + e) * LD + n];
          v[4 + e] = s0[(r0 + 8 + e) * LD + n];
        }
        if (folded) {
          const float4 ra = *reinterpret_cast<const float4*>(rs + r0), rb = *reinterpret_cast<const float4*>(rs + r0 + 8);
          const float bn = bias.present ? bias.at(n) : 0.f;
          v[0] = v[0] * ra.x + bn; v[1] = v[1] * ra.y + bn; v[2] = v[2] * ra.z + bn; v[3] = v[3] * ra.w + bn;
          v[4] = v[4] * rb.x + bn; v[5] = v[5] * rb.y + bn; v[6] = v[6] * rb.z + bn; v[7] = v[7] * rb.w + bn;
        }
        const int mg = m0 + kg * 16, seg = mg / seg_len, key = mg % seg_len;   // (a 16-group never straddles a segment)
        const size_t off = ((size_t)seg * vt_rows + (n0 + n - v_start)) * vt_ld + key + 8 * hh;
        store_h16x8<NP>(vt, off, v, rc);
      }
    }
    rc.commit(sf.p, sf.tag);
  }
};

// x[M, ldx] (fp32 residual stream) += acc
struct EpiResidual {
  float* x;
  int ldx;
  template <int BM, int BN> static constexpr int aux_bytes() { return 0; }
  template <int BM, int BN>
  __device__ void prefetch(char*, int, int, int, int) const {}
  template <int BM, int LD>
  __device__ void stats(float*, int, int, const char*) const {}
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    for (int item = tid; item < BM * BN / 8; item += 256) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
      float v[8];
      tile_row8<LD>(s0, m, n, v);
      float4* px = reinterpret_cast<float4*>(x + (size_t)(m0 + m) * ldx + n0 + n);
      float4 a = px[0], b = px[1];
      a.x += v[0]; a.y += v[1]; a.z += v[2]; a.w += v[3];
      b.x += v[4]; b.y += v[5]; b.z += v[6]; b.w += v[7];
      px[0] = a; px[1] = b;
    }
  }
};

// w = v (.) g as ROUNDED float32 products.  The planes written from w are hi = r16(w), lo = r16(w - hi): left to the
// compiler, the product is contracted into that subtraction as an FMA in some instantiations of an epilogue and not in
// others -- the first version of the duplicating epilogue below wrote y planes one ulp off the one-pass form's (found
// with tools/diag/dedup_diff.py on a debug build with stop points, profiles/r05c_dedup_diff.log).  Under
// `fp contract(off)` these multiplications carry no contract flag and cannot be fused (HIP's __fmul_rn is a plain
// `x * y` and would be), so every form of the residual epilogue writes the same bits.
__device__ __forceinline__ void gain8(const float (&v)[8], float4 g0, float4 g1, float (&w)[8]) {
#pragma clang fp contract(off)
  w[0] = v[0] * g0.x; w[1] = v[1] * g0.y; w[2] = v[2] * g0.z; w[3] = v[3] * g0.w;
  w[4] = v[4] * g1.x; w[5] = v[5] * g1.y; w[6] = v[6] * g1.z; w[7] = v[7] * g1.w;
}

// Residual add that also PRODUCES the folded-norm inputs of the next projection:
//   x += acc ;  ssq[m][col/32] = sum over each 32-column group of x^2 ;
//   y = x (.) g  as bf16 planes, g = g_lo for rows < split_row, g_hi otherwise
//   (g pointer = base + step * step_stride; a null base skips y for that row range).
// DUP (layer 0 of a CFG step): the launch runs on the conditional pass's rows only and every row r is
// ALSO written as row r + dup_rows -- the unconditional pass starts from the same z, the same FiLM and the same
// self-attention (models/diffusion/models.py:373-386, network.py:174-193), so up to here its rows are bit-for-bit the
// conditional ones: x and ssq are copied, y[r] = x (.) g_lo and y[r + dup_rows] = x (.) g_hi (split_row is not used).
// Y2 (the folded cross-attention query projection, msd_api.hip decoder_layers / DESIGN.md 5 S6): rows < y2_rows are ALSO
// written as y2 = x (.) g2 -- g2 = the next cross-attention norm's plain scale, not step-indexed -- the A operand of the
// half of that projection which does not wait for the self-attention block (narrow tiles only).
template <int NP, bool DUP = false, bool Y2 = false>
struct EpiResidualNorm {
  float* x;
  int ldx;
  h16_t* y[2];
  float* ssq;
  int tiles;
  const float* g_lo; int g_lo_stride;
  const float* g_hi; int g_hi_stride;
  int split_row;
  const int* step_ptr;
  int dup_rows = 0;   // DUP only: distance to the second copy of a row (= rows of one pass)
  h16_t* y2[2] = {nullptr, nullptr};   // Y2 only
  const float* g2 = nullptr;
  int y2_rows = 0;
  // aux layout (BN == 32 or 48): [x tile BM x BN fp32][g_lo slice, 1 KiB][g_hi slice, 1 KiB][Y2: g2 slice, 1 KiB]
  template <int BN> static constexpr bool narrow() { return BN == 32 || BN == 48; }
  template <int BM, int BN> static constexpr int aux_bytes() { return narrow<BN>() ? BM * BN * 4 + 2048 + (Y2 ? 1024 : 0) : 0; }
  template <int BM, int LD>
  __device__ void stats(float*, int, int, const char*) const {}
  template <int BM, int BN>
  __device__ void prefetch(char* aux, int m0, int n0, int wave, int lane) const {
    if (!narrow<BN>()) return;
    // the residual tile, row-major: 16-byte chunk id = 64 i + lane is chunk id % (BN/4) of row id / (BN/4)
    // (BN == 32: lane (r = lane>>3, c = lane&7) fetches 16 B of row 8i + r)
    constexpr int CPR = BN / 4;
    static_assert((BM * CPR) % 64 == 0, "whole DMA instructions");
    for (int i = wave; i < BM * CPR / 64; i += 4) {
      const int id = i * 64 + lane, r = id / CPR, ch = id % CPR;
      __builtin_amdgcn_global_load_lds((aux_gptr_t)(x + (size_t)(m0 + r) * ldx + n0 + ch * 4),
                                       lds_ptr_of(aux + i * 1024), 16, 0, 0);
    }
    // (only the two waves that fetch a step-indexed row read the scan index)
    if (g_lo && wave == 2) aux_dma_row(g_lo + (size_t)scan_index(step_ptr) * g_lo_stride + n0, aux + BM * BN * 4, BN * 4, lane);
    if (g_hi && wave == 3) aux_dma_row(g_hi + (size_t)scan_index(step_ptr) * g_hi_stride + n0, aux + BM * BN * 4 + 1024, BN * 4, lane);
    if constexpr (Y2) {
      if (g2 && wave == 1) aux_dma_row(g2 + n0, aux + BM * BN * 4 + 2048, BN * 4, lane);
    }
  }
  // 32 x 48 tiles: 8 lanes per row, 6 of them with 8 columns each; ONE partial sum of squares per row and tile, in
  // slot n0 / 48 of the row's `tiles` (= D / 32) slots -- the D / 48 slots a row gets this way are fewer than `tiles`,
  // so the tiles of the first columns also zero one of the unused slots each: a consumer adds all `tiles` slots of a
  // row whichever launch produced that row (rows of one x are produced by launches of different tile widths).
  template <int BM, int LD>
  __device__ void run48(float* s0, int m0, int n0, int tid, const char* aux, SatFlag sf) const {
    constexpr int BN = 48;
    static_assert(BM * 8 == 256, "one 8-lane group per row");
    static_assert(!DUP, "the duplicating form runs on 32-column (or batched 96-column) tiles");
    typedef const __attribute__((address_space(3))) f32x4* lds_cf32x4;
    lds_cf32x4 xs = (lds_cf32x4)(aux);
    lds_cf32x4 gl = (lds_cf32x4)(aux + BM * BN * 4), gh = (lds_cf32x4)(aux + BM * BN * 4 + 1024);
    const int m = tid >> 3, c = tid & 7;
    const bool act = c < BN / 8;
    const int n = act ? c * 8 : 0;
    const int row = m0 + m, col = n0 + n;
    float v[8];
    tile_row8<LD>(s0, m, n, v);
    const f32x4 a = xs[(m * BN + n) / 4], b = xs[(m * BN + n) / 4 + 1];
    v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
    v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    float4* px = reinterpret_cast<float4*>(x + (size_t)row * ldx + col);
    if (act) {
      px[0] = make_float4(v[0], v[1], v[2], v[3]);
      px[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sq += v[e] * v[e];
    sq = act ? sq : 0.f;
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    const int slot = n0 / BN, used = ldx / BN;   // (ldx == D: the residual stream is [rows][D])
    if (c == 0) ssq[(size_t)row * tiles + slot] = sq;
    if (c == 6 && used + slot < tiles) ssq[(size_t)row * tiles + used + slot] = 0.f;
    RangeCheck rc;
    const bool lo_rows = row < split_row;
    if (act && (lo_rows ? (g_lo != nullptr) : (g_hi != nullptr))) {
      const f32x4 g0 = lo_rows ? gl[n / 4] : gh[n / 4], g1 = lo_rows ? gl[n / 4 + 1] : gh[n / 4 + 1];
      float w[8];
      gain8(v, make_float4(g0[0], g0[1], g0[2], g0[3]), make_float4(g1[0], g1[1], g1[2], g1[3]), w);
      store_h16x8<NP>(y, (size_t)row * ldx + col, w, rc);
    }
    if constexpr (Y2) {
      if (act && g2 != nullptr && row < y2_rows) {
        lds_cf32x4 gc = (lds_cf32x4)(aux + BM * BN * 4 + 2048);
        const f32x4 c0 = gc[n / 4], c1 = gc[n / 4 + 1];
        float w2[8];
        gain8(v, make_float4(c0[0], c0[1], c0[2], c0[3]), make_float4(c1[0], c1[1], c1[2], c1[3]), w2);
        store_h16x8<NP>(y2, (size_t)row * ldx + col, w2, rc);
      }
    }
    rc.commit(sf.p, sf.tag);
  }
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    if constexpr (BN == 48) {
      return run48<BM, LD>(s0, m0, n0, tid, aux, sf);
    } else {
    static_assert(BN % 32 == 0, "partial sums of squares are per 32-column group (tiles = D / 32)");
    static_assert(!Y2 || !DUP, "the duplicating form never feeds a folded query projection");
    // (Y2 on a tile wider than 32 columns -- the batched path's -- writes no second pair: the launcher folds the query
    // projection only where the producer runs on narrow tiles, msd_api.hip fold_cross_q)
    const bool pre = BN == 32 && aux_present(aux);
    const int step = pre ? 0 : *step_ptr;
    RangeCheck rc;
    // one tile-element group (8 columns of one row); LX / LG fetch the residual and the gain
    auto body = [&](int item, auto LX, auto LG) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;   // BN/8 consecutive lanes share a row
      float v[8];
      tile_row8<LD>(s0, m, n, v);
      const int row = m0 + m, col = n0 + n;
      float4* px = reinterpret_cast<float4*>(x + (size_t)row * ldx + col);
      float4 a, b;
      LX(m, n, px, a, b);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      px[0] = make_float4(v[0], v[1], v[2], v[3]);
      px[1] = make_float4(v[4], v[5], v[6], v[7]);
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sq += v[e] * v[e];
      sq += __shfl_xor(sq, 1, 64);   // 4 consecutive lanes = one 32-column group of one row
      sq += __shfl_xor(sq, 2, 64);
      if ((item & 3) == 0) ssq[(size_t)row * tiles + col / 32] = sq;
      if constexpr (DUP) {
        const size_t row2 = (size_t)row + dup_rows;
        float4* px2 = reinterpret_cast<float4*>(x + row2 * ldx + col);
        px2[0] = make_float4(v[0], v[1], v[2], v[3]);
        px2[1] = make_float4(v[4], v[5], v[6], v[7]);
        if ((item & 3) == 0) ssq[row2 * tiles + col / 32] = sq;
        float4 g0, g1, h0, h1;
        LG(true, n, col, g0, g1);
        LG(false, n, col, h0, h1);
        // (gain8: both copies must carry the bits the one-pass form computes)
        float w[8], u[8];
        gain8(v, g0, g1, w);
        gain8(v, h0, h1, u);
        store_h16x8<NP>(y, (size_t)row * ldx + col, w, rc);
        store_h16x8<NP>(y, row2 * ldx + col, u, rc);
      } else {
        const bool lo_rows = row < split_row;
        if (lo_rows ? (g_lo != nullptr) : (g_hi != nullptr)) {
          float4 g0, g1;
          LG(lo_rows, n, col, g0, g1);
          float w[8];
          gain8(v, g0, g1, w);
          store_h16x8<NP>(y, (size_t)row * ldx + col, w, rc);
        }
        if constexpr (Y2 && BN == 32) {
          if (g2 != nullptr && row < y2_rows) {
            typedef const __attribute__((address_space(3))) f32x4* lds_g2_t;
            lds_g2_t gc = (lds_g2_t)(aux + BM * 128 + 2048);
            const f32x4 c0 = gc[n / 4], c1 = gc[n / 4 + 1];
            float w2[8];
            gain8(v, make_float4(c0[0], c0[1], c0[2], c0[3]), make_float4(c1[0], c1[1], c1[2], c1[3]), w2);
            store_h16x8<NP>(y2, (size_t)row * ldx + col, w2, rc);
          }
        }
      }
    };
    if (pre) {   // operands prefetched into the aux LDS region: explicit LDS pointers (ds_read)
      typedef const __attribute__((address_space(3))) f32x4* lds_cf32x4;   // native vector: loadable from LDS
      auto f4 = [](f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
      lds_cf32x4 xs = (lds_cf32x4)(aux);
      lds_cf32x4 gl = (lds_cf32x4)(aux + BM * 128), gh = (lds_cf32x4)(aux + BM * 128 + 1024);
      for (int item = tid; item < BM * BN / 8; item += 256)
        body(item,
             [&](int m, int n, float4*, float4& a, float4& b) { a = f4(xs[(m * BN + n) / 4]); b = f4(xs[(m * BN + n) / 4 + 1]); },
             [&](bool lo_rows, int n, int, float4& g0, float4& g1) {
               g0 = f4(lo_rows ? gl[n / 4] : gh[n / 4]);
               g1 = f4(lo_rows ? gl[n / 4 + 1] : gh[n / 4 + 1]);
             });
    } else {
      // The batched path's 128 x 96 tiles come here (no aux copy of the residual tile: it would not fit behind the
      // ring).  Round 3's form loaded an item's residual and gain rows INSIDE the item loop: a rolled loop whose every
      // iteration waited twice with vmcnt(0) -- for its own loads and, vmcnt counting stores too, for the previous
      // iteration's stores -- 12 dependent memory round trips per thread, the 6.9 us "epilogue" of the phase stamps
      // (profiles/r03p_phase_times_b8.txt).  Now every load of the thread is in flight before the first item is
      // computed (up to 24 x 16 bytes per thread; the accumulators are dead by now: they live in the slab).
      const float* glo = g_lo ? g_lo + (size_t)step * g_lo_stride : nullptr;
      const float* ghi = g_hi ? g_hi + (size_t)step * g_hi_stride : nullptr;
      constexpr int ITEMS = BM * BN / 8, PER = (ITEMS + 255) / 256;
      f32x4 xa[PER], xb[PER], ga[PER], gb[PER], ha[DUP ? PER : 1], hb[DUP ? PER : 1];
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int item = tid + it * 256;
        if (ITEMS % 256 == 0 || item < ITEMS) {
          const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
          const int row = m0 + m, col = n0 + n;
          const f32x4* px = reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + col);
          xa[it] = px[0]; xb[it] = px[1];
          if constexpr (DUP) {   // both gain rows of every item (the launcher sets both)
            ga[it] = *reinterpret_cast<const f32x4*>(glo + col); gb[it] = *reinterpret_cast<const f32x4*>(glo + col + 4);
            ha[it] = *reinterpret_cast<const f32x4*>(ghi + col); hb[it] = *reinterpret_cast<const f32x4*>(ghi + col + 4);
          } else {
            const float* g = row < split_row ? glo : ghi;
            if (g != nullptr) {
              ga[it] = *reinterpret_cast<const f32x4*>(g + col);
              gb[it] = *reinterpret_cast<const f32x4*>(g + col + 4);
            }
          }
        }
      }
      auto f4 = [](f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int item = tid + it * 256;
        if (ITEMS % 256 == 0 || item < ITEMS)
          body(item,
               [&](int, int, float4*, float4& a, float4& b2) { a = f4(xa[it]); b2 = f4(xb[it]); },
               [&](bool lo, int, int, float4& g0, float4& g1) {
                 if (DUP && !lo) { g0 = f4(ha[DUP ? it : 0]); g1 = f4(hb[DUP ? it : 0]); }
                 else { g0 = f4(ga[it]); g1 = f4(gb[it]); }
               });
      }
    }
    rc.commit(sf.p, sf.tag);
    }
  }
};

// Decoder input (network.py:420-427): x[pass][m][:] = z[m] . W_in + pos[m % T] for every
// pass (the conditional and the unconditional CFG pass start from the same rows), plus
// the folded-norm inputs of layer 0's self-attention projection (y = x (.) g, ssq).
template <int NP>
struct EpiInProj {
  float* x;
  int ldx;
  const float* pos;
  int T, pass_rows, passes;
  h16_t* y[2];
  float* ssq;
  int tiles;
  const float* g; int g_stride;
  const int* step_ptr;
  int* step_copy = nullptr;   // = step_ptr when the sampler follows in the same step
  // optional: y2 = x (.) g2 for the first pass's rows -- layer 0's folded cross-attention query projection (EpiResidualNorm Y2)
  h16_t* y2[2] = {nullptr, nullptr};
  const float* g2 = nullptr;
  template <int BM, int BN> static constexpr int aux_bytes() { return 0; }
  template <int BM, int BN>
  __device__ void prefetch(char*, int, int, int, int) const {}
  template <int BM, int LD>
  __device__ void stats(float*, int, int, const char*) const {}
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    static_assert(BN == 64 || BN == 32, "partial sums of squares are per BN-column tile");
    const int step = *step_ptr;
    // first kernel of the DDPM step: publish the index in slot 1 for the sampler (elementwise.h),
    // which then owns slot 0 and decrements it without a separate launch
    if (step_copy && m0 == 0 && n0 == 0 && tid == 0) step_copy[1] = step;
    const float* gs = g + (size_t)step * g_stride;
    RangeCheck rc;
    for (int item = tid; item < BM * BN / 8; item += 256) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
      float v[8];
      tile_row8<LD>(s0, m, n, v);
      const int row = m0 + m, col = n0 + n;
      const float4 p0 = *reinterpret_cast<const float4*>(pos + (size_t)(row % T) * ldx + col);
      const float4 p1 = *reinterpret_cast<const float4*>(pos + (size_t)(row % T) * ldx + col + 4);
      v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
      v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sq += v[e] * v[e];
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      if (BN == 64) sq += __shfl_xor(sq, 4, 64);
      const float4 g0 = *reinterpret_cast<const float4*>(gs + col);
      const float4 g1 = *reinterpret_cast<const float4*>(gs + col + 4);
      float w[8] = {v[0] * g0.x, v[1] * g0.y, v[2] * g0.z, v[3] * g0.w,
                    v[4] * g1.x, v[5] * g1.y, v[6] * g1.z, v[7] * g1.w};
      for (int ps = 0; ps < passes; ++ps) {
        const size_t r = (size_t)ps * pass_rows + row;
        float4* px = reinterpret_cast<float4*>(x + r * ldx + col);
        px[0] = make_float4(v[0], v[1], v[2], v[3]);
        px[1] = make_float4(v[4], v[5], v[6], v[7]);
        if ((item % (BN / 8)) == 0) ssq[r * tiles + n0 / BN] = sq;
        store_h16x8<NP>(y, r * ldx + col, w, rc);
      }
      if (g2 != nullptr) {
        const float4 c0 = *reinterpret_cast<const float4*>(g2 + col), c1 = *reinterpret_cast<const float4*>(g2 + col + 4);
        float w2[8];
        gain8(v, c0, c1, w2);
        store_h16x8<NP>(y2, (size_t)row * ldx + col, w2, rc);
      }
    }
    rc.commit(sf.p, sf.tag);
  }
};

// out fp32 [M, ldc] = acc [* rstd[m] + bias[n]]
struct EpiStoreF32 {
  float* out;
  int ldc;
  RowScale rsc;
  template <int BM, int BN> static constexpr int aux_bytes() { return rowscale_aux_bytes<BM>(); }
  template <int BM, int BN>
  __device__ void prefetch(char* aux, int m0, int n0, int wave, int lane) const {
    rowscale_prefetch<BM, BN>(rsc, aux, m0, n0, wave, lane);
  }
  template <int BM, int LD>
  __device__ void stats(float* s0, int m0, int tid, const char* aux) const {
    if (rsc.ssq) tile_rstd_compute<BM>(rsc, s0 + BM * LD, m0, tid, aux);
  }
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    float* rs = s0 + BM * LD;
    BiasRow bias;
    if (rsc.ssq) bias = tile_rstd<BM>(rsc, rs, m0, n0, tid, aux, stats_done);
    rowscale_variants(rsc.ssq != nullptr, bias.present, [&](auto mode) {
      MSD_EPI_ITEMS(BM * BN / 8, item) {
        const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
        float v[8];
        tile_row8<LD>(s0, m, n, v);
        rowscale8<decltype(mode)::value>(v, decltype(mode)::value ? rs[m] : 1.f, bias, n);
        float4* po = reinterpret_cast<float4*>(out + (size_t)(m0 + m) * ldc + n0 + n);
        po[0] = make_float4(v[0], v[1], v[2], v[3]);
        po[1] = make_float4(v[4], v[5], v[6], v[7]);
      }
    });
  }
};

// Gated GELU (layers.py:483-497 with activations ('gelu','linear')).  The packed
// weight interleaves wi_0 / wi_1 in blocks of 16 output columns: packed columns
// [32g, 32g+16) are wi_0 columns [16g, 16g+16), [32g+16, 32g+32) the matching wi_1.
//   g_out[m][n0/2 + j] = gelu(tile[m][pc(j)]) * tile[m][pc(j) + 16]
template <int NP>
struct EpiGeglu {
  h16_t* out[2];
  int ldc;  // = F
  RowScale rsc;  // bias table is indexed by PACKED column
  template <int BM, int BN> static constexpr int aux_bytes() { return rowscale_aux_bytes<BM>(); }
  template <int BM, int BN>
  __device__ void prefetch(char* aux, int m0, int n0, int wave, int lane) const {
    rowscale_prefetch<BM, BN>(rsc, aux, m0, n0, wave, lane);
  }
  template <int BM, int LD>
  __device__ void stats(float* s0, int m0, int tid, const char* aux) const {
    if (rsc.ssq) tile_rstd_compute<BM>(rsc, s0 + BM * LD, m0, tid, aux);
  }
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    static_assert(BN % 32 == 0, "gated epilogue needs whole wi_0/wi_1 groups");
    constexpr int OUT_N = BN / 2;  // output columns per tile
    float* rs = s0 + BM * LD;
    BiasRow bias;
    if (rsc.ssq) bias = tile_rstd<BM>(rsc, rs, m0, n0, tid, aux, stats_done);
    RangeCheck rc;
    rowscale_variants(rsc.ssq != nullptr, bias.present, [&](auto mode) {
      MSD_EPI_ITEMS(BM * OUT_N / 8, item) {
        const int m = item / (OUT_N / 8), j = (item % (OUT_N / 8)) * 8;  // 8 output cols j..j+7
        const int pc = (j / 16) * 32 + (j % 16);
        float a[8], b[8], v[8];
        tile_row8<LD>(s0, m, pc, a);
        tile_row8<LD>(s0, m, pc + 16, b);
        const float r = decltype(mode)::value ? rs[m] : 1.f;
        rowscale8<decltype(mode)::value>(a, r, bias, pc);
        rowscale8<decltype(mode)::value>(b, r, bias, pc + 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(a[e]) * b[e];
        store_h16x8<NP>(out, (size_t)(m0 + m) * ldc + n0 / 2 + j, v, rc);
      }
    });
    rc.commit(sf.p, sf.tag);
  }
};

// Which epilogues ever carry a weight prefetch (one kernel instantiation per PF value): the encoders' plain residual
// and the fp32 store never do, so their PF = 1 twins are not built.
template <class Epi> struct epi_may_prefetch : std::true_type {};
template <> struct epi_may_prefetch<EpiResidual> : std::false_type {};
template <> struct epi_may_prefetch<EpiStoreF32> : std::false_type {};

template <int NP, int BM, int BN, int NS, class Epi>
constexpr int gemm_h16_dma_smem() { return NS * NP * (BM + BN) * 128 + Epi::template aux_bytes<BM, BN>(); }

template <int NP, int BM, int BN, int NS, class Epi, int PF>
inline hipError_t gemm_h16_dma_prepare_one() {
  constexpr int smem = gemm_h16_dma_smem<NP, BM, BN, NS, Epi>();
  if (smem < 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_dma_kernel<NP, BM, BN, NS, Epi, PF>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

// one-time opt-in to > 64 KiB dynamic LDS; call for every instantiation OUTSIDE stream capture
template <int NP, int BM, int BN, int NS, class Epi>
inline hipError_t gemm_h16_dma_prepare() {
  hipError_t e = gemm_h16_dma_prepare_one<NP, BM, BN, NS, Epi, 0>(), r;
  if constexpr (NP == 2 && epi_may_prefetch<Epi>::value) {   // the single-plane mode never prefetches
    if ((r = gemm_h16_dma_prepare_one<NP, BM, BN, NS, Epi, 1>()) != hipSuccess) e = r;
  }
  return e;
}

template <int NP, int BM, int BN, int NS, class Epi>
inline hipError_t launch_gemm_h16_dma(const GemmParams& p_in, const Epi& epi, hipStream_t stream) {
  constexpr int smem = gemm_h16_dma_smem<NP, BM, BN, NS, Epi>();
  static const hipError_t attr = gemm_h16_dma_prepare<NP, BM, BN, NS, Epi>();
  if (attr != hipSuccess) return attr;
  GemmParams p = p_in;
  p.map.fill(p.M, p.N, BM, BN, p.xcd_rows);
  const int rx = p.xcd_rows, cx = 8 / rx;
  const int grid = 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / BM + rx - 1) / rx);
  // one kernel instantiation per number of prefetch targets (single path: see prefetch_weights)
#define MSD_LAUNCH_PF(PF_) \
  hipLaunchKernelGGL((gemm_h16_dma_kernel<NP, BM, BN, NS, Epi, PF_>), dim3(grid), dim3(256 + pf_threads(PF_)), smem, stream, p, epi)
  const int npf = NP == 2 ? prefetch_kind(p.pf) : 0;
  if constexpr (NP == 2 && epi_may_prefetch<Epi>::value) {
    if (npf >= 1) MSD_LAUNCH_PF(1);   // product: the first target only
    else MSD_LAUNCH_PF(0);
  } else {
    MSD_LAUNCH_PF(0);
  }
#undef MSD_LAUNCH_PF
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// Batched songs: the gated-MLP input projection as a PERSISTENT tile loop with a register epilogue (round 6; VERDICT r05
// next #3).  At >= 4 songs per handle a CU runs 4 .. 8 of the 128 x 128 tiles back to back and a tile lives 17.4 us of
// which 11.8 are its main loop (profiles/r05z_phase_times_b8.txt): 2.0 issuing the two ring stages (the ingest rate),
// 0.5 waiting for the first, 1.1 storing the accumulators to the LDS slab, 1.9 in the epilogue.  The slab lives in the
// ring, so nothing of the next tile could start before the epilogue had read it.  Here
//   * the epilogue works on the accumulator REGISTERS: in the packed wi_0 | wi_1 column order a lane's accumulators
//     (i, 2q) and (i, 2q + 1) are the gelu input and the gate of the SAME four output columns -- no slab, no transpose;
//   * the block stays alive over its tiles (virtual block v = blockIdx + k * gridDim: the XCD-aware tile map of the
//     plain launch, same XCD for every k) and issues tile k + 1's two ring stages right behind tile k's main loop, in
//     front of tile k's epilogue: the stages land while the epilogue computes and stores.
// vmcnt counts the epilogue's stores, which are younger than those DMAs: the next tile starts with vmcnt(0) (its stages
// have had the whole epilogue to land) and keeps the loop's counted waits from there.
// ----------------------------------------------------------------------------

// The first NS K-tiles of tile (bm, bn) into the ring, 2 x 2 wave layout: the prologue of gemm_tile as a function, its
// NS * NP * (BM + BN) / 32 LDS-DMA instructions per wave addressable one by one (Q0 <= q < Q1) so that a caller can
// spread them between other work: a wave cannot issue past a DMA the address unit has not taken yet, and a whole
// prologue is ~2 us of that (the "entry -> prologue DMAs issued" of the phase stamps).
template <int NP, int BM, int BN, int NS>
struct GemmTileIssue {
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = NP * (A_BYTES + B_BYTES);
  static constexpr int A_LD = BM / 32, B_LD = BN / 32, PER_PLANE = A_LD + B_LD, PER_STAGE = NP * PER_PLANE, COUNT = NS * PER_STAGE;
  const h16_t* ga[NP];
  const h16_t* gb[NP];
  size_t a_step, b_step;
  char* smem;
  int wave;
  // `as_wave` >= 0: issue the share of compute wave `as_wave` (a loader wave standing in for the four of them)
  __device__ __forceinline__ GemmTileIssue(const GemmParams& p, int bm, int bn, char* smem_, int as_wave = -1) : smem(smem_) {
    const int lane = threadIdx.x & 63;
    wave = as_wave >= 0 ? as_wave : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r8 = lane >> 3, csrc = (lane & 7) ^ r8;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      ga[pl] = p.A[pl] + (size_t)(bm * BM + wave * (BM / 4) + r8) * p.lda + csrc * 8;
      gb[pl] = p.B[pl] + (size_t)(bn * BN + wave * (BN / 4) + r8) * p.ldb + csrc * 8;
    }
    a_step = (size_t)8 * p.lda; b_step = (size_t)8 * p.ldb;
  }
  template <int Q0, int Q1>
  __device__ __forceinline__ void issue() const {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
      if (q >= COUNT) break;
      const int st = q / PER_STAGE, pl = (q % PER_STAGE) / PER_PLANE, r = q % PER_PLANE;
      char* base = smem + st * STAGE_BYTES;
      if (r < A_LD)
        __builtin_amdgcn_global_load_lds((gptr_t)(ga[pl] + r * a_step + st * kGemmBK),
                                         (lptr_t)(base + pl * A_BYTES + (wave * (BM / 4) + 8 * r) * 128), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((gptr_t)(gb[pl] + (r - A_LD) * b_step + st * kGemmBK),
                                         (lptr_t)(base + NP * A_BYTES + pl * B_BYTES + (wave * (BN / 4) + 8 * (r - A_LD)) * 128), 16, 0, 0);
    }
  }
};

template <int NP>
__device__ __forceinline__ void store_h16x4(h16_t* const* planes, size_t off, const float v[4], RangeCheck& rc) {
  uint32_t wh[2], wl[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rc.see(v[2 * e], v[2 * e + 1]);
    if (NP == 2) split2_h16(v[2 * e], v[2 * e + 1], wh[e], wl[e]);
    else wh[e] = cvt2_h16(v[2 * e], v[2 * e + 1]);
  }
  *reinterpret_cast<uint2*>(planes[0] + off) = make_uint2(wh[0], wh[1]);
  if (NP == 2) *reinterpret_cast<uint2*>(planes[1] + off) = make_uint2(wl[0], wl[1]);
}

template <int NP, int BM, int BN, int NS>
constexpr int gemm_h16_geglu_persist_smem() { return NS * NP * (BM + BN) * 128 + rowscale_aux_bytes<BM>() + BM * 4; }   // ring, aux rows, rstd

// `vblocks`: blocks of the plain launch's grid (GemmParams::TileMap: 8 x column groups x row groups; some map to no
// tile).  The launcher folds the norm (rsc.ssq) and has a step-indexed bias row (rsc.bias): the decoder's MLP blocks.
// The row statistics are per ROW tile: a block whose successive tiles keep their row tile (grid / 8 a multiple of the
// row tiles per XCD row group: every shipped batch) computes them -- and fetches their partial sums -- once.
//
// What was measured on the way (8 songs per handle, same-process A/B against the per-tile launch, stamps of a debug
// build: profiles/r06i ... r06l_persist_ab_b*.log, r06l_phase_times_b8.txt):
//   * this form (the next tile's stages issued as ONE block behind the main loop, in front of the epilogue): tile
//     lifetime 18.97 -> 17.6 us, -0.8 ... -1.9 % end to end.  The issue itself blocks the compute waves for ~2 us (a wave
//     stalls at a DMA the address unit has not taken, and 128 KiB are 2 us of the CU's ingest rate), so what overlaps
//     the epilogue's arithmetic is the LANDING, not the issue;
//   * the same DMAs spread between the epilogue's eight output groups: the epilogue 1.9 -> 3.45 us, no better;
//   * a fifth, LOADER wave that issues them while the four compute waves run the epilogue (it mirrors their barriers):
//     one wave issues a DMA per ~65 clocks -- 128 of them are 5 us -- and the next tile waited 2.9 us for it: +1 ... +1.5 %.
// A tile is 768 KiB through a CU that ingests ~35 B/clk: 11.9 us of its 17.6 at these clocks; the rest is the per-K-tile
// synchronisation of a two-stage ring (the loop's 12.9 us against 10.3 of MFMA issue), which no epilogue overlap touches.
// (One song, one 64 x 128 tile per block through this kernel -- the register epilogue without a tile loop: +0.5 % step,
// its 8-byte stores cover 32 contiguous bytes per row where the slab's cover 128; not used.)
template <int NP, int BM, int BN, int NS, int PF = kPfNone>
__global__ void __launch_bounds__(256 + pf_threads(PF)) gemm_h16_geglu_persist_kernel(GemmParams p, EpiGeglu<NP> epi, int vblocks) {
  static_assert(BM % 32 == 0 && BN % 64 == 0, "2 x 2 waves; whole wi_0 | wi_1 groups per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  warm_kernargs<kernarg_lines<GemmParams, EpiGeglu<NP>, int>()>();
  if constexpr (kPfWave && PF != kPfNone) {
    if (threadIdx.x >= 256) {
      prefetch_wave<PF>(p.pf, blockIdx.x, gridDim.x, p.B[0]);
      return;
    }
  }
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  constexpr int STAGE_BYTES = NP * (BM + BN) * 128;
  typedef GemmTileIssue<NP, BM, BN, NS> Issue;
  char* const aux = smem + NS * STAGE_BYTES;
  float* const rs = reinterpret_cast<float*>(aux + rowscale_aux_bytes<BM>());
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lm = lane & 15, ln = (lane >> 4) * 4;
  const int step = (int)gridDim.x;
  int v = blockIdx.x, bm = 0, bn = 0;
  bool have = false;
  for (; v < vblocks; v += step)
    if (gemm_block_tile(p, v, bm, bn)) { have = true; break; }
  if (!have) return;
  // epilogue operands of a tile: the bias row of its columns and -- a new row tile only -- the partial sums of squares of
  // its rows (EpiGeglu::prefetch = rowscale_prefetch, split so that the statistics move once per row tile)
  auto operands = [&](int tbm, int tbn, bool with_stats) {
    if (with_stats) aux_dma_linear(epi.rsc.ssq + (size_t)tbm * BM * epi.rsc.tiles, aux, BM * epi.rsc.tiles * 4, wave, lane);
    if (wave == 3)
      aux_dma_row(epi.rsc.bias + (size_t)scan_index(epi.rsc.step_ptr) * epi.rsc.bias_step_stride + tbn * BN,
                  aux + rowscale_ssq_bytes<BM>(epi.rsc.tiles), BN * 4, lane);
  };
  {
    const Issue first(p, bm, bn, smem);
    first.template issue<0, Issue::COUNT>();
  }
  operands(bm, bn, true);
  bool stats_stale = true;
  for (;;) {
    GemmAcc<BM, BN> acc;
    const int vcur = v;   // (debug builds: this tile's phase-stamp record)
    gemm_tile<NP, BM, BN, NS, EpiGeglu<NP>, PF, /*PROLOGUE=*/false, /*REG_EPI=*/true>(p, epi, bm, bn, smem, &acc, vcur, /*fresh=*/stats_stale && vcur == (int)blockIdx.x);
    const int m0 = bm * BM, n0 = bn * BN;
    int nbm = 0, nbn = 0;
    bool more = false;
    for (v += step; v < vblocks; v += step)
      if (gemm_block_tile(p, v, nbm, nbn)) { more = true; break; }
    if (more) {   // the ring is free: the accumulators stay in registers
      const Issue next(p, nbm, nbn, smem);
      next.template issue<0, Issue::COUNT>();
    }
    if (stats_stale) tile_rstd_compute<BM>(epi.rsc, rs, m0, tid, aux);
    typedef const __attribute__((address_space(3))) f32x4* lds_cf32x4_t;
    lds_cf32x4_t brow = (lds_cf32x4_t)(size_t)(unsigned)(size_t)(aux + rowscale_ssq_bytes<BM>(epi.rsc.tiles));
    f32x4 bias[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) bias[j] = brow[(wn * WN + j * 16 + ln) >> 2];
    __syncthreads();   // rstd of the tile's rows visible; every lane holds its bias entries: the aux rows are free
    if (more) operands(nbm, nbn, nbm != bm);
    MSD_TS_AT(0, vcur, 4)
    RangeCheck rc;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int r = wm * WM + i * 16 + lm;
      const float rstd = rs[r];
      const size_t orow = (size_t)(m0 + r) * epi.ldc + (n0 + wn * WN) / 2 + ln;
#pragma unroll
      for (int q = 0; q < FN / 2; ++q) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float h0 = __builtin_fmaf(acc.a[i][2 * q][e] * kWScaleInv, rstd, bias[2 * q][e]);
          const float h1 = __builtin_fmaf(acc.a[i][2 * q + 1][e] * kWScaleInv, rstd, bias[2 * q + 1][e]);
          o[e] = gelu_tanh(h0) * h1;
        }
        store_h16x4<NP>(epi.out, orow + q * 16, o, rc);
      }
    }
    rc.commit(p.sat, p.sat_tag);
    MSD_TS_AT(0, vcur, 5)
    MSD_TS_AT(0, vcur, 6)
#if MSD_TIMESTAMPS
    if (threadIdx.x == 0 && vcur < kTsBlocks) {
      g_msd_ts[0][vcur][9] = (unsigned long long)vblocks;
      g_msd_ts[0][vcur][11] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (!more) break;
    stats_stale = nbm != bm;
    bm = nbm; bn = nbn;
  }
}

template <int NP, int BM, int BN, int NS>
inline hipError_t gemm_h16_geglu_persist_prepare() {
  constexpr int smem = gemm_h16_geglu_persist_smem<NP, BM, BN, NS>();
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_geglu_persist_kernel<NP, BM, BN, NS, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h16_geglu_persist_kernel<NP, BM, BN, NS, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  return e != hipSuccess ? e : r;
}

// `blocks`: resident blocks to launch (a multiple of 8, at most one per CU)
template <int NP, int BM, int BN, int NS>
inline hipError_t launch_gemm_h16_geglu_persist(const GemmParams& p_in, const EpiGeglu<NP>& epi, int blocks, hipStream_t stream) {
  static_assert(NP == 2, "two-plane modes");
  constexpr int smem = gemm_h16_geglu_persist_smem<NP, BM, BN, NS>();
  static const hipError_t attr = gemm_h16_geglu_persist_prepare<NP, BM, BN, NS>();
  if (attr != hipSuccess) return attr;
  if (!epi.rsc.ssq || !epi.rsc.bias || p_in.K < NS * kGemmBK) return hipErrorInvalidValue;
  GemmParams p = p_in;
  p.map.fill(p.M, p.N, BM, BN, p.xcd_rows);
  const int rx = p.xcd_rows, cx = 8 / rx;
  const int vblocks = 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / BM + rx - 1) / rx);
  const int grid = vblocks < blocks ? vblocks : blocks;
  if (prefetch_kind(p.pf) >= 1)
    hipLaunchKernelGGL((gemm_h16_geglu_persist_kernel<NP, BM, BN, NS, 1>), dim3(grid), dim3(256 + pf_threads(1)), smem, stream, p, epi, vblocks);
  else
    hipLaunchKernelGGL((gemm_h16_geglu_persist_kernel<NP, BM, BN, NS, 0>), dim3(grid), dim3(256), smem, stream, p, epi, vblocks);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// The folded cross-attention query projection (round 6; DESIGN.md 5 S6, msd_api.hip decoder_layers).
//   q = rstd(x1) ((x1 (.) gamma) . Wq),  x1 = x0 + ao . Wo   (network.py:196-198 on the residual of :174-193)
//     = rstd(x1) ((x0 (.) gamma) . Wq  +  ao . (Wo diag(gamma) Wq))
// Neither term needs the self-attention output projection: the first rides on the QKV launch's idle CUs (its A operand
// is written by whichever epilogue produced x0: EpiResidualNorm Y2 / EpiInProj y2), the second runs BESIDE the output
// projection -- same A operand, one launch -- and adds the first in its epilogue (EpiAddStoreH16); the 1/rms moves
// onto the logits inside the attention kernel (attention.h AttnParams::q_ssq).  One launch less per decoder layer.
// Round 3 measured this algebra at 0 ... +1 % (docs/history.md "Hoisting"): its dual launch put both problems on
// 32 x 32 tiles -- 576 blocks for 512 resident slots, 13.1 us where the two launches took 7.2 + 6.2.  Here each
// problem keeps its own tile shape and the launch is at most one block per CU (192 + 64 at base).
// ----------------------------------------------------------------------------

// C (row-major 16-bit planes) = acc + addend[m][n] (fp32).  The addend tile is prefetched into the aux LDS region like
// the residual tile of EpiResidualNorm, so the epilogue issues no global load.
template <int NP>
struct EpiAddStoreH16 {
  h16_t* out[2];
  int ldc;
  const float* addend;
  int ld_add;
  template <int BM, int BN> static constexpr int aux_bytes() { return BM * BN * 4; }
  template <int BM, int BN>
  __device__ void prefetch(char* aux, int m0, int n0, int wave, int lane) const {
    constexpr int CPR = BN / 4;   // 16-byte chunks per tile row
    static_assert((BM * CPR) % 64 == 0, "whole DMA instructions");
    for (int i = wave; i < BM * CPR / 64; i += 4) {
      const int id = i * 64 + lane, r = id / CPR, ch = id % CPR;
      __builtin_amdgcn_global_load_lds((aux_gptr_t)(addend + (size_t)(m0 + r) * ld_add + n0 + ch * 4),
                                       lds_ptr_of(aux + i * 1024), 16, 0, 0);
    }
  }
  template <int BM, int LD>
  __device__ void stats(float*, int, int, const char*) const {}
  template <int BM, int BN, int LD>
  __device__ void run(float* s0, int m0, int n0, int tid, const char* aux = nullptr, bool stats_done = false,
                      SatFlag sf = SatFlag()) const {
    typedef const __attribute__((address_space(3))) f32x4* lds_cf32x4;
    lds_cf32x4 xs = (lds_cf32x4)(aux);
    RangeCheck rc;
    MSD_EPI_ITEMS(BM * BN / 8, item) {
      const int m = item / (BN / 8), n = (item % (BN / 8)) * 8;
      float v[8];
      tile_row8<LD>(s0, m, n, v);
      const f32x4 a = xs[(m * BN + n) / 4], b = xs[(m * BN + n) / 4 + 1];
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
      v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
      store_h16x8<NP>(out, (size_t)(m0 + m) * ldc + n0 + n, v, rc);
    }
    rc.commit(sf.p, sf.tag);
  }
};

// Two independent GEMMs in ONE launch (no data flows between them), each on its own tile shape and epilogue: blocks
// [0, n1) run problem 1, the rest problem 2.  n1 is a multiple of 8, so a block's XCD (blockIdx % 8) is the same in the
// launch-wide and in the problem-local numbering and both problems keep their XCD-aware tile maps.  The launch's weight
// prefetch target (at most one, p1.pf) is touched by the prefetch waves of all blocks.
template <int NP, int BM1, int BN1, int NS1, class Epi1, int BM2, int BN2, int NS2, class Epi2, int PF = kPfNone>
__global__ void __launch_bounds__(256 + pf_threads(PF)) gemm_h16_dual_kernel(GemmParams p1, Epi1 e1, GemmParams p2, Epi2 e2, int n1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  warm_kernargs<kernarg_lines<GemmParams, Epi1, GemmParams, Epi2, int>()>();
  if constexpr (kPfWave && PF != kPfNone) {
    if (threadIdx.x >= 256) {
      prefetch_wave<PF>(p1.pf, blockIdx.x, gridDim.x, p1.B[0]);
      return;
    }
  }
  int bm, bn;
  if ((int)blockIdx.x < n1) {
    if (!gemm_block_tile(p1, (int)blockIdx.x, bm, bn)) return;
    gemm_tile<NP, BM1, BN1, NS1, Epi1, PF>(p1, e1, bm, bn, smem);
  } else {
    if (!gemm_block_tile(p2, (int)blockIdx.x - n1, bm, bn)) return;
    gemm_tile<NP, BM2, BN2, NS2, Epi2, PF>(p2, e2, bm, bn, smem);
  }
}

template <int NP, int BM1, int BN1, int NS1, class Epi1, int BM2, int BN2, int NS2, class Epi2>
constexpr int gemm_h16_dual_smem() {
  constexpr int a = gemm_h16_dma_smem<NP, BM1, BN1, NS1, Epi1>(), b = gemm_h16_dma_smem<NP, BM2, BN2, NS2, Epi2>();
  return a > b ? a : b;
}

// one-time opt-in to > 64 KiB dynamic LDS; call OUTSIDE stream capture
template <int NP, int BM1, int BN1, int NS1, class Epi1, int BM2, int BN2, int NS2, class Epi2>
inline hipError_t gemm_h16_dual_prepare() {
  constexpr int smem = gemm_h16_dual_smem<NP, BM1, BN1, NS1, Epi1, BM2, BN2, NS2, Epi2>();
  if (smem < 64 * 1024) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(
      reinterpret_cast<const void*>(gemm_h16_dual_kernel<NP, BM1, BN1, NS1, Epi1, BM2, BN2, NS2, Epi2, 0>),
      hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const hipError_t r = hipFuncSetAttribute(
      reinterpret_cast<const void*>(gemm_h16_dual_kernel<NP, BM1, BN1, NS1, Epi1, BM2, BN2, NS2, Epi2, 1>),
      hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  return e != hipSuccess ? e : r;
}

inline int gemm_grid_blocks(const GemmParams& p, int BM, int BN) {
  const int rx = p.xcd_rows, cx = 8 / rx;
  return 8 * ((p.N / BN + cx - 1) / cx) * ((p.M / BM + rx - 1) / rx);
}

template <int NP, int BM1, int BN1, int NS1, class Epi1, int BM2, int BN2, int NS2, class Epi2>
inline hipError_t launch_gemm_h16_dual(const GemmParams& p1_in, const Epi1& e1, const GemmParams& p2_in, const Epi2& e2,
                                       hipStream_t stream) {
  static_assert(NP == 2, "the folded query projection exists in the two-plane modes only");
  constexpr int smem = gemm_h16_dual_smem<NP, BM1, BN1, NS1, Epi1, BM2, BN2, NS2, Epi2>();
  static const hipError_t attr = gemm_h16_dual_prepare<NP, BM1, BN1, NS1, Epi1, BM2, BN2, NS2, Epi2>();
  if (attr != hipSuccess) return attr;
  GemmParams p1 = p1_in, p2 = p2_in;
  p1.map.fill(p1.M, p1.N, BM1, BN1, p1.xcd_rows);
  p2.map.fill(p2.M, p2.N, BM2, BN2, p2.xcd_rows);
  const int n1 = gemm_grid_blocks(p1, BM1, BN1), n2 = gemm_grid_blocks(p2, BM2, BN2);
  if (prefetch_kind(p1.pf) >= 1)
    hipLaunchKernelGGL((gemm_h16_dual_kernel<NP, BM1, BN1, NS1, Epi1, BM2, BN2, NS2, Epi2, 1>), dim3(n1 + n2),
                       dim3(256 + pf_threads(1)), smem, stream, p1, e1, p2, e2, n1);
  else
    hipLaunchKernelGGL((gemm_h16_dual_kernel<NP, BM1, BN1, NS1, Epi1, BM2, BN2, NS2, Epi2, 0>), dim3(n1 + n2), dim3(256),
                       smem, stream, p1, e1, p2, e2, n1);
  return hipGetLastError();
}

}  // namespace msd
