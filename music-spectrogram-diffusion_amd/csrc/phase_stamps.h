// Per-block phase stamps of the kernels (a DEBUG instrument: -DMSD_TIMESTAMPS=1, read by tools/diag/phase_times.py).
// In the product build every MSD_TS_* site expands to nothing and the library has no msd_debug_timestamps symbol.
#pragma once
#include "common.h"

namespace msd {

// Phase timestamps (debug builds only: -DMSD_TIMESTAMPS=1, tools/diag/phase_times.py): every block of a GEMM
// launch records s_memtime at entry, when K-tile 0 has landed, behind the main loop and behind the epilogue; the
// last launch of each tile shape stays in g_msd_ts and is read back through msd_debug_timestamps().  Compiled out
// of the product (the default build is bit-identical with and without this block).
#ifndef MSD_TIMESTAMPS
#define MSD_TIMESTAMPS 0
#endif
#if MSD_TIMESTAMPS
// classes: 0 BN = 128 | 1 BN = 96 | 2 other 64-row tiles | 3 narrow tiles | 4 attention QB = 1 | 5 attention QB = 2 | 6 merge
// fields : 0 entry | 1 prologue issued | 2 first tile landed | 3 loop end | 4 slab + stats | 5 epilogue issued |
//          6 stores left | 8 XCC_ID | 9 grid | 10 / 11 s_memrealtime at entry / end
constexpr int kTsClasses = 8, kTsBlocks = 1024, kTsFields = 12;
__device__ unsigned long long g_msd_ts[kTsClasses][kTsBlocks][kTsFields];
template <int BM, int BN> constexpr int ts_class() { return BN == 128 ? 0 : (BN == 96 ? 1 : (BM == 64 ? 2 : 3)); }
#define MSD_TS_AT(CLS, BLK, FIELD)                                                                       \
  if (threadIdx.x == 0 && (BLK) < kTsBlocks) g_msd_ts[CLS][BLK][FIELD] = __builtin_amdgcn_s_memtime();
#define MSD_TS_BEGIN(CLS, BLK)                                                                           \
  MSD_TS_AT(CLS, BLK, 0)                                                                                 \
  if (threadIdx.x == 0 && (BLK) < kTsBlocks) g_msd_ts[CLS][BLK][10] = __builtin_amdgcn_s_memrealtime();
#define MSD_TS_END(CLS, BLK, GRID)                                                                       \
  {                                                                                                      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    MSD_TS_AT(CLS, BLK, 6)                                                                               \
    if (threadIdx.x == 0 && (BLK) < kTsBlocks) {                                                         \
      unsigned xcc_;                                                                                     \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                \
      g_msd_ts[CLS][BLK][8] = xcc_ & 0xf;                                                                \
      g_msd_ts[CLS][BLK][9] = (unsigned long long)(GRID);                                                \
      g_msd_ts[CLS][BLK][11] = __builtin_amdgcn_s_memrealtime();                                         \
    }                                                                                                    \
  }
#define MSD_TS_STAMP(BM_, BN_, FIELD) MSD_TS_AT((ts_class<BM_, BN_>()), msd_ts_blk, FIELD)   /* msd_ts_blk: the enclosing gemm_tile's record */
#else
#define MSD_TS_AT(CLS, BLK, FIELD)
#define MSD_TS_BEGIN(CLS, BLK)
#define MSD_TS_END(CLS, BLK, GRID)
#define MSD_TS_STAMP(BM_, BN_, FIELD)
#endif

}  // namespace msd

#if MSD_TIMESTAMPS   // debug builds only (tools/diag/phase_times.py); the product library has no such symbol
extern "C" int msd_debug_timestamps(unsigned long long* host_out) {   // [kTsClasses][kTsBlocks][kTsFields]
  if (hipDeviceSynchronize() != hipSuccess) return 5;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(msd::g_msd_ts), sizeof(msd::g_msd_ts)) == hipSuccess ? 0 : 5;
}
#endif
