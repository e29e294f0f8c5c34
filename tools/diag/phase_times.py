"""Where does a weight-streaming GEMM launch of the DDPM step spend its time?  (GPU box, debug build only.)

Runs a few denoising steps of base_with_context on a library built with -DMSD_TIMESTAMPS=1 (the stamps live in the
sources behind that macro since round 4; tools/README.md: "phase timestamps"), then reads the per-block stamps of the
LAST launch of each tile shape:

  entry      block starts (first instruction behind the accumulator clear)
  issued     the prologue's LDS-DMA instructions (NS K-tiles + the epilogue's aux rows) are issued
  landed     K-tile 0 is in LDS (first counted wait + barrier)
  loop end   main loop finished (all MFMAs issued, fragment reads done)
  slab       accumulators in the LDS slab, row statistics done, barrier
  epi issued epilogue arithmetic done, its global stores issued
  left       the stores have left (s_waitcnt vmcnt(0))

and the same for the attention kernels (Q loads + ring DMAs issued, stage 0 landed, key loop end, partials in LDS,
merged + stored) and the key-split merge kernel (entry, end).

s_memtime (core clock) gives the deltas inside one block, s_memrealtime (100 MHz, one counter for the chip) aligns
blocks with each other and calibrates the core clock.  The product library has none of this (every stamp macro is
empty without -DMSD_TIMESTAMPS=1: tests/test_diag_tools.py).

With MSD_CHAIN=1|2 (experiments build) the gated-MLP-in / MLP-out / QKV tiles run inside ONE launch (chain.h); the
script then also prints the launch's timeline: per phase the span first entry -> last end, and per block the time
between the end of its phase-0 tile and the entry of its phase-1 tile (= XCD barrier + waiting for the slowest block).

usage (GPU box):  [BATCH=8] [MSD_CHAIN=2] MSD_AMD_LIB=tools/ubench/exp/libmsd_amd_exp_ts.so python tools/diag/phase_times.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import msd_amd
from msd_amd import native
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as helpers   # (not tests.helpers: that imports oracle/)

CLASSES = ['GEMM, BN = 128 (gated MLP-in, last layer: 64 x 128 at one song, 128 x 128 batched)',
           'GEMM, BN = 96 (one song: QKV, last layer, 64 x 96; batched: the last 128 x 96 launch of the step)',
           'GEMM, other 64-row tiles (one song: the MLP output projection, 64 x 32 over K = 2048)',
           'GEMM, narrow tiles (the last such launch of the step)',
           'attention, 32 query rows per block (one song: decoder self-attention, last layer)',
           'attention, 64 query rows per block (cross-attention, last layer)',
           'key-split merge (last layer)',
           'GEMM, 256 x 128 tile with loader waves (gemm_h16_ls.h, MSD_BIG_LS=1; consumer wave 0)']
GEMM_PHASES = ['entry -> prologue DMAs issued', 'issued -> K-tile 0 landed', 'landed -> main loop end',
               'loop end -> slab + row statistics', 'slab -> epilogue stores issued', 'issued -> stores have left']
LS_PHASES = ['entry -> tile 0 visible (first barrier)', 'first barrier -> main loop end', 'loop end -> aux rows landed',
             'aux -> first 64-row pass done', 'first pass -> fourth pass done', 'issued -> stores have left']
ATT_PHASES = ['entry -> Q loads + ring DMAs issued', 'issued -> stage 0 landed', 'landed -> key loop end',
              'loop end -> partials in LDS', 'partials -> merged + stores issued', 'issued -> stores have left']


def pct(v):
  return '%7.2f %7.2f %7.2f' % tuple(np.percentile(v, [10, 50, 90]))


def main():
  steps, nb = int(os.environ.get('STEPS', '16')), int(os.environ.get('BATCH', '1'))
  spec = msd_amd.config.preset('base_with_context', num_steps=steps)
  import ast
  kw = {k.strip(): ast.literal_eval(v.strip()) for k, v in (it.split('=', 1) for it in os.environ.get('KNOB', '').split(',') if it.strip())}
  model = msd_amd.InferenceModel('synthetic:0', spec, batch_size=nb, **kw)   # KNOB='mlp_in_persistent=False,...': InferenceModel keywords
  batch = helpers.make_batch(spec, batch=nb)
  init_z, noise = helpers.make_noise(spec, batch=nb)
  print('base_with_context, %d song(s) per handle, %d steps' % (nb, steps))
  for _ in range(2):   # the second run is the graph replay, weights staged, clocks up
    model.predict(batch, init_z=init_z, noise=noise)
  torch.cuda.synchronize()
  lib = native.load('f16')
  if not hasattr(lib, 'msd_debug_timestamps'):
    raise SystemExit('this library has no msd_debug_timestamps: build it with the patch and -DMSD_TIMESTAMPS=1')
  ts = np.zeros((8, 1024, 12), np.uint64)
  rc = lib.msd_debug_timestamps(ctypes.c_void_p(ts.ctypes.data))
  assert rc == 0, rc
  for c, name in enumerate(CLASSES):
    grid = int(ts[c, 0, 9])
    if grid == 0 or name is None:
      continue
    n = min(grid, 1024)
    t = ts[c, :n].astype(np.int64)
    t = t[t[:, 11] > 0]                                  # blocks that left early (no work) never stamp the end
    if c == 6:                                           # merge: entry and end only
      t[:, 1:6] = t[:, :1]
    if c == 7:                                           # loader-wave kernel: fields 0 1 2 3 4 5 6, [7] = loader's "prologue issued"
      print('\n  (loader wave 4: entry -> 36 prologue DMAs issued, us p10/p50/p90: %s)' % pct((t[:, 7] - t[:, 0]) / 2.0e3))
    core = t[:, :7] - t[:, :1]                           # core-clock ticks since this block's entry
    real = (t[:, 11] - t[:, 10]).astype(np.float64) * 10.  # ns, 10 ns resolution
    ghz = core[:, 6].sum() / real.sum()                  # ticks per ns
    ph = np.diff(core, axis=1) / ghz / 1e3               # us per phase
    entry = (t[:, 10] - t[:, 10].min()) * 0.01           # us since the first block of the launch started
    end = (t[:, 11] - t[:, 10].min()) * 0.01
    print('\n%s: grid %d blocks (%d stamped), core clock %.3f GHz (s_memtime / s_memrealtime)' % (name, grid, len(t), ghz))
    print('  launch span first entry -> last block end : %7.2f us' % end.max())
    print('  per block (us)                            p10     p50     p90')
    print('  entry after first block               %s' % pct(entry))
    for k, label in enumerate(GEMM_PHASES if c < 4 else (LS_PHASES if c == 7 else ATT_PHASES)):
      if c == 6 and k != 5:
        continue
      print('  %-37s %s' % (label if c != 6 else 'entry -> stores have left', pct(ph[:, k])))
    print('  block lifetime                        %s' % pct(ph.sum(1)))
    xcc = t[:, 8]
    print('  per XCD (blocks, last end us): ' + '  '.join('%d: %d, %.2f' % (x, (xcc == x).sum(), end[xcc == x].max())
                                                          for x in sorted(set(xcc.tolist()))))
  if os.environ.get('MSD_CHAIN', '0') not in ('', '0'):
    chain_timeline(ts)


def chain_timeline(ts):
  """classes 0 (gated MLP-in) and 2 (MLP-out) were stamped by the SAME launch (the last layer's chain; its QKV phase
  does not exist).  s_memrealtime (10 ns) is one counter for the chip: block b's phase-1 entry minus its phase-0 end is
  what the phase boundary cost that block."""
  a, b = ts[0].astype(np.int64), ts[2].astype(np.int64)
  n = int(min(a[0, 9], 1024))
  a, b = a[:n], b[:n]
  t0 = a[a[:, 11] > 0, 10].min()
  print('\nchain launch timeline (last layer; us since the first block entered phase 0)')
  for name, t in (('phase 0 gated MLP-in', a), ('phase 1 MLP-out', b)):
    live = (t[:, 11] > 0) & (t[:, 9] == a[0, 9]) & (t[:, 10] >= t0)
    e0, e1 = (t[live, 10] - t0) * 0.01, (t[live, 11] - t0) * 0.01
    print('  %-22s blocks %3d   first entry %6.2f   median entry %6.2f   median end %6.2f   last end %6.2f'
          % (name, live.sum(), e0.min(), np.median(e0), np.median(e1), e1.max()))
  both = (a[:, 11] > 0) & (b[:, 11] > 0) & (a[:, 9] == b[:, 9]) & (b[:, 10] >= a[:, 11])   # same launch (grid), not a stale stamp
  gap = (b[both, 10] - a[both, 11]) * 0.01
  print('  phase boundary per block (phase-0 end -> phase-1 entry): p10 %.2f  p50 %.2f  p90 %.2f us' % tuple(np.percentile(gap, [10, 50, 90])))


if __name__ == '__main__':
  main()
