"""Where does a weight-streaming GEMM launch of the DDPM step spend its time?  (GPU box, debug build only.)

Runs a few denoising steps of base_with_context on a library built from csrc/ + tools/diag/phase_timestamps.patch
with -DMSD_TIMESTAMPS=1 (tools/README.md: "phase timestamps"), then reads the per-block stamps of the LAST launch of
each tile shape:

  entry      block starts (first instruction behind the accumulator clear)
  landed     K-tile 0 is in LDS (prologue DMA issued + first counted wait + barrier)
  loop end   main loop finished (all MFMAs issued, fragment reads done)
  epi end    epilogue finished and its stores have left (s_waitcnt vmcnt(0))

s_memtime (core clock) gives the deltas inside one block, s_memrealtime (100 MHz, one counter for the chip) aligns
blocks with each other and calibrates the core clock.  The product library has none of this (the patch is not
applied to the tree; the default build's hash is unchanged).

usage (GPU box):  [BATCH=8] MSD_AMD_LIB=tools/ab/libs/libmsd_amd_ts.so python tools/diag/phase_times.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import msd_amd
from msd_amd import native
from tests import helpers

CLASSES = ['BN = 128 (gated MLP-in, last layer: 64 x 128 at one song, 128 x 128 batched)',
           'BN = 96 (one song: QKV, last layer, 64 x 96; batched: the last 128 x 96 launch of the step)',
           '64 x 64 (last such launch of the step)', 'narrow tiles (the last such launch of the step)']


def pct(v):
  return '%7.2f %7.2f %7.2f' % tuple(np.percentile(v, [10, 50, 90]))


def main():
  steps, nb = int(os.environ.get('STEPS', '16')), int(os.environ.get('BATCH', '1'))
  spec = msd_amd.config.preset('base_with_context', num_steps=steps)
  model = msd_amd.InferenceModel('synthetic:0', spec, batch_size=nb)
  batch = helpers.make_batch(spec, batch=nb)
  init_z, noise = helpers.make_noise(spec, batch=nb)
  print('base_with_context, %d song(s) per handle, %d steps' % (nb, steps))
  for _ in range(2):   # the second run is the graph replay, weights staged, clocks up
    model.predict(batch, init_z=init_z, noise=noise)
  torch.cuda.synchronize()
  lib = native.load('f16')
  if not hasattr(lib, 'msd_debug_timestamps'):
    raise SystemExit('this library has no msd_debug_timestamps: build it with the patch and -DMSD_TIMESTAMPS=1')
  ts = np.zeros((4, 1024, 8), np.uint64)
  rc = lib.msd_debug_timestamps(ctypes.c_void_p(ts.ctypes.data))
  assert rc == 0, rc
  for c, name in enumerate(CLASSES):
    grid = int(ts[c, 0, 5])
    if grid == 0:
      continue
    n = min(grid, 1024)
    t = ts[c, :n].astype(np.int64)
    core = t[:, :4] - t[:, :1]                           # core-clock ticks since this block's entry
    real = (t[:, 7] - t[:, 6]).astype(np.float64) * 10.  # ns, 10 ns resolution
    ghz = core[:, 3].sum() / real.sum()                  # ticks per ns
    ph = np.diff(core, axis=1) / ghz / 1e3               # us: landed-entry, loop-landed, epi-loop
    entry = (t[:, 6] - t[:, 6].min()) * 0.01             # us since the first block of the launch started
    end = (t[:, 7] - t[:, 6].min()) * 0.01
    print('\n%s: grid %d blocks, core clock %.3f GHz (s_memtime / s_memrealtime)' % (name, grid, ghz))
    print('  launch span first entry -> last epilogue end : %7.2f us' % end.max())
    print('  per block (us)                     p10     p50     p90')
    print('  entry after first block        %s' % pct(entry))
    print('  entry -> K-tile 0 landed       %s' % pct(ph[:, 0]))
    print('  landed -> main loop end        %s' % pct(ph[:, 1]))
    print('  loop end -> epilogue + stores  %s' % pct(ph[:, 2]))
    print('  block lifetime                 %s' % pct(ph.sum(1)))
    xcc = t[:, 4]
    for x in sorted(set(xcc.tolist())):
      m = xcc == x
      print('  XCD %d: %3d blocks, entries %6.2f..%6.2f us, last end %6.2f us, median lifetime %6.2f us'
            % (x, m.sum(), entry[m].min(), entry[m].max(), end[m].max(), np.median(ph[m].sum(1))))


if __name__ == '__main__':
  main()
