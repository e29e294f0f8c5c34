#!/usr/bin/env python3
"""Songs per GPU through SEVERAL handles at once (one Python thread + one HIP stream per handle) against one handle with
all the songs: independent step graphs on independent streams let the hardware run one chain's launch heads and tails
(and partial rounds of blocks) under another chain's kernels -- no device-side hand-off is needed between chains that
never exchange data.  The library is thread-compatible per handle (include/msd_amd.h); ctypes releases the GIL.

  python tools/ab/multi_handle.py [--steps 200] [--rounds 3] 1x8 2x4 4x2 8x1 2x8      (handles x songs per handle)

Prints mel-frames/s of every layout (median over rounds; round 0 = restore + graph capture, not timed)."""
from __future__ import annotations

import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--preset', default='base_with_context')
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--rounds', type=int, default=3)
  ap.add_argument('layouts', nargs='+')
  args = ap.parse_args()
  import torch
  import msd_amd
  spec = msd_amd.config.preset(args.preset, num_steps=args.steps)
  t_frames = spec.task_feature_lengths['targets']
  c_len = spec.task_feature_lengths.get('targets_context') if spec.has_context else None
  for layout in args.layouts:
    nh, nb = (int(v) for v in layout.split('x'))
    models = [msd_amd.InferenceModel('synthetic:0', spec, batch_size=nb) for _ in range(nh)]
    rates = []
    for r in range(args.rounds + 1):
      batches = []
      for h in range(nh):
        b = {'encoder_input_tokens': np.concatenate([msd_amd.synthetic.segment_tokens(spec, 1000 * (h * nb + s) + r) for s in range(nb)], 0)}
        if c_len:
          b['encoder_continuous_inputs'] = torch.zeros((nb, c_len, 128), dtype=torch.float32, device=models[h].device)
          b['encoder_continuous_mask'] = np.ones((nb, c_len), np.int32)
        batches.append(b)
      outs = [None] * nh

      def work(h):
        outs[h], _ = models[h].predict(batches[h], seed=r, segment=r, return_torch=True)

      torch.cuda.synchronize()
      t0 = time.perf_counter()
      if r == 0:   # restore + graph capture one handle after the other: weight loading copies through the legacy stream,
        for h in range(nh):   # which must not meet another thread's stream capture
          work(h)
      else:
        threads = [threading.Thread(target=work, args=(h,)) for h in range(nh)]
        for t in threads:
          t.start()
        for t in threads:
          t.join()
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      assert all(o is not None and bool(torch.isfinite(o).all()) for o in outs)
      if r > 0:
        rates.append(nh * nb * t_frames * (1000.0 / args.steps) / dt / (1000.0 / args.steps))   # frames per second of THIS run
    # normalise to 1000-step segments: a run of `steps` steps synthesizes steps / 1000 of a segment's work
    per_1000 = [v * args.steps / 1000.0 for v in rates]
    print('%-6s %2d handle(s) x %2d song(s): %8.1f mel-frames/s at 1000 steps (median of %d; %s)'
          % (layout, nh, nb, float(np.median(per_1000)), len(per_1000), ' '.join('%.1f' % v for v in per_1000)), flush=True)
    del models
    torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
