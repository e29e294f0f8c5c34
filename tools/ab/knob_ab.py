#!/usr/bin/env python3
"""Same-box, same-PROCESS A/B of InferenceModel knobs (msd_config fields): one model per variant, all alive at once,
segments strictly alternating between them (variant 0, 1, ..., 0, 1, ...), so box-to-box spread and slow drift hit
every variant alike.  A process start costs ~20 s of GPU box time; this costs one.

  python tools/ab/knob_ab.py [--preset base_with_context] [--steps 1000] [--rounds 5] [--batch 1]
        [--tokens 1100 [--tokens 300 ...]]      valid token counts of the synthetic segment (default: the bench's seeds)
        'dedup_layer0=None' 'dedup_layer0=False' ...     one variant per argument: comma-separated kwargs of InferenceModel

Prints, per token count: median / min sample time per segment of every variant, the ratio to variant 0, and whether the
outputs are bit-identical to variant 0's.  Nothing here imports the oracle."""
from __future__ import annotations

import argparse
import ast
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def parse_variant(text):
  kw = {}
  for part in [p for p in text.split(',') if p.strip()]:
    k, v = part.split('=', 1)
    kw[k.strip()] = ast.literal_eval(v.strip())
  return kw


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--preset', default='base_with_context')
  ap.add_argument('--steps', type=int, default=1000)
  ap.add_argument('--rounds', type=int, default=5)
  ap.add_argument('--batch', type=int, default=1)
  ap.add_argument('--tokens', type=int, action='append', default=None)
  ap.add_argument('--json', default='')
  ap.add_argument('variants', nargs='+')
  args = ap.parse_args()
  import torch
  import msd_amd
  spec = msd_amd.config.preset(args.preset, num_steps=args.steps)
  variants = [parse_variant(v) for v in args.variants]
  models = [msd_amd.InferenceModel('synthetic:0', spec, batch_size=args.batch, **kw) for kw in variants]
  c_len = models[0].targets_context_length
  nb = args.batch
  rows = []
  for tok in (args.tokens or [None]):
    def tokens(seed):
      t = np.concatenate([msd_amd.synthetic.segment_tokens(spec, 1000 * b + seed) for b in range(nb)], 0)
      if tok is not None:
        t = np.concatenate([msd_amd.synthetic.segment_tokens(spec, 1000 * b + seed, min_len=tok, max_len=tok) for b in range(nb)], 0)
      return t
    ctx = torch.zeros((nb, c_len, 128), dtype=torch.float32, device=models[0].device) if c_len else None
    times = [[] for _ in models]
    same = [True] * len(models)
    for r in range(args.rounds + 1):      # round 0: restore, tables, graph capture -- not timed
      batch = {'encoder_input_tokens': tokens(7 + r)}
      if c_len:
        batch['encoder_continuous_inputs'] = ctx
        batch['encoder_continuous_mask'] = np.ones((nb, c_len), np.int32)
      outs = []
      for i, m in enumerate(models):
        out, _ = m.predict(batch, seed=3, segment=r, return_torch=True)
        torch.cuda.synchronize()
        if r > 0:
          times[i].append(m.last_timing['sample_s'] * 1e3)
        outs.append(out)
      for i in range(1, len(models)):
        same[i] = same[i] and bool(torch.equal(outs[0], outs[i]))
      if c_len:
        ctx = outs[0]
    n_keys = int((batch['encoder_input_tokens'][0] > 0).sum()) + (c_len or 0)
    base = float(np.median(times[0]))
    print('--- %s, %d steps, %d song(s), %d valid keys (%s tokens)' % (args.preset, args.steps, nb, n_keys, tok if tok is not None else 'seeded'))
    for i, kw in enumerate(variants):
      med = float(np.median(times[i]))
      print('  %-44s median %8.2f ms  min %8.2f  x%.4f  %s' % (args.variants[i] or '(default)', med, min(times[i]), med / base,
                                                                 '' if i == 0 else ('bit-identical' if same[i] else 'DIFFERENT BITS')))
      rows.append({'tokens': tok, 'keys': n_keys, 'variant': args.variants[i], 'median_ms': med, 'min_ms': min(times[i]),
                   'ratio': med / base, 'identical_to_first': same[i], 'all_ms': times[i]})
    sys.stdout.flush()
  if args.json:
    with open(args.json, 'w') as f:
      json.dump(rows, f, indent=1)


if __name__ == '__main__':
  main()
