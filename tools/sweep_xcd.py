"""Same-process sweep of the per-GEMM-class XCD grid (rows x columns of the 8 XCDs over the tile grid) and
walk order, one class at a time against the current best (coordinate descent), on the headline workload.
The switches are read while the step graph is captured, so each setting = msd_reset_graph + one warm-up
segment + `--reps` timed segments.  Prints one line per setting and the best found.
  python tools/sweep_xcd.py [--reps 2] [--classes gemm_qkv,gemm_mlp_in_geglu,...]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = ['gemm_mlp_in_geglu', 'gemm_mlp_out', 'gemm_qkv', 'gemm_attn_out', 'gemm_cross_out', 'gemm_cross_q']


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=2)
  ap.add_argument('--classes', default=','.join(CLASSES))
  ap.add_argument('--extra', default='', help='semicolon-separated extra env settings to time, e.g. "MSD_GRAPH_STEPS=4;MSD_X=1"')
  args = ap.parse_args()
  import torch
  import msd_amd
  spec = msd_amd.config.preset('base_with_context')
  model = msd_amd.InferenceModel('synthetic:0', spec)
  c = model.targets_context_length
  state = {'pred': torch.zeros((1, c, 128), dtype=torch.float32, device=model.device), 'k': 0}

  def segment():
    k = state['k']
    batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, k),
             'encoder_continuous_inputs': state['pred'],
             'encoder_continuous_mask': (np.zeros if k == 0 else np.ones)((1, c), np.int32)}
    out, _ = model.predict(batch, seed=0, segment=k, return_torch=True)
    state['pred'] = out
    state['k'] = k + 1
    return model.last_timing['sample_s']

  def measure(env):
    for kv in env:
      os.environ[kv[0]] = kv[1]
    model._get_native().reset_graph()
    segment()
    t = min(segment() for _ in range(args.reps))
    for kv in env:
      os.environ.pop(kv[0], None)
    return t * 1e3 / spec.diffusion.sampler.schedule.num_steps   # us per DDPM step... (ms/1000 steps = us)

  segment()
  base = measure([])
  print('baseline %.2f us/step' % (base * 1e3), flush=True)
  best_env, best = [], base
  results = {'baseline_us': base * 1e3, 'settings': []}
  for cls in [c for c in args.classes.split(',') if c]:
    key = 'MSD_XCD_' + cls.upper()
    cls_best = None
    for rows in (1, 2, 4, 8):
      for walk in (0, 1):
        if (rows, walk) == (2, 1):
          continue   # the default
        t = measure(best_env + [(key, '%d,%d' % (rows, walk))])
        results['settings'].append({'class': cls, 'rows': rows, 'walk': walk, 'us': t * 1e3})
        print('%-20s rows %d walk %d : %.2f us/step (best so far %.2f)' % (cls, rows, walk, t * 1e3, best * 1e3), flush=True)
        if t < best * 0.997:   # > 0.3 % better than the running best
          best, cls_best = t, (key, '%d,%d' % (rows, walk))
    if cls_best:
      best_env.append(cls_best)
  again = measure([])
  final = measure(best_env)
  print('baseline again %.2f us/step; best env %s -> %.2f us/step' % (again * 1e3, best_env, final * 1e3))
  for e in [x for x in args.extra.split(';') if x]:
    kv = [tuple(p.split('=', 1)) for p in e.split()]
    print('extra [%s]: %.2f us/step' % (e, measure(kv) * 1e3), flush=True)
  results.update({'baseline_again_us': again * 1e3, 'best_env': best_env, 'best_us': final * 1e3})
  print(json.dumps(results))


if __name__ == '__main__':
  main()
