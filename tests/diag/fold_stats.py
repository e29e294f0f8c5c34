#!/usr/bin/env python3
"""Short-chain error statistics of the folded cross-attention query projection (msd_config.cross_q_fold) against the
unfolded order, on the device: fraction of elements beyond 1e-3 / 1e-4 of the float64 oracle, relative to the float32
oracle's own (tests/helpers.py assert_fp32_class), over seeds / styles / step counts.  Test infrastructure (imports the
oracle); run on the GPU box:  python tests/diag/fold_stats.py [--seeds 6]"""
import argparse
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--seeds', type=int, default=6)
  ap.add_argument('--preset', default='tiny_context')
  args = ap.parse_args()
  import msd_amd
  from oracle import backend, fast
  from tests import helpers
  rows = []
  for style in ('concat', 'sum'):
    for steps in (4, 8):
      for seed in range(args.seeds):
        spec = msd_amd.config.preset(args.preset, num_steps=steps)
        if style == 'sum':
          spec = dataclasses.replace(spec, t5=dataclasses.replace(spec.t5, decoder_cross_attend_style='sum_cross_attends'))
        params = msd_amd.synthetic.init_params(spec, 100 + seed, norm_scale_jitter=0.1)
        batch = helpers.make_batch(spec, batch=2, ctx_mask='ragged', seed=seed)
        init_z, noise = helpers.make_noise(spec, batch=2, seed=50 + seed)
        cfg, dc = helpers.oracle_configs(spec)
        ref = {}
        for dt in ('float64', 'float32'):
          xp = backend.TorchBackend(dt)
          fm = fast.FastModel(xp, cfg, dc, params, spec.has_context)
          ref[dt] = xp.to_numpy(fm.predict(batch, init_z, noise)[0]).astype(np.float64)
        e32 = np.abs(ref['float32'] - ref['float64']).ravel()
        out = {}
        for fold in (False, True):
          model = msd_amd.InferenceModel(params, spec, batch_size=2, cross_q_fold=fold, **helpers.ALL_PLANES)
          got, _ = model.predict(batch, init_z=init_z, noise=noise)
          e = np.abs(np.asarray(got, np.float64) - ref['float64']).ravel()
          out[fold] = [float((e > t).mean()) / max(float((e32 > t).mean()), 1e-9) for t in (1e-3, 1e-4)] + [float(np.median(e) / max(np.median(e32), 1e-12))]
          del model
        rows.append((style, steps, seed, out))
        print('%-6s steps %d seed %d  f32 beyond 1e-3 %.4f | ratio to f32 oracle (1e-3, 1e-4, median): unfolded %.3f %.3f %.2f   folded %.3f %.3f %.2f'
              % (style, steps, seed, float((e32 > 1e-3).mean()), *out[False], *out[True]), flush=True)
  for style in ('concat', 'sum'):
    for k, name in ((False, 'unfolded'), (True, 'folded')):
      r = np.array([o[k][:2] for s, _, _, o in rows if s == style])
      print('%-6s %-8s mean ratio beyond 1e-3 %.3f (max %.3f)  beyond 1e-4 %.3f (max %.3f)' % (style, name, r[:, 0].mean(), r[:, 0].max(), r[:, 1].mean(), r[:, 1].max()))


if __name__ == '__main__':
  main()
