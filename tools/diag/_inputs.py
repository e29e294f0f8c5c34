"""Seeded model features and noise for the diagnostics in this directory.  (tests/helpers.py has the same two
functions, but importing it pulls in `oracle/`, which nothing outside tests/, smoke() and bench.py's cpu_baseline leg
may do -- tools/README.md.)"""
import numpy as np

import msd_amd


def make_batch(spec, batch=1, seed=7):
  """Model features with seeded content, full context mask."""
  rng = np.random.default_rng(seed)
  toks = np.concatenate([msd_amd.synthetic.segment_tokens(spec, b, min_len=8,
                                                           max_len=spec.task_feature_lengths['inputs'] - 2)
                         for b in range(batch)], 0)
  out = {'encoder_input_tokens': toks}
  n = 128
  if spec.has_context:
    c = spec.task_feature_lengths['targets_context']
    out['encoder_continuous_inputs'] = rng.uniform(-13, 5, (batch, c, n)).astype(np.float32)
    out['encoder_continuous_mask'] = np.ones((batch, c), np.int32)
  out['decoder_target_tokens'] = np.zeros((batch, spec.task_feature_lengths['targets'], n), np.float32)
  return out


def make_noise(spec, batch=1, seed=11):
  rng = np.random.default_rng(seed)
  t, n = spec.task_feature_lengths['targets'], 128
  steps = spec.diffusion.sampler.schedule.num_steps
  return (rng.standard_normal((batch, t, n)).astype(np.float32),
          rng.standard_normal((steps, batch, t, n)).astype(np.float32))
