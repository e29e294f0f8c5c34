"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py with
the float64 oracle).  CPU: the oracle still reproduces its tiny fixture (guards the
oracle against silent drift).  GPU: the HIP path against the FULL-SIZE fixtures at
BASELINE.json's shapes with the absolute bar of north_star: mel frames within 1e-3 rms."""
import os

import numpy as np
import pytest

import msd_amd
from tests import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_oracle_reproduces_tiny_golden():
  from oracle import backend, fast
  g = np.load(os.path.join(GOLD, 'tiny_context_n6.npz'))
  spec = msd_amd.config.preset('tiny_context', num_steps=6)
  params = msd_amd.synthetic.init_params(spec, int(g['weight_seed']), norm_scale_jitter=float(g['jitter']))
  batch = helpers.make_batch(spec, batch=2, seed=int(g['batch_seed']), ctx_mask='ragged')
  init_z, noise = helpers.make_noise(spec, batch=2, seed=int(g['noise_seed']))
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend('float64')
  out = xp.to_numpy(fast.FastModel(xp, cfg, dc, params, True).predict(batch, init_z, noise)[0])
  assert helpers.rms(out, g['mel']) < 1e-5


@pytest.mark.gpu
def test_tiny_golden_on_device():
  g = np.load(os.path.join(GOLD, 'tiny_context_n6.npz'))
  spec = msd_amd.config.preset('tiny_context', num_steps=6)
  params = msd_amd.synthetic.init_params(spec, int(g['weight_seed']), norm_scale_jitter=float(g['jitter']))
  batch = helpers.make_batch(spec, batch=2, seed=int(g['batch_seed']), ctx_mask='ragged')
  init_z, noise = helpers.make_noise(spec, batch=2, seed=int(g['noise_seed']))
  model = msd_amd.InferenceModel(params, spec, batch_size=2)
  got, _ = model.predict(batch, init_z=init_z, noise=noise)
  # 6 huge steps: float32 itself is ~1e-2 from float64 here (tests/test_gpu_model.py)
  assert helpers.rms(got, g['mel']) < 5e-2


def _song(preset, n_segments, noise_seed):
  """Device run with the SAME inputs make_golden.py used (tokens, Philox noise, chaining)."""
  from oracle import philox
  import torch
  spec = msd_amd.config.preset(preset, num_steps=1000)
  model = msd_amd.InferenceModel('synthetic:0', spec)
  t, n = spec.task_feature_lengths['targets'], 128
  c = spec.task_feature_lengths.get('targets_context')
  pred = np.zeros((1, c or 0, n), np.float32)
  outs = []
  for k in range(n_segments):
    batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, k)}
    if spec.has_context:
      batch['encoder_continuous_inputs'] = pred
      batch['encoder_continuous_mask'] = (np.zeros if k == 0 else np.ones)((1, c), np.int32)
    init_z, noise = philox.segment_noise((1, t, n), 1000, seed=noise_seed, segment=k)
    pred, _ = model.predict(batch, init_z=init_z, noise=noise)
    outs.append(pred)
  return model, np.concatenate(outs, 1)


@pytest.mark.gpu
def test_small_1000_steps_within_1e3_rms():
  """BASELINE config 2 shape (small, no context, 1000-step DDPM, CFG 5)."""
  g = np.load(os.path.join(GOLD, 'small_n1000.npz'))
  _, got = _song('small', int(g['n_segments']), int(g['noise_seed']))
  err = helpers.rms(got, g['mel'])
  print('small 1000-step rms vs float64 oracle: %.3e' % err)
  assert err <= 1e-3


@pytest.mark.gpu
def test_base_with_context_1000_steps_within_1e3_rms():
  """BASELINE config 3 shape: base_with_context, 2 chained segments, 1000 steps, CFG 5."""
  g = np.load(os.path.join(GOLD, 'base_with_context_n1000.npz'))
  model, got = _song('base_with_context', int(g['n_segments']), int(g['noise_seed']))
  t = 256
  e0, e1 = helpers.rms(got[:, :t], g['mel'][:, :t]), helpers.rms(got[:, t:], g['mel'][:, t:])
  print('base_with_context 1000-step rms vs float64 oracle: segment 0 %.3e, segment 1 (chained) %.3e' % (e0, e1))
  assert e0 <= 1e-3 and e1 <= 1e-3
  # size-independent properties at full size: determinism and range
  lo, hi = model.audio_codec.min_value, model.audio_codec.max_value
  assert got.min() >= lo - 1e-4 and got.max() <= hi + 1e-4   # final x0 is clipped to [-1, 1]
  assert np.isfinite(got).all()
