"""The exact-shortcut oracle (oracle/fast.py: tables, cached cross K/V, dropped
padding, skipped unconditional cross-attention) against the faithful restatement."""
import numpy as np
import pytest

import msd_amd
from oracle import backend, fast, predict
from tests import helpers


@pytest.mark.parametrize('preset,mask', [('tiny', 'ones'), ('tiny_context', 'ones'),
                                         ('tiny_context', 'zeros'), ('tiny_context', 'ragged')])
def test_fast_equals_faithful_float64(preset, mask):
  spec = msd_amd.config.preset(preset, num_steps=5)
  params = msd_amd.synthetic.init_params(spec, 1, norm_scale_jitter=0.1)
  batch = helpers.make_batch(spec, batch=2, ctx_mask=mask)
  init_z, noise = helpers.make_noise(spec, batch=2)
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.NumpyBackend('float64')
  ref, _ = predict.predict_batch_with_aux(xp, cfg, dc, params, batch, init_z, noise,
                                          context=spec.has_context)
  got, _ = fast.FastModel(xp, cfg, dc, params, spec.has_context).predict(batch, init_z, noise)
  np.testing.assert_allclose(got, ref, atol=1e-9)


@pytest.mark.parametrize('preset,mask', [('tiny', 'ones'), ('tiny_context', 'zeros'), ('tiny_context', 'ragged')])
def test_fast_equals_faithful_sum_cross_attends(preset, mask):
  """decoder_cross_attend_style='sum_cross_attends' (the T5Config default, network.py:199-216): one
  cross-attention module per encoding; the shortcuts (cached K/V per module, dropped padding, S4) hold."""
  import dataclasses
  spec = msd_amd.config.preset(preset, num_steps=5)
  spec = dataclasses.replace(spec, t5=dataclasses.replace(spec.t5, decoder_cross_attend_style='sum_cross_attends'))
  params = msd_amd.synthetic.init_params(spec, 1, norm_scale_jitter=0.1)
  assert ('decoder/layers_0/MultiHeadDotProductAttention_1/query/kernel' in params) == spec.has_context
  batch = helpers.make_batch(spec, batch=2, ctx_mask=mask)
  init_z, noise = helpers.make_noise(spec, batch=2)
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.NumpyBackend('float64')
  ref, _ = predict.predict_batch_with_aux(xp, cfg, dc, params, batch, init_z, noise, context=spec.has_context)
  got, _ = fast.FastModel(xp, cfg, dc, params, spec.has_context).predict(batch, init_z, noise)
  np.testing.assert_allclose(got, ref, atol=1e-9)


def test_fast_torch_backend_and_bf16x3_close_to_f32():
  spec = msd_amd.config.preset('tiny_context', num_steps=5)
  params = msd_amd.synthetic.init_params(spec, 1)
  batch = helpers.make_batch(spec)
  init_z, noise = helpers.make_noise(spec)
  cfg, dc = helpers.oracle_configs(spec)
  xp64 = backend.NumpyBackend('float64')
  ref = fast.FastModel(xp64, cfg, dc, params, True).predict(batch, init_z, noise)[0]
  xt = backend.TorchBackend('float32')
  f32 = xt.to_numpy(fast.FastModel(xt, cfg, dc, params, True).predict(batch, init_z, noise)[0])
  x3 = xt.to_numpy(fast.FastModel(xt, cfg, dc, params, True, precision='bf16x3').predict(
      batch, init_z, noise)[0])
  b16 = xt.to_numpy(fast.FastModel(xt, cfg, dc, params, True, precision='bf16').predict(
      batch, init_z, noise)[0])
  e32, e3, e16 = helpers.rms(f32, ref), helpers.rms(x3, ref), helpers.rms(b16, ref)
  # few, huge DDPM steps amplify rounding (x0 = sqrt(1+e^-l)(z - sigma eps)): float32
  # itself is ~1e-2 mel units from float64 here; bf16x3 must sit at that same floor
  print('rms vs float64: f32 %.3e  bf16x3 %.3e  bf16 %.3e' % (e32, e3, e16))
  assert e3 < 3 * e32 + 1e-4
  assert e16 > 5 * e3              # plain bf16 is far off: why bf16x3 is the parity mode


def test_numpy_bf16_rounding_matches_torch():
  rng = np.random.default_rng(0)
  x = (rng.standard_normal(10000) * 10.0 ** rng.integers(-6, 6, 10000)).astype(np.float32)
  a = backend.NumpyBackend('float32').round_bf16(x)
  xt = backend.TorchBackend('float32')
  b = xt.to_numpy(xt.round_bf16(xt.asarray(x)))
  np.testing.assert_array_equal(a, b)


def test_numpy_f16_rounding_matches_torch_and_saturates():
  rng = np.random.default_rng(1)
  x = (rng.standard_normal(10000) * 10.0 ** rng.integers(-9, 7, 10000)).astype(np.float32)
  a = backend.NumpyBackend('float32').round_f16(x)
  xt = backend.TorchBackend('float32')
  b = xt.to_numpy(xt.round_f16(xt.asarray(x)))
  np.testing.assert_array_equal(a, b)
  assert np.isfinite(a).all() and np.abs(a).max() == 65504.0


def test_half_planes_beat_bfloat16_planes_on_a_decoder_pass():
  """One decoder pass (no sampler in the way): operands as fp16 hi + lo (22 significand bits) are an order of
  magnitude closer to float64 than as bf16 hi + lo (16 bits) -- the emulation behind DESIGN 3's fp16-plane plan."""
  spec = msd_amd.config.preset('tiny_context', num_steps=10)
  params = msd_amd.synthetic.init_params(spec, 2, norm_scale_jitter=0.1)
  batch = helpers.make_batch(spec, batch=2, ctx_mask='ragged')
  cfg, dc = helpers.oracle_configs(spec)
  z = np.random.default_rng(0).standard_normal((2, 64, 128))
  outs = {}
  for prec, dtype in (('f32', 'float64'), ('f32', 'float32'), ('bf16x3', 'float32'), ('f16x3', 'float32')):
    xp = backend.TorchBackend(dtype, threads=1)
    fm = fast.FastModel(xp, cfg, dc, params, True, precision=prec)
    fm.encode(batch['encoder_input_tokens'], batch['encoder_continuous_inputs'], batch['encoder_continuous_mask'])
    outs[prec, dtype] = xp.to_numpy(fm.decoder_pass(xp.asarray(z), 4, True)).astype(np.float64)
  # against the float32 run of the same statements: float32's own distance to float64 (4e-5 here) is the time
  # embedding (sin / cos of arguments up to 2e4 in float32), shared by every float32-class mode
  f32 = outs['f32', 'float32']
  e32 = helpers.rms(f32, outs['f32', 'float64'])
  eb, eh = helpers.rms(outs['bf16x3', 'float32'], f32), helpers.rms(outs['f16x3', 'float32'], f32)
  print('decoder pass: float32 vs float64 %.2e | vs float32: bf16x3 %.2e  f16x3 %.2e' % (e32, eb, eh))
  assert eh < 0.1 * eb
