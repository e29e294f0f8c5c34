"""Pins derived from the maths for the parts of the path no reference test covers
(diffusion_utils / network / models / audio_codecs), SURVEY.md 8(c) "Extra pins"."""
import numpy as np
import pytest

import msd_amd
from oracle import backend, net, ops, predict, sampler
from tests import helpers

XP64 = backend.NumpyBackend('float64')
XP32 = backend.NumpyBackend('float32')
COS = sampler.DiffusionSchedule('cosine', num_steps=1000)


def test_logsnr_cosine_endpoints():
  t = np.array([0.0, 0.5, 1.0])
  np.testing.assert_allclose(sampler.get_logsnr_t(XP64, t, COS), [20.0, 0.0, -20.0], atol=1e-9)


def test_x0_eps_round_trip():
  rng = np.random.default_rng(0)
  z, eps = rng.standard_normal((2, 5, 7)), rng.standard_normal((2, 5, 7))
  logsnr = np.array([-3.0, 4.0])
  x0 = sampler.predict_x0_from_eps(XP64, z=z, eps=eps, logsnr=logsnr)
  np.testing.assert_allclose(sampler.predict_eps_from_x0(XP64, z=z, x0=x0, logsnr=logsnr), eps,
                             atol=1e-10)


def test_reverse_large_variance_and_mean():
  lt, ls = np.array([-1.0]), np.array([0.5])
  rng = np.random.default_rng(1)
  x0, zt = rng.standard_normal((1, 3)), rng.standard_normal((1, 3))
  d = sampler.diffusion_reverse(XP64, x0=x0, z_t=zt, logsnr_s=ls, logsnr_t=lt, logvar_type='large')
  sig = lambda x: 1 / (1 + np.exp(-x))
  r = np.exp(lt - ls)
  np.testing.assert_allclose(d['var'], (1 - r) * sig(-lt), rtol=1e-12)
  a_t, a_s = np.sqrt(sig(lt)), np.sqrt(sig(ls))
  np.testing.assert_allclose(d['mean'], r * (a_s / a_t) * zt + (1 - r) * a_s * x0, rtol=1e-12)
  np.testing.assert_allclose(np.exp(d['logvar']), d['var'], rtol=1e-10)


def test_step_zero_returns_clipped_x0():
  dc = sampler.DiffusionConfig(
      sampler=sampler.SamplerConfig(schedule=sampler.DiffusionSchedule('cosine', num_steps=4)))
  rng = np.random.default_rng(2)
  z = rng.standard_normal((1, 3, 4))
  pred = lambda z, time, include_conditioning: 3.0 * z
  body = sampler.eval_step(XP64, rng.standard_normal((4, 1, 3, 4)), dc, 1, pred)
  out = body(z, 0)
  w = dc.classifier_free_guidance.eval_condition_weight
  eps = w * 3 * z + (1 - w) * 3 * z
  lt = sampler.get_logsnr_t(XP64, np.array([0.25]), dc.sampler.schedule)
  x0 = np.clip(sampler.predict_x0_from_eps(XP64, z=z, eps=eps, logsnr=lt), -1, 1)
  np.testing.assert_allclose(out, x0, atol=1e-12)


def test_scale_round_trip():
  codec = predict.MelGANCodec()
  x = np.linspace(codec.min_value, codec.max_value, 101)
  y = codec.scale_features(XP64, x, clip=True)
  assert y.min() == pytest.approx(-1) and y.max() == pytest.approx(1)
  np.testing.assert_allclose(codec.scale_to_features(XP64, y), x, atol=1e-12)
  host = msd_amd.audio_codecs.MelGAN()
  np.testing.assert_allclose(host.scale_features(x, clip=True), y, atol=1e-12)
  np.testing.assert_allclose(host.scale_to_features(y), x, atol=1e-12)


def test_param_counts():
  assert msd_amd.config.param_count(msd_amd.config.preset('base_with_context')) == 411_665_664
  assert msd_amd.config.param_count(msd_amd.config.preset('small')) == 84_956_160
  assert msd_amd.config.num_embeddings(1) == 1536


@pytest.fixture(scope='module')
def tiny():
  spec = msd_amd.config.preset('tiny_context', num_steps=6)
  params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
  return spec, params


def _encode(xp, spec, params, batch):
  cfg, _ = helpers.oracle_configs(spec)
  p = {k: xp.asarray(v) for k, v in params.items()}
  ctx = predict.MelGANCodec().scale_features(xp, xp.asarray(batch['encoder_continuous_inputs']),
                                             clip=True)
  return cfg, p, net.context_transformer_encode(xp, cfg, p, batch['encoder_input_tokens'], ctx,
                                                xp.asarray(batch['encoder_continuous_mask']))


def test_uncond_pass_is_decoder_without_cross_attention(tiny):
  """encodings*0 and masks*0 -> zero_activations_if_masked
  zeroes the cross-attention branch exactly."""
  spec, params = tiny
  batch = helpers.make_batch(spec)
  cfg, p, enc = _encode(XP64, spec, params, batch)
  z = np.random.default_rng(0).standard_normal((1, 64, 128))
  t = np.array([0.4])
  uncond = net.decode(XP64, cfg, p, [(e * 0.0, m * 0.0) for e, m in enc], z, t)
  # decoder with zero-width encodings is not expressible; instead check invariance
  # to the encoding VALUES once masks are zero:
  junk = [(np.random.default_rng(5).standard_normal(e.shape), m * 0.0) for e, m in enc]
  np.testing.assert_allclose(net.decode(XP64, cfg, p, junk, z, t), uncond, atol=1e-12)


def test_fully_masked_context_equals_token_only(tiny):
  spec, params = tiny
  batch = helpers.make_batch(spec, ctx_mask='zeros')
  cfg, p, enc = _encode(XP64, spec, params, batch)
  z = np.random.default_rng(0).standard_normal((1, 64, 128))
  t = np.array([0.7])
  both = net.decode(XP64, cfg, p, enc, z, t)
  tok_only = net.decode(XP64, cfg, p, enc[:1], z, t)
  np.testing.assert_allclose(both, tok_only, atol=1e-10)


def test_cfg_weight_one_is_single_pass(tiny):
  spec, params = tiny
  batch = helpers.make_batch(spec)
  init_z, noise = helpers.make_noise(spec)
  cfg, dc = helpers.oracle_configs(spec)
  calls = []
  dc1 = sampler.DiffusionConfig(
      classifier_free_guidance=sampler.ClassifierFreeGuidanceConfig(eval_condition_weight=1.0),
      sampler=dc.sampler)
  orig = net.decode

  def counting(*a, **k):
    calls.append(1)
    return orig(*a, **k)

  net.decode = counting
  try:
    predict.predict_batch_with_aux(XP32, cfg, dc1, params, batch, init_z, noise)
  finally:
    net.decode = orig
  assert len(calls) == dc.sampler.schedule.num_steps  # one decoder call per step


def test_float32_oracle_tracks_float64_single_pass(tiny):
  spec, params = tiny
  batch = helpers.make_batch(spec)
  z = np.random.default_rng(0).standard_normal((1, 64, 128))
  outs = []
  for xp in (XP64, XP32):
    cfg, p, enc = _encode(xp, spec, params, batch)
    outs.append(np.asarray(net.decode(xp, cfg, p, enc, xp.asarray(z), xp.asarray([0.3])), np.float64))
  assert helpers.rms(outs[0], outs[1]) <= 1e-5 * max(1.0, np.abs(outs[0]).max())


def test_terminal_relative_positions():
  xp = XP32
  assert net.get_sequence_length(xp, np.array([1, 1, 0, 0, 0])) == 2
  assert net.get_sequence_length(xp, np.array([1, 1, 1, 1, 1])) == 5
  assert net.get_sequence_length(xp, np.array([0, 0, 0])) == 0
  # network.py:45-51 docstring example: max length 5, seq len 2 -> [3, 4, 0, 1, 2]
  np.testing.assert_array_equal(np.roll(np.arange(5), 2), [3, 4, 0, 1, 2])
