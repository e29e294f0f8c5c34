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
  model = msd_amd.InferenceModel(params, spec, batch_size=2, **helpers.ALL_PLANES)
  got, _ = model.predict(batch, init_z=init_z, noise=noise)
  # 6 huge steps are ill-conditioned (helpers.assert_fp32_class): the yardstick is the float32 oracle's
  # own deviation from the float64 fixture, on the bulk, on the outlier count AND on the rms (x3)
  from oracle import backend, fast
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend('float32')
  ref32 = xp.to_numpy(fast.FastModel(xp, cfg, dc, params, True).predict(batch, init_z, noise)[0])
  helpers.assert_fp32_class(got, g['mel'].astype(np.float64), ref32, 'tiny golden')
  assert helpers.rms(got, g['mel']) <= 3 * helpers.rms(ref32, g['mel']) + 1e-4


def _song(preset, n_segments, noise_seed, teacher=None, precision='f16x3'):
  """Device run with the SAME inputs make_golden.py used (tokens, Philox noise, chaining: every
  segment's context is the DEVICE's own previous prediction, as in beam/evaluation.py:191-223; with
  `teacher` [1, n_segments*T, n] the context of segment k is teacher's segment k-1 instead)."""
  from oracle import philox
  import torch
  spec = msd_amd.config.preset(preset, num_steps=1000)
  model = msd_amd.InferenceModel('synthetic:0', spec, precision=precision)
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
    if teacher is not None:
      assert c == t
      pred = np.ascontiguousarray(teacher[:, k * t:(k + 1) * t])
  return model, np.concatenate(outs, 1)


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['f16x3', 'bf16x3'])
def test_small_1000_steps_within_1e3_rms(precision):
  """BASELINE config 2 shape (small, no context, 1000-step DDPM, CFG 5); both library builds (half planes:
  emulation 6.8e-5 = the float32 oracle's own error; bfloat16 planes: 1.4e-4)."""
  g = np.load(os.path.join(GOLD, 'small_n1000.npz'))
  _, got = _song('small', int(g['n_segments']), int(g['noise_seed']), precision=precision)
  err = helpers.rms(got, g['mel'])
  print('small 1000-step rms vs float64 oracle, %s: %.3e' % (precision, err))
  assert err <= (1e-4 if precision == 'f16x3' else 2e-4)


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


@pytest.mark.gpu
def test_base_with_context_chain_depth():
  """Chain depth (beam/evaluation.py:191-223): segment k+1 is conditioned on the DEVICE's own segment k; the
  bench chains 20 segments, a 10-minute song 118.  Fixture: the float64 oracle's free-running 12-segment song
  plus the float32 oracle's OWN free-running song (`rms_f32`: how far the reference's arithmetic drifts from
  float64 at each depth: 0.6e-4 -> 4.9e-4, roughly linear -- a free-running song is a feedback loop).
  Measured (MI355X): with the half operand planes (default) the device is ON the float32 oracle's drift, x0.9 ..
  x1.1 at every depth, 0.63e-4 -> 5.3e-4 (profiles/r02m_gpu_tests.log); with the bfloat16 planes of the other
  build it runs 3x .. 8x above it and leaves 1e-3 at depth 6 (profiles/r02k_chain12.log).
  Asserted: north_star's 1e-3 at every depth, and never more than twice the float32 oracle's own drift."""
  g = np.load(os.path.join(GOLD, 'base_chain_n1000.npz'))
  n_seg, t = int(g['n_segments']), 256
  assert n_seg >= 12
  _, got = _song('base_with_context', n_seg, int(g['noise_seed']))
  rows, ok = [], True
  for k in range(n_seg):
    e = helpers.rms(got[:, k * t:(k + 1) * t], g['mel'][:, k * t:(k + 1) * t])
    f = float(g['rms_f32'][k])
    bar = min(1e-3, 2 * f)
    rows.append('segment %2d: device %.3e | float32 oracle %.3e | x%.1f | bar %.1e %s'
                % (k, e, f, e / f, bar, 'ok' if e <= bar else 'FAIL'))
    ok = ok and e <= bar
  print('base_with_context free-running song, rms vs float64 oracle per segment:\n  ' + '\n  '.join(rows))
  assert ok, rows


@pytest.mark.gpu
def test_base_with_context_every_segment_on_the_reference_context():
  """The same 12 segments with the context the float64 fixture holds (segment k conditioned on the FIXTURE's
  segment k-1): identical inputs, so north_star's bar applies at every depth -- rms <= 1e-3 per segment
  (measured 0.57e-4 .. 0.87e-4 with half planes, 1.5e-4 .. 2.2e-4 with bfloat16 planes)."""
  g = np.load(os.path.join(GOLD, 'base_chain_n1000.npz'))
  n_seg, t = int(g['n_segments']), 256
  _, got = _song('base_with_context', n_seg, int(g['noise_seed']), teacher=g['mel'].astype(np.float32))
  errs = [helpers.rms(got[:, k * t:(k + 1) * t], g['mel'][:, k * t:(k + 1) * t]) for k in range(n_seg)]
  print('base_with_context, every segment on the fixture\'s context, rms vs float64 oracle: '
        + ' '.join('%.2e' % e for e in errs))
  assert max(errs) <= 1e-3, errs


def test_trained_like_reshapes_the_dynamic_range():
  """synthetic.trained_like: log-normal channel gains + outlier channels, same power, same keys/shapes."""
  spec = msd_amd.config.preset('tiny_context', num_steps=4)
  base = msd_amd.synthetic.init_params(spec, 0)
  tl = msd_amd.synthetic.trained_like(base, seed=1)
  assert tl.keys() == base.keys()
  kernels = [k for k in base if k.endswith('/kernel')]
  assert kernels
  spread = []
  for k in kernels:
    assert tl[k].shape == base[k].shape and tl[k].dtype == np.float32
    g = np.sqrt((tl[k].astype(np.float64) ** 2).mean(0) / np.maximum((base[k].astype(np.float64) ** 2).mean(0), 1e-30))
    assert abs(np.sqrt((g ** 2).mean()) - 1) < 1e-3          # overall output power kept
    spread.append(g.max() / g.min())
  assert np.median(spread) > 4                                # channels now span > 4x in gain
  scales = [k for k in base if k.endswith('/scale')]
  assert any(np.std(tl[k] / base[k]) > 0.2 for k in scales)
  again = msd_amd.synthetic.trained_like(base, seed=1)
  assert all(np.array_equal(again[k], tl[k]) for k in tl)


@pytest.mark.gpu
def test_small_trained_like_weights_1000_steps():
  """Same bar with weights whose dynamic range is reshaped towards a trained model's (log-normal channel
  gains, x6 outlier channels, log-normal norm scales: synthetic.trained_like): the split-bf16 planes must
  not be tuned to fresh initialisers.  Bar: rms <= 1e-3, or 2x the float32 oracle's own deviation from the
  float64 fixture where that is beyond 1e-3."""
  from oracle import philox
  path = os.path.join(GOLD, 'small_trained_like_n1000.npz')
  g = np.load(path)
  spec = msd_amd.config.preset('small', num_steps=1000)
  params = msd_amd.synthetic.trained_like(msd_amd.synthetic.init_params(spec, int(g['weight_seed'])),
                                          seed=int(g['reshape_seed']))
  model = msd_amd.InferenceModel(params, spec)
  t = spec.task_feature_lengths['targets']
  batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 0)}
  init_z, noise = philox.segment_noise((1, t, 128), 1000, seed=int(g['noise_seed']), segment=0)
  got, _ = model.predict(batch, init_z=init_z, noise=noise)
  err, f = helpers.rms(got, g['mel']), float(g['rms_f32'])
  bar = 1e-3 if f <= 1e-3 else 2 * f
  print('small, trained-like weights, 1000 steps: device %.3e | float32 oracle %.3e | bar %.1e' % (err, f, bar))
  assert err <= bar


@pytest.mark.gpu
@pytest.mark.parametrize('gain', [1, 2, 4])
def test_small_sharp_attention_1000_steps(gain):
  """SHARP attention: every decoder query kernel times `gain` (synthetic.sharp_attention), i.e.
  every attention logit times `gain` -- competing keys 10 - 30 apart instead of the O(1) logits of fresh initialisers.
  Fixture: the float64 oracle's 1000-step `small` segment with those weights plus the float32 oracle's own rms
  (tests/diag/sharp_attention_study.py --golden).  The DEFAULT mode (all planes) must sit on the float32 floor
  (<= 1.3x: the adoption criterion of round 3); the opt-in single query-side plane is run beside it and only has to
  meet north_star's 1e-3 bar -- its ratio to the floor is printed (emulation: 2.8x at gain 4).  (Gain 8 is no fixture:
  the segment is then chaotic in float32 itself -- the float32 oracle ends 0.67 rms away from the float64 one.)"""
  from oracle import philox
  path = os.path.join(GOLD, 'small_sharp%d_n1000.npz' % gain)
  if not os.path.exists(path):
    pytest.skip('fixture not generated: python -m tests.diag.sharp_attention_study --golden %d' % gain)
  g = np.load(path)
  spec = msd_amd.config.preset('small', num_steps=1000)
  params = msd_amd.synthetic.sharp_attention(msd_amd.synthetic.init_params(spec, 0), float(g['gain']))
  t = spec.task_feature_lengths['targets']
  batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 0)}
  init_z, noise = philox.segment_noise((1, t, 128), 1000, seed=int(g['noise_seed']), segment=0)
  floor = float(g['rms_f32'])
  errs = {}
  for name, kw in (('default', {}), ('P one plane', dict(attention_query_planes=(2, 1))),
                   ('Q and P one plane', helpers.ONE_QUERY_PLANE)):
    model = msd_amd.InferenceModel(params, spec, **kw)
    got, _ = model.predict(batch, init_z=init_z, noise=noise)
    errs[name] = helpers.rms(got, g['mel'])
    del model
  print('small, decoder logits x%d, 1000 steps: float32 oracle %.3e | device default %.3e (x%.2f) | P one plane %.3e (x%.2f) '
        '| Q and P one plane %.3e (x%.2f)' % (gain, floor, errs['default'], errs['default'] / floor, errs['P one plane'],
                                              errs['P one plane'] / floor, errs['Q and P one plane'], errs['Q and P one plane'] / floor))
  assert errs['default'] <= 1.3 * floor and errs['default'] <= 1e-3
  # the opt-in modes are gated too: north_star's bar for both, and P alone stays float32-class
  # (measured at gain 4: P one plane x1.10 of the floor, Q and P one plane 4.5e-4)
  assert errs['P one plane'] <= 1.3 * floor and errs['P one plane'] <= 1e-3
  assert errs['Q and P one plane'] <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['trained', 'sharp2'])
def test_base_with_context_robustness_1000_steps(kind):
  """The HEADLINE model off its fresh-initialiser comfort zone (the real checkpoints are trained,
  gin/models/diffusion/context/t5_base.gin:70-83): base_with_context, one 1000-step segment conditioned on a realistic
  previous prediction, with `trained` = synthetic.trained_like weights (log-normal channel gains, x6 outlier channels,
  log-normal norm scales) and with `sharp2` = every decoder attention logit doubled.  Fixture: float64 oracle + the
  float32 oracle's own rms (tests/golden/make_golden.py robust_*).  Default mode (f16x3, all planes): north_star's
  1e-3, float32-class (<= 1.3 x the float32 oracle's own error), and the half-plane range holds -- the model is built
  with range_fallback=False, so an activation beyond 65504 would fail the test instead of silently switching planes."""
  import sys
  from oracle import philox
  path = os.path.join(GOLD, 'base_%s_n1000.npz' % kind)
  if not os.path.exists(path):
    pytest.skip('fixture not generated: python tests/golden/make_golden.py robust_%s64 robust_%s32 robust_%s_pack' % (kind, kind, kind))
  sys.path.insert(0, GOLD)
  import make_golden
  g = np.load(path)
  spec = msd_amd.config.preset('base_with_context', num_steps=1000)
  params = make_golden.robust_params(spec, kind)
  batch = make_golden.robust_batch(spec)
  t = spec.task_feature_lengths['targets']
  init_z, noise = philox.segment_noise((1, t, 128), 1000, seed=int(g['noise_seed']), segment=int(g['segment']))
  model = msd_amd.InferenceModel(params, spec, range_fallback=False)
  got, _ = model.predict(batch, init_z=init_z, noise=noise)
  assert model.precision == 'f16x3'
  err, floor = helpers.rms(got, g['mel']), float(g['rms_f32'])
  print('base_with_context, %s weights, 1000 steps: device %.3e | float32 oracle %.3e | x%.2f' % (kind, err, floor, err / floor))
  assert err <= 1e-3 and err <= 1.3 * floor
