"""Cases of tests/golden/ref_*.npz: fixtures produced by the REFERENCE's own code
(tests/golden/make_ref_golden.py runs layers.py / network.py / diffusion_utils.py / models.py from
/root/reference over the NumPy stand-in of jax + flax in tests/golden/ref_shim.py, float64).
Inputs are rebuilt from seeds here, so generator, CPU tests (oracle vs fixture) and GPU tests (device vs
fixture) see identical tensors; the fixture stores digests of them."""
import dataclasses
import hashlib

import numpy as np

import msd_amd
from tests import helpers


def _variant(preset, steps, cross=None, sampler=None, model_output=None, logvar=None, schedule=None,
             train_schedule=None, cfg_weight=5.0, clip=True, t5=None):
  spec = msd_amd.config.preset(preset, num_steps=steps, cfg_weight=cfg_weight)
  if t5:
    spec = dataclasses.replace(spec, t5=dataclasses.replace(spec.t5, **t5))
  if cross:
    spec = dataclasses.replace(spec, t5=dataclasses.replace(spec.t5, decoder_cross_attend_style=cross))
  d = spec.diffusion
  s = d.sampler
  if schedule:
    s = dataclasses.replace(s, schedule=dataclasses.replace(s.schedule, **schedule))
  s = dataclasses.replace(s, name=sampler or s.name, logvar_type=logvar or s.logvar_type, clip_x0=clip)
  d = dataclasses.replace(d, sampler=s, model_output=model_output or d.model_output)
  if train_schedule:
    d = dataclasses.replace(d, train_schedule=dataclasses.replace(d.train_schedule, **train_schedule))
  return dataclasses.replace(spec, diffusion=d)


# name -> (spec, batch size, context mask kind, weight seed, batch seed, noise seed)
def cases():
  lin = dict(name='linear', start=1e-4, stop=2e-2)
  return {
      'tiny_context_ddpm': (_variant('tiny_context', 6), 2, 'ragged', 0, 3, 11),
      'tiny_context_zero_mask': (_variant('tiny_context', 4), 2, 'zeros', 1, 4, 12),
      'tiny_ddpm': (_variant('tiny', 6), 2, 'ones', 2, 5, 13),
      'tiny_context_sum_cross': (_variant('tiny_context', 5, cross='sum_cross_attends'), 2, 'ragged', 3, 6, 14),
      'tiny_sum_cross': (_variant('tiny', 4, cross='sum_cross_attends'), 1, 'ones', 4, 7, 15),
      'tiny_context_ddim': (_variant('tiny_context', 6, sampler='ddim'), 2, 'ragged', 5, 8, 16),
      'tiny_context_v_small': (_variant('tiny_context', 6, model_output='v', logvar='small'), 2, 'ones', 6, 9, 17),
      'tiny_context_x0_medium_noclip': (_variant('tiny_context', 6, model_output='x0', logvar='medium:0.3', clip=False),
                                        2, 'ragged', 7, 10, 18),
      'tiny_context_linear': (_variant('tiny_context', 8, schedule=lin, train_schedule=dict(lin, num_steps=8)),
                              2, 'ragged', 8, 11, 19),
      'tiny_context_w1': (_variant('tiny_context', 4, cfg_weight=1.0), 2, 'ragged', 9, 12, 20),
      'tiny_context_regular_positions': (_variant('tiny_context', 5, t5=dict(context_positions='regular',
                                                                             position_encoding='fixed')),
                                         2, 'ragged', 11, 16, 24),
      # the full 1000-step chain on the tiny model (the bar of north_star applies: 1e-3 rms)
      'tiny_context_n1000': (_variant('tiny_context', 1000), 1, 'ragged', 10, 15, 23),
      # full-size shapes (BASELINE configs 2 and 3), a few steps
      'base_with_context_n3': (_variant('base_with_context', 3), 1, 'ones', 0, 13, 21),
      'small_n3': (_variant('small', 3), 1, 'ones', 0, 14, 22),
  }


WITH_ENCODINGS = ('tiny_context_ddpm', 'tiny_context_zero_mask', 'tiny_ddpm')


def pass_z(shape):
  """z of the stored single decoder passes."""
  return np.random.default_rng(99).standard_normal(shape)


def inputs(name):
  spec, b, mask, wseed, bseed, nseed = cases()[name]
  params = msd_amd.synthetic.init_params(spec, wseed, norm_scale_jitter=0.1)
  batch = helpers.make_batch(spec, batch=b, seed=bseed, ctx_mask=mask)
  if name.startswith(('base', 'small')):   # realistic token counts at full size, not the random 8..2046
    batch['encoder_input_tokens'] = np.concatenate(
        [msd_amd.synthetic.segment_tokens(spec, k) for k in range(b)], 0)
  init_z, noise = helpers.make_noise(spec, batch=b, seed=nseed)
  return spec, params, batch, init_z, noise


def digest(params, batch, init_z, noise):
  h = hashlib.sha256()
  for k in sorted(params):
    h.update(k.encode())
    h.update(np.ascontiguousarray(params[k]).tobytes())
  for k in sorted(batch):
    h.update(k.encode())
    h.update(np.ascontiguousarray(batch[k]).tobytes())
  h.update(np.ascontiguousarray(init_z).tobytes())
  h.update(np.ascontiguousarray(noise).tobytes())
  return h.hexdigest()[:16]
