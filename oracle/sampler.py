"""Oracle restatement of the reference ``models/diffusion/diffusion_utils.py``
(evaluation path only).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  No reference test covers
this file; it is a line-by-line restatement, pinned by the identities in
tests/test_oracle_identities.py and, through whole sampled segments of every
branch (ddpm / ddim, eps / x0 / v, large / small / medium, cosine / linear, w = 1),
against the reference's own diffusion_utils.eval_scan executed over a NumPy
stand-in of jax (tests/golden/ref_*.npz, tests/test_ref_golden.py).

RNG contract: the reference draws ``init_z = normal(rng)`` and per-step
``normal(fold_in(rng, i))`` from jax.random (threefry; jax version unpinned,
jax not installable here).  The oracle therefore takes the noise as EXPLICIT
inputs: ``init_z [B,T,n]`` and ``noise [N,B,T,n]`` where ``noise[i]`` is the draw
used at scan index ``i`` (``noise[0]`` is drawn but unused, diffusion_utils.py:395).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Callable, Optional

import numpy as np


@dataclasses.dataclass(frozen=True)
class DiffusionSchedule:
  """diffusion_utils.py:25-30."""
  name: str
  start: Optional[float] = None
  stop: Optional[float] = None
  num_steps: Optional[int] = None


@dataclasses.dataclass(frozen=True)
class ClassifierFreeGuidanceConfig:
  """diffusion_utils.py:33-36."""
  drop_condition_prob: float = 0.1
  eval_condition_weight: float = 5.0


@dataclasses.dataclass(frozen=True)
class SamplerConfig:
  """diffusion_utils.py:39-46."""
  name: str = 'ddpm'
  schedule: DiffusionSchedule = DiffusionSchedule(name='cosine', num_steps=1000)
  clip_x0: bool = True
  logvar_type: str = 'large'


@dataclasses.dataclass(frozen=True)
class DiffusionConfig:
  """diffusion_utils.py:49-59."""
  time_continuous_or_discrete: str = 'continuous'
  train_schedule: DiffusionSchedule = DiffusionSchedule(name='cosine')
  loss_norm: str = 'l1'
  loss_type: str = 'eps'
  model_output: str = 'eps'
  classifier_free_guidance: ClassifierFreeGuidanceConfig = ClassifierFreeGuidanceConfig()
  sampler: SamplerConfig = SamplerConfig()


def broadcast_to_shape_from_left(xp, x, shape):
  """diffusion_utils.py:62-66."""
  assert len(shape) >= x.ndim
  return xp.reshape(x, tuple(x.shape) + (1,) * (len(shape) - x.ndim))


def get_timing_signal_1d(xp, position, num_channels, min_timescale=1.0,
                         max_timescale=2.0e4):
  """diffusion_utils.py:69-97: [sin(p w_k) | cos(p w_k)], shape [batch, channels]."""
  assert position.ndim == 1
  assert num_channels % 2 == 0
  num_timescales = float(num_channels // 2)
  log_timescale_increment = (
      math.log(max_timescale / min_timescale) / (num_timescales - 1.0))
  inv_timescales = min_timescale * xp.exp(
      xp.arange(int(num_timescales)) * -log_timescale_increment)
  scaled_time = xp.expand_dims(position, 1) * xp.expand_dims(inv_timescales, 0)
  return xp.concatenate([xp.sin(scaled_time), xp.cos(scaled_time)], 1)


def log1mexp(xp, x):
  """diffusion_utils.py:100-106."""
  return xp.where(x > math.log(2.0), xp.log1p(-xp.exp(-x)), xp.log(-xp.expm1(-x)))


def log_sigmoid(xp, x):
  """jax.nn.log_sigmoid(x) = -softplus(-x)."""
  return -(xp.maximum(-x, 0.0 * x) + xp.log1p(xp.exp(-abs(x))))


def diffusion_reverse(xp, *, x0, z_t, logsnr_s, logsnr_t, logvar_type):
  """q(z_s | z_t, x0), diffusion_utils.py:120-163."""
  alpha_st = xp.sqrt((1. + xp.exp(-logsnr_t)) / (1. + xp.exp(-logsnr_s)))
  alpha_s = xp.sqrt(xp.sigmoid(logsnr_s))
  r = xp.exp(logsnr_t - logsnr_s)
  one_minus_r = -xp.expm1(logsnr_t - logsnr_s)
  log_one_minus_r = log1mexp(xp, logsnr_s - logsnr_t)
  mean = r * alpha_st * z_t + one_minus_r * alpha_s * x0
  if logvar_type == 'small':
    var = one_minus_r * xp.sigmoid(-logsnr_s)
    logvar = log_one_minus_r + log_sigmoid(xp, -logsnr_s)
  elif logvar_type == 'large':
    var = one_minus_r * xp.sigmoid(-logsnr_t)
    logvar = log_one_minus_r + log_sigmoid(xp, -logsnr_t)
  elif logvar_type.startswith('medium:'):
    _, frac = logvar_type.split(':')
    frac = float(frac)
    assert 0 <= frac <= 1
    min_logvar = log_one_minus_r + log_sigmoid(xp, -logsnr_s)
    max_logvar = log_one_minus_r + log_sigmoid(xp, -logsnr_t)
    logvar = frac * max_logvar + (1 - frac) * min_logvar
    var = xp.exp(logvar)
  else:
    raise ValueError('Unknown logvar_type: %s' % logvar_type)
  return {'mean': mean, 'std': xp.sqrt(var), 'var': var, 'logvar': logvar}


def get_logsnr_t(xp, t, schedule):
  """diffusion_utils.py:166-202.  ``t`` is an array in the backend dtype."""
  logsnr_min, logsnr_max = -20.0, 20.0
  if schedule.name == 'cosine':
    b = float(np.arctan(np.exp(-0.5 * logsnr_max)))
    a = float(np.arctan(np.exp(-0.5 * logsnr_min)) - b)
    return -2.0 * xp.log(xp.tan(a * t + b))
  elif schedule.name == 'linear':
    assert schedule.num_steps > 0
    betas = np.linspace(schedule.start, schedule.stop, schedule.num_steps,
                        dtype=np.float64)
    alphas_cumprod = np.cumprod(1. - betas, axis=0)
    logsnr = np.log(alphas_cumprod) - np.log1p(-alphas_cumprod)
    logsnr = np.clip(logsnr, logsnr_min, logsnr_max)
    grid = np.linspace(0, 1, schedule.num_steps)
    return xp.asarray(np.interp(xp.to_numpy(t).astype(np.float64), grid, logsnr))
  raise ValueError('Schedule %s not identified.' % schedule.name)


def predict_eps_from_x0(xp, *, z, x0, logsnr):
  """eps = (z - alpha x0) / sigma (diffusion_utils.py:205-212)."""
  logsnr = broadcast_to_shape_from_left(xp, logsnr, z.shape)
  return xp.sqrt(1.0 + xp.exp(logsnr)) * (z - x0 * (1.0 / xp.sqrt(1.0 + xp.exp(-logsnr))))


def predict_x0_from_eps(xp, *, z, eps, logsnr):
  """x0 = (z - sigma eps) / alpha (diffusion_utils.py:215-222)."""
  logsnr = broadcast_to_shape_from_left(xp, logsnr, z.shape)
  return xp.sqrt(1.0 + xp.exp(-logsnr)) * (z - eps * (1.0 / xp.sqrt(1.0 + xp.exp(logsnr))))


def predict_x0_from_v(xp, *, z, v, logsnr):
  """x0 = alpha z - sigma v (diffusion_utils.py:225-233)."""
  logsnr = broadcast_to_shape_from_left(xp, logsnr, z.shape)
  return xp.sqrt(xp.sigmoid(logsnr)) * z - xp.sqrt(xp.sigmoid(-logsnr)) * v


def get_x0_and_eps_from_model_output(xp, z, time, model_output, diffusion_config):
  """diffusion_utils.py:288-322."""
  logsnr = get_logsnr_t(xp, time, diffusion_config.train_schedule)
  mo = diffusion_config.model_output
  if mo == 'eps':
    return {'eps': model_output,
            'x0': predict_x0_from_eps(xp, z=z, eps=model_output, logsnr=logsnr)}
  if mo == 'x0':
    return {'eps': predict_eps_from_x0(xp, z=z, x0=model_output, logsnr=logsnr),
            'x0': model_output}
  if mo == 'v':
    x0_out = predict_x0_from_v(xp, z=z, v=model_output, logsnr=logsnr)
    return {'x0': x0_out,
            'eps': predict_eps_from_x0(xp, z=z, x0=x0_out, logsnr=logsnr)}
  if mo == 'x0_and_eps':
    n = model_output.shape[-1] // 2
    x0_, eps_ = model_output[..., :n], model_output[..., n:]
    x0 = predict_x0_from_eps(xp, z=z, eps=eps_, logsnr=logsnr)
    wx = broadcast_to_shape_from_left(xp, xp.sigmoid(-logsnr), z.shape)
    x0_out = wx * x0_ + (1. - wx) * x0
    return {'x0': x0_out,
            'eps': predict_eps_from_x0(xp, z=z, x0=x0_out, logsnr=logsnr)}
  raise ValueError('Unknown model_output: %s' % mo)


def ddim_step(xp, i, logsnr_s, logsnr_t, pred_x_t, pred_eps_t):
  """diffusion_utils.py:369-379."""
  del logsnr_t
  logsnr_s = broadcast_to_shape_from_left(xp, logsnr_s, pred_x_t.shape)
  stdv_s = xp.sqrt(xp.sigmoid(-logsnr_s))
  alpha_s = xp.sqrt(xp.sigmoid(logsnr_s))
  z_s_pred = alpha_s * pred_x_t + stdv_s * pred_eps_t
  return pred_x_t if i == 0 else z_s_pred


def ddpm_step(xp, i, eps, logsnr_s, logsnr_t, pred_x0, z_t, logvar_type):
  """diffusion_utils.py:382-395; ``eps`` = the draw normal(fold_in(rng, i))."""
  logsnr_s = broadcast_to_shape_from_left(xp, logsnr_s, pred_x0.shape)
  logsnr_t = broadcast_to_shape_from_left(xp, logsnr_t, pred_x0.shape)
  z_s_dist = diffusion_reverse(xp, x0=pred_x0, z_t=z_t, logsnr_s=logsnr_s,
                               logsnr_t=logsnr_t, logvar_type=logvar_type)
  return pred_x0 if i == 0 else z_s_dist['mean'] + z_s_dist['std'] * eps


def eval_step(xp, noise, diffusion_config, batch_size, pred_fn: Callable):
  """diffusion_utils.py:398-453.  ``noise[i]`` replaces normal(fold_in(rng, i))."""
  schedule = diffusion_config.sampler.schedule
  num_steps = schedule.num_steps

  def body(z_t, i):
    # float32 scalars as in the reference (i.astype(float32) / num_steps).
    t = xp.full((batch_size,), 0.0) + (float(i) + 1.0)
    t = t / float(num_steps)
    s = (xp.full((batch_size,), 0.0) + float(i)) / float(num_steps)
    logsnr_t = get_logsnr_t(xp, t, schedule)
    logsnr_s = get_logsnr_t(xp, s, schedule)
    time = t

    model_output = pred_fn(z=z_t, time=time, include_conditioning=True)
    outputs = get_x0_and_eps_from_model_output(xp, z_t, time, model_output,
                                               diffusion_config)
    pred_eps, pred_x0 = outputs['eps'], outputs['x0']

    cond_wt = diffusion_config.classifier_free_guidance.eval_condition_weight
    if cond_wt != 1:
      uncond_wt = 1. - cond_wt
      uncond_model_output = pred_fn(z=z_t, time=time, include_conditioning=False)
      uncond_outputs = get_x0_and_eps_from_model_output(
          xp, z_t, time, uncond_model_output, diffusion_config)
      pred_eps = cond_wt * pred_eps + uncond_wt * uncond_outputs['eps']
      pred_x0 = predict_x0_from_eps(xp, z=z_t, eps=pred_eps, logsnr=logsnr_t)

    if diffusion_config.sampler.clip_x0:
      pred_x0 = xp.clip(pred_x0, -1.0, 1.0)
      pred_eps = predict_eps_from_x0(xp, z=z_t, x0=pred_x0, logsnr=logsnr_t)

    if diffusion_config.sampler.name == 'ddim':
      return ddim_step(xp, i, logsnr_s, logsnr_t, pred_x0, pred_eps)
    elif diffusion_config.sampler.name == 'ddpm':
      eps = None if noise is None else noise[i]
      return ddpm_step(xp, i, eps, logsnr_s, logsnr_t, pred_x0, z_t,
                       diffusion_config.sampler.logvar_type)
    raise ValueError('Unknown sampler type: %s' % diffusion_config.sampler.name)

  return body


def eval_scan(xp, init_z, noise, pred_fn, diffusion_config, trace=None):
  """diffusion_utils.py:456-476: reversed scan i = N-1 .. 0 from ``init_z``.

  ``trace`` (optional list) receives z after every step for debugging/parity.
  """
  batch_size = init_z.shape[0]
  step_fn = eval_step(xp, noise, diffusion_config, batch_size, pred_fn)
  z = init_z
  for i in reversed(range(diffusion_config.sampler.schedule.num_steps)):
    z = step_fn(z, i)
    if trace is not None:
      trace.append(z)
  return z
