"""Seeded synthetic weights and inputs for the hot path.

No checkpoint is reachable from this environment (README.md:24-26 of the
reference: GCS zips), so -- like ``TrainStateInitializer.from_checkpoint_or_scratch``
falling back to ``init_fn`` (inference.py:163-175) -- the model can be
initialised "from scratch" with the reference's own initialisers:

  * DenseGeneral default: variance_scaling(1.0, 'fan_in', 'truncated_normal')
    (layers.py:409-410); fan_in = kernel.shape[0]
  * attention q/k/v/out: variance_scaling(1.0, 'fan_in', 'normal')
    (layers.py:206-207), query additionally / sqrt(head_dim) (layers.py:257-258)
  * token embedding: normal(stddev=1.0) (network.py:282)
  * position tables: layers.sinusoidal(permute_bands, random_phase_offsets)
    (layers.py:51-106; network.py:84-91), frozen parameters
  * RMSNorm scales: ones (layers.py:636)

The draws come from ``numpy.random.default_rng`` (not jax.random), so the values
are NOT bit-identical to a JAX init; they follow the same distributions.
Synthetic inputs follow SURVEY.md 8(d) / BASELINE.md 4.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import config as config_lib

_TRUNC_STD = 0.87962566103423978  # stddev of a unit normal truncated to [-2, 2]


def _variance_scaling(rng, shape, distribution):
  fan_in = shape[0]
  std = np.sqrt(1.0 / fan_in)
  if distribution == 'normal':
    return (rng.standard_normal(shape) * std).astype(np.float32)
  # truncated normal on [-2, 2] std units, rescaled (jax variance_scaling)
  x = rng.standard_normal(shape)
  bad = np.abs(x) > 2.0
  while bad.any():
    x[bad] = rng.standard_normal(int(bad.sum()))
    bad = np.abs(x) > 2.0
  return (x * (std / _TRUNC_STD)).astype(np.float32)


def _sinusoidal(rng, max_len, features, permute_bands, random_phase_offsets,
                min_scale=1.0, max_scale=10000.0):
  position = np.arange(0, max_len)[:, np.newaxis]
  scale_factor = -np.log(max_scale / min_scale) / (features // 2 - 1)
  div_term = min_scale * np.exp(np.arange(0, features // 2) * scale_factor)
  rads = (position * div_term).astype(np.float32)
  if random_phase_offsets:
    sin_off = rng.uniform(0, 2 * np.pi, features // 2).astype(np.float32)
    cos_off = rng.uniform(0, 2 * np.pi, features // 2).astype(np.float32)
  else:
    sin_off = cos_off = np.float32(0)
  pe = np.zeros((max_len, features), np.float32)
  pe[:, :features // 2] = np.sin(rads + sin_off)
  pe[:, features // 2:2 * (features // 2)] = np.cos(rads + cos_off)
  if permute_bands:
    pe = pe[:, rng.permutation(features)]
  return np.ascontiguousarray(pe)


def init_params(spec: config_lib.ModelSpec, seed: int = 0,
                norm_scale_jitter: float = 0.0) -> Dict[str, np.ndarray]:
  """Flat ``name -> float32 array`` parameter dict for ``spec``.

  ``norm_scale_jitter`` > 0 perturbs the RMSNorm scales (tests use it so that a
  kernel ignoring a scale vector cannot pass).
  """
  rng = np.random.default_rng(seed)
  c = spec.t5
  pos_kind = c.position_encoding
  params: Dict[str, np.ndarray] = {}
  for name, shape in config_lib.param_shapes(spec).items():
    leaf = name.rsplit('/', 2)
    if name.endswith('/scale'):
      v = np.ones(shape, np.float32)
      if norm_scale_jitter:
        v = (v + norm_scale_jitter * rng.standard_normal(shape)).astype(np.float32)
    elif name.endswith('token_embedder/embedding'):
      v = rng.standard_normal(shape).astype(np.float32)
    elif name.endswith('Embed_0/embedding'):
      if pos_kind == 'fixed':
        v = _sinusoidal(rng, shape[0], shape[1], False, False)
      elif pos_kind in ('fixed_permuted_offset', 'learnable_permuted_offset'):
        v = _sinusoidal(rng, shape[0], shape[1], True, True)
      elif pos_kind == 'random':
        v = _variance_scaling(rng, shape[::-1], 'normal').T.copy()  # out_axis=0
      else:
        raise ValueError(f'Unknown position_encoding: {pos_kind}')
    elif leaf[-2] in ('query', 'key', 'value', 'out') and leaf[-1] == 'kernel':
      v = _variance_scaling(rng, shape, 'normal')
      if leaf[-2] == 'query':
        v = (v / np.sqrt(np.float32(c.head_dim))).astype(np.float32)
    else:
      v = _variance_scaling(rng, shape, 'truncated_normal')
    params[name] = np.ascontiguousarray(v, dtype=np.float32)
  return params


def trained_like(params: Dict[str, np.ndarray], seed: int = 0, channel_sigma: float = 0.5,
                 outlier_frac: float = 0.01, outlier_gain: float = 6.0, norm_sigma: float = 0.4) -> Dict[str, np.ndarray]:
  """Reshape the dynamic range of initialiser weights towards what TRAINED transformers look like (no checkpoint
  is reachable here): every kernel's output channels get log-normal gains exp(channel_sigma N(0,1)), a fraction
  `outlier_frac` of them an extra `outlier_gain` (the few dominant feature channels of trained residual streams),
  and the RMSNorm scales become log-normal around 1.  Parity tests use it to check that the split-bf16 arithmetic
  is not tuned to the narrow range of fresh initialisers."""
  rng = np.random.default_rng(seed)
  out = {}
  for name, v in params.items():
    if name.endswith('/scale'):
      v = (v * np.exp(norm_sigma * rng.standard_normal(v.shape))).astype(np.float32)
    elif name.endswith('/kernel'):
      gain = np.exp(channel_sigma * rng.standard_normal(v.shape[1]))
      gain[rng.random(v.shape[1]) < outlier_frac] *= outlier_gain
      # keep the layer's overall output power: trained layers are not 30 % louder than fresh ones on average
      gain /= np.sqrt(np.mean(gain ** 2))
      v = (v * gain[None, :]).astype(np.float32)
    out[name] = np.ascontiguousarray(v, dtype=np.float32)
  return out


def sharp_attention(params: Dict[str, np.ndarray], gain: float, where: str = 'decoder') -> Dict[str, np.ndarray]:
  """Multiply every query kernel under `where` (self- and cross-attention of the decoder by default) by `gain`.
  The projections read RMS-normalised inputs, so every attention logit q.k scales by `gain`: fresh initialisers give
  logits of O(1) (soft attention over all keys), gain 4 / 8 put competing keys 10 - 30 apart -- the SHARP attention
  of a trained model, where a rounding of Q that is harmless at O(1) logits (|s| 2^-12 with one half plane) decides
  which key wins (DESIGN.md 3 "sharp attention")."""
  out = {}
  for name, v in params.items():
    if name.startswith(where + '/') and name.endswith('/query/kernel'):
      v = (v * np.float32(gain)).astype(np.float32)
    out[name] = np.ascontiguousarray(v, dtype=np.float32)
  return out


def segment_tokens(spec: config_lib.ModelSpec, segment: int, seed: int = 1234,
                   min_len: int = 128, max_len: int = 1536) -> np.ndarray:
  """int32 [1, inputs_length]: ``len ~ U{min..max}`` regular ids ``U{3..1390}``,
  then EOS=1, then PAD=0.  Lengths are clamped to the model's
  input length / vocabulary for the tiny presets."""
  length = spec.task_feature_lengths['inputs']
  rng = np.random.default_rng(seed + segment)
  hi_len = min(max_len, length - 1)
  lo_len = min(min_len, hi_len)
  n = int(rng.integers(lo_len, hi_len + 1))
  hi_id = min(1390, spec.t5.vocab_size - 1)
  toks = np.zeros((1, length), np.int32)
  toks[0, :n] = rng.integers(3, hi_id + 1, n)
  toks[0, n] = 1
  return toks
