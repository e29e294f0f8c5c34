"""Oracle restatement of the prediction wrappers: ``models/diffusion/models.py``
``predict_batch_with_aux``, ``audio_codecs.py`` feature scaling, and the
per-song segment loop of ``beam/evaluation.py`` ``InferSong.process``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  No reference test touches
these files; ``predict_batch_with_aux`` is pinned at 1e-9 against the reference's
own method executed over a NumPy stand-in of jax / flax
(tests/golden/ref_*.npz, tests/test_ref_golden.py); the segment loop is not.
"""
from __future__ import annotations

import math

import numpy as np

from oracle import net
from oracle import ops
from oracle import sampler as diffusion_utils


class MelGANCodec:
  """Constants of audio_codecs.MelGAN (audio_codecs.py:204-218) + scaling
  (audio_codecs.py:166-183)."""
  name = 'melgan'
  n_dims = 128
  sample_rate = 16000
  hop_size = 320
  min_value = math.log(1e-5)
  max_value = 4.0

  def scale_features(self, xp, features, output_range=(-1.0, 1.0), clip=False):
    min_out, max_out = output_range
    if clip:
      features = xp.clip(features, self.min_value, self.max_value)
    zero_one = (features - self.min_value) / (self.max_value - self.min_value)
    return zero_one * (max_out - min_out) + min_out

  def scale_to_features(self, xp, outputs, input_range=(-1.0, 1.0), clip=False):
    min_out, max_out = input_range
    outputs = xp.clip(outputs, min_out, max_out) if clip else outputs
    zero_one = (outputs - min_out) / (max_out - min_out)
    return zero_one * (self.max_value - self.min_value) + self.min_value


def _mul_flag(xp, encodings_and_masks, include_conditioning):
  """models.py:181-182 / 376-377: every encoding AND mask times the flag."""
  f = 1.0 if include_conditioning else 0.0
  return [(e * f, m * f) for e, m in encodings_and_masks]


def predict_batch_with_aux(xp, cfg, diffusion_config, params, batch, init_z, noise,
                           context=None, codec=None, trace=None):
  """{Diffusion,ContextDiffusion}Model.predict_batch_with_aux
  (models.py:149-205 / 340-400), noise explicit.

  ``context`` True selects ContextDiffusionModel; default: inferred from batch.
  Returns (decodes [B,T,n] in mel units, scores zeros[B]).
  """
  codec = codec or MelGANCodec()
  if context is None:
    context = 'encoder_continuous_inputs' in batch
  params = {k: xp.asarray(v) for k, v in params.items()}
  tokens = np.asarray(batch['encoder_input_tokens'])
  if context:
    ctx = xp.asarray(batch['encoder_continuous_inputs'])
    ctx = codec.scale_features(xp, ctx, output_range=(-1., 1.), clip=True)
    encodings_and_masks = net.context_transformer_encode(
        xp, cfg, params, tokens, ctx, xp.asarray(batch['encoder_continuous_mask']))
  else:
    encodings_and_masks = net.transformer_encode(xp, cfg, params, tokens)

  def pred_fn(z, time, include_conditioning):
    step = _mul_flag(xp, encodings_and_masks, include_conditioning)
    return net.decode(xp, cfg, params, step, z, time)

  pred_x0 = diffusion_utils.eval_scan(
      xp, xp.asarray(init_z), None if noise is None else xp.asarray(noise), pred_fn,
      diffusion_config, trace=trace)
  decodes = codec.scale_to_features(xp, pred_x0, input_range=(-1., 1.))
  scores = xp.zeros((tokens.shape[0],))
  return decodes, scores


def predict_song(xp, cfg, diffusion_config, params, segment_tokens, init_zs, noises,
                 context_length=None, always_mask_context=False, codec=None):
  """Segment loop of InferSong.process (beam/evaluation.py:161-223).

  segment_tokens: list of int32 [1, L]; init_zs[k]/noises[k]: the explicit noise
  of segment k.  With a context model, segment 0 (or every segment if
  ``always_mask_context``) runs with context zeros + mask 0; later segments get
  the previous PREDICTION (mel units) with mask 1.  Returns [1, T*K, n].
  """
  codec = codec or MelGANCodec()
  n = codec.n_dims
  pred_encoded = np.zeros([1, context_length or 0, n], np.float32)
  full = []
  for i, tokens in enumerate(segment_tokens):
    batch = {'encoder_input_tokens': np.asarray(tokens)}
    if context_length is not None:
      batch['encoder_continuous_inputs'] = pred_encoded[:1]
      if i == 0 or always_mask_context:
        batch['encoder_continuous_mask'] = np.zeros([1, context_length], np.int32)
      else:
        batch['encoder_continuous_mask'] = np.ones([1, context_length], np.int32)
    decodes, _ = predict_batch_with_aux(
        xp, cfg, diffusion_config, params, batch, init_zs[i], noises[i],
        context=context_length is not None, codec=codec)
    pred_encoded = xp.to_numpy(decodes).astype(np.float32)
    full.append(pred_encoded[:1])
  return np.concatenate(full, axis=1)
