"""Oracle restatement of the reference ``models/diffusion/network.py``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  No reference test covers
this file; it is a line-by-line restatement, inference path only
(deterministic=True, dropout inactive), pinned at 1e-9 against the reference's
own network.py executed over a NumPy stand-in of jax / flax
(tests/golden/ref_*.npz, tests/test_ref_golden.py).

Parameters are a flat dict ``name -> array`` using the Flax auto-naming the
reference produces, e.g.
``decoder/layers_3/FiLMLayer_0/DenseGeneral_0/kernel``.
"""
from __future__ import annotations

import dataclasses
from typing import Sequence

from oracle import ops
from oracle import sampler as diffusion_utils


@dataclasses.dataclass(frozen=True)
class T5Config:
  """network.py:54-72 (inference-relevant fields)."""
  vocab_size: int
  emb_dim: int = 512
  num_heads: int = 8
  num_encoder_layers: int = 6
  num_decoder_layers: int = 6
  head_dim: int = 64
  mlp_dim: int = 2048
  mlp_activations: Sequence[str] = ('relu',)
  max_decoder_noise_time: float = 2e4
  decoder_cross_attend_style: str = 'sum_cross_attends'
  position_encoding: str = 'fixed'
  context_positions: str = 'regular'


def get_sequence_length(xp, sequence):
  """network.py:28-39: index of the first 0, or the full length if none."""
  seq = xp.to_numpy(sequence)
  zeros = (seq == 0)
  length = int(zeros.argmax()) if zeros.any() else 0
  if length == 0 and seq[0] != 0:
    length = seq.shape[0]
  return length


def encoder_layer(xp, cfg, p, prefix, inputs, encoder_inputs_mask):
  """EncoderLayer (network.py:109-158)."""
  encoder_mask = ops.make_attention_mask(xp, encoder_inputs_mask, encoder_inputs_mask)
  x = ops.rms_layer_norm(xp, inputs, p[prefix + '/pre_attention_layer_norm/scale'])
  x = ops.mha(xp, p, prefix + '/attention', x, x, cfg.num_heads, cfg.head_dim,
              mask=encoder_mask)
  x = x + inputs
  y = ops.rms_layer_norm(xp, x, p[prefix + '/pre_mlp_layer_norm/scale'])
  y = ops.mlp_block(xp, p, prefix + '/mlp', y, cfg.mlp_activations)
  return y + x


def token_encoder(xp, cfg, p, prefix, tokens, mask):
  """TokenEncoder (network.py:261-303)."""
  seq_length = tokens.shape[1]
  x = ops.embed_one_hot(xp, tokens, p[prefix + '/token_embedder/embedding'])
  x = x + p[prefix + '/Embed_0/embedding'][:seq_length][None]
  for lyr in range(cfg.num_encoder_layers):
    x = encoder_layer(xp, cfg, p, '%s/layers_%d' % (prefix, lyr), x, mask)
  x = ops.rms_layer_norm(xp, x, p[prefix + '/encoder_norm/scale'])
  return x, mask


def continuous_encoder(xp, cfg, p, prefix, inputs, mask):
  """ContinuousEncoder (network.py:306-357)."""
  b, max_positions = inputs.shape[0], inputs.shape[1]
  x = ops.dense_general(xp, inputs, p[prefix + '/input_proj/kernel'])
  table = p[prefix + '/Embed_0/embedding']
  if cfg.context_positions == 'regular':
    pos = table[:max_positions][None]
  elif cfg.context_positions == 'terminal_relative':
    # network.py:329-334: positions rolled by each row's sequence length.
    rows = []
    for i in range(b):
      seq_len = get_sequence_length(xp, mask[i])
      idx = xp.roll(xp.asint(list(range(max_positions))), seq_len, 0)
      rows.append(xp.expand_dims(xp.take(table, idx), 0))
    pos = xp.concatenate(rows, 0)
  else:
    raise ValueError(f'Unknown context_positions: {cfg.context_positions}')
  x = x + pos
  for lyr in range(cfg.num_encoder_layers):
    x = encoder_layer(xp, cfg, p, '%s/layers_%d' % (prefix, lyr), x, mask)
  x = ops.rms_layer_norm(xp, x, p[prefix + '/encoder_norm/scale'])
  return x, mask


def decoder_layer(xp, cfg, p, prefix, inputs, encodings_and_encdec_masks,
                  conditioning_emb):
  """DecoderLayer (network.py:161-258)."""
  x = ops.rms_layer_norm(xp, inputs, p[prefix + '/pre_self_attention_layer_norm/scale'])
  if conditioning_emb is not None:
    x = ops.film(xp, x, conditioning_emb, p[prefix + '/FiLMLayer_0/DenseGeneral_0/kernel'])
  x = ops.mha(xp, p, prefix + '/self_attention', x, x, cfg.num_heads, cfg.head_dim)
  x = x + inputs

  y = ops.rms_layer_norm(xp, x, p[prefix + '/pre_cross_attention_layer_norm/scale'])
  if cfg.decoder_cross_attend_style == 'sum_cross_attends':
    ys = []
    for n, (encoded, encdec_mask) in enumerate(encodings_and_encdec_masks):
      y_n = ops.mha(xp, p, '%s/MultiHeadDotProductAttention_%d' % (prefix, n), y,
                    encoded, cfg.num_heads, cfg.head_dim, mask=encdec_mask)
      ys.append(ops.zero_activations_if_masked(xp, y_n, encdec_mask))
    acc = ys[0]
    for y_n in ys[1:]:
      acc = acc + y_n
    y = acc + x
  elif cfg.decoder_cross_attend_style == 'concat_encodings':
    encoded = xp.concatenate([e for e, _ in encodings_and_encdec_masks], 1)
    encdec_mask = xp.concatenate([m for _, m in encodings_and_encdec_masks], -1)
    y = ops.mha(xp, p, prefix + '/MultiHeadDotProductAttention_0', y, encoded,
                cfg.num_heads, cfg.head_dim, mask=encdec_mask)
    y = ops.zero_activations_if_masked(xp, y, encdec_mask)
    y = y + x
  else:
    raise ValueError('Unknown decoder_cross_attend_style: '
                     f'{cfg.decoder_cross_attend_style}')

  z = ops.rms_layer_norm(xp, y, p[prefix + '/pre_mlp_layer_norm/scale'])
  if conditioning_emb is not None:
    z = ops.film(xp, z, conditioning_emb, p[prefix + '/FiLMLayer_1/DenseGeneral_0/kernel'])
  z = ops.mlp_block(xp, p, prefix + '/mlp', z, cfg.mlp_activations)
  return z + y


def time_conditioning(xp, cfg, p, prefix, decoder_noise_time):
  """network.py:377-392: sinusoid -> Dense -> swish -> Dense -> swish, [b,1,4D]."""
  emb = diffusion_utils.get_timing_signal_1d(
      xp, decoder_noise_time * cfg.max_decoder_noise_time, cfg.emb_dim,
      max_timescale=cfg.max_decoder_noise_time)
  emb = ops.swish(xp, ops.dense_general(xp, emb, p[prefix + '/time_emb_dense0/kernel']))
  emb = ops.swish(xp, ops.dense_general(xp, emb, p[prefix + '/time_emb_dense1/kernel']))
  return xp.expand_dims(emb, 1)


def decoder(xp, cfg, p, prefix, encodings_and_masks, decoder_input_tokens,
            decoder_noise_time):
  """Decoder (network.py:360-457)."""
  batch, seq_length = decoder_input_tokens.shape[0], decoder_input_tokens.shape[1]
  assert tuple(decoder_noise_time.shape) == (batch,)
  conditioning_emb = time_conditioning(xp, cfg, p, prefix, decoder_noise_time)
  position_encodings = p[prefix + '/Embed_0/embedding'][:seq_length][None]
  decoder_mask = xp.ones((batch, seq_length))
  encdec = [(e, ops.make_attention_mask(xp, decoder_mask, xp.cast(m)))
            for e, m in encodings_and_masks]
  inputs = ops.dense_general(
      xp, decoder_input_tokens, p[prefix + '/continuous_inputs_projection/kernel'])
  y = inputs + position_encodings
  for lyr in range(cfg.num_decoder_layers):
    y = decoder_layer(xp, cfg, p, '%s/layers_%d' % (prefix, lyr), y, encdec,
                      conditioning_emb)
  y = ops.rms_layer_norm(xp, y, p[prefix + '/decoder_norm/scale'])
  return ops.dense_general(xp, y, p[prefix + '/spec_out_dense/kernel'])


# ---------------------------------------------------------------------------
# Transformer / ContinuousContextTransformer (network.py:460-606)
# ---------------------------------------------------------------------------
def transformer_encode(xp, cfg, p, encoder_input_tokens):
  """Transformer.encode (network.py:470-482)."""
  mask = xp.cast(xp.to_numpy(encoder_input_tokens) > 0)
  encoded, mask = token_encoder(xp, cfg, p, 'encoder', encoder_input_tokens, mask)
  return [(encoded, mask)]


def context_transformer_encode(xp, cfg, p, input_tokens, continuous_inputs,
                               continuous_mask):
  """ContinuousContextTransformer.encode (network.py:537-559)."""
  tokens_mask = xp.cast(xp.to_numpy(input_tokens) > 0)
  tokens_encoded, tokens_mask = token_encoder(
      xp, cfg, p, 'token_encoder', input_tokens, tokens_mask)
  continuous_mask = xp.cast(continuous_mask)
  continuous_encoded, continuous_mask = continuous_encoder(
      xp, cfg, p, 'continuous_encoder', continuous_inputs, continuous_mask)
  return [(tokens_encoded, tokens_mask), (continuous_encoded, continuous_mask)]


def decode(xp, cfg, p, encodings_and_masks, input_tokens, noise_time):
  """{Transformer,ContinuousContextTransformer}.decode (network.py:484-496,561-573)."""
  return decoder(xp, cfg, p, 'decoder', encodings_and_masks, input_tokens, noise_time)
