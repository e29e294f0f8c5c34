"""Oracle restatement of the hot-path ops in the reference ``layers.py``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows (paths relative to
/root/reference/music_spectrogram_diffusion/).  ``xp`` is an oracle backend
(oracle/backend.py); parameters are plain arrays in a flat dict.

Pinned by the reference's own tests: ``mha``/``dot_product_attention``
(layers_test.py:285-330, 375-387), ``make_attention_mask``
(layers_test.py:117-125), ``dense_general`` (layers_test.py:450-484).
Unpinned by any reference test (standard definitions followed): tanh-GELU
(flax.linen.gelu default approximate=True), swish, RMS LayerNorm, FiLM.
"""
from __future__ import annotations

import math

import numpy as np

MASK_BIAS = -1e10  # layers.py:346


def dense_general(xp, x, kernel, n_contract_axes=1):
  """Bias-free DenseGeneral (layers.py:397-442).

  ``kernel`` is stored 2-D ``[prod(in_axes), prod(features)]`` (layers.py:430-431);
  contraction is over the last ``n_contract_axes`` axes of ``x``.
  """
  lead = tuple(x.shape[:x.ndim - n_contract_axes])
  k_in = int(np.prod(x.shape[x.ndim - n_contract_axes:]))
  assert kernel.shape[0] == k_in, (kernel.shape, x.shape)
  y = xp.matmul(xp.reshape(x, lead + (k_in,)), kernel)
  return y


def gelu_tanh(xp, x):
  """flax.linen.gelu with the JAX default ``approximate=True`` (layers.py:445-456)."""
  c = math.sqrt(2.0 / math.pi)
  return 0.5 * x * (1.0 + xp.tanh(c * (x + 0.044715 * (x * x * x))))


def swish(xp, x):
  """flax.linen.swish = x * sigmoid(x) (network.py:385,391)."""
  return x * xp.sigmoid(x)


def activation(xp, name, x):
  """layers.py:445-456 (_convert_to_activation_function)."""
  if name == 'linear':
    return x
  if name == 'gelu':
    return gelu_tanh(xp, x)
  if name == 'relu':
    return xp.maximum(x, 0.0 * x)
  if name in ('swish', 'silu'):
    return swish(xp, x)
  raise ValueError("don't know how to convert %s to an activation function" % (name,))


def rms_layer_norm(xp, x, scale, epsilon=1e-6):
  """T5 LayerNorm = RMSNorm, no mean subtraction, no bias (layers.py:632-649)."""
  mean2 = xp.mean(xp.square(x), axis=-1, keepdims=True)
  y = x * (1.0 / xp.sqrt(mean2 + epsilon))  # lax.rsqrt
  return y * scale


def film(xp, x, conditioning_emb, kernel):
  """FiLMLayer (layers.py:652-666): x * (scale + 1) + bias."""
  scale_bias = dense_general(xp, conditioning_emb, kernel)
  d = x.shape[-1]
  scale, bias = scale_bias[..., :d], scale_bias[..., d:]
  return x * (scale + 1.0) + bias


def make_attention_mask(xp, query_input, key_input):
  """Multiplicative padding mask [batch, 1, len_q, len_kv] (layers.py:672-704)."""
  mask = xp.expand_dims(query_input, -1) * xp.expand_dims(key_input, -2)
  return xp.expand_dims(mask, -3)


def softmax_last(xp, x):
  """jax.nn.softmax over the last axis (layers.py:165)."""
  m = xp.max(x, axis=-1, keepdims=True)
  e = xp.exp(x - m)
  return e / xp.sum(e, axis=-1, keepdims=True)


def dot_product_attention(xp, query, key, value, bias=None):
  """layers.py:109-181, deterministic path (no dropout), no 1/sqrt(d) scaling.

  query [b,q,h,d], key/value [b,k,h,d], bias broadcastable to [b,h,q,k].
  """
  attn_weights = xp.einsum('bqhd,bkhd->bhqk', query, key)
  if bias is not None:
    attn_weights = attn_weights + bias
  attn_weights = softmax_last(xp, attn_weights)
  return xp.einsum('bhqk,bkhd->bqhd', attn_weights, value)


def mha(xp, params, prefix, inputs_q, inputs_kv, num_heads, head_dim, mask=None,
        bias=None):
  """MultiHeadDotProductAttention, non-decode path (layers.py:188-268,340-379).

  Kernels: query/key/value ``[features, heads*head_dim]`` (reshaped
  ``[f, h, d]``), out ``[heads*head_dim, features]`` (layers_test.py:306-319).
  The mask becomes an additive 0 / -1e10 bias (layers.py:341-346).
  """
  b, lq = inputs_q.shape[0], inputs_q.shape[1]
  lk = inputs_kv.shape[1]
  q = xp.reshape(dense_general(xp, inputs_q, params[prefix + '/query/kernel']),
                 (b, lq, num_heads, head_dim))
  k = xp.reshape(dense_general(xp, inputs_kv, params[prefix + '/key/kernel']),
                 (b, lk, num_heads, head_dim))
  v = xp.reshape(dense_general(xp, inputs_kv, params[prefix + '/value/kernel']),
                 (b, lk, num_heads, head_dim))
  attention_bias = None
  if mask is not None:
    attention_bias = xp.where(mask > 0, 0.0 * mask, 0.0 * mask + MASK_BIAS)
  if bias is not None:  # combine_biases (layers.py:761-778)
    attention_bias = bias if attention_bias is None else attention_bias + bias
  x = dot_product_attention(xp, q, k, v, bias=attention_bias)
  return dense_general(xp, x, params[prefix + '/out/kernel'], n_contract_axes=2)


def mlp_block(xp, params, prefix, inputs, activations):
  """MlpBlock (layers.py:459-510): wo( prod_i act_i(x . wi_i) )."""
  acts = []
  for idx, act in enumerate(activations):
    name = 'wi' if len(activations) == 1 else 'wi_%d' % idx
    h = dense_general(xp, inputs, params['%s/%s/kernel' % (prefix, name)])
    acts.append(activation(xp, act, h))
  x = acts[0]
  for a in acts[1:]:
    x = x * a
  return dense_general(xp, x, params[prefix + '/wo/kernel'])


def embed_one_hot(xp, tokens, embedding):
  """Embed with one_hot=True (layers.py:556-559): a row gather in exact arithmetic.  Non-integer
  inputs are an error in the reference (layers.py:546-547; layers_test.py:392-401)."""
  kind = getattr(getattr(tokens, 'dtype', None), 'kind', None)
  if kind is None:   # torch tensors
    import torch
    kind = 'f' if isinstance(tokens, torch.Tensor) and tokens.dtype.is_floating_point else 'i'
  if kind not in 'iu':
    raise ValueError('Input type must be an integer or unsigned integer.')
  return xp.take(embedding, xp.asint(tokens))


def zero_activations_if_masked(xp, y, mask):
  """layers.py:882-902: zero the output if no key is valid for that batch row.

  mask is [batch, 1, len_q, len_kv]; is_not_empty is [batch, len_q, 1].
  """
  is_not_empty = xp.any(xp.squeeze(mask, 1) == 1, axis=-1, keepdims=True)
  return y * xp.cast(is_not_empty)


def sinusoidal_table(max_len, features, min_scale=1.0, max_scale=10000.0,
                     sin_offsets=0.0, cos_offsets=0.0, permutation=None):
  """layers.sinusoidal (layers.py:51-106) as a float32 table.

  The reference draws ``sin_offsets``/``cos_offsets`` (uniform [0, 2pi)) and the
  feature permutation from a JAX key at init time and stores the table as a
  (frozen) checkpoint parameter; here the draws are explicit inputs.
  """
  position = np.arange(0, max_len)[:, np.newaxis]
  scale_factor = -np.log(max_scale / min_scale) / (features // 2 - 1)
  div_term = min_scale * np.exp(np.arange(0, features // 2) * scale_factor)
  rads = (position * div_term).astype(np.float32)  # jnp.array(...) -> float32
  pe = np.zeros((max_len, features), np.float32)
  pe[:, :features // 2] = np.sin(rads + np.float32(0) + np.asarray(sin_offsets, np.float32))
  pe[:, features // 2:2 * (features // 2)] = np.cos(
      rads + np.asarray(cos_offsets, np.float32))
  if permutation is not None:
    pe = pe[:, np.asarray(permutation)]
  return pe
