"""Pin the oracle against the reference's OWN known-answer tests for this path
(layers_test.py of the reference, restated in NumPy -- SURVEY.md 8(c)):
  * test_multihead_dot_product_attention   layers_test.py:285-330
  * test_dot_product_attention             layers_test.py:375-387
  * test_make_attention_mask_multiply_pairwise_fn  layers_test.py:117-125
  * DenseTest (ones kernels)               layers_test.py:450-484
"""
import numpy as np
import pytest

from oracle import backend, ops

BACKENDS = [backend.NumpyBackend('float64'), backend.NumpyBackend('float32'),
            backend.TorchBackend('float64')]


def _softmax(x):
  e = np.exp(x - x.max(-1, keepdims=True))
  return e / e.sum(-1, keepdims=True)


@pytest.mark.parametrize('f', [20, 22])
@pytest.mark.parametrize('xp', BACKENDS, ids=lambda b: '%s-%s' % (b.name, b.dtype))
def test_multihead_dot_product_attention(f, xp):
  # same seed, same draw order, same shapes as layers_test.py:288-301
  b, q, h, d, k = 2, 3, 4, 5, 6
  np.random.seed(0)
  inputs_q = np.random.randn(b, q, f)
  inputs_kv = np.random.randn(b, k, f)
  query_kernel = np.random.randn(f, h, d)
  key_kernel = np.random.randn(f, h, d)
  value_kernel = np.random.randn(f, h, d)
  out_kernel = np.random.randn(h, d, f)
  params = {'a/query/kernel': query_kernel.reshape(f, -1), 'a/key/kernel': key_kernel.reshape(f, -1),
            'a/value/kernel': value_kernel.reshape(f, -1), 'a/out/kernel': out_kernel.reshape(-1, f)}
  params = {n: xp.asarray(v) for n, v in params.items()}
  y = xp.to_numpy(ops.mha(xp, params, 'a', xp.asarray(inputs_q), xp.asarray(inputs_kv), h, d))
  # expected, as layers_test.py:322-329
  query = np.einsum('bqf,fhd->bqhd', inputs_q, query_kernel)
  key = np.einsum('bkf,fhd->bkhd', inputs_kv, key_kernel)
  value = np.einsum('bkf,fhd->bkhd', inputs_kv, value_kernel)
  logits = np.einsum('bqhd,bkhd->bhqk', query, key)
  weights = _softmax(logits)
  combined_value = np.einsum('bhqk,bkhd->bqhd', weights, value)
  y_expected = np.einsum('bqhd,hdf->bqf', combined_value, out_kernel)
  tol = 1e-5 if str(xp.dtype).endswith('64') else 2e-3  # reference: rtol=atol=1e-5 (float64 inputs)
  np.testing.assert_allclose(y, y_expected, rtol=tol, atol=tol)


@pytest.mark.parametrize('xp', BACKENDS, ids=lambda b: '%s-%s' % (b.name, b.dtype))
def test_dot_product_attention_with_bias(xp):
  b, q, h, d, k = 2, 3, 4, 5, 6
  np.random.seed(0)
  query = np.random.randn(b, q, h, d)
  key = np.random.randn(b, k, h, d)
  value = np.random.randn(b, k, h, d)
  bias = np.random.randn(b, h, q, k)
  out = xp.to_numpy(ops.dot_product_attention(xp, xp.asarray(query), xp.asarray(key),
                                              xp.asarray(value), bias=xp.asarray(bias)))
  logits = np.einsum('bqhd,bkhd->bhqk', query, key)
  expected = np.einsum('bhqk,bkhd->bqhd', _softmax(logits + bias), value)
  atol = 1e-6 if str(xp.dtype).endswith('64') else 1e-4
  np.testing.assert_allclose(out, expected, atol=atol)


def test_make_attention_mask_multiply_pairwise_fn():
  xp = backend.NumpyBackend('float32')
  tokens = np.array([[7, 0, 0], [8, 5, 0]])
  m = ops.make_attention_mask(xp, xp.cast(tokens > 0), xp.cast(tokens > 0))
  assert m.shape == (2, 1, 3, 3)
  np.testing.assert_array_equal(m[0, 0], np.array([[1, 0, 0], [0, 0, 0], [0, 0, 0]]))
  np.testing.assert_array_equal(m[1, 0], np.array([[1, 1, 0], [1, 1, 0], [0, 0, 0]]))


def test_dense_general_ones_kernels():
  xp = backend.NumpyBackend('float32')
  # no bias, 3 -> 4 (layers_test.py:452-461)
  y = ops.dense_general(xp, np.ones((1, 3), np.float32), np.ones((3, 4), np.float32))
  np.testing.assert_allclose(y, np.full((1, 4), 3.))
  # two output features (2, 2): kernel stored [3, 4] (layers_test.py:463-472)
  y = ops.dense_general(xp, np.ones((1, 3), np.float32), np.ones((3, 4), np.float32)).reshape(1, 2, 2)
  np.testing.assert_allclose(y, np.full((1, 2, 2), 3.))
  # two contraction axes (layers_test.py:474-484)
  y = ops.dense_general(xp, np.ones((1, 2, 2), np.float32), np.ones((4, 3), np.float32), 2)
  np.testing.assert_allclose(y, np.full((1, 3), 4.))


def test_mask_bias_is_exact_zero_weight():
  """layers.py:341-346: a masked key gets -1e10 and therefore weight exactly 0
  in float32 -- the fact that justifies dropping padded keys (shortcut S3)."""
  xp = backend.NumpyBackend('float32')
  rng = np.random.default_rng(0)
  q, k, v = (rng.standard_normal((1, 4, 2, 8)).astype(np.float32) for _ in range(3))
  k2 = rng.standard_normal((1, 7, 2, 8)).astype(np.float32)
  v2 = rng.standard_normal((1, 7, 2, 8)).astype(np.float32)
  k2[:, :4], v2[:, :4] = k, v
  mask = np.zeros((1, 1, 4, 7), np.float32)
  mask[..., :4] = 1
  bias = np.where(mask > 0, 0.0, ops.MASK_BIAS).astype(np.float32)
  full = ops.dot_product_attention(xp, q, k2, v2, bias=bias)
  dropped = ops.dot_product_attention(xp, q, k, v)
  np.testing.assert_array_equal(full, dropped)


def test_embedder_raises_for_non_integer_input():
  """layers_test.py:392-401 ('Input type must be an integer or unsigned integer.')."""
  from oracle import backend, ops
  xp = backend.NumpyBackend('float64')
  table = np.arange(50, dtype=np.float64).reshape(10, 5)
  ids = np.arange(5, dtype=np.int64)[:, None]
  np.testing.assert_array_equal(xp.to_numpy(ops.embed_one_hot(xp, ids, xp.asarray(table))), table[ids])
  with pytest.raises(ValueError, match='Input type must be an integer or unsigned integer.'):
    ops.embed_one_hot(xp, ids.astype(np.float32), xp.asarray(table))
