"""-m gpu: the HIP building blocks through the C-ABI against NumPy/oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
  import torch
  import msd_amd
  assert torch.cuda.is_available(), 'these tests need the MI355X'
  msd_amd.native.load()
  return torch, msd_amd.native


def _dev(torch, a):
  return torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()


@pytest.mark.parametrize('prec,tol', [('f16', 3e-3), ('f16x3', 1e-5), ('bf16', 2e-2), ('bf16x3', 2e-5)])
@pytest.mark.parametrize('m,n,k', [(64, 64, 64), (256, 768, 768), (512, 128, 2048), (192, 320, 128)])
def test_gemm_h16(env, prec, tol, m, n, k):
  torch, native = env
  rng = np.random.default_rng(m + n + k)
  a = rng.standard_normal((m, k)).astype(np.float32)   # asymmetric, transpose-detecting
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  c = torch.zeros((m, n), dtype=torch.float32, device='cuda')
  native.op_gemm_h16(prec, _dev(torch, a), _dev(torch, w), c)
  ref = a.astype(np.float64) @ w.astype(np.float64)
  err = np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max()
  assert err < tol, err


def test_gemm_h16_identity_detects_transposes(env):
  torch, native = env
  k = 128
  a = np.eye(k, dtype=np.float32)[:64] * 1.0
  w = (np.arange(k * 192).reshape(k, 192) % 61).astype(np.float32)  # 512 w is exactly representable in one half plane
  c = torch.zeros((64, 192), dtype=torch.float32, device='cuda')
  native.op_gemm_h16('f16', _dev(torch, a), _dev(torch, w), c)
  np.testing.assert_array_equal(c.cpu().numpy(), w[:64])


@pytest.mark.parametrize('wscale,ascale,tol', [(1e-4, 1.0, 5e-4), (30.0, 1.0, 1e-5), (1.0, 300.0, 1e-5), (1.0, 1e-2, 5e-5)])
def test_gemm_operand_magnitudes(env, wscale, ascale, tol):
  """Half planes have 5 exponent bits: weights are packed times 2^9 (|w| < 128 representable) and the lo plane of
  small operands becomes subnormal.  Weights 1e-4 x the initialiser scale keep ~18 significant bits (hi normal,
  lo subnormal; the bound also covers an MFMA that flushed subnormal inputs: 2^-12); weights x30, activations
  x300 and x0.01 stay float32-class."""
  torch, native = env
  rng = np.random.default_rng(17)
  m, n, k = 128, 256, 768
  a = (rng.standard_normal((m, k)) * ascale).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k) * wscale).astype(np.float32)
  c = torch.zeros((m, n), dtype=torch.float32, device='cuda')
  native.op_gemm_h16('f16x3', _dev(torch, a), _dev(torch, w), c)
  ref = a.astype(np.float64) @ w.astype(np.float64)
  err = np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max()
  print('gemm f16x3, weights x%g, activations x%g: max rel err %.2e' % (wscale, ascale, err))
  assert err < tol, err


def test_out_of_range_operands_fail_loudly_on_half_planes(env):
  """Half planes hold |x| <= 65504 and |w| < 128.  Round 2 clamped both silently in the stand-alone ops (ADVICE r02:
  a saturated GEGLU came back as 30 * 255.875 with MSD_OK); now an activation beyond the range is MSD_ERR_RANGE
  (RangeError), a weight beyond it MSD_ERR_UNSUPPORTED (NotImplementedError, as msd_finalize_weights answers), and
  the bfloat16-plane build takes both in its stride."""
  torch, native = env
  rng = np.random.default_rng(3)
  m, n, k = 128, 128, 256
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  c = torch.zeros((m, n), dtype=torch.float32, device='cuda')
  a_big = a.copy()
  a_big[5, 7] = 2e5
  with pytest.raises(native.RangeError):
    native.op_gemm_h16('f16x3', _dev(torch, a_big), _dev(torch, w), c)
  with pytest.raises(native.RangeError):
    native.op_gemm_h16('f16', _dev(torch, a_big), _dev(torch, w), c)
  a_edge = a.copy()
  a_edge[5, 7] = 65504.0                      # the largest half: still in range
  native.op_gemm_h16('f16x3', _dev(torch, a_edge), _dev(torch, w), c)
  ref = a_edge.astype(np.float64) @ w.astype(np.float64)
  assert np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-5
  w_big = w.copy()
  w_big[3, 3] = 200.0
  with pytest.raises(NotImplementedError):
    native.op_gemm_h16('f16x3', _dev(torch, a), _dev(torch, w_big), c)
  for x, y in ((a_big, w), (a, w_big)):       # bfloat16 planes: float32's exponent range
    native.op_gemm_h16('bf16x3', _dev(torch, x), _dev(torch, y), c)
    ref = x.astype(np.float64) @ y.astype(np.float64)
    assert np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-5
  # attention: a query beyond the range
  q = rng.standard_normal((64, 64)).astype(np.float32)
  kk = rng.standard_normal((64, 64)).astype(np.float32)
  o = torch.zeros((64, 64), dtype=torch.float32, device='cuda')
  q[0, 0] = 1e5
  with pytest.raises(native.RangeError):
    native.op_attention('f16x3', _dev(torch, q), _dev(torch, kk), _dev(torch, kk), o, 1)
  assert native.op_gemm_bf16 is native.op_gemm_h16      # ABI <= 2 name


@pytest.mark.parametrize('m,n,k', [(64, 64, 16), (1000, 128, 768), (256, 768, 128), (7, 64, 32)])
def test_gemm_f32(env, m, n, k):
  torch, native = env
  rng = np.random.default_rng(m * n + k)
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = rng.standard_normal((k, n)).astype(np.float32)
  c = torch.zeros((m, n), dtype=torch.float32, device='cuda')
  native.op_gemm_f32(_dev(torch, a), _dev(torch, w), c)
  ref = a.astype(np.float64) @ w.astype(np.float64)
  np.testing.assert_allclose(c.cpu().numpy(), ref, rtol=0, atol=2e-6 * np.sqrt(k) * np.abs(ref).max())


@pytest.mark.parametrize('prec,tol', [('f16', 5e-3), ('f16x3', 2e-5), ('bf16', 3e-2), ('bf16x3', 3e-5)])
@pytest.mark.parametrize('nq,nk,valid,heads', [(64, 32, 32, 1), (256, 256, 256, 2), (64, 2304, 2304, 3),
                                               (128, 512, 301, 2), (64, 64, 1, 1), (64, 64, 0, 2),
                                               (64, 160, 129, 1), (192, 1344, 1337, 2)])
def test_attention(env, prec, tol, nq, nk, valid, heads):
  """Unscaled softmax(q k^T) v with a key-count bound (== the reference's -1e10
  padding bias, layers.py:341-346) and the all-masked -> 0 rule (layers.py:882-902)."""
  torch, native = env
  from oracle import backend, ops
  rng = np.random.default_rng(nq + nk + valid)
  j = heads * 64
  q = (rng.standard_normal((nq, j)) * 0.35).astype(np.float32)
  k = (rng.standard_normal((nk, j)) * 0.35).astype(np.float32)
  v = rng.standard_normal((nk, j)).astype(np.float32)
  o = torch.zeros((nq, j), dtype=torch.float32, device='cuda')
  native.op_attention(prec, _dev(torch, q), _dev(torch, k), _dev(torch, v), o, heads, n_keys_valid=valid)
  got = o.cpu().numpy()
  if valid == 0:
    np.testing.assert_array_equal(got, 0.0)
    return
  xp = backend.NumpyBackend('float64')
  sh = lambda x, n: x.reshape(1, n, heads, 64).astype(np.float64)
  ref = ops.dot_product_attention(xp, sh(q, nq), sh(k[:valid], valid), sh(v[:valid], valid))
  ref = ref.reshape(nq, j)
  assert np.abs(got - ref).max() < tol * max(1.0, np.abs(ref).max())
  if prec == 'f16x3':
    # the decoder's form of this mode: Q and the softmax weights as ONE half plane each (11 significand bits), K / V
    # as hi + lo.  Bounds: the logits move by |s| 2^-12 ~ 1e-3 at |s| ~ 4, so the weights by ~1e-3 relative, and
    # rounding the weights themselves adds 2^-12 -- a few 1e-4 of the output range (single-plane 'f16' is 5e-3)
    for qp, bound in ((1, 6e-4), (2, 3e-4), (3, 8e-4)):
      o.zero_()
      native.op_attention(prec, _dev(torch, q), _dev(torch, k), _dev(torch, v), o, heads, n_keys_valid=valid, qp=qp)
      err = np.abs(o.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
      assert err < bound, (qp, err)


@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
@pytest.mark.parametrize('nq,nk,valid,heads,ksplit', [(256, 2304, 2304, 12, 4), (256, 1408, 1356, 12, 4), (256, 768, 700, 12, 2),
                                                      (64, 512, 301, 2, 2), (128, 1024, 1000, 3, 8), (64, 512, 130, 1, 4)])
def test_attention_key_split_merged_inside_the_launch(env, prec, nq, nk, valid, heads, ksplit):
  """Round 6: a key-split attention finishes INSIDE its launch -- every block publishes its partial write-through (sc1),
  takes a ticket, the last block of a (query tile, head) group merges the partials (attention.h
  attention_inlaunch_merge; MI355X guide G16 R1).  Against the separate merge launch the result must be BIT-identical
  (one merge function), on the first launch and on the 25th back-to-back launch (the arrival counters reset
  themselves; a stale partial from another XCD's L2 or a counter off by one would show here), at the decoder's own
  shape (12 heads x 4 query tiles x 4 splits = 192 blocks) and on ragged / short key axes where some splits are empty;
  and it must sit on the float64 oracle like the plain kernel."""
  torch, native = env
  from oracle import backend, ops
  rng = np.random.default_rng(nq + nk + valid + ksplit)
  j = heads * 64
  q = (rng.standard_normal((nq, j)) * 0.35).astype(np.float32)
  k = (rng.standard_normal((nk, j)) * 0.35).astype(np.float32)
  v = rng.standard_normal((nk, j)).astype(np.float32)
  qd, kd, vd = _dev(torch, q), _dev(torch, k), _dev(torch, v)
  outs = {}
  for name, inl, reps in (('merge launch', False, 1), ('in launch', True, 1), ('in launch x25', True, 25), ('merge launch x3', False, 3)):
    o = torch.full((nq, j), float('nan'), dtype=torch.float32, device='cuda')
    native.op_attention_split(prec, qd, kd, vd, o, heads, ksplit, inl, repeats=reps, n_keys_valid=valid)
    outs[name] = o.cpu().numpy()
  for name, got in outs.items():
    assert np.isfinite(got).all(), name
    assert np.array_equal(got, outs['merge launch']), (name, np.abs(got - outs['merge launch']).max())
  xp = backend.NumpyBackend('float64')
  sh = lambda x, n: x.reshape(1, n, heads, 64).astype(np.float64)
  ref = ops.dot_product_attention(xp, sh(q, nq), sh(k[:valid], valid), sh(v[:valid], valid)).reshape(nq, j)
  tol = 2e-5 if prec == 'f16x3' else 5e-3
  assert np.abs(outs['in launch'] - ref).max() < tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('s_std', [3.5, 10.0, 20.0])
def test_attention_sharp_logits_by_query_side_planes(env, s_std):
  """SHARP attention: logits with standard deviation `s_std` -- the top competing keys of a
  row sit ~2.8 s_std from the mean, i.e. |s| ~ 10 / 30 / 60 -- against the float64 oracle, for each query-side plane
  choice of the two-plane mode.  What one half plane (11 significand bits) costs where:
    P (softmax weights, qp bit 1): 2^-12 relative on every weight, whatever the logits  -> flat, ~2e-4 of the output
    Q (queries, qp bit 0):         a logit moves by ~|s| 2^-12                           -> GROWS with s_std
  The bounds below are those two laws; a NumPy emulation of the planes (hi = f16(x), lo = f16(x - hi)) gives, for
  s_std 3.5 / 10 / 20: all planes 4e-7 / 9e-7 / 1.2e-6, P one plane 1.0e-4 / 1.0e-4 / 1.2e-4, Q one plane 8e-4 / 2.3e-3 / 3.3e-3."""
  torch, native = env
  from oracle import backend, ops
  rng = np.random.default_rng(int(s_std * 10))
  nq, nk, heads = 128, 512, 2
  j = heads * 64
  scale = np.sqrt(s_std / 8.0)          # q, k ~ N(0, scale^2): q.k over 64 dims has std 8 scale^2
  q = (rng.standard_normal((nq, j)) * scale).astype(np.float32)
  k = (rng.standard_normal((nk, j)) * scale).astype(np.float32)
  v = rng.standard_normal((nk, j)).astype(np.float32)
  xp = backend.NumpyBackend('float64')
  sh = lambda x, n: x.reshape(1, n, heads, 64).astype(np.float64)
  ref = ops.dot_product_attention(xp, sh(q, nq), sh(k, nk), sh(v, nk)).reshape(nq, j)
  logits = np.einsum('qhd,khd->hqk', q.reshape(nq, heads, 64).astype(np.float64), k.reshape(nk, heads, 64).astype(np.float64))
  top = np.sort(logits, -1)[..., -1].mean()
  errs = {}
  for qp in (0, 2, 1, 3):
    o = torch.zeros((nq, j), dtype=torch.float32, device='cuda')
    native.op_attention('f16x3', _dev(torch, q), _dev(torch, k), _dev(torch, v), o, heads, qp=qp)
    errs[qp] = np.abs(o.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
  print('sharp attention, logit std %.1f (mean top logit %.1f): max err all planes %.2e | P one plane %.2e | Q one plane %.2e | both %.2e'
        % (s_std, top, errs[0], errs[2], errs[1], errs[3]))
  q_law = 2.0 * s_std * 2.0 ** -12 + 5e-4
  assert errs[0] < 2e-5 and errs[2] < 4e-4
  assert errs[1] < q_law and errs[3] < q_law + 4e-4
  assert errs[2] < errs[1]   # the P plane is the cheap one to drop, at every sharpness


def test_attention_two_equal_keys_at_logit_40(env):
  """Two keys with EQUAL logits ~40 per query (k_b = k_a reflected about q: q.k_b == q.k_a), far above a soft
  background: the output is the 50 / 50 mix of their values, and any perturbation of the logit DIFFERENCE moves the
  mix.  All planes keep it to ~1e-5; one plane for Q moves the two logits by different amounts (~|q| |k_a - k_b| 2^-12)."""
  torch, native = env
  from oracle import backend, ops
  rng = np.random.default_rng(40)
  nq, nbg = 64, 128
  q = rng.standard_normal((nq, 64)).astype(np.float32)
  ka = (q * (40.0 / (q * q).sum(-1, keepdims=True)) + 0.4 * rng.standard_normal((nq, 64))).astype(np.float32)
  kb = (2.0 * (q * ka).sum(-1, keepdims=True) / (q * q).sum(-1, keepdims=True) * q - ka).astype(np.float32)
  k = np.concatenate([np.stack([ka, kb], 1).reshape(2 * nq, 64), 0.35 * rng.standard_normal((nbg, 64)).astype(np.float32)], 0)
  v = rng.standard_normal((k.shape[0], 64)).astype(np.float32)
  xp = backend.NumpyBackend('float64')
  sh = lambda x, n: x.reshape(1, n, 1, 64).astype(np.float64)
  ref = ops.dot_product_attention(xp, sh(q, nq), sh(k, k.shape[0]), sh(v, k.shape[0])).reshape(nq, 64)
  sd = (q.astype(np.float64) * (ka.astype(np.float64) - kb.astype(np.float64))).sum(-1)
  assert np.abs(sd).max() < 1e-3 and 30 < (q.astype(np.float64) * ka).sum(-1).min()   # equal logits, ~40
  errs = {}
  for qp in (0, 2, 1, 3):
    o = torch.zeros((nq, 64), dtype=torch.float32, device='cuda')
    native.op_attention('f16x3', _dev(torch, q), _dev(torch, k), _dev(torch, v), o, 1, qp=qp)
    errs[qp] = np.abs(o.cpu().numpy() - ref).max()
  print('two equal keys at logit ~40: max err all planes %.2e | P one plane %.2e | Q one plane %.2e | both %.2e'
        % (errs[0], errs[2], errs[1], errs[3]))
  assert errs[0] < 5e-5 and errs[2] < 3e-4 and errs[1] < 8e-3 and errs[3] < 8e-3   # (emulation: 1.3e-6 / 1.5e-5 / 3.7e-3 / 3.8e-3)


def test_attention_spiked_key_forces_online_rescale(env):
  """A key block whose max jumps by ~60 forces the running-max rescale path."""
  torch, native = env
  from oracle import backend, ops
  rng = np.random.default_rng(5)
  nq, nk, heads = 64, 256, 1
  q = (rng.standard_normal((nq, 64)) * 0.3).astype(np.float32)
  k = (rng.standard_normal((nk, 64)) * 0.3).astype(np.float32)
  v = rng.standard_normal((nk, 64)).astype(np.float32)
  k[200] = q[3] * 40.0  # q[3].k[200] >> every other score, in the 7th key block
  o = torch.zeros((nq, 64), dtype=torch.float32, device='cuda')
  native.op_attention('f16x3', _dev(torch, q), _dev(torch, k), _dev(torch, v), o, heads)
  xp = backend.NumpyBackend('float64')
  sh = lambda x, n: x.reshape(1, n, 1, 64).astype(np.float64)
  ref = ops.dot_product_attention(xp, sh(q, nq), sh(k, nk), sh(v, nk)).reshape(nq, 64)
  assert np.abs(o.cpu().numpy() - ref).max() < 1e-4


def test_philox_normal_matches_oracle(env):
  torch, native = env
  from oracle import philox
  for n, seed, stream, sub in [(4096, 0, 0, 0), (1001, 123456789012345, 7, 3), (32768, 1, 2 ** 33 + 5, 1000)]:
    out = torch.empty(n, dtype=torch.float32, device='cuda')
    native.fill_normal(out, seed, stream, sub)
    ref = philox.normal(n, seed, stream, sub)
    # identical counters/bits; the float32 log/sin/cos of device and NumPy differ by ulps
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-5)
