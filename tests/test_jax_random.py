"""jax.random-compatible draws (jax itself is absent).  Pinned: the Threefry-2x32-20
core by the Random123 known-answer vectors (the same three jax's own random_test uses) and the
counter/bit layout by jax's published `random.bits(PRNGKey(1701), (3,))` values.  The float stages are
checked for distribution and against scipy's erfinv, not bit-exactly (see jax_random.py).  CPU only."""
import numpy as np

import msd_amd
from msd_amd import jax_random as jr


def test_threefry2x32_known_answers():
  kat = [((0x0, 0x0), (0x0, 0x0), (0x6b200159, 0x99ba4efe)),
         ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
         ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]
  for key, ctr, want in kat:
    a, b = jr.threefry2x32(key, np.array([ctr[0]], np.uint32), np.array([ctr[1]], np.uint32))
    assert (int(a[0]), int(b[0])) == want


def test_bit_layout_matches_jax_vector():
  # jax/tests/random_test.py: random.bits(PRNGKey(1701), (3,)) with the default threefry impl
  np.testing.assert_array_equal(jr.random_bits(jr.prng_key(1701), 3), [56197195, 4200222568, 961309823])
  # even / odd sizes share their blocks: element i < ceil(n/2) is word 0 of block (i, i + ceil(n/2))
  a = jr.random_bits((1, 2), 8)
  x0, x1 = jr.threefry2x32((1, 2), np.arange(4, dtype=np.uint32), np.arange(4, 8, dtype=np.uint32))
  np.testing.assert_array_equal(a, np.concatenate([x0, x1]))


def test_key_derivation():
  assert jr.prng_key(7) == (0, 7) and jr.prng_key((5 << 32) | 9) == (5, 9)
  k = jr.prng_key(0)
  f3 = jr.fold_in(k, 3)
  a, b = jr.threefry2x32(k, np.array([0], np.uint32), np.array([3], np.uint32))
  assert f3 == (int(a[0]), int(b[0])) and f3 != jr.fold_in(k, 4)
  s = jr.split(k, 2)
  bits = jr.random_bits(k, 4)
  assert s == [(int(bits[0]), int(bits[1])), (int(bits[2]), int(bits[3]))]


def test_normal_distribution_and_erfinv():
  from scipy.special import erfinv
  x = np.linspace(-0.999999, 0.999999, 20001).astype(np.float32)
  # the float32 polynomial is good to a few ulp (3e-6 in the tails, where 1 - x^2 loses bits)
  np.testing.assert_allclose(jr.erfinv_f32(x), erfinv(x.astype(np.float64)), rtol=5e-6, atol=2e-7)
  assert jr.erfinv_f32(np.float32(1.0)) == np.inf and jr.erfinv_f32(np.float32(-1.0)) == -np.inf
  z = jr.normal(jr.prng_key(42), (64, 256, 8))
  assert z.shape == (64, 256, 8) and z.dtype == np.float32 and np.isfinite(z).all()
  assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
  assert abs((np.abs(z) < 1.0).mean() - 0.6827) < 0.01
  init_z, noise = jr.reference_noise(3, (1, 8, 16), 5)
  assert noise.shape == (5, 1, 8, 16)
  np.testing.assert_array_equal(init_z, jr.normal(jr.prng_key(3), (1, 8, 16)))
  np.testing.assert_array_equal(noise[2], jr.normal(jr.fold_in(jr.prng_key(3), 2), (1, 8, 16)))


def test_float_stage_against_scipy_and_torch_erfinv():
  """External pin for N5's FLOAT stage (round 5): the Giles polynomial `erfinv_f32` restates XLA's float32 ErfInv; what
  it must compute is erf^-1.  Checked against scipy.special.erfinv (float64) and torch.erfinv (an independent float32
  implementation) over the whole range the uniform stage can produce -- every float32 step near +-1 included, where
  the draw's magnitude is largest: <= 1e-6 relative in the core and <= 2e-5 in the tails (the accuracy of the float32 formula itself), i.e.
  the restated constants and the two-branch structure are right.  (XLA's own rounding of the same polynomial may
  differ from NumPy's by an ulp: `normal` is seed-compatible to ~1e-7, as the module says.)"""
  import scipy.special
  import torch
  lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
  grid = np.concatenate([
      np.linspace(lo, np.float32(1.0) - np.float32(2.0 ** -24), 200001, dtype=np.float64).astype(np.float32),
      np.float32(1.0) - np.float32(2.0 ** -24) * np.arange(1, 4000, dtype=np.float32),      # the last float32 steps below 1
      lo + np.float32(2.0 ** -24) * np.arange(0, 4000, dtype=np.float32)])                  # ... and above -1
  grid = grid[(grid > -1) & (grid < 1)]
  got = jr.erfinv_f32(grid).astype(np.float64)
  want = scipy.special.erfinv(grid.astype(np.float64))
  rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
  core = np.abs(grid) < 0.9
  # float32 evaluation of w = -log1p(-x^2) loses bits as |x| -> 1 (1 - x^2 cancels): a few ulp in the core, ~1e-5 in
  # the tails -- a property of the float32 formula XLA evaluates too, not of this restatement
  assert rel[core & (np.abs(grid) > 1e-3)].max() <= 1e-6 and rel.max() <= 2e-5, (rel[core].max(), rel.max(), grid[np.argmax(rel)])
  t = torch.erfinv(torch.as_tensor(grid)).numpy().astype(np.float64)
  rel_t = np.abs(got - t) / np.maximum(np.abs(t), 1e-30)
  assert rel_t[core & (np.abs(grid) > 1e-3)].max() <= 2e-6 and rel_t.max() <= 4e-5
  # the uniform stage: [nextafter(-1, 0), 1) exactly, monotone in the 23 mantissa bits it keeps
  bits = np.array([0, 1 << 9, 0x7FFFFFFF, 0xFFFFFE00, 0xFFFFFFFF], np.uint32)
  floats = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
  u = np.maximum(lo, floats * np.float32(np.float32(1.0) - lo) + lo)
  assert u[0] == lo and u.max() < 1.0 and (np.diff(u[[0, 1, 2, 3]]) > 0).all() and u[3] == u[4]
