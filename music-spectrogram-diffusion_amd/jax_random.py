"""jax.random-compatible noise for `seed=` parity with the reference.

The reference draws `init_z = jax.random.normal(PRNGKey(seed), [B, T, n])` and, at scan index i,
`eps = jax.random.normal(fold_in(PRNGKey(seed), i), [B, T, n])` (inference.py:203,
models/diffusion/diffusion_utils.py:389-390, 462).  jax is a third-party dependency that is absent
here and whose version the reference does not pin; this module restates the published algorithm of
its default generator (jax/_src/prng.py, random.py, non-partitionable threefry -- the default up to
jax 0.4.x):
  * threefry2x32, 20 rounds, key schedule constant 0x1BD11BDA (Random123);
  * PRNGKey(seed) = (seed >> 32, seed & 0xffffffff);  fold_in(key, d) = threefry(key, (0, d));
  * random bits for n values: counters 0..n-1 (zero-padded to even), first half = word 0, second
    half = word 1 of the blocks; outputs concatenated the same way;
  * uniform in [nextafter(-1, 0), 1): mantissa trick (bits >> 9 | 0x3f800000) - 1, scaled in float32;
  * normal = sqrt(2) * erfinv(u) with XLA's float32 erfinv (Giles' polynomial).
Pinned: the threefry core by the Random123 / jax known-answer vectors and the bit layout by jax's
own `random.bits(PRNGKey(1701), (3,))` vector (tests/test_jax_random.py).  NOT pinned: the float
stages (XLA may contract a multiply-add or round log1p differently by 1 ulp), so `normal` is
seed-compatible to ~1e-7, not guaranteed bit-exact; jax >= 0.5 defaults to the partitionable
layout, which differs."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
  return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(key: Tuple[int, int], x0, x1):
  """Random123 Threefry-2x32-20 on arrays of counter words."""
  with np.errstate(over='ignore'):
    k0, k1 = np.asarray(key[0], _U32), np.asarray(key[1], _U32)   # scalars, or arrays broadcast against x
    ks = (k0, k1, k0 ^ k1 ^ _U32(0x1BD11BDA))
    x0 = np.asarray(x0, _U32) + ks[0]
    x1 = np.asarray(x1, _U32) + ks[1]
    for g in range(5):
      for r in _ROT[g % 2]:
        x0 = x0 + x1
        x1 = _rotl(x1, r) ^ x0
      x0 = x0 + ks[(g + 1) % 3]
      x1 = x1 + ks[(g + 2) % 3] + _U32(g + 1)
  return x0, x1


def prng_key(seed: int) -> Tuple[int, int]:
  seed = int(seed)
  return ((seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF)


def fold_in(key: Tuple[int, int], data: int) -> Tuple[int, int]:
  a, b = threefry2x32(key, np.array([0], _U32), np.array([int(data) & 0xFFFFFFFF], _U32))
  return int(a[0]), int(b[0])


def split(key: Tuple[int, int], num: int = 2):
  bits = random_bits(key, 2 * num)
  return [(int(bits[2 * i]), int(bits[2 * i + 1])) for i in range(num)]


def random_bits(key: Tuple[int, int], n: int) -> np.ndarray:
  """uint32 [n], the layout of jax's threefry_random_bits for 32-bit values (n < 2**32)."""
  half = (n + 1) // 2
  counts = np.arange(2 * half, dtype=np.uint64).astype(_U32)
  if n % 2:
    counts[-1] = 0                       # the zero pad of threefry_2x32's odd-size path
  a, b = threefry2x32(key, counts[:half], counts[half:])
  return np.concatenate([a, b])[:n]


_ERFINV_LT5 = np.array([2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
                        -0.00125372503, -0.00417768164, 0.246640727, 1.50140941], np.float32)
_ERFINV_GE5 = np.array([-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
                        -0.0076224613, 0.00943887047, 1.00167406, 2.83297682], np.float32)


def erfinv_f32(x: np.ndarray) -> np.ndarray:
  """XLA's float32 ErfInv (M. Giles, 'Approximating the erfinv function')."""
  x = np.asarray(x, np.float32)
  with np.errstate(divide='ignore'):
    w = -np.log1p(-(x * x)).astype(np.float32)
  lt = w < np.float32(5.0)
  w = np.where(lt, w - np.float32(2.5), np.sqrt(w) - np.float32(3.0)).astype(np.float32)
  p = np.where(lt, _ERFINV_LT5[0], _ERFINV_GE5[0]).astype(np.float32)
  for i in range(1, 9):
    p = (np.where(lt, _ERFINV_LT5[i], _ERFINV_GE5[i]).astype(np.float32) + p * w).astype(np.float32)
  return np.where(np.abs(x) == 1, np.copysign(np.float32(np.inf), x), p * x).astype(np.float32)


def uniform_f32(key: Tuple[int, int], n: int, minval: np.float32, maxval: np.float32) -> np.ndarray:
  bits = random_bits(key, n)
  floats = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
  return np.maximum(minval, floats * np.float32(maxval - minval) + minval).astype(np.float32)


def normal(key: Tuple[int, int], shape: Sequence[int]) -> np.ndarray:
  """jax.random.normal(key, shape, float32)."""
  n = int(np.prod(shape))
  lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
  u = uniform_f32(key, n, lo, np.float32(1.0))
  return (np.float32(np.sqrt(2)) * erfinv_f32(u)).reshape(shape)


def reference_noise(seed: int, shape: Sequence[int], num_steps: int):
  """(init_z [B,T,n], noise [N,B,T,n]) as eval_scan draws them for `InferenceModel.predict(batch, seed)`
  (noise[i] = the draw at scan index i, diffusion_utils.py:389-390).  All steps in one vectorised pass."""
  key = prng_key(seed)
  init_z = normal(key, shape)
  n = int(np.prod(shape))
  half = (n + 1) // 2
  # fold_in(key, i) for every i at once
  f0, f1 = threefry2x32(key, np.zeros(num_steps, _U32), np.arange(num_steps, dtype=_U32))
  counts = np.arange(2 * half, dtype=np.uint64).astype(_U32)
  if n % 2:
    counts[-1] = 0
  a, b = threefry2x32((f0[:, None], f1[:, None]), counts[None, :half], counts[None, half:])
  bits = np.concatenate([a, b], axis=1)[:, :n]
  lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
  floats = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
  u = np.maximum(lo, floats * np.float32(np.float32(1.0) - lo) + lo).astype(np.float32)
  noise = (np.float32(np.sqrt(2)) * erfinv_f32(u)).reshape((num_steps,) + tuple(shape))
  return init_z, noise
