"""NumPy restatement of the library's counter-based normal generator
(csrc/elementwise.h ``philox_normal_kernel``; msd_amd.h ``msd_fill_normal``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

This generator REPLACES ``jax.random.normal(PRNGKey(seed))`` /
``normal(fold_in(rng, i))`` (diffusion_utils.py:389-390,462): jax's threefry is
version-dependent and cannot be imported here, so "identical seeds" is defined
on this documented generator and parity with the reference is defined on
identical noise TENSORS.

  Philox4x32-10 (Salmon et al. 2011), key (seed_lo, seed_hi),
  counter (block, subseq, stream_lo, stream_hi); element e = 4*block + j.
  subseq 0 = init_z, subseq 1+i = the step-i draw; stream = segment index.
  u = ((x >> 8) + 0.5) * 2^-24;  (z0,z1) = sqrt(-2 ln u0) * (cos, sin)(2 pi u1)
  from words (0,1); (z2,z3) from words (2,3).  float32 arithmetic.
Known answer pinning Philox itself: tests/test_oracle_philox.py uses the
Random123 KAT vectors for philox4x32_10.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key) -> np.ndarray:
  """counter uint32 [..., 4], key (k0, k1) -> uint32 [..., 4]."""
  c = [counter[..., i].astype(np.uint64) for i in range(4)]
  k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
  for _ in range(10):
    p0 = M0 * c[0]
    p1 = M1 * c[2]
    hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
    hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
    c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
    k0 = (k0 + W0) & 0xFFFFFFFF
    k1 = (k1 + W1) & 0xFFFFFFFF
  return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def normal(n: int, seed: int, stream_id: int, subseq: int) -> np.ndarray:
  """float32 [n] standard normals, element order identical to the device kernel."""
  nblk = (n + 3) // 4
  ctr = np.zeros((nblk, 4), np.uint32)
  ctr[:, 0] = np.arange(nblk, dtype=np.uint64).astype(np.uint32)
  ctr[:, 1] = np.uint32(subseq & 0xFFFFFFFF)
  ctr[:, 2] = np.uint32(stream_id & 0xFFFFFFFF)
  ctr[:, 3] = np.uint32((stream_id >> 32) & 0xFFFFFFFF)
  x = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
  u = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
  out = np.empty((nblk, 4), np.float32)
  for h in range(2):
    rad = np.sqrt(np.float32(-2.0) * np.log(u[:, 2 * h]))
    ang = np.float32(6.283185307179586) * u[:, 2 * h + 1]
    out[:, 2 * h] = rad * np.cos(ang)
    out[:, 2 * h + 1] = rad * np.sin(ang)
  return out.reshape(-1)[:n]


def segment_noise(shape_btn, num_steps: int, seed: int, segment: int):
  """(init_z [B,T,n], noise [N,B,T,n]) exactly as msd_sample generates them."""
  n = int(np.prod(shape_btn))
  init_z = normal(n, seed, segment, 0).reshape(shape_btn)
  noise = np.stack([normal(n, seed, segment, 1 + i).reshape(shape_btn)
                    for i in range(num_steps)])
  return init_z, noise
