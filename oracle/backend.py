"""Array backends for the oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

The oracle is written once against this tiny interface and can run on
  * ``NumpyBackend('float64' | 'float32')`` -- the canonical restatement, and
  * ``TorchBackend('float32' | 'float64', threads)`` -- the same arithmetic on
    torch-CPU (about 4-5x faster sgemm here), used for full-size segments and
    for bench.py's ``cpu_baseline``.
Both keep every array in the requested dtype (no silent float64 promotion).
"""
from __future__ import annotations

import numpy as np


def effective_cpus() -> int:
  """CPUs this process may actually use: min(affinity mask, cgroup v2/v1 quota).  (The GPU
  box shows 256 logical CPUs under a 16-CPU quota; 256 torch threads there never finish.)"""
  import os
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
    if q != 'max':
      n = min(n, max(1, int(int(q) / int(p))))
  except Exception:
    try:
      q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
      p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
      if q > 0:
        n = min(n, max(1, q // p))
    except Exception:
      pass
  return max(1, n)


class NumpyBackend:
  name = 'numpy'

  def __init__(self, dtype='float32'):
    self.dtype = np.dtype(dtype)

  # -- construction ---------------------------------------------------------
  def asarray(self, x):
    return np.asarray(x, dtype=self.dtype)

  def asint(self, x):
    return np.asarray(x, dtype=np.int64)

  def to_numpy(self, x):
    return np.asarray(x)

  def zeros(self, shape):
    return np.zeros(shape, self.dtype)

  def ones(self, shape):
    return np.ones(shape, self.dtype)

  def arange(self, n):
    return np.arange(n, dtype=self.dtype)

  def full(self, shape, v):
    return np.full(shape, v, dtype=self.dtype)

  # -- elementwise ----------------------------------------------------------
  exp = staticmethod(np.exp)
  expm1 = staticmethod(np.expm1)
  log = staticmethod(np.log)
  log1p = staticmethod(np.log1p)
  tan = staticmethod(np.tan)
  tanh = staticmethod(np.tanh)
  sin = staticmethod(np.sin)
  cos = staticmethod(np.cos)
  sqrt = staticmethod(np.sqrt)
  square = staticmethod(np.square)
  maximum = staticmethod(np.maximum)
  where = staticmethod(np.where)

  def clip(self, x, lo, hi):
    return np.clip(x, lo, hi)

  def sigmoid(self, x):
    # jax.nn.sigmoid == expit
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)

  # -- reductions / shape ---------------------------------------------------
  def sum(self, x, axis, keepdims=False):
    return np.sum(x, axis=axis, keepdims=keepdims)

  def mean(self, x, axis, keepdims=False):
    return np.mean(x, axis=axis, keepdims=keepdims)

  def max(self, x, axis, keepdims=False):
    return np.max(x, axis=axis, keepdims=keepdims)

  def any(self, x, axis, keepdims=False):
    return np.any(x, axis=axis, keepdims=keepdims)

  def concatenate(self, xs, axis):
    return np.concatenate(xs, axis=axis)

  def reshape(self, x, shape):
    return np.reshape(x, shape)

  def expand_dims(self, x, axis):
    return np.expand_dims(x, axis)

  def squeeze(self, x, axis):
    return np.squeeze(x, axis)

  def take(self, table, idx):
    return table[idx]

  def roll(self, x, shift, axis):
    return np.roll(x, shift, axis=axis)

  def cast(self, x):
    return np.asarray(x).astype(self.dtype)

  def round_bf16(self, x):
    """Round-to-nearest-even to bfloat16 precision, result kept in self.dtype."""
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return u.view(np.float32).astype(self.dtype)

  def round_f16(self, x):
    """Round-to-nearest-even to IEEE half precision, saturating at +-65504, kept in self.dtype."""
    return np.clip(np.asarray(x, np.float32), -65504.0, 65504.0).astype(np.float16).astype(self.dtype)

  # -- contractions ---------------------------------------------------------
  def matmul(self, a, b):
    return np.matmul(a, b)

  def einsum(self, eq, *ops):
    return np.einsum(eq, *ops, optimize=True)


class TorchBackend:
  name = 'torch'

  def __init__(self, dtype='float32', threads=None):
    import torch  # local import: torch is plumbing for the fast CPU leg only
    self.t = torch
    torch.set_num_threads(int(threads) if threads else effective_cpus())
    self.dtype = {'float32': torch.float32, 'float64': torch.float64}[
        str(np.dtype(dtype))]
    t = torch
    self.exp, self.expm1, self.log, self.log1p = t.exp, t.expm1, t.log, t.log1p
    self.tan, self.tanh, self.sin, self.cos = t.tan, t.tanh, t.sin, t.cos
    self.sqrt, self.square = t.sqrt, t.square
    self.sigmoid = t.sigmoid
    self.matmul, self.einsum = t.matmul, t.einsum

  def asarray(self, x):
    t = self.t
    if isinstance(x, t.Tensor):
      return x.to(self.dtype)
    return t.as_tensor(np.ascontiguousarray(x)).to(self.dtype)

  def asint(self, x):
    t = self.t
    if isinstance(x, t.Tensor):
      return x.to(t.int64)
    return t.as_tensor(np.ascontiguousarray(x)).to(t.int64)

  def to_numpy(self, x):
    return x.detach().cpu().numpy() if isinstance(x, self.t.Tensor) else np.asarray(x)

  def zeros(self, shape):
    return self.t.zeros(tuple(shape), dtype=self.dtype)

  def ones(self, shape):
    return self.t.ones(tuple(shape), dtype=self.dtype)

  def arange(self, n):
    return self.t.arange(n, dtype=self.dtype)

  def full(self, shape, v):
    return self.t.full(tuple(shape), float(v), dtype=self.dtype)

  def maximum(self, a, b):
    t = self.t
    if not isinstance(a, t.Tensor):
      a = t.as_tensor(a, dtype=b.dtype)
    if not isinstance(b, t.Tensor):
      b = t.as_tensor(b, dtype=a.dtype)
    return t.maximum(a, b)

  def where(self, c, a, b):
    t = self.t
    if not isinstance(a, t.Tensor):
      a = t.as_tensor(a, dtype=b.dtype if isinstance(b, t.Tensor) else self.dtype)
    if not isinstance(b, t.Tensor):
      b = t.as_tensor(b, dtype=a.dtype)
    return t.where(c, a, b)

  def clip(self, x, lo, hi):
    return self.t.clamp(x, float(lo), float(hi))

  def sum(self, x, axis, keepdims=False):
    return self.t.sum(x, dim=axis, keepdim=keepdims)

  def mean(self, x, axis, keepdims=False):
    return self.t.mean(x, dim=axis, keepdim=keepdims)

  def max(self, x, axis, keepdims=False):
    return self.t.amax(x, dim=axis, keepdim=keepdims)

  def any(self, x, axis, keepdims=False):
    return self.t.any(x, dim=axis, keepdim=keepdims)

  def concatenate(self, xs, axis):
    return self.t.cat(list(xs), dim=axis)

  def reshape(self, x, shape):
    return x.reshape(tuple(shape))

  def expand_dims(self, x, axis):
    return x.unsqueeze(axis)

  def squeeze(self, x, axis):
    return x.squeeze(axis)

  def take(self, table, idx):
    return table[idx]

  def roll(self, x, shift, axis):
    return self.t.roll(x, int(shift), dims=axis)

  def cast(self, x):
    return self.asarray(x)

  def round_bf16(self, x):
    return x.to(self.t.bfloat16).to(self.dtype)

  def round_f16(self, x):
    return x.clamp(-65504.0, 65504.0).to(self.t.float16).to(self.dtype)
