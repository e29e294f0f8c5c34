"""Run the REFERENCE's own Python for the hot path without JAX / Flax / T5X.

`/root/reference/music_spectrogram_diffusion/{layers.py, models/diffusion/network.py,
models/diffusion/diffusion_utils.py}` are plain Python over the jax.numpy / jax.lax / flax.linen API.
None of those packages is installed here, so this file installs NumPy stand-ins for exactly the slice
of that API the three files touch (`install()`), then imports the reference modules from where they lie
(`load_reference()`): the network, the schedules, the sampler steps and the scan that make_ref_golden.py
turns into fixtures are therefore the reference's OWN statements, executed line by line -- not a
restatement.  What is ours is only the arithmetic underneath (NumPy instead of XLA), which is why the
fixtures are taken in float64 ("a jax whose float32 is 64 bits wide": WIDE = True below serves every
float32 request in float64) and compared with the float64 oracle at 1e-9, far below any float32 effect.

Test infrastructure, used only in THIS container (the GPU box has no /root/reference): nothing here is
imported by the package, the tests or bench.py; the committed outputs are tests/golden/ref_*.npz.

Stand-in semantics that matter:
  * flax.linen.Module: dataclass transform, `@compact` auto-naming (`Class_n` per parent, per call),
    `setup()` attribute naming, `self.param` / `param_with_axes` looked up in (or, mode 'init', created into)
    a nested {'params': ...} tree along the module path; shape-checked against the reference's request;
    every access recorded so a fixture can assert that the tree was consumed exactly.
  * jax.random: keys are paths; `normal` asks `noise_provider(path, shape)` (the fixture feeds the Philox
    tensors the oracle and the device get).
  * jax.lax.scan / jax.vmap: Python loops.  Dropout: identity (deterministic paths only).
"""
from __future__ import annotations

import copy
import dataclasses
import functools
import importlib
import os
import sys
import types
import zlib
from typing import Any, Optional

import numpy as np
from scipy import special as _sp

REFERENCE_ROOT = '/root/reference'
WIDE = True                       # float32 requests -> float64
noise_provider = None             # callable(path: tuple, shape) -> ndarray, set by the fixture generator


# ---------------------------------------------------------------------------------------- arrays / dtypes
def _float():
  return np.dtype(np.float64 if WIDE else np.float32)


def _dt(dtype):
  if dtype is None:
    return None
  d = np.dtype(dtype)
  return _float() if d.kind == 'f' else d


class _AtIndex:
  def __init__(self, arr, idx):
    self.arr, self.idx = arr, idx

  def set(self, v):
    out = np.array(self.arr, copy=True)
    out[self.idx] = v
    return _wrap(out)

  def add(self, v):
    out = np.array(self.arr, copy=True)
    out[self.idx] += v
    return _wrap(out)


class _At:
  def __init__(self, arr):
    self.arr = arr

  def __getitem__(self, idx):
    return _AtIndex(self.arr, idx)


class JArr(np.ndarray):
  """ndarray with the jax.Array extras the reference uses (.at[].set, dtype-mapped astype)."""

  def astype(self, dtype, *a, **k):
    return _wrap(np.asarray(self).astype(_dt(dtype), *a, **k))

  @property
  def at(self):
    return _At(self)

  def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kw):
    ins = [np.asarray(i) if isinstance(i, JArr) else i for i in inputs]
    if out is not None:
      kw['out'] = tuple(np.asarray(o) if isinstance(o, JArr) else o for o in out)
    res = getattr(ufunc, method)(*ins, **kw)
    if isinstance(res, tuple):
      return tuple(_wrap(r) for r in res)
    return _wrap(res)

  def __getitem__(self, idx):
    return _wrap(np.asarray(self)[idx])


def _wrap(x):
  if isinstance(x, (np.ndarray, np.generic)):
    a = np.asarray(x)
    if a.dtype.kind == 'f' and a.dtype != _float():
      a = a.astype(_float())
    return a.view(JArr)
  if isinstance(x, (list, tuple)) and x and all(isinstance(e, (np.ndarray, np.generic)) for e in x):
    return type(x)(_wrap(e) for e in x)
  return x


def _np_fn(name):
  f = getattr(np, name)
  if not callable(f) or isinstance(f, type):
    return f

  @functools.wraps(f)
  def g(*a, **k):
    if 'dtype' in k:
      k['dtype'] = _dt(k['dtype'])
    a = [np.asarray(x) if isinstance(x, JArr) else x for x in a]
    return _wrap(f(*a, **k))
  return g


def _asarray(x, dtype=None):
  a = np.asarray(x)
  if dtype is not None:
    a = a.astype(_dt(dtype))
  return _wrap(a)


def _make_jnp():
  m = types.ModuleType('jax.numpy')

  def _getattr(name):
    try:
      return _np_fn(name)
    except AttributeError:
      raise AttributeError('jax.numpy stand-in has no %r' % name)
  m.__getattr__ = _getattr
  m.ndarray = np.ndarray
  m.dtype = np.dtype
  for t in ('float32', 'float64', 'int32', 'int64', 'uint32', 'uint8', 'int8', 'bool_', 'integer', 'floating'):
    setattr(m, t, getattr(np, t))
  m.bfloat16 = np.float32
  m.newaxis, m.pi, m.inf = np.newaxis, np.pi, np.inf
  m.issubdtype = np.issubdtype
  m.asarray = _asarray
  m.array = lambda x, dtype=None, copy=True: _asarray(np.array(x), dtype)

  def _creator(fn):
    def c(shape, *a, dtype=None, **k):
      out = fn(shape, *a, **k)
      if dtype is not None:
        out = out.astype(_dt(dtype))
      elif out.dtype.kind == 'f':
        out = out.astype(_float())
      return _wrap(out)
    return c
  m.zeros, m.ones, m.empty = _creator(np.zeros), _creator(np.ones), _creator(np.zeros)

  def full(shape, fill_value, dtype=None):
    fv = np.asarray(fill_value)
    out = np.broadcast_to(fv, shape).copy() if fv.ndim else np.full(shape, fv[()])
    if dtype is not None:
      out = out.astype(_dt(dtype))
    return _wrap(out)
  m.full = full
  m.shape = np.shape
  m.split = lambda x, n, axis=0: [_wrap(p) for p in np.split(np.asarray(x), n, axis)]
  return m


def _make_lax():
  m = types.ModuleType('jax.lax')
  m.rsqrt = lambda x: _wrap(1.0 / np.sqrt(np.asarray(x)))
  m.square = lambda x: _wrap(np.square(np.asarray(x)))
  m.select = lambda p, a, b: _wrap(np.where(np.asarray(p), np.asarray(a), np.asarray(b)))
  m.iota = lambda dtype, n: _wrap(np.arange(n, dtype=dtype))
  m.stop_gradient = lambda x: x

  def dot_general(lhs, rhs, dimension_numbers, precision=None, preferred_element_type=None):
    (lc, rc), (lb, rb) = dimension_numbers
    if tuple(lb) or tuple(rb):
      raise NotImplementedError('batch dimensions')
    return _wrap(np.tensordot(np.asarray(lhs), np.asarray(rhs), axes=(tuple(lc), tuple(rc))))
  m.dot_general = dot_general

  def dynamic_slice_in_dim(x, start, size, axis=0):
    idx = [slice(None)] * np.ndim(x)
    idx[axis] = slice(int(start), int(start) + size)
    return _wrap(np.asarray(x)[tuple(idx)])
  m.dynamic_slice_in_dim = dynamic_slice_in_dim

  def dynamic_slice(x, start, sizes):
    idx = tuple(slice(int(s), int(s) + int(n)) for s, n in zip(start, sizes))
    return _wrap(np.asarray(x)[idx])
  m.dynamic_slice = dynamic_slice

  def scan(f, init, xs, length=None, reverse=False, unroll=1):
    n = len(xs) if xs is not None else length
    order = range(n - 1, -1, -1) if reverse else range(n)
    carry, ys = init, [None] * n
    for k in order:
      x = None if xs is None else _wrap(np.asarray(xs)[k])
      carry, ys[k] = f(carry, x)
    if all(y is None for y in ys):
      return carry, None
    return carry, _wrap(np.stack([np.asarray(y) for y in ys]))
  m.scan = scan
  return m


def _make_nn():
  m = types.ModuleType('jax.nn')
  m.sigmoid = lambda x: _wrap(_sp.expit(np.asarray(x)))
  m.log_sigmoid = lambda x: _wrap(-np.logaddexp(0.0, -np.asarray(x)))

  def softmax(x, axis=-1):
    x = np.asarray(x)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return _wrap(e / e.sum(axis=axis, keepdims=True))
  m.softmax = softmax
  m.one_hot = lambda x, n, dtype=None: _wrap((np.asarray(x)[..., None] == np.arange(n)).astype(_dt(dtype) or _float()))
  m.relu = lambda x: _wrap(np.maximum(np.asarray(x), 0))
  m.silu = m.swish = lambda x: _wrap(np.asarray(x) * _sp.expit(np.asarray(x)))
  m.tanh = lambda x: _wrap(np.tanh(np.asarray(x)))

  def gelu(x, approximate=True):
    x = np.asarray(x)
    if approximate:   # jax.nn.gelu's default: the tanh form
      return _wrap(0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3))))
    return _wrap(0.5 * x * (1.0 + _sp.erf(x / np.sqrt(2.0))))
  m.gelu = gelu
  return m


class Key:
  """A PRNG key is the path of splits / fold_ins that made it."""

  def __init__(self, path):
    self.path = tuple(path)

  def __repr__(self):
    return 'Key%r' % (self.path,)


def _make_random():
  m = types.ModuleType('jax.random')
  m.PRNGKey = lambda seed: Key(('key', int(seed)))
  m.split = lambda key, num=2: [Key(key.path + ('split', i)) for i in range(num)]
  m.fold_in = lambda key, data: Key(key.path + ('fold_in', int(data)))

  def normal(key, shape=(), dtype=None):
    if noise_provider is None:
      raise RuntimeError('ref_shim.noise_provider is not set')
    return _wrap(np.asarray(noise_provider(key.path, tuple(shape))).astype(_float()))
  m.normal = normal

  # initialisers only (MODE 'init': the VALUES are placeholders, a run takes them from the checkpoint tree)
  def uniform(key, shape=(), dtype=None, minval=0.0, maxval=1.0):
    return _wrap(_key_rng(key).uniform(minval, maxval, tuple(shape)))
  m.uniform = uniform
  m.permutation = lambda key, x, axis=0: _wrap(_key_rng(key).permutation(np.asarray(x), axis=axis))

  def _absent(name):
    def f(*a, **k):
      raise NotImplementedError('jax.random.%s: not on the deterministic inference path' % name)
    return f
  m.bernoulli = lambda key, p=0.5, shape=(): _wrap(_key_rng(key).random(tuple(shape)) < p)
  m.randint = _absent('randint')
  return m


def _key_rng(key):
  return np.random.default_rng(zlib.crc32(repr(key.path).encode()))


def _tree_map(f, tree, *rest):
  if isinstance(tree, (list, tuple)):
    return type(tree)(_tree_map(f, t, *[r[i] for r in rest]) for i, t in enumerate(tree))
  if isinstance(tree, dict):
    return {k: _tree_map(f, v, *[r[k] for r in rest]) for k, v in tree.items()}
  return f(tree, *rest)


def _vmap(fn, in_axes=0, out_axes=0):
  def g(*args):
    axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
    n = next(np.shape(a)[ax] for a, ax in zip(args, axes) if ax is not None)
    outs = []
    for i in range(n):
      outs.append(fn(*[a if ax is None else _wrap(np.take(np.asarray(a), i, axis=ax)) for a, ax in zip(args, axes)]))
    if isinstance(outs[0], tuple):
      return tuple(_wrap(np.stack([np.asarray(o[j]) for o in outs])) for j in range(len(outs[0])))
    return _wrap(np.stack([np.asarray(o) for o in outs]))
  return g


# ------------------------------------------------------------------------------------------- flax.linen
_stack = []      # modules whose method is executing, innermost last
MODE = 'apply'   # or 'init': missing parameters are created with the reference's initialiser
accessed = set() # parameter paths read since the last reset_accessed()


def reset_accessed():
  accessed.clear()


def compact(fn):
  fn._compact = True
  return fn


def _wrap_method(fn):
  is_compact = getattr(fn, '_compact', False)

  @functools.wraps(fn)
  def w(self, *a, **k):
    self._ensure_setup()
    if is_compact:
      self.__dict__['_auto'] = {}
    _stack.append(self)
    try:
      return fn(self, *a, **k)
    finally:
      _stack.pop()
  return w


@dataclasses.dataclass(eq=False, repr=False)
class Module:
  _: dataclasses.KW_ONLY
  name: Optional[str] = None
  parent: Any = None

  def __init_subclass__(cls, **kw):
    super().__init_subclass__(**kw)
    fields = cls.__dict__.get('__annotations__', {})   # a function-valued FIELD (kernel_init = ...) is data
    for k, v in list(cls.__dict__.items()):
      if (isinstance(v, types.FunctionType) and k != 'setup' and k not in fields
          and (k == '__call__' or not k.startswith('_'))):
        setattr(cls, k, _wrap_method(v))
    dataclasses.dataclass(cls, eq=False, repr=False)

  def __post_init__(self):
    d = self.__dict__
    d.update(_auto={}, _setup_done=False, _in_setup=False, _vars=None)
    if _stack and self.parent is None:
      p = _stack[-1]
      d['parent'] = p
      if self.name is None and not p.__dict__['_in_setup']:
        n = type(self).__name__
        i = p.__dict__['_auto'].get(n, 0)
        p.__dict__['_auto'][n] = i + 1
        d['name'] = '%s_%d' % (n, i)

  def __setattr__(self, k, v):
    if self.__dict__.get('_in_setup') and isinstance(v, Module):
      v.__dict__['parent'] = self
      v.__dict__['name'] = k
    object.__setattr__(self, k, v)

  def _ensure_setup(self):
    d = self.__dict__
    if d['_setup_done']:
      return
    d['_setup_done'] = True
    setup = getattr(type(self), 'setup', None)
    if setup is None:
      return
    d['_in_setup'] = True
    _stack.append(self)
    try:
      setup(self)
    finally:
      _stack.pop()
      d['_in_setup'] = False

  def _path(self):
    return () if self.parent is None else self.parent._path() + (self.name,)

  def _root(self):
    return self if self.parent is None else self.parent._root()

  def param(self, name, init_fn, *init_args):
    path = self._path() + (name,)
    tree = self._root().__dict__['_vars']['params']
    node = tree
    for p in path[:-1]:
      if p not in node:
        if MODE != 'init':
          raise KeyError('no parameter collection %r (looking for %s)' % (p, '/'.join(path)))
        node[p] = {}
      node = node[p]
    if name not in node:
      if MODE != 'init':
        raise KeyError('no parameter %s' % '/'.join(path))
      node[name] = np.asarray(init_fn(Key(('init',) + path), *init_args))
    v = np.asarray(node[name])
    if init_args:
      want = tuple(int(s) for s in init_args[0])
      if tuple(v.shape) != want:
        raise ValueError('parameter %s: tree holds %s, the reference asks for %s' % ('/'.join(path), v.shape, want))
    accessed.add('/'.join(path))
    return _wrap(v)

  def has_variable(self, col, name):
    return False

  def make_rng(self, name):
    raise NotImplementedError('make_rng(%r): dropout is not on the inference path' % name)

  def apply(self, variables, *args, method=None, rngs=None, mutable=False, **kwargs):
    clone = copy.copy(self)
    clone.__dict__.update(_auto={}, _setup_done=False, _in_setup=False, parent=None,
                          _vars={'params': variables['params']})
    if method is None:
      fn = type(self).__call__
    else:
      fn = getattr(method, '__func__', method)
    return fn(clone, *args, **kwargs)

  def init_with_output(self, rngs, *args, method=None, **kwargs):
    global MODE
    prev, MODE = MODE, 'init'
    try:
      variables = {'params': {}}
      out = self.apply(variables, *args, method=method, **kwargs)
      return out, variables
    finally:
      MODE = prev

  def init(self, rngs, *args, method=None, **kwargs):
    return self.init_with_output(rngs, *args, method=method, **kwargs)[1]


class Dropout(Module):
  rate: float = 0.0
  broadcast_dims: Any = ()
  deterministic: Optional[bool] = None

  def __call__(self, x, deterministic=None):
    det = self.deterministic if deterministic is None else deterministic
    if not det and self.rate > 0:
      raise NotImplementedError('stochastic dropout')
    return x


def _make_initializers():
  m = types.ModuleType('flax.linen.initializers')

  def variance_scaling(scale, mode, distribution, in_axis=-2, out_axis=-1, **kw):
    def init(key, shape, dtype=np.float32):
      shape = tuple(int(s) for s in shape)
      fan_in = shape[in_axis] if len(shape) > 1 else shape[0]
      fan_out = shape[out_axis] if len(shape) > 1 else shape[0]
      n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2}[mode]
      rng = _key_rng(key)
      return _wrap(rng.standard_normal(shape) * np.sqrt(scale / n))
    return init
  m.variance_scaling = variance_scaling

  def normal(stddev=1e-2):
    def init(key, shape, dtype=np.float32):
      rng = _key_rng(key)
      return _wrap(rng.standard_normal(tuple(int(s) for s in shape)) * stddev)
    return init
  m.normal = normal
  m.ones = lambda key, shape, dtype=np.float32: _wrap(np.ones(tuple(int(s) for s in shape)))
  m.zeros = lambda key, shape, dtype=np.float32: _wrap(np.zeros(tuple(int(s) for s in shape)))
  m.lecun_normal = lambda **kw: variance_scaling(1.0, 'fan_in', 'truncated_normal')
  return m


def _param_with_axes(name, init_fn, *init_args, axes=None, module=None):
  return (module or _stack[-1]).param(name, init_fn, *init_args)


def _struct_dataclass(cls):
  cls = dataclasses.dataclass(cls, frozen=True)
  cls.replace = lambda self, **kw: dataclasses.replace(self, **kw)
  return cls


# ------------------------------------------------------------------------------------------------ install
def install():
  """Put the stand-ins into sys.modules (refuses if a real jax is importable: then use the real one)."""
  if 'jax' in sys.modules and not getattr(sys.modules['jax'], '_msd_ref_shim', False):
    raise RuntimeError('a real jax is already imported')
  jnp, lax, jnn, jrandom = _make_jnp(), _make_lax(), _make_nn(), _make_random()
  jax = types.ModuleType('jax')
  jax._msd_ref_shim = True
  jax.numpy, jax.lax, jax.nn, jax.random = jnp, lax, jnn, jrandom
  jax.Array = np.ndarray
  jax.vmap = _vmap
  jax.jit = lambda f, **kw: f
  tree = types.ModuleType('jax.tree')
  tree.map = _tree_map
  jax.tree = tree
  jax.tree_map = _tree_map

  linen = types.ModuleType('flax.linen')
  linen.Module, linen.compact, linen.Dropout = Module, compact, Dropout
  linen.initializers = _make_initializers()
  linear = types.ModuleType('flax.linen.linear')
  linear.default_kernel_init = linen.initializers.lecun_normal()
  linen.linear = linear
  for act in ('relu', 'gelu', 'swish', 'silu', 'tanh', 'sigmoid', 'softmax', 'log_sigmoid'):
    setattr(linen, act, getattr(jnn, act))
  part = types.ModuleType('flax.linen.partitioning')
  part.param_with_axes = _param_with_axes
  part.with_sharding_constraint = lambda x, axes: x
  linen.partitioning = part
  struct = types.ModuleType('flax.struct')
  struct.dataclass = _struct_dataclass
  part.AxisMetadata = type('AxisMetadata', (), {})
  core = types.ModuleType('flax.core')
  core.freeze = core.unfreeze = lambda tree: tree
  flax = types.ModuleType('flax')
  flax.linen, flax.struct, flax.core = linen, struct, core
  flax._msd_ref_shim = True
  jnn.initializers = linen.initializers
  jax.config = types.SimpleNamespace(parse_flags_with_absl=lambda: None, update=lambda *a, **k: None)

  sys.modules.update({
      'jax': jax, 'jax.numpy': jnp, 'jax.lax': lax, 'jax.nn': jnn, 'jax.random': jrandom, 'jax.tree': tree,
      'flax': flax, 'flax.linen': linen, 'flax.linen.partitioning': part, 'flax.linen.initializers': linen.initializers,
      'flax.linen.linear': linear, 'flax.struct': struct, 'flax.core': core, 'jax.nn.initializers': linen.initializers,
  })


# ----------------------------------------------------- permissive stubs for what models.py merely imports
class _StubMeta(type):
  def __getattr__(cls, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return _stub(cls.__name__ + '.' + name)


class _Stub(metaclass=_StubMeta):
  """Subclassable, callable, attribute-chasable nothing."""

  def __init__(self, *a, **k):
    pass

  def __call__(self, *a, **k):
    return _Stub()

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return _Stub()


def _stub(name):
  return _StubMeta(name, (_Stub,), {})


class _StubModule(types.ModuleType):
  __path__ = []

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    full = self.__name__ + '.' + name
    if full in sys.modules:
      return sys.modules[full]
    return _stub(full)


STUBBED = ('tensorflow', 'tensorflow_hub', 'seqio', 'clu', 'gin', 't5x', 'mt3', 'imageio', 'matplotlib')


class _StubFinder:
  """Serve `import tensorflow`, `from t5x import models`, ... with empty modules: models.py imports
  them for training, metrics and feature conversion; predict_batch_with_aux touches none of them."""

  @staticmethod
  def find_spec(name, path=None, target=None):
    if name.split('.')[0] in STUBBED:
      from importlib.machinery import ModuleSpec
      return ModuleSpec(name, _StubFinder, is_package=True)
    return None

  @staticmethod
  def create_module(spec):
    return _StubModule(spec.name)

  @staticmethod
  def exec_module(module):
    pass


class BaseTransformerModel:
  """t5x.models.BaseTransformerModel as far as predict_batch_with_aux needs it: holds the module."""

  def __init__(self, module=None, *a, **k):
    self.module = module


def load_models():
  """Additionally import the reference's models.py (+ audio_codecs.py) over the stubs above."""
  ref = load_reference()
  if not any(f is _StubFinder for f in sys.meta_path):
    sys.meta_path.append(_StubFinder)
  t5x_models = importlib.import_module('t5x.models')
  t5x_models.BaseTransformerModel = BaseTransformerModel
  t5x_models.Array = np.ndarray
  core = sys.modules['flax.core']
  scope = types.ModuleType('flax.core.scope')
  scope.FrozenVariableDict = dict
  core.scope = scope
  sys.modules['flax.core.scope'] = scope
  ref.audio_codecs = importlib.import_module('music_spectrogram_diffusion.audio_codecs')
  ref.models = importlib.import_module('music_spectrogram_diffusion.models.diffusion.models')
  return ref


def install_absl_testing():
  """absl.testing.{absltest, parameterized} as far as the reference's layers_test.py uses them, over unittest."""
  import unittest

  def _expand(sets, named):
    def deco(fn):
      @functools.wraps(fn)
      def run(self):
        for s in sets:
          kw = dict(s) if isinstance(s, dict) else None
          label = kw.pop('testcase_name', None) if (kw is not None and named) else None
          with self.subTest(case=label if label is not None else s):
            fn(self, **kw) if kw is not None else fn(self, *s)
      return run
    return deco
  absl = types.ModuleType('absl')
  testing = types.ModuleType('absl.testing')
  absltest = types.ModuleType('absl.testing.absltest')
  absltest.TestCase, absltest.main = unittest.TestCase, unittest.main
  param = types.ModuleType('absl.testing.parameterized')
  param.TestCase = unittest.TestCase
  param.parameters = lambda *sets: _expand(sets, False)
  param.named_parameters = lambda *sets: _expand(sets, True)
  absl.testing, testing.absltest, testing.parameterized = testing, absltest, param
  sys.modules.update({'absl': absl, 'absl.testing': testing, 'absl.testing.absltest': absltest,
                      'absl.testing.parameterized': param})


def load_reference():
  """Import layers / network / diffusion_utils from /root/reference WITHOUT running the package
  __init__ (which pulls in TensorFlow, seqio, t5x ...): the packages are registered as bare
  namespaces whose __path__ points at the reference tree."""
  install()
  pkg_dir = os.path.join(REFERENCE_ROOT, 'music_spectrogram_diffusion')
  if not os.path.isdir(pkg_dir):
    raise RuntimeError('%s is not here: fixtures can only be regenerated next to the reference' % pkg_dir)
  for name, sub in (('music_spectrogram_diffusion', ''), ('music_spectrogram_diffusion.models', 'models'),
                    ('music_spectrogram_diffusion.models.diffusion', 'models/diffusion')):
    if name not in sys.modules:
      m = types.ModuleType(name)
      m.__path__ = [os.path.join(pkg_dir, sub)]
      sys.modules[name] = m
  layers = importlib.import_module('music_spectrogram_diffusion.layers')
  diffusion_utils = importlib.import_module('music_spectrogram_diffusion.models.diffusion.diffusion_utils')
  network = importlib.import_module('music_spectrogram_diffusion.models.diffusion.network')
  return types.SimpleNamespace(layers=layers, network=network, diffusion_utils=diffusion_utils)
