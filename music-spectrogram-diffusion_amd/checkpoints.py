"""T5X checkpoint directories <-> the flat parameter dict of this package.

The reference restores weights with t5x (`inference.py:159-176`: RestoreCheckpointConfig(path=<dir>,
mode='specific', dtype='float32')); t5x, flax, tensorstore and zarr are third-party there and absent
here, so this module restates the published on-disk layout from scratch:

  <dir>/checkpoint                      msgpack (flax.serialization) of
                                        {'version': 3, 'optimizer': {'target': <param tree>, 'state': ...}}
                                        where a leaf is either an inline array (msgpack ExtType 1 =
                                        packb((shape, dtype name, bytes))) or a TensorStore spec
                                        {'driver': 'zarr', 'kvstore': {'driver': 'file', 'path': P}, ...}
  <dir>/<P>/.zarray + chunk files       zarr v2 array; P = 'target.' + the tree path joined with '.'
                                        (e.g. target.decoder.layers_3.FiLMLayer_0.DenseGeneral_0.kernel),
                                        chunk keys 'i.j', compressor gzip (t5x default), zlib or none

Only the 'target' tree is read (the optimizer state is skipped); names come out '/'-joined, i.e. exactly
the Flax names `config.param_shapes` lists.  No checkpoint of the reference is available in this
environment (no network): the reader is pinned by a write -> read round trip through `save_t5x_checkpoint`
(which follows the same layout) and by hand-built zarr cases; InferenceModel additionally checks every
name and shape against the model's parameter tree when it loads one."""
from __future__ import annotations

import gzip
import itertools
import json
import os
import re
import zlib
from typing import Dict, Mapping, Optional

import numpy as np

_EXT_NDARRAY, _EXT_NATIVE_COMPLEX, _EXT_NPSCALAR = 1, 2, 3     # flax/serialization.py


class CheckpointError(ValueError):
  pass


# ---- zarr v2 ------------------------------------------------------------------------------
def _decompress(raw: bytes, compressor: Optional[dict]) -> bytes:
  if compressor is None:
    return raw
  cid = compressor.get('id')
  if cid == 'gzip':
    return gzip.decompress(raw)
  if cid == 'zlib':
    return zlib.decompress(raw)
  raise CheckpointError("zarr compressor '%s' is not supported (gzip, zlib, none are)" % cid)


def read_zarr_array(path: str) -> np.ndarray:
  """One zarr v2 array directory -> ndarray (C or F order, '.' or '/' chunk separators, missing
  chunks = fill_value, edge chunks stored full-size)."""
  meta_path = os.path.join(path, '.zarray')
  if not os.path.isfile(meta_path):
    raise CheckpointError('no .zarray in %s' % path)
  with open(meta_path) as f:
    meta = json.load(f)
  if meta.get('zarr_format', 2) != 2:
    raise CheckpointError('zarr_format %s is not supported' % meta.get('zarr_format'))
  if meta.get('filters'):
    raise CheckpointError('zarr filters are not supported')
  shape, chunks = tuple(meta['shape']), tuple(meta['chunks'])
  dtype = np.dtype(meta['dtype'])
  order = meta.get('order', 'C')
  sep = meta.get('dimension_separator', '.')
  fill = meta.get('fill_value')
  out = np.empty(shape, dtype)
  if fill is not None:
    out[...] = fill
  else:
    out[...] = 0
  grid = [range(-(-s // c)) for s, c in zip(shape, chunks)] if shape else [range(1)]
  for idx in itertools.product(*grid):
    key = sep.join(str(i) for i in idx) if shape else '0'
    fn = os.path.join(path, *key.split('/')) if sep == '/' else os.path.join(path, key)
    if not os.path.isfile(fn):
      continue
    with open(fn, 'rb') as f:
      buf = _decompress(f.read(), meta.get('compressor'))
    n_elem = int(np.prod(chunks)) if shape else 1
    block = np.frombuffer(buf, dtype=dtype, count=n_elem).reshape(chunks if shape else (), order=order)
    if not shape:
      out[...] = block
      continue
    sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
    out[sel] = block[tuple(slice(0, sl.stop - sl.start) for sl in sel)]
  return out


def write_zarr_array(path: str, arr: np.ndarray, compressor: Optional[str] = 'gzip', chunks=None) -> None:
  arr = np.asarray(arr)
  os.makedirs(path, exist_ok=True)
  chunks = tuple(chunks) if chunks is not None else (arr.shape if arr.ndim else ())
  meta = {'chunks': list(chunks), 'compressor': {'id': compressor, 'level': 1} if compressor else None,
          'dtype': arr.dtype.str, 'fill_value': None, 'filters': None, 'order': 'C',
          'shape': list(arr.shape), 'zarr_format': 2}
  with open(os.path.join(path, '.zarray'), 'w') as f:
    json.dump(meta, f)
  pack = {'gzip': lambda b: gzip.compress(b, 1), 'zlib': lambda b: zlib.compress(b, 1), None: lambda b: b}[compressor]
  grid = [range(-(-s // c)) for s, c in zip(arr.shape, chunks)] if arr.ndim else [range(1)]
  for idx in itertools.product(*grid):
    if arr.ndim:
      block = np.zeros(chunks, arr.dtype)
      sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
      block[tuple(slice(0, sl.stop - sl.start) for sl in sel)] = arr[sel]
      key = '.'.join(str(i) for i in idx)
    else:
      block, key = arr, '0'
    with open(os.path.join(path, key), 'wb') as f:
      f.write(pack(np.ascontiguousarray(block).tobytes()))


# ---- flax msgpack ----------------------------------------------------------------------------
def _ext_hook(code, data):
  import msgpack
  if code == _EXT_NDARRAY:
    shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape)
  if code == _EXT_NPSCALAR:
    shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape)[()]
  if code == _EXT_NATIVE_COMPLEX:
    re_, im_ = msgpack.unpackb(data, raw=False)
    return complex(re_, im_)
  return data


def _unchunk(node):
  """flax splits arrays > 2^30 bytes into {'__msgpack_chunked_array__': True, 'shape': ..., 'chunks': {...}}."""
  if isinstance(node, dict) and node.get('__msgpack_chunked_array__'):
    parts = [node['chunks'][k] for k in sorted(node['chunks'], key=int)]
    return np.concatenate([np.asarray(p).reshape(-1) for p in parts]).reshape(node['shape'])
  return node


def _is_ts_spec(node) -> bool:
  return isinstance(node, dict) and 'driver' in node and ('kvstore' in node or 'path' in node)


def _spec_path(spec: dict) -> str:
  kv = spec.get('kvstore')
  if isinstance(kv, dict) and 'path' in kv:
    return kv['path']
  if isinstance(kv, str):                          # 'file://...' URL form
    return kv.split('://', 1)[-1]
  return spec['path']


def _find_target(tree):
  if isinstance(tree, dict):
    if 'target' in tree and isinstance(tree['target'], dict):
      return tree['target']
    for key in ('optimizer', 'state', 'train_state'):
      if key in tree:
        t = _find_target(tree[key])
        if t is not None:
          return t
  return None


def resolve_checkpoint_dir(path: str) -> str:
  """`path` may be a checkpoint_<step> directory or a model directory holding several (highest step
  wins, like t5x's 'latest')."""
  if os.path.isfile(os.path.join(path, 'checkpoint')) or any(
      n.startswith('target.') for n in os.listdir(path)):
    return path
  steps = []
  for n in os.listdir(path):
    m = re.fullmatch(r'checkpoint_(\d+)', n)
    if m and os.path.isdir(os.path.join(path, n)):
      steps.append((int(m.group(1)), n))
  if not steps:
    raise CheckpointError('%s holds neither a T5X checkpoint nor checkpoint_<step> directories' % path)
  return os.path.join(path, max(steps)[1])


def load_t5x_checkpoint(path: str) -> Dict[str, np.ndarray]:
  """-> {'decoder/layers_0/.../kernel': float32 array, ..., '__step__': int} from a T5X checkpoint."""
  ckpt = resolve_checkpoint_dir(path)
  flat: Dict[str, np.ndarray] = {}
  step = 0
  m = re.search(r'checkpoint_(\d+)$', ckpt.rstrip('/'))
  if m:
    step = int(m.group(1))
  index = os.path.join(ckpt, 'checkpoint')
  if os.path.isfile(index):
    import msgpack
    with open(index, 'rb') as f:
      tree = msgpack.unpackb(f.read(), raw=False, ext_hook=_ext_hook, strict_map_key=False)
    target = _find_target(tree)
    if target is None:
      raise CheckpointError("no 'target' parameter tree in %s" % index)
    # the reference reads train_state.step (inference.py:178-181), not the directory name: a renamed
    # or copied checkpoint directory must still report the step it was saved at
    try:
      saved = _unchunk(tree['optimizer']['state']['step'])
      if _is_ts_spec(saved):
        saved = read_zarr_array(os.path.join(ckpt, _spec_path(saved)))
      step = int(np.asarray(saved).reshape(-1)[0])
    except (KeyError, TypeError, ValueError, IndexError):
      pass   # older layouts without optimizer.state.step: keep the directory name

    def walk(node, prefix):
      node = _unchunk(node)
      if _is_ts_spec(node):
        if node.get('driver') != 'zarr':
          raise CheckpointError("TensorStore driver '%s' is not supported" % node.get('driver'))
        flat['/'.join(prefix)] = read_zarr_array(os.path.join(ckpt, _spec_path(node)))
      elif isinstance(node, dict):
        for k, v in node.items():
          walk(v, prefix + [str(k)])
      elif isinstance(node, (np.ndarray, np.generic, int, float)):
        flat['/'.join(prefix)] = np.asarray(node)
      else:
        raise CheckpointError('unexpected leaf %r at %s' % (type(node), '/'.join(prefix)))
    walk(target, [])
  else:   # index missing: take every target.* array directory
    for n in sorted(os.listdir(ckpt)):
      if n.startswith('target.') and os.path.isfile(os.path.join(ckpt, n, '.zarray')):
        flat[n[len('target.'):].replace('.', '/')] = read_zarr_array(os.path.join(ckpt, n))
    if not flat:
      raise CheckpointError('no parameters found in %s' % ckpt)
  out = {k: np.asarray(v, np.float32) for k, v in flat.items()}
  out['__step__'] = np.asarray(step)
  return out


def save_t5x_checkpoint(params: Mapping[str, np.ndarray], path: str, step: int = 0,
                        inline_below: int = 0, compressor: Optional[str] = 'gzip') -> str:
  """Write `params` (flat, '/'-joined Flax names) in the T5X layout above -> the checkpoint_<step> dir.
  Arrays with fewer than `inline_below` elements go inline into the msgpack index."""
  import msgpack
  ckpt = os.path.join(path, 'checkpoint_%d' % step)
  os.makedirs(ckpt, exist_ok=True)
  tree: dict = {}
  for name, value in params.items():
    if name.startswith('__'):
      continue
    arr = np.asarray(value, np.float32)
    node = tree
    parts = name.split('/')
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    if arr.size < inline_below:
      node[parts[-1]] = msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(arr.shape), arr.dtype.name, arr.tobytes()),
                                                                    use_bin_type=True))
    else:
      rel = 'target.' + '.'.join(parts)
      write_zarr_array(os.path.join(ckpt, rel), arr, compressor)
      node[parts[-1]] = {'driver': 'zarr', 'kvstore': {'driver': 'file', 'path': rel},
                         'metadata': {'compressor': {'id': compressor} if compressor else None,
                                      'shape': list(arr.shape), 'chunks': list(arr.shape)}, 'dtype': 'float32'}
  state = {'version': 3, 'optimizer': {'target': tree, 'state': {'step': int(step), 'param_states': {}}}}
  with open(os.path.join(ckpt, 'checkpoint'), 'wb') as f:
    f.write(msgpack.packb(state, use_bin_type=True))
  return ckpt
