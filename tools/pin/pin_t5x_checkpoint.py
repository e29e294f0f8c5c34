#!/usr/bin/env python3
"""N3 pin: does checkpoints.load_t5x_checkpoint read a REAL T5X checkpoint of the reference, and is the tree the model's?

  python tools/pin/pin_t5x_checkpoint.py <checkpoint dir or model dir> [--preset base_with_context] [--json out.json]
  python tools/pin/pin_t5x_checkpoint.py --self-test        (writes a full-tree synthetic checkpoint with flax's msgpack
                                                             layout + raw zarr chunks, NOT with this package's writer,
                                                             and runs the same checks on it)

Checks (inference.py:159-176; README.md:24-25 names the released checkpoints):
  * every parameter the C-ABI declares (msd_weight_info == the tree the reference's module.init creates) is present
    with its shape; nothing else is (optimizer slots skipped);
  * the parameter count (base_with_context: 411.67 M; small: 84.96 M -- SURVEY 8);
  * every value finite, no all-zero matrix; the train step is reported.
Exit code: 0 the tree is the model's | 1 mismatch."""
from __future__ import annotations

import argparse
import gzip
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

EXPECTED_MILLIONS = {'base_with_context': 411.67, 'small': 84.96}


def expected_tree(preset: str):
  """{name: shape} of the model (config.param_shapes; tests/test_ref_golden.py
  test_parameter_tree_is_the_one_the_reference_creates pins that tree to the reference's own module.init)."""
  import msd_amd
  spec = msd_amd.config.preset(preset)
  return {k: tuple(v) for k, v in msd_amd.config.param_shapes(spec).items()}


def check(path: str, preset: str) -> dict:
  from msd_amd import checkpoints
  want = expected_tree(preset)
  got = checkpoints.load_t5x_checkpoint(path)
  step = int(np.asarray(got.pop('__step__', -1)))
  missing = sorted(set(want) - set(got))
  extra = sorted(set(got) - set(want))
  misshapen = sorted(k for k in set(want) & set(got) if tuple(got[k].shape) != want[k])
  count = sum(int(np.prod(v.shape)) for k, v in got.items() if k in want)
  nonfinite = sorted(k for k, v in got.items() if not np.isfinite(v).all())
  zero = sorted(k for k, v in got.items() if v.ndim == 2 and not v.any())
  exp = EXPECTED_MILLIONS.get(preset)
  ok = not (missing or extra or misshapen or nonfinite or zero) and (exp is None or abs(count / 1e6 - exp) < 0.01)
  return {'path': path, 'preset': preset, 'step': step, 'parameters': count, 'parameters_M': round(count / 1e6, 2),
          'expected_M': exp, 'entries': len(got), 'missing': missing[:20], 'unexpected': extra[:20], 'misshapen': misshapen[:20],
          'nonfinite': nonfinite[:20], 'all_zero_matrices': zero[:20], 'ok': bool(ok)}


def write_flax_layout(params, root, step=7000, inline_below=2 ** 12, rows_per_chunk=512):
  """A checkpoint directory the way flax.serialization / t5x.checkpoints lay it out, packed with the `msgpack` library and
  hand-written zarr v2 chunk files (NOT checkpoints.save_t5x_checkpoint): small arrays inline as ExtType 1, large ones as
  TensorStore specs + `target.<dotted name>` zarr directories, optimizer slots beside them."""
  import msgpack
  ckpt = os.path.join(root, 'checkpoint_%d' % step)
  os.makedirs(ckpt)

  def inline(arr):
    arr = np.asarray(arr)
    return msgpack.ExtType(1, msgpack.packb((list(arr.shape), arr.dtype.name, arr.tobytes()), use_bin_type=True))

  def zarr_dir(rel, arr):
    d = os.path.join(ckpt, rel)
    os.makedirs(d)
    chunks = [min(rows_per_chunk, arr.shape[0])] + list(arr.shape[1:])
    with open(os.path.join(d, '.zarray'), 'w') as f:
      json.dump({'chunks': chunks, 'compressor': {'id': 'gzip', 'level': 1}, 'dtype': arr.dtype.str, 'fill_value': None,
                 'filters': None, 'order': 'C', 'shape': list(arr.shape), 'zarr_format': 2}, f)
    for i in range(-(-arr.shape[0] // chunks[0])):
      block = np.zeros(chunks, arr.dtype)
      part = arr[i * chunks[0]:(i + 1) * chunks[0]]
      block[:len(part)] = part
      with open(os.path.join(d, '.'.join([str(i)] + ['0'] * (arr.ndim - 1))), 'wb') as f:
        f.write(gzip.compress(block.tobytes(), 1))
    return {'driver': 'zarr', 'dtype': arr.dtype.name, 'kvstore': {'driver': 'file', 'path': rel},
            'metadata': {'chunks': chunks, 'compressor': {'id': 'gzip'}, 'shape': list(arr.shape)}}

  target, slots = {}, {}
  for name, arr in params.items():
    parts = name.split('/')
    node, snode = target, slots
    for p in parts[:-1]:
      node = node.setdefault(p, {})
      snode = snode.setdefault(p, {})
    if arr.size < inline_below:
      node[parts[-1]] = inline(arr)
    else:
      node[parts[-1]] = zarr_dir('target.' + '.'.join(parts), arr)
      snode[parts[-1]] = {'v_row': inline(np.zeros((arr.shape[0],), np.float32))}
  state = {'version': 3, 'optimizer': {'state': {'step': inline(np.asarray(step, np.int32)), 'param_states': slots}, 'target': target}}
  with open(os.path.join(ckpt, 'checkpoint'), 'wb') as f:
    f.write(msgpack.packb(state, use_bin_type=True))
  return ckpt


def self_test(preset='tiny_context') -> dict:
  import msd_amd
  spec = msd_amd.config.preset(preset)
  params = msd_amd.synthetic.init_params(spec, 0)
  with tempfile.TemporaryDirectory(prefix='msd_pin_ckpt_') as tmp:
    write_flax_layout(params, tmp)
    res = check(tmp, preset)
    # a damaged copy must be caught: drop one matrix, reshape another
    broken = dict(params)
    victim = sorted(k for k, v in broken.items() if v.ndim == 2)[0]
    broken.pop(victim)
    other = sorted(k for k, v in broken.items() if v.ndim == 2 and v.shape[0] != v.shape[1])[0]
    broken[other] = np.ascontiguousarray(broken[other].T)
    with tempfile.TemporaryDirectory(prefix='msd_pin_ckpt_bad_') as tmp2:
      write_flax_layout(broken, tmp2)
      bad = check(tmp2, preset)
  res['self_test_damaged_copy_caught'] = bool(not bad['ok'] and victim in bad['missing'] and other in bad['misshapen'])
  res['ok'] = bool(res['ok'] and res['self_test_damaged_copy_caught'])
  return res


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('path', nargs='?')
  ap.add_argument('--preset', default='base_with_context')
  ap.add_argument('--json', default='')
  ap.add_argument('--self-test', action='store_true')
  args = ap.parse_args(argv)
  if args.self_test:
    res = self_test()
  elif args.path:
    res = check(args.path, args.preset)
  else:
    ap.error('a checkpoint directory (or --self-test)')
  if args.json:
    with open(args.json, 'w') as f:
      json.dump(res, f, indent=1)
  print(json.dumps(res, indent=1))
  print('N3 pin: %s' % ('OK: the tree is the model\'s (%s M parameters, step %s)' % (res['parameters_M'], res['step']) if res['ok'] else 'MISMATCH'))
  return 0 if res['ok'] else 1


if __name__ == '__main__':
  sys.exit(main())
