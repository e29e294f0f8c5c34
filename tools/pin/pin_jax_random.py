#!/usr/bin/env python3
"""N5 pin: are msd_amd.jax_random's draws the ones jax.random makes?

  python tools/pin/pin_jax_random.py [--seeds 0 1 42] [--steps 8] [--shape 1 256 128] [--json out.json]
  python tools/pin/pin_jax_random.py --self-test        (no jax: the comparison code against a stand-in)

Compares, bitwise (float32 viewed as uint32):
  * normal(PRNGKey(seed), shape)                              -- init_z of eval_scan (diffusion_utils.py:462)
  * normal(fold_in(PRNGKey(seed), i), shape) for i < steps    -- the step noise (diffusion_utils.py:389-390)
  * random.bits / fold_in key words                           -- the integer stage (already pinned by KATs: sanity)
Exit code: 0 bit-identical | 1 differs | 2 jax not importable (and no --self-test)."""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def ulp_report(got: np.ndarray, want: np.ndarray) -> dict:
  """Bitwise comparison of two float32 arrays: count of differing elements and the histogram of their ulp distances."""
  g = np.ascontiguousarray(got, np.float32).view(np.uint32).astype(np.int64).ravel()
  w = np.ascontiguousarray(want, np.float32).view(np.uint32).astype(np.int64).ravel()
  # map the sign-magnitude float order onto a monotonic integer line
  g = np.where(g & 0x80000000, 0x80000000 - (g & 0x7FFFFFFF), g + 0x80000000)
  w = np.where(w & 0x80000000, 0x80000000 - (w & 0x7FFFFFFF), w + 0x80000000)
  d = np.abs(g - w)
  bad = np.nonzero(d)[0]
  hist = {str(int(k)): int((d == k).sum()) for k in np.unique(d[bad])[:8]}
  return {'elements': int(g.size), 'differing': int(bad.size), 'max_ulp': int(d.max()) if d.size else 0, 'ulp_histogram': hist,
          'first': [int(i) for i in bad[:5]]}


def compare(theirs, seeds, steps, shape) -> dict:
  """`theirs`: an object with PRNGKey(seed), fold_in(key, i), normal(key, shape), key_words(key) -> (hi, lo)."""
  from msd_amd import jax_random as ours
  out = {'cases': [], 'ok': True}
  for seed in seeds:
    key_t, key_o = theirs.PRNGKey(seed), ours.prng_key(seed)
    rec = {'seed': seed, 'key_words_equal': tuple(int(v) for v in theirs.key_words(key_t)) == tuple(int(v) for v in key_o)}
    rec['init_z'] = ulp_report(ours.normal(key_o, shape), theirs.normal(key_t, shape))
    z, noise = ours.reference_noise(seed, shape, steps)
    rec['init_z_vectorised'] = ulp_report(z, theirs.normal(key_t, shape))
    worst = {'differing': 0}
    for i in range(steps):
      r = ulp_report(noise[i], theirs.normal(theirs.fold_in(key_t, i), shape))
      r2 = ulp_report(ours.normal(ours.fold_in(key_o, i), shape), theirs.normal(theirs.fold_in(key_t, i), shape))
      if r['differing'] + r2['differing'] >= worst['differing']:
        worst = dict(r, step=i, scalar_path_differing=r2['differing'], differing=r['differing'] + r2['differing'])
    rec['step_noise_worst'] = worst
    rec['ok'] = bool(rec['key_words_equal'] and rec['init_z']['differing'] == 0 and rec['init_z_vectorised']['differing'] == 0
                     and worst['differing'] == 0)
    out['ok'] = out['ok'] and rec['ok']
    out['cases'].append(rec)
  return out


class JaxSide:
  """jax.random behind the four calls compare() needs."""

  def __init__(self):
    import jax
    jax.config.update('jax_platforms', 'cpu') if hasattr(jax, 'config') else None
    self.jax = jax
    self.version = jax.__version__

  def PRNGKey(self, seed):
    return self.jax.random.PRNGKey(seed)

  def fold_in(self, key, i):
    return self.jax.random.fold_in(key, i)

  def normal(self, key, shape):
    return np.asarray(self.jax.random.normal(key, tuple(shape), dtype=np.float32))

  def key_words(self, key):
    k = np.asarray(self.jax.random.key_data(key) if hasattr(self.jax.random, 'key_data') else key).ravel()
    return int(k[0]), int(k[1])


class StandInSide:
  """--self-test: this package's own generator dressed as the other side (optionally with one flipped bit, to prove
  that the comparison sees a difference)."""

  def __init__(self, flip=False):
    from msd_amd import jax_random as ours
    self.ours, self.flip, self.version = ours, flip, 'stand-in'

  def PRNGKey(self, seed):
    return self.ours.prng_key(seed)

  def fold_in(self, key, i):
    return self.ours.fold_in(key, i)

  def normal(self, key, shape):
    x = self.ours.normal(key, shape).copy()
    if self.flip:
      v = x.view(np.uint32).ravel()
      v[7] ^= 1
    return x

  def key_words(self, key):
    return key


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--seeds', type=int, nargs='+', default=[0, 1, 42])
  ap.add_argument('--steps', type=int, default=8)
  ap.add_argument('--shape', type=int, nargs='+', default=[1, 256, 128])
  ap.add_argument('--json', default='')
  ap.add_argument('--self-test', action='store_true')
  ap.add_argument('--self-test-flip', action='store_true', help=argparse.SUPPRESS)
  args = ap.parse_args(argv)
  if args.self_test or args.self_test_flip:
    side = StandInSide(flip=args.self_test_flip)
  else:
    try:
      side = JaxSide()
    except Exception as e:
      print('jax is not importable here (%s): nothing to pin against; --self-test exercises the comparison' % (repr(e)[:120],))
      return 2
  res = compare(side, args.seeds, args.steps, args.shape)
  res['against'] = 'jax %s' % side.version if isinstance(side, JaxSide) else side.version
  if args.json:
    with open(args.json, 'w') as f:
      json.dump(res, f, indent=1)
  for c in res['cases']:
    print('seed %-4d key words %s | init_z differing %d (max %d ulp) | worst step %s: %d differing (max %d ulp) %s'
          % (c['seed'], 'ok' if c['key_words_equal'] else 'DIFFER', c['init_z']['differing'], c['init_z']['max_ulp'],
             c['step_noise_worst'].get('step'), c['step_noise_worst']['differing'], c['step_noise_worst'].get('max_ulp', 0),
             c['step_noise_worst'].get('ulp_histogram', '')))
  print('N5 pin against %s: %s' % (res['against'], 'BIT-IDENTICAL' if res['ok'] else 'DIFFERS'))
  return 0 if res['ok'] else 1


if __name__ == '__main__':
  sys.exit(main())
