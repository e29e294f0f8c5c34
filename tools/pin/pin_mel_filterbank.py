#!/usr/bin/env python3
"""N4 pin: is audio_codecs.linear_to_mel_weight_matrix tf.signal's (audio_codecs.py:43-143 builds MelGAN.encode on it)?

  python tools/pin/pin_mel_filterbank.py [--json out.json]
  python tools/pin/pin_mel_filterbank.py --self-test          (no tensorflow: against a second construction)

The codec's own bank (128 mel bins, 513 spectrogram bins, 16 kHz, 0 - 8000 Hz) and two more shapes.  tf.signal computes
in float64 and casts to float32 -- so does this package: expected max |difference| 0, allowed 2e-6.  Also compares the
frame / window helpers when tensorflow is present (tf.signal.stft of a fixed signal against stft_magnitude).
Exit code: 0 within tolerance | 1 differs | 2 tensorflow not importable (and no --self-test)."""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [(128, 513, 16000.0, 0.0, 8000.0), (80, 513, 16000.0, 20.0, 7600.0), (64, 257, 22050.0, 0.0, 11025.0)]


def second_construction(num_mel_bins, num_spectrogram_bins, sample_rate, lo, hi):
  """The HTK bank built another way (numpy.interp of each triangle over the bins' mel values)."""
  mel = lambda f: 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
  edges_mel = np.linspace(mel(lo), mel(hi), num_mel_bins + 2)
  bins_mel = mel(np.linspace(0.0, sample_rate / 2.0, num_spectrogram_bins))
  m = np.stack([np.interp(bins_mel, edges_mel[j:j + 3], [0.0, 1.0, 0.0], left=0.0, right=0.0) for j in range(num_mel_bins)], 1)
  m[0] = 0.0
  return m.astype(np.float32)


def compare(their_matrix, their_stft=None, tol=2e-6) -> dict:
  from msd_amd import audio_codecs as ac
  out = {'cases': [], 'ok': True}
  for case in CASES:
    ours = ac.linear_to_mel_weight_matrix(*case)
    theirs = np.asarray(their_matrix(*case), np.float32)
    err = float(np.abs(ours - theirs).max()) if ours.shape == theirs.shape else float('inf')
    ok = ours.shape == theirs.shape and err <= tol
    out['cases'].append({'case': case, 'shape': list(ours.shape), 'max_abs_diff': err, 'bitwise_equal': bool(np.array_equal(ours, theirs)), 'ok': ok})
    out['ok'] = out['ok'] and ok
  if their_stft is not None:
    rng = np.random.default_rng(0)
    sig = rng.standard_normal((2, 16000)).astype(np.float32)
    ours = ac.stft_magnitude(sig, 640, 320, 1024)
    theirs = np.asarray(their_stft(sig, 640, 320, 1024), np.float32)
    err = float(np.abs(ours - theirs).max() / np.abs(theirs).max()) if ours.shape == theirs.shape else float('inf')
    out['stft'] = {'shape': list(ours.shape), 'their_shape': list(theirs.shape), 'max_rel_diff': err, 'ok': err < 1e-5}
    out['ok'] = out['ok'] and out['stft']['ok']
  return out


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--json', default='')
  ap.add_argument('--self-test', action='store_true')
  args = ap.parse_args(argv)
  stft = None
  if args.self_test:
    matrix, against = second_construction, 'a second construction (stand-in)'
  else:
    try:
      import tensorflow as tf
    except Exception as e:
      print('tensorflow is not importable here (%s): nothing to pin against; --self-test exercises the comparison' % (repr(e)[:120],))
      return 2
    matrix = lambda nm, nb, sr, lo, hi: tf.signal.linear_to_mel_weight_matrix(nm, nb, sr, lo, hi).numpy()
    stft = lambda s, fl, fs, n: tf.abs(tf.signal.stft(s, frame_length=fl, frame_step=fs, fft_length=n, pad_end=True)).numpy()
    against = 'tensorflow %s' % tf.__version__
  res = compare(matrix, stft)
  res['against'] = against
  if args.json:
    with open(args.json, 'w') as f:
      json.dump(res, f, indent=1)
  for c in res['cases']:
    print('mel bank %s: max |diff| %.2e%s' % (c['case'], c['max_abs_diff'], ' (bitwise equal)' if c['bitwise_equal'] else ''))
  if 'stft' in res:
    print('stft magnitude: max rel diff %.2e' % res['stft']['max_rel_diff'])
  print('N4 pin against %s: %s' % (against, 'OK' if res['ok'] else 'DIFFERS'))
  return 0 if res['ok'] else 1


if __name__ == '__main__':
  sys.exit(main())
