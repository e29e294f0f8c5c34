"""MelGAN.encode (audio -> log-mel).  The reference computes it with tf.signal
(absent); parity with TensorFlow is unpinned.  Pinned here: STFT magnitudes against a direct O(N^2)
DFT of the windowed, zero-padded frames; frame count of pad_end=True; the mel filter bank against
its closed form (HTK mel, triangles, zero DC weight); silence -> log(1e-5) = the codec's
min/pad value; a pure tone lands in the right mel bin.  CPU only."""
import math

import numpy as np

import msd_amd
from msd_amd import audio_codecs as ac


def test_stft_matches_direct_dft_and_frame_count():
  rng = np.random.default_rng(0)
  x = rng.standard_normal((2, 1000)).astype(np.float32)
  mag = ac.stft_magnitude(x, 640, 320, 1024)
  assert mag.shape == (2, 4, 513)                         # ceil(1000 / 320) frames (pad_end=True)
  w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(640) / 640)
  k = np.arange(513)[:, None]
  n = np.arange(1024)[None, :]
  dft = np.exp(-2j * np.pi * k * n / 1024)
  for b, f in [(0, 0), (1, 2), (0, 3)]:                   # frame 3 runs past the end: zero extended
    frame = np.zeros(1024)
    seg = x[b, f * 320:f * 320 + 640]
    frame[:len(seg)] = seg * w[:len(seg)]
    np.testing.assert_allclose(mag[b, f], np.abs(dft @ frame), rtol=2e-4, atol=2e-4)


def test_mel_matrix_closed_form():
  m = ac.linear_to_mel_weight_matrix(128, 513, 16000, 0.0, 8000.0)
  assert m.shape == (513, 128) and m.dtype == np.float32
  assert (m[0] == 0).all() and (m >= 0).all() and m.max() <= 1.0
  mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
  edges = np.linspace(mel(0.0), mel(8000.0), 130)
  for j in (0, 17, 64, 127):
    for i in (1, 40, 200, 512):
      f = mel(i * 8000.0 / 512)
      want = max(0.0, min((f - edges[j]) / (edges[j + 1] - edges[j]), (edges[j + 2] - f) / (edges[j + 2] - edges[j + 1])))
      assert abs(m[i, j] - want) < 1e-6
  # every filter has support; neighbouring triangles overlap so that interior bins are covered
  assert (m.sum(0) > 0).all() and (m[5:500].sum(1) > 0.5).all()


def test_encode_shapes_silence_and_tone():
  codec = ac.MelGAN()
  assert codec.encode(np.zeros((0, 100), np.float32)).shape == (0, 128)
  sil = codec.encode(np.zeros((1, 3200), np.float32))
  assert sil.shape == (1, 10, 128) and sil.dtype == np.float32
  np.testing.assert_allclose(sil, math.log(1e-5), rtol=0, atol=1e-6)
  assert abs(codec.min_value - math.log(1e-5)) < 1e-12 and codec.pad_value == codec.min_value
  t = np.arange(16000) / 16000.0
  tone = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
  mel = codec.encode(tone)                                # 1-d input = one example
  assert mel.shape == (1, 50, 128)
  peak = int(np.argmax(mel[0, 10]))
  edges_hz = 700.0 * (np.exp(np.linspace(0.0, 1127.0 * math.log(1 + 8000 / 700.0), 130) / 1127.0) - 1.0)
  assert edges_hz[peak] <= 1000.0 <= edges_hz[peak + 2]
  # features scale into the network range like any context spectrogram (audio_codecs.py:166-174)
  scaled = codec.scale_features(mel, clip=True)
  assert scaled.min() >= -1.0 and scaled.max() <= 1.0
