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


def test_stft_magnitude_against_scipy_and_torch():
  """External pins for N4: the framing + window + FFT of `stft_magnitude` against two STFTs this
  package did not write.
    scipy.signal.stft(window='hann' (periodic: get_window's default), nperseg=640, noverlap=320, nfft=1024,
      boundary=None, padded=False) frames the signal exactly like tf.signal.stft (frame k = samples [320 k, 320 k + 640),
      zero-padded to 1024 BEHIND the window) and divides by the window's sum;
    torch.stft(n_fft=1024, hop_length=320, win_length=640, center=False) centres the 640-sample window inside its
      1024-sample frame, i.e. it sees the samples tf.signal sees when the signal is shifted right by 192 -- a shift of
      the windowed frame inside the FFT buffer changes phases, not magnitudes.
  The zero extension of pad_end=True (ceil(n / 320) frames) is applied to the inputs of both."""
  import scipy.signal
  import torch
  rng = np.random.default_rng(3)
  for n in (3200, 3333, 641):                                 # whole frames, a ragged end, barely two frames
    x = rng.standard_normal((2, n)).astype(np.float32)
    mag = ac.stft_magnitude(x, 640, 320, 1024)
    n_frames = -(-n // 320)
    assert mag.shape == (2, n_frames, 513)
    padded = np.zeros((2, (n_frames - 1) * 320 + 640), np.float32)
    padded[:, :n] = x
    _, _, z = scipy.signal.stft(padded, fs=16000, window='hann', nperseg=640, noverlap=320, nfft=1024, boundary=None,
                                padded=False, return_onesided=True)
    win_sum = scipy.signal.get_window('hann', 640).sum()
    want = np.abs(z).transpose(0, 2, 1) * win_sum              # [batch, frames, bins]
    assert want.shape == mag.shape
    np.testing.assert_allclose(mag, want, rtol=2e-4, atol=2e-3)
    shifted = np.zeros((2, (n_frames - 1) * 320 + 1024), np.float32)
    shifted[:, 192:192 + padded.shape[1]] = padded
    zt = torch.stft(torch.as_tensor(shifted), n_fft=1024, hop_length=320, win_length=640,
                    window=torch.hann_window(640, periodic=True), center=False, return_complex=True)
    want_t = zt.abs().numpy().transpose(0, 2, 1)
    assert want_t.shape == mag.shape
    np.testing.assert_allclose(mag, want_t, rtol=2e-4, atol=2e-3)
  np.testing.assert_allclose(ac.hann_window_periodic(640), scipy.signal.get_window('hann', 640), atol=1e-7)
  np.testing.assert_allclose(ac.hann_window_periodic(640), torch.hann_window(640, periodic=True).numpy(), atol=1e-6)


def test_mel_matrix_against_a_second_construction():
  """The HTK filter bank built a second way (no library offers tf.signal's variant: torchaudio / librosa -- absent here
  anyway -- draw their triangles linear in HERTZ between mel-spaced points, tf.signal linear in MEL): numpy.interp of
  the three-point function (lower, 0), (centre, 1), (upper, 0) over the bins' mel values, band edges from the inverse
  formula f = 700 (e^{m / 1127} - 1) instead of a mel-domain linspace of the end points.  HTK's own anchor: 1000 Hz is
  1000 mel (999.99)."""
  m = ac.linear_to_mel_weight_matrix(128, 513, 16000, 0.0, 8000.0)
  assert abs(1127.0 * math.log(1.0 + 1000.0 / 700.0) - 1000.0) < 0.02
  top = 1127.0 * math.log(1.0 + 8000.0 / 700.0)
  edges_hz = 700.0 * (np.exp(np.arange(130) * (top / 129.0) / 1127.0) - 1.0)
  edges_mel = 1127.0 * np.log1p(edges_hz / 700.0)
  bins_mel = 1127.0 * np.log1p(np.arange(513) * (8000.0 / 512.0) / 700.0)
  second = np.stack([np.interp(bins_mel, edges_mel[j:j + 3], [0.0, 1.0, 0.0], left=0.0, right=0.0) for j in range(128)], 1)
  second[0] = 0.0                                             # tf.signal drops the DC bin
  np.testing.assert_allclose(m, second, atol=2e-6)
