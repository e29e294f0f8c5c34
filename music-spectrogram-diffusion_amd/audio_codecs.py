"""Host mirror of the part of the reference ``audio_codecs.py`` that is on the
synthesis hot path: the MelGAN codec constants (audio_codecs.py:204-218) and the
linear feature scaling (audio_codecs.py:166-183).  The scaling itself runs fused
inside the HIP kernels (context-encoder input / final store); these NumPy
versions serve callers that want the same helpers the reference exposes.

``encode`` (audio -> log-mel, audio_codecs.py:43-143, 226-247) is restated
in NumPy below: the reference computes it with tf.signal (absent here), so it follows the
published definitions of tf.signal.stft(pad_end=True), hann_window(periodic) and
linear_to_mel_weight_matrix (HTK mel scale); parity with TensorFlow is UNPINNED (no TF, no
reference vectors) -- tests/test_audio_encode.py pins it against a direct DFT and the closed-form
filter bank.  ``decode`` (TF-Hub SoundStream, audio_codecs.py:249-264) is row N2 -- not built.
"""
from __future__ import annotations

import math

import numpy as np


class AudioCodec:
  name: str
  n_dims: int
  sample_rate: int
  hop_size: int
  min_value: float
  max_value: float
  pad_value: float
  additional_frames_for_encoding: int = 0

  @property
  def abbrev_str(self):
    return self.name

  @property
  def frame_rate(self):
    return int(self.sample_rate // self.hop_size)

  def scale_features(self, features, output_range=(-1.0, 1.0), clip=False):
    """Linearly scale features to the network range (audio_codecs.py:166-174)."""
    min_out, max_out = output_range
    features = np.asarray(features)
    if clip:
      features = np.clip(features, self.min_value, self.max_value)
    zero_one = (features - self.min_value) / (self.max_value - self.min_value)
    return zero_one * (max_out - min_out) + min_out

  def scale_to_features(self, outputs, input_range=(-1.0, 1.0), clip=False):
    """Inverse scaling (audio_codecs.py:176-183)."""
    min_out, max_out = input_range
    outputs = np.asarray(outputs)
    outputs = np.clip(outputs, min_out, max_out) if clip else outputs
    zero_one = (outputs - min_out) / (max_out - min_out)
    return zero_one * (self.max_value - self.min_value) + self.min_value

  def encode(self, audio):
    raise NotImplementedError('codec %s has no encoder' % getattr(self, 'name', '?'))

  def decode(self, features):
    raise NotImplementedError(
        'mel -> audio needs the TF-Hub SoundStream artifact (audio_codecs.py:31-33), '
        'unavailable here: SURVEY.md 8(f) N2')

  @property
  def context_codec(self):
    return self


class MelGAN(AudioCodec):
  """Invertible mel spectrogram, 128 dims @ 16 kHz (audio_codecs.py:204-218)."""
  name = 'melgan'
  n_dims = 128
  sample_rate = 16000
  hop_size = 320
  min_value = math.log(1e-5)
  max_value = 4.0
  pad_value = math.log(1e-5)
  additional_frames_for_encoding = 16

  def __init__(self, decode_dither_amount: float = 0.0):
    self._frame_length = 640
    self._fft_size = 1024
    self._lo_hz = 0.0
    self._decode_dither_amount = decode_dither_amount
    self._mel_basis = None

  def encode(self, audio):
    """audio float [batch, n_samples] -> log-mel float32 [batch, ceil(n_samples / hop), 128]
    (audio_codecs.py:226-247 -> Audio2Mel.call :107-143)."""
    audio = np.asarray(audio, np.float32)
    if audio.ndim == 1:
      audio = audio[None]
    if audio.shape[0] == 0:
      return np.zeros((0, self.n_dims), np.float32)       # audio_codecs.py:235-238
    if self._mel_basis is None:
      self._mel_basis = linear_to_mel_weight_matrix(self.n_dims, self._fft_size // 2 + 1, self.sample_rate,
                                                    self._lo_hz, float(self.sample_rate // 2))
    mag = stft_magnitude(audio, self._frame_length, self.hop_size, self._fft_size)
    mel = mag @ self._mel_basis
    return np.log(np.clip(mel, 1e-5, 1e8)).astype(np.float32)


def hann_window_periodic(n: int) -> np.ndarray:
  """tf.signal.hann_window(n, periodic=True): 0.5 - 0.5 cos(2 pi k / n)."""
  return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def stft_magnitude(signals: np.ndarray, frame_length: int, frame_step: int, fft_length: int) -> np.ndarray:
  """|tf.signal.stft(signals, frame_length, frame_step, fft_length, hann_window, pad_end=True)|:
  frame k covers samples [k*step, k*step + frame_length) of the zero-extended signal, k < ceil(n / step);
  windowed, zero-padded to fft_length, real FFT -> fft_length // 2 + 1 bins."""
  b, n = signals.shape
  n_frames = -(-n // frame_step)
  padded = np.zeros((b, (n_frames - 1) * frame_step + frame_length), np.float32)
  padded[:, :n] = signals
  idx = np.arange(frame_length)[None, :] + frame_step * np.arange(n_frames)[:, None]
  frames = padded[:, idx] * hann_window_periodic(frame_length)
  return np.abs(np.fft.rfft(frames, n=fft_length, axis=-1)).astype(np.float32)


def linear_to_mel_weight_matrix(num_mel_bins: int, num_spectrogram_bins: int, sample_rate: float,
                                lower_edge_hertz: float, upper_edge_hertz: float) -> np.ndarray:
  """tf.signal.linear_to_mel_weight_matrix: HTK mel scale mel(f) = 1127 ln(1 + f / 700); triangular
  filters between num_mel_bins + 2 band edges equally spaced in mel; the DC bin gets zero weight."""
  def hz_to_mel(f):
    return 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
  bins_hz = np.linspace(0.0, sample_rate / 2.0, num_spectrogram_bins)[1:]
  bins_mel = hz_to_mel(bins_hz)[:, None]
  edges = np.linspace(hz_to_mel(lower_edge_hertz), hz_to_mel(upper_edge_hertz), num_mel_bins + 2)
  lower, center, upper = edges[:-2][None], edges[1:-1][None], edges[2:][None]
  w = np.maximum(0.0, np.minimum((bins_mel - lower) / (center - lower), (upper - bins_mel) / (upper - center)))
  return np.concatenate([np.zeros((1, num_mel_bins)), w], 0).astype(np.float32)


def get_codec(name: str) -> AudioCodec:
  if name in ('MelGAN', 'melgan', 'audio_codecs.MelGAN'):
    return MelGAN()
  raise ValueError('Unknown audio codec: %s' % name)
