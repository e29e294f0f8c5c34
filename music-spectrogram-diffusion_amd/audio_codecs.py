"""Host mirror of the part of the reference ``audio_codecs.py`` that is on the
synthesis hot path: the MelGAN codec constants (audio_codecs.py:204-218) and the
linear feature scaling (audio_codecs.py:166-183).  The scaling itself runs fused
inside the HIP kernels (context-encoder input / final store); these NumPy
versions serve callers that want the same helpers the reference exposes.

``encode`` (TF STFT + mel, audio_codecs.py:226-247) and ``decode`` (TF-Hub
SoundStream, audio_codecs.py:249-264) are SURVEY.md 8(f) rows N4/N2 -- not built.
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
    raise NotImplementedError('audio -> mel is SURVEY.md 8(f) N4 (not on the hot path)')

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
    self._decode_dither_amount = decode_dither_amount


def get_codec(name: str) -> AudioCodec:
  if name in ('MelGAN', 'melgan', 'audio_codecs.MelGAN'):
    return MelGAN()
  raise ValueError('Unknown audio codec: %s' % name)
