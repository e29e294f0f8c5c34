"""Vocabulary configuration, the MT3-style codec, velocity bins, program granularities and the
pass-through token vocabulary (reference: vocabularies.py:20-281).  NumPy only."""
from __future__ import annotations

import dataclasses
import math
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import event_codec

# note_seq constants the reference imports (note_seq/constants.py, note-seq 0.0.3: MIDI ranges)
MIN_MIDI_PITCH, MAX_MIDI_PITCH = 0, 127
MIN_MIDI_PROGRAM, MAX_MIDI_PROGRAM = 0, 127
MAX_MIDI_VELOCITY = 127

DECODED_EOS_ID = -1          # vocabularies.py:27-28
DECODED_INVALID_ID = -2
DEFAULT_EXTRA_IDS = 100      # t5.data.DEFAULT_EXTRA_IDS (vocabularies.py:137)


@dataclasses.dataclass
class VocabularyConfig:      # vocabularies.py:36-52
  steps_per_second: int = 100
  max_shift_seconds: int = 10
  num_velocity_bins: int = 127

  @property
  def abbrev_str(self) -> str:
    s = ''
    if self.steps_per_second != 100:
      s += 'ss%d' % self.steps_per_second
    if self.max_shift_seconds != 10:
      s += 'ms%d' % self.max_shift_seconds
    if self.num_velocity_bins != 127:
      s += 'vb%d' % self.num_velocity_bins
    return s


def num_velocity_bins_from_codec(codec: event_codec.Codec) -> int:     # vocabularies.py:55-58
  lo, hi = codec.event_type_range('velocity')
  return hi - lo


def velocity_to_bin(velocity: int, num_velocity_bins: int) -> int:     # vocabularies.py:61-65
  return 0 if velocity == 0 else math.ceil(num_velocity_bins * velocity / MAX_MIDI_VELOCITY)


def bin_to_velocity(velocity_bin: int, num_velocity_bins: int) -> int:  # vocabularies.py:68-72
  return 0 if velocity_bin == 0 else int(MAX_MIDI_VELOCITY * velocity_bin / num_velocity_bins)


def drop_programs(tokens, codec: event_codec.Codec):                   # vocabularies.py:75-78
  tokens = np.asarray(tokens)
  lo, hi = codec.event_type_range('program')
  return tokens[(tokens < lo) | (tokens > hi)]


def programs_to_midi_classes(tokens, codec: event_codec.Codec):        # vocabularies.py:81-89
  tokens = np.asarray(tokens)
  lo, hi = codec.event_type_range('program')
  is_program = (tokens >= lo) & (tokens <= hi)
  return np.where(is_program, lo + 8 * ((tokens - lo) // 8), tokens)


@dataclasses.dataclass
class ProgramGranularity:    # vocabularies.py:92-96
  tokens_map_fn: Callable
  program_map_fn: Callable[[int], int]


PROGRAM_GRANULARITIES = {    # vocabularies.py:99-116
    'flat': ProgramGranularity(drop_programs, lambda program: 0),
    'midi_class': ProgramGranularity(programs_to_midi_classes, lambda program: 8 * (program // 8)),
    'full': ProgramGranularity(lambda tokens, codec: np.asarray(tokens), lambda program: program),
}


def build_codec(vocab_config: VocabularyConfig) -> event_codec.Codec:  # vocabularies.py:119-141
  ranges = [
      event_codec.EventRange('pitch', MIN_MIDI_PITCH, MAX_MIDI_PITCH),
      event_codec.EventRange('velocity', 0, vocab_config.num_velocity_bins),   # bin 0 = note-off
      event_codec.EventRange('tie', 0, 0),
      event_codec.EventRange('program', MIN_MIDI_PROGRAM, MAX_MIDI_PROGRAM),
      event_codec.EventRange('drum', MIN_MIDI_PITCH, MAX_MIDI_PITCH),
  ]
  return event_codec.Codec(
      max_shift_steps=vocab_config.steps_per_second * vocab_config.max_shift_seconds,
      steps_per_second=vocab_config.steps_per_second, event_ranges=ranges)


class GenericTokenVocabulary:
  """Pass-through vocabulary: ids 0/1/2 = PAD/EOS/UNK, regular token t -> t + 3, then `extra_ids`
  sentinel ids on top (vocabularies.py:149-268; the seqio.Vocabulary base contributes
  vocab_size = base size + extra_ids)."""

  def __init__(self, regular_ids: int, extra_ids: int = 0):
    self._num_special_tokens = 3
    self._num_regular_tokens = regular_ids
    self.extra_ids = extra_ids

  pad_id, eos_id, unk_id = 0, 1, 2

  @property
  def _base_vocab_size(self) -> int:
    return self._num_special_tokens + self._num_regular_tokens

  @property
  def vocab_size(self) -> int:
    return self._base_vocab_size + self.extra_ids

  def encode(self, token_ids: Sequence[int]) -> List[int]:              # vocabularies.py:176-197
    out = []
    for t in token_ids:
      if not 0 <= t < self._num_regular_tokens:
        raise ValueError('token_id %s does not fall within valid range of [0, %d)'
                         % (t, self._num_regular_tokens))
      out.append(int(t) + self._num_special_tokens)
    return out

  def encode_array(self, token_ids) -> np.ndarray:
    """Array form (the reference's _encode_tf, vocabularies.py:224-241): dtype preserved."""
    a = np.asarray(token_ids)
    if a.size and (a.min() < 0 or a.max() >= self._num_regular_tokens):
      raise ValueError('token ids outside [0, %d)' % self._num_regular_tokens)
    return a + a.dtype.type(self._num_special_tokens) if a.dtype.kind in 'iu' else a + self._num_special_tokens

  def _decode_one(self, i: int) -> int:
    if i == self.eos_id:
      return DECODED_EOS_ID
    if i < self._num_special_tokens or i >= self._base_vocab_size:
      return DECODED_INVALID_ID
    return i - self._num_special_tokens

  def decode(self, ids: Sequence[int]) -> List[int]:
    """Python decode: clipped after the first EOS (seqio.Vocabulary.decode) then mapped
    (vocabularies.py:199-222; vocabularies_test.py:76-81)."""
    ids = [int(i) for i in ids]
    if self.eos_id in ids:
      ids = ids[:ids.index(self.eos_id) + 1]
    return [self._decode_one(i) for i in ids]

  def decode_array(self, ids) -> np.ndarray:
    """Length-preserving decode (the reference's _decode_tf, vocabularies.py:243-268): EOS and
    everything after it -> DECODED_EOS_ID."""
    ids = np.asarray(ids)
    after = np.cumsum(ids == self.eos_id, axis=-1) > 0
    ok = (ids >= self._num_special_tokens) & (ids < self._base_vocab_size)
    return np.where(after, DECODED_EOS_ID, np.where(ok, ids - self._num_special_tokens, DECODED_INVALID_ID))

  def __eq__(self, other):                                               # vocabularies.py:270-275
    return (isinstance(other, GenericTokenVocabulary) and self.extra_ids == other.extra_ids and
            self._num_regular_tokens == other._num_regular_tokens)


def vocabulary_from_codec(codec: event_codec.Codec) -> GenericTokenVocabulary:   # vocabularies.py:144-146
  return GenericTokenVocabulary(codec.num_classes, extra_ids=DEFAULT_EXTRA_IDS)


def num_embeddings(vocabulary: GenericTokenVocabulary) -> int:           # vocabularies.py:278-281
  return 128 * math.ceil(vocabulary.vocab_size / 128)
