"""NoteSequence / MIDI file -> the `encoder_input_tokens` of every 5.12 s segment of a song, i.e. the
reference's full-song synthesis input pipeline (tasks.py:405-464 with full_song_eval=True) as plain
functions:

  tokenize_transcription_example   preprocessors.py:101-197   sustain, note events, unit-shift stream,
                                                              per-frame indices, state (tie) events
  rekey                            tasks.py:88-101            events become 'inputs', frames 'targets'
  split_full_song                  preprocessors.py:863-921   consecutive segments of 256 frames
                                   (+ t5.data.preprocessors.split_tokens: the last segment keeps its
                                   true, shorter length)
  extract_sequence_with_indices    run_length_encoding.py:179 tie section + events of the segment
  map_midi_programs                preprocessors.py:735-748
  run_length_encode_shifts         run_length_encoding.py:208 (state changes: velocity, program)
  handle_too_long                  preprocessors.py:699-732   more than inputs_length-1 tokens is an error
  tokenize_and_append_eos          seqio: vocabulary.encode (+3), EOS = 1
  feature converter                feature_converters.py: right-pad with 0 to inputs_length

Defaults are the MT3 task the shipped models use (gin/tasks/mt3/base.gin: 1 velocity bin, 'full'
programs; gin/tasks/base.gin: ties on; gin/tasks/mt3/context_mega.gin: inputs 2048, targets 256;
MelGAN codec: 16 kHz, hop 320 -> 50 frames/s, audio_codecs.py:204-218)."""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np

from . import event_codec, midi_io, note_sequences, run_length_encoding, vocabularies


@dataclasses.dataclass
class FrontendConfig:
  sample_rate: int = 16000
  hop_size: int = 320
  segment_frames: int = 256           # TASK_FEATURE_LENGTHS['targets']
  inputs_length: int = 2048           # TASK_FEATURE_LENGTHS['inputs']
  vocab: vocabularies.VocabularyConfig = dataclasses.field(
      default_factory=lambda: vocabularies.VocabularyConfig(num_velocity_bins=1))
  include_ties: bool = True
  onsets_only: bool = False
  program_granularity: str = 'full'
  additional_frames_for_encoding: int = 16  # MelGAN (audio_codecs.py:216-221): widens the FRAME slice only, not the tokens

  @property
  def frame_rate(self) -> float:
    return self.sample_rate / self.hop_size

  @classmethod
  def from_spec(cls, spec) -> 'FrontendConfig':
    """Lengths from a msd_amd ModelSpec (task_feature_lengths)."""
    lengths = spec.task_feature_lengths
    return cls(segment_frames=int(lengths['targets']), inputs_length=int(lengths['inputs']))


def audio_frame_times(num_samples: int, hop_size: int, frame_rate: float) -> np.ndarray:
  """preprocessors.py:60-81: samples are padded by hop - n % hop (a FULL extra frame when n is a
  multiple of hop) and cut into non-overlapping frames; frame k starts at k / frame_rate."""
  padded = num_samples + (hop_size - num_samples % hop_size)
  return np.arange(padded // hop_size) / frame_rate


@dataclasses.dataclass
class TokenizedSong:
  """Output of tokenize_transcription_example after the rekey (events = 'inputs')."""
  events: np.ndarray
  event_start_indices: np.ndarray
  event_end_indices: np.ndarray
  state_events: np.ndarray
  state_event_indices: np.ndarray
  frame_times: np.ndarray
  note_sequence: note_sequences.NoteSequence


def tokenize_note_sequence(ns: note_sequences.NoteSequence, config: FrontendConfig, codec: event_codec.Codec,
                           num_samples: Optional[int] = None) -> TokenizedSong:
  """preprocessors.py:141-197.  `num_samples`: length of the audio the frames are cut from; for
  MIDI-only synthesis there is no audio and the song length is used (floor(total_time * rate))."""
  if config.onsets_only and config.include_ties:
    raise ValueError('Ties not supported when only modeling onsets.')
  if not ns.notes:
    raise ValueError('no notes: the tie section of a segment is undefined for an empty sequence')
  note_sequences.validate_note_sequence(ns)
  if num_samples is None:
    num_samples = int(ns.total_time * config.sample_rate)
  frame_times = audio_frame_times(num_samples, config.hop_size, config.frame_rate)
  if config.onsets_only:
    times, values = note_sequences.note_sequence_to_onsets(ns)
  else:
    ns = midi_io.apply_sustain_control_changes(ns)
    times, values = note_sequences.note_sequence_to_onsets_and_offsets_and_programs(ns)
  ns.control_changes = []
  ev, start, end, st_ev, st_idx = run_length_encoding.encode_and_index_events(
      state=note_sequences.NoteEncodingState() if config.include_ties else None,
      event_times=times, event_values=values, encode_event_fn=note_sequences.note_event_data_to_events,
      codec=codec, frame_times=frame_times,
      encoding_state_to_events_fn=note_sequences.note_encoding_state_to_events if config.include_ties else None)
  return TokenizedSong(ev, start, end, st_ev, st_idx, frame_times, ns)


def split_full_song(song: TokenizedSong, config: FrontendConfig) -> List[dict]:
  """preprocessors.py:863-921: frames are cut into consecutive runs of segment_frames (the last one
  shorter), the per-frame index arrays with them; the event arrays pass through whole."""
  n = len(song.frame_times)
  out = []
  for lo in range(0, n, config.segment_frames):
    hi = min(lo + config.segment_frames, n)
    out.append({
        'frame_range': (lo, hi + config.additional_frames_for_encoding),
        'inputs': song.events,
        'state_events': song.state_events,
        'event_start_indices': song.event_start_indices[lo:hi],
        'event_end_indices': song.event_end_indices[lo:hi],
        'state_event_indices': song.state_event_indices[lo:hi],
    })
  return out


def segment_to_model_tokens(segment: dict, config: FrontendConfig, codec: event_codec.Codec,
                            vocabulary: vocabularies.GenericTokenVocabulary, on_too_long: str = 'error') -> np.ndarray:
  """note_representation_processor_chain (tasks.py:147-171) + handle_too_long + EOS + padding ->
  int32 [inputs_length]."""
  tie = codec.encode_event(event_codec.Event('tie', 0)) if config.include_ties else None
  f = run_length_encoding.extract_sequence_with_indices(segment, state_events_end_token=tie, feature_key='inputs')
  f['inputs'] = vocabularies.PROGRAM_GRANULARITIES[config.program_granularity].tokens_map_fn(f['inputs'], codec)
  f = run_length_encoding.run_length_encode_shifts_fn(
      codec, feature_key='inputs', state_change_event_types=['velocity', 'program'])(f)
  toks = np.asarray(f['inputs'], dtype=np.int64)
  limit = config.inputs_length - 1                      # room for EOS (handle_too_long)
  if len(toks) > limit:
    if on_too_long == 'truncate':
      toks = toks[:limit]
    else:
      raise ValueError('Value for "inputs" field exceeds maximum length: %d > %d' % (len(toks), limit))
  ids = vocabulary.encode_array(toks)
  out = np.zeros((config.inputs_length,), dtype=np.int32)
  out[:len(ids)] = ids
  out[len(ids)] = vocabulary.eos_id
  return out


def note_sequence_to_model_inputs(ns: note_sequences.NoteSequence, config: Optional[FrontendConfig] = None,
                                  num_samples: Optional[int] = None, on_too_long: str = 'error') -> List[np.ndarray]:
  """One int32 [1, inputs_length] `encoder_input_tokens` array per segment of the song, ready for
  InferenceModel.predict_sequence."""
  config = config or FrontendConfig()
  codec = vocabularies.build_codec(config.vocab)
  vocabulary = vocabularies.vocabulary_from_codec(codec)
  song = tokenize_note_sequence(ns, config, codec, num_samples)
  return [segment_to_model_tokens(seg, config, codec, vocabulary, on_too_long)[None]
          for seg in split_full_song(song, config)]


def midi_file_to_model_inputs(path: str, config: Optional[FrontendConfig] = None, **kw) -> List[np.ndarray]:
  return note_sequence_to_model_inputs(midi_io.midi_file_to_note_sequence(path), config, **kw)


def decode_model_inputs(segments: Sequence[np.ndarray], config: Optional[FrontendConfig] = None
                        ) -> note_sequences.NoteSequence:
  """Inverse of note_sequence_to_model_inputs (times on the codec's 10 ms grid): the round-trip
  property the tests use.  Per segment: strip EOS/padding, vocabulary.decode, open the tie section,
  decode with start_time = segment index * segment_frames / frame_rate (the reference's
  decode_and_combine_predictions walks segments the same way)."""
  config = config or FrontendConfig()
  codec = vocabularies.build_codec(config.vocab)
  vocabulary = vocabularies.vocabulary_from_codec(codec)
  spec = note_sequences.NoteEncodingWithTiesSpec if config.include_ties else note_sequences.NoteEncodingSpec
  state = spec.init_decoding_state_fn()
  seg_seconds = config.segment_frames / config.frame_rate
  for k, seg in enumerate(segments):
    ids = vocabulary.decode(np.asarray(seg).reshape(-1))
    toks = [t for t in ids if t >= 0]
    spec.begin_decoding_segment_fn(state)
    run_length_encoding.decode_events(state, toks, start_time=k * seg_seconds, max_time=None, codec=codec,
                                      decode_event_fn=spec.decode_event_fn)
  return spec.flush_decoding_state_fn(state)
