"""tests/golden/ref_frontend.npz: the tokenisation chain of the REFERENCE, executed by the reference's own code.

note_sequences.py, run_length_encoding.py, event_codec.py and vocabularies.py are imported from /root/reference
(build container only).  None of their dependencies is installed, so stand-ins are registered first:
  note_seq     a NoteSequence with the few members those files touch (notes.add / extend / del [:], CopyFrom,
               total_time, ticks_per_quarter) and the MIDI range constants;
  tensorflow   the eight names run_length_encode_shifts / extract_sequence_with_indices use, over NumPy
               (tf.function and autograph are identities: the function bodies are plain Python loops);
  seqio, t5    map_over_dataset = identity, an empty Vocabulary base class; absl.logging.
For every seeded case of tests/ref_frontend_cases.py:
  note arrays -> note_arrays_to_note_sequence -> (trim_overlapping_notes) -> note_sequence_to_onsets_and_offsets_and_programs
  -> encode_and_index_events (note_event_data_to_events, note_encoding_state_to_events)
  -> per 256-frame segment: extract_sequence_with_indices (tie token) -> run_length_encode_shifts (velocity, program)
  -> GenericTokenVocabulary._encode
  -> and back: NoteEncodingWithTiesSpec decoding of the segment tokens -> flush -> notes
and everything along the way is stored.  tests/test_ref_frontend.py holds the package's frontend/ to it bit for bit.

    python tests/golden/make_ref_frontend_golden.py
"""
import copy
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from tests import ref_frontend_cases as cases  # noqa: E402


def install_standins():
  # ---- note_seq
  class Note:
    def __init__(self, start_time=0.0, end_time=0.0, pitch=0, velocity=0, program=0, is_drum=False, instrument=0):
      self.start_time, self.end_time, self.pitch, self.velocity = start_time, end_time, pitch, velocity
      self.program, self.is_drum, self.instrument = program, is_drum, instrument

  class Notes(list):
    def add(self, **kw):
      n = Note(**kw)
      self.append(n)
      return n

  class NoteSequence:
    def __init__(self, ticks_per_quarter=0):
      self.notes, self.total_time, self.ticks_per_quarter = Notes(), 0.0, ticks_per_quarter

    def CopyFrom(self, other):
      self.notes = Notes(copy.copy(n) for n in other.notes)
      self.total_time, self.ticks_per_quarter = other.total_time, other.ticks_per_quarter
  NoteSequence.Note = Note
  ns = types.ModuleType('note_seq')
  ns.NoteSequence = NoteSequence
  ns.MIN_MIDI_PITCH, ns.MAX_MIDI_PITCH = 0, 127
  ns.MIN_MIDI_PROGRAM, ns.MAX_MIDI_PROGRAM = 0, 127
  ns.MIN_MIDI_VELOCITY, ns.MAX_MIDI_VELOCITY = 1, 127
  # ---- tensorflow (the names run_length_encoding.py uses)
  tf = types.ModuleType('tensorflow')
  tf.int32 = np.int32
  tf.Tensor = np.ndarray        # annotations of the (unused) *_tf methods of GenericTokenVocabulary
  tf.function = lambda f: f
  tf.constant = lambda v, dtype=None: np.array(v, dtype=dtype)
  tf.zeros = lambda n, dtype=None: np.zeros(n, dtype=dtype)
  tf.concat = lambda parts, axis=0: np.concatenate([np.asarray(p) for p in parts], axis=axis)
  tf.minimum = lambda a, b: min(a, b)
  tf.TensorShape = lambda dims: tuple(dims)

  def tensor_scatter_nd_update(t, indices, updates):
    out = np.array(t, copy=True)
    for (i,), u in zip(indices, updates):
      out[i] = u
    return out
  tf.tensor_scatter_nd_update = tensor_scatter_nd_update
  tf.autograph = types.SimpleNamespace(experimental=types.SimpleNamespace(set_loop_options=lambda **kw: None))
  # ---- seqio / t5 / absl
  seqio = types.ModuleType('seqio')
  seqio.map_over_dataset = lambda f=None, **kw: f

  class Vocabulary:   # seqio.Vocabulary as published: extra_ids kept, vocab_size = _base_vocab_size + extra_ids
    def __init__(self, extra_ids=0):
      self._extra_ids = extra_ids

    @property
    def extra_ids(self):
      return self._extra_ids

    @property
    def vocab_size(self):
      return self._base_vocab_size + self._extra_ids
  seqio.Vocabulary = Vocabulary
  t5 = types.ModuleType('t5')
  t5.data = types.ModuleType('t5.data')
  t5.data.DEFAULT_EXTRA_IDS = 100        # the constant t5 publishes
  absl = types.ModuleType('absl')
  absl.logging = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
  sys.modules.update({'note_seq': ns, 'tensorflow': tf, 'seqio': seqio, 't5': t5, 't5.data': t5.data,
                      'absl': absl, 'absl.logging': absl.logging})


def ragged(rows, dtype=np.int64):
  rows = [np.asarray(r, dtype) for r in rows]
  off = np.cumsum([0] + [len(r) for r in rows])
  return (np.concatenate(rows) if rows else np.zeros(0, dtype)), off.astype(np.int64)


def main():
  install_standins()
  pkg = types.ModuleType('music_spectrogram_diffusion')
  pkg.__path__ = [os.path.join(ref_shim.REFERENCE_ROOT, 'music_spectrogram_diffusion')]
  sys.modules['music_spectrogram_diffusion'] = pkg
  ec = importlib.import_module('music_spectrogram_diffusion.event_codec')
  rle = importlib.import_module('music_spectrogram_diffusion.run_length_encoding')
  voc = importlib.import_module('music_spectrogram_diffusion.vocabularies')
  nsq = importlib.import_module('music_spectrogram_diffusion.note_sequences')

  out = {k: [] for k in ('events', 'start', 'end', 'state_events', 'state_idx', 'times', 'val_pitch', 'val_velocity',
                         'val_program', 'val_drum', 'trim_start', 'trim_end', 'trim_pitch', 'instrument',
                         'seg_tokens', 'seg_vocab_ids', 'dec_start', 'dec_end', 'dec_pitch', 'dec_velocity', 'dec_program',
                         'dec_drum', 'dec_instrument')}
  seg_case, meta = [], []
  for i in range(cases.N_CASES):
    c = cases.case(i)
    codec = voc.build_codec(voc.VocabularyConfig(num_velocity_bins=c['num_velocity_bins']))
    vocab = voc.vocabulary_from_codec(codec)
    ns = nsq.note_arrays_to_note_sequence(
        onset_times=c['onsets'].tolist(), pitches=c['pitches'].tolist(), offset_times=c['offsets'].tolist(),
        velocities=c['velocities'].tolist(), programs=c['programs'].tolist(), is_drums=c['is_drums'].tolist())
    out['instrument'].append([n.instrument for n in ns.notes])
    trimmed = nsq.trim_overlapping_notes(ns)
    out['trim_start'].append(np.array([n.start_time for n in trimmed.notes], np.float64))
    out['trim_end'].append(np.array([n.end_time for n in trimmed.notes], np.float64))
    out['trim_pitch'].append([n.pitch for n in trimmed.notes])
    nsq.validate_note_sequence(trimmed)
    times, values = nsq.note_sequence_to_onsets_and_offsets_and_programs(trimmed)
    out['times'].append(np.array(times, np.float64))
    out['val_pitch'].append([v.pitch for v in values])
    out['val_velocity'].append([v.velocity for v in values])
    out['val_program'].append([v.program for v in values])
    out['val_drum'].append([int(bool(v.is_drum)) for v in values])
    frame_times = np.arange(c['n_frames']) / cases.FRAME_RATE
    events, start, end, state_events, state_idx = rle.encode_and_index_events(
        state=nsq.NoteEncodingState(), event_times=times, event_values=values,
        encode_event_fn=nsq.note_event_data_to_events, codec=codec, frame_times=frame_times,
        encoding_state_to_events_fn=nsq.note_encoding_state_to_events)
    for k, v in (('events', events), ('start', start), ('end', end), ('state_events', state_events), ('state_idx', state_idx)):
      out[k].append(v)
    tie = codec.encode_event(ec.Event('tie', 0))
    encode_shifts = rle.run_length_encode_shifts_fn(codec, state_change_event_types=['velocity', 'program'])
    spec = nsq.NoteEncodingWithTiesSpec
    dstate = spec.init_decoding_state_fn()
    invalid = dropped = 0
    for f0 in range(0, c['n_frames'], cases.SEGMENT_FRAMES):
      f1 = min(f0 + cases.SEGMENT_FRAMES, c['n_frames'])
      feats = {'targets': events, 'event_start_indices': start[f0:f1], 'event_end_indices': end[f0:f1],
               'state_events': state_events, 'state_event_indices': state_idx[f0:f1]}
      feats = rle.extract_sequence_with_indices(feats, state_events_end_token=tie)
      feats = encode_shifts(feats)
      toks = np.asarray(feats['targets'], np.int64)
      out['seg_tokens'].append(toks)
      out['seg_vocab_ids'].append(np.asarray(vocab._encode(toks.tolist()), np.int64))
      seg_case.append(i)
      # and back: the decoder of the reference on the tokens just made (note_sequences.py:301-408)
      spec.begin_decoding_segment_fn(dstate)
      a, b = rle.decode_events(dstate, toks, start_time=f0 / cases.FRAME_RATE, max_time=None, codec=codec,
                               decode_event_fn=spec.decode_event_fn)
      invalid, dropped = invalid + a, dropped + b
    dec = spec.flush_decoding_state_fn(dstate)
    out['dec_start'].append(np.array([n.start_time for n in dec.notes], np.float64))
    out['dec_end'].append(np.array([n.end_time for n in dec.notes], np.float64))
    for k, f in (('dec_pitch', lambda n: n.pitch), ('dec_velocity', lambda n: n.velocity), ('dec_program', lambda n: n.program),
                 ('dec_drum', lambda n: int(bool(n.is_drum))), ('dec_instrument', lambda n: n.instrument)):
      out[k].append([f(n) for n in dec.notes])
    meta.append((codec.num_classes, vocab._base_vocab_size, voc.num_embeddings(vocab), tie, invalid, dropped))
  save = {}
  for k, rows in out.items():
    dtype = np.float64 if k in ('times', 'trim_start', 'trim_end', 'dec_start', 'dec_end') else np.int64
    save[k], save[k + '_off'] = ragged(rows, dtype)
  save['seg_case'] = np.array(seg_case, np.int64)
  save['meta'] = np.array(meta, np.int64)     # per case: codec classes, base vocabulary size, num_embeddings, tie token, invalid / dropped events of the decode
  np.savez_compressed(os.path.join(HERE, 'ref_frontend.npz'), **save)
  print('%d cases, %d segments, %d events, %d tokens' % (cases.N_CASES, len(seg_case), len(save['events']), len(save['seg_tokens'])))


if __name__ == '__main__':
  main()
