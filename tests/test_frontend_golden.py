"""SURVEY.md 8(f) N1 -- the MIDI/NoteSequence front end against the golden vectors of the reference's
OWN tests (restated here; integer results must match exactly):
  event_codec_test.py:26-52, vocabularies_test.py:27-109, run_length_encoding_test.py:45-87,
  note_sequences_test.py:41-504.  CPU only."""
import numpy as np
import pytest

import msd_amd  # noqa: F401  (registers the dashed package)
from msd_amd.frontend import event_codec, note_sequences, run_length_encoding, vocabularies

Event, EventRange = event_codec.Event, event_codec.EventRange

# the codec of run_length_encoding_test.py:25-37 / note_sequences_test.py:24-36
CODEC = event_codec.Codec(
    max_shift_steps=100, steps_per_second=100,
    event_ranges=[EventRange('pitch', 0, 127), EventRange('velocity', 0, 127), EventRange('drum', 0, 127),
                  EventRange('program', 0, 127), EventRange('tie', 0, 0)])


# ---- event_codec_test.py ----------------------------------------------------------------------
def test_codec_encode_decode():
  ec = event_codec.Codec(100, 100, [EventRange('pitch', 0, 127)])
  events = [Event('pitch', 60), Event('shift', 5), Event('pitch', 62)]
  enc = [ec.encode_event(e) for e in events]
  assert enc == [161, 5, 163]
  assert [ec.decode_event_index(i) for i in enc] == events


def test_codec_shift_steps():
  ec = event_codec.Codec(100, 100, [EventRange('pitch', 0, 127)])
  assert ec.max_shift_steps == 100
  assert [ec.is_shift_event_index(i) for i in (-1, 0, 100, 101)] == [False, True, True, False]
  with pytest.raises(ValueError):
    ec.encode_event(Event('pitch', 128))
  with pytest.raises(ValueError):
    ec.encode_event(Event('nope', 0))
  with pytest.raises(ValueError):
    ec.decode_event_index(ec.num_classes)


# ---- vocabularies_test.py ---------------------------------------------------------------------
def test_velocity_quantization():
  v = vocabularies
  assert v.velocity_to_bin(0, 1) == 0 and v.velocity_to_bin(0, 127) == 0
  assert v.bin_to_velocity(0, 1) == 0 and v.bin_to_velocity(0, 127) == 0
  assert v.velocity_to_bin(v.bin_to_velocity(1, 1), 1) == 1
  for b in range(1, 128):
    assert v.velocity_to_bin(v.bin_to_velocity(b, 127), 127) == b


def test_vocab_encode_decode():
  vocab = vocabularies.GenericTokenVocabulary(32)
  assert vocab.encode([1, 2, 3]) == [4, 5, 6]
  np.testing.assert_array_equal(vocab.encode_array(np.array([1, 2, 3])), [4, 5, 6])
  assert vocab.decode([4, 5, 6]) == [1, 2, 3]
  np.testing.assert_array_equal(vocab.decode_array(np.array([4, 5, 6])), [1, 2, 3])


def test_vocab_decode_invalid_ids():
  vocab = vocabularies.GenericTokenVocabulary(32, extra_ids=4)
  enc = [0, 2, 3, 4, 34, 35]
  want = [-2, -2, 0, 1, 31, -2]
  assert vocab.decode(enc) == want
  np.testing.assert_array_equal(vocab.decode_array(np.array(enc)), want)


def test_vocab_decode_eos():
  vocab = vocabularies.GenericTokenVocabulary(32)
  enc = [0, 2, 3, 4, 1, 0, 1, 0]
  assert vocab.decode(enc) == [-2, -2, 0, 1, -1]                       # python decode stops at EOS
  np.testing.assert_array_equal(vocab.decode_array(np.array(enc)), [-2, -2, 0, 1, -1, -1, -1, -1])


def test_vocab_encode_invalid_id_and_dtypes():
  vocab = vocabularies.GenericTokenVocabulary(32)
  vocab.encode([0, 15, 31])
  vocab.encode_array(np.array([0, 15, 31]))
  for bad in ([-1, 15, 31], [0, 15, 32]):
    with pytest.raises(ValueError):
      vocab.encode(bad)
    with pytest.raises(ValueError):
      vocab.encode_array(np.array(bad))
  assert vocab.encode_array(np.array([0, 15, 31], np.int32)).dtype == np.int32
  assert vocab.encode_array(np.array([0, 15, 31], np.int64)).dtype == np.int64


def test_mt3_vocabulary_size():
  """1388 codec classes + 3 special + 100 extra ids -> 1491 -> 1536 embeddings."""
  codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
  assert codec.num_classes == 1388
  vocab = vocabularies.vocabulary_from_codec(codec)
  assert vocab.vocab_size == 1491 and vocabularies.num_embeddings(vocab) == 1536
  assert msd_amd.config.preset('base_with_context').t5.vocab_size == 1536


# ---- run_length_encoding_test.py --------------------------------------------------------------
@pytest.mark.parametrize('src,want,state_types', [
    ([1, 1, 1, 161, 1, 1, 1, 162, 1, 1, 1], [3, 161, 6, 162], ()),
    ([1] * 202 + [161, 1, 1, 1], [100, 100, 2, 161], ()),
    ([1, 1, 1, 161, 162, 1, 1, 1], [3, 161, 162], ()),
    ([1, 1, 1, 525, 356, 161, 1, 1, 525, 356, 161, 355, 394], [3, 525, 356, 161, 5, 161, 355, 394],
     ('velocity', 'program')),
])
def test_run_length_encode_shifts(src, want, state_types):
  fn = run_length_encoding.run_length_encode_shifts_fn(CODEC, state_change_event_types=state_types)
  out = fn({'targets': np.array(src)})['targets']
  np.testing.assert_array_equal(out, want)


# ---- note_sequences_test.py: encode + index ----------------------------------------------------
def _ns(notes):
  ns = note_sequences.NoteSequence()
  for kw in notes:
    ns.add_note(**kw)
  ns.total_time = ns.notes[-1].end_time
  return ns


def test_encode_and_index_onsets():
  ns = _ns([dict(start_time=1.0, end_time=1.1, pitch=61, velocity=100),
            dict(start_time=2.0, end_time=2.1, pitch=62, velocity=100),
            dict(start_time=3.0, end_time=3.1, pitch=63, velocity=100)])
  frame_times = np.arange(0, 4, step=.001)
  t, v = note_sequences.note_sequence_to_onsets(ns)
  ev, si, ei, _, _ = run_length_encoding.encode_and_index_events(
      None, t, v, note_sequences.note_event_data_to_events, CODEC, frame_times)
  assert len(si) == len(ei) == len(frame_times) and len(ev) == 403
  np.testing.assert_array_equal(ev, [1] * 100 + [162] + [1] * 100 + [163] + [1] * 100 + [164] + [1] * 100)
  assert (si[0], ei[0]) == (0, 0)
  assert frame_times[1000] == 1.0 and (si[1000], ei[1000]) == (100, 100)
  assert frame_times[2000] == 2.0 and (si[2000], ei[2000]) == (201, 201)
  assert frame_times[3000] == 3.0 and (si[3000], ei[3000]) == (302, 302)
  assert ev[-1] == 1 and frame_times[-1] == 3.999 and si[-1] == 402 and ei[-1] == 403


def test_encode_and_index_velocity():
  ns = _ns([dict(start_time=1.0, end_time=3.0, pitch=61, velocity=1),
            dict(start_time=2.0, end_time=4.0, pitch=62, velocity=127)])
  frame_times = np.arange(0, 4, step=.001)
  t, v = note_sequences.note_sequence_to_onsets_and_offsets(ns)
  ev, si, ei, _, _ = run_length_encoding.encode_and_index_events(
      None, t, v, note_sequences.note_event_data_to_events, CODEC, frame_times)
  assert len(ev) == 408
  np.testing.assert_array_equal(
      ev, [1] * 100 + [230, 162] + [1] * 100 + [356, 163] + [1] * 100 + [229, 162] + [1] * 100 + [229, 163])
  assert (si[1000], ei[1000]) == (100, 100)
  assert (si[2000], ei[2000]) == (202, 202)
  assert (si[3000], ei[3000]) == (304, 304)
  assert si[-1] == 405 and ei[-1] == 408


def test_encode_and_index_multitrack_with_ties():
  ns = _ns([dict(start_time=0.0, end_time=1.0, pitch=37, velocity=127, is_drum=True),
            dict(start_time=1.0, end_time=3.0, pitch=61, velocity=127, program=0),
            dict(start_time=2.0, end_time=4.0, pitch=62, velocity=127, program=40)])
  frame_times = np.arange(0, 4, step=.001)
  t, v = note_sequences.note_sequence_to_onsets_and_offsets_and_programs(ns)
  tok, si, ei, st, sti = run_length_encoding.encode_and_index_events(
      note_sequences.NoteEncodingState(), t, v, note_sequences.note_event_data_to_events, CODEC, frame_times,
      encoding_state_to_events_fn=note_sequences.note_encoding_state_to_events)
  assert len(si) == len(ei) == len(sti) == len(frame_times) and len(tok) == 414
  E = Event
  want = ([E('velocity', 127), E('drum', 37)] + [E('shift', 1)] * 100 +
          [E('program', 0), E('velocity', 127), E('pitch', 61)] + [E('shift', 1)] * 100 +
          [E('program', 40), E('velocity', 127), E('pitch', 62)] + [E('shift', 1)] * 100 +
          [E('program', 0), E('velocity', 0), E('pitch', 61)] + [E('shift', 1)] * 100 +
          [E('program', 40), E('velocity', 0), E('pitch', 62)])
  np.testing.assert_array_equal(tok, [CODEC.encode_event(e) for e in want])
  want_state = [E('tie', 0), E('tie', 0), E('program', 0), E('pitch', 61), E('tie', 0),
                E('program', 0), E('pitch', 61), E('program', 40), E('pitch', 62), E('tie', 0),
                E('program', 40), E('pitch', 62), E('tie', 0)]
  np.testing.assert_array_equal(st, [CODEC.encode_event(e) for e in want_state])
  assert (si[0], ei[0], sti[0]) == (0, 0, 0)
  assert (si[1000], ei[1000], sti[1000]) == (102, 102, 1)
  assert (si[2000], ei[2000], sti[2000]) == (205, 205, 2)
  assert (si[3000], ei[3000], sti[3000]) == (308, 308, 5)
  assert si[-1] == 410 and ei[-1] == len(want) and sti[-1] == 10


def test_encode_and_index_last_token_alignment():
  ns = _ns([dict(start_time=0.0, end_time=0.1, pitch=60, velocity=100)])
  frame_times = np.arange(0, 1.008, step=.008)
  t, v = note_sequences.note_sequence_to_onsets(ns)
  ev, si, ei, _, _ = run_length_encoding.encode_and_index_events(
      None, t, v, note_sequences.note_event_data_to_events, CODEC, frame_times)
  assert len(si) == len(ei) == len(frame_times) and len(ev) == 102
  np.testing.assert_array_equal(ev, [161] + [1] * 101)
  assert (si[0], ei[0]) == (0, 0) and (si[125], ei[125]) == (101, 102)


# ---- note_sequences_test.py: decode ---------------------------------------------------------------
def _decode(events, fn, start_time=0, max_time=None):
  state = note_sequences.NoteDecodingState()
  invalid, dropped = run_length_encoding.decode_events(state, events, start_time, max_time, CODEC, fn)
  ns = note_sequences.flush_note_decoding_state(state)
  notes = [(n.pitch, n.velocity, round(n.start_time, 9), round(n.end_time, 9), n.program, n.is_drum, n.instrument)
           for n in ns.notes]
  return invalid, dropped, notes, round(ns.total_time, 9)


ONSET = note_sequences.decode_note_onset_event
NOTE = note_sequences.decode_note_event


@pytest.mark.parametrize('events,fn,kw,want', [
    ([25, 161, 50, 162], ONSET, {}, (0, 0, [(60, 100, 0.25, 0.26, 0, False, 0), (61, 100, 0.50, 0.51, 0, False, 0)], 0.51)),
    ([5, 161, 25, 162], ONSET, {}, (0, 0, [(60, 100, 0.05, 0.06, 0, False, 0), (61, 100, 0.25, 0.26, 0, False, 0)], 0.26)),
    ([5, 356, 161, 25, 229, 161], NOTE, {}, (0, 0, [(60, 127, 0.05, 0.25, 0, False, 0)], 0.25)),
    ([5, 356, 161, 10, 161, 25, 229, 161], NOTE, {},
     (0, 0, [(60, 127, 0.05, 0.10, 0, False, 0), (60, 127, 0.10, 0.25, 0, False, 0)], 0.25)),
    ([5, 525, 356, 161, 15, 356, 394, 25, 525, 229, 161], NOTE, {},
     (0, 0, [(37, 127, 0.15, 0.16, 0, True, 9), (60, 127, 0.05, 0.25, 40, False, 0)], 0.25)),
    ([5, -1, 161, -2, 25, 162, 9999], ONSET, {},
     (3, 0, [(60, 100, 0.05, 0.06, 0, False, 0), (61, 100, 0.25, 0.26, 0, False, 0)], 0.26)),
    ([161, 25, 162], ONSET, dict(start_time=1.0, max_time=1.25),
     (0, 0, [(60, 100, 1.00, 1.01, 0, False, 0), (61, 100, 1.25, 1.26, 0, False, 0)], 1.26)),
    ([5, 161, 30, 162], ONSET, dict(start_time=1.0, max_time=1.25), (0, 2, [(60, 100, 1.05, 1.06, 0, False, 0)], 1.06)),
    ([25, 230, 50, 161], ONSET, {}, (1, 0, [(60, 100, 0.50, 0.51, 0, False, 0)], 0.51)),
])
def test_decode_note_sequence_events(events, fn, kw, want):
  assert _decode(events, fn, **kw) == want
