"""tests/golden/ref_frontend.npz = the reference's OWN tokenisation code (note_sequences.py, run_length_encoding.py,
event_codec.py, vocabularies.py) executed on 48 seeded random note sets over stand-ins of note_seq / tensorflow /
seqio (tests/golden/make_ref_frontend_golden.py).  The package's frontend/ must reproduce every intermediate bit
for bit: instrument assignment, overlap trimming, the (time, value) event list and its ordering, the unit-shift event
stream with its frame indices and state dumps, and -- per 256-frame segment -- the tie-prefixed, run-length encoded
tokens and their vocabulary ids; then the way back (NoteEncodingWithTiesSpec decoding of those tokens: the decoded
notes, invalid / dropped counts); plus codec / vocabulary sizes."""
import os

import numpy as np
import pytest

from msd_amd.frontend import event_codec, note_sequences, run_length_encoding, vocabularies
from tests import ref_frontend_cases as cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_frontend.npz')


@pytest.fixture(scope='module')
def gold():
  return np.load(GOLD)


def _row(g, key, i):
  off = g[key + '_off']
  return g[key][off[i]:off[i + 1]]


def test_frontend_reproduces_the_references_tokenisation(gold):
  g = gold
  seg = 0
  for i in range(cases.N_CASES):
    c = cases.case(i)
    codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=c['num_velocity_bins']))
    vocab = vocabularies.vocabulary_from_codec(codec)
    tie = codec.encode_event(event_codec.Event('tie', 0))
    assert (codec.num_classes, vocab._base_vocab_size, vocabularies.num_embeddings(vocab), tie) == tuple(g['meta'][i][:4])
    ns = note_sequences.note_arrays_to_note_sequence(
        c['onsets'].tolist(), c['pitches'].tolist(), c['offsets'].tolist(), c['velocities'].tolist(),
        c['programs'].tolist(), c['is_drums'].tolist())
    assert [n.instrument for n in ns.notes] == _row(g, 'instrument', i).tolist()
    trimmed = note_sequences.trim_overlapping_notes(ns)
    np.testing.assert_array_equal([n.start_time for n in trimmed.notes], _row(g, 'trim_start', i))
    np.testing.assert_array_equal([n.end_time for n in trimmed.notes], _row(g, 'trim_end', i))
    assert [n.pitch for n in trimmed.notes] == _row(g, 'trim_pitch', i).tolist()
    note_sequences.validate_note_sequence(trimmed)
    times, values = note_sequences.note_sequence_to_onsets_and_offsets_and_programs(trimmed)
    np.testing.assert_array_equal(times, _row(g, 'times', i))
    assert [v.pitch for v in values] == _row(g, 'val_pitch', i).tolist()
    assert [v.velocity for v in values] == _row(g, 'val_velocity', i).tolist()
    assert [v.program for v in values] == _row(g, 'val_program', i).tolist()
    assert [int(bool(v.is_drum)) for v in values] == _row(g, 'val_drum', i).tolist()
    frame_times = np.arange(c['n_frames']) / cases.FRAME_RATE
    events, start, end, state_events, state_idx = run_length_encoding.encode_and_index_events(
        note_sequences.NoteEncodingState(), times, values, note_sequences.note_event_data_to_events, codec,
        frame_times, note_sequences.note_encoding_state_to_events)
    for got, key in ((events, 'events'), (start, 'start'), (end, 'end'), (state_events, 'state_events'),
                     (state_idx, 'state_idx')):
      np.testing.assert_array_equal(np.asarray(got), _row(g, key, i), err_msg='case %d %s' % (i, key))
    encode_shifts = run_length_encoding.run_length_encode_shifts_fn(codec, state_change_event_types=['velocity', 'program'])
    spec = note_sequences.NoteEncodingWithTiesSpec
    dstate = spec.init_decoding_state_fn()
    invalid = dropped = 0
    for f0 in range(0, c['n_frames'], cases.SEGMENT_FRAMES):
      f1 = min(f0 + cases.SEGMENT_FRAMES, c['n_frames'])
      assert int(g['seg_case'][seg]) == i
      feats = {'targets': np.asarray(events), 'event_start_indices': np.asarray(start)[f0:f1],
               'event_end_indices': np.asarray(end)[f0:f1], 'state_events': np.asarray(state_events),
               'state_event_indices': np.asarray(state_idx)[f0:f1]}
      feats = run_length_encoding.extract_sequence_with_indices(feats, state_events_end_token=tie)
      feats = encode_shifts(feats)
      np.testing.assert_array_equal(np.asarray(feats['targets']), _row(g, 'seg_tokens', seg), err_msg='case %d frame %d' % (i, f0))
      np.testing.assert_array_equal(vocab.encode(np.asarray(feats['targets']).tolist()), _row(g, 'seg_vocab_ids', seg))
      # and back, with the package's decoder on the same tokens
      spec.begin_decoding_segment_fn(dstate)
      a, b = run_length_encoding.decode_events(dstate, np.asarray(feats['targets']), f0 / cases.FRAME_RATE, None, codec,
                                               spec.decode_event_fn)
      invalid, dropped = invalid + a, dropped + b
      seg += 1
    dec = spec.flush_decoding_state_fn(dstate)
    assert (invalid, dropped) == tuple(g['meta'][i][4:6])
    np.testing.assert_array_equal([n.start_time for n in dec.notes], _row(g, 'dec_start', i), err_msg='case %d' % i)
    np.testing.assert_array_equal([n.end_time for n in dec.notes], _row(g, 'dec_end', i))
    for key, f in (('dec_pitch', lambda n: n.pitch), ('dec_velocity', lambda n: n.velocity), ('dec_program', lambda n: n.program),
                   ('dec_drum', lambda n: int(bool(n.is_drum))), ('dec_instrument', lambda n: n.instrument)):
      assert [f(n) for n in dec.notes] == _row(g, key, i).tolist(), (i, key)
  assert seg == len(g['seg_case'])


def test_the_cases_exercise_the_interesting_paths(gold):
  """Drums, several programs, 127 velocity bins, overlapping notes that get trimmed, state dumps that are not just
  the tie token, and shifts longer than one max_shift token all occur in the fixture."""
  g = gold
  assert any(_row(g, 'val_drum', i).any() for i in range(cases.N_CASES))
  assert any(len(set(_row(g, 'val_program', i).tolist())) > 2 for i in range(cases.N_CASES))
  assert any(len(_row(g, 'trim_pitch', i)) < len(cases.case(i)['pitches']) or
             not np.array_equal(np.sort(_row(g, 'trim_end', i)), np.sort(cases.case(i)['offsets'])) for i in range(cases.N_CASES))
  assert max(len(_row(g, 'state_events', i)) for i in range(cases.N_CASES)) > 200
  assert len(g['seg_case']) > cases.N_CASES
