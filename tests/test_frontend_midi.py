"""Front end beyond the reference's golden vectors: the MIDI reader, the sustain pass and the
full-song segment pipeline (frontend/midi_io.py, frontend/tokenizer.py).  These have no vectors in the
reference's tests (note_seq / pretty_midi / tf.data are third-party there); they are pinned by
hand-computed cases and by the domain's round trips: MIDI bytes -> notes -> MIDI bytes, and
notes -> segment tokens -> notes.  CPU only."""
import numpy as np
import pytest

import msd_amd
from msd_amd.frontend import midi_io, note_sequences, tokenizer, vocabularies


def _notes(ns):
  return sorted((n.is_drum, n.program, n.pitch, round(n.start_time, 6), round(n.end_time, 6), n.velocity)
                for n in ns.notes)


def test_midi_write_read_round_trip():
  ns = note_sequences.NoteSequence()
  ns.add_note(pitch=60, velocity=80, start_time=0.0, end_time=0.5, program=0)
  ns.add_note(pitch=64, velocity=90, start_time=0.25, end_time=1.0, program=40)
  ns.add_note(pitch=60, velocity=70, start_time=0.5, end_time=0.75, program=0)     # re-struck at the release tick
  ns.add_note(pitch=36, velocity=100, start_time=0.125, end_time=0.25, is_drum=True)
  got = midi_io.parse_midi(midi_io.note_sequence_to_midi(ns, ticks_per_quarter=480, tempo_us=500000))
  assert _notes(got) == _notes(ns)
  assert got.total_time == 1.0 and got.ticks_per_quarter == 480
  drums = [n for n in got.notes if n.is_drum]
  assert len(drums) == 1 and all(n.program == 0 for n in drums)


def test_midi_tempo_map_running_status_and_velocity_zero_off():
  tpq = 100
  track = bytes([
      0x00, 0xFF, 0x51, 0x03, 0x0F, 0x42, 0x40,      # tempo 1,000,000 us/qn -> 10 ms per tick
      0x00, 0xC1, 0x28,                              # ch 1 program 40
      0x00, 0x91, 0x3C, 0x64,                        # tick 0: note on 60
      0x64, 0x3E, 0x50,                              # tick 100 (running status): note on 62
      0x00, 0x3C, 0x00,                              # tick 100: note on 60 vel 0 = off -> 60 lasts 1.0 s
      0x00, 0xFF, 0x51, 0x03, 0x07, 0xA1, 0x20,      # tick 100: tempo 500,000 -> 5 ms per tick
      0x64, 0x81, 0x3E, 0x00,                        # tick 200: note off 62 -> lasts 0.5 s
      0x00, 0xFF, 0x2F, 0x00])
  data = b'MThd' + (6).to_bytes(4, 'big') + (0).to_bytes(2, 'big') + (1).to_bytes(2, 'big') + tpq.to_bytes(2, 'big')
  data += b'MTrk' + len(track).to_bytes(4, 'big') + track
  ns = midi_io.parse_midi(data)
  assert _notes(ns) == [(False, 40, 60, 0.0, 1.0, 100), (False, 40, 62, 1.0, 1.5, 80)]


def test_midi_rejects_garbage():
  with pytest.raises(midi_io.MidiError):
    midi_io.parse_midi(b'RIFF' + b'\0' * 20)
  good = midi_io.note_sequence_to_midi(note_sequences.note_arrays_to_note_sequence([0.0], [60], [0.5]))
  with pytest.raises(midi_io.MidiError):
    midi_io.parse_midi(good[:14] + b'XXXX' + good[18:])


def test_sustain_pedal_semantics():
  ns = note_sequences.NoteSequence()
  a = dict(program=0, instrument=0)
  ns.add_note(pitch=60, velocity=100, start_time=0.0, end_time=0.5, **a)     # released under the pedal
  ns.add_note(pitch=60, velocity=100, start_time=1.0, end_time=1.2, **a)     # re-strike ends the first one
  ns.add_note(pitch=64, velocity=100, start_time=0.2, end_time=3.0, **a)     # outlasts the pedal: unchanged
  ns.add_note(pitch=67, velocity=100, start_time=2.5, end_time=2.6, **a)     # pedal already up: unchanged
  ns.add_note(pitch=36, velocity=100, start_time=0.1, end_time=0.2, is_drum=True, instrument=9)
  ns.add_note(pitch=72, velocity=100, start_time=0.3, end_time=0.4, program=40, instrument=1)  # other instrument
  ns.total_time = 3.0
  CC = note_sequences.ControlChange
  ns.control_changes = [CC(0.1, 64, 127, instrument=0), CC(2.0, 64, 0, instrument=0), CC(0.5, 7, 100, instrument=0)]
  out = midi_io.apply_sustain_control_changes(ns)
  got = {(n.pitch, round(n.start_time, 3)): round(n.end_time, 3) for n in out.notes}
  assert got == {(60, 0.0): 1.0, (60, 1.0): 2.0, (64, 0.2): 3.0, (67, 2.5): 2.6, (36, 0.1): 0.2, (72, 0.3): 0.4}
  assert [n.end_time for n in ns.notes][:2] == [0.5, 1.2]       # input untouched
  # pedal never released: held notes end at the last event time
  ns.control_changes = [CC(0.1, 64, 127, instrument=0)]
  out = midi_io.apply_sustain_control_changes(ns)
  got = {(n.pitch, round(n.start_time, 3)): round(n.end_time, 3) for n in out.notes}
  assert got[(60, 1.0)] == 3.0 and got[(67, 2.5)] == 3.0 and got[(60, 0.0)] == 1.0 and got[(72, 0.3)] == 0.4


def test_frame_times_padding_quirk():
  # preprocessors.py:66-69 pads by hop - n % hop: a multiple of hop gains a whole extra frame
  assert len(tokenizer.audio_frame_times(0, 320, 50.0)) == 1
  assert len(tokenizer.audio_frame_times(319, 320, 50.0)) == 1
  assert len(tokenizer.audio_frame_times(320, 320, 50.0)) == 2
  np.testing.assert_allclose(tokenizer.audio_frame_times(641, 320, 50.0), [0.0, 0.02, 0.04])


def _random_song(seed, seconds=14.0, programs=(0, 40, 73), n=120, drums=True):
  """Notes on the codec's 10 ms grid, no overlap within a (program, pitch) lane."""
  rng = np.random.default_rng(seed)
  ns = note_sequences.NoteSequence()
  busy = {}
  for _ in range(n):
    program = int(rng.choice(programs))
    pitch = int(rng.integers(40, 90))
    start = int(rng.integers(0, int(seconds * 100) - 60))
    dur = int(rng.integers(2, 300))
    lane = busy.setdefault((program, pitch), [])
    if any(s < start + dur + 1 and start < e + 1 for s, e in lane):
      continue
    lane.append((start, start + dur))
    ns.add_note(pitch=pitch, velocity=int(rng.integers(1, 128)), start_time=start / 100, end_time=(start + dur) / 100,
                program=program)
  if drums:
    for k in range(20):
      t = int(rng.integers(0, int(seconds * 100))) / 100
      ns.add_note(pitch=int(rng.choice([36, 38, 42])), velocity=100, start_time=t, end_time=t + 0.05, is_drum=True)
  ns.total_time = max(n.end_time for n in ns.notes)
  note_sequences.assign_instruments(ns)
  return ns


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_song_to_segments_round_trip(seed):
  ns = _random_song(seed)
  cfg = tokenizer.FrontendConfig()
  segs = tokenizer.note_sequence_to_model_inputs(ns, cfg)
  n_frames = len(tokenizer.audio_frame_times(int(ns.total_time * 16000), 320, 50.0))
  assert len(segs) == -(-n_frames // 256)
  codec = vocabularies.build_codec(cfg.vocab)
  tie = codec.encode_event(msd_amd.frontend.event_codec.Event('tie', 0)) + 3
  for k, s in enumerate(segs):
    assert s.shape == (1, 2048) and s.dtype == np.int32
    row = s[0]
    n_valid = int((row > 0).sum())
    assert row[n_valid - 1] == 1 and (row[n_valid:] == 0).all() and (row[:n_valid - 1] >= 3).all()
    assert row.max() < 1536 and tie in row[:n_valid]
  back = tokenizer.decode_model_inputs(segs, cfg)
  def key(seq, drum_len):
    out = []
    for n in seq.notes:
      end = round(n.start_time * 100) + 1 if (n.is_drum and drum_len) else round(n.end_time * 100)
      out.append((n.is_drum, 0 if n.is_drum else n.program, n.pitch, round(n.start_time * 100), end))
    return sorted(out)
  # 1 velocity bin: velocities come back as 127; drums have no offsets (decoded 10 ms long)
  assert key(back, True) == key(ns, True)
  assert all(n.velocity == 127 for n in back.notes)


def test_segment_tokens_hand_computed():
  """5.12 s segments; a note held across the boundary re-appears in the tie section."""
  ns = note_sequences.NoteSequence()
  ns.add_note(pitch=60, velocity=100, start_time=0.5, end_time=6.0, program=0)
  ns.add_note(pitch=64, velocity=90, start_time=1.0, end_time=2.0, program=40)
  ns.add_note(pitch=36, velocity=90, start_time=5.5, end_time=5.6, is_drum=True)
  ns.total_time = 6.0
  segs = tokenizer.note_sequence_to_model_inputs(ns)
  # codec blocks (+3): shift 3..1003, pitch 1004.., velocity 1132/1133, tie 1134, program 1135.., drum 1263..
  want0 = [1134, 53, 1135, 1133, 1064, 103, 1175, 1068, 203, 1132, 1068, 1]
  want1 = [1135, 1064, 1134, 41, 1133, 1299, 91, 1132, 1064, 1]
  np.testing.assert_array_equal(segs[0][0, :len(want0) + 1], want0 + [0])
  np.testing.assert_array_equal(segs[1][0, :len(want1) + 1], want1 + [0])
  assert len(segs) == 2


def test_too_long_and_empty():
  rng = np.random.default_rng(0)
  ns = note_sequences.NoteSequence()
  for i in range(1500):   # 1500 notes inside one segment: > 2047 tokens
    t = round(float(rng.uniform(0, 4.9)), 2)
    ns.add_note(pitch=int(20 + i % 100), velocity=100, start_time=t, end_time=t + 0.01 + 0.01 * (i % 3), program=i % 8)
  ns.total_time = 5.0
  ns = note_sequences.trim_overlapping_notes(ns)
  with pytest.raises(ValueError, match='exceeds maximum length'):
    tokenizer.note_sequence_to_model_inputs(ns)
  segs = tokenizer.note_sequence_to_model_inputs(ns, on_too_long='truncate')
  assert segs[0][0, 2047] == 1 and (segs[0][0, :2047] >= 3).all()
  bad = note_sequences.NoteSequence()
  bad.add_note(pitch=60, velocity=0, start_time=0.0, end_time=1.0)
  with pytest.raises(ValueError, match='zero velocity'):
    tokenizer.note_sequence_to_model_inputs(bad)
  with pytest.raises(ValueError, match='no notes'):
    tokenizer.note_sequence_to_model_inputs(note_sequences.NoteSequence())


def test_midi_file_to_model_inputs(tmp_path):
  ns = _random_song(5, seconds=11.0)
  path = tmp_path / 'song.mid'
  path.write_bytes(midi_io.note_sequence_to_midi(ns, ticks_per_quarter=500, tempo_us=500000))   # 1 ms ticks
  a = tokenizer.midi_file_to_model_inputs(str(path))
  b = tokenizer.note_sequence_to_model_inputs(ns)
  assert len(a) == len(b)
  for x, y in zip(a, b):
    np.testing.assert_array_equal(x, y)


def test_trim_and_assign_instruments():
  ns = note_sequences.NoteSequence()
  ns.add_note(pitch=60, velocity=100, start_time=0.0, end_time=1.0, program=0)
  ns.add_note(pitch=60, velocity=100, start_time=0.5, end_time=0.8, program=0)
  ns.add_note(pitch=60, velocity=100, start_time=0.5, end_time=0.9, program=1)
  out = note_sequences.trim_overlapping_notes(ns)
  assert sorted((n.program, n.start_time, n.end_time) for n in out.notes) == [(0, 0.0, 0.5), (0, 0.5, 0.8), (1, 0.5, 0.9)]
  many = note_sequences.note_arrays_to_note_sequence([0.0] * 12, list(range(60, 72)), programs=list(range(12)),
                                                     is_drums=[False] * 11 + [True])
  assert [n.instrument for n in many.notes] == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 9]


def test_synthesize_cli_dry_run(tmp_path, capsys):
  from msd_amd import synthesize
  ns = _random_song(9, seconds=12.0)
  path = tmp_path / 'cli.mid'
  path.write_bytes(midi_io.note_sequence_to_midi(ns, ticks_per_quarter=500))
  assert synthesize.main([str(path), '--dry-run']) == 0
  err = capsys.readouterr().err
  assert '3 segments of 256 frames' in err and 'tokens per segment' in err
