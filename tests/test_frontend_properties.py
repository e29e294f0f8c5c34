"""Property tests (hypothesis) for the front end's integer transforms: size-independent invariants the
domain offers -- run-length encode -> decode preserves every event time; the codec is a bijection on its
index space; the vocabulary is a bijection on valid tokens; songs survive tokenise -> decode for random
segment-boundary positions.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

import msd_amd  # noqa: F401
from msd_amd.frontend import event_codec, note_sequences, run_length_encoding, tokenizer, vocabularies

CODEC = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))


@settings(max_examples=200, deadline=None)
@given(st.integers(0, CODEC.num_classes - 1))
def test_codec_is_a_bijection(index):
  ev = CODEC.decode_event_index(index)
  assert CODEC.encode_event(ev) == index
  lo, hi = CODEC.event_type_range(ev.type)
  assert lo <= index <= hi
  assert CODEC.is_shift_event_index(index) == (ev.type == 'shift')


@settings(max_examples=100, deadline=None)
@given(st.lists(st.integers(0, CODEC.num_classes - 1), max_size=50))
def test_vocabulary_round_trip(tokens):
  vocab = vocabularies.vocabulary_from_codec(CODEC)
  ids = vocab.encode(tokens)
  assert vocab.decode(ids) == tokens
  np.testing.assert_array_equal(vocab.decode_array(vocab.encode_array(np.array(tokens, np.int64))), tokens)
  assert all(3 <= i < vocab.vocab_size - vocab.extra_ids for i in ids)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 2500), st.integers(0, 127)), min_size=1, max_size=40))
def test_rle_preserves_event_times(events):
  """Unit-shift stream -> run-length encoded (absolute shifts, chunks of max_shift) -> decode: every
  pitch event comes back at its own step, in order."""
  events = sorted(events)
  pitch0 = CODEC.event_type_range('pitch')[0]
  stream, step = [], 0
  for t, p in events:
    stream += [1] * (t - step)            # 'shift 1' has index 1
    step = t
    stream.append(pitch0 + p)
  out = run_length_encoding.run_length_encode_shifts_fn(CODEC)({'targets': np.array(stream)})['targets']
  assert all(0 < tok <= CODEC.max_shift_steps or tok >= pitch0 for tok in out)
  got = []
  class S: pass
  def fn(state, time, event, codec):
    got.append((round(time * CODEC.steps_per_second), event.value))
  run_length_encoding.decode_events(S(), out, 0, None, CODEC, fn)
  assert got == events


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10_000), st.integers(64, 512))
def test_song_round_trip_for_any_segment_length(seed, segment_frames):
  rng = np.random.default_rng(seed)
  ns = note_sequences.NoteSequence()
  busy = {}
  for _ in range(40):
    program, pitch = int(rng.choice([0, 25, 60])), int(rng.integers(30, 100))
    start, dur = int(rng.integers(0, 1500)), int(rng.integers(2, 400))
    lane = busy.setdefault((program, pitch), [])
    if any(s < start + dur + 1 and start < e + 1 for s, e in lane):
      continue
    lane.append((start, start + dur))
    ns.add_note(pitch=pitch, velocity=64, start_time=start / 100, end_time=(start + dur) / 100, program=program)
  ns.total_time = max(n.end_time for n in ns.notes)
  cfg = tokenizer.FrontendConfig(segment_frames=segment_frames)
  back = tokenizer.decode_model_inputs(tokenizer.note_sequence_to_model_inputs(ns, cfg), cfg)
  key = lambda s: sorted((n.program, n.pitch, round(n.start_time * 100), round(n.end_time * 100)) for n in s.notes)
  assert key(back) == key(ns)
