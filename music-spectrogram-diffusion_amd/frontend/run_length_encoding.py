"""Timed events -> unit-shift event stream indexed by audio frame, segment extraction with the tie
section, run-length encoding of shifts, and the inverse decode loop
(reference: run_length_encoding.py:61-326).  Plain NumPy / Python; the reference's tf.data /
autograph wrappers are replaced by ordinary functions on dicts of arrays."""
from __future__ import annotations

from typing import Callable, List, Mapping, MutableMapping, Optional, Sequence, Tuple

import numpy as np

from . import event_codec

Event = event_codec.Event


def encode_and_index_events(state, event_times: Sequence[float], event_values: Sequence, encode_event_fn,
                            codec: event_codec.Codec, frame_times: Sequence[float],
                            encoding_state_to_events_fn=None):
  """run_length_encoding.py:61-176.

  Events are quantised to codec steps (Python round = half-to-even, like the reference), stably sorted
  by time, and written out with ONE 'shift 1' token per elapsed step.  For every audio frame the index
  of the first event at/after the frame start is recorded (event_start_indices; event_end_indices is
  the same array shifted by one frame, so consecutive slices abut), plus -- when a state->events
  function is given -- the position in `state_events` of the state dump that precedes that event.

  Returns (events, event_start_indices, event_end_indices, state_events, state_event_indices)."""
  sps = codec.steps_per_second
  order = np.argsort(np.asarray(event_times, dtype=np.float64), kind='stable')
  steps = [round(float(event_times[i]) * sps) for i in order]
  values = [event_values[i] for i in order]
  shift_one = codec.encode_event(Event('shift', 1))
  n_frames = len(frame_times)

  events: List[int] = []
  state_events: List[int] = []
  start_idx: List[int] = []
  state_idx: List[int] = []
  cur_step = 0
  cur_event = 0          # index of the first event of the current step
  cur_state_event = 0

  def cover_frames():
    # frames that start before the current step begin with the events of the previous step
    t = cur_step / sps
    while len(start_idx) < n_frames and frame_times[len(start_idx)] < t:
      start_idx.append(cur_event)
      state_idx.append(cur_state_event)

  for step, value in zip(steps, values):
    while step > cur_step:
      events.append(shift_one)
      cur_step += 1
      cover_frames()
      cur_event = len(events)
      cur_state_event = len(state_events)
    if encoding_state_to_events_fn is not None:
      # the state BEFORE this event (run_length_encoding.py:139-143)
      state_events.extend(codec.encode_event(e) for e in encoding_state_to_events_fn(state))
    events.extend(codec.encode_event(e) for e in encode_event_fn(state, value, codec))

  # trailing shifts until the last frame is covered; "<=": a step that coincides with a frame start
  # needs one more shift to cover that frame (run_length_encoding.py:147-155)
  while cur_step / sps <= frame_times[-1]:
    events.append(shift_one)
    cur_step += 1
    cover_frames()
    cur_event = len(events)

  end_idx = start_idx[1:] + [len(events)]
  return (np.array(events), np.array(start_idx), np.array(end_idx), np.array(state_events),
          np.array(state_idx))


def extract_sequence_with_indices(features: Mapping[str, np.ndarray], state_events_end_token: Optional[int] = None,
                                  feature_key: str = 'targets') -> dict:
  """Events of one audio segment = events[start of its first frame : end of its last frame], preceded --
  with ties -- by the state dump in force at the segment start, up to and including its tie token
  (run_length_encoding.py:179-205)."""
  out = dict(features)
  lo = int(features['event_start_indices'][0])
  hi = int(features['event_end_indices'][-1])
  seq = np.asarray(features[feature_key])[lo:hi]
  if state_events_end_token is not None:
    st = np.asarray(features['state_events'])
    a = int(features['state_event_indices'][0])
    b = a + 1
    while st[b - 1] != state_events_end_token:
      b += 1
    seq = np.concatenate([st[a:b], seq], axis=0)
  out[feature_key] = seq
  return out


def run_length_encode_shifts_fn(codec: event_codec.Codec, feature_key: str = 'targets',
                                state_change_event_types: Sequence[str] = ()) -> Callable[[MutableMapping], Mapping]:
  """run_length_encoding.py:208-275.  Returns f(features) -> features with features[feature_key]
  re-encoded: runs of unit shifts become shift tokens carrying the time SINCE THE SEGMENT START (the
  reference re-emits the running total, in chunks of max_shift_steps), shifts after the last event are
  dropped, and a state-change token (e.g. velocity / program) equal to the current state is dropped."""
  ranges = [codec.event_type_range(t) for t in state_change_event_types]
  max_shift = codec.max_shift_steps

  def run_length_encode_shifts(features: MutableMapping) -> Mapping:
    pending = 0        # unit shifts since the last emitted event
    total = 0          # unit shifts since the segment start
    current = [0] * len(ranges)
    out: List[int] = []
    for ev in np.asarray(features[feature_key]).tolist():
      if codec.is_shift_event_index(ev):
        pending += 1
        total += 1
        continue
      redundant = False
      for i, (lo, hi) in enumerate(ranges):
        if lo <= ev <= hi:
          if current[i] == ev:
            redundant = True
          current[i] = ev
      if redundant:
        continue
      if pending > 0:
        left = total
        while left > 0:
          step = min(max_shift, left)
          out.append(step)
          left -= step
        pending = 0
      out.append(ev)
    features[feature_key] = np.array(out, dtype=np.int32)
    return features

  return run_length_encode_shifts


def decode_events(state, tokens, start_time: float, max_time: Optional[float], codec: event_codec.Codec,
                  decode_event_fn) -> Tuple[int, int]:
  """Inverse walk (run_length_encoding.py:278-326): a shift token sets the time to start_time + its
  steps (shifts are absolute within a segment; consecutive shift tokens add up), any other token is
  handed to decode_event_fn at the current time.  Returns (invalid_events, dropped_events)."""
  invalid = dropped = 0
  steps = 0
  now = start_time
  tokens = list(tokens)
  for pos, tok in enumerate(tokens):
    try:
      ev = codec.decode_event_index(tok)
    except ValueError:
      invalid += 1
      continue
    if ev.type == 'shift':
      steps += ev.value
      now = start_time + steps / codec.steps_per_second
      if max_time and now > max_time:
        dropped = len(tokens) - pos
        break
    else:
      steps = 0
      try:
        decode_event_fn(state, now, ev, codec)
      except ValueError:
        invalid += 1
  return invalid, dropped
