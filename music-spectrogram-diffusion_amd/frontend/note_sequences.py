"""Notes <-> timed note events <-> codec events (reference: note_sequences.py:28-445).

The reference works on note_seq.NoteSequence protos (note-seq is not installed here and is not part
of /root/reference); this module carries a plain-dataclass container with the fields the path reads
(notes with pitch / velocity / start / end / program / is_drum / instrument, total_time, control
changes) and restates the reference functions on it."""
from __future__ import annotations

import dataclasses
from typing import Dict, List, MutableMapping, Optional, Sequence, Set, Tuple

from . import event_codec, vocabularies

Event = event_codec.Event

DEFAULT_VELOCITY = 100          # note_sequences.py:25-26
DEFAULT_NOTE_DURATION = 0.01
MIN_NOTE_DURATION = 0.01        # note_sequences.py:29


@dataclasses.dataclass
class Note:
  pitch: int
  velocity: int
  start_time: float
  end_time: float
  program: int = 0
  is_drum: bool = False
  instrument: int = 0


@dataclasses.dataclass
class ControlChange:
  time: float
  control_number: int
  control_value: int
  instrument: int = 0
  program: int = 0
  is_drum: bool = False


@dataclasses.dataclass
class NoteSequence:
  notes: List[Note] = dataclasses.field(default_factory=list)
  total_time: float = 0.0
  ticks_per_quarter: int = 220
  control_changes: List[ControlChange] = dataclasses.field(default_factory=list)
  id: str = ''
  filename: str = ''

  def add_note(self, **kw) -> Note:
    n = Note(**kw)
    self.notes.append(n)
    return n

  def copy(self) -> 'NoteSequence':
    return NoteSequence([dataclasses.replace(n) for n in self.notes], self.total_time,
                        self.ticks_per_quarter, [dataclasses.replace(c) for c in self.control_changes],
                        self.id, self.filename)


def extract_track(ns: NoteSequence, program: int, is_drum: bool) -> NoteSequence:   # note_sequences.py:39-47
  notes = [dataclasses.replace(n) for n in ns.notes if n.program == program and n.is_drum == is_drum]
  return NoteSequence(notes, max((n.end_time for n in notes), default=0.0), 220)


def trim_overlapping_notes(ns: NoteSequence) -> NoteSequence:
  """Per (pitch, program, is_drum): cut a note where the next one starts; drop empty notes
  (note_sequences.py:50-68)."""
  out = ns.copy()
  lanes: Dict[Tuple[int, int, bool], List[Note]] = {}
  for n in out.notes:
    lanes.setdefault((n.pitch, n.program, n.is_drum), []).append(n)
  for lane in lanes.values():
    lane.sort(key=lambda n: n.start_time)       # stable, like sorted()
    for prev, nxt in zip(lane, lane[1:]):
      if prev.end_time > nxt.start_time:
        prev.end_time = nxt.start_time
  out.notes = [n for n in out.notes if n.start_time < n.end_time]
  return out


def assign_instruments(ns: NoteSequence) -> None:
  """One instrument number per program in order of first appearance, skipping 9, which is the drum
  instrument (note_sequences.py:71-84)."""
  by_program: Dict[int, int] = {}
  for n in ns.notes:
    if n.is_drum:
      n.instrument = 9
    else:
      if n.program not in by_program:
        k = len(by_program)
        by_program[n.program] = k if k < 9 else k + 1
      n.instrument = by_program[n.program]


def validate_note_sequence(ns: NoteSequence) -> None:      # note_sequences.py:87-95
  for n in ns.notes:
    if n.start_time >= n.end_time:
      raise ValueError('note has start time >= end time: %f >= %f' % (n.start_time, n.end_time))
    if n.velocity == 0:
      raise ValueError('note has zero velocity')


def note_arrays_to_note_sequence(onset_times, pitches, offset_times=None, velocities=None,
                                 programs=None, is_drums=None) -> NoteSequence:
  """note_sequences.py:98-133 (missing arrays take the defaults)."""
  ns = NoteSequence(ticks_per_quarter=220)
  n = max(len(onset_times), len(pitches))
  def at(seq, i, default):
    return default if seq is None or i >= len(seq) else seq[i]
  for i in range(n):
    on = onset_times[i]
    off = at(offset_times, i, None)
    if off is None:
      off = on + DEFAULT_NOTE_DURATION
    ns.add_note(start_time=on, end_time=off, pitch=pitches[i], velocity=at(velocities, i, DEFAULT_VELOCITY),
                program=at(programs, i, 0), is_drum=at(is_drums, i, False))
    ns.total_time = max(ns.total_time, off)
  assign_instruments(ns)
  return ns


@dataclasses.dataclass
class NoteEventData:             # note_sequences.py:136-142
  pitch: int
  velocity: Optional[int] = None
  program: Optional[int] = None
  is_drum: Optional[bool] = None
  instrument: Optional[int] = None


def note_sequence_to_onsets(ns: NoteSequence):
  """Onset times + pitches; pitch order is the tie-breaker of the later stable time sort
  (note_sequences.py:145-152)."""
  notes = sorted(ns.notes, key=lambda n: n.pitch)
  return [n.start_time for n in notes], [NoteEventData(pitch=n.pitch) for n in notes]


def note_sequence_to_onsets_and_offsets(ns: NoteSequence):
  """All offsets (velocity 0) first, then all onsets, both in pitch order: offsets win ties
  (note_sequences.py:155-178)."""
  notes = sorted(ns.notes, key=lambda n: n.pitch)
  times = [n.end_time for n in notes] + [n.start_time for n in notes]
  values = ([NoteEventData(pitch=n.pitch, velocity=0) for n in notes] +
            [NoteEventData(pitch=n.pitch, velocity=n.velocity) for n in notes])
  return times, values


def note_sequence_to_onsets_and_offsets_and_programs(ns: NoteSequence):
  """As above with programs, ordered by (is_drum, program, pitch); drums have no offsets
  (note_sequences.py:181-207)."""
  notes = sorted(ns.notes, key=lambda n: (n.is_drum, n.program, n.pitch))
  pitched = [n for n in notes if not n.is_drum]
  times = [n.end_time for n in pitched] + [n.start_time for n in notes]
  values = ([NoteEventData(pitch=n.pitch, velocity=0, program=n.program, is_drum=False) for n in pitched] +
            [NoteEventData(pitch=n.pitch, velocity=n.velocity, program=n.program, is_drum=n.is_drum)
             for n in notes])
  return times, values


@dataclasses.dataclass
class NoteEncodingState:
  """Velocity bin of every (pitch, program) seen so far; 0 = currently off (note_sequences.py:210-215)."""
  active_pitches: MutableMapping[Tuple[int, int], int] = dataclasses.field(default_factory=dict)


def note_event_data_to_events(state: Optional[NoteEncodingState], value: NoteEventData,
                              codec: event_codec.Codec) -> Sequence[Event]:
  """note_sequences.py:218-252: onset-only / velocity+pitch / velocity+drum / program+velocity+pitch."""
  if value.velocity is None:
    return [Event('pitch', value.pitch)]
  vbin = vocabularies.velocity_to_bin(value.velocity, vocabularies.num_velocity_bins_from_codec(codec))
  if value.program is None:
    if state is not None:
      state.active_pitches[(value.pitch, 0)] = vbin
    return [Event('velocity', vbin), Event('pitch', value.pitch)]
  if value.is_drum:
    return [Event('velocity', vbin), Event('drum', value.pitch)]
  if state is not None:
    state.active_pitches[(value.pitch, value.program)] = vbin
  return [Event('program', value.program), Event('velocity', vbin), Event('pitch', value.pitch)]


def note_encoding_state_to_events(state: NoteEncodingState) -> Sequence[Event]:
  """(program, pitch) of every sounding note in (program, pitch) order, then the tie marker
  (note_sequences.py:255-266)."""
  out: List[Event] = []
  for pitch, program in sorted(state.active_pitches, key=lambda k: (k[1], k[0])):
    if state.active_pitches[(pitch, program)]:
      out.append(Event('program', program))
      out.append(Event('pitch', pitch))
  out.append(Event('tie', 0))
  return out


@dataclasses.dataclass
class NoteDecodingState:         # note_sequences.py:269-288
  current_time: float = 0.0
  current_velocity: int = DEFAULT_VELOCITY
  current_program: int = 0
  active_pitches: MutableMapping[Tuple[int, int], Tuple[float, int]] = dataclasses.field(default_factory=dict)
  tied_pitches: Set[Tuple[int, int]] = dataclasses.field(default_factory=set)
  is_tie_section: bool = False
  note_sequence: NoteSequence = dataclasses.field(default_factory=lambda: NoteSequence(ticks_per_quarter=220))


def _add_note(ns: NoteSequence, start_time, end_time, pitch, velocity, program=0, is_drum=False) -> None:
  end_time = max(end_time, start_time + MIN_NOTE_DURATION)       # note_sequences.py:300-309
  ns.add_note(start_time=start_time, end_time=end_time, pitch=pitch, velocity=velocity,
              program=program, is_drum=is_drum)
  ns.total_time = max(ns.total_time, end_time)


def decode_note_onset_event(state: NoteDecodingState, time: float, event: Event, codec) -> None:
  """note_sequences.py:283-297: onsets-only vocabulary; anything but 'pitch' is invalid."""
  if event.type != 'pitch':
    raise ValueError('unexpected event type: %s' % event.type)
  ns = state.note_sequence
  ns.add_note(start_time=time, end_time=time + DEFAULT_NOTE_DURATION, pitch=event.value, velocity=DEFAULT_VELOCITY)
  ns.total_time = max(ns.total_time, time + DEFAULT_NOTE_DURATION)


def decode_note_event(state: NoteDecodingState, time: float, event: Event, codec: event_codec.Codec) -> None:
  """note_sequences.py:312-389."""
  if time < state.current_time:
    raise ValueError('event time < current time, %f < %f' % (time, state.current_time))
  state.current_time = time
  kind = event.type
  if kind == 'pitch':
    key = (event.value, state.current_program)
    if state.is_tie_section:
      if key not in state.active_pitches:
        raise ValueError('inactive pitch/program in tie section: %d/%d' % key)
      if key in state.tied_pitches:
        raise ValueError('pitch/program is already tied: %d/%d' % key)
      state.tied_pitches.add(key)
    elif state.current_velocity == 0:
      if key not in state.active_pitches:
        raise ValueError('note-off for inactive pitch/program: %d/%d' % key)
      on, vel = state.active_pitches.pop(key)
      _add_note(state.note_sequence, on, time, key[0], vel, program=key[1])
    else:
      if key in state.active_pitches:      # re-struck while sounding: close the old note first
        on, vel = state.active_pitches.pop(key)
        _add_note(state.note_sequence, on, time, key[0], vel, program=key[1])
      state.active_pitches[key] = (time, state.current_velocity)
  elif kind == 'drum':
    if state.current_velocity == 0:
      raise ValueError('velocity cannot be zero for drum event')
    _add_note(state.note_sequence, time, time + DEFAULT_NOTE_DURATION, event.value, state.current_velocity,
              is_drum=True)
  elif kind == 'velocity':
    state.current_velocity = vocabularies.bin_to_velocity(
        event.value, vocabularies.num_velocity_bins_from_codec(codec))
  elif kind == 'program':
    state.current_program = event.value
  elif kind == 'tie':
    if not state.is_tie_section:
      raise ValueError('tie section end event when not in tie section')
    for key in list(state.active_pitches):   # notes not declared tied ended with the last segment
      if key not in state.tied_pitches:
        on, vel = state.active_pitches.pop(key)
        _add_note(state.note_sequence, on, state.current_time, key[0], vel, program=key[1])
    state.is_tie_section = False
  else:
    raise ValueError('unexpected event type: %s' % kind)


def begin_tied_pitches_section(state: NoteDecodingState) -> None:     # note_sequences.py:392-395
  state.tied_pitches = set()
  state.is_tie_section = True


def flush_note_decoding_state(state: NoteDecodingState) -> NoteSequence:
  """End whatever still sounds (note_sequences.py:398-409)."""
  for on, _ in state.active_pitches.values():
    state.current_time = max(state.current_time, on + MIN_NOTE_DURATION)
  for key in list(state.active_pitches):
    on, vel = state.active_pitches.pop(key)
    _add_note(state.note_sequence, on, state.current_time, key[0], vel, program=key[1])
  assign_instruments(state.note_sequence)
  return state.note_sequence


@dataclasses.dataclass
class EventEncodingSpec:         # run_length_encoding.py:38-58 (kept here to avoid a module cycle)
  init_encoding_state_fn: object
  encode_event_fn: object
  encoding_state_to_events_fn: object
  init_decoding_state_fn: object
  begin_decoding_segment_fn: object
  decode_event_fn: object
  flush_decoding_state_fn: object


# note_sequences.py:416-445
NoteOnsetEncodingSpec = EventEncodingSpec(lambda: None, note_event_data_to_events, None, NoteDecodingState,
                                          lambda state: None, decode_note_onset_event,
                                          lambda state: state.note_sequence)
NoteEncodingSpec = EventEncodingSpec(lambda: None, note_event_data_to_events, None, NoteDecodingState,
                                     lambda state: None, decode_note_event, flush_note_decoding_state)
NoteEncodingWithTiesSpec = EventEncodingSpec(NoteEncodingState, note_event_data_to_events,
                                             note_encoding_state_to_events, NoteDecodingState,
                                             begin_tied_pitches_section, decode_note_event,
                                             flush_note_decoding_state)
