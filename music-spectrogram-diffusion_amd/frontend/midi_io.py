"""Standard MIDI File -> NoteSequence, and the sustain-pedal pass the reference applies before
tokenising (preprocessors.py:167).

The reference reads MIDI through third-party code that is neither installed here nor vendored under
/root/reference: note_seq.midi_file_to_note_sequence (which parses with pretty_midi) and
note_seq.sequences_lib.apply_sustain_control_changes (note-seq; the reference's setup.py does not pin a
version).  This module restates their published behaviour from scratch; it has no golden vectors
from the reference's own tests (parity unpinned for this file; tests/test_frontend_midi.py pins it
against hand-computed cases and a write -> read round trip).

Parsing rules followed (pretty_midi 0.2.x `PrettyMIDI._load_instruments` / `_load_tempo_changes`):
  * times = tempo map over ticks, default 500000 us per quarter; seconds per tick of a tempo segment
    = 60 / (bpm * ticks_per_quarter) with bpm = 6e7 / tempo_us;
  * one instrument per (program at note-off, channel, track); channel 10 (index 9) = drums;
  * a note-off (or note-on with velocity 0) closes every open note of that (channel, pitch) that did
    not start on the same tick; zero-length notes are never produced;
  * notes still open at the end of the file are dropped.
Tempo events are honoured from every track (pretty_midi reads them from track 0 only and warns
otherwise; identical for well-formed type-0/1 files)."""
from __future__ import annotations

import struct
from typing import Dict, List, Sequence, Tuple

from .note_sequences import ControlChange, Note, NoteSequence

SUSTAIN_CC = 64


class MidiError(ValueError):
  pass


def _varlen(data: bytes, pos: int) -> Tuple[int, int]:
  value = 0
  while True:
    if pos >= len(data):
      raise MidiError('truncated variable-length quantity')
    b = data[pos]
    pos += 1
    value = (value << 7) | (b & 0x7F)
    if not b & 0x80:
      return value, pos


def _parse_track(data: bytes):
  """-> list of (abs_tick, kind, a, b, c): kind in 'on','off','cc','pc','tempo'."""
  out = []
  pos, tick, status = 0, 0, None
  n = len(data)
  while pos < n:
    delta, pos = _varlen(data, pos)
    tick += delta
    if pos >= n:
      raise MidiError('truncated track event')
    b0 = data[pos]
    if b0 == 0xFF:                                    # meta
      if pos + 1 >= n:
        raise MidiError('truncated meta event')
      mtype = data[pos + 1]
      length, p = _varlen(data, pos + 2)
      payload = data[p:p + length]
      pos = p + length
      if mtype == 0x51 and length == 3:
        out.append((tick, 'tempo', (payload[0] << 16) | (payload[1] << 8) | payload[2], 0, 0))
      elif mtype == 0x2F:
        break
      status = None                                   # meta / sysex cancel running status
      continue
    if b0 in (0xF0, 0xF7):                            # sysex
      length, p = _varlen(data, pos + 1)
      pos = p + length
      status = None
      continue
    if b0 & 0x80:
      status = b0
      pos += 1
    elif status is None:
      raise MidiError('data byte without running status')
    hi, ch = status & 0xF0, status & 0x0F
    need = 1 if hi in (0xC0, 0xD0) else 2
    if pos + need > n:
      raise MidiError('truncated channel message')
    d1 = data[pos]
    d2 = data[pos + 1] if need == 2 else 0
    pos += need
    if hi == 0x90 and d2 > 0:
      out.append((tick, 'on', ch, d1, d2))
    elif hi == 0x80 or hi == 0x90:
      out.append((tick, 'off', ch, d1, 0))
    elif hi == 0xB0:
      out.append((tick, 'cc', ch, d1, d2))
    elif hi == 0xC0:
      out.append((tick, 'pc', ch, d1, 0))
  return out


def parse_midi(data: bytes) -> NoteSequence:
  if data[:4] != b'MThd' or len(data) < 14:
    raise MidiError('not a Standard MIDI File')
  hlen, fmt, ntrks, division = struct.unpack('>IHHH', data[4:14])
  if division & 0x8000:
    raise MidiError('SMPTE time division is not supported')
  if fmt not in (0, 1):
    raise MidiError('MIDI format %d is not supported' % fmt)
  pos = 8 + hlen
  tracks = []
  for _ in range(ntrks):
    if data[pos:pos + 4] != b'MTrk':
      raise MidiError('missing MTrk chunk')
    (tlen,) = struct.unpack('>I', data[pos + 4:pos + 8])
    tracks.append(_parse_track(data[pos + 8:pos + 8 + tlen]))
    pos += 8 + tlen

  # ---- tempo map -> seconds -----------------------------------------------------------
  tempi = sorted((t, us) for tr in tracks for (t, kind, us, _, _) in tr if kind == 'tempo')
  seg_tick, seg_time, seg_scale = [0], [0.0], [60.0 / ((60000000.0 / 500000) * division)]
  for t, us in tempi:
    scale = 60.0 / ((60000000.0 / us) * division)
    if t == seg_tick[-1]:
      seg_scale[-1] = scale                            # later event at the same tick wins
    else:
      seg_time.append(seg_time[-1] + seg_scale[-1] * (t - seg_tick[-1]))
      seg_tick.append(t)
      seg_scale.append(scale)

  import bisect
  def seconds(tick: int) -> float:
    k = bisect.bisect_right(seg_tick, tick) - 1
    return seg_time[k] + seg_scale[k] * (tick - seg_tick[k])

  # ---- notes / control changes ----------------------------------------------------------
  ns = NoteSequence(ticks_per_quarter=division)
  instruments: Dict[Tuple[int, int, int], int] = {}     # (program, channel, track) -> index
  def instrument_of(program, ch, trk):
    key = (program, ch, trk)
    if key not in instruments:
      instruments[key] = len(instruments)
    return instruments[key]

  for trk, events in enumerate(tracks):
    program = [0] * 16
    open_notes: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}
    for tick, kind, ch, d1, d2 in events:
      if kind == 'pc':
        program[ch] = d1
      elif kind == 'on':
        open_notes.setdefault((ch, d1), []).append((tick, d2))
      elif kind == 'off':
        pending = open_notes.get((ch, d1))
        if pending:
          keep = [(s, v) for s, v in pending if s == tick]
          for s, v in pending:
            if s != tick:
              ns.notes.append(Note(pitch=d1, velocity=v, start_time=seconds(s), end_time=seconds(tick),
                                   program=program[ch], is_drum=(ch == 9),
                                   instrument=instrument_of(program[ch], ch, trk)))
          if keep:
            open_notes[(ch, d1)] = keep
          else:
            del open_notes[(ch, d1)]
      elif kind == 'cc':
        ns.control_changes.append(ControlChange(time=seconds(tick), control_number=d1, control_value=d2,
                                                instrument=instrument_of(program[ch], ch, trk),
                                                program=program[ch], is_drum=(ch == 9)))
  ns.total_time = max((n.end_time for n in ns.notes), default=0.0)
  return ns


def midi_file_to_note_sequence(path: str) -> NoteSequence:
  """note_seq.midi_file_to_note_sequence (entry point of the reference's MIDI notebooks)."""
  with open(path, 'rb') as f:
    ns = parse_midi(f.read())
  ns.filename = path
  return ns


_SUS_ON, _SUS_OFF, _NOTE_ON, _NOTE_OFF = 0, 1, 2, 3


def apply_sustain_control_changes(ns: NoteSequence, sustain_control_number: int = SUSTAIN_CC) -> NoteSequence:
  """note_seq.sequences_lib.apply_sustain_control_changes, per instrument, drums untouched: while the
  pedal is down (controller value >= 64) a released note keeps sounding until the pedal comes up or
  the same pitch is struck again; notes still held by a pedal at the end of the sequence end at the
  last event time.  Events at equal times are processed pedal-down, pedal-up, note-on, note-off."""
  out = ns.copy()
  events = []
  for note in out.notes:
    if not note.is_drum:
      events.append((note.start_time, _NOTE_ON, note))
      events.append((note.end_time, _NOTE_OFF, note))
  for cc in out.control_changes:
    if cc.control_number == sustain_control_number:
      events.append((cc.time, _SUS_ON if cc.control_value >= 64 else _SUS_OFF, cc))
  events.sort(key=lambda e: (e[0], e[1]))      # stable: insertion order breaks the remaining ties

  held: Dict[int, List[Note]] = {}
  pedal: Dict[int, bool] = {}
  removed = set()
  now = 0.0
  for now, kind, obj in events:
    inst = obj.instrument
    if kind == _SUS_ON:
      pedal[inst] = True
    elif kind == _SUS_OFF:
      pedal[inst] = False
      still = []
      for note in held.get(inst, []):
        if note.end_time < now:                 # was being extended by the pedal
          note.end_time = now
          out.total_time = max(out.total_time, now)
        else:
          still.append(note)
      held[inst] = still
    elif kind == _NOTE_ON:
      if pedal.get(inst, False):
        still = []
        for note in held.get(inst, []):
          if note.pitch == obj.pitch:
            note.end_time = now
            if note.start_time == note.end_time:
              removed.add(id(note))             # same pitch re-struck at the same instant
          else:
            still.append(note)
        held[inst] = still
      held.setdefault(inst, []).append(obj)
    else:
      if not pedal.get(inst, False):
        lst = held.get(inst, [])
        for k, note in enumerate(lst):
          if note is obj:
            del lst[k]
            break
  for lst in held.values():
    for note in lst:
      note.end_time = now
      out.total_time = now
  if removed:
    out.notes = [n for n in out.notes if id(n) not in removed]
  return out


# ---- writer (tests, synthetic corpora) ------------------------------------------------------
def _vl(n: int) -> bytes:
  chunks = [n & 0x7F]
  n >>= 7
  while n:
    chunks.append((n & 0x7F) | 0x80)
    n >>= 7
  return bytes(reversed(chunks))


def write_midi(tracks: Sequence[Sequence[Tuple[int, bytes]]], ticks_per_quarter: int = 480, fmt: int = 1) -> bytes:
  """tracks: per track a list of (absolute_tick, raw message bytes incl. status; meta as FF tt len ..)."""
  out = [b'MThd', struct.pack('>IHHH', 6, fmt, len(tracks), ticks_per_quarter)]
  for events in tracks:
    body = bytearray()
    last = 0
    for tick, msg in sorted(events, key=lambda e: e[0]):
      body += _vl(tick - last) + msg
      last = tick
    body += b'\x00\xff\x2f\x00'
    out += [b'MTrk', struct.pack('>I', len(body)), bytes(body)]
  return b''.join(out)


def note_sequence_to_midi(ns: NoteSequence, ticks_per_quarter: int = 480, tempo_us: int = 500000) -> bytes:
  """One track per (program, is_drum); times quantised to ticks at a constant tempo."""
  per_tick = tempo_us / 1e6 / ticks_per_quarter
  groups: Dict[Tuple[int, bool], List[Note]] = {}
  for n in ns.notes:
    groups.setdefault((n.program, n.is_drum), []).append(n)
  tracks = [[(0, b'\xff\x51\x03' + tempo_us.to_bytes(3, 'big'))]]
  next_ch = 0
  for (program, is_drum), notes in sorted(groups.items()):
    if is_drum:
      ch = 9
    else:
      ch = next_ch if next_ch != 9 else 10
      next_ch = ch + 1
      if ch > 15:
        raise MidiError('more than 15 pitched programs need several files/ports')
    ev = [(0, bytes([0xC0 | ch, program]))]
    for n in notes:
      ev.append((int(round(n.start_time / per_tick)), bytes([0x90 | ch, n.pitch, n.velocity])))
      ev.append((int(round(n.end_time / per_tick)), bytes([0x80 | ch, n.pitch, 0])))
    # note-offs before note-ons at equal ticks
    ev.sort(key=lambda e: (e[0], 0 if (e[1][0] & 0xF0) in (0x80, 0xC0) else 1))
    tracks.append(ev)
  return write_midi(tracks, ticks_per_quarter)
