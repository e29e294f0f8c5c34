"""Event <-> integer index codec (reference: event_codec.py:20-112).

The index space is the concatenation of the value ranges of the event types, with 'shift' always
first (so shift indices start at 0: event_codec.py:54-58).  Ranges are laid out once at construction;
encode/decode are table lookups instead of the reference's linear scans."""
from __future__ import annotations

import bisect
import dataclasses
from typing import List, Sequence, Tuple


@dataclasses.dataclass(frozen=True)
class EventRange:           # event_codec.py:20-24
  type: str
  min_value: int
  max_value: int

  @property
  def size(self) -> int:
    return self.max_value - self.min_value + 1


@dataclasses.dataclass(frozen=True)
class Event:                # event_codec.py:27-30
  type: str
  value: int


class Codec:
  """event_codec.py:33-112.  `steps_per_second`: one shift step lasts 1/steps_per_second."""

  def __init__(self, max_shift_steps: int, steps_per_second: float, event_ranges: Sequence[EventRange]):
    self.steps_per_second = steps_per_second
    self._ranges: List[EventRange] = [EventRange('shift', 0, max_shift_steps)] + list(event_ranges)
    names = [r.type for r in self._ranges]
    if len(set(names)) != len(names):                     # event_codec.py:59-61
      raise ValueError('event types must be unique: %s' % names)
    self._first = {}                                      # type -> first index of its block
    self._starts: List[int] = []                          # block starts, ascending
    off = 0
    for r in self._ranges:
      self._first[r.type] = off
      self._starts.append(off)
      off += r.size
    self._num_classes = off
    self._by_type = {r.type: r for r in self._ranges}

  @property
  def num_classes(self) -> int:                           # event_codec.py:63-65
    return self._num_classes

  @property
  def max_shift_steps(self) -> int:                       # event_codec.py:74-76
    return self._ranges[0].max_value

  def is_shift_event_index(self, index: int) -> bool:     # event_codec.py:70-72
    return 0 <= index <= self._ranges[0].max_value

  def encode_event(self, event: Event) -> int:            # event_codec.py:78-90
    r = self._by_type.get(event.type)
    if r is None:
      raise ValueError('Unknown event type: %s' % event.type)
    if not r.min_value <= event.value <= r.max_value:
      raise ValueError('Event value %d is not within valid range [%d, %d] for type %s'
                       % (event.value, r.min_value, r.max_value, event.type))
    return self._first[event.type] + event.value - r.min_value

  def event_type_range(self, event_type: str) -> Tuple[int, int]:   # event_codec.py:92-100
    r = self._by_type.get(event_type)
    if r is None:
      raise ValueError('Unknown event type: %s' % event_type)
    lo = self._first[event_type]
    return lo, lo + r.size - 1

  def decode_event_index(self, index: int) -> Event:      # event_codec.py:102-112
    if not 0 <= index < self._num_classes:
      raise ValueError('Unknown event index: %s' % index)
    b = bisect.bisect_right(self._starts, index) - 1
    r = self._ranges[b]
    return Event(r.type, r.min_value + int(index) - self._starts[b])
