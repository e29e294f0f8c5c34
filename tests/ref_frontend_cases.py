"""Seeded random note material for tests/golden/ref_frontend.npz (the reference's own note_sequences.py /
run_length_encoding.py / event_codec.py / vocabularies.py executed over stand-ins of note_seq / tensorflow /
seqio: tests/golden/make_ref_frontend_golden.py).  Shared by the generator and tests/test_ref_frontend.py."""
import numpy as np

N_CASES = 48
FRAME_RATE = 50.0          # 16 kHz / hop 320 (audio_codecs.py:204-210)
SEGMENT_FRAMES = 256


def case(i):
  """-> dict(onsets, offsets, pitches, velocities, programs, is_drums, num_velocity_bins, n_frames)"""
  rng = np.random.default_rng(1000 + i)
  n = int(rng.integers(1, 70))
  dur = float(rng.choice([3.0, 7.5, 12.0]))
  # times on a 1 ms grid plus a few exact frame / step boundaries (ties between onsets and offsets)
  onsets = np.round(rng.uniform(0, dur, n), 3)
  if i % 3 == 0:
    onsets[: n // 3] = np.round(onsets[: n // 3] * 100) / 100       # 10 ms codec steps
  lengths = np.round(rng.choice([0.01, 0.05, 0.12, 0.5, 1.0, 2.5], n) * rng.uniform(0.8, 1.2, n), 3)
  lengths = np.maximum(lengths, 0.005)
  pitches = rng.integers(21, 109, n)
  velocities = rng.integers(1, 128, n)
  programs = rng.choice([0, 0, 0, 24, 40, 73], n) if i % 2 else np.zeros(n, int)
  is_drums = (rng.random(n) < (0.15 if i % 4 == 1 else 0.0))
  n_frames = int(np.ceil((float((onsets + lengths).max()) + 0.3) * FRAME_RATE))
  return dict(onsets=onsets, offsets=onsets + lengths, pitches=pitches.astype(int), velocities=velocities.astype(int),
              programs=np.asarray(programs).astype(int), is_drums=is_drums.astype(bool),
              num_velocity_bins=127 if i % 5 == 0 else 1, n_frames=n_frames)
