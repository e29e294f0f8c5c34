"""Multi-process (gloo, world_size 2/3, CPU) coverage of the N>1 host logic in
sharding.py: chained hand-off == sequential run; round-robin dealing; gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import msd_amd
from msd_amd import sharding

C, NDIM, T = 4, 3, 4


def stub_predict_sequence(tokens_list, seed=0, init_context=None, first_segment_index=0,
                          always_mask_context=False):
  """Deterministic stand-in with the same data dependence as the real model:
  segment k depends on its tokens, the seed, its global index and the previous
  prediction (or nothing when the context is masked)."""
  prev = None if init_context is None else np.asarray(init_context, np.float32)
  outs = []
  for i, toks in enumerate(tokens_list):
    gi = first_segment_index + i
    base = np.full((1, T, NDIM), float(np.sum(toks) % 97) + 0.01 * seed + gi, np.float32)
    if prev is not None:
      base = base + 0.5 * prev[:, -T:, :]
    prev = base
    outs.append(base)
  return np.concatenate(outs, 1)


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, n_seg, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    segs = [np.arange(5, dtype=np.int32) + 3 * k for k in range(n_seg)]
    local = sharding.chained_predict(stub_predict_sequence, segs, (1, C, NDIM), rank, world, seed=2)
    full = sharding.gather_song(local, rank, world)
    songs = [[np.arange(4, dtype=np.int32) + j + k for k in range(n_seg)] for j in range(3)]
    wave = sharding.chained_wavefront(stub_predict_sequence, songs, (1, C, NDIM), rank, world, seed=5)
    wave_full = [sharding.gather_song(w, rank, world) for w in wave]
    if rank == 0:
      q.put((full, wave_full))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_seg', [(2, 5), (3, 4), (2, 1)])
def test_chained_handoff_equals_sequential(world, n_seg):
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, n_seg, q)) for r in range(world)]
  for p in procs:
    p.start()
  full, wave_full = q.get(timeout=120)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  segs = [np.arange(5, dtype=np.int32) + 3 * k for k in range(n_seg)]
  np.testing.assert_array_equal(full, stub_predict_sequence(segs, seed=2))
  for j, got in enumerate(wave_full):
    song = [np.arange(4, dtype=np.int32) + j + k for k in range(n_seg)]
    np.testing.assert_array_equal(got, stub_predict_sequence(song, seed=5 + j))


def _wavefront_worker(rank, world, port, n_songs, q, delays):
  """world ranks x n_songs songs of `world` segments; rank `r` sleeps delays[r] seconds inside every segment, so that
  fast ranks run ahead and several sends are in flight at once (what would reorder if order were an accident)."""
  import time
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    def slow_predict(tokens_list, **kw):
      time.sleep(delays[rank])
      return stub_predict_sequence(tokens_list, **kw)
    songs = [[np.arange(4, dtype=np.int32) + 7 * j + k for k in range(world)] for j in range(n_songs)]
    t0 = time.perf_counter()
    wave = sharding.chained_wavefront(slow_predict, songs, (1, C, NDIM), rank, world, seed=11)
    dt = time.perf_counter() - t0
    full = [sharding.gather_song(w, rank, world) for w in wave]
    if rank == 0:
      q.put((full, dt))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('delays', [(0.0, 0.05, 0.0), (0.05, 0.0, 0.02)])
def test_wavefront_world3_four_songs_in_order_and_overlapped(delays):
  """world 3 x 4 songs.  The message header (song, source rank) is checked on arrival, so a
  reordered hand-off raises instead of silently conditioning a song on another song's prediction; the sends are
  asynchronous, so rank 0 does not wait for rank 1 between its songs (its wall time stays near 4 x its own work)."""
  world, n_songs = 3, 4
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_wavefront_worker, args=(r, world, port, n_songs, q, delays)) for r in range(world)]
  for p in procs:
    p.start()
  full, dt0 = q.get(timeout=120)           # a deadlock would time out here
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for j, got in enumerate(full):
    song = [np.arange(4, dtype=np.int32) + 7 * j + k for k in range(world)]
    np.testing.assert_array_equal(got, stub_predict_sequence(song, seed=11 + j))
  if delays[0] == 0.0:                      # rank 0 is fast, rank 1 slow: rank 0 must not be throttled to rank 1's pace
    assert dt0 < n_songs * delays[1] * 0.75 + 1.0


def test_handoff_header_rejects_a_reordered_message():
  payload = torch.arange(C * NDIM, dtype=torch.float32).reshape(1, C, NDIM)
  msg = sharding.pack_handoff(payload, song=2, src_rank=1)
  assert msg.shape == (sharding.HEADER + C * NDIM,)
  np.testing.assert_array_equal(sharding.unpack_handoff(msg, 2, 1, (1, C, NDIM)).numpy(), payload.numpy())
  for song, src, shape in [(3, 1, (1, C, NDIM)), (2, 0, (1, C, NDIM)), (2, 1, (1, C + 1, NDIM))]:
    with pytest.raises(sharding.HandoffError):
      sharding.unpack_handoff(msg, song, src, shape)
  bad = msg.clone()
  bad[0] = 0.0
  with pytest.raises(sharding.HandoffError):
    sharding.unpack_handoff(bad, 2, 1, (1, C, NDIM))


def test_dealing_helpers():
  assert sharding.deal_round_robin(10, 1, 4) == [1, 5, 9]
  assert sorted(sum((sharding.deal_round_robin(10, r, 4) for r in range(4)), [])) == list(range(10))
  chunks = [sharding.contiguous_chunk(118, r, 8) for r in range(8)]
  assert chunks[0][0] == 0 and chunks[-1][1] == 118
  assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
  assert max(b - a for a, b in chunks) - min(b - a for a, b in chunks) <= 1
  assert sharding.contiguous_chunk(3, 5, 8) == (3, 3)  # more ranks than segments: empty chunk


def test_masked_boundary_mode_matches_always_masked_heads():
  segs = [np.arange(5, dtype=np.int32) + k for k in range(6)]
  parts = [sharding.masked_boundary_predict(stub_predict_sequence, segs, r, 3) for r in range(3)]
  got = np.concatenate(parts, 1)
  want = np.concatenate([stub_predict_sequence(segs[a:b], first_segment_index=a)
                         for a, b in [(0, 2), (2, 4), (4, 6)]], 1)
  np.testing.assert_array_equal(got, want)
