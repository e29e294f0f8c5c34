"""-m gpu: the chained context hand-off (BASELINE config 4; beam/evaluation.py:191-223) with the REAL
model on the device: one song cut across two ranks must be BIT-identical to the sequential song.

  * two processes on the one GPU of the test box, `sharding.chained_predict` over
    `InferenceModel.predict_sequence`, message over torch.distributed (gloo here: RCCL refuses two ranks on
    one device; on a multi-GPU node the same code path runs with comm_device=cuda over RCCL/xGMI,
    bench.py --mode chained|wavefront);
  * two handles in one process with the hand-off message kept in device memory
    (`predict_sequence(return_torch=True)` -> `init_context=<cuda tensor>`).
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _spec(preset, steps):
  import msd_amd
  return msd_amd.config.preset(preset, num_steps=steps)


def _segments(spec, n):
  import msd_amd
  return [msd_amd.synthetic.segment_tokens(spec, k, min_len=8, max_len=min(400, spec.task_feature_lengths['inputs'] - 2))
          for k in range(n)]


def _worker(rank, world, port, preset, steps, n_seg, q):
  import torch
  import torch.distributed as dist
  import msd_amd
  from msd_amd import sharding
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    spec = _spec(preset, steps)
    model = msd_amd.InferenceModel('synthetic:0', spec)
    segs = _segments(spec, n_seg)
    c = spec.task_feature_lengths['targets_context']
    local = sharding.chained_predict(model.predict_sequence, segs, (1, c, 128), rank, world, seed=4)
    full = sharding.gather_song(local, rank, world)
    wave = sharding.chained_wavefront(model.predict_sequence, [segs[:3], segs[1:]], (1, c, 128), rank, world, seed=9)
    wave_full = [sharding.gather_song(w, rank, world) for w in wave]
    if rank == 0:
      seq = model.predict_sequence(segs, seed=4)
      wseq = [model.predict_sequence(segs[:3], seed=9), model.predict_sequence(segs[1:], seed=10)]
      q.put((full, seq, wave_full, wseq))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('preset,steps,n_seg', [('tiny_context', 8, 5), ('base_with_context', 12, 4)])
def test_two_ranks_one_song_bit_identical_to_sequential(preset, steps, n_seg):
  import torch.multiprocessing as mp
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, preset, steps, n_seg, q)) for r in range(2)]
  for p in procs:
    p.start()
  import queue
  import time
  result, deadline = None, time.time() + 600
  try:
    while result is None:
      try:
        result = q.get(timeout=2)
      except queue.Empty:
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        assert not dead, 'a rank died with exit code %s before delivering its result' % dead
        assert time.time() < deadline, 'ranks did not finish in 600 s'
  finally:
    for p in procs:
      p.join(timeout=60)
      if p.is_alive():
        p.kill()
  full, seq, wave_full, wseq = result
  assert all(p.exitcode == 0 for p in procs)
  assert full.shape == seq.shape and np.isfinite(seq).all() and seq.std() > 0.1
  np.testing.assert_array_equal(full, seq)       # bit-for-bit: same kernels, same inputs, same noise keys
  for got, want in zip(wave_full, wseq):
    np.testing.assert_array_equal(got, want)


def test_device_resident_handoff_between_two_handles():
  """The hand-off message stays a device tensor: handle A's last prediction -> handle B's context."""
  import torch
  import msd_amd
  spec = _spec('tiny_context', 8)
  a = msd_amd.InferenceModel('synthetic:0', spec)
  b = msd_amd.InferenceModel('synthetic:0', spec)
  segs = _segments(spec, 5)
  c = spec.task_feature_lengths['targets_context']
  seq = a.predict_sequence(segs, seed=1)
  head = a.predict_sequence(segs[:2], seed=1, return_torch=True)
  assert isinstance(head, torch.Tensor) and head.is_cuda
  msg = head[:, -c:, :].contiguous()
  tail = b.predict_sequence(segs[2:], seed=1, init_context=msg, first_segment_index=2, return_torch=True)
  got = torch.cat([head, tail], dim=1).cpu().numpy()
  np.testing.assert_array_equal(got, seq)
  # and the masked-boundary mode differs exactly from the cut on (the reference's i == 0 behaviour there)
  from msd_amd import sharding
  parts = [sharding.masked_boundary_predict(a.predict_sequence, segs, r, 2, seed=1) for r in range(2)]
  cut = sharding.contiguous_chunk(len(segs), 1, 2)[0] * spec.task_feature_lengths['targets']
  masked = np.concatenate(parts, 1)
  np.testing.assert_array_equal(masked[:, :cut], seq[:, :cut])
  assert np.abs(masked[:, cut:] - seq[:, cut:]).max() > 1e-3
