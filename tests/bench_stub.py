"""CPU stand-in for ``msd_amd.InferenceModel`` used ONLY by tests/test_bench_launch.py: it lets
``bench.py --gpus N --dist-backend gloo --model-factory tests.bench_stub:StubInferenceModel`` run its whole
multi-rank plumbing (self-launch, rendezvous, barriers, the four --mode drivers of sharding.py, the hand-off
probe, the max-over-ranks clock, the one JSON line) in this GPU-less container.  It computes nothing of the
product: a segment is a cheap deterministic function with the real model's DATA DEPENDENCE (its tokens, the seed,
its global index, the previous prediction unless the context is masked)."""
from __future__ import annotations

import time

import numpy as np
import torch


class _StubNative:
  """profile_steps of native.NativeModel: fixed per-class figures (ms total, launches) for `n_steps` steps."""

  def profile_steps(self, batch, n_steps, stream=0):
    per_launch_us = {'gemm_qkv': 10.0, 'attn_self': 8.0, 'gemm_attn_out': 6.0, 'gemm_cross_q': 5.0, 'attn_cross': 12.0,
                     'gemm_cross_out': 6.0, 'gemm_mlp_in_geglu': 14.0, 'gemm_mlp_out': 12.0}
    out = {k: (v * 1e-3 * 12 * n_steps, 12 * n_steps) for k, v in per_launch_us.items()}
    # (in-proj: the cold first launch of a step outlasts the MLP-in launch under eager hipEvents -- it did on the GPU,
    # profiles/r03w_bench_default.json -- and must still not be picked as the roofline's dominant kernel)
    for k, v in (('final_proj_f32', 9.0), ('sampler_step', 5.0), ('in_proj_f32', 19.0)):
      out[k] = (v * 1e-3 * n_steps, n_steps)
    return out


class StubInferenceModel:

  def __init__(self, checkpoint_path, spec, batch_size=1, precision='f16x3', device=None, **_unused):
    self.spec = spec
    self.batch_size = batch_size
    self.precision = precision
    self.device = torch.device('cpu')
    lens = spec.task_feature_lengths
    self.inputs_length, self.targets_length = lens['inputs'], lens['targets']
    self.targets_context_length = lens.get('targets_context') if spec.has_context else None
    self.last_timing = {}
    self._native = _StubNative()
    self.params = {}

  def _get_native(self):
    return self._native

  def _segment(self, toks, seed, gi, prev):
    t = self.targets_length
    base = torch.full((1, t, 128), float(int(np.sum(toks)) % 97) * 0.01 + 0.001 * seed + 0.1 * gi, dtype=torch.float32)
    if prev is not None:
      base = base + 0.5 * torch.as_tensor(prev, dtype=torch.float32)[:, -t:, :]
    return base

  def predict(self, batch, seed=0, segment=0, init_z=None, noise=None, return_torch=False, rng='philox'):
    toks = np.asarray(batch['encoder_input_tokens'])
    t0 = time.perf_counter()
    outs = []
    for b in range(toks.shape[0]):
      prev = None
      if self.targets_context_length is not None and np.asarray(batch['encoder_continuous_mask'])[b].any():
        prev = torch.as_tensor(batch['encoder_continuous_inputs'])[b:b + 1]
      outs.append(self._segment(toks[b], seed, segment, prev))
    out = torch.cat(outs, 0)
    dt = time.perf_counter() - t0
    self.last_timing = {'encode_s': 0.1 * dt, 'sample_s': 0.9 * dt, 'total_s': dt}
    scores = torch.zeros((toks.shape[0],))
    return (out, scores) if return_torch else (out.numpy(), scores.numpy())

  def predict_sequence(self, segments_tokens, seed=0, always_mask_context=False, init_context=None,
                       first_segment_index=0, return_timing=False, rng='philox', return_torch=False):
    prev = None if init_context is None else init_context
    import os
    # test hook (tests/test_bench_launch.py): the rank named here never returns from the hand-off leg's segments
    # (seeds >= 100 are that leg's songs) -- the stand-in for a hung point-to-point transport
    if os.environ.get('MSD_STUB_HANG_RANK') == os.environ.get('RANK') and seed >= 100:
      time.sleep(3600)
    outs = []
    for i, toks in enumerate(segments_tokens):
      masked = always_mask_context or self.targets_context_length is None
      out = self._segment(np.asarray(toks), seed, first_segment_index + i, None if masked else prev)
      prev = out
      outs.append(out)
    full = torch.cat(outs, 1)
    return full if return_torch else full.numpy()
