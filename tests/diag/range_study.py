"""How much operand range do the half planes have to spare?  CPU study with the oracle's emulation (test
infrastructure): the `small` model with weights reshaped by synthetic.trained_like at growing outlier gains, 100
DDPM steps, error of the f16x3 and bf16x3 emulations against float64 and the largest |activation| / |weight|
that reaches a GEMM or attention operand plane (half saturates at 65504; weights are packed times 2^9, |w| < 128).
  python -m tests.diag.range_study [gain ...]
  python -m tests.diag.range_study folded [preset ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msd_amd
from oracle import backend, fast, philox
from tests import helpers


class Probe(fast.FastModel):
  top_a = 0.0
  top_w = 0.0

  def _split(self, a):
    m = float(abs(a).max()) if a.numel() else 0.0
    Probe.top_a = max(Probe.top_a, m)
    return super()._split(a)

  def _w(self, name):
    if name not in self._wcache:
      Probe.top_w = max(Probe.top_w, float(np.abs(self.xp.to_numpy(self.p[name])).max()))
    return super()._w(name)


def main(gains):
  steps = 100
  spec = msd_amd.config.preset('small', num_steps=steps)
  base = msd_amd.synthetic.init_params(spec, 0)
  cfg, dc = helpers.oracle_configs(spec)
  t, n = spec.task_feature_lengths['targets'], 128
  batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 0)}
  init_z, noise = philox.segment_noise((1, t, n), steps, seed=0, segment=0)
  for gain in gains:
    params = msd_amd.synthetic.trained_like(base, seed=1, outlier_gain=gain, channel_sigma=0.5 if gain < 100 else 0.8)
    t0 = time.time()
    x64 = backend.TorchBackend('float64')
    ref = x64.to_numpy(fast.FastModel(x64, cfg, dc, params, False).predict(batch, init_z, noise)[0])
    row = []
    for prec in ('f32', 'f16x3', 'bf16x3'):
      xp = backend.TorchBackend('float32')
      Probe.top_a = Probe.top_w = 0.0
      out = xp.to_numpy(Probe(xp, cfg, dc, params, False, precision=prec).predict(batch, init_z, noise)[0])
      row.append('%s %.2e' % (prec, helpers.rms(out, ref)))
    print('outlier gain x%-5g  rms vs float64: %s | largest operand: activation %.3g, weight %.3g   (%.0f s)'
          % (gain, '  '.join(row), Probe.top_a, Probe.top_w, time.time() - t0), flush=True)


def folded_plane_range(presets):
  """The folded-norm operand planes of the device hold y = x (.) gamma (.) (film_scale + 1) -- the RAW residual
  stream times the gain, the 1/rms comes after the GEMM -- so what has to fit the half range is the
  stream itself, not its normalised value.  Largest |x (.) gamma| at any RMSNorm of a 100-step run, per preset."""
  from oracle import ops
  steps = 100
  for name in presets:
    spec = msd_amd.config.preset(name, num_steps=steps)
    params = msd_amd.synthetic.init_params(spec, 0)
    cfg, dc = helpers.oracle_configs(spec)
    t, n = spec.task_feature_lengths['targets'], 128
    batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 0)}
    if spec.has_context:
      c = spec.task_feature_lengths['targets_context']
      batch['encoder_continuous_inputs'] = np.random.default_rng(0).uniform(-11, 4, (1, c, n)).astype(np.float32)
      batch['encoder_continuous_mask'] = np.ones((1, c), np.int32)
    init_z, noise = philox.segment_noise((1, t, n), steps, seed=0, segment=0)
    top = {'fold': 0.0, 'stream': 0.0}
    orig = ops.rms_layer_norm

    def probe(xp, x, scale, epsilon=1e-6):
      top['stream'] = max(top['stream'], float(abs(x).max()))
      top['fold'] = max(top['fold'], float(abs(x * scale).max()))
      return orig(xp, x, scale, epsilon)
    ops.rms_layer_norm = probe
    try:
      xp = backend.TorchBackend('float32')
      fast.FastModel(xp, cfg, dc, params, spec.has_context).predict(batch, init_z, noise)
    finally:
      ops.rms_layer_norm = orig
    print('%-18s largest |residual stream| %.3g, largest |x (.) gamma| %.3g -> %.0fx below the half range (65504)'
          % (name, top['stream'], top['fold'], 65504.0 / top['fold']), flush=True)


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == 'folded':
    folded_plane_range(sys.argv[2:] or ['small', 'base_with_context'])
  else:
    main([float(a) for a in sys.argv[1:]] or [6.0, 50.0, 400.0, 3000.0])
