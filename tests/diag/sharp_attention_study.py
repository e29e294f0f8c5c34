"""Does the single-plane QUERY side of the decoder's attentions survive SHARP (trained-like) attention?

VERDICT r03 item 3: one half plane for Q perturbs a logit by ~|s| 2^-12; every fixture so far has logits of O(1).
`synthetic.sharp_attention(params, gain)` scales every decoder query kernel, i.e. every logit, by `gain`.  For each
gain this script runs the `small` 1000-step segment (identical noise) as

  float64            the yardstick of this gain (fresh: the committed golden belongs to gain 1 only)
  float32            the reference's own arithmetic: the FLOOR any float32-class evaluation sits on
  f16x3              device emulation, hi + lo half planes everywhere (attention.h QP = 0)
  f16_noplo_dec      ... P (softmax weights) as one plane in the decoder's attentions (QP = 2)
  f16_noqlo_dec      ... Q as one plane (QP = 1)
  f16_noqplo_dec     ... both (QP = 3: round 3's default)

and prints rms vs float64 and the ratio to the float32 floor.  The emulation has predicted the device to three digits
on every fixture so far (tests/diag/precision_study.py).  Test infrastructure; changes nothing in the product.

  python -m tests.diag.sharp_attention_study [gain ...]         (default: 1 4 8; ~4 min per run on 8 cores)
  python -m tests.diag.sharp_attention_study --golden 4 8        also writes tests/golden/small_sharp<gain>_n1000.npz"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msd_amd
from oracle import backend, fast, philox
from tests import helpers
from tests.diag import precision_study as ps
from tests.test_golden import GOLD

VARIANTS = ['f16x3', 'f16_noplo_dec', 'f16_noqlo_dec', 'f16_noqplo_dec']
ps.F16_VARIANTS.setdefault('f16_noplo_dec', dict(scale_w=True, mm=[(0, 0), (0, 1), (1, 0)]))
ps.F16_ATT.setdefault('f16_noplo_dec', (ps._X3, ps._NOL, 'dec'))
NOISE_SEED = 20250925


def run(spec, params, batch, init_z, noise, dtype, variant=None):
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend(dtype, threads=backend.effective_cpus())
  if variant is None:
    m = fast.FastModel(xp, cfg, dc, params, spec.has_context)
  else:
    m = ps.StudyModel(xp, cfg, dc, params, spec.has_context, precision='bf16x3')
    m.variant = variant
  return xp.to_numpy(m.predict(batch, init_z, noise)[0]).astype(np.float64)


def logit_spread(spec, params, batch, init_z):
  """|s| statistics of the decoder's attentions at one mid-schedule pass: what 'sharp' means in numbers."""
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend('float64', threads=backend.effective_cpus())
  seen = []

  class Probe(fast.FastModel):
    def _attend(self, q, k, v):
      if getattr(self, '_dec', False):
        s = xp.to_numpy(xp.einsum('qhd,khd->hqk', q, k))
        seen.append((float(np.abs(s).max()), float(np.mean(s.max(-1) - np.median(s, -1))), float(np.mean(np.sort(np.exp(s - s.max(-1, keepdims=True)) / np.exp(s - s.max(-1, keepdims=True)).sum(-1, keepdims=True), -1)[..., -1]))))
      return super()._attend(q, k, v)

    def decoder_pass(self, z, i, cond):
      self._dec = True
      try:
        return super().decoder_pass(z, i, cond)
      finally:
        self._dec = False
  m = Probe(xp, cfg, dc, params, spec.has_context)
  m.encode(batch['encoder_input_tokens'])
  m.decoder_pass(xp.asarray(init_z), 500, True)
  a = np.array(seen)
  return a[:, 0].max(), a[:, 1].mean(), a[:, 2].mean()


def main(argv):
  golden = '--golden' in argv
  gains = [float(a) for a in argv if not a.startswith('--')] or [1.0, 4.0, 8.0]
  spec = msd_amd.config.preset('small', num_steps=1000)
  base = msd_amd.synthetic.init_params(spec, 0)
  t, n = spec.task_feature_lengths['targets'], 128
  batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 0)}
  init_z, noise = philox.segment_noise((1, t, n), 1000, seed=NOISE_SEED, segment=0)
  for gain in gains:
    params = msd_amd.synthetic.sharp_attention(base, gain)
    smax, gap, top = logit_spread(spec, params, batch, init_z)
    print('gain %g: decoder attention logits max |s| %.1f, mean (max - median) %.1f, mean top softmax weight %.3f'
          % (gain, smax, gap, top), flush=True)
    t0 = time.perf_counter()
    ref = run(spec, params, batch, init_z, noise, 'float64')
    print('gain %g  %-16s (%.0f s)' % (gain, 'float64', time.perf_counter() - t0), flush=True)
    f32 = run(spec, params, batch, init_z, noise, 'float32')
    floor = helpers.rms(f32, ref)
    print('gain %g  %-16s rms vs float64 %.3e   x1.00' % (gain, 'float32', floor), flush=True)
    for v in VARIANTS:
      t0 = time.perf_counter()
      out = run(spec, params, batch, init_z, noise, 'float32', v)
      e = helpers.rms(out, ref)
      print('gain %g  %-16s rms vs float64 %.3e   x%.2f   (%.0f s)' % (gain, v, e, e / floor, time.perf_counter() - t0), flush=True)
    if golden:
      np.savez_compressed(os.path.join(GOLD, 'small_sharp%g_n1000.npz' % gain), mel=ref, rms_f32=floor, gain=gain,
                          noise_seed=NOISE_SEED)


if __name__ == '__main__':
  main(sys.argv[1:])
