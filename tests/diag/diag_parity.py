"""Diagnostic: device vs float64/float32 oracle on short chains (not a test)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import msd_amd
from tests import helpers
from oracle import backend, fast

import itertools
cases = [('tiny_context', 1, b, m, 5.0) for b, m in itertools.product((1, 2), ('ones', 'ragged', 'zeros'))]
cases += [('tiny', 1, 2, 'ones', 5.0), ('tiny', 1, 1, 'ones', 5.0)]
for preset, steps, batch, mask, w in cases:
  spec = msd_amd.config.preset(preset, num_steps=steps, cfg_weight=w)
  params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
  model = msd_amd.InferenceModel(params, spec, batch_size=batch)
  b = helpers.make_batch(spec, batch=batch, ctx_mask=mask)
  init_z, noise = helpers.make_noise(spec, batch=batch)
  got, _ = model.predict(b, init_z=init_z, noise=noise)
  cfg, dc = helpers.oracle_configs(spec)
  out = {}
  for dt in ('float64', 'float32'):
    xp = backend.TorchBackend(dt)
    out[dt] = xp.to_numpy(fast.FastModel(xp, cfg, dc, params, spec.has_context).predict(b, init_z, noise)[0]).astype(np.float64)
  print('%-13s steps %d batch %d mask %-6s w %.0f: device %.3e  f32-oracle %.3e   (rms vs f64, mel units)' % (
      preset, steps, batch, mask, w, helpers.rms(got, out['float64']), helpers.rms(out['float32'], out['float64'])),
      ' per-elem:', ['%.1e' % helpers.rms(got[i], out['float64'][i]) for i in range(batch)])
