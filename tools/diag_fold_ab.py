"""Diagnostic: folded vs unfolded norm path on the device, same inputs."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import msd_amd
from tests import helpers
from oracle import backend, fast

for preset, steps, batch, mask in [('tiny_context', 1, 2, 'ragged'), ('tiny_context', 6, 2, 'ragged'), ('tiny_context', 6, 1, 'ones'),
                                   ('small_with_context', 20, 2, 'ones')]:
  spec = msd_amd.config.preset(preset, num_steps=steps)
  params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
  b = helpers.make_batch(spec, batch=batch, ctx_mask=mask)
  init_z, noise = helpers.make_noise(spec, batch=batch)
  outs = {}
  for fold in ('1', '0'):
    os.environ['MSD_FOLD_NORM'] = fold
    model = msd_amd.InferenceModel(params, spec, batch_size=batch)
    outs[fold], _ = model.predict(b, init_z=init_z, noise=noise)
  cfg, dc = helpers.oracle_configs(spec)
  o = {}
  for dt in ('float64', 'float32'):
    xp = backend.TorchBackend(dt)
    o[dt] = xp.to_numpy(fast.FastModel(xp, cfg, dc, params, True).predict(b, init_z, noise)[0]).astype(np.float64)
  print('%-18s steps %2d batch %d %-6s: fold-vs-unfold %.3e | vs f64: fold %.3e unfold %.3e f32-oracle %.3e' % (
      preset, steps, batch, mask, helpers.rms(outs['1'], outs['0']), helpers.rms(outs['1'], o['float64']),
      helpers.rms(outs['0'], o['float64']), helpers.rms(o['float32'], o['float64'])))
