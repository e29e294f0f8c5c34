"""Diagnostic: folded vs unfolded norm path on the device, same inputs, clip_x0 off and one
DDPM step so that the output is LINEAR in the CFG-combined eps (no clip-boundary chaos)."""
import dataclasses, os, sys
import numpy as np
sys.path.insert(0, '.')
import msd_amd
from tests import helpers
from oracle import backend, fast

for preset, batch, mask, w in [('tiny_context', 1, 'ones', 5.0), ('tiny_context', 2, 'ragged', 5.0), ('tiny_context', 2, 'ragged', 1.0),
                               ('tiny', 2, 'ones', 5.0), ('small_with_context', 2, 'ones', 5.0)]:
  spec = msd_amd.config.preset(preset, num_steps=1, cfg_weight=w)
  d = spec.diffusion
  spec = dataclasses.replace(spec, diffusion=dataclasses.replace(d, sampler=dataclasses.replace(d.sampler, clip_x0=False)))
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
    o[dt] = xp.to_numpy(fast.FastModel(xp, cfg, dc, params, spec.has_context).predict(b, init_z, noise)[0]).astype(np.float64)
  sc = np.sqrt(np.mean(o['float64'] ** 2))
  print('%-18s batch %d %-6s w %.0f: REL rms fold-vs-unfold %.3e | vs f64: fold %.3e unfold %.3e f32-oracle %.3e | per-elem fold %s' % (
      preset, batch, mask, w, helpers.rms(outs['1'], outs['0']) / sc, helpers.rms(outs['1'], o['float64']) / sc,
      helpers.rms(outs['0'], o['float64']) / sc, helpers.rms(o['float32'], o['float64']) / sc,
      ['%.1e' % (helpers.rms(outs['1'][i], o['float64'][i]) / sc) for i in range(batch)]))
