"""Diagnostic: single decoder-pass error (device vs float64 oracle), cond and uncond."""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
import msd_amd
from tests import helpers
from oracle import backend, fast

for preset in ('tiny_context', 'small_with_context'):
  spec = msd_amd.config.preset(preset, num_steps=6)
  params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
  model = msd_amd.InferenceModel(params, spec, batch_size=2)
  nm = model._get_native()
  b = helpers.make_batch(spec, batch=2, ctx_mask='ragged')
  cfg, dc = helpers.oracle_configs(spec)
  res = {}
  for dt in ('float64', 'float32'):
    xp = backend.TorchBackend(dt)
    fm = fast.FastModel(xp, cfg, dc, params, True)
    fm.encode(b['encoder_input_tokens'], b['encoder_continuous_inputs'], b['encoder_continuous_mask'])
    res[dt] = fm
  nm.encode(2, b['encoder_input_tokens'], torch.as_tensor(b['encoder_continuous_inputs']).cuda(), b['encoder_continuous_mask'])
  t = spec.task_feature_lengths['targets']
  z = np.random.default_rng(0).standard_normal((2, t, 128)).astype(np.float32)
  zd = torch.as_tensor(z).cuda()
  for step, cond in [(5, True), (3, True), (0, True), (3, False)]:
    eps = torch.zeros_like(zd)
    nm.decoder_pass(2, step, zd, cond, eps)
    torch.cuda.synchronize()
    w64 = res['float64'].xp.to_numpy(res['float64'].decoder_pass(res['float64'].xp.asarray(z), step, cond))
    w32 = res['float32'].xp.to_numpy(res['float32'].decoder_pass(res['float32'].xp.asarray(z), step, cond)).astype(np.float64)
    sc = np.sqrt(np.mean(w64 ** 2))
    print('%-18s step %d cond %d: rel rms device %.3e  f32-oracle %.3e' % (
        preset, step, cond, helpers.rms(eps.cpu().numpy(), w64) / sc, helpers.rms(w32, w64) / sc))
