import sys, dataclasses, numpy as np, torch
sys.path.insert(0, '/root/repo')
import msd_amd
from tests import helpers
from oracle import backend, fast
base = msd_amd.config.preset('tiny_context', num_steps=3)
spec = dataclasses.replace(base, task_feature_lengths={'inputs': 1024, 'targets': 64, 'targets_context': 64})
params = msd_amd.synthetic.init_params(spec, 6, norm_scale_jitter=0.1)
model = msd_amd.InferenceModel(params, spec)
nm = model._get_native()
cfg, dc = helpers.oracle_configs(spec)
for valid in [1, 130, 1022]:
  for mask in ['ragged', 'ones', 'zeros']:
    batch = helpers.make_batch(spec, batch=1, ctx_mask=mask)
    toks = batch['encoder_input_tokens']; toks[0, valid:] = 0; toks[0, :valid] = np.maximum(toks[0, :valid], 3); toks[0, valid-1] = 1
    fm = fast.FastModel(backend.NumpyBackend('float64'), cfg, dc, params, True)
    fm.encode(toks, batch['encoder_continuous_inputs'], batch['encoder_continuous_mask'])
    nm.encode(1, toks, torch.as_tensor(batch['encoder_continuous_inputs']).cuda(), batch['encoder_continuous_mask'])
    z = np.random.default_rng(0).standard_normal((1, 64, 128)).astype(np.float32)
    zd = torch.as_tensor(z).cuda()
    for step, cond in [(2, True), (0, True)]:
      eps = torch.zeros_like(zd); nm.decoder_pass(1, step, zd, cond, eps); torch.cuda.synchronize()
      want = fm.decoder_pass(z.astype(np.float64), step, cond)
      err = np.abs(eps.cpu().numpy() - want).max() / np.abs(want).max()
      print('valid', valid, mask, 'nkeys', int((toks>0).sum() + batch['encoder_continuous_mask'].sum()), 'step', step, 'relerr %.2e' % err)
