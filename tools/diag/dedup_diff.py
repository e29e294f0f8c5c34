"""Where do two launch structures of the same step part ways?  One DDPM step (msd_profile_steps: eager launches, fixed
z) of a ONE-decoder-layer model with msd_config.dedup_layer0 on and off; every internal buffer msd_debug_read exposes
is compared, per CFG pass.  (Round 5: found why the first version of the layer-0 de-duplication was not bit-exact.)

  python tools/diag/dedup_diff.py [preset] [layers] [songs]"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import msd_amd
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as helpers

preset = sys.argv[1] if len(sys.argv) > 1 else 'tiny_context'
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
base = msd_amd.config.preset(preset, num_steps=4)
spec = dataclasses.replace(base, t5=dataclasses.replace(base.t5, num_decoder_layers=layers))
params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
batch = helpers.make_batch(spec, batch=nb)
bufs = {}
for dedup in (False, None):
  model = msd_amd.InferenceModel(params, spec, batch_size=nb, dedup_layer0=dedup)
  nm = model._get_native()
  toks = batch['encoder_input_tokens']
  if spec.has_context:
    nm.encode(nb, toks, torch.as_tensor(batch['encoder_continuous_inputs']).cuda(), batch['encoder_continuous_mask'])
  else:
    nm.encode(nb, toks)
  nm.profile_steps(nb, 1)
  torch.cuda.synchronize()
  bufs[dedup] = {b: nm.debug_read(b) for b in ('qk', 'vt', 'ao', 'x', 'ssq', 'y', 'g', 'eps', 'z')}
  for b in ('xg', 'qp', 'cq'):   # (the folded cross-attention query projection's buffers: conditional rows only)
    try:
      bufs[dedup][b] = nm.debug_read(b)
    except Exception:
      pass
  del model
t = spec.task_feature_lengths['targets']
for b in bufs[False]:
  a, c = bufs[False][b], bufs[None][b]
  if b in ('z', 'xg', 'qp', 'cq'):
    print('%-4s max |diff| %.3e' % (b, np.abs(a - c).max()))
    continue
  a2, c2 = a.reshape(2, -1), c.reshape(2, -1)   # [pass][...] (one song)
  print('%-4s pass 0: max |diff| %.3e (%d of %d differ) | pass 1: %.3e (%d differ) | off: pass 0 == pass 1 ? %s'
        % (b, np.abs(a2[0] - c2[0]).max(), (a2[0] != c2[0]).sum(), a2[0].size, np.abs(a2[1] - c2[1]).max(), (a2[1] != c2[1]).sum(),
           bool(np.array_equal(a2[0], a2[1]))))
