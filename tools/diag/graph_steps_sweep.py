"""How many DDPM steps per hipGraph?  (GPU box.)  One model per value of msd_config.graph_steps, 3 timed 1000-step
segments each, same box, same process: the replay overhead of a graph launch is what separates them."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msd_amd
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as helpers

spec = msd_amd.config.preset('base_with_context', num_steps=1000)
batch = helpers.make_batch(spec, batch=1)
init_z, noise = helpers.make_noise(spec, batch=1)
for gs in [int(v) for v in (sys.argv[1:] or ['8', '4', '20', '40', '8'])]:
  model = msd_amd.InferenceModel('synthetic:0', spec, graph_steps=gs)
  model.predict(batch, init_z=init_z, noise=noise)
  torch.cuda.synchronize()
  ts = []
  for _ in range(3):
    t0 = time.perf_counter()
    model.predict(batch, init_z=init_z, noise=noise)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
  print('graph_steps %3d: %s ms per segment' % (gs, ' '.join('%.1f' % t for t in ts)), flush=True)
  del model
  torch.cuda.empty_cache()
