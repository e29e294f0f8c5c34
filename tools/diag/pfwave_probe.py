"""Debug probe (GPU box): one base_with_context decoder pass + a 3-step sample on the library named by MSD_AMD_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import msd_amd
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as helpers   # (not tests.helpers: that imports oracle/)
spec = msd_amd.config.preset('base_with_context', num_steps=3)
model = msd_amd.InferenceModel('synthetic:0', spec)
batch = helpers.make_batch(spec)
init_z, noise = helpers.make_noise(spec)
print('lib', os.environ.get('MSD_AMD_LIB'), flush=True)
got, _ = model.predict(batch, init_z=init_z, noise=noise)
torch.cuda.synchronize()
print('predict ok', float(np.abs(got).mean()), flush=True)
