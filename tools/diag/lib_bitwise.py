"""Do two builds of the library compute the same bits?  (GPU box.)  Runs one base_with_context decoder pass and an
8-step sample on the in-tree library and on $MSD_AMD_LIB_B (each in its own process: one library per process),
prints the sha256 of the outputs.

usage: python tools/diag/lib_bitwise.py            # prints digests for the library named by MSD_AMD_LIB (or in-tree)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import msd_amd
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as helpers   # (not tests.helpers: that imports oracle/)

out = []
for preset, nb in (('base_with_context', 1), ('small', 1), ('base_with_context', 8)):
  spec = msd_amd.config.preset(preset, num_steps=8)
  model = msd_amd.InferenceModel('synthetic:0', spec, batch_size=nb)
  batch = helpers.make_batch(spec, batch=nb)
  init_z, noise = helpers.make_noise(spec, batch=nb)
  got, _ = model.predict(batch, init_z=init_z, noise=noise)
  torch.cuda.synchronize()
  out.append('%s x%d %s' % (preset, nb, hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()[:16]))
print('lib=%s | %s' % (os.path.basename(os.environ.get('MSD_AMD_LIB', 'in-tree')), ' | '.join(out)))
