"""Run the base_with_context golden song several times on the device: prints the rms against the
float64 fixture and whether repeated runs are bit-identical (they must be: no atomics on the path;
a difference means a race in a kernel).  GPU only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msd_amd
from tests import helpers
from tests.test_golden import GOLD
from oracle import philox

preset = sys.argv[1] if len(sys.argv) > 1 else 'base_with_context'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fix = {'base_with_context': 'base_with_context_n1000.npz', 'small': 'small_n1000.npz'}[preset]
g = np.load(os.path.join(GOLD, fix))
spec = msd_amd.config.preset(preset, num_steps=1000)
model = msd_amd.InferenceModel('synthetic:0', spec)
t, n = spec.task_feature_lengths['targets'], 128
c = spec.task_feature_lengths.get('targets_context')
first = None
for rep in range(reps):
  pred = np.zeros((1, c or 0, n), np.float32)
  outs = []
  for k in range(int(g['n_segments'])):
    batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, k)}
    if spec.has_context:
      batch['encoder_continuous_inputs'] = pred
      batch['encoder_continuous_mask'] = (np.zeros if k == 0 else np.ones)((1, c), np.int32)
    init_z, noise = philox.segment_noise((1, t, n), 1000, seed=int(g['noise_seed']), segment=k)
    pred, _ = model.predict(batch, init_z=init_z, noise=noise)
    outs.append(pred)
  got = np.concatenate(outs, 1)
  errs = [helpers.rms(got[:, i * t:(i + 1) * t], g['mel'][:, i * t:(i + 1) * t]) for i in range(int(g['n_segments']))]
  if first is None:
    first = got
  print('run %d: rms per segment %s; max |diff| vs run 0 = %.3e; identical=%s' % (
      rep, ['%.3e' % e for e in errs], np.abs(got - first).max(), np.array_equal(got, first)), flush=True)
