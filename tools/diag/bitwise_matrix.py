"""Run-to-run and knob-to-knob bit identity of one sampled segment (one process, fresh model per run).
  python tools/diag/bitwise_matrix.py [preset] [songs] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import msd_amd
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as helpers

preset = sys.argv[1] if len(sys.argv) > 1 else 'base_with_context'
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
spec = msd_amd.config.preset(preset, num_steps=steps)
params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
batch = helpers.make_batch(spec, batch=nb)
rng = np.random.default_rng(5)
t = spec.task_feature_lengths['targets']
init_z = rng.standard_normal((nb, t, 128)).astype(np.float32)
noise = rng.standard_normal((steps, nb, t, 128)).astype(np.float32)
runs = [('default', {}), ('default again', {}), ('dedup off', dict(dedup_layer0=False)), ('dedup off again', dict(dedup_layer0=False)),
        ('graph_steps 1', dict(graph_steps=1)), ('graph_steps 1, dedup off', dict(graph_steps=1, dedup_layer0=False)),
        ('fold off', dict(cross_q_fold=False)), ('fold off, dedup off', dict(cross_q_fold=False, dedup_layer0=False)),
        ('merge launch', dict(cross_merge_in_launch=False)), ('merge launch, dedup off', dict(cross_merge_in_launch=False, dedup_layer0=False)),
        ('prefetch off', dict(weight_prefetch=False)), ('prefetch off, dedup off', dict(weight_prefetch=False, dedup_layer0=False))]
outs = {}
for name, kw in runs:
  kw = {k: v for k, v in kw.items()}
  gs = kw.pop('graph_steps', None)
  try:
    model = msd_amd.InferenceModel(params, spec, batch_size=nb, **kw) if gs is None else msd_amd.InferenceModel(params, spec, batch_size=nb, graph_steps=gs, **kw)
  except TypeError as e:
    print(name, 'skipped:', e)
    continue
  got, _ = model.predict(batch, init_z=init_z, noise=noise)
  outs[name] = np.asarray(got)
  del model
ref = outs['default']
for name, got in outs.items():
  d = np.abs(got - ref)
  print('%-28s vs default: max |diff| %.3e, %d of %d elements differ' % (name, d.max(), int((got != ref).sum()), got.size))
for a, b in (('fold off', 'fold off, dedup off'), ('merge launch', 'merge launch, dedup off'), ('prefetch off', 'prefetch off, dedup off'),
             ('graph_steps 1', 'graph_steps 1, dedup off')):
  if a in outs and b in outs:
    print('%-28s vs %-28s: %d differ' % (a, b, int((outs[a] != outs[b]).sum())))
