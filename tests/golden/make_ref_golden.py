"""Fixtures produced by the REFERENCE's own code: tests/golden/ref_*.npz.

Runs only in the build container (needs /root/reference); see ref_shim.py for how the reference's
layers.py / network.py / diffusion_utils.py / models.py execute without JAX.  For every case of
tests/ref_cases.py:

    model = models.{Context,}DiffusionModel(module=network.{ContinuousContext,}Transformer(T5Config(..)),
                                            diffusion_config=DiffusionConfig(..), audio_codec=MelGAN())
    mel, _ = model.predict_batch_with_aux(params, batch, rng)           # models.py:149-205 / 340-400

in float64, with jax.random.normal served from the case's seeded noise (init_z for the scan key,
noise[i] for fold_in(rng, i): diffusion_utils.py:389-390,462).  Stored per case: the mel output, (for
three cases) the encodings, one conditional and one unconditional decoder pass, a digest of the inputs, and the
parameter tree (names + shapes) the reference's OWN `module.init` creates -- which the package's
synthetic / checkpoint tree must equal.

    python tests/golden/make_ref_golden.py [--add-f32] [case ...]      # default: all cases
    (--add-f32: only the float32 pass, added to existing files)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from tests import ref_cases  # noqa: E402


def nest(flat):
  out = {}
  for k, v in flat.items():
    node = out
    parts = k.split('/')
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = np.asarray(v, np.float64)
  return out


def flatten(tree, prefix=''):
  out = {}
  for k, v in tree.items():
    if isinstance(v, dict):
      out.update(flatten(v, prefix + k + '/'))
    else:
      out[prefix + k] = v
  return out


def reference_model(ref, spec):
  """The reference's objects for a package ModelSpec (gin would build the same: base_with_context.gin)."""
  t5, d = spec.t5, spec.diffusion
  du, net = ref.diffusion_utils, ref.network
  cfg = net.T5Config(
      vocab_size=t5.vocab_size, emb_dim=t5.emb_dim, num_heads=t5.num_heads,
      num_encoder_layers=t5.num_encoder_layers, num_decoder_layers=t5.num_decoder_layers, head_dim=t5.head_dim,
      mlp_dim=t5.mlp_dim, mlp_activations=tuple(t5.mlp_activations), dropout_rate=t5.dropout_rate,
      max_decoder_noise_time=t5.max_decoder_noise_time, decoder_cross_attend_style=t5.decoder_cross_attend_style,
      position_encoding=t5.position_encoding, context_positions=t5.context_positions)

  def sched(s):
    return du.DiffusionSchedule(name=s.name, start=s.start, stop=s.stop, num_steps=s.num_steps)
  dc = du.DiffusionConfig(
      train_schedule=sched(d.train_schedule), model_output=d.model_output,
      classifier_free_guidance=du.ClassifierFreeGuidanceConfig(
          eval_condition_weight=d.classifier_free_guidance.eval_condition_weight),
      sampler=du.SamplerConfig(name=d.sampler.name, schedule=sched(d.sampler.schedule), clip_x0=d.sampler.clip_x0,
                               logvar_type=d.sampler.logvar_type))
  codec = ref.audio_codecs.MelGAN()
  if spec.has_context:
    return ref.models.ContextDiffusionModel(module=net.ContinuousContextTransformer(config=cfg),
                                            diffusion_config=dc, audio_codec=codec)
  return ref.models.DiffusionModel(module=net.Transformer(config=cfg), diffusion_config=dc, audio_codec=codec)


def predict(ref, name, wide):
  """The reference's predict_batch_with_aux on the case's inputs; wide: float64, else float32 arrays."""
  import jax  # the stand-in
  spec, params, batch, init_z, noise = ref_cases.inputs(name)
  ref_shim.WIDE = wide
  model = reference_model(ref, spec)
  tree = nest(params)
  b = {k: np.asarray(v, np.float64) if np.asarray(v).dtype.kind == 'f' else np.asarray(v) for k, v in batch.items()}
  root = jax.random.PRNGKey(0)

  def provider(path, shape):
    if path == root.path:
      assert tuple(shape) == init_z.shape
      return init_z
    assert path[:-2] == root.path and path[-2] == 'fold_in', path
    return noise[path[-1]]
  ref_shim.noise_provider = provider
  ref_shim.reset_accessed()
  mel, _ = model.predict_batch_with_aux(tree, b, rng=root)
  assert np.asarray(mel).dtype == (np.float64 if wide else np.float32)
  return model, tree, b, root, np.asarray(mel)


def add_f32(ref, name):
  """Second pass: the same reference statements over float32 NumPy arrays (`mel_f32`): how far float32
  arithmetic moves THIS case away from the float64 answer -- the yardstick of the device test."""
  path = os.path.join(HERE, 'ref_%s.npz' % name)
  g = dict(np.load(path))
  t0 = time.time()
  g['mel_f32'] = predict(ref, name, wide=False)[4]
  ref_shim.WIDE = True
  np.savez_compressed(path, **g)
  print('%-32s %6.1fs  float32 pass: rms vs float64 %.2e' % (
      name, time.time() - t0, float(np.sqrt(np.mean((g['mel_f32'].astype(np.float64) - g['mel']) ** 2)))), flush=True)


def run_case(ref, name):
  import jax  # the stand-in
  spec, params, batch, init_z, noise = ref_cases.inputs(name)
  t0 = time.time()
  model, tree, b, root, mel = predict(ref, name, wide=True)
  dt = time.time() - t0
  unused = sorted(set(params) - ref_shim.accessed)
  assert not unused, 'the reference never read %s' % unused
  assert ref_shim.accessed <= set(params)

  # pieces: encodings, one conditional / one unconditional decoder pass at fixed (z, time)
  module = model.module
  if spec.has_context:
    ctx = model.audio_codec.scale_features(b['encoder_continuous_inputs'], output_range=[-1., 1.], clip=True)
    enc = module.apply({'params': tree}, input_tokens=b['encoder_input_tokens'], continuous_inputs=ctx,
                       continuous_mask=b['encoder_continuous_mask'], enable_dropout=False, method=module.encode)
  else:
    enc = module.apply({'params': tree}, encoder_input_tokens=b['encoder_input_tokens'], enable_dropout=False,
                       method=module.encode)
  z = ref_cases.pass_z(init_z.shape)
  steps = spec.diffusion.sampler.schedule.num_steps
  step = steps // 2
  tm = np.full((init_z.shape[0],), (step + 1.0) / steps)
  passes = {}
  for cond in (1, 0):
    e = jax.tree.map(lambda x: x * bool(cond), enc)
    kw = (dict(input_tokens=z, noise_time=tm) if spec.has_context else
          dict(decoder_input_tokens=z, decoder_noise_time=tm))    # network.py:561-566 / 484-489
    passes[cond] = module.apply({'params': tree}, encodings_and_masks=e, enable_dropout=False,
                                method=module.decode, **kw)

  # the tree the reference's own init creates (names + shapes; values are placeholders except the
  # sinusoidal tables of position_encoding='fixed', which are deterministic)
  init_tree = flatten(module.init(
      root, **({'encoder_input_tokens': b['encoder_input_tokens'],
                'encoder_continuous_inputs': b['encoder_continuous_inputs'],
                'encoder_continuous_mask': b['encoder_continuous_mask']} if spec.has_context else
               {'encoder_input_tokens': b['encoder_input_tokens']}),
      decoder_input_tokens=z, decoder_noise_time=tm, enable_dropout=False)['params'])
  shapes = {k: tuple(v.shape) for k, v in init_tree.items()}
  mine = {k: tuple(v.shape) for k, v in params.items()}
  assert shapes == mine, ('parameter trees differ', sorted(set(shapes.items()) ^ set(mine.items()))[:8])

  out = dict(mel=np.asarray(mel), pass_step=step, pass_cond=np.asarray(passes[1]),
             pass_uncond=np.asarray(passes[0]), digest=ref_cases.digest(params, batch, init_z, noise),
             tree_names=np.array(sorted(shapes)), tree_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
  if spec.t5.position_encoding == 'fixed':   # layers.sinusoidal() without permutation / offsets is deterministic
    out['decoder_position_table'] = np.asarray(init_tree['decoder/Embed_0/embedding'])
  if name in ref_cases.WITH_ENCODINGS:   # random float64 does not compress: encodings for three cases only
    for i, (e, m) in enumerate(enc):
      out['enc%d' % i] = np.asarray(e)
      out['mask%d' % i] = np.asarray(m).astype(np.int8)
  np.savez_compressed(os.path.join(HERE, 'ref_%s.npz' % name), **out)
  print('%-32s %6.1fs  mel %s rms %.3f  digest %s' % (name, dt, np.asarray(mel).shape,
                                                      float(np.sqrt(np.mean(np.asarray(mel) ** 2))), out['digest']), flush=True)


def main():
  ref = ref_shim.load_models()
  args = sys.argv[1:]
  only_f32 = '--add-f32' in args
  names = [a for a in args if not a.startswith('--')] or list(ref_cases.cases())
  for n in names:
    if not only_f32:
      run_case(ref, n)
    add_f32(ref, n)


if __name__ == '__main__':
  main()
