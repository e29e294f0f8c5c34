"""Generate the committed golden fixtures with the ORACLE (float64, torch-CPU).

  python tests/golden/make_golden.py [tiny] [small] [base]

JAX/T5X cannot be imported in this environment, so the goldens
come from the oracle restatement -- pinned itself by the reference's
layers_test.py known-answer tests (tests/test_oracle_kat.py) and by the
faithful-vs-fast equivalence tests -- not from the reference binary.  Every
input is regenerated from seeds by the tests (weights: synthetic.init_params(spec,
seed); tokens: synthetic.segment_tokens; noise: oracle/philox.py), so a fixture
holds only the expected mel output and the seeds.

  tiny_context_n6.npz        tiny_context preset, 6 steps, batch 2, ragged context
  small_n1000.npz            BASELINE config 2 shape: small/no-context, 1000 steps, 1 segment
  base_with_context_n1000.npz  BASELINE config 3 shape: 2 chained segments, 1000 steps
  base_chain_n1000.npz       the same song continued to 6 chained segments (float64), plus the float32
                             oracle's OWN chained run of the same song as the yardstick (`mel_f32`,
                             per-segment `rms_f32`): how far the reference's arithmetic itself drifts
                             along the chain.  Two processes: `chain64` and `chain32` (hours of CPU);
                             `chainpack` merges them.  Saved after every segment.
  base_trained_n1000.npz, base_sharp2_n1000.npz   (round 5) the headline model with trained-like weights / with every
                             decoder attention logit doubled: one 1000-step segment conditioned on a realistic context,
                             float64 fixture + the float32 oracle's own rms (`robust_<kind>64`, `robust_<kind>32`,
                             `robust_<kind>_pack`; ~10 CPU-minutes per run)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import msd_amd  # noqa: E402
from oracle import backend, fast, philox  # noqa: E402
from tests import helpers  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def tiny():
  spec = msd_amd.config.preset('tiny_context', num_steps=6)
  params = msd_amd.synthetic.init_params(spec, 3, norm_scale_jitter=0.1)
  batch = helpers.make_batch(spec, batch=2, ctx_mask='ragged')
  init_z, noise = helpers.make_noise(spec, batch=2)
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend('float64')
  out = xp.to_numpy(fast.FastModel(xp, cfg, dc, params, True).predict(batch, init_z, noise)[0])
  np.savez_compressed(os.path.join(HERE, 'tiny_context_n6.npz'), mel=out.astype(np.float32),
                      weight_seed=3, jitter=0.1, batch_seed=7, noise_seed=11)


def song(preset, n_segments, name, weight_seed=0, seed=0, dtype='float64', threads=None):
  spec = msd_amd.config.preset(preset, num_steps=1000)
  params = msd_amd.synthetic.init_params(spec, weight_seed)
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend(dtype, threads=threads or os.cpu_count())
  fm = fast.FastModel(xp, cfg, dc, params, spec.has_context)
  t, n = spec.task_feature_lengths['targets'], 128
  c = spec.task_feature_lengths.get('targets_context')
  pred = np.zeros((1, c or 0, n), np.float32)
  outs = []
  path = os.path.join(HERE, name)
  if os.environ.get('CHAIN_RESUME') and os.path.exists(path):   # continue an interrupted chain
    done = np.load(path)
    outs = [done['mel'][:, k * t:(k + 1) * t] for k in range(int(done['n_segments']))]
    pred = outs[-1]
    print('%s: resuming after segment %d' % (name, len(outs) - 1), flush=True)
  for k in range(len(outs), n_segments):
    t0 = time.time()
    batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, k)}
    if spec.has_context:
      batch['encoder_continuous_inputs'] = pred
      batch['encoder_continuous_mask'] = (np.zeros if k == 0 else np.ones)((1, c), np.int32)
    init_z, noise = philox.segment_noise((1, t, n), 1000, seed=seed, segment=k)
    out = xp.to_numpy(fm.predict(batch, init_z, noise)[0]).astype(np.float32)
    pred = out
    outs.append(out)
    print('%s segment %d: %.0fs' % (name, k, time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, name), mel=np.concatenate(outs, 1), weight_seed=weight_seed,
                        noise_seed=seed, n_segments=k + 1)


def trained_like(dtype, threads=None):
  """small / no-context, one 1000-step segment, weights reshaped by synthetic.trained_like (log-normal channel
  gains, outlier channels, log-normal norm scales): float64 fixture + the float32 oracle's run as yardstick."""
  spec = msd_amd.config.preset('small', num_steps=1000)
  params = msd_amd.synthetic.trained_like(msd_amd.synthetic.init_params(spec, 0), seed=1)
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend(dtype, threads=threads or os.cpu_count())
  fm = fast.FastModel(xp, cfg, dc, params, False)
  t, n = spec.task_feature_lengths['targets'], 128
  batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 0)}
  init_z, noise = philox.segment_noise((1, t, n), 1000, seed=0, segment=0)
  t0 = time.time()
  out = xp.to_numpy(fm.predict(batch, init_z, noise)[0]).astype(np.float32)
  print('trained_like %s: %.0fs' % (dtype, time.time() - t0), flush=True)
  np.save(os.path.join(HERE, '_trained_like_%s.npy' % dtype), out)


def trained_pack():
  a = np.load(os.path.join(HERE, '_trained_like_float64.npy'))
  b = np.load(os.path.join(HERE, '_trained_like_float32.npy'))
  print('trained-like: float32 oracle vs float64 rms %.3e' % helpers.rms(b, a))
  np.savez_compressed(os.path.join(HERE, 'small_trained_like_n1000.npz'), mel=a, rms_f32=helpers.rms(b, a),
                      weight_seed=0, reshape_seed=1, noise_seed=0)


def robust_params(spec, kind):
  """Weights of the full-size robustness fixtures: `trained` = synthetic.trained_like (log-normal
  channel gains, outlier channels, log-normal norm scales), `sharp2` = every decoder query kernel x2 (every attention
  logit x2: max |s| ~ 13)."""
  base = msd_amd.synthetic.init_params(spec, 0)
  return msd_amd.synthetic.trained_like(base, seed=1) if kind == 'trained' else msd_amd.synthetic.sharp_attention(base, 2.0)


def robust_batch(spec):
  """One base_with_context segment WITH its context: segment 1 of the bench's song (tokens of segment 1), conditioned on
  segment 0 of the committed float64 fixture base_with_context_n1000.npz (a realistic previous prediction)."""
  c = spec.task_feature_lengths['targets_context']
  prev = np.load(os.path.join(HERE, 'base_with_context_n1000.npz'))['mel'][:, :c].astype(np.float32)
  return {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 1), 'encoder_continuous_inputs': prev,
          'encoder_continuous_mask': np.ones((1, c), np.int32)}


def robust(kind, dtype, threads=None):
  """base_with_context, ONE 1000-step segment, `kind` weights: float64 fixture / float32 yardstick (two processes)."""
  spec = msd_amd.config.preset('base_with_context', num_steps=1000)
  params = robust_params(spec, kind)
  cfg, dc = helpers.oracle_configs(spec)
  xp = backend.TorchBackend(dtype, threads=threads or os.cpu_count())
  fm = fast.FastModel(xp, cfg, dc, params, True)
  t, n = spec.task_feature_lengths['targets'], 128
  init_z, noise = philox.segment_noise((1, t, n), 1000, seed=0, segment=1)
  t0 = time.time()
  out = xp.to_numpy(fm.predict(robust_batch(spec), init_z, noise)[0]).astype(np.float32)
  print('robust %s %s: %.0fs' % (kind, dtype, time.time() - t0), flush=True)
  np.save(os.path.join(HERE, '_robust_%s_%s.npy' % (kind, dtype)), out)


def robust_pack(kind):
  a = np.load(os.path.join(HERE, '_robust_%s_float64.npy' % kind))
  b = np.load(os.path.join(HERE, '_robust_%s_float32.npy' % kind))
  print('base_with_context, %s weights: float32 oracle vs float64 rms %.3e' % (kind, helpers.rms(b, a)))
  np.savez_compressed(os.path.join(HERE, 'base_%s_n1000.npz' % kind), mel=a, rms_f32=helpers.rms(b, a), kind=kind,
                      weight_seed=0, reshape_seed=1, noise_seed=0, segment=1)


def chainpack():
  """Merge the float64 chain and the float32 oracle's own chain into one fixture."""
  a = np.load(os.path.join(HERE, '_chain64.npz'))
  b = np.load(os.path.join(HERE, '_chain32.npz'))
  n = min(int(a['n_segments']), int(b['n_segments']))
  t = 256
  m64, m32 = a['mel'][:, :n * t], b['mel'][:, :n * t]
  rms32 = [helpers.rms(m32[:, k * t:(k + 1) * t], m64[:, k * t:(k + 1) * t]) for k in range(n)]
  print('float32 oracle vs float64 oracle, per segment:', ' '.join('%.2e' % r for r in rms32))
  np.savez_compressed(os.path.join(HERE, 'base_chain_n1000.npz'), mel=m64, mel_f32=m32,
                      rms_f32=np.asarray(rms32), weight_seed=int(a['weight_seed']),
                      noise_seed=int(a['noise_seed']), n_segments=n)


if __name__ == '__main__':
  what = sys.argv[1:] or ['tiny']
  if 'tiny' in what:
    tiny()
  if 'small' in what:
    song('small', 1, 'small_n1000.npz')
  if 'base' in what:
    song('base_with_context', 2, 'base_with_context_n1000.npz')
  nseg = int(os.environ.get('CHAIN_SEGMENTS', 6))
  nthr = int(os.environ.get('CHAIN_THREADS', 0)) or None
  if 'chain64' in what:
    song('base_with_context', nseg, '_chain64.npz', dtype='float64', threads=nthr)
  if 'chain32' in what:
    song('base_with_context', nseg, '_chain32.npz', dtype='float32', threads=nthr)
  if 'chainpack' in what:
    chainpack()
  if 'trained64' in what:
    trained_like('float64', nthr)
  if 'trained32' in what:
    trained_like('float32', nthr)
  if 'trainedpack' in what:
    trained_pack()
  for kind in ('trained', 'sharp2'):   # e.g. robust_trained64 robust_trained32 robust_trained_pack
    for dt, tag in (('float64', '64'), ('float32', '32')):
      if 'robust_%s%s' % (kind, tag) in what:
        robust(kind, dt, nthr)
    if 'robust_%s_pack' % kind in what:
      robust_pack(kind)
