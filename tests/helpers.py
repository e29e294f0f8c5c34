"""Shared test helpers: product spec -> oracle configs, seeded inputs."""
import numpy as np

import msd_amd
from oracle import backend, fast, net, predict, sampler

T5_FIELDS = ['vocab_size', 'emb_dim', 'num_heads', 'num_encoder_layers', 'num_decoder_layers',
             'head_dim', 'mlp_dim', 'mlp_activations', 'max_decoder_noise_time',
             'decoder_cross_attend_style', 'position_encoding', 'context_positions']


def oracle_configs(spec):
  t5 = net.T5Config(**{k: getattr(spec.t5, k) for k in T5_FIELDS})
  d = spec.diffusion

  def sched(x):
    return sampler.DiffusionSchedule(x.name, start=x.start, stop=x.stop, num_steps=x.num_steps)

  dc = sampler.DiffusionConfig(
      model_output=d.model_output, train_schedule=sched(d.train_schedule),
      classifier_free_guidance=sampler.ClassifierFreeGuidanceConfig(
          eval_condition_weight=d.classifier_free_guidance.eval_condition_weight),
      sampler=sampler.SamplerConfig(
          name=d.sampler.name, clip_x0=d.sampler.clip_x0, logvar_type=d.sampler.logvar_type,
          schedule=sched(d.sampler.schedule)))
  return t5, dc


def make_batch(spec, batch=1, seed=7, ctx_mask='ones', segment0=0):
  """Model features with seeded content.  ctx_mask: 'ones' | 'zeros' | 'ragged'."""
  rng = np.random.default_rng(seed)
  toks = np.concatenate([msd_amd.synthetic.segment_tokens(spec, segment0 + b, min_len=8,
                                                           max_len=spec.task_feature_lengths['inputs'] - 2)
                         for b in range(batch)], 0)
  out = {'encoder_input_tokens': toks}
  n = 128
  if spec.has_context:
    c = spec.task_feature_lengths['targets_context']
    out['encoder_continuous_inputs'] = rng.uniform(-13, 5, (batch, c, n)).astype(np.float32)
    if ctx_mask == 'ones':
      m = np.ones((batch, c), np.int32)
    elif ctx_mask == 'zeros':
      m = np.zeros((batch, c), np.int32)
    else:  # a valid prefix of different length per row (exercises terminal_relative)
      m = np.zeros((batch, c), np.int32)
      for b in range(batch):
        m[b, :int(rng.integers(1, c))] = 1
    out['encoder_continuous_mask'] = m
  out['decoder_target_tokens'] = np.zeros((batch, spec.task_feature_lengths['targets'], n), np.float32)
  return out


def make_noise(spec, batch=1, seed=11):
  rng = np.random.default_rng(seed)
  t, n = spec.task_feature_lengths['targets'], 128
  steps = spec.diffusion.sampler.schedule.num_steps
  return (rng.standard_normal((batch, t, n)).astype(np.float32),
          rng.standard_normal((steps, batch, t, n)).astype(np.float32))


def rms(a, b):
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  return float(np.sqrt(np.mean((a - b) ** 2)))


# Short DDPM chains (3 - 8 huge steps) are ill-conditioned amplifiers: assert_fp32_class below can tell float32-class
# arithmetic from anything coarser.  Since round 4 the product's default 'f16x3' mode keeps hi + lo planes EVERYWHERE,
# the query side of the decoder's attentions included (round 3's default ran Q and the softmax weights on one half
# plane: 2.5 % faster, fine on O(1) logits, 2.8x the float32 floor on sharp attention -- DESIGN.md 3).  One plane is an
# opt-in (ONE_QUERY_PLANE): its single passes are asserted against the same 2e-4 / 3e-4 bounds as the default's, its
# op-level behaviour on sharp logits in tests/test_gpu_ops.py.
ALL_PLANES = dict(attention_query_planes=2)        # = the library default since round 4 (kept explicit where a test's
                                                    # statistics REQUIRE it, whatever the default becomes)
ONE_QUERY_PLANE = dict(attention_query_planes=1)    # the opt-in: Q and the softmax weights as one half plane each


def assert_fp32_class(got, ref64, ref32, what=''):
  """Short DDPM chains are ill-conditioned by construction: at the first step
  (t = 1, logsnr = -20) x0 = 22026 (z - eps) is clipped to +-1 for all but a few
  marginal elements, where a 1e-6 difference in eps decides the outcome.  Any two valid
  float32-class evaluations (the float32 oracle included) therefore disagree on a
  handful of elements by O(1) while agreeing to ~1e-6 on the rest, and rms is
  dominated by those few.  So the device is compared with the float64 oracle on the
  BULK (median error) and on the COUNT of outliers, using the float32 oracle's own
  deviation from float64 as the yardstick.  (The 1000-step runs use the absolute
  1e-3 rms bar: tests/test_golden.py.)"""
  e_dev = np.abs(np.asarray(got, np.float64) - ref64).ravel()
  e_f32 = np.abs(np.asarray(ref32, np.float64) - ref64).ravel()
  med_dev, med_f32 = np.median(e_dev), np.median(e_f32)
  out_dev, out_f32 = float((e_dev > 1e-2).mean()), float((e_f32 > 1e-2).mean())
  print('%s median |err| device %.2e / f32-oracle %.2e; outliers(>1e-2) device %.4f / f32-oracle %.4f; rms %.2e / %.2e'
        % (what, med_dev, med_f32, out_dev, out_f32, rms(got, ref64), rms(ref32, ref64)))
  assert med_dev <= 1.5 * med_f32 + 1e-6, 'bulk error is not fp32-class'   # (worst over round 4's GPU suite: 1.08)
  # Outliers: at most twice the float32 oracle's own count plus 0.3 % of the elements.  bf16x3 products carry
  # ~2^-17 relative error against float32's 2^-24, so more marginal elements flip at the clip boundary of the
  # first steps, and WHICH ones flip depends on the last bits: two builds of the same kernels whose single
  # decoder passes agree to 4e-6 relative (round 2: a code-generation difference in the attention kernel,
  # DESIGN 11) gave 0.0001 and 0.0020 on the tiny golden, against the float32 oracle's 0.0002.
  assert out_dev <= 2 * out_f32 + 3e-3, 'too many outliers'
  # A third criterion, on the error DISTRIBUTION below the outlier threshold (the two above
  # are nearly blind to a kernel that is moderately wrong everywhere).  A short chain turns one rounding into a
  # spread of errors between 1e-6 and 1e-2 whatever the arithmetic, on different elements for every evaluation -- so an
  # rms over "the elements the float32 oracle gets right" cannot work (measured on the MI355X: the device's rms over
  # those is 3 - 8x the float32 oracle's, simply because its amplified elements are other ones).  What two float32-
  # class evaluations do share is HOW MANY elements end up beyond a given error: on tiny_context the float32 oracle,
  # the half-plane emulation and the bfloat16-plane emulation have 5.4 / 5.5 / 6.5 % of their elements beyond 1e-3 and
  # 27.1 / 27.5 / 30.5 % beyond 1e-4, a single half plane ('f16') 33 % and 35 %.
  # MI355X, 46 short-chain cases (profiles/r03l_tests.log): the device is within +-0.02 of the float32 oracle's fraction at
  # both thresholds (worst ratio 1.07)
  # Round 4 (the bounds of round 3, x1.25 / x1.15 + 0.01, would have passed the bfloat16-plane
  # emulation's 6.5 % against the float32 oracle's 5.4 %): x1.10 / x1.12 + 0.005.  The bfloat16-plane figure is now
  # outside (bound 6.44 %); the worst device case of round 4's suite (profiles/r04z_gpu_tests.log, 32 short-chain cases
  # in both attention modes) is x1.017 / x1.070, +0.003 / +0.0185 absolute.
  for tau, factor, slack in ((1e-3, 1.10, 0.005), (1e-4, 1.12, 0.005)):
    f_dev, f_f32 = float((e_dev > tau).mean()), float((e_f32 > tau).mean())
    print('%s elements beyond %.0e: device %.4f / f32-oracle %.4f' % (what, tau, f_dev, f_f32))
    assert f_dev <= factor * f_f32 + slack, 'too many elements beyond %.0e for float32-class arithmetic' % tau
