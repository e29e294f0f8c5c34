"""Typed configuration for the DDPM synthesis hot path.

Mirrors the reference dataclasses that decide every shape on the path:
  * ``T5Config``            models/diffusion/network.py:54-72
  * ``DiffusionSchedule`` / ``ClassifierFreeGuidanceConfig`` / ``SamplerConfig``
    / ``DiffusionConfig``   models/diffusion/diffusion_utils.py:25-59
and the presets bound by the shipped gin files
  * base_with_context       gin/models/diffusion/context/t5_base.gin:41-83
                            + gin/tasks/mt3/context_mega.gin:5
  * small (no context)      gin/models/diffusion/basic/t5_small.gin:5-11 over
                            gin/models/diffusion/basic/t5_base.gin:43-83
                            + gin/tasks/mt3/base.gin:5
  * tiny / tiny_context     shape recipe of gin/models/diffusion/*/local_tiny.gin
                            (head_dim widened to 64: the HIP attention kernels are
                            specialised for d=64, the only value any shipped gin uses).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional, Sequence, Tuple


@dataclasses.dataclass(frozen=True)
class T5Config:
  vocab_size: int
  dtype: str = 'float32'
  emb_dim: int = 512
  num_heads: int = 8
  num_encoder_layers: int = 6
  num_decoder_layers: int = 6
  head_dim: int = 64
  mlp_dim: int = 2048
  mlp_activations: Sequence[str] = ('relu',)
  dropout_rate: float = 0.1
  max_decoder_noise_time: float = 2e4
  decoder_cross_attend_style: str = 'sum_cross_attends'
  position_encoding: str = 'fixed'
  context_positions: str = 'regular'


@dataclasses.dataclass(frozen=True)
class DiffusionSchedule:
  name: str
  start: Optional[float] = None
  stop: Optional[float] = None
  num_steps: Optional[int] = None


@dataclasses.dataclass(frozen=True)
class ClassifierFreeGuidanceConfig:
  drop_condition_prob: float = 0.1
  eval_condition_weight: float = 5.0


@dataclasses.dataclass(frozen=True)
class SamplerConfig:
  name: str = 'ddpm'
  schedule: DiffusionSchedule = DiffusionSchedule(name='cosine', num_steps=1000)
  clip_x0: bool = True
  logvar_type: str = 'large'


@dataclasses.dataclass(frozen=True)
class DiffusionConfig:
  time_continuous_or_discrete: str = 'continuous'
  train_schedule: DiffusionSchedule = DiffusionSchedule(name='cosine')
  loss_norm: str = 'l1'
  loss_type: str = 'eps'
  model_output: str = 'eps'
  classifier_free_guidance: ClassifierFreeGuidanceConfig = ClassifierFreeGuidanceConfig()
  sampler: SamplerConfig = SamplerConfig()


@dataclasses.dataclass(frozen=True)
class ModelSpec:
  """Everything ``InferenceModel`` reads out of the gin string (inference.py:97-111)."""
  model: str                      # 'DiffusionModel' | 'ContextDiffusionModel'
  t5: T5Config
  diffusion: DiffusionConfig
  task_feature_lengths: Dict[str, int]
  audio_codec: str = 'MelGAN'
  num_velocity_bins: int = 1

  @property
  def has_context(self) -> bool:
    return self.model == 'ContextDiffusionModel'


def num_embeddings(num_velocity_bins: int = 1) -> int:
  """vocabularies.num_embeddings (vocabularies.py:279-281) for the MT3 codec.

  Codec classes (vocabularies.py:118-139): shift steps 1001 (10 s * 100 steps/s
  + 1), pitch 128, velocity num_velocity_bins+1, tie 1, program 128, drum 128
  -> 1388 for num_velocity_bins=1; +3 special ids +100 extra ids, rounded up
  to a multiple of 128.
  """
  classes = 1001 + 128 + (num_velocity_bins + 1) + 1 + 128 + 128
  vocab = classes + 3 + 100
  return 128 * ((vocab + 127) // 128)


def _ddpm(num_steps: int, cfg_weight: float) -> DiffusionConfig:
  return DiffusionConfig(
      classifier_free_guidance=ClassifierFreeGuidanceConfig(
          eval_condition_weight=cfg_weight),
      sampler=SamplerConfig(
          schedule=DiffusionSchedule(name='cosine', num_steps=num_steps)))


def preset(name: str, num_steps: int = 1000, cfg_weight: float = 5.0) -> ModelSpec:
  """Typed presets equivalent to the shipped gin files (see module docstring)."""
  common = dict(mlp_activations=('gelu', 'linear'),
                decoder_cross_attend_style='concat_encodings',
                position_encoding='fixed_permuted_offset')
  if name == 'base_with_context':
    t5 = T5Config(vocab_size=num_embeddings(1), emb_dim=768, num_heads=12,
                  num_encoder_layers=12, num_decoder_layers=12, head_dim=64,
                  mlp_dim=2048, context_positions='terminal_relative', **common)
    return ModelSpec('ContextDiffusionModel', t5, _ddpm(num_steps, cfg_weight),
                     {'inputs': 2048, 'targets': 256, 'targets_context': 256})
  if name == 'small_with_context':
    t5 = T5Config(vocab_size=num_embeddings(1), emb_dim=512, num_heads=6,
                  num_encoder_layers=8, num_decoder_layers=8, head_dim=64,
                  mlp_dim=1024, context_positions='terminal_relative', **common)
    return ModelSpec('ContextDiffusionModel', t5, _ddpm(num_steps, cfg_weight),
                     {'inputs': 2048, 'targets': 256, 'targets_context': 256})
  if name == 'base':
    t5 = T5Config(vocab_size=num_embeddings(1), emb_dim=768, num_heads=12,
                  num_encoder_layers=12, num_decoder_layers=12, head_dim=64,
                  mlp_dim=2048, **common)
    return ModelSpec('DiffusionModel', t5, _ddpm(num_steps, cfg_weight),
                     {'inputs': 2048, 'targets': 256})
  if name == 'small':
    t5 = T5Config(vocab_size=num_embeddings(1), emb_dim=512, num_heads=6,
                  num_encoder_layers=8, num_decoder_layers=8, head_dim=64,
                  mlp_dim=1024, **common)
    return ModelSpec('DiffusionModel', t5, _ddpm(num_steps, cfg_weight),
                     {'inputs': 2048, 'targets': 256})
  if name == 'tiny_context':
    t5 = T5Config(vocab_size=256, emb_dim=128, num_heads=2, num_encoder_layers=2,
                  num_decoder_layers=2, head_dim=64, mlp_dim=256,
                  context_positions='terminal_relative', **common)
    return ModelSpec('ContextDiffusionModel', t5, _ddpm(num_steps, cfg_weight),
                     {'inputs': 128, 'targets': 64, 'targets_context': 64})
  if name == 'tiny':
    t5 = T5Config(vocab_size=256, emb_dim=128, num_heads=2, num_encoder_layers=2,
                  num_decoder_layers=2, head_dim=64, mlp_dim=256, **common)
    return ModelSpec('DiffusionModel', t5, _ddpm(num_steps, cfg_weight),
                     {'inputs': 128, 'targets': 64})
  raise ValueError('Unknown preset: %s' % name)


def param_shapes(spec: ModelSpec, n_dims: int = 128) -> Dict[str, Tuple[int, ...]]:
  """Flat parameter tree ``name -> shape`` (Flax auto-naming of the reference
  modules; SURVEY.md 8(a) lists the name= citations in network.py / layers.py)."""
  c = spec.t5
  d, j, f = c.emb_dim, c.num_heads * c.head_dim, c.mlp_dim
  lens = spec.task_feature_lengths
  shapes: Dict[str, Tuple[int, ...]] = {}

  def attention(prefix):
    shapes[prefix + '/query/kernel'] = (d, j)
    shapes[prefix + '/key/kernel'] = (d, j)
    shapes[prefix + '/value/kernel'] = (d, j)
    shapes[prefix + '/out/kernel'] = (j, d)

  def mlp(prefix):
    if len(c.mlp_activations) == 1:
      shapes[prefix + '/wi/kernel'] = (d, f)
    else:
      for i in range(len(c.mlp_activations)):
        shapes['%s/wi_%d/kernel' % (prefix, i)] = (d, f)
    shapes[prefix + '/wo/kernel'] = (f, d)

  def encoder_layers(prefix):
    for l in range(c.num_encoder_layers):
      lp = '%s/layers_%d' % (prefix, l)
      shapes[lp + '/pre_attention_layer_norm/scale'] = (d,)
      attention(lp + '/attention')
      shapes[lp + '/pre_mlp_layer_norm/scale'] = (d,)
      mlp(lp + '/mlp')
    shapes[prefix + '/encoder_norm/scale'] = (d,)

  tok = 'token_encoder' if spec.has_context else 'encoder'
  shapes[tok + '/token_embedder/embedding'] = (c.vocab_size, d)
  shapes[tok + '/Embed_0/embedding'] = (lens['inputs'], d)
  encoder_layers(tok)
  if spec.has_context:
    ce = 'continuous_encoder'
    shapes[ce + '/input_proj/kernel'] = (n_dims, d)
    shapes[ce + '/Embed_0/embedding'] = (lens['targets_context'], d)
    encoder_layers(ce)

  dec = 'decoder'
  shapes[dec + '/time_emb_dense0/kernel'] = (d, 4 * d)
  shapes[dec + '/time_emb_dense1/kernel'] = (4 * d, 4 * d)
  shapes[dec + '/Embed_0/embedding'] = (lens['targets'], d)
  shapes[dec + '/continuous_inputs_projection/kernel'] = (n_dims, d)
  n_cross = 1
  if c.decoder_cross_attend_style == 'sum_cross_attends':
    n_cross = 2 if spec.has_context else 1
  for l in range(c.num_decoder_layers):
    lp = '%s/layers_%d' % (dec, l)
    shapes[lp + '/pre_self_attention_layer_norm/scale'] = (d,)
    shapes[lp + '/FiLMLayer_0/DenseGeneral_0/kernel'] = (4 * d, 2 * d)
    attention(lp + '/self_attention')
    shapes[lp + '/pre_cross_attention_layer_norm/scale'] = (d,)
    for n in range(n_cross):
      attention('%s/MultiHeadDotProductAttention_%d' % (lp, n))
    shapes[lp + '/pre_mlp_layer_norm/scale'] = (d,)
    shapes[lp + '/FiLMLayer_1/DenseGeneral_0/kernel'] = (4 * d, 2 * d)
    mlp(lp + '/mlp')
  shapes[dec + '/decoder_norm/scale'] = (d,)
  shapes[dec + '/spec_out_dense/kernel'] = (d, n_dims)
  return shapes


def param_count(spec: ModelSpec) -> int:
  n = 0
  for s in param_shapes(spec).values():
    k = 1
    for v in s:
      k *= v
    n += k
  return n
