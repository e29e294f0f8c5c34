"""gin_lite: the subset reader behind parse_training_gin_file / InferenceModel."""
import os

import pytest

import msd_amd
from msd_amd import config, gin_lite

REF_GIN = '/root/reference/music_spectrogram_diffusion/gin'


def test_preset_round_trips_through_config_string():
  for name in ('base_with_context', 'small', 'tiny', 'tiny_context'):
    spec = config.preset(name, num_steps=37, cfg_weight=2.5)
    text = gin_lite.spec_to_config_str(spec)
    assert gin_lite.model_spec_from_bindings(gin_lite.parse(text)) == spec


def test_block_scoped_and_macro_syntax():
  text = '''
# comment
from __gin__ import dynamic_registration
import seqio
TASK_FEATURE_LENGTHS = {'inputs': 2048, 'targets': 256}
NUM_VELOCITY_BINS = 1
AUDIO_CODEC = @audio_codecs.MelGAN()
MODEL = @models.DiffusionModel()
models.DiffusionModel:
  module = @network.Transformer()
  diffusion_config = @diffusion_utils.DiffusionConfig()
diffusion_utils.DiffusionConfig:
  sampler = @diffusion_utils.SamplerConfig()
diffusion_utils.SamplerConfig:
  schedule = @sampler/diffusion_utils.DiffusionSchedule()
sampler/diffusion_utils.DiffusionSchedule:
  name = 'cosine'
  num_steps = 250
network.T5Config:
  vocab_size = @vocabularies.num_embeddings()
  emb_dim = 512
  num_heads = 6
  mlp_activations = ('gelu', 'linear')
  decoder_cross_attend_style = 'concat_encodings'
network.T5Config.mlp_dim = 1024
'''
  spec = gin_lite.model_spec_from_bindings(gin_lite.parse(text))
  assert spec.model == 'DiffusionModel' and not spec.has_context
  assert spec.t5.emb_dim == 512 and spec.t5.num_heads == 6 and spec.t5.mlp_dim == 1024
  assert spec.t5.vocab_size == 1536
  assert spec.t5.mlp_activations == ('gelu', 'linear')
  assert spec.diffusion.sampler.schedule.num_steps == 250
  assert spec.diffusion.classifier_free_guidance.eval_condition_weight == 5.0  # reference default


def test_missing_bindings_raise_value_error():
  with pytest.raises(ValueError):
    gin_lite.model_spec_from_bindings(gin_lite.parse('MODEL = @models.DiffusionModel()'))
  with pytest.raises(ValueError):
    gin_lite.model_spec_from_bindings(gin_lite.parse(
        "TASK_FEATURE_LENGTHS = {'inputs': 8, 'targets': 8}\nMODEL = @models.ContinuousOutputsEncoderDecoderModel()"))


@pytest.mark.skipif(not os.path.isdir(REF_GIN), reason='reference tree not mounted')
@pytest.mark.parametrize('model_gin,task_gin,preset', [
    ('models/diffusion/context/t5_base.gin', 'tasks/mt3/context_mega.gin', 'base_with_context'),
    ('models/diffusion/basic/t5_small.gin', 'tasks/mt3/base.gin', 'small'),
    ('models/diffusion/context/t5_small.gin', 'tasks/mt3/context_mega.gin', 'small_with_context'),
])
def test_shipped_gin_files_equal_typed_presets(model_gin, task_gin, preset):
  cfg = msd_amd.parse_training_gin_file(
      os.path.join(REF_GIN, model_gin),
      ["include '%s'" % os.path.join(REF_GIN, task_gin),
       "include '%s'" % os.path.join(REF_GIN, 'audio_codecs/melgan.gin')])
  assert gin_lite.model_spec_from_bindings(gin_lite.parse(cfg)) == config.preset(preset)


def test_backslash_wrapped_bindings_as_config_str_emits_them():
  """gin.config_str() wraps every binding longer than 80 columns as `key = \\` + an indented value
  line; the continuation must yield the value (a Ref / dict / tuple), not the string '\\ ...'.  Uses a
  non-default schedule scope so that no hard-coded fallback scope can hide a dropped reference."""
  text = '''
TASK_FEATURE_LENGTHS = \\
    {'inputs': 2048, 'targets': 256}
MODEL = \\
    @models.DiffusionModel()
AUDIO_CODEC = @audio_codecs.MelGAN()
models.DiffusionModel.module = @network.Transformer()
models.DiffusionModel.diffusion_config = \\
    @diffusion_utils.DiffusionConfig()
diffusion_utils.DiffusionConfig.sampler = @diffusion_utils.SamplerConfig()
diffusion_utils.SamplerConfig.schedule = \\
    @fast/diffusion_utils.DiffusionSchedule()
fast/diffusion_utils.DiffusionSchedule.num_steps = 100
network.T5Config.emb_dim = 512
network.T5Config.num_heads = 6
network.T5Config.mlp_dim = 1024
network.T5Config.mlp_activations = \\
    ('gelu', 'linear')
network.T5Config.decoder_cross_attend_style = \\
    'concat_encodings'
'''
  b = gin_lite.parse(text)
  assert not any(isinstance(v, str) and v.startswith('\\') for v in b.values()), b
  spec = gin_lite.model_spec_from_bindings(b)
  assert spec.diffusion.sampler.schedule.num_steps == 100
  assert spec.task_feature_lengths == {'inputs': 2048, 'targets': 256}
  assert spec.t5.mlp_activations == ('gelu', 'linear') and spec.model == 'DiffusionModel'
