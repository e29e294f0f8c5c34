"""Host-side logic of the InferenceModel mirror that needs no GPU: shapes/types
(inference.py:113-157), error behaviour, and the InferSong segment loop
(beam/evaluation.py:161-223) with the device call stubbed out."""
import numpy as np
import pytest
import torch

import msd_amd
from msd_amd import config, inference, native


def _bare_model(preset='tiny_context'):
  """An InferenceModel without a device: everything but the native handle."""
  spec = config.preset(preset, num_steps=4)
  m = object.__new__(inference.InferenceModel)
  m.spec = spec
  m.batch_size = 1
  m.sequence_length = dict(spec.task_feature_lengths)
  m.inputs_length = m.sequence_length['inputs']
  m.targets_length = m.sequence_length['targets']
  m.targets_context_length = m.sequence_length.get('targets_context') if spec.has_context else None
  m.model = inference._ModelInfo(spec)
  m.audio_codec = msd_amd.audio_codecs.MelGAN()
  m._torch = torch
  m.device = torch.device('cpu')
  return m


def test_input_shapes_and_types_follow_reference():
  m = _bare_model('tiny_context')
  assert m.input_shapes == {
      'encoder_input_tokens': (1, 128), 'decoder_target_tokens': (1, 64, 128),
      'encoder_continuous_inputs': (1, 64, 128), 'encoder_continuous_mask': (1, 64)}
  assert m.input_types['encoder_input_tokens'] == np.int32
  assert m.input_types['encoder_continuous_mask'] == np.int32
  m2 = _bare_model('tiny')
  assert set(m2.input_shapes) == {'encoder_input_tokens', 'decoder_target_tokens'}
  assert m2.targets_context_length is None


def test_no_gpu_means_loud_failure_not_fallback():
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  with pytest.raises(native.NativeLibraryError):
    msd_amd.InferenceModel(None, config.preset('tiny'))


def test_unsupported_and_unknown_configs_raise_like_the_reference():
  codec = msd_amd.audio_codecs.MelGAN()
  import dataclasses
  base = config.preset('tiny_context')
  bad = dataclasses.replace(base, t5=dataclasses.replace(base.t5, decoder_cross_attend_style='nope'))
  with pytest.raises(ValueError, match='Unknown decoder_cross_attend_style'):
    inference._to_native_config(bad, codec, 1, 'bf16x3')
  bad = dataclasses.replace(base, t5=dataclasses.replace(base.t5, context_positions='nope'))
  with pytest.raises(ValueError, match='Unknown context_positions'):
    inference._to_native_config(bad, codec, 1, 'bf16x3')
  d = base.diffusion
  bad = dataclasses.replace(base, diffusion=dataclasses.replace(
      d, sampler=dataclasses.replace(d.sampler, name='euler')))
  with pytest.raises(ValueError, match='Unknown sampler type'):
    inference._to_native_config(bad, codec, 1, 'bf16x3')
  bad = dataclasses.replace(base, diffusion=dataclasses.replace(d, model_output='zzz'))
  with pytest.raises(ValueError, match='Unknown model_output'):
    inference._to_native_config(bad, codec, 1, 'bf16x3')
  with pytest.raises(ValueError):
    inference._to_native_config(base, codec, 1, 'fp8')
  summed = dataclasses.replace(base, t5=dataclasses.replace(base.t5, decoder_cross_attend_style='sum_cross_attends'))
  assert inference._to_native_config(summed, codec, 1, 'bf16x3').cross_attend_sum == 1
  cfg = inference._to_native_config(base, codec, 2, 'bf16')
  assert (cfg.emb_dim, cfg.context_length, cfg.max_batch, cfg.precision) == (128, 64, 2, 2)   # MSD_PREC_BF16: its own value since ABI 3


@pytest.mark.parametrize('always_mask', [False, True])
def test_predict_sequence_context_hand_off(always_mask):
  m = _bare_model('tiny_context')
  seen = []

  def fake_predict(batch, seed=0, segment=0, return_torch=False, **kw):
    ctx = batch['encoder_continuous_inputs']
    mask = np.asarray(batch['encoder_continuous_mask'])
    seen.append((segment, float(torch.as_tensor(ctx).sum()), int(mask.sum()), mask.dtype))
    out = torch.full((1, 64, 128), float(segment + 1))
    return out, torch.zeros(1)

  m.predict = fake_predict
  segs = [np.zeros(128, np.int32) for _ in range(3)]
  full = m.predict_sequence(segs, seed=3, always_mask_context=always_mask)
  assert full.shape == (1, 192, 128)
  np.testing.assert_array_equal(full[0, ::64, 0], [1, 2, 3])
  # segment 0: zeros + mask 0; later: previous prediction + mask 1 (unless always masked)
  assert seen[0][1:3] == (0.0, 0)
  assert seen[1][1] == 64 * 128 * 1.0 and seen[2][1] == 64 * 128 * 2.0
  assert [s[2] for s in seen[1:]] == ([0, 0] if always_mask else [64, 64])
  assert all(s[3] == np.int32 for s in seen)
  # resuming mid-song (chained multi-GPU hand-off): first local segment has context
  seen.clear()
  m.predict_sequence(segs[:1], init_context=np.ones((1, 64, 128), np.float32), first_segment_index=5)
  assert seen[0][0] == 5 and seen[0][2] == 64


def test_checkpoint_loader_paths(tmp_path):
  spec = config.preset('tiny')
  params, step = inference._load_checkpoint('synthetic:4', spec)
  assert step == 0 and set(params) == set(config.param_shapes(spec))
  np.savez(tmp_path / 'w.npz', __step__=np.int64(1234), **params)
  p2, step2 = inference._load_checkpoint(str(tmp_path / 'w.npz'), spec)
  assert step2 == 1234 and all(np.array_equal(p2[k], params[k]) for k in params)
  with pytest.raises(ValueError):       # not a directory (T5X dirs are read: tests/test_checkpoints.py)
    inference._load_checkpoint('/some/t5x/checkpoint_500000', spec)


def test_chunking_properties():
  """sharding.contiguous_chunk / deal_round_robin over every (segments, world) up to 40 x 12: the chunks tile
  [0, n) in rank order without gaps, sizes differ by at most one with the larger ones first (the sender /
  receiver conditions of chained_predict rely on it), and round-robin dealing is a partition."""
  from msd_amd import sharding
  for n in range(0, 41):
    for world in range(1, 13):
      chunks = [sharding.contiguous_chunk(n, r, world) for r in range(world)]
      assert chunks[0][0] == 0 and chunks[-1][1] == n
      assert all(chunks[r][1] == chunks[r + 1][0] for r in range(world - 1))
      sizes = [b - a for a, b in chunks]
      assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
      # a rank receives the hand-off iff the previous rank sends it (chained_predict's two conditions)
      for r in range(1, world):
        recv = 0 < chunks[r][0] < n
        send = chunks[r - 1][1] < n
        assert recv == (send and chunks[r - 1][1] > 0)
      dealt = sorted(i for r in range(world) for i in sharding.deal_round_robin(n, r, world))
      assert dealt == list(range(n))
