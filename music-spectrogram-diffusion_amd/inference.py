"""Drop-in mirror of the reference ``inference.py`` for the diffusion path.

Same public surface as the reference (file:line = music_spectrogram_diffusion/):
  * ``parse_training_gin_file(gin_file, gin_bindings) -> str``      inference.py:32-65
  * ``InferenceModel(checkpoint_path, gin_config, batch_size=1)``    inference.py:71-111
      .sequence_length / .inputs_length / .targets_length /
      .targets_context_length                                        inference.py:97-101
      .model.FEATURE_CONVERTER_CLS, .audio_codec, .codec, .step      inference.py:104-111,178-181
      .input_shapes / .input_types                                    inference.py:113-157
      .predict(batch, seed=0) -> (decodes, scores)                    inference.py:200-203
plus ``predict_sequence`` -- the per-song segment loop of
``InferSong.process`` (beam/evaluation.py:161-223) that BASELINE.json's
north_star names.

The compute is the HIP library behind include/msd_amd.h (native.py); there is
no fallback.  Differences that a caller can observe, all documented in
DESIGN.md: (1) the RNG is the library's Philox generator, not jax threefry
(pass ``init_z``/``noise`` for bit-identical noise across implementations);
(2) ``checkpoint_path`` accepts a T5X checkpoint directory like the reference
(checkpoints.py restates the flax-msgpack + zarr layout), and additionally
``None`` / ``'synthetic[:seed]'`` (scratch init with the reference initialisers,
as ``from_checkpoint_or_scratch`` does without a checkpoint), a ``.safetensors``
/ ``.npz`` flat dict, or an in-memory dict.  MIDI / NoteSequence input goes
through ``frontend/`` (``synthesize_midi``); ``.codec`` is the event codec the
tokeniser uses (inference.py:108-111).
"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from . import audio_codecs
from . import config as config_lib
from . import gin_lite
from . import native
from . import synthetic


def parse_training_gin_file(gin_file: str, gin_bindings: Sequence[str] = ()) -> str:
  """Parse a T5X training ``.gin`` (with includes) + extra bindings into the flat
  operative config string ``InferenceModel`` consumes (inference.py:32-65)."""
  with open(gin_file) as f:
    text = f.read()
  roots = [os.path.dirname(os.path.abspath(gin_file)), os.getcwd()]
  bindings = gin_lite.parse(text, roots)
  for b in gin_bindings:
    bindings.update(gin_lite.parse(b, roots))
  return gin_lite.to_config_str(bindings)


class _FeatureConverter:
  """Stand-in for the seqio feature-converter classes: only the names the model
  consumes matter at this boundary (models/diffusion/feature_converters.py:32-43,
  feature_converters.py:35-36)."""

  def __init__(self, model_features):
    self.MODEL_FEATURES = dict.fromkeys(model_features)


class _ModelInfo:
  """What callers read from ``InferenceModel.model`` (beam/evaluation.py:148)."""

  def __init__(self, spec: config_lib.ModelSpec):
    self.name = spec.model
    self.diffusion_config = spec.diffusion
    if spec.has_context:
      feats = ['encoder_input_tokens', 'encoder_continuous_inputs', 'encoder_continuous_mask',
               'decoder_target_tokens', 'decoder_target_mask']
    else:
      feats = ['encoder_input_tokens', 'decoder_target_tokens', 'decoder_target_mask']
    self.FEATURE_CONVERTER_CLS = _FeatureConverter(feats)


def _to_native_config(spec: config_lib.ModelSpec, codec: audio_codecs.AudioCodec,
                      batch_size: int, precision: str, attention_query_planes=None, graph_steps: int = 0,
                      weight_prefetch: Optional[bool] = None, dedup_layer0: Optional[bool] = None,
                      cross_key_split: int = 0, keep_raw_weights: bool = False,
                      kv_touch_ahead: Optional[int] = None, cross_merge_in_launch: Optional[bool] = None,
                      cross_q_fold: Optional[bool] = None, mlp_in_persistent: Optional[bool] = None) -> native.MsdConfig:
  t5, d = spec.t5, spec.diffusion
  # Everything the kernels fix by construction is validated here with the
  # reference's own error type (ValueError; msd_amd.h msd_config comment).
  if t5.decoder_cross_attend_style not in ('concat_encodings', 'sum_cross_attends'):   # network.py:236-238
    raise ValueError(f'Unknown decoder_cross_attend_style: {t5.decoder_cross_attend_style}')
  if tuple(t5.mlp_activations) != ('gelu', 'linear'):
    raise NotImplementedError('only gated-GELU MLPs (mlp_activations=(gelu, linear)) are built')
  if t5.position_encoding not in ('fixed', 'fixed_permuted_offset', 'learnable_permuted_offset',
                                  'random'):
    raise ValueError(f'Unknown position_encoding: {t5.position_encoding}')
  if t5.context_positions not in ('regular', 'terminal_relative'):
    raise ValueError(f'Unknown context_positions: {t5.context_positions}')
  if d.sampler.name not in ('ddpm', 'ddim'):
    raise ValueError('Unknown sampler type: %s' % d.sampler.name)
  for sched in (d.sampler.schedule, d.train_schedule):   # diffusion_utils.py:181-202
    if sched.name not in native.SCHEDULES:
      raise ValueError('Schedule %s not identified.' % sched.name)
    if sched.name == 'linear' and (sched.start is None or sched.stop is None or not sched.num_steps):
      raise ValueError('linear schedule needs start, stop and num_steps')
  if d.model_output not in native.MODEL_OUTPUTS:       # diffusion_utils.py:288-322
    if d.model_output == 'x0_and_eps':
      raise NotImplementedError(
          'model_output="x0_and_eps" splits a 2n-channel network output (diffusion_utils.py:306-311), '
          "which the reference's own Decoder cannot produce (n_out = input channels, network.py:451-456)")
    raise ValueError('Unknown model_output: %s' % d.model_output)
  lv = d.sampler.logvar_type                            # diffusion_utils.py:141-157
  lv_frac = 0.0
  if lv == 'large':
    lv_kind = native.MSD_LOGVAR_LARGE
  elif lv == 'small':
    lv_kind = native.MSD_LOGVAR_SMALL
  elif lv.startswith('medium:'):
    lv_kind, lv_frac = native.MSD_LOGVAR_MEDIUM, float(lv.split(':')[1])
    if not 0 <= lv_frac <= 1:
      raise ValueError('logvar_type medium:<frac> needs 0 <= frac <= 1')
  else:
    raise ValueError('Unknown logvar_type: %s' % lv)
  if precision not in native.PRECISIONS:
    raise ValueError('precision must be one of %s' % sorted(native.PRECISIONS))
  lens = spec.task_feature_lengths
  cfg = native.MsdConfig()
  cfg.has_context = int(spec.has_context)
  cfg.vocab_size = t5.vocab_size
  cfg.emb_dim = t5.emb_dim
  cfg.num_heads = t5.num_heads
  cfg.head_dim = t5.head_dim
  cfg.mlp_dim = t5.mlp_dim
  cfg.num_encoder_layers = t5.num_encoder_layers
  cfg.num_decoder_layers = t5.num_decoder_layers
  cfg.inputs_length = lens['inputs']
  cfg.targets_length = lens['targets']
  cfg.context_length = lens.get('targets_context', 0) if spec.has_context else 0
  cfg.n_dims = codec.n_dims
  cfg.num_steps = d.sampler.schedule.num_steps
  cfg.sampler = native.MSD_SAMPLER_DDIM if d.sampler.name == 'ddim' else native.MSD_SAMPLER_DDPM
  cfg.clip_x0 = int(d.sampler.clip_x0)
  cfg.context_terminal_relative = int(t5.context_positions == 'terminal_relative')
  cfg.precision = native.PRECISIONS[precision]
  cfg.max_batch = batch_size
  cfg.max_decoder_noise_time = t5.max_decoder_noise_time
  cfg.cfg_weight = d.classifier_free_guidance.eval_condition_weight
  cfg.feature_min = codec.min_value
  cfg.feature_max = codec.max_value
  cfg.model_output = native.MODEL_OUTPUTS[d.model_output]
  cfg.logvar_type, cfg.logvar_frac = lv_kind, lv_frac
  ss, ts = d.sampler.schedule, d.train_schedule
  cfg.sampler_schedule = native.SCHEDULES[ss.name]
  cfg.sampler_schedule_start, cfg.sampler_schedule_stop = float(ss.start or 0.0), float(ss.stop or 0.0)
  cfg.train_schedule = native.SCHEDULES[ts.name]
  cfg.train_schedule_start, cfg.train_schedule_stop = float(ts.start or 0.0), float(ts.stop or 0.0)
  cfg.train_schedule_num_steps = int(ts.num_steps or 0)
  cfg.cross_attend_sum = int(t5.decoder_cross_attend_style == 'sum_cross_attends')
  # query side of the decoder's attentions: one number for both Q (in q.k^T) and the softmax weights (in P.V), or a
  # (q_planes, p_planes) pair; None / 0 = the library's choice
  qp = attention_query_planes
  if not isinstance(qp, (tuple, list)):
    qp = (qp, qp)
  if len(qp) != 2 or any(v not in (None, 0, 1, 2) for v in qp):
    raise ValueError('attention_query_planes must be None, 1, 2 or a (q_planes, p_planes) pair of those')
  cfg.attn_q_planes, cfg.attn_p_planes = int(qp[0] or 0), int(qp[1] or 0)
  if not 0 <= int(graph_steps) <= 64:
    raise ValueError('graph_steps must be in [0, 64] (0 = library default)')
  cfg.graph_steps = int(graph_steps)
  cfg.weight_prefetch = 0 if weight_prefetch is None else (1 if weight_prefetch else 2)
  cfg.dedup_layer0 = 0 if dedup_layer0 is None else (1 if dedup_layer0 else 2)
  if int(cross_key_split) not in (0, 1, 2, 4, 8):
    raise ValueError('cross_key_split must be 0 (chosen per segment), 1, 2, 4 or 8')
  cfg.cross_key_split = int(cross_key_split)
  cfg.keep_raw_weights = int(bool(keep_raw_weights))
  if kv_touch_ahead is not None and not 0 <= int(kv_touch_ahead) <= 16:
    raise ValueError('kv_touch_ahead must be None (library default), 0 (off) or 1 .. 16 stages')
  cfg.kv_touch_ahead = 0 if kv_touch_ahead is None else (-1 if int(kv_touch_ahead) == 0 else int(kv_touch_ahead))
  cfg.cross_merge_in_launch = 0 if cross_merge_in_launch is None else (1 if cross_merge_in_launch else 2)
  cfg.cross_q_fold = 0 if cross_q_fold is None else (1 if cross_q_fold else 2)
  cfg.mlp_in_persistent = 0 if mlp_in_persistent is None else (1 if mlp_in_persistent else 2)
  return cfg


def _load_checkpoint(path, spec: config_lib.ModelSpec) -> Tuple[Dict[str, np.ndarray], int]:
  if isinstance(path, Mapping):
    return dict(path), 0
  if path is None or str(path).startswith('synthetic'):
    seed = 0
    if path is not None and ':' in str(path):
      seed = int(str(path).split(':', 1)[1])
    return synthetic.init_params(spec, seed), 0
  path = str(path)
  if path.endswith('.safetensors'):
    from safetensors.numpy import load_file
    flat = load_file(path)
  elif path.endswith('.npz'):
    with np.load(path) as f:
      flat = {k: f[k] for k in f.files}
  elif os.path.isdir(path):
    # a T5X checkpoint directory, what the reference restores (inference.py:159-176)
    from . import checkpoints
    flat = checkpoints.load_t5x_checkpoint(path)
  else:
    raise ValueError(
        'checkpoint_path must be a T5X checkpoint directory, a .safetensors/.npz flat dict keyed by '
        'the Flax parameter names, or "synthetic[:seed]": %r' % path)
  step = int(np.asarray(flat.pop('__step__', 0)))
  return {k: np.asarray(v, np.float32) for k, v in flat.items()}, step


class InferenceModel(object):
  """Wrapper of the HIP synthesizer with the reference's InferenceModel API."""

  def __init__(self, checkpoint_path, gin_config: Union[str, config_lib.ModelSpec],
               batch_size: int = 1, precision: str = 'f16x3', device: Optional[int] = None,
               range_fallback: bool = True, attention_query_planes=None, graph_steps: int = 0,
               weight_prefetch: Optional[bool] = None, dedup_layer0: Optional[bool] = None,
               cross_key_split: int = 0, keep_raw_weights: bool = False, kv_touch_ahead: Optional[int] = None,
               cross_merge_in_launch: Optional[bool] = None, cross_q_fold: Optional[bool] = None,
               mlp_in_persistent: Optional[bool] = None):
    """Args mirror inference.py:71-88.

    gin_config: the parsed gin string (``parse_training_gin_file``) or a typed
      ``config.ModelSpec`` preset.
    precision: 'f16x3' (default; operands as hi + lo IEEE-half planes: float32-class
      results, 5x inside the 1e-3 rms parity bar; weights must satisfy |w| < 128),
      'bf16x3' (hi + lo bfloat16 planes: float32's exponent range, 2x the error; the
      other library build), 'f16' / 'bf16' (one plane, fastest; do NOT meet the bar).
    device: HIP device index (default: torch's current device).
    attention_query_planes: planes of the query side of the decoder's attentions in the hi + lo modes (msd_config
      attn_q_planes / attn_p_planes): None = the library's choice; 1 or 2 for both Q (in q.k^T) and
      the softmax weights (in P.V), or a (q_planes, p_planes) pair.  The memory side (K, V) always keeps hi + lo.
    graph_steps: DDPM steps captured per hipGraph (0 = the library's choice, 8).
    weight_prefetch: None = the library decides from the model's size; True / False force it.
    dedup_layer0: a CFG step computes decoder layer 0's self-attention block once for both passes (exact: they are
      bit-identical up to the first cross-attention, models/diffusion/models.py:373-386); None = the library's choice
      (on), False turns it off (A/B and bitwise tests).
    cross_key_split: blocks sharing the key axis of one (head, query tile) of the decoder's cross-attention; 0 = the
      library chooses per segment from its key count, else 1, 2, 4 or 8.
    kv_touch_ahead: 128-key ring stages by which the cross-attention launches' prefetch wave touches the cached K / V
      lines ahead of their LDS-DMA (None = the library's choice, 0 = off).
    cross_merge_in_launch: a key-split cross-attention finishes inside its launch (the last block of a group merges
      the partials; no merge launch); None = the library's choice, False = the separate merge launch.  Bit-identical.
    cross_q_fold: the cross-attention's query projection has no launch of its own (folded into the QKV and the
      self-attention output projection launches by exact algebra, msd_amd.h); None = the library's choice (on)
    mlp_in_persistent: batched songs (>= 4 per call): the decoder's gated-MLP input projection as one resident block per
      CU walking its tiles (register epilogue, the next tile's operands land under it); None = the library's choice (on)
    keep_raw_weights: keep the float32 staging copies of the packed matrices on the device (default: freed after
      packing -- 1.5 GB per handle at base_with_context).
    range_fallback: what to do when an activation leaves the range of the half planes (|x| > 65504; the
      library detects it and fails the call with native.RangeError -- the reference is float32 and has no such
      limit): True (default) switches this model to 'bf16x3' (bfloat16 planes: float32's exponent range, twice
      the rounding error), once and for good, with a RuntimeWarning, and repeats the call -- a valid workload
      never fails; False lets the error out.
    """
    import torch  # device memory + streams only
    if isinstance(gin_config, config_lib.ModelSpec):
      spec = gin_config
      self.gin_config = gin_lite.spec_to_config_str(spec)
    else:
      self.gin_config = gin_config
      spec = gin_lite.model_spec_from_bindings(gin_lite.parse(gin_config))
    self.spec = spec
    self.checkpoint_path = checkpoint_path
    self.batch_size = batch_size
    self.precision = precision
    self.range_fallback = bool(range_fallback)
    self.attention_query_planes = attention_query_planes
    self.graph_steps, self.weight_prefetch = graph_steps, weight_prefetch
    self.dedup_layer0, self.cross_key_split, self.keep_raw_weights = dedup_layer0, cross_key_split, keep_raw_weights
    self.kv_touch_ahead = kv_touch_ahead
    self.cross_merge_in_launch, self.cross_q_fold = cross_merge_in_launch, cross_q_fold
    self.mlp_in_persistent = mlp_in_persistent

    self.sequence_length = dict(spec.task_feature_lengths)
    self.inputs_length = self.sequence_length['inputs']
    self.targets_length = self.sequence_length['targets']
    self.targets_context_length = self.sequence_length.get('targets_context', None)
    if spec.has_context and self.targets_context_length is None:
      raise ValueError('ContextDiffusionModel needs TASK_FEATURE_LENGTHS["targets_context"]')
    if not spec.has_context:
      self.targets_context_length = None

    self.model = _ModelInfo(spec)
    self.audio_codec = audio_codecs.get_codec(spec.audio_codec)
    # inference.py:104-111: the event codec of the tokeniser side, built from the gin's velocity bins
    from .frontend import vocabularies
    self.vocab_config = vocabularies.VocabularyConfig(num_velocity_bins=spec.num_velocity_bins)
    self.codec = vocabularies.build_codec(self.vocab_config)

    if not torch.cuda.is_available():
      raise native.NativeLibraryError(
          'no HIP device visible: InferenceModel runs only on the GPU (no CPU fallback)')
    self._torch = torch
    self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
    self._params_np: Optional[Dict[str, np.ndarray]] = None
    self._native: Optional[native.NativeModel] = None
    self._step = 0
    self._stream = None
    self.last_timing: Dict[str, float] = {}

  # -- shapes / types (inference.py:113-157) -------------------------------------
  @property
  def input_shapes(self):
    shapes = {
        'encoder_input_tokens': (self.batch_size, self.inputs_length),
        'decoder_target_tokens': (self.batch_size, self.targets_length, self.audio_codec.n_dims),
    }
    if self.targets_context_length is not None:
      shapes.update({
          'encoder_continuous_inputs':
              (self.batch_size, self.targets_context_length, self.audio_codec.n_dims),
          'encoder_continuous_mask': (self.batch_size, self.targets_context_length),
      })
    if 'decoder_input_tokens' in self.model.FEATURE_CONVERTER_CLS.MODEL_FEATURES:
      shapes['decoder_input_tokens'] = shapes['decoder_target_tokens']
    return shapes

  @property
  def input_types(self):
    types = {'encoder_input_tokens': np.int32, 'decoder_target_tokens': np.float32}
    if self.targets_context_length is not None:
      types.update({'encoder_continuous_inputs': np.float32,
                    'encoder_continuous_mask': np.int32})
    if 'decoder_input_tokens' in self.model.FEATURE_CONVERTER_CLS.MODEL_FEATURES:
      types['decoder_input_tokens'] = types['decoder_target_tokens']
    return types

  # -- restore (inference.py:159-198): lazy, once per process ----------------------
  def _get_native(self) -> native.NativeModel:
    if self._native is None:
      torch = self._torch
      with torch.cuda.device(self.device):
        if self._params_np is not None:   # rebuilt after a range fallback: same weights
          params = self._params_np
        else:
          params, self._step = _load_checkpoint(self.checkpoint_path, self.spec)
        cfg = _to_native_config(self.spec, self.audio_codec, self.batch_size, self.precision,
                                self.attention_query_planes, self.graph_steps, self.weight_prefetch,
                                self.dedup_layer0, self.cross_key_split, self.keep_raw_weights, self.kv_touch_ahead,
                                self.cross_merge_in_launch, self.cross_q_fold, self.mlp_in_persistent)
        nm = native.NativeModel(cfg)   # the library build (plane format) follows from cfg.precision
        self._stream = torch.cuda.Stream(device=self.device)
        nm.load_weights(params, stream=self._stream.cuda_stream)
        self._params_np = params
        self._native = nm
    return self._native

  @property
  def step(self):
    self._get_native()
    return self._step

  @property
  def params(self) -> Dict[str, np.ndarray]:
    self._get_native()
    return self._params_np

  # -- predict (inference.py:200-203) -----------------------------------------------
  def predict(self, batch: Mapping[str, Any], seed: int = 0, segment: int = 0,
              init_z=None, noise=None, return_torch: bool = False, rng: str = 'philox'):
    """Predict one batch of 256-frame segments.

    batch: the model features of inference.py:113-136 (NumPy arrays or torch
      tensors); ``decoder_target_tokens`` is used for its shape only.
    seed / segment: key of the Philox generator (replaces PRNGKey(seed)).
    init_z [B,T,n] / noise [N,B,T,n]: explicit draws (the parity contract).
    rng: 'philox' (default: the library's device generator, one stream per segment) or 'jax':
      the draws jax.random would make for PRNGKey(seed) -- init_z = normal(key), step-i noise =
      normal(fold_in(key, i)) -- restated on the host (jax_random.py) and cached per
      (seed, batch).  Like the reference (beam/evaluation.py:209 calls predict(batch) with the default
      seed for EVERY segment), `segment` does not enter the key in this mode.
    Returns (decodes float32 [B,T,n] in mel units, scores float32 [B] zeros).
    """
    try:
      return self._predict_once(batch, seed, segment, init_z, noise, return_torch, rng)
    except native.RangeError:
      if not (self.range_fallback and self.precision in ('f16x3', 'f16')):
        raise
      import warnings
      new = 'bf16x3' if self.precision == 'f16x3' else 'bf16'
      warnings.warn("an activation left the half-plane range (|x| > 65504): switching this model from precision "
                    "'%s' to '%s' (bfloat16 planes) and repeating the call" % (self.precision, new), RuntimeWarning)
      self.precision = new
      if self._native is not None:
        self._native.close()
      self._native = None
      return self._predict_once(batch, seed, segment, init_z, noise, return_torch, rng)

  def _predict_once(self, batch, seed, segment, init_z, noise, return_torch, rng):
    torch = self._torch
    nm = self._get_native()
    dev = self.device
    tokens = _to_numpy(batch['encoder_input_tokens'])
    if tokens.dtype.kind not in 'iu':   # layers.Embed (layers.py:546-547; layers_test.py:392-401)
      raise ValueError('Input type must be an integer or unsigned integer.')
    tokens = np.ascontiguousarray(tokens, dtype=np.int32)
    b = tokens.shape[0]
    if tokens.ndim != 2 or tokens.shape[1] != self.inputs_length:
      raise ValueError('encoder_input_tokens must be [batch, %d]' % self.inputs_length)
    if b > self.batch_size:
      raise ValueError('batch %d exceeds batch_size %d' % (b, self.batch_size))
    t, n = self.targets_length, self.audio_codec.n_dims
    t0 = time.perf_counter()
    with torch.cuda.device(dev):
      # tensors the caller produced on its own stream (e.g. the previous prediction) are consumed on
      # the model's stream: order the two
      self._stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.device(dev), torch.cuda.stream(self._stream):
      s = self._stream.cuda_stream
      ctx = mask = None
      if self.targets_context_length is not None:
        ctx = _to_device(torch, batch['encoder_continuous_inputs'], dev, torch.float32)
        if tuple(ctx.shape) != (b, self.targets_context_length, n):
          raise ValueError('encoder_continuous_inputs must be [batch, %d, %d]'
                           % (self.targets_context_length, n))
        mask = np.ascontiguousarray(_to_numpy(batch['encoder_continuous_mask']), dtype=np.int32)
        if mask.shape != (b, self.targets_context_length):   # msd_encode copies batch * C ints from it
          raise ValueError('encoder_continuous_mask must be [batch, %d]' % self.targets_context_length)
      nm.encode(b, tokens, ctx, mask, stream=s)
      t1 = time.perf_counter()
      out = torch.empty((b, t, n), dtype=torch.float32, device=dev)
      if rng == 'jax' and init_z is None and noise is None:
        init_z, noise = self._jax_noise(seed, b)
      elif rng not in ('philox', 'jax'):
        raise ValueError("rng must be 'philox' or 'jax': %r" % (rng,))
      z0 = None if init_z is None else _to_device(torch, init_z, dev, torch.float32)
      nz = None if noise is None else _to_device(torch, noise, dev, torch.float32)
      if z0 is not None and tuple(z0.shape) != (b, t, n):
        raise ValueError('init_z must be [batch, %d, %d]' % (t, n))
      if nz is not None and tuple(nz.shape) != (self.spec.diffusion.sampler.schedule.num_steps, b, t, n):
        raise ValueError('noise must be [num_steps, batch, %d, %d]' % (t, n))
      nm.sample(b, out, seed=seed, stream_id=segment, init_z=z0, noise=nz, stream=s)
      self._stream.synchronize()
    t2 = time.perf_counter()
    self.last_timing = {'encode_s': t1 - t0, 'sample_s': t2 - t1, 'total_s': t2 - t0}
    scores = np.zeros((b,), np.float32)
    if return_torch:
      return out, torch.zeros((b,), dtype=torch.float32, device=dev)
    return out.cpu().numpy(), scores

  def _jax_noise(self, seed: int, b: int):
    """Device-resident (init_z, noise) of jax_random.reference_noise, cached for the last (seed, b)."""
    key = (int(seed), int(b))
    if getattr(self, '_jax_noise_key', None) != key:
      from . import jax_random
      t, n = self.targets_length, self.audio_codec.n_dims
      steps = self.spec.diffusion.sampler.schedule.num_steps
      z, nz = jax_random.reference_noise(seed, (b, t, n), steps)
      self._jax_noise_val = (self._torch.as_tensor(z).to(self.device), self._torch.as_tensor(nz).to(self.device))
      self._jax_noise_key = key
    return self._jax_noise_val

  # -- InferSong.process segment loop (beam/evaluation.py:161-223) ---------------------
  def predict_sequence(self, segments_tokens: Sequence[np.ndarray], seed: int = 0,
                       always_mask_context: bool = False, init_context: Optional[np.ndarray] = None,
                       first_segment_index: int = 0, return_timing: bool = False, rng: str = 'philox',
                       return_torch: bool = False):
    """Synthesize a whole song: segments of int32 [inputs_length] (or [1, L]).

    Segment 0 runs with context zeros + mask 0 (beam/evaluation.py:195-198);
    segment i > 0 gets the previous PREDICTION (mel units) with mask 1
    (:194,199-205); ``always_mask_context`` masks every segment (:66-68).
    ``init_context`` [1, C, n] (NumPy or a device tensor; + ``first_segment_index`` > 0)
    resumes a song in the middle: the chained multi-GPU hand-off (sharding.py) uses it.
    Returns float32 [1, T*K, n] (NumPy; with ``return_torch`` the device tensor, so that the
    hand-off message never leaves the GPU); with ``return_timing`` also a dict with the
    reference's own metric (evaluation.py:217-220,244-250).
    """
    torch = self._torch
    n = self.audio_codec.n_dims
    c_len = self.targets_context_length
    pred = None
    if c_len is not None:
      pred = torch.zeros((1, c_len, n), dtype=torch.float32, device=self.device)
      if init_context is not None:
        pred = _to_device(torch, init_context, self.device, torch.float32).reshape(1, c_len, n)
    outs, seconds = [], []
    for i, toks in enumerate(segments_tokens):
      gi = first_segment_index + i
      toks = np.asarray(toks, np.int32).reshape(1, -1)
      batch = {'encoder_input_tokens': toks}
      if c_len is not None:
        batch['encoder_continuous_inputs'] = pred
        no_ctx = always_mask_context or (i == 0 and init_context is None)
        batch['encoder_continuous_mask'] = (np.zeros if no_ctx else np.ones)((1, c_len), np.int32)
      tick = time.perf_counter()
      out, _ = self.predict(batch, seed=seed, segment=gi, return_torch=True, rng=rng)
      if i != 0:
        seconds.append(time.perf_counter() - tick)
      if c_len is not None:
        pred = out[:1]
      outs.append(out[:1])
    full = torch.cat(outs, dim=1)
    if not return_torch:
      full = full.cpu().numpy()
    if not return_timing:
      return full
    seconds_per_chunk = self.targets_length * (self.audio_codec.hop_size / self.audio_codec.sample_rate)
    per_chunk = float(np.mean(seconds)) if seconds else float('nan')
    return full, {'prediction_seconds_per_chunk': per_chunk,
                  'predictions_seconds_per_audio_second': per_chunk / seconds_per_chunk}


  # ---- MIDI in -----------------------------------------------------
  def tokenize_note_sequence(self, ns, on_too_long: str = 'error'):
    """NoteSequence -> list of int32 [1, inputs_length] segment inputs through the reference's
    full-song pipeline (frontend/tokenizer.py; tasks.py:405-464 with full_song_eval=True)."""
    from .frontend import tokenizer
    cfg = tokenizer.FrontendConfig(sample_rate=self.audio_codec.sample_rate, hop_size=self.audio_codec.hop_size,
                                   segment_frames=self.targets_length, inputs_length=self.inputs_length)
    return tokenizer.note_sequence_to_model_inputs(ns, cfg, on_too_long=on_too_long)

  def synthesize_note_sequence(self, ns, seed: int = 0, **kw):
    """Notes -> mel frames of the whole song, float32 [1, K * targets_length, n_dims] (the tail past
    ns.total_time is the padding of the last segment).  kw: predict_sequence options."""
    return self.predict_sequence(self.tokenize_note_sequence(ns), seed=seed, **kw)

  def synthesize_midi(self, path: str, seed: int = 0, **kw):
    """Standard MIDI File -> mel frames (the reference's notebooks read MIDI with
    note_seq.midi_file_to_note_sequence; frontend/midi_io.py restates that reader)."""
    from .frontend import midi_io
    return self.synthesize_note_sequence(midi_io.midi_file_to_note_sequence(path), seed=seed, **kw)


def _to_numpy(x) -> np.ndarray:
  if isinstance(x, np.ndarray):
    return x
  if hasattr(x, 'detach'):
    return x.detach().cpu().numpy()
  return np.asarray(x)


def _to_device(torch, x, dev, dtype):
  if isinstance(x, torch.Tensor):
    return x.to(device=dev, dtype=dtype).contiguous()
  return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(dev)
