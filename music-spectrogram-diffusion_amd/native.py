"""ctypes binding of the C ABI in include/msd_amd.h (libmsd_amd.so).

This is the ONLY compute path of the package: there is no CPU or PyTorch
fallback.  ``load()`` raises ``NativeLibraryError`` when the HIP library has not
been built (run ``python music-spectrogram-diffusion_amd/build_native.py`` or
``__graft_entry__.build()``), and every wrapper converts a non-zero msd_status
into the Python exception the reference raises for the same condition
(ValueError for unknown sampler/schedule/..., see msd_amd.h).

PyTorch is plumbing here: device memory (``tensor.data_ptr()``), streams and
``torch.distributed``; no torch op is on the hot path.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libmsd_amd.so')
LIB_PATHS = {'f16': LIB_PATH, 'bf16': os.path.join(_HERE, 'csrc', 'libmsd_amd_bf16.so')}

MSD_PREC_F16 = 0      # one IEEE-half plane per operand
MSD_PREC_F16X3 = 1    # hi + lo half planes, three MFMAs per product (the parity mode, the default)
MSD_PREC_BF16 = 2     # one bfloat16 plane                       } libmsd_amd_bf16.so; msd_create of the other
MSD_PREC_BF16X3 = 3   # hi + lo bfloat16 planes                  } build answers MSD_ERR_UNSUPPORTED
ABI_VERSION = 6        # MSD_AMD_ABI_VERSION of include/msd_amd.h (tests/test_abi.py)
MSD_SAMPLER_DDPM = 0
MSD_SAMPLER_DDIM = 1
MAX_KERNEL_CLASSES = 16

# precision name -> (msd_precision value, plane format).  The plane format is a property of the LIBRARY build
# (csrc/common.h): libmsd_amd.so holds operand planes in IEEE half -- 'f16x3' (hi + lo planes, 22 significand bits,
# three MFMAs per product: the parity mode and the default) and 'f16' (one plane) --, libmsd_amd_bf16.so in
# bfloat16 -- 'bf16x3' / 'bf16': 16 / 8 significand bits but float32's exponent range, for weights or activations
# beyond the half range (|w| >= 128, |x| >= 131008).
PRECISIONS = {'f16': MSD_PREC_F16, 'f16x3': MSD_PREC_F16X3, 'bf16': MSD_PREC_BF16, 'bf16x3': MSD_PREC_BF16X3}
_PLANES_OF = {MSD_PREC_F16: 'f16', MSD_PREC_F16X3: 'f16', MSD_PREC_BF16: 'bf16', MSD_PREC_BF16X3: 'bf16'}


def plane_format(precision: str) -> str:
  if precision not in PRECISIONS:
    raise ValueError('precision must be one of %s' % sorted(PRECISIONS))
  return 'bf16' if precision.startswith('bf16') else 'f16'


# every symbol include/msd_amd.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = (
    'msd_version', 'msd_device_count', 'msd_create', 'msd_destroy', 'msd_last_error',
    'msd_num_weights', 'msd_weight_info', 'msd_set_weight', 'msd_finalize_weights',
    'msd_encode', 'msd_sample', 'msd_reset_graph', 'msd_decoder_pass', 'msd_fill_normal', 'msd_get_schedule',
    'msd_debug_read', 'msd_profile_steps', 'msd_op_gemm_h16', 'msd_op_gemm_bf16', 'msd_op_gemm_f32',
    'msd_op_attention', 'msd_op_attention_qp', 'msd_op_attention_split', 'msd_op_sampler_step', 'msd_op_residual_norm_gemm', 'msd_op_geglu',
    'msd_op_qkv', 'msd_op_final_proj')


class NativeLibraryError(RuntimeError):
  pass


class RangeError(ArithmeticError):
  """MSD_ERR_RANGE: an activation left the range of the IEEE-half operand planes (|x| > 65504) during the call;
  its result is invalid.  The bfloat16-plane precisions ('bf16x3') have float32's exponent range."""


MSD_SCHEDULE_COSINE, MSD_SCHEDULE_LINEAR = 0, 1
MSD_OUTPUT_EPS, MSD_OUTPUT_X0, MSD_OUTPUT_V = 0, 1, 2
MSD_LOGVAR_LARGE, MSD_LOGVAR_SMALL, MSD_LOGVAR_MEDIUM = 0, 1, 2
SCHEDULES = {'cosine': MSD_SCHEDULE_COSINE, 'linear': MSD_SCHEDULE_LINEAR}
MODEL_OUTPUTS = {'eps': MSD_OUTPUT_EPS, 'x0': MSD_OUTPUT_X0, 'v': MSD_OUTPUT_V}


class MsdConfig(ctypes.Structure):
  """msd_config of include/msd_amd.h (ABI 6), field for field."""
  _fields_ = [(n, ctypes.c_int32) for n in (
      'struct_size', 'has_context', 'vocab_size', 'emb_dim', 'num_heads', 'head_dim',
      'mlp_dim', 'num_encoder_layers', 'num_decoder_layers', 'inputs_length',
      'targets_length', 'context_length', 'n_dims', 'num_steps', 'sampler', 'clip_x0',
      'context_terminal_relative', 'precision', 'max_batch')] + [
          (n, ctypes.c_float) for n in (
              'max_decoder_noise_time', 'cfg_weight', 'feature_min', 'feature_max')] + [
      ('model_output', ctypes.c_int32), ('logvar_type', ctypes.c_int32), ('logvar_frac', ctypes.c_float),
      ('sampler_schedule', ctypes.c_int32), ('sampler_schedule_start', ctypes.c_float),
      ('sampler_schedule_stop', ctypes.c_float), ('train_schedule', ctypes.c_int32),
      ('train_schedule_start', ctypes.c_float), ('train_schedule_stop', ctypes.c_float),
      ('train_schedule_num_steps', ctypes.c_int32), ('cross_attend_sum', ctypes.c_int32),
      ('attn_q_planes', ctypes.c_int32), ('attn_p_planes', ctypes.c_int32), ('graph_steps', ctypes.c_int32),
      ('weight_prefetch', ctypes.c_int32),
      ('dedup_layer0', ctypes.c_int32), ('cross_key_split', ctypes.c_int32), ('keep_raw_weights', ctypes.c_int32),
      ('kv_touch_ahead', ctypes.c_int32),
      ('cross_merge_in_launch', ctypes.c_int32), ('cross_q_fold', ctypes.c_int32),
      ('mlp_in_persistent', ctypes.c_int32)]


# msd_config only ever grows at its end, so an OLDER library can be driven by passing it the struct size it knows
# (same-box A/B of a previous round's binary through MSD_AMD_LIB: tools/ab/); the newer fields are then simply not seen.
ABI_STRUCT_SIZES = {4: MsdConfig.weight_prefetch.offset + 4, 5: MsdConfig.kv_touch_ahead.offset + 4, 6: ctypes.sizeof(MsdConfig)}

_libs = {}


def load(planes: str = 'f16') -> ctypes.CDLL:
  """dlopen libmsd_amd.so (planes 'f16') or libmsd_amd_bf16.so ('bf16') and declare prototypes.  Fails loudly
  if missing."""
  if planes in _libs:
    return _libs[planes]
  LIB_PATH = LIB_PATHS[planes]
  # same-box A/B of two builds (tools/ab_bench.sh, the experiments build of tools/ubench/exp): MSD_AMD_LIB=<path>
  # replaces the half-plane library.  This is the ONLY environment variable the package reads; the library itself
  # reads none.  An older build may lack the newest entry points: they stay unbound (calling one raises
  # AttributeError) and a warning names them.
  override = os.environ.get('MSD_AMD_LIB') if planes == 'f16' else None
  if override:
    LIB_PATH = override
  if not os.path.exists(LIB_PATH):
    raise NativeLibraryError(
        'HIP library not built: %s is missing. Build it with '
        '`python music-spectrogram-diffusion_amd/build_native.py` (needs hipcc); '
        'there is no CPU fallback for this path.' % LIB_PATH)
  try:
    lib = ctypes.CDLL(LIB_PATH)
  except OSError as e:  # missing ROCm runtime etc.
    raise NativeLibraryError('cannot load %s: %s' % (LIB_PATH, e)) from e
  present = [sym for sym in EXPORTED_SYMBOLS if hasattr(lib, sym)]
  if len(present) != len(EXPORTED_SYMBOLS):
    missing = sorted(set(EXPORTED_SYMBOLS) - set(present))
    if not override:
      raise NativeLibraryError('%s does not export %s' % (LIB_PATH, missing))
    import warnings
    warnings.warn('MSD_AMD_LIB=%s does not export %s: those entry points are unbound' % (LIB_PATH, missing), RuntimeWarning)
  if override:
    if 'msd_version' not in present:   # (the warning above is no help for the one symbol this check needs)
      raise NativeLibraryError('MSD_AMD_LIB=%s does not export msd_version: cannot verify its ABI (this package binds ABI %d)'
                               % (LIB_PATH, ABI_VERSION))
    lib.msd_version.restype = ctypes.c_char_p
    ver = lib.msd_version() or b''
    lib._msd_abi = next((a for a in ABI_STRUCT_SIZES if ('abi %d' % a).encode() in ver), None)
    if lib._msd_abi is None:
      raise NativeLibraryError('MSD_AMD_LIB=%s is %r: this package binds ABI %d (and drives ABI %s through their msd_config size)'
                               % (LIB_PATH, ver, ABI_VERSION, sorted(set(ABI_STRUCT_SIZES) - {ABI_VERSION})))
  c = ctypes
  vp, i32, i64, u64, u32 = c.c_void_p, c.c_int, c.c_int64, c.c_uint64, c.c_uint32
  lib.msd_version.restype = c.c_char_p
  lib.msd_device_count.restype = i32
  lib.msd_create.argtypes = [c.POINTER(MsdConfig), c.POINTER(vp)]
  lib.msd_destroy.argtypes = [vp]
  lib.msd_destroy.restype = None
  lib.msd_last_error.argtypes = [vp]
  lib.msd_last_error.restype = c.c_char_p
  lib.msd_num_weights.argtypes = [vp]
  lib.msd_weight_info.argtypes = [vp, i32, c.POINTER(c.c_char_p), c.POINTER(i64), c.POINTER(i32)]
  lib.msd_set_weight.argtypes = [vp, c.c_char_p, vp, c.POINTER(i64), i32]
  lib.msd_finalize_weights.argtypes = [vp, vp]
  lib.msd_encode.argtypes = [vp, i32, vp, vp, vp, vp]
  lib.msd_sample.argtypes = [vp, i32, u64, u64, vp, vp, vp, vp]
  lib.msd_decoder_pass.argtypes = [vp, i32, i32, vp, i32, vp, vp]
  lib.msd_reset_graph.argtypes = [vp]
  lib.msd_fill_normal.argtypes = [u64, u64, u32, vp, i64, vp]
  lib.msd_get_schedule.argtypes = [vp, vp]
  lib.msd_debug_read.argtypes = [vp, c.c_char_p, vp, i64, c.POINTER(i64)]
  lib.msd_profile_steps.argtypes = [vp, i32, i32, c.POINTER(c.POINTER(c.c_char_p)),
                                    c.POINTER(c.c_double), c.POINTER(i64), vp]
  if 'msd_op_gemm_h16' in present:
    lib.msd_op_gemm_h16.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp]
  lib.msd_op_gemm_bf16.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp]
  lib.msd_op_gemm_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp]
  lib.msd_op_attention.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
  if 'msd_op_attention_qp' in present:
    lib.msd_op_attention_qp.argtypes = [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
  if 'msd_op_attention_split' in present:
    lib.msd_op_attention_split.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
  lib.msd_op_sampler_step.argtypes = [c.POINTER(MsdConfig), i32, vp, vp, vp, vp, vp, i64, vp]
  lib.msd_op_residual_norm_gemm.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
  lib.msd_op_geglu.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
  lib.msd_op_qkv.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
  lib.msd_op_final_proj.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
  for name in present:
    fn = getattr(lib, name)
    if name not in ('msd_version', 'msd_last_error', 'msd_destroy'):
      fn.restype = i32
  _libs[planes] = lib
  return lib


_EXC = {1: ValueError, 2: KeyError, 3: ValueError, 4: RuntimeError, 5: RuntimeError,
        6: NotImplementedError, 7: RangeError}


def _check(lib, handle, rc, what):
  if rc == 0:
    return
  msg = lib.msd_last_error(handle).decode('utf-8', 'replace') if handle else ''
  raise _EXC.get(rc, RuntimeError)('%s failed (msd_status %d): %s' % (what, rc, msg))


def _ptr(t) -> Optional[int]:
  """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
  if t is None:
    return None
  if isinstance(t, np.ndarray):
    return t.ctypes.data
  return t.data_ptr()


class NativeModel:
  """Owns one ``msd_model*`` on the current HIP device."""

  def __init__(self, cfg: MsdConfig, planes: Optional[str] = None):
    """`planes` (which library build) follows from cfg.precision; passing the other build's name is an error the
    library itself reports (msd_create -> MSD_ERR_UNSUPPORTED -> NotImplementedError)."""
    if planes is None:
      if cfg.precision not in _PLANES_OF:
        raise ValueError('unknown msd_precision %r' % (cfg.precision,))
      planes = _PLANES_OF[cfg.precision]
    self.lib = load(planes)
    self.planes = planes
    self.cfg = cfg
    self.handle = ctypes.c_void_p()
    cfg.struct_size = ABI_STRUCT_SIZES[getattr(self.lib, '_msd_abi', None) or ABI_VERSION]
    rc = self.lib.msd_create(ctypes.byref(cfg), ctypes.byref(self.handle))
    if rc != 0:
      msg = self.lib.msd_last_error(self.handle).decode() if self.handle else 'invalid config'
      if self.handle:
        self.lib.msd_destroy(self.handle)
        self.handle = ctypes.c_void_p()
      raise _EXC.get(rc, RuntimeError)('msd_create failed (msd_status %d): %s' % (rc, msg))

  def close(self):
    if getattr(self, 'handle', None):
      self.lib.msd_destroy(self.handle)
      self.handle = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass

  # -- weights ---------------------------------------------------------------
  def weight_specs(self) -> List[Tuple[str, Tuple[int, ...]]]:
    out = []
    for i in range(self.lib.msd_num_weights(self.handle)):
      name = ctypes.c_char_p()
      shape = (ctypes.c_int64 * 2)()
      ndim = ctypes.c_int()
      _check(self.lib, self.handle,
             self.lib.msd_weight_info(self.handle, i, ctypes.byref(name), shape, ctypes.byref(ndim)),
             'msd_weight_info')
      out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value))))
    return out

  def set_weight(self, name: str, array):
    """array: float32 numpy array or torch tensor (host or device), C-contiguous."""
    if isinstance(array, np.ndarray):
      array = np.ascontiguousarray(array, dtype=np.float32)
      shape = array.shape
    else:
      array = array.contiguous().float()
      shape = tuple(array.shape)
    cshape = (ctypes.c_int64 * max(len(shape), 1))(*shape)
    _check(self.lib, self.handle,
           self.lib.msd_set_weight(self.handle, name.encode(), _ptr(array), cshape, len(shape)),
           'msd_set_weight(%s)' % name)

  def load_weights(self, params: Dict[str, np.ndarray], stream: int = 0):
    specs = dict(self.weight_specs())
    missing = [k for k in specs if k not in params]
    if missing:
      raise KeyError('parameter dict is missing %d weights, e.g. %s' % (len(missing), missing[:3]))
    for name in specs:
      self.set_weight(name, params[name])
    _check(self.lib, self.handle, self.lib.msd_finalize_weights(self.handle, stream),
           'msd_finalize_weights')

  # -- hot path ----------------------------------------------------------------
  def encode(self, batch: int, tokens, ctx=None, ctx_mask=None, stream: int = 0):
    _check(self.lib, self.handle,
           self.lib.msd_encode(self.handle, batch, _ptr(tokens), _ptr(ctx), _ptr(ctx_mask), stream),
           'msd_encode')

  def sample(self, batch: int, out, seed: int = 0, stream_id: int = 0, init_z=None,
             noise=None, stream: int = 0):
    _check(self.lib, self.handle,
           self.lib.msd_sample(self.handle, batch, seed, stream_id, _ptr(init_z), _ptr(noise),
                               _ptr(out), stream), 'msd_sample')

  def reset_graph(self):
    _check(self.lib, self.handle, self.lib.msd_reset_graph(self.handle), 'msd_reset_graph')

  def decoder_pass(self, batch: int, step_index: int, z, include_conditioning: bool, eps_out,
                   stream: int = 0):
    _check(self.lib, self.handle,
           self.lib.msd_decoder_pass(self.handle, batch, step_index, _ptr(z),
                                     int(bool(include_conditioning)), _ptr(eps_out), stream),
           'msd_decoder_pass')

  def schedule(self) -> np.ndarray:
    out = np.zeros((self.cfg.num_steps, 8), np.float32)
    _check(self.lib, self.handle, self.lib.msd_get_schedule(self.handle, out.ctypes.data),
           'msd_get_schedule')
    return out

  def debug_read(self, buffer: str, max_elems: Optional[int] = None) -> np.ndarray:
    n = ctypes.c_int64()
    probe = np.zeros(1, np.float32)
    _check(self.lib, self.handle,
           self.lib.msd_debug_read(self.handle, buffer.encode(), probe.ctypes.data, 0, ctypes.byref(n)),
           'msd_debug_read')
    count = n.value if max_elems is None else min(n.value, max_elems)
    out = np.zeros(count, np.float32)
    _check(self.lib, self.handle,
           self.lib.msd_debug_read(self.handle, buffer.encode(), out.ctypes.data, count, ctypes.byref(n)),
           'msd_debug_read')
    return out

  def profile_steps(self, batch: int, n_steps: int, stream: int = 0) -> Dict[str, Tuple[float, int]]:
    names = ctypes.POINTER(ctypes.c_char_p)()
    ms = (ctypes.c_double * MAX_KERNEL_CLASSES)()
    launches = (ctypes.c_int64 * MAX_KERNEL_CLASSES)()
    _check(self.lib, self.handle,
           self.lib.msd_profile_steps(self.handle, batch, n_steps, ctypes.byref(names), ms,
                                      launches, stream), 'msd_profile_steps')
    out = {}
    i = 0
    while i < MAX_KERNEL_CLASSES and names[i]:
      out[names[i].decode()] = (float(ms[i]), int(launches[i]))
      i += 1
    return out


def fill_normal(out, seed: int, stream_id: int, subseq: int, stream: int = 0):
  lib = load()
  rc = lib.msd_fill_normal(seed, stream_id, subseq, _ptr(out), out.numel(), stream)
  if rc:
    raise RuntimeError('msd_fill_normal failed (%d)' % rc)


def op_gemm_h16(precision: str, a, w, c, stream: int = 0):
  """C = A.W with 16-bit operand planes in the format `precision` names."""
  lib = load(plane_format(precision))
  m, k = a.shape
  n = w.shape[1]
  rc = lib.msd_op_gemm_h16(PRECISIONS[precision], _ptr(a), _ptr(w), _ptr(c), m, n, k, stream)
  if rc:
    raise _EXC.get(rc, RuntimeError)('msd_op_gemm_h16 failed (%d)' % rc)


op_gemm_bf16 = op_gemm_h16   # ABI <= 2 name


def op_gemm_f32(a, w, c, stream: int = 0):
  lib = load()
  m, k = a.shape
  n = w.shape[1]
  rc = lib.msd_op_gemm_f32(_ptr(a), _ptr(w), _ptr(c), m, n, k, stream)
  if rc:
    raise _EXC.get(rc, RuntimeError)('msd_op_gemm_f32 failed (%d)' % rc)


def op_attention(precision: str, q, k, v, o, heads: int, n_keys_valid: Optional[int] = None,
                 stream: int = 0, qp: int = 0):
  """qp: query-side single-plane switches (bit 0: Q, bit 1: P); the decoder runs 'f16x3' with qp = 3."""
  lib = load(plane_format(precision))
  n_q, n_keys = q.shape[0], k.shape[0]
  nv = n_keys if n_keys_valid is None else n_keys_valid
  rc = lib.msd_op_attention_qp(PRECISIONS[precision], qp, _ptr(q), _ptr(k), _ptr(v), _ptr(o), n_q,
                               n_keys, nv, heads, stream)
  if rc:
    raise _EXC.get(rc, RuntimeError)('msd_op_attention failed (%d)' % rc)


def op_attention_split(precision: str, q, k, v, o, heads: int, ksplit: int, merge_in_launch: bool, repeats: int = 1,
                       n_keys_valid: Optional[int] = None, stream: int = 0, qp: int = 0):
  """op_attention with the key axis split over `ksplit` blocks per (head, query tile); the partials are merged by the
  separate merge launch or -- merge_in_launch -- inside the attention launch by the last block to arrive; `repeats`
  launches back to back (the in-launch merge's arrival counters must be zero again after each)."""
  lib = load(plane_format(precision))
  n_q, n_keys = q.shape[0], k.shape[0]
  nv = n_keys if n_keys_valid is None else n_keys_valid
  rc = lib.msd_op_attention_split(PRECISIONS[precision], qp, ksplit, int(bool(merge_in_launch)), repeats, _ptr(q), _ptr(k),
                                  _ptr(v), _ptr(o), n_q, n_keys, nv, heads, stream)
  if rc:
    raise _EXC.get(rc, RuntimeError)('msd_op_attention_split failed (%d)' % rc)


def _op_check(rc, what):
  if rc:
    raise _EXC.get(rc, RuntimeError)('%s failed (msd_status %d)' % (what, rc))


def op_sampler_step(cfg: MsdConfig, step_index: int, z, out_cond, out_uncond, noise, z_out, stream: int = 0):
  """One sampler update (msd_op_sampler_step).  Tensors: float32 device tensors of equal numel."""
  lib = load()
  cfg.struct_size = ABI_STRUCT_SIZES[getattr(lib, '_msd_abi', None) or ABI_VERSION]
  _op_check(lib.msd_op_sampler_step(ctypes.byref(cfg), step_index, _ptr(z), _ptr(out_cond), _ptr(out_uncond),
                                    _ptr(noise), _ptr(z_out), z.numel(), stream), 'msd_op_sampler_step')


def op_residual_norm_gemm(folded: bool, x_in, a, w1, gamma, film_scale, film_bias, w2, x_out, h_out,
                          stream: int = 0):
  lib = load()
  m, k = a.shape
  d, n = w2.shape
  _op_check(lib.msd_op_residual_norm_gemm(int(folded), _ptr(x_in), _ptr(a), _ptr(w1), _ptr(gamma),
                                          _ptr(film_scale), _ptr(film_bias), _ptr(w2), _ptr(x_out), _ptr(h_out),
                                          m, k, d, n, stream), 'msd_op_residual_norm_gemm')


def op_geglu(a, wi0, wi1, out, stream: int = 0):
  lib = load()
  m, k = a.shape
  _op_check(lib.msd_op_geglu(_ptr(a), _ptr(wi0), _ptr(wi1), _ptr(out), m, k, wi0.shape[1], stream), 'msd_op_geglu')


def op_qkv(a, wq, wk, wv, q, k_out, v, seg_len: int, stream: int = 0):
  lib = load()
  m, k = a.shape
  _op_check(lib.msd_op_qkv(_ptr(a), _ptr(wq), _ptr(wk), _ptr(wv), _ptr(q), _ptr(k_out), _ptr(v), m, k,
                           wq.shape[1], seg_len, stream), 'msd_op_qkv')


def op_final_proj(x, gamma, w, out, stream: int = 0):
  lib = load()
  m, d = x.shape
  _op_check(lib.msd_op_final_proj(_ptr(x), _ptr(gamma), _ptr(w), _ptr(out), m, d, w.shape[1], stream),
            'msd_op_final_proj')
