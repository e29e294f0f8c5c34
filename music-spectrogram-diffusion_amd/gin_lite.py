"""A reader for the subset of gin the synthesis path depends on.

``gin-config`` is not installed here and only a handful of bindings decide the
hot path's shapes (SURVEY.md 8(b) "Config"): ``TASK_FEATURE_LENGTHS``, ``MODEL``,
``network.T5Config.*``, ``diffusion_utils.{DiffusionConfig, SamplerConfig,
DiffusionSchedule, ClassifierFreeGuidanceConfig}.*``, ``AUDIO_CODEC`` and
``NUM_VELOCITY_BINS``.  This module parses gin *text* -- either raw ``.gin``
files with ``include`` lines (what ``parse_training_gin_file`` reads,
inference.py:32-65) or the flat operative string ``gin.config_str()`` returns
(what ``InferenceModel.__init__`` is given, inference.py:71-88) -- into a flat
``{key: python value}`` dict, and turns that into a ``config.ModelSpec``.

It is not a re-implementation of gin: references (``@x``, ``@x()``, ``%MACRO``)
are kept as ``Ref`` tokens and resolved only where the spec needs them; unknown
bindings are kept verbatim and ignored.
"""
from __future__ import annotations

import ast
import dataclasses
import os
import re
import warnings
from typing import Any, Dict, List, Optional, Sequence

from . import config as config_lib


@dataclasses.dataclass(frozen=True)
class Ref:
  """``@scope/name`` (configurable reference, ``call`` if ``()``) or ``%MACRO``."""
  kind: str   # '@' or '%'
  name: str
  call: bool = False

  def __repr__(self):
    return '%s%s%s' % (self.kind, self.name, '()' if self.call else '')


_REF_RE = re.compile(r'(@[\w./]+(?:\(\))?|%[\w.]+)')


def _parse_value(text: str) -> Any:
  text = text.strip()
  refs: List[Ref] = []

  def sub(m):
    tok = m.group(0)
    if tok[0] == '@':
      call = tok.endswith('()')
      refs.append(Ref('@', tok[1:-2] if call else tok[1:], call))
    else:
      refs.append(Ref('%', tok[1:]))
    return '__gin_ref_%d__' % (len(refs) - 1)

  py = _REF_RE.sub(sub, text)
  if re.fullmatch(r'__gin_ref_\d+__', py):
    return refs[0]
  try:
    tree = ast.parse(py, mode='eval')
  except SyntaxError:
    return text

  def build(node):
    if isinstance(node, ast.Name):
      m = re.fullmatch(r'__gin_ref_(\d+)__', node.id)
      if m:
        return refs[int(m.group(1))]
      if node.id in ('True', 'False', 'None'):
        return {'True': True, 'False': False, 'None': None}[node.id]
      return node.id
    if isinstance(node, ast.Constant):
      return node.value
    if isinstance(node, ast.Tuple):
      return tuple(build(e) for e in node.elts)
    if isinstance(node, ast.List):
      return [build(e) for e in node.elts]
    if isinstance(node, ast.Dict):
      return {build(k): build(v) for k, v in zip(node.keys, node.values)}
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
      return -build(node.operand)
    return ast.unparse(node)

  return build(tree.body)


def _logical_lines(text: str):
  """Yield (indent, line) with comments stripped and bracketed values joined."""
  buf, depth, indent0 = '', 0, 0
  for raw in text.splitlines():
    line = re.sub(r'(?<![\'"\w])#.*$', '', raw) if '#' in raw else raw
    if not line.strip():
      continue
    if not buf:
      indent0 = len(line) - len(line.lstrip())
    piece = line.strip()
    # `gin.config_str()` wraps any binding longer than 80 columns with a trailing backslash
    # ("a.b = \\" / "    @scope/c()"): the backslash continues the line and is not part of the value
    continued = piece.endswith('\\')
    if continued:
      piece = piece[:-1].rstrip()
    buf += (' ' if buf and piece else '') + piece
    depth = sum(buf.count(c) for c in '([{') - sum(buf.count(c) for c in ')]}')
    if depth <= 0 and not continued:
      yield indent0, buf
      buf = ''
  if buf:
    yield indent0, buf


def parse(text: str, search_paths: Sequence[str] = (), _seen=None) -> Dict[str, Any]:
  """Parse gin text into a flat ``{binding_key: value}`` dict (later wins)."""
  out: Dict[str, Any] = {}
  _seen = _seen if _seen is not None else set()
  block: Optional[str] = None
  for indent, line in _logical_lines(text):
    if indent == 0:
      block = None
    if line.startswith(('import ', 'from ')):
      continue
    m = re.match(r"include\s+['\"](.+)['\"]", line)
    if m:
      path = _find(m.group(1), search_paths)
      if path is None:
        warnings.warn('gin_lite: include not found: %s' % m.group(1))
        continue
      if path in _seen:
        continue
      _seen.add(path)
      with open(path) as f:
        out.update(parse(f.read(), list(search_paths) + [os.path.dirname(path)], _seen))
      continue
    if indent == 0 and line.endswith(':') and '=' not in line:
      block = line[:-1].strip()
      continue
    if '=' not in line:
      continue
    key, value = line.split('=', 1)
    key = key.strip()
    if block is not None and indent > 0:
      key = block + '.' + key
    out[key] = _parse_value(value)
  return out


def _find(rel: str, search_paths: Sequence[str]) -> Optional[str]:
  cands = [rel] + [os.path.join(p, rel) for p in search_paths]
  # gin files name includes relative to the python package root, e.g.
  # 'music_spectrogram_diffusion/gin/tasks/base.gin': also try every suffix of
  # the path under each search root and its parents.
  parts = rel.split('/')
  for p in search_paths:
    root = os.path.abspath(p)
    for _ in range(8):
      for k in range(len(parts)):
        cands.append(os.path.join(root, *parts[k:]))
      root = os.path.dirname(root)
  for c in cands:
    if os.path.isfile(c):
      return os.path.abspath(c)
  return None


def to_config_str(bindings: Dict[str, Any]) -> str:
  """Flat, re-parseable operative string (stand-in for ``gin.config_str()``)."""
  return '\n'.join('%s = %r' % (k, v) for k, v in bindings.items()) + '\n'


# ---------------------------------------------------------------------------
# bindings -> ModelSpec
# ---------------------------------------------------------------------------
def _lookup(b: Dict[str, Any], cls: str, param: str, scope: Optional[str] = None):
  """Value of ``[scope/]...cls.param`` (gin selectors may be partially qualified)."""
  hits = []
  for k, v in b.items():
    sc, _, sel = k.rpartition('/')
    if sel.endswith(cls + '.' + param) or sel == cls + '.' + param:
      hits.append((sc, v))
  if scope is not None:
    for sc, v in hits:
      if sc == scope:
        return v
  for sc, v in hits:
    if sc == '':
      return v
  return None


def _resolve(b: Dict[str, Any], v):
  n = 0
  while isinstance(v, Ref) and v.kind == '%' and n < 16:
    v = b.get(v.name)
    n += 1
  return v


def model_spec_from_bindings(b: Dict[str, Any]) -> config_lib.ModelSpec:
  lengths = _resolve(b, b.get('TASK_FEATURE_LENGTHS'))
  if not isinstance(lengths, dict):
    raise ValueError('gin config does not bind TASK_FEATURE_LENGTHS')
  model = _resolve(b, b.get('MODEL'))
  if not isinstance(model, Ref):
    raise ValueError('gin config does not bind MODEL')
  model_cls = model.name.split('.')[-1]
  if model_cls not in ('DiffusionModel', 'ContextDiffusionModel'):
    raise ValueError('Unsupported MODEL for the diffusion hot path: %r' % (model,))

  nvb = _resolve(b, b.get('NUM_VELOCITY_BINS'))
  nvb = 1 if nvb is None else int(nvb)

  fields = {}
  for f in dataclasses.fields(config_lib.T5Config):
    v = _resolve(b, _lookup(b, 'T5Config', f.name))
    if v is None:
      continue
    if isinstance(v, Ref):  # vocab_size = @vocabularies.num_embeddings()
      if f.name == 'vocab_size':
        v = config_lib.num_embeddings(nvb)
      else:
        continue
    fields[f.name] = tuple(v) if isinstance(v, list) else v
  fields.setdefault('vocab_size', config_lib.num_embeddings(nvb))
  t5 = config_lib.T5Config(**fields)

  def schedule(scope, default):
    kw = {}
    for p in ('name', 'start', 'stop', 'num_steps'):
      v = _resolve(b, _lookup(b, 'DiffusionSchedule', p, scope))
      if v is not None:
        kw[p] = v
    if not kw:
      return default
    kw.setdefault('name', default.name)
    return config_lib.DiffusionSchedule(**kw)

  sampler_defaults = config_lib.SamplerConfig()
  skw = {}
  for p in ('name', 'clip_x0', 'logvar_type'):
    v = _resolve(b, _lookup(b, 'SamplerConfig', p))
    if v is not None and not isinstance(v, Ref):
      skw[p] = v
  sched_ref = _lookup(b, 'SamplerConfig', 'schedule')
  sscope = sched_ref.name.rpartition('/')[0] if isinstance(sched_ref, Ref) else 'sampler'
  skw['schedule'] = schedule(sscope or None, sampler_defaults.schedule)
  sampler = config_lib.SamplerConfig(**skw)

  ckw = {}
  for p in ('drop_condition_prob', 'eval_condition_weight'):
    v = _resolve(b, _lookup(b, 'ClassifierFreeGuidanceConfig', p))
    if v is not None and not isinstance(v, Ref):
      ckw[p] = float(v)
  dkw = {}
  for p in ('time_continuous_or_discrete', 'loss_norm', 'loss_type', 'model_output'):
    v = _resolve(b, _lookup(b, 'DiffusionConfig', p))
    if v is not None and not isinstance(v, Ref):
      dkw[p] = v
  tref = _lookup(b, 'DiffusionConfig', 'train_schedule')
  tscope = tref.name.rpartition('/')[0] if isinstance(tref, Ref) else 'train'
  diffusion = config_lib.DiffusionConfig(
      train_schedule=schedule(tscope or None, config_lib.DiffusionSchedule('cosine')),
      classifier_free_guidance=config_lib.ClassifierFreeGuidanceConfig(**ckw),
      sampler=sampler, **dkw)

  codec = _resolve(b, b.get('AUDIO_CODEC'))
  codec_name = codec.name.split('.')[-1] if isinstance(codec, Ref) else 'MelGAN'
  return config_lib.ModelSpec(model_cls, t5, diffusion,
                              {k: int(v) for k, v in lengths.items()},
                              audio_codec=codec_name, num_velocity_bins=nvb)


def spec_to_config_str(spec: config_lib.ModelSpec) -> str:
  """Operative gin-style string for a typed preset (round-trips through ``parse``)."""
  t5, d = spec.t5, spec.diffusion
  lines = [
      'TASK_FEATURE_LENGTHS = %r' % dict(spec.task_feature_lengths),
      'NUM_VELOCITY_BINS = %d' % spec.num_velocity_bins,
      'AUDIO_CODEC = @audio_codecs.%s()' % spec.audio_codec,
      'MODEL = @models.%s()' % spec.model,
  ]
  for f in dataclasses.fields(config_lib.T5Config):
    lines.append('network.T5Config.%s = %r' % (f.name, getattr(t5, f.name)))
  lines += [
      'diffusion_utils.DiffusionConfig.model_output = %r' % d.model_output,
      'diffusion_utils.DiffusionConfig.train_schedule = @train/diffusion_utils.DiffusionSchedule()',
      'diffusion_utils.DiffusionConfig.sampler = @diffusion_utils.SamplerConfig()',
      'diffusion_utils.DiffusionConfig.classifier_free_guidance = '
      '@diffusion_utils.ClassifierFreeGuidanceConfig()',
      'diffusion_utils.ClassifierFreeGuidanceConfig.eval_condition_weight = %r'
      % d.classifier_free_guidance.eval_condition_weight,
      'diffusion_utils.SamplerConfig.name = %r' % d.sampler.name,
      'diffusion_utils.SamplerConfig.clip_x0 = %r' % d.sampler.clip_x0,
      'diffusion_utils.SamplerConfig.logvar_type = %r' % d.sampler.logvar_type,
      'diffusion_utils.SamplerConfig.schedule = @sampler/diffusion_utils.DiffusionSchedule()',
      'train/diffusion_utils.DiffusionSchedule.name = %r' % d.train_schedule.name,
      'sampler/diffusion_utils.DiffusionSchedule.name = %r' % d.sampler.schedule.name,
      'sampler/diffusion_utils.DiffusionSchedule.num_steps = %r' % d.sampler.schedule.num_steps,
  ]
  for p in ('start', 'stop'):
    v = getattr(d.sampler.schedule, p)
    if v is not None:
      lines.append('sampler/diffusion_utils.DiffusionSchedule.%s = %r' % (p, v))
  return '\n'.join(lines) + '\n'
