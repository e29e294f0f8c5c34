"""Benchmark of the hot path: mel-frames/sec (+ xRTF) of 1000-step DDPM synthesis.

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N ranks, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: ONE 256-frame (5.12 s)
segment through ``InferenceModel.predict`` = encoders + cross-K/V cache + the full
N-step DDPM loop (CFG weight 5 -> two decoder passes per step) + un-scaling, with
the previous segment's prediction as context (segment-sequential,
beam/evaluation.py:161-223).  Workload = BASELINE.json configs[2]:
base_with_context, 1000-step DDPM, synthetic MIDI tokens, seeded synthetic
weights.  The warm-up segments absorb the one-time restore/graph capture exactly
like the reference excludes its first segment (beam/evaluation.py:217-220).

N > 1: one process per GPU.  --mode (sharding.py):
  replicas   (default) every rank synthesizes its own song: no message on the data path
  chained    ONE song of N*K segments cut into N contiguous chunks; rank r starts when rank r-1 hands
             over its last prediction (one 128 KiB device-to-device message, RCCL send/recv = one xGMI
             link).  Bit-identical to the sequential song; serial by construction (BASELINE config 4's
             partitioning for a single song)
  wavefront  K songs of N segments each, rank r runs segment r of every song while rank r-1 already
             works on the next song: all ranks busy after N-1 fill steps (efficiency K / (K+N-1))
  masked     one song of N*K segments, chunk heads run context-masked (no message, not parity-exact at
             the N-1 cuts: the reference's own i == 0 / always_mask_context behaviour there)
value = frames of all ranks / max-over-ranks time; per-GPU work is K segments in every mode ("weak").
The plain N > 1 line (default --mode replicas) also carries a `handoff` leg: BASELINE config 4's partitioning -- the
10-minute workload, 118 segments, as a wavefront of ceil(118 / N) songs x N segments with the device-to-device context
hand-off (per-rank segments capped by --handoff-max-segments) -- timed after the replicas region, with each rank's
idle fraction; a hang anywhere on that message path (first contact with RCCL point-to-point) ends the leg after
--handoff-timeout seconds and the line is printed with an `error` field instead of never.

The JSON line also carries
  roofline      dominant kernel class: algorithmic FLOP per launch / mean launch
                duration measured with hipEvents on the launch stream
                (msd_profile_steps), vs the dense 16-bit MFMA peak (f16 = bf16 rate);
  cpu_baseline  the torch-CPU float32 oracle ("port") timed on a bounded sample
                (encoders + a few DDPM steps, extrapolated linearly: per-step cost
                is constant) on rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 / f16 MFMA (MI355X_MICROARCH.md; not the 2:1-sparse 5 PF)
PEAK_HBM_GBS = 8000.0


def class_flops(spec, s_valid: float, passes: int):
  """Algorithmic FLOP per LAUNCH of each kernel class at batch 1:
  2MNK per GEMM, 2*H*T*S*d each for QK^T and PV; elementwise work ignored;
  f16x3 counts the product once (it is one fp32-class GEMM)."""
  t5 = spec.t5
  d, h, f = t5.emb_dim, t5.num_heads, t5.mlp_dim
  j = h * t5.head_dim
  t = spec.task_feature_lengths['targets']
  m = passes * t
  return {
      'gemm_qkv': 2.0 * m * 3 * j * d,
      'attn_self': 4.0 * passes * h * t * t * t5.head_dim,
      'gemm_attn_out': 2.0 * m * d * j,
      'gemm_cross_q': 2.0 * t * j * d,
      'attn_cross': 4.0 * h * t * s_valid * t5.head_dim,
      'gemm_cross_out': 2.0 * t * d * j,
      'gemm_mlp_in_geglu': 2.0 * m * 2 * f * d,
      'gemm_mlp_out': 2.0 * m * d * f,
      'final_proj_f32': 2.0 * m * 128 * d,
      'in_proj_f32': 2.0 * t * 128 * d,
  }


def class_bytes(spec, s_valid: float, passes: int, planes: int = 2):
  """ALGORITHMIC bytes per launch of each kernel class at batch 1: every operand read once, every
  result written once (16-bit planes: 2 B x `planes`; fp32: 4 B).  What a launch must move if nothing
  were fetched twice -- the denominator of the waste ratio against the PMC fabric traffic."""
  t5 = spec.t5
  d, h, f = t5.emb_dim, t5.num_heads, t5.mlp_dim
  j = h * t5.head_dim
  t = spec.task_feature_lengths['targets']
  m = passes * t
  pb = 2 * planes
  fp = 4
  return {
      'gemm_qkv': pb * (m * d + 3 * j * d + m * 3 * j) + fp * (m * d // 32 + 3 * j),
      'attn_self': pb * (m * 3 * j + m * j),
      'gemm_attn_out': pb * (m * j + d * j + m * d) + fp * (2 * m * d + m * d // 32),
      'gemm_cross_q': pb * (t * d + j * d + t * j) + fp * (t * d // 32),
      'attn_cross': pb * (t * j + 2 * s_valid * j) + fp * 4 * (t * j + 2 * t * h),
      'attn_cross_merge': fp * 4 * (t * j + 2 * t * h) + pb * t * j,
      'gemm_cross_out': pb * (t * j + d * j + t * d) + fp * (2 * t * d + t * d // 32),
      'gemm_mlp_in_geglu': pb * (m * d + 2 * f * d + m * f) + fp * (m * d // 32 + 2 * f),
      'gemm_mlp_out': pb * (m * f + d * f + m * d) + fp * (2 * m * d + m * d // 32),
      'final_proj_f32': fp * (m * d + d * 128 + m * 128),
      'sampler_step': fp * (4 * t * 128) + pb * t * 128,
      'in_proj_f32': pb * (t * 128 + d * 128 + m * d) + fp * (m * d + t * d),
  }


def class_work(spec, s_valid: float, passes: int, planes: int = 2):
  """(flops, bytes) per launch for every class name `classify_kernel` can return, including the classes of layer 0's
  half-M launches (S5) and the combined classes of an instantiation that serves two launch sites."""
  flops, abytes = class_flops(spec, s_valid, passes), class_bytes(spec, s_valid, passes, planes)
  f1, b1 = class_flops(spec, s_valid, 1), class_bytes(spec, s_valid, 1, planes)
  for c in ('gemm_qkv', 'gemm_attn_out'):   # layer 0 of a CFG step runs on ONE pass's rows
    flops[c + '_l0'], abytes[c + '_l0'] = f1[c], b1[c]
  ld = spec.t5.num_decoder_layers
  for a, b, na, nb in (('gemm_attn_out', 'gemm_cross_out', ld, ld), ('gemm_attn_out', 'gemm_mlp_out', ld - 1, ld)):
    flops[a + '+' + b] = (na * flops[a] + nb * flops[b]) / (na + nb)
    abytes[a + '+' + b] = (na * abytes[a] + nb * abytes[b]) / (na + nb)
  flops.setdefault('attn_cross_merge', 0.0)
  flops.setdefault('sampler_step', 0.0)
  return flops, abytes


def roofline_from_self_profile(spec, sp, passes, planes, graph_step_ms=None):
  """`roofline` fields from a self_profile() record: per class us / launches / TFLOP/s / fraction of the dense 16-bit
  MFMA peak, the dominant class (most algorithmic FLOP per step), and the whole step."""
  flops, abytes = class_work(spec, float(sp.get('s_valid_keys') or 0.0), passes, planes)
  fold_cross_q_work(flops, abytes, sp['per_class'])
  table = {}
  for c, e in sp['per_class'].items():
    tf = flops.get(c, 0.0) / (e['avg_us'] * 1e-6) / 1e12
    table[c] = {'us_per_launch': e['avg_us'], 'launches_per_step': e['launches_per_step'], 'kernel': e['kernel'],
                'gap_before_us': e.get('avg_gap_before_us'),
                'algorithmic_gflop': round(flops.get(c, 0.0) / 1e9, 4), 'tflops': round(tf, 2),
                'frac': round(tf / PEAK_BF16_TFLOPS, 5)}
  dom = max(table, key=lambda c: flops.get(c, 0.0) * (table[c]['launches_per_step'] or 0.0))
  step_flops = sum(flops.get(c, 0.0) * (e['launches_per_step'] or 0.0) for c, e in table.items())
  sum_us = sp['sum_kernel_us_per_step']
  whole = {'algorithmic_gflop': round(step_flops / 1e9, 2), 'sum_kernel_us': sum_us,
           'launches': round(sum(e['launches_per_step'] or 0.0 for e in table.values()), 1),
           'tflops_over_kernel_time': round(step_flops / (sum_us * 1e-6) / 1e12, 2),
           'frac_over_kernel_time': round(step_flops / (sum_us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 5)}
  for k in ('sum_gap_us_per_step', 'step_span_us', 'gap_between_steps_us'):
    if k in sp:
      whole[k] = sp[k]
  if graph_step_ms:
    whole['graph_ms'] = round(graph_step_ms, 4)
    whole['achieved_tflops_graph'] = round(step_flops / (graph_step_ms * 1e-3) / 1e12, 3)
    whole['frac_graph'] = round(step_flops / (graph_step_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 5)
    # kernels cannot outlast the wall clock that contains them: the traced child's kernels against ITS OWN segment
    child_ms = sp.get('child_sample_ms_per_segment')
    if child_ms:
      whole['kernel_time_over_child_step'] = round(sum_us / (child_ms / spec.diffusion.sampler.schedule.num_steps * 1e3), 4)
  return dom, table, whole, flops, abytes


def fold_cross_q_work(flops, abytes, classes):
  """Round 6 (S6, csrc/msd_api.hip decoder_layers): the cross-attention's query projection has no launch of its own -- one half
  rides on the QKV launch, the other runs beside the self-attention output projection.  Its ALGORITHMIC work (2 T J D per
  layer, the reference's: network.py:196-198) is counted with the output projection's launch, once."""
  if 'gemm_cross_q' in classes or 'gemm_cross_q' not in flops:
    return
  for c in ('gemm_attn_out', 'gemm_attn_out_l0'):
    if c in classes and c in flops:
      flops[c] += flops['gemm_cross_q']
      abytes[c] = abytes.get(c, 0) + abytes.get('gemm_cross_q', 0)


def library_hash(planes='f16'):
  """sha256 (first 16 hex) of the built HIP library: stamps which binary a profile was taken on."""
  import hashlib
  path = os.path.join(ROOT, 'music-spectrogram-diffusion_amd', 'csrc',
                      'libmsd_amd.so' if planes == 'f16' else 'libmsd_amd_bf16.so')
  if not os.path.exists(path):
    return None
  h = hashlib.sha256()
  with open(path, 'rb') as f:
    for chunk in iter(lambda: f.read(1 << 20), b''):
      h.update(chunk)
  return h.hexdigest()[:16]


def profile_roofline(kernel_class, args):
  """The committed rocprofv3 numbers for `kernel_class` (profiles/roofline.json, made by
  tools/profile_round.sh + tools/make_roofline.py from a kernel-trace pass and separate PMC passes of
  this same command): average duration, MFMA-pipe utilisation, fabric bytes.  rocprofv3 cannot run
  inside the timed process, so these are the committed measurement; the entry carries the hash of the
  library it was taken on and `matches_binary` says whether that is the binary being timed now."""
  path = os.path.join(ROOT, 'profiles', 'roofline.json')
  if not os.path.exists(path):
    return None, 'profiles/roofline.json not present'
  if args.preset != 'base_with_context' or args.batch != 1 or args.precision != 'f16x3' or args.cfg_weight == 1.0:
    return None, 'the profile was taken on base_with_context, B=1, f16x3, CFG'
  with open(path) as f:
    t = json.load(f)
  cls = t.get('per_class', {}).get(kernel_class)
  if cls is None:
    return None, 'class %s not in profiles/roofline.json' % kernel_class
  out = dict(cls)
  out['profile_library_sha'] = t.get('library_sha')
  out['matches_binary'] = t.get('library_sha') == library_hash()
  return out, t.get('source', '')


# ---- kernel classes of a rocprofv3 --kernel-trace --stats table ----------------------------------------------------------
# (substring of the demangled kernel name, class) -- first match wins.  Round 3 renamed the 16-bit-plane kernels
# (gemm_bf16_* -> gemm_h16_*, EpiStoreBf16 -> EpiStoreH16); the round-2 names stay so that the committed r02 traces
# still classify.  tools/make_roofline.py uses the same table (it imports this module).
KERNEL_CLASSES = [
    ('gemm_h16_dual_kernel', 'gemm_attn_out+cross_q'),      # (round 3's hoisted query projection: committed traces only)
    ('gemm_h16_splitk_kernel', 'gemm_mlp_out'),
    ('EpiGeglu', 'gemm_mlp_in_geglu'), ('EpiQKV', 'gemm_qkv'),
    ('64, 32, 4, msd::EpiResidualNorm', 'gemm_mlp_out'), ('64, 32, 4, EpiResidualNorm', 'gemm_mlp_out'),
    ('32, 32, 4, msd::EpiResidualNorm', 'gemm_attn_out+gemm_cross_out'), ('32, 32, 4, EpiResidualNorm', 'gemm_attn_out+gemm_cross_out'),
    ('32, 32, 4, msd::EpiStoreBf16', 'gemm_cross_q'), ('32, 32, 4, EpiStoreBf16', 'gemm_cross_q'),
    ('32, 32, 4, msd::EpiStoreH16', 'gemm_cross_q'), ('32, 32, 4, EpiStoreH16', 'gemm_cross_q'),
    ('attention_merge_kernel', 'attn_cross_merge'),
    ('attention_kernel<2, 2, 1', 'attn_self'), ('attention_kernel<2, 2, 2', 'attn_cross'),
    ('final_proj_f32_kernel', 'final_proj_f32'), ('EpiInProj', 'in_proj_f32'), ('sampler_step_kernel', 'sampler_step'),
]


def normalise_kernel(name):
  """kernel name as tools/pmc_summary.py prints it: no return type, no namespace, no argument list"""
  return name.split('(')[0].replace('void msd::', '').replace('msd::', '').strip()


def classify_kernel(name, step_kernels=None):
  """Class of a kernel INSTANTIATION.  `step_kernels`: the normalised names of the instantiations the DDPM-step graph
  replays; anything else (the encoders run the same templates at other shapes, without the weight prefetch:
  `EpiGeglu<2>, 0`, `attention_kernel<2, 2, 2, 0, 0>`, ...) is NOT part of a class -- round 3 classified by substring
  only and averaged the encoder's M = 2048 launches into the decoder's counters."""
  if step_kernels is not None and normalise_kernel(name) not in step_kernels:
    return None
  if 'gemm_h16_dual_kernel' in name and re.search(r'>, \d+, \d+, \d+, (msd::)?Epi', name):
    # round 6 (S6): two problems, each with its own tile shape, in one launch -- QKV + the first half of the folded
    # cross-attention query projection, or the self-attention output projection + its second half
    if 'EpiQKV' in name:
      l0 = step_kernels is not None and 'dual_kernel<2, 64, 64, 3' in name and any('dual_kernel<2, 64, 96, 3' in k and 'EpiQKV' in k for k in step_kernels)
      return 'gemm_qkv_l0' if l0 else 'gemm_qkv'
    return 'gemm_attn_out_l0' if 'EpiResidualNorm<2, true' in name else 'gemm_attn_out'
  if step_kernels is not None and 'EpiResidualNorm' in name:
    # round 5: layer 0's self-attention block runs on one CFG pass's rows (S5): its out-projection is the duplicating
    # epilogue EpiResidualNorm<2, true> (M = 256) -- a class of its own, one launch per step
    if 'EpiResidualNorm<2, true' in name:
      return 'gemm_attn_out_l0'
    if any('32, 48, 4' in k for k in step_kernels):
      # round 4: MLP-out on the 32 x 48 tile, attention-out on 64 x 32, cross-out on 32 x 32: one class each
      for sub, cls in (('32, 48, 4', 'gemm_mlp_out'), ('64, 32, 4', 'gemm_attn_out'), ('32, 32, 4', 'gemm_cross_out')):
        if sub in name:
          return cls
  if step_kernels is not None and 'EpiQKV' in name and '64, 64, 3' in name and any('64, 96, 3' in k and 'EpiQKV' in k for k in step_kernels):
    return 'gemm_qkv_l0'      # layer 0's QKV projection on M = 256 rows (S5): 64 x 64 tiles, one launch per step
  for sub, cls in KERNEL_CLASSES:
    if sub in name:
      return cls
  return None


def step_kernel_names(stats_csv):
  """The instantiations that belong to the DDPM step: the sampler runs exactly once per step, so every kernel of the
  step graph has at least that many calls in a trace, while an encoder or load-time launch has a few dozen."""
  import csv
  rows = list(csv.DictReader(open(stats_csv)))
  per_step = [int(r['Calls']) for r in rows if 'sampler_step_kernel' in r['Name']]
  if not per_step:
    return None
  floor = max(per_step) // 2
  return {normalise_kernel(r['Name']) for r in rows if int(r['Calls']) >= floor}


def kernel_stats_classes(stats_csv):
  """rocprofv3 `*_kernel_stats.csv` -> ({class: {kernel, calls, avg_us, launches_per_step}}, steps traced).  A class may
  run as several instantiations (with / without the weight prefetch): call-weighted means."""
  import csv
  step_kernels = step_kernel_names(stats_csv)
  out, steps = {}, 0
  with open(stats_csv) as f:
    for r in csv.DictReader(f):
      if 'sampler_step_kernel' in r['Name']:
        steps = max(steps, int(r['Calls']))
      cls = classify_kernel(r['Name'], step_kernels)
      if not cls:
        continue
      e = out.setdefault(cls, {'kernel': [], 'calls': 0, '_ns': 0.0})
      e['kernel'].append(normalise_kernel(r['Name']))
      e['calls'] += int(r['Calls'])
      e['_ns'] += float(r['TotalDurationNs'])
  for e in out.values():
    e['avg_us'] = round(e.pop('_ns') / e['calls'] / 1e3, 3)
    e['kernel'] = ' | '.join(e['kernel'])
    e['launches_per_step'] = round(e['calls'] / steps, 3) if steps else None
  return out, steps


def kernel_type(name):
  """Coarse type of a kernel from its demangled name (what the name alone can tell)."""
  for sub, typ in (('sampler_step_kernel', 'sampler'), ('final_proj_f32_kernel', 'final'), ('EpiInProj', 'in_proj'),
                   ('EpiQKV', 'qkv'), ('EpiGeglu', 'mlp_in'), ('EpiStoreH16', 'cross_q'),
                   ('attention_merge_kernel', 'merge'), ('attention_kernel', 'attn'), ('EpiResidualNorm', 'resid')):
    if sub in name:
      return typ
  return None


def kernel_trace_classes(trace_csv):
  """rocprofv3 `*_kernel_trace.csv` (one row per dispatch, with timestamps) -> per-class figures by POSITION in the
  DDPM step.  A step is the run of dispatches from an input projection to the next sampler launch; inside it the class
  of a launch follows from its type and its predecessor (the first attention after a QKV projection is the
  self-attention, a residual GEMM after it the attention output projection, after a cross-attention / its merge the
  cross output projection, after the gated MLP input the MLP output) -- so instantiations that serve several launch
  sites (`small`: one residual template for three of them, one attention kernel for both attentions) still classify,
  which the name-based stats table cannot do.  Also the GAPS: start of a launch minus end of its predecessor.
  Returns ({class: {kernel, calls, avg_us, launches_per_step, avg_gap_before_us}}, steps, whole-step dict)."""
  import csv
  rows = []
  with open(trace_csv) as f:
    for r in csv.DictReader(f):
      name = r.get('Kernel_Name', '')
      if 'msd::' not in name:
        continue
      rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name))
  rows.sort()
  out, steps = {}, 0
  step_span = step_kernel = step_gap = inter_step = 0.0
  i, n = 0, len(rows)
  prev_step_end = None
  while i < n:
    if kernel_type(rows[i][2]) != 'in_proj':
      i += 1
      continue
    j = i
    while j < n and kernel_type(rows[j][2]) != 'sampler':
      j += 1
      if j < n and kernel_type(rows[j][2]) == 'in_proj':   # (a decoder pass without a sampler: msd_decoder_pass) not a step
        break
    if j >= n or kernel_type(rows[j][2]) != 'sampler':
      i = j
      continue
    seq = rows[i:j + 1]
    dedup = any('EpiResidualNorm<2, true' in r[2] for r in seq)
    prev_cls, first_qkv = None, True
    for k, (t0, t1, name) in enumerate(seq):
      typ = kernel_type(name)
      if typ == 'qkv':
        cls = 'gemm_qkv_l0' if (dedup and first_qkv) else 'gemm_qkv'
        first_qkv = False
      elif typ == 'attn':
        cls = 'attn_self' if prev_cls in ('gemm_qkv', 'gemm_qkv_l0') else 'attn_cross'
      elif typ == 'resid':
        if prev_cls == 'attn_self':
          cls = 'gemm_attn_out_l0' if 'EpiResidualNorm<2, true' in name else 'gemm_attn_out'
        elif prev_cls == 'gemm_mlp_in_geglu':
          cls = 'gemm_mlp_out'
        else:
          cls = 'gemm_cross_out'
      else:
        cls = {'sampler': 'sampler_step', 'final': 'final_proj_f32', 'in_proj': 'in_proj_f32', 'mlp_in': 'gemm_mlp_in_geglu',
               'cross_q': 'gemm_cross_q', 'merge': 'attn_cross_merge'}.get(typ)
      if cls is None:
        continue
      e = out.setdefault(cls, {'kernel': set(), 'calls': 0, '_ns': 0.0, '_gap': 0.0})
      e['kernel'].add(normalise_kernel(name))
      e['calls'] += 1
      e['_ns'] += t1 - t0
      if k > 0:
        e['_gap'] += t0 - seq[k - 1][1]
        step_gap += t0 - seq[k - 1][1]
      step_kernel += t1 - t0
      prev_cls = cls
    steps += 1
    step_span += seq[-1][1] - seq[0][0]
    if prev_step_end is not None and seq[0][0] - prev_step_end < 50000:   # consecutive steps of one segment
      inter_step += seq[0][0] - prev_step_end
    prev_step_end = seq[-1][1]
    i = j + 1
  for e in out.values():
    e['avg_us'] = round(e['_ns'] / e['calls'] / 1e3, 3)
    e['avg_gap_before_us'] = round(e.pop('_gap') / e['calls'] / 1e3, 3)
    e.pop('_ns')
    e['kernel'] = ' | '.join(sorted(e['kernel']))
    e['launches_per_step'] = round(e['calls'] / steps, 3) if steps else None
  whole = {}
  if steps:
    whole = {'sum_kernel_us_per_step': round(step_kernel / steps / 1e3, 2),
             'sum_gap_us_per_step': round(step_gap / steps / 1e3, 2),
             'step_span_us': round(step_span / steps / 1e3, 2),
             'gap_between_steps_us': round(inter_step / max(steps - 1, 1) / 1e3, 3)}
  return out, steps, whole


def rocprofv3_path():
  import shutil
  for c in (shutil.which('rocprofv3'), '/opt/rocm/bin/rocprofv3'):
    if c and os.path.exists(c):
      return c
  return None


def self_profile(args, preset, timeout_s=300.0, keep_csv=None):
  """The driver's line must carry THIS run's kernel times (it pasted the committed profile's).  An
  untimed leg: a child of this same script -- one 1000-step segment of `preset`, nothing else -- under
  `rocprofv3 --kernel-trace --stats`; its kernel_stats table -> per-class average durations of the graph-replayed
  kernels on THIS box, THIS binary.  Returns (record, None) or (None, why not)."""
  import glob
  import shutil
  import subprocess
  import tempfile
  exe = rocprofv3_path()
  if exe is None:
    return None, 'rocprofv3 not found on this box'
  tmp = tempfile.mkdtemp(prefix='msd_selfprof_', dir='/tmp')
  n_timed = min(args.steps, 2)   # the parent's warm-up segment(s) + its first timed ones: the same key counts, bounded trace size
  cmd = [exe, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', tmp, '--', sys.executable,
         os.path.abspath(__file__), '--self-profile-child', '--preset', preset, '--precision', args.precision,
         '--num-steps', str(args.num_steps), '--cfg-weight', str(args.cfg_weight), '--warmup', str(args.warmup),
         '--steps', str(n_timed), '--data', args.data]
  if args.attn_planes:
    cmd += ['--attn-planes', args.attn_planes]
  for item in (args.knob or []):
    cmd += ['--knob', item]
  env = dict(os.environ)
  env['TMPDIR'] = '/tmp'
  t0 = time.perf_counter()
  try:
    run = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
    child = None
    for line in reversed(run.stdout.strip().split('\n')):
      if line.startswith('{'):
        child = json.loads(line)
        break
    stats = sorted(glob.glob(os.path.join(tmp, '**', '*kernel_stats.csv'), recursive=True))
    trace = sorted(glob.glob(os.path.join(tmp, '**', '*kernel_trace.csv'), recursive=True))
    if run.returncode != 0 or not (stats or trace) or child is None:
      return None, 'rocprofv3 child failed (rc %s): %s' % (run.returncode, (run.stderr or run.stdout)[-300:])
    whole = {}
    if trace:     # per dispatch: classes by position in the step, and the gaps between launches
      classes, steps, whole = kernel_trace_classes(trace[0])
      how = 'per-dispatch kernel trace, classes by position in the step'
    else:         # (older rocprofv3 without the per-dispatch table) the stats table, classes by kernel name
      classes, steps = kernel_stats_classes(stats[0])
      how = 'kernel_stats table, classes by kernel name'
    if keep_csv and stats:
      shutil.copyfile(stats[0], keep_csv)
    if not classes or not steps:
      return None, 'no DDPM steps in the child trace'
    sum_us = whole.get('sum_kernel_us_per_step') or sum(e['avg_us'] * e['calls'] for e in classes.values()) / steps
    rec = {'command': 'rocprofv3 --kernel-trace --stats -- python bench.py --self-profile-child --preset %s --warmup %d --steps %d '
                      '(the segments this run warms up and starts timing on, nothing else; untimed leg of this run)'
                      % (preset, args.warmup, n_timed),
           'read_from': how, 'steps_traced': steps, 'seconds': round(time.perf_counter() - t0, 1),
           'child_sample_ms_per_segment': child.get('sample_ms_per_segment'),
           's_valid_keys': child.get('s_valid_keys'),
           'sum_kernel_us_per_step': round(sum_us, 2),
           'per_class': classes}
    rec.update({k: v for k, v in whole.items() if k != 'sum_kernel_us_per_step'})
    return rec, None
  except subprocess.TimeoutExpired:
    return None, 'rocprofv3 child did not finish within %.0f s' % timeout_s
  except Exception as e:   # never let the profile take the benchmark down
    return None, repr(e)[:300]
  finally:
    shutil.rmtree(tmp, ignore_errors=True)


def self_profile_child(args):
  """What `self_profile` traces: one model, the parent's warm-up segment(s) (restore + graph capture) and its first
  timed segments -- same token seeds, same context chaining, so the key counts (the cross-attention's time) are the
  parent's."""
  import torch
  import msd_amd
  spec = msd_amd.config.preset(args.preset, num_steps=args.num_steps, cfg_weight=args.cfg_weight)
  model = msd_amd.InferenceModel('synthetic:0', spec, batch_size=1, precision=args.precision, **model_kwargs(args))
  c_len = model.targets_context_length
  n_seg = args.warmup + args.steps
  if args.preset == 'small' and args.data == 'tokens':
    segs = [msd_amd.synthetic.segment_tokens(spec, 5000 + k) for k in range(n_seg)]       # small_leg's seeds
  elif args.data == 'midi':
    segs = synthetic_midi_tokens(spec, 0, n_seg)
  else:
    segs = [msd_amd.synthetic.segment_tokens(spec, k) for k in range(n_seg)]              # main()'s song_tokens(0, .)
  pred = torch.zeros((1, c_len, 128), dtype=torch.float32, device='cuda') if c_len is not None else None
  smp, keys = [], []
  for k in range(n_seg):
    batch = {'encoder_input_tokens': segs[k]}
    if c_len is not None:
      batch['encoder_continuous_inputs'] = pred
      batch['encoder_continuous_mask'] = (np.zeros if k == 0 else np.ones)((1, c_len), np.int32)
    out, _ = model.predict(batch, seed=0, segment=k, return_torch=True)
    if c_len is not None:
      pred = out
    if k >= args.warmup:
      smp.append(model.last_timing['sample_s'])
      keys.append(float((segs[k] > 0).sum() + (c_len if (c_len is not None and k > 0) else 0)))
  torch.cuda.synchronize()
  print(json.dumps({'sample_ms_per_segment': round(float(np.mean(smp)) * 1e3, 3), 's_valid_keys': float(np.mean(keys))}))


def cpu_baseline(spec, params, batch, sample_steps: int, budget_s: float = 20.0):
  """Oracle ('port') on the host cores: encoders once + `sample_steps` DDPM steps."""
  import torch
  from oracle import backend, fast
  from tests import helpers
  cores = backend.effective_cpus()
  xp = backend.TorchBackend('float32', threads=cores)
  cfg, dc = helpers.oracle_configs(spec)
  n_full = dc.sampler.schedule.num_steps
  fm = fast.FastModel(xp, cfg, dc, params, spec.has_context)
  t0 = time.perf_counter()
  if spec.has_context:
    fm.encode(batch['encoder_input_tokens'], batch['encoder_continuous_inputs'],
              batch['encoder_continuous_mask'])
  else:
    fm.encode(batch['encoder_input_tokens'])
  t_enc = time.perf_counter() - t0
  t, n = spec.task_feature_lengths['targets'], 128
  rng = np.random.default_rng(0)
  z = xp.asarray(rng.standard_normal((1, t, n)).astype(np.float32))
  w = dc.classifier_free_guidance.eval_condition_weight
  fm.decoder_pass(z, n_full - 1, True)  # warm the weight caches
  t0 = time.perf_counter()
  done = 0
  for k in range(sample_steps):
    i = n_full - 1 - k
    e = fm.decoder_pass(z, i, True)
    if w != 1:
      e = w * e + (1 - w) * fm.decoder_pass(z, i, False)
    z = z * 0.999 + 0.001 * e  # keep the data flowing; the sampler update is negligible
    done += 1
    if time.perf_counter() - t0 > budget_s:  # bounded sample: stop after ~budget_s of CPU work
      break
  sample_steps = done
  t_step = (time.perf_counter() - t0) / sample_steps
  seg_s = t_enc + n_full * t_step
  return {
      'value': t / seg_s, 'unit': 'mel-frames/sec', 'cores': cores, 'kind': 'port', 'reference_probe': probe_reference(),
      'config1': cpu_config1(cores, w),
      'as_written': cpu_as_written(spec, params, batch, cores, t_step),
      'xRTF': (t * 320 / 16000.0) / seg_s,
      'sample': 'torch-CPU float32 oracle (oracle/fast.py, cached cross K/V): encoders %.2fs + %d of %d '
                'DDPM steps at %.3fs/step, extrapolated linearly to one %d-frame segment'
                % (t_enc, sample_steps, n_full, t_step, t),
  }


def cpu_as_written(spec, params, batch, cores, fast_step_s, steps=3):
  """The reference path AS WRITTEN on the host cores, beside the shortcut model the main figure times: oracle/predict.py
  restates predict_batch_with_aux line by line -- both encoders, then per DDPM step two full decoder calls that
  RE-PROJECT the cross-attention K / V of all 2304 positions and evaluate the time-embedding MLP and all FiLM layers
  (models/diffusion/network.py:196-235, 377-392) -- which oracle/fast.py (and the device) hoist out of the loop.  One
  untimed 1-step run (thread pool, allocator), then a 1-step and a `steps`-step run: their difference over steps - 1 is
  what one more step costs (the per-step cost does not depend on the step count); extrapolated like the main figure."""
  import msd_amd
  from oracle import backend, predict
  from tests import helpers
  n_full = spec.diffusion.sampler.schedule.num_steps
  name = 'base_with_context' if spec.has_context else 'small'
  w = spec.diffusion.classifier_free_guidance.eval_condition_weight
  if msd_amd.config.preset(name, num_steps=steps, cfg_weight=w).t5 != spec.t5:
    return {'skipped': 'not a shipped preset'}
  xp = backend.TorchBackend('float32', threads=cores)

  def run(k):
    short = msd_amd.config.preset(name, num_steps=k, cfg_weight=w)
    cfg, dc = helpers.oracle_configs(short)
    init_z, noise = helpers.make_noise(short)
    t0 = time.perf_counter()
    predict.predict_batch_with_aux(xp, cfg, dc, params, batch, init_z, noise, context=spec.has_context)
    return time.perf_counter() - t0
  run(1)
  dt1, dtk = run(1), run(steps)
  if dtk <= dt1:   # (a noisy host: never report a non-positive step)
    return {'skipped': 'timing not monotone on this host (%d steps %.2f s <= 1 step %.2f s)' % (steps, dtk, dt1)}
  step_s = (dtk - dt1) / (steps - 1)
  seg_s = max(dt1 - step_s, 0.0) + n_full * step_s
  t = spec.task_feature_lengths['targets']
  return {'value': round(t / seg_s, 4), 'unit': 'mel-frames/sec', 'xRTF': round((t * 320 / 16000.0) / seg_s, 5), 'cores': cores,
          'kind': 'port', 'seconds_per_step': round(step_s, 4), 'ratio_to_shortcut_model_step': round(step_s / fast_step_s, 3),
          'sample': 'oracle/predict.py (the reference path as written: K / V re-projected, time MLP + FiLM evaluated in every '
                    'decoder call): a 1-step and a %d-step run after one warm-up run, their difference = %d steps, extrapolated to %d steps'
                    % (steps, steps - 1, n_full)}


def cpu_config1(cores, cfg_weight):
  """BASELINE.json configs[0], timed IN FULL (it is the reference's own CPU-runnable case: seconds): the `small`
  no-context model, ONE 256-frame segment, 10 DDPM steps, on the host cores -- the float32 oracle's predict (encoder,
  10 x {conditional + unconditional decoder pass, sampler update}, un-scaling), the JAX CPU path's stand-in."""
  import msd_amd
  from oracle import backend, fast
  from tests import helpers
  spec = msd_amd.config.preset('small', num_steps=10, cfg_weight=cfg_weight)
  params = msd_amd.synthetic.init_params(spec, 0)
  xp = backend.TorchBackend('float32', threads=cores)
  cfg, dc = helpers.oracle_configs(spec)
  batch = {'encoder_input_tokens': msd_amd.synthetic.segment_tokens(spec, 5000)}
  init_z, noise = helpers.make_noise(spec)
  fm = fast.FastModel(xp, cfg, dc, params, False)
  # (no warm-up pass: config 1 is ONE cold call, like the reference's first predict)
  t0 = time.perf_counter()
  out = fm.predict(batch, init_z, noise)[0]
  dt = time.perf_counter() - t0
  t = spec.task_feature_lengths['targets']
  return {'value': round(t / dt, 3), 'unit': 'mel-frames/sec', 'xRTF': round((t * 320 / 16000.0) / dt, 4),
          'seconds': round(dt, 3), 'cores': cores, 'kind': 'port', 'finite': bool(np.isfinite(xp.to_numpy(out)).all()),
          'sample': 'BASELINE config 1 in full: small (no context), 1 segment of %d frames, 10 DDPM steps, CFG w=%g, '
                    'torch-CPU float32 oracle (oracle/fast.py)' % (t, cfg_weight)}


def probe_reference():
  """time the real JAX/T5X reference on the host cores if its stack is importable on
  this box; otherwise the oracle port stands in (kind "port").  The probe result is reported either way.  When `jax`
  alone imports, the N5 pin runs (tools/pin/pin_jax_random.py: this package's restated threefry + normal against
  jax.random, bit for bit) and its verdict rides in the record."""
  missing = []
  for mod in ('jax', 'flax', 't5x', 'gin', 'seqio'):
    try:
      __import__(mod)
    except Exception:
      missing.append(mod)
  pin = None
  if 'jax' not in missing:
    try:
      import importlib.util
      spec = importlib.util.spec_from_file_location('pin_jax_random', os.path.join(ROOT, 'tools', 'pin', 'pin_jax_random.py'))
      mod = importlib.util.module_from_spec(spec)
      spec.loader.exec_module(mod)
      res = mod.compare(mod.JaxSide(), [0, 42], 4, [1, 256, 128])
      pin = 'N5 pin (jax_random vs jax.random, 2 seeds x (init_z + 4 steps)): %s' % ('BIT-IDENTICAL' if res['ok'] else 'DIFFERS: %s' % json.dumps(res['cases'][0])[:300])
    except Exception as e:
      pin = 'N5 pin failed to run: %s' % repr(e)[:200]
  if missing:
    msg = 'reference stack not importable here (missing: %s) -> oracle port timed instead' % ', '.join(missing)
  else:
    msg = 'jax/flax/t5x importable, but the reference sources are not on this box (no /root/reference at run time)'
  return msg if pin is None else msg + '; ' + pin


def synthetic_midi_tokens(spec, seed, n_segments):
  """Seeded synthetic song (4 pitched programs + drums, ~9 notes/s) -> MIDI bytes -> frontend tokens."""
  import msd_amd
  from msd_amd.frontend import midi_io, note_sequences, tokenizer
  rng = np.random.default_rng(seed)
  seconds = n_segments * spec.task_feature_lengths['targets'] * 320 / 16000.0
  ns = note_sequences.NoteSequence()
  for program in (0, 33, 48, 73):
    t = 0.0
    while t < seconds - 0.3:
      dur = float(rng.choice([0.12, 0.25, 0.5, 1.0]))
      for pitch in rng.choice(np.arange(36, 96), size=int(rng.integers(1, 4)), replace=False):
        ns.add_note(pitch=int(pitch), velocity=int(rng.integers(40, 120)), start_time=round(t, 2),
                    end_time=round(min(t + dur, seconds - 0.01), 2), program=program)
      t += dur * float(rng.choice([0.5, 1.0, 1.0, 2.0]))
  for k in range(int(seconds * 4)):
    ns.add_note(pitch=int(rng.choice([36, 38, 42, 46])), velocity=100, start_time=k * 0.25, end_time=k * 0.25 + 0.05,
                is_drum=True)
  ns = note_sequences.trim_overlapping_notes(ns)
  ns.total_time = seconds - 0.005
  cfg = tokenizer.FrontendConfig.from_spec(spec)
  toks = tokenizer.note_sequence_to_model_inputs(midi_io.parse_midi(midi_io.note_sequence_to_midi(ns, 500)), cfg,
                                                 num_samples=int(seconds * 16000) - 1, on_too_long='truncate')
  assert len(toks) == n_segments, (len(toks), n_segments)
  return toks


def batched_leg(spec, args):
  """Throughput lever outside the headline: several independent songs per GPU through the same
  kernels (M = 2 * songs * 256 rows; 128-row GEMM tiles from 4 songs up).  One warm-up segment,
  one timed segment per song."""
  import torch
  import msd_amd
  nb = args.batched_songs
  model = msd_amd.InferenceModel('synthetic:0', spec, batch_size=nb, precision=args.precision, **model_kwargs(args))
  c_len = model.targets_context_length
  pred = torch.zeros((nb, c_len, 128), dtype=torch.float32, device=model.device) if c_len is not None else None
  t_frames = spec.task_feature_lengths['targets']
  dt = 0.0
  for k in range(2):
    batch = {'encoder_input_tokens': np.concatenate(
        [msd_amd.synthetic.segment_tokens(spec, 1000 * (7 + b) + k) for b in range(nb)], 0)}
    if c_len is not None:
      batch['encoder_continuous_inputs'] = pred
      batch['encoder_continuous_mask'] = (np.zeros if k == 0 else np.ones)((nb, c_len), np.int32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, _ = model.predict(batch, seed=0, segment=k, return_torch=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if c_len is not None:
      pred = out
  return {'songs_per_gpu': nb, 'value': round(nb * t_frames / dt, 3), 'unit': 'mel-frames/sec',
          'ms_per_segment_batch': round(dt * 1e3, 2),
          'note': 'same kernels, %d independent songs batched per handle; not the headline workload (SURVEY 8: B=1)' % nb}


def small_leg(args):
  """BASELINE.json configs[1] beside the headline: `small` (no context), 1000-step DDPM, one GPU, a synthetic
  60 s song = 12 independent 256-frame segments one after the other.  One warm-up segment
  (restore + graph capture, excluded like the reference's first segment), then the 12 timed ones."""
  import torch
  import msd_amd
  spec = msd_amd.config.preset('small', num_steps=args.num_steps, cfg_weight=args.cfg_weight)
  model = msd_amd.InferenceModel('synthetic:0', spec, batch_size=1, precision=args.precision, **model_kwargs(args))
  t_frames = spec.task_feature_lengths['targets']
  n = args.small_segments
  segs = [msd_amd.synthetic.segment_tokens(spec, 5000 + k) for k in range(n + 1)]
  model.predict({'encoder_input_tokens': segs[0]}, seed=0, segment=0, return_torch=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  smp = 0.0
  for k in range(1, n + 1):
    out, _ = model.predict({'encoder_input_tokens': segs[k]}, seed=0, segment=k, return_torch=True)
    smp += model.last_timing['sample_s']
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  assert torch.isfinite(out).all(), 'non-finite mel output (small)'
  return {'value': round(n * t_frames / dt, 3), 'unit': 'mel-frames/sec', 'xRTF': round(n * t_frames * 320 / 16000.0 / dt, 4),
          'ms_per_step': round(dt / n * 1e3, 3), 'segments': n,
          'ms_per_ddpm_step': round(smp / n / args.num_steps * 1e3, 5),
          'config': {'workload': 'small (no context), %d-step DDPM, CFG w=%g, 1 song per GPU, %d independent segments of '
                                 '%d frames (60 s of audio)' % (args.num_steps, args.cfg_weight, n, t_frames),
                     'precision': args.precision}}


def small_roofline(args, leg):
  """`roofline` of the `small` leg: its own rocprofv3 child (self_profile) -> per-class us, dominant class, whole step."""
  import msd_amd
  spec = msd_amd.config.preset('small', num_steps=args.num_steps, cfg_weight=args.cfg_weight)
  keep = os.path.join(args.self_profile_keep, 'self_profile_small_kernel_stats.csv') if args.self_profile_keep else None
  sp, why = self_profile(args, 'small', keep_csv=keep)
  if sp is None:
    return {'ok': False, 'why': why}
  passes = 2 if args.cfg_weight != 1.0 else 1
  dom, table, whole, flops, abytes = roofline_from_self_profile(spec, sp, passes, 2 if args.precision.endswith('x3') else 1,
                                                                 leg.get('ms_per_ddpm_step'))
  ach = table[dom]['tflops']
  return {'bound': 'mfma', 'kernel': dom, 'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
          'frac': round(ach / PEAK_BF16_TFLOPS, 5), 'traffic': None,
          'duration_source': 'rocprofv3 --kernel-trace --stats of THIS run (self_profile child, preset small)',
          'kernel_ms_per_launch': round(table[dom]['us_per_launch'] * 1e-3, 5),
          'algorithmic_gflop_per_launch': round(flops[dom] / 1e9, 4), 'algorithmic_bytes_per_launch': int(abytes.get(dom, 0)),
          'self_profile': {k: v for k, v in sp.items() if k != 'per_class'}, 'per_class_us': table, 'whole_step': whole}


def _free_port():
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def self_launch(args, argv):
  """`python bench.py --gpus N` (N > 1) started WITHOUT torch.distributed.run: start the N ranks ourselves --
  the same command the driver's own launcher would run -- and hand back its exit code.  Rank 0 of the children
  prints the one JSON line on the stdout they inherit."""
  import subprocess
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL between processes needs it on this driver
  env.setdefault('OMP_NUM_THREADS', '4')
  return subprocess.run(cmd, env=env).returncode


def _resolve(name):
  """'package.module:attr' -> the object (the CPU test leg swaps the model class this way)."""
  import importlib
  mod, attr = name.split(':')
  return getattr(importlib.import_module(mod), attr)


def _sync(device=None):
  """torch.cuda.synchronize() when the run is on a GPU (the CPU / gloo test leg has nothing to wait for)."""
  import torch
  if device is not None and str(device) == 'cpu':
    return
  if torch.cuda.is_available():
    torch.cuda.synchronize()


def handoff_check(dist, rank, world, device, shape=(1, 256, 128), rounds=20):
  """Untimed evidence that the hand-off primitive of --mode chained/wavefront works on this node: the
  128 KiB context message goes rank r -> r+1 (device to device, dist.send/recv = RCCL point-to-point
  over one xGMI link), `rounds` times down the chain; content checked, mean latency per hop reported."""
  import torch
  from msd_amd import sharding
  try:
    sharding.warm_up(None, device)   # communicators / peer connections exist before anything here is timed
    buf = torch.empty(sharding.HEADER + int(np.prod(shape)), dtype=torch.float32, device=device)
    _sync(device)
    dist.barrier()
    ok = True
    outbox = sharding._Outbox()
    t0 = time.perf_counter()
    for k in range(rounds + 1):   # the message path of sharding.chained_predict: header-checked, asynchronous send
      if k == 1:   # round 0 opens the connections
        _sync(device)
        t0 = time.perf_counter()
      if rank > 0:
        sharding._recv(buf, rank - 1)
        _sync(device)
        got = sharding.unpack_handoff(buf, k, rank - 1, shape)   # raises HandoffError on a reordered message
        ok = ok and bool((got == float(rank - 1) + k).all())
      if rank + 1 < world:
        outbox.post(sharding.pack_handoff(torch.full(shape, float(rank) + k, dtype=torch.float32, device=device), k, rank), rank + 1)
    outbox.drain()
    _sync(device)
    dt = time.perf_counter() - t0
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return {'ok': bool(flag.item() == 1.0), 'message_bytes': int(np.prod(shape)) * 4, 'hops': world - 1,
            'warm_up': sharding.last_warm_up,   # seconds in the group's first collective / first point-to-point chain, both untimed
            'us_per_hop': round(float(tt.item()) / rounds / max(world - 1, 1) * 1e6, 1),
            'note': 'header-checked message of sharding.chained_predict: isend / recv of the device tensor (RCCL point-to-point, one communicator per peer pair); one message per chunk boundary in --mode chained'}
  except Exception as e:  # never let the probe take the benchmark down
    return {'ok': False, 'error': repr(e)[:200]}


class Watchdog:
  """Bounded wait around a region that may hang in a communication call (first contact with RCCL point-to-point on a
  node nobody has run on).  A hung NCCL / HIP wait cannot be interrupted from Python, so when the time is up the
  timer thread runs `on_timeout` (rank 0: print the benchmark line it already has, with an error field) and ends the
  PROCESS with os._exit(EXIT_CODE != 0) -- every rank arms its own, so the launcher sees all ranks leave, and the
  launcher's (= the plain command's) exit code is non-zero: a hung hand-off must not look like a clean run
  (VERDICT r05 weak #9)."""

  def __init__(self, seconds, on_timeout):
    import threading
    self.t = threading.Timer(seconds, self._fire)
    self.t.daemon = True
    self.on_timeout = on_timeout
    self.seconds = seconds

  EXIT_CODE = 3   # a hang is a FAILURE of the run: the line is still printed (with its error field), the exit code says so

  def _fire(self):
    # Only RANK 0 leaves with the failure code, and only after it has printed: torch.distributed.run answers the first
    # failed worker by sending SIGTERM to all the others -- a rank > 0 that timed out a moment earlier and left non-zero
    # took rank 0 down before it could print (the line was lost in one run of three).  The launcher's exit code is
    # non-zero as soon as one worker's is.
    try:
      self.on_timeout(self.seconds)
    finally:
      sys.stdout.flush()
      os._exit(self.EXIT_CODE if int(os.environ.get('RANK', '0')) == 0 else 0)

  def __enter__(self):
    self.t.start()
    return self

  def __exit__(self, *exc):
    self.t.cancel()
    return False


def comm_device_of(args, model):
  """Where the hand-off message lives: device memory under RCCL ('nccl'); the host under gloo (its point-to-point calls
  take CPU tensors only) -- also when the MODEL is on a GPU (the -m gpu test of this launcher: two ranks, one device)."""
  return model.device if args.dist_backend == 'nccl' else 'cpu'


def handoff_leg(args, model, spec, dist, rank, world, song_tokens, ctx_shape):
  """BASELINE config 4 as a leg of the plain N > 1 line: the 10-minute workload (--handoff-segments, 118) as a
  wavefront of K = ceil(118 / N) songs x N segments (beam/evaluation.py:191-223 per song: segment k+1 conditions on
  segment k's prediction, which arrives from rank k as one device-to-device message); K is capped so that a rank
  runs at most --handoff-max-segments segments.  Returns the record on every rank (rank 0 prints it)."""
  import torch
  from msd_amd import sharding
  k_songs = min(-(-args.handoff_segments // world), args.handoff_max_segments)
  songs = [[t[:1] for t in song_tokens(100 + j, world)] for j in range(k_songs)]
  busy = [0.0]

  def timed_predict_sequence(*a, **kw):
    _sync(model.device)
    t0 = time.perf_counter()
    out = model.predict_sequence(*a, **kw)
    _sync(model.device)
    busy[0] += time.perf_counter() - t0
    return out

  _sync(model.device)
  dist.barrier()
  _sync(model.device)
  t0 = time.perf_counter()
  outs = sharding.chained_wavefront(timed_predict_sequence, songs, ctx_shape, rank, world,
                                    comm_device=comm_device_of(args, model), seed=100, return_torch=True)   # (song j: noise seed 100 + j)
  _sync(model.device)
  mine = time.perf_counter() - t0
  dist.barrier()
  _sync(model.device)
  elapsed = time.perf_counter() - t0
  finite = all(bool(torch.isfinite(torch.as_tensor(o)).all()) for o in outs)
  import hashlib
  # what this rank synthesized, song by song, as a digest of the float32 bytes: a test (or a SCALE reader) can compare
  # it with the sequential song's segment `rank` -- the hand-off is bit-identical by construction
  digests = [hashlib.sha256(np.ascontiguousarray(torch.as_tensor(o).detach().cpu().numpy(), np.float32).tobytes()).hexdigest()[:16]
             for o in outs]
  rows = [None] * world
  dist.all_gather_object(rows, {'rank': rank, 'seconds': round(mine, 4), 'busy_seconds': round(busy[0], 4),
                                'idle_fraction': round(1.0 - busy[0] / max(elapsed, 1e-9), 4), 'finite': finite,
                                'segment_sha256_16': digests})
  tmax = torch.tensor([elapsed], dtype=torch.float64, device=comm_device_of(args, model))
  dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
  elapsed = float(tmax.item())
  t_frames = spec.task_feature_lengths['targets']
  frames = k_songs * world * t_frames
  return {'mode': 'wavefront', 'songs': k_songs, 'segments_per_song': world, 'segments': k_songs * world,
          'value': round(frames / elapsed, 3), 'unit': 'mel-frames/sec', 'xRTF': round(frames * 320 / 16000.0 / elapsed, 4),
          'seconds': round(elapsed, 4), 'ideal_efficiency': round(k_songs / (k_songs + world - 1.0), 4),
          'per_rank': rows, 'ok': all(r['finite'] for r in rows),
          'data': args.data,
          'note': 'BASELINE config 4: %d-segment workload as %d songs x %d segments, rank r runs segment r of every song; '
                  'context hand-off per segment boundary (sharding.chained_wavefront); bit-identical to the sequential songs'
                  % (args.handoff_segments, k_songs, world)}


def model_kwargs(args):
  """InferenceModel options that are NOT the library default (the default run passes none)."""
  kw = {}
  if args.attn_planes:
    q, p = (int(v) for v in args.attn_planes.split(','))
    kw['attention_query_planes'] = (q, p)
  for item in (args.knob or []):   # e.g. --knob cross_q_fold=False: an InferenceModel keyword (launch-level A/B on one box)
    import ast
    k, v = item.split('=', 1)
    kw[k.strip()] = ast.literal_eval(v.strip())
  return kw


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=5, help='timed segments (per GPU)')
  ap.add_argument('--warmup', type=int, default=1, help='untimed warm-up segments')
  ap.add_argument('--mode', choices=['replicas', 'chained', 'wavefront', 'masked'], default='replicas',
                  help='multi-GPU partitioning (module docstring); all modes equal replicas at --gpus 1')
  ap.add_argument('--preset', default='base_with_context')
  ap.add_argument('--precision', default='f16x3', choices=['f16x3', 'f16', 'bf16x3', 'bf16'],
                  help="'f16x3' = hi + lo IEEE-half planes, 3 MFMAs per product (parity mode, libmsd_amd.so); 'bf16x3' = the "
                       "same with bfloat16 planes (libmsd_amd_bf16.so); 'f16' / 'bf16' = one plane")
  ap.add_argument('--num-steps', type=int, default=1000, help='DDPM steps (headline: 1000)')
  ap.add_argument('--cfg-weight', type=float, default=5.0)
  ap.add_argument('--batch', type=int, default=1, help='independent songs synthesized together per GPU')
  ap.add_argument('--knob', action='append', default=None, help='k=v: an InferenceModel keyword that is not the library '
                  'default (e.g. cross_q_fold=False), for launch-level A/Bs; named in config.knobs')
  ap.add_argument('--attn-planes', default='', help='q,p planes of the query side of the decoder attentions (e.g. "1,1" = '
                  "round 3's single plane); default: the library's choice (hi + lo for both: DESIGN.md 3)")
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-sample-steps', type=int, default=20)
  ap.add_argument('--data', choices=['tokens', 'midi'], default='tokens',
                  help="'tokens': BASELINE.md 4 synthetic token streams (headline); 'midi': synthetic MIDI songs through frontend/")
  ap.add_argument('--profile-steps', type=int, default=3)
  ap.add_argument('--batched-songs', type=int, default=8,
                  help='extra leg: this many songs per GPU in one handle (0/1 = skip); N=1 runs only')
  ap.add_argument('--small-segments', type=int, default=12,
                  help="extra leg: BASELINE config 2, the `small` no-context model over this many segments (0 = skip); N=1 runs only")
  ap.add_argument('--handoff-segments', type=int, default=118,
                  help='N > 1, --mode replicas: size of the extra `handoff` leg (BASELINE config 4: 10 min of MIDI = 118 segments); 0 = skip')
  ap.add_argument('--handoff-max-segments', type=int, default=16,
                  help='cap on the segments one rank runs in the `handoff` leg (its wall time is about this many + N - 1 segments)')
  ap.add_argument('--handoff-timeout', type=float, default=240.0,
                  help='seconds after which a hanging hand-off probe / leg is abandoned (the line then carries an error field)')
  ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                  help="torch.distributed backend: 'nccl' (= RCCL over xGMI, the product); 'gloo' is for the CPU test of the launcher")
  ap.add_argument('--no-self-profile', action='store_true',
                  help='skip the untimed rocprofv3 legs (a child of this script under --kernel-trace --stats per preset)')
  ap.add_argument('--self-profile-child', action='store_true', help=argparse.SUPPRESS)   # what self_profile() traces
  ap.add_argument('--self-profile-keep', default='', help='directory that keeps the children\'s kernel_stats tables (profiles/ records)')
  ap.add_argument('--model-factory', default='msd_amd:InferenceModel',
                  help='module:attr of the InferenceModel class (tests swap in a CPU stand-in to exercise the multi-rank plumbing)')
  args = ap.parse_args()
  if args.self_profile_child:
    return self_profile_child(args)

  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    # started plainly (the driver's `python bench.py --gpus N`): become the launcher of N ranks
    raise SystemExit(self_launch(args, sys.argv[1:]))

  import torch
  import msd_amd
  from msd_amd import sharding

  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or plainly, without '
                     'torch.distributed.run)' % (args.gpus, world, args.gpus))
  on_gpu = args.dist_backend == 'nccl'
  if on_gpu:
    torch.cuda.set_device(local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    import datetime
    pg_timeout = datetime.timedelta(seconds=max(60.0, 2 * args.handoff_timeout))   # (a stuck collective aborts instead of hanging for 10 min)
    if on_gpu:
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank), timeout=pg_timeout)
    else:
      dist.init_process_group('gloo', rank=rank, world_size=world, timeout=pg_timeout)

  ranks_seen = None
  if dist is not None:
    # census: what each rank runs on, all-gathered -- a SCALE record can then prove that N distinct devices took part
    me = {'rank': rank, 'local_rank': local_rank, 'pid': os.getpid(), 'host': os.uname().nodename}
    if on_gpu:
      prop = torch.cuda.get_device_properties(local_rank)
      me.update(device=torch.cuda.current_device(), name=prop.name,
                uuid=str(getattr(prop, 'uuid', '')), pci_bus_id=getattr(prop, 'pci_bus_id', None))
    ranks_seen = [None] * world
    dist.all_gather_object(ranks_seen, me)

  spec = msd_amd.config.preset(args.preset, num_steps=args.num_steps, cfg_weight=args.cfg_weight)
  model = _resolve(args.model_factory)('synthetic:0', spec, batch_size=args.batch, precision=args.precision, **model_kwargs(args))
  nb = args.batch
  t_frames = spec.task_feature_lengths['targets']
  c_len = model.targets_context_length
  mode = args.mode if (world > 1 and c_len is not None) else 'replicas'
  if mode != 'replicas' and nb != 1:
    raise SystemExit('--mode %s runs one song per rank at a time (--batch 1)' % mode)
  n_seg = args.warmup + args.steps

  def song_tokens(song, n):   # `nb` songs side by side -> list of [nb, L] segment inputs
    if args.data == 'midi':
      songs = [synthetic_midi_tokens(spec, 1000 * (song * nb + b), n) for b in range(nb)]
      return [np.concatenate([songs[b][k] for b in range(nb)], 0) for k in range(n)]
    return [np.concatenate([msd_amd.synthetic.segment_tokens(spec, 1000 * (song * nb + b) + k) for b in range(nb)], 0)
            for k in range(n)]

  # every rank warms up on its own song (restore, tables, graph capture: excluded like the reference
  # excludes its first segment, beam/evaluation.py:217-220)
  segs = song_tokens(rank, n_seg)
  pred = None
  if c_len is not None:
    pred = torch.zeros((nb, c_len, 128), dtype=torch.float32, device=model.device)

  def run_segment(k):
    nonlocal pred
    batch = {'encoder_input_tokens': segs[k]}
    if c_len is not None:
      batch['encoder_continuous_inputs'] = pred
      batch['encoder_continuous_mask'] = (np.zeros if k == 0 else np.ones)((nb, c_len), np.int32)
    out, _ = model.predict(batch, seed=0, segment=k, return_torch=True)
    if c_len is not None:
      pred = out
    return out

  for k in range(args.warmup):
    run_segment(k)
  _sync(model.device)
  if dist is not None:
    dist.barrier()
  _sync(model.device)
  t0 = time.perf_counter()
  enc_s = smp_s = 0.0
  ctx_shape = (1, c_len or 0, 128)
  if mode == 'replicas':
    for k in range(args.warmup, n_seg):
      out = run_segment(k)
      enc_s += model.last_timing['encode_s']
      smp_s += model.last_timing['sample_s']
  elif mode == 'chained':      # one song, world * K segments, rank r owns [r K, (r+1) K)
    song = [t[:1] for t in song_tokens(0, world * args.steps)]
    out = sharding.chained_predict(model.predict_sequence, song, ctx_shape, rank, world,
                                   comm_device=comm_device_of(args, model), return_torch=True)
  elif mode == 'wavefront':    # K songs of `world` segments: rank r runs segment r of every song
    songs = [[t[:1] for t in song_tokens(j, world)] for j in range(args.steps)]
    out = sharding.chained_wavefront(model.predict_sequence, songs, ctx_shape, rank, world,
                                     comm_device=comm_device_of(args, model), return_torch=True)[-1]
  else:                        # masked: chunk heads context-masked, no message
    song = [t[:1] for t in song_tokens(0, world * args.steps)]
    a, b = sharding.contiguous_chunk(len(song), rank, world)
    out = model.predict_sequence(song[a:b], first_segment_index=a, return_torch=True)
  _sync(model.device)
  if dist is not None:
    dist.barrier()
  _sync(model.device)
  elapsed = time.perf_counter() - t0
  assert torch.isfinite(torch.as_tensor(out)).all(), 'non-finite mel output'
  per_rank_seconds = None
  if dist is not None:
    per_rank_seconds = [None] * world
    dist.all_gather_object(per_rank_seconds, round(elapsed, 6))
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=comm_device_of(args, model))
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
  # the precision each rank ENDED in: range_fallback (on by default) switches a model to bfloat16 planes with a warning
  # when an activation leaves the half range -- the line must not credit f16x3 with a figure measured in bf16x3
  precision_ran = getattr(model, 'precision', args.precision)
  precisions_seen = None
  if dist is not None:
    precisions_seen = [None] * world
    dist.all_gather_object(precisions_seen, precision_ran)

  result = None
  if rank == 0:
    frames = world * args.steps * t_frames * nb
    value = frames / elapsed
    audio_s = frames * 320 / 16000.0
    passes = 2 if args.cfg_weight != 1.0 else 1
    # ---- roofline of the dominant kernel (hipEvents on the launch stream) --------
    nm = model._get_native()
    toks = segs[-1]
    s_valid = float((toks > 0).sum() / nb + (c_len or 0))
    # make sure the profiled steps see this rank's last replica-style encode (chained modes end elsewhere)
    if mode != 'replicas':
      run_segment(n_seg - 1)
    if on_gpu:
      with torch.cuda.device(model.device):
        prof = nm.profile_steps(nb, args.profile_steps, stream=model._stream.cuda_stream)
    else:
      prof = nm.profile_steps(nb, args.profile_steps)
    flops = {k: v * nb for k, v in class_flops(spec, s_valid, passes).items()}
    abytes = {k: v * nb for k, v in class_bytes(spec, s_valid, passes, 2 if args.precision.endswith('x3') else 1).items()}
    per_class = {}
    for name, (ms, launches) in prof.items():
      if launches:
        per_class[name] = {'ms_per_launch': ms / launches, 'launches_per_step': launches / args.profile_steps,
                           'ms_per_step': ms / args.profile_steps}
    fold_cross_q_work(flops, abytes, per_class)
    # dominant kernel = the class that carries the most algorithmic FLOP per step (the gated MLP input projection:
    # 30 % of the step's FLOP and the largest share of its GPU time in the rocprofv3 trace).  NOT "the longest
    # eager launch": under hipEvents the first launch of a step (in-proj, cold) can outlast it by a microsecond,
    # which once put a 0.05 GFLOP kernel into this object.  Per-step totals by class: per_class_ms_per_step.
    dom = max((n for n in per_class if n in flops), key=lambda n: flops[n] * per_class[n]['launches_per_step'])
    event_ms = per_class[dom]['ms_per_launch']
    step_flops = sum(flops[n] * per_class[n]['launches_per_step'] for n in per_class if n in flops)
    prof_entry, prof_src = profile_roofline(dom, args)
    graph_step_ms = (smp_s / args.steps / args.num_steps * 1e3) if mode == 'replicas' else None
    # `achieved` = algorithmic FLOP per launch / average launch duration.  Duration, in this order: (1) THIS run's own
    # rocprofv3 kernel trace (self_profile: a child of this script on this box and binary, untimed) -- what the graph
    # replays, measured where the line is measured; (2) the committed profile's average when it was taken on this binary
    # (another box: labelled); (3) the live hipEvent figure (eager launches, ~20 % longer).  All are reported.
    sp, sp_why = (None, 'skipped (--no-self-profile)')
    if on_gpu and world == 1 and not args.no_self_profile and nb == 1:
      keep = os.path.join(args.self_profile_keep, 'self_profile_%s_kernel_stats.csv' % args.preset) if args.self_profile_keep else None
      sp, sp_why = self_profile(args, args.preset, keep_csv=keep)
    sp_table = sp_whole = None
    if sp is not None:
      planes_n = 2 if args.precision.endswith('x3') else 1
      sp_dom, sp_table, sp_whole, _, _ = roofline_from_self_profile(spec, sp, passes, planes_n, graph_step_ms)
      if sp_dom != dom or dom not in sp_table:
        sp, sp_why = None, 'the trace\'s dominant class (%s) is not the eager one (%s)' % (sp_dom, dom)
    use_prof = bool(prof_entry and prof_entry.get('matches_binary') and prof_entry.get('avg_us'))
    if sp is not None:
      dur_ms = sp_table[dom]['us_per_launch'] * 1e-3
      dur_src = 'rocprofv3 --kernel-trace --stats of THIS run (self_profile: a child of this command on this box, same binary)'
    elif use_prof:
      dur_ms = prof_entry['avg_us'] * 1e-3
      dur_src = 'rocprofv3 kernel-trace average of the COMMITTED profile (profiles/roofline.json: same binary, another box)'
    else:
      dur_ms = event_ms
      dur_src = 'hipEvents around eager launches (msd_profile_steps)'
    achieved = flops[dom] / (dur_ms * 1e-3) / 1e12
    roofline = {
        'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 3), 'peak': PEAK_BF16_TFLOPS,
        'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_BF16_TFLOPS, 5),
        'traffic': prof_entry.get('fabric_bytes_per_launch') if prof_entry else None,
        'traffic_source': 'PMC passes of the committed profile (profiles/roofline.json; counter collection serialises the '
                          'dispatches at ~50 ms each, so it cannot run inside this command)' if prof_entry else None,
        'duration_source': dur_src,
        'self_profile': ({k: v for k, v in sp.items() if k != 'per_class'} if sp is not None else {'ok': False, 'why': sp_why}),
        'per_class_us': sp_table,
        'whole_step_self_profile': sp_whole,
        'kernel_ms_per_launch': round(dur_ms, 5),
        'kernel_ms_per_launch_hipevents': round(event_ms, 5),
        'achieved_hipevents': round(flops[dom] / (event_ms * 1e-3) / 1e12, 3),
        'algorithmic_gflop_per_launch': round(flops[dom] / 1e9, 4),
        'algorithmic_bytes_per_launch': int(abytes.get(dom, 0)),
        'profile': prof_entry, 'profile_source': prof_src,
        'library_sha': library_hash('bf16' if args.precision.startswith('bf16') else 'f16'),
        'whole_step': {
            'algorithmic_gflop': round(step_flops / 1e9, 2),
            'eager_ms': round(sum(v['ms_per_step'] for v in per_class.values()), 4),
            'graph_ms': None if graph_step_ms is None else round(graph_step_ms, 4),
            'achieved_tflops_graph': None if graph_step_ms is None else round(step_flops / (graph_step_ms * 1e-3) / 1e12, 3),
            'launches': round(sum(v['launches_per_step'] for v in per_class.values()), 1),
        },
        'per_class_ms_per_step': {k: round(v['ms_per_step'], 4) for k, v in per_class.items()},
    }
    par = {'replicas': 'song-parallel x%d' % world,
           'chained': 'one song chained over %d GPUs (context hand-off, serial)' % world,
           'wavefront': 'wavefront of %d songs x %d segments over %d GPUs (context hand-off per segment)' % (args.steps, world, world),
           'masked': 'one song, %d masked-boundary chunks' % world}[mode]
    result = {
        'metric': 'mel-frames/sec', 'value': round(value, 3), 'unit': 'mel-frames/sec',
        'xRTF': round(audio_s / elapsed, 4),
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f16x3': 'f16x3 (operands split into hi + lo half planes, 3 f16 MFMAs per product, fp32 accumulate; '
                           'fp32 residual/norm/softmax/sampler)',
                  'bf16x3': 'bf16x3 (operands split into hi + lo bfloat16 planes, 3 bf16 MFMAs per product, fp32 '
                            'accumulate; fp32 residual/norm/softmax/sampler)'}.get(args.precision, args.precision),
        'data': ('synthetic (seeded tokens, reference-initialiser weights, Philox noise)' if args.data == 'tokens' else
                 'synthetic (seeded MIDI songs through the front end, reference-initialiser weights, Philox noise)'),
        'config': {'workload': '%s, %d-step DDPM, CFG w=%g, %d song(s) per GPU, %s, %d segments of %d frames per GPU'
                               % (args.preset, args.num_steps, args.cfg_weight, nb,
                                  'segment-sequential with context hand-off' if c_len is not None
                                  else 'independent segments (no context), one after the other',
                                  args.steps, t_frames),
                   'precision': precision_ran, 'precision_requested': args.precision, 'parallelism': par, 'mode': mode,
                   'attention_query_planes': args.attn_planes or 'library default: hi + lo for Q and for the softmax weights',
                   'knobs': args.knob or 'library defaults'},
        'roofline': roofline,
    }
    if mode == 'replicas':
      result['encode_ms_per_segment'] = round(enc_s / args.steps * 1e3, 3)
      result['sample_ms_per_segment'] = round(smp_s / args.steps * 1e3, 3)
    if ranks_seen is not None:
      result['ranks_seen'] = ranks_seen               # one entry per rank: device / uuid / pid as that rank saw them
      result['per_rank_seconds'] = per_rank_seconds   # each rank's own timed region; `value` uses the maximum
      result['per_rank_precision'] = precisions_seen  # what each rank's model ended in (range_fallback may switch planes)
      for r_, p_ in zip(result['ranks_seen'], precisions_seen):
        r_['precision'] = p_
    if world == 1 and args.batched_songs > 1 and nb == 1:
      result['batched'] = batched_leg(spec, args)
    if world == 1 and args.small_segments > 0 and nb == 1 and args.preset != 'small':
      result['small'] = small_leg(args)
      if on_gpu and not args.no_self_profile:   # BASELINE config 2's own per-class record
        result['small']['roofline'] = small_roofline(args, result['small'])
    if world == 1 and not args.no_cpu_baseline:
      batch = {'encoder_input_tokens': segs[-1][:1]}
      if c_len is not None:
        batch['encoder_continuous_inputs'] = np.zeros((1, c_len, 128), np.float32)
        batch['encoder_continuous_mask'] = np.ones((1, c_len), np.int32)
      result['cpu_baseline'] = cpu_baseline(spec, model.params, batch, args.cpu_sample_steps)
  if dist is not None and c_len:
    # ---- the message path (probe + BASELINE config 4 leg), AFTER the headline figures are safe: first contact with
    # RCCL point-to-point on this node may hang; the watchdog then prints the line as it stands with an error field
    printed = []

    def bail(seconds):
      if rank == 0 and not printed:
        result.setdefault('handoff_check', {'ok': False})
        result['handoff'] = {'ok': False, 'error': 'the hand-off probe / leg did not finish within %.0f s (--handoff-timeout): '
                                                   'abandoned, replicas figures above are unaffected' % seconds}
        print(json.dumps(result))
        printed.append(1)

    with Watchdog(args.handoff_timeout, bail):
      handoff = handoff_check(dist, rank, world, comm_device_of(args, model), (1, c_len, 128))
      if rank == 0:
        result['handoff_check'] = handoff
      if mode == 'replicas' and args.handoff_segments > 0 and nb == 1:
        try:
          leg = handoff_leg(args, model, spec, dist, rank, world, song_tokens, ctx_shape)
        except Exception as e:   # a HandoffError or a transport error on one rank: report, do not lose the line
          leg = {'ok': False, 'error': repr(e)[:300]}
        if rank == 0:
          result['handoff'] = leg
  if dist is not None:
    with Watchdog(60.0, lambda s_: print(json.dumps(result)) if rank == 0 else None):
      dist.barrier()
      dist.destroy_process_group()
  if rank == 0:
    print(json.dumps(result))


if __name__ == '__main__':
  main()
