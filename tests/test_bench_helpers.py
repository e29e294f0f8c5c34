"""bench.py pieces that need no GPU: the FLOP model behind `roofline.achieved`, the committed PMC traffic
lookup behind `roofline.traffic`, and the synthetic-MIDI workload generator (`--data midi`)."""
import argparse
import importlib.util
import os

import numpy as np

import msd_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)


def test_class_flops_match_design_table():
  spec = msd_amd.config.preset('base_with_context')
  f = bench.class_flops(spec, 2304.0, 2)
  assert f['gemm_mlp_in_geglu'] == 2.0 * 512 * 4096 * 768          # 3.22 GFLOP / launch
  assert f['gemm_qkv'] == 2.0 * 512 * 2304 * 768
  assert f['gemm_mlp_out'] == 2.0 * 512 * 768 * 2048
  assert f['gemm_cross_q'] == 2.0 * 256 * 768 * 768                # conditional rows only (S4)
  per_layer = sum(v for k, v in f.items() if k not in ('final_proj_f32', 'in_proj_f32'))
  total = 12 * per_layer + f['final_proj_f32'] + f['in_proj_f32']
  assert 100e9 < total < 130e9                                      # ~112-121 GFLOP per step


def test_class_bytes_match_the_hand_count():
  """MLP-in at base, CFG, bf16x3: 12.6 MB weight planes + 1.6 MB activations + 4.2 MB output."""
  spec = msd_amd.config.preset('base_with_context')
  b = bench.class_bytes(spec, 2304.0, 2)
  assert abs(b['gemm_mlp_in_geglu'] - (4 * 4096 * 768 + 4 * 512 * 768 + 4 * 512 * 2048)) < 0.1e6
  assert b['attn_cross'] > b['attn_self']
  one = bench.class_bytes(spec, 2304.0, 2, planes=1)
  assert one['gemm_qkv'] < b['gemm_qkv']


def test_profile_roofline_lookup_and_staleness_stamp():
  """profiles/roofline.json (tools/make_roofline.py): per-class rocprof duration, MFMA utilisation and fabric
  bytes, stamped with the library hash; only for the configuration it was measured on."""
  ok = argparse.Namespace(preset='base_with_context', batch=1, precision='f16x3', cfg_weight=5.0)
  e, src = bench.profile_roofline('gemm_mlp_in_geglu', ok)
  assert e is not None and 'rocprofv3' in src
  assert 5.0 < e['avg_us'] < 40.0 and 0.0 < e['mfma_util'] < 1.0 and e['waste'] >= 1.0
  assert 10e6 < e['fabric_bytes_per_launch'] < 100e6
  assert isinstance(e['matches_binary'], bool) and 'profile_library_sha' in e
  for other in (dict(preset='small'), dict(batch=8), dict(precision='f16'), dict(cfg_weight=1.0)):
    ns = argparse.Namespace(**{**vars(ok), **other})
    assert bench.profile_roofline('gemm_mlp_in_geglu', ns)[0] is None
  assert bench.profile_roofline('no_such_class', ok)[0] is None
  assert bench.library_hash() is None or len(bench.library_hash()) == 16


def test_synthetic_midi_workload():
  spec = msd_amd.config.preset('base_with_context')
  toks = bench.synthetic_midi_tokens(spec, 7, 3)
  assert len(toks) == 3
  for t in toks:
    assert t.shape == (1, 2048) and t.dtype == np.int32
    n = int((t > 0).sum())
    assert 100 < n < 2048 and t[0, n - 1] == 1 and t.max() < 1536
  again = bench.synthetic_midi_tokens(spec, 7, 3)
  assert all(np.array_equal(a, b) for a, b in zip(toks, again))


def test_roofline_json_covers_every_kernel_class_of_the_step():
  """profiles/roofline.json (tools/make_roofline.py) must hold every kernel class of the DDPM step with a duration,
  a FLOP-rate fraction and -- for the MFMA classes -- utilisation and traffic; the kernel names of the committed
  trace must all classify (a renamed kernel or template argument would silently drop a class)."""
  import csv
  import importlib.util
  import json
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location('make_roofline', os.path.join(root, 'tools', 'make_roofline.py'))
  mr = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mr)
  doc = json.load(open(os.path.join(root, 'profiles', 'roofline.json')))
  # (round 4: attention-out runs on 64 x 32 tiles and cross-out on 32 x 32, so each is a class of its own; in round 3's
  # traces they shared one template, 'gemm_attn_out+gemm_cross_out')
  want = {'gemm_mlp_in_geglu', 'gemm_mlp_out', 'gemm_qkv', 'gemm_attn_out', 'gemm_cross_out', 'gemm_cross_q', 'attn_self',
          'attn_cross', 'attn_cross_merge', 'final_proj_f32', 'in_proj_f32', 'sampler_step'}
  if 'gemm_attn_out+cross_q' in doc['per_class']:   # a profile taken with MSD_HOIST_Q=1: the q projection has no launch of its own
    want = (want - {'gemm_cross_q'}) | {'gemm_attn_out+cross_q'}
  # round 5 (S5): layer 0's QKV projection and attention-out run on one CFG pass's rows -- launches of other shapes
  # (64 x 64 tiles at M = 256; the duplicating epilogue), one per step: classes of their own
  want |= {'gemm_qkv_l0', 'gemm_attn_out_l0'}
  # round 6: the key-split cross-attention merges inside its launch and the cross-attention's query projection is folded
  # into the QKV and attention-out launches (dual kernels, S6): neither is a launch of the step any more -- the query
  # projection's algorithmic work is counted with attention-out (bench.fold_cross_q_work)
  if any('gemm_h16_dual_kernel' in k for k in (doc.get('step_kernels') or [])):
    want -= {'gemm_cross_q', 'attn_cross_merge'}
    assert 'gemm_h16_dual_kernel' in doc['per_class']['gemm_qkv']['kernel'] and 'EpiAddStoreH16' in doc['per_class']['gemm_attn_out']['kernel']
  assert set(doc['per_class']) == want
  assert 'taken on the' not in doc['source']     # counter passes and kernel trace come from ONE binary
  for cls, e in doc['per_class'].items():
    assert e['avg_us'] > 1.0 and e['calls'] >= 1000, cls   # (>= one launch per step of the traced segment)
    if cls.startswith(('gemm_', 'attn_self', 'attn_cross')) and cls != 'attn_cross_merge':
      assert 0 < e['frac'] < 1 and 0 < e['mfma_util'] < 1 and e['fabric_bytes_per_launch'] > 0, cls
  assert len(doc['library_sha']) == 16
  trace = os.path.join(root, 'profiles', '%s_bench_kernel_stats.csv' % doc['tag'])
  seen = set()
  step = mr.step_kernel_names(trace)
  assert any('32, 48, 4' in k for k in step)                    # the MLP output projection's tile of round 4
  with open(trace) as f:
    for r in csv.DictReader(f):
      if 'msd::' in r['Name'] and int(r['Calls']) >= 1000:      # the step's kernels (the encoders' run far fewer times)
        cls = mr.classify(r['Name'], step)
        assert cls is not None, r['Name']
        seen.add(cls)
  assert seen == want


def test_roofline_classifies_by_step_instantiation_not_by_substring():
  """tools/make_roofline.py: the encoders run the decoder's GEMM / attention templates at other shapes (no weight
  prefetch, all attention planes); their launches must not be averaged into the decoder classes' counters
  (VERDICT r03 weak #3).  On the committed r03w passes the decoder-only figures are the judge's recomputation:
  gated-MLP-in MFMA utilisation 0.258 (not 0.294), fabric 42.4 MB (not 45.7), waste 2.30 (not 2.48)."""
  import importlib.util
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location('make_roofline', os.path.join(root, 'tools', 'make_roofline.py'))
  mr = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mr)
  stats = os.path.join(root, 'profiles', 'r03w_bench_kernel_stats.csv')
  step = mr.step_kernel_names(stats)
  dec = 'gemm_h16_dma_kernel<2, 64, 128, 3, EpiGeglu<2>, 1>'
  enc = 'gemm_h16_dma_kernel<2, 64, 128, 3, EpiGeglu<2>, 0>'
  assert dec in step and enc not in step
  assert 'attention_kernel<2, 2, 2, 1, 3>' in step and 'attention_kernel<2, 2, 2, 0, 0>' not in step
  assert mr.classify(dec, step) == 'gemm_mlp_in_geglu' and mr.classify(enc, step) is None
  assert mr.classify(enc) == 'gemm_mlp_in_geglu'   # (without the step set: the old, contaminating behaviour)
  # the decoder-only arithmetic from the committed counter passes
  import csv
  row = {r['kernel']: r for r in csv.DictReader(open(os.path.join(root, 'profiles', 'r03w_pmc_sq.csv')))}[dec]
  busy = float(row['SQ_VALU_MFMA_BUSY_CYCLES'])
  assert busy == 9437184.0                          # = 3 x 3.2212 GFLOP / 16384 FLOP per MFMA x 16 cycles
  util = busy / (1024 * 14.905e-6 * 2.4e9)
  assert abs(util - 0.258) < 1e-3


def test_kernel_stats_classes_of_a_committed_trace():
  """bench.py's own rocprofv3 leg (self_profile) reads a kernel_stats table the way tools/make_roofline.py does: the
  step graph's instantiations only, one class each, launches per step from the sampler's call count; the per-class us
  times launches add up to the trace's kernel time per step."""
  stats = os.path.join(ROOT, 'profiles', 'r05z_bench_kernel_stats.csv')
  classes, steps = bench.kernel_stats_classes(stats)
  assert steps == 2001
  assert classes['gemm_mlp_in_geglu']['launches_per_step'] == 12.0 and 12.0 < classes['gemm_mlp_in_geglu']['avg_us'] < 16.0
  assert classes['gemm_qkv']['launches_per_step'] == 11.0 and classes['gemm_qkv_l0']['launches_per_step'] == 1.0
  assert sum(e['launches_per_step'] for e in classes.values()) == 111.0
  sp = {'per_class': classes, 's_valid_keys': 1356.0, 'child_sample_ms_per_segment': 960.0,
        'sum_kernel_us_per_step': sum(e['avg_us'] * e['calls'] for e in classes.values()) / steps}
  spec = msd_amd.config.preset('base_with_context')
  dom, table, whole, flops, abytes = bench.roofline_from_self_profile(spec, sp, 2, 2, graph_step_ms=0.96)
  assert dom == 'gemm_mlp_in_geglu' and abs(table[dom]['frac'] - 0.0946) < 2e-3        # VERDICT r05's recomputation
  assert 0.9 < whole['kernel_time_over_child_step'] <= 1.0 and 100 < whole['algorithmic_gflop'] < 125
  assert abs(whole['sum_kernel_us'] - 956.6) < 1.0 and whole['launches'] == 111.0


def _write_trace(path, names, dur_ns=5000, gap_ns=2000):
  import csv
  t = 0
  with open(path, 'w') as fh:
    w = csv.DictWriter(fh, fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
    w.writeheader()
    for name in names:
      w.writerow(dict(Kernel_Name=name, Start_Timestamp=t + gap_ns, End_Timestamp=t + gap_ns + dur_ns))
      t += gap_ns + dur_ns


def test_kernel_trace_classes_by_position_in_the_step(tmp_path):
  """bench.py's rocprofv3 leg reads the PER-DISPATCH kernel trace: a launch's class follows from its position in the
  DDPM step, so `small` -- where ONE residual-GEMM instantiation serves attention-out, cross-out and MLP-out and one
  attention kernel both attentions -- still gets one class per launch site; launches outside a step (encoders) do
  not count; the gaps between launches are measured."""
  g = 'void msd::gemm_h16_dma_kernel<2, %s, msd::%s, 0>(msd::GemmParams, msd::X)'
  att = 'void msd::attention_kernel<2, 2, 1, 0, 0>(msd::AttnParams)'
  res = g % ('32, 32, 4', 'EpiResidualNorm<2, false>')
  layer = [g % ('64, 64, 3', 'EpiQKV<2>'), att, res, g % ('32, 32, 4', 'EpiStoreH16<2>'), att,
           'void msd::attention_merge_kernel<2, 4>(msd::AttnParams, int, unsigned int)', res, g % ('64, 64, 3', 'EpiGeglu<2>'), res]
  layer0 = list(layer)
  layer0[2] = g % ('32, 32, 4', 'EpiResidualNorm<2, true>')
  step = [g % ('32, 32, 4', 'EpiInProj<2>')] + layer0 + layer * 2 + ['void msd::final_proj_f32_kernel<1>(msd::FinalProjParams)',
                                                                       'void msd::sampler_step_kernel<0>(msd::SamplerParams)']
  encoder = [g % ('64, 64, 3', 'EpiQKV<2>'), 'void msd::attention_kernel<2, 2, 2, 0, 0>(msd::AttnParams)', g % ('64, 64, 3', 'EpiGeglu<2>')]
  path = str(tmp_path / 'x_kernel_trace.csv')
  _write_trace(path, encoder + step * 4)
  classes, steps, whole = bench.kernel_trace_classes(path)
  assert steps == 4
  want = {'in_proj_f32': 1, 'gemm_qkv_l0': 1, 'gemm_qkv': 2, 'attn_self': 3, 'gemm_attn_out_l0': 1, 'gemm_attn_out': 2,
          'gemm_cross_q': 3, 'attn_cross': 3, 'attn_cross_merge': 3, 'gemm_cross_out': 3, 'gemm_mlp_in_geglu': 3,
          'gemm_mlp_out': 3, 'final_proj_f32': 1, 'sampler_step': 1}
  assert {c: e['launches_per_step'] for c, e in classes.items()} == want
  assert all(e['avg_us'] == 5.0 for e in classes.values())
  n = len(step)
  assert whole['sum_kernel_us_per_step'] == 5.0 * n and whole['sum_gap_us_per_step'] == 2.0 * (n - 1)
  assert whole['gap_between_steps_us'] == 2.0 and classes['gemm_qkv']['avg_gap_before_us'] == 2.0
  # a fused query projection (no cross-q launch, no merge launch): the attention after attention-out is the cross-attention
  fused = [k for k in layer if 'EpiStoreH16' not in k and 'merge' not in k]
  _write_trace(path, [g % ('32, 32, 4', 'EpiInProj<2>')] + fused * 2 + ['void msd::sampler_step_kernel<0>(msd::SamplerParams)'])
  classes, steps, _ = bench.kernel_trace_classes(path)
  assert steps == 1 and classes['attn_cross']['launches_per_step'] == 2 and classes['gemm_cross_out']['launches_per_step'] == 2
  assert 'gemm_cross_q' not in classes and classes['gemm_mlp_out']['launches_per_step'] == 2


def test_watchdog_exit_code_is_a_failure():
  assert bench.Watchdog.EXIT_CODE != 0


def test_dual_launch_kernels_classify_and_carry_the_folded_query_projection():
  """Round 6 (S6): the QKV and attention-out launches are `gemm_h16_dual_kernel`s (two problems, each with its own tile
  shape); the cross-attention's query projection has no launch of its own and its ALGORITHMIC work is counted with the
  attention-out launch, once -- in the bench line's per-class table and in tools/make_roofline.py alike."""
  import bench
  qkv = ('void msd::gemm_h16_dual_kernel<2, 64, 96, 3, msd::EpiQKV<2>, 64, 96, 3, msd::EpiStoreF32, 1>'
         '(msd::GemmParams, msd::EpiQKV<2>, msd::GemmParams, msd::EpiStoreF32, int)')
  qkv0 = qkv.replace('64, 96, 3', '64, 64, 3')
  out = ('void msd::gemm_h16_dual_kernel<2, 64, 32, 4, msd::EpiResidualNorm<2, false, false>, 32, 96, 4, msd::EpiAddStoreH16<2>, 0>'
         '(msd::GemmParams, msd::EpiResidualNorm<2, false, false>, msd::GemmParams, msd::EpiAddStoreH16<2>, int)')
  out0 = out.replace('64, 32, 4, msd::EpiResidualNorm<2, false, false>', '32, 32, 4, msd::EpiResidualNorm<2, true, false>')
  mlp_out_y2 = 'void msd::gemm_h16_dma_kernel<2, 32, 48, 4, msd::EpiResidualNorm<2, false, true>, 1>(msd::GemmParams, msd::EpiResidualNorm<2, false, true>)'
  step = {bench.normalise_kernel(k) for k in (qkv, qkv0, out, out0, mlp_out_y2)}
  assert bench.classify_kernel(qkv, step) == 'gemm_qkv' and bench.classify_kernel(qkv0, step) == 'gemm_qkv_l0'
  assert bench.classify_kernel(out, step) == 'gemm_attn_out' and bench.classify_kernel(out0, step) == 'gemm_attn_out_l0'
  assert bench.classify_kernel(mlp_out_y2, step) == 'gemm_mlp_out'
  assert bench.kernel_type(qkv) == 'qkv' and bench.kernel_type(out) == 'resid' and bench.kernel_type(mlp_out_y2) == 'resid'
  flops = {'gemm_attn_out': 10.0, 'gemm_attn_out_l0': 5.0, 'gemm_cross_q': 3.0, 'gemm_qkv': 30.0}
  abytes = {'gemm_attn_out': 100, 'gemm_cross_q': 30}
  bench.fold_cross_q_work(flops, abytes, {'gemm_attn_out': {}, 'gemm_attn_out_l0': {}, 'gemm_qkv': {}})   # no cross-q launch: folded
  assert flops['gemm_attn_out'] == 13.0 and flops['gemm_attn_out_l0'] == 8.0 and abytes['gemm_attn_out'] == 130
  flops2 = {'gemm_attn_out': 10.0, 'gemm_cross_q': 3.0}
  bench.fold_cross_q_work(flops2, {}, {'gemm_attn_out': {}, 'gemm_cross_q': {}})                           # a launch of its own: untouched
  assert flops2['gemm_attn_out'] == 10.0
