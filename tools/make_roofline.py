"""profiles/<tag>_{bench_kernel_stats,pmc_fetch,pmc_write,pmc_sq}.csv -> profiles/roofline.json

One entry per kernel class of the DDPM step (the names msd_profile_steps / bench.py use), all per launch:
  avg_us                    rocprofv3 --kernel-trace --stats average duration (the graph-replayed kernels)
  algorithmic_gflop, frac   2MNK (or 4 H T S d) / avg_us against the dense bf16 MFMA peak (2.5 PFLOP/s)
  mfma_util                 SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x avg_us x clock): the fraction of SIMD
                            cycles the MFMA pipe was busy (bf16x3 issues 3 MFMAs per algorithmic product)
  fabric_bytes_per_launch   2 x FETCH_SIZE + WRITE_SIZE (KiB -> B; FETCH_SIZE doubled: gfx950 counts 16 B/lane
                            reads at half size, MI355X_MICROARCH.md "HBM"); includes Infinity-Cache hits
  fabric_gbs                the same / avg_us
  algorithmic_bytes, waste  every operand read once + every result written once; waste = fabric / algorithmic
and the sha256 of the library the passes ran on (profiles/<tag>_library_sha.txt, written on the GPU box by
tools/profile_round.sh) so that bench.py can tell a stale profile from a current one.

Usage: python tools/make_roofline.py <tag> [s_valid]     (s_valid: mean cross-attention keys of the run)"""
import csv
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)
import msd_amd  # noqa: E402

CLOCK_GHZ = 2.4      # MI355X peak engine clock (MI355X_MICROARCH.md); SQ cycle counters tick at the engine clock
SIMDS = 256 * 4
# kernel name -> class: ONE table for this tool and for bench.py's own rocprofv3 leg (bench.py KERNEL_CLASSES)
CLASS = bench.KERNEL_CLASSES
normalise = bench.normalise_kernel
classify = bench.classify_kernel
step_kernel_names = bench.step_kernel_names


def small_record(tag):
  """BASELINE config 2 (`small`): profiles/<tag>_kernel_stats.csv (bench.py's own rocprofv3 leg, --self-profile-keep) +
  profiles/<tag>_pmc_{fetch,write,sq}.csv (PRESET=small tools/profile_round.sh) + the bench line's position-based classes
  (profiles/<bench json>: small.roofline.per_class_us) -> profiles/roofline_small.json.  In this model ONE residual-GEMM
  instantiation serves attention-out, cross-out and MLP-out and one attention kernel both attentions, so the counter
  passes (which aggregate by kernel NAME) are reported per instantiation, the durations per class."""
  prof = os.path.join(ROOT, 'profiles')
  stats_csv = os.path.join(prof, '%s_kernel_stats.csv' % tag)
  step_kernels = step_kernel_names(stats_csv)
  rows = {normalise(r['Name']): r for r in csv.DictReader(open(stats_csv))}
  steps = max(int(r['Calls']) for n, r in rows.items() if 'sampler_step_kernel' in n)

  def pmc(name):
    path = os.path.join(prof, '%s_pmc_%s.csv' % (tag, name))
    return {r['kernel']: r for r in csv.DictReader(open(path))} if os.path.exists(path) else {}
  fetch, write, sq = pmc('fetch'), pmc('write'), pmc('sq')
  per_kernel = {}
  for k in sorted(step_kernels):
    r = rows[k]
    us = float(r['TotalDurationNs']) / int(r['Calls']) / 1e3
    e = {'launches_per_step': round(int(r['Calls']) / steps, 3), 'avg_us': round(us, 3)}
    if k in sq:
      busy = float(sq[k]['SQ_VALU_MFMA_BUSY_CYCLES'])
      e['mfma_busy_cycles'] = busy
      e['mfma_util'] = round(busy / (SIMDS * us * 1e-6 * CLOCK_GHZ * 1e9), 4)
    if k in fetch:
      fb = (2.0 * float(fetch[k]['FETCH_SIZE']) + float(write.get(k, {}).get('WRITE_SIZE', 0.0))) * 1024
      e['fabric_bytes_per_launch'] = int(round(fb))
      e['fabric_gbs'] = round(fb / (us * 1e-6) / 1e9, 1)
    per_kernel[k] = e
  doc = {'tag': tag, 'preset': 'small', 'library_sha': open(os.path.join(prof, '%s_library_sha.txt' % tag)).read().strip(),
         'source': 'bench.py self_profile child (rocprofv3 --kernel-trace --stats, preset small) + PRESET=small tools/profile_round.sh '
                   'counter passes (12 DDPM steps each; the encoder shares the step\'s PF = 0 instantiations, 8 launches in ~900)',
         'assumptions': {'clock_ghz': CLOCK_GHZ, 'simds': SIMDS, 'peak_bf16_tflops': bench.PEAK_BF16_TFLOPS,
                         'fetch_correction': 'FETCH_SIZE x2 (gfx950 counts 16 B/lane reads at half size); WRITE_SIZE as reported'},
         'steps_traced': steps, 'per_kernel': per_kernel}
  for cand in sorted(os.listdir(prof)):
    if cand.startswith(tag.split('_')[0]) and cand.endswith('bench_default.json'):
      line = json.load(open(os.path.join(prof, cand)))
      rl = line.get('small', {}).get('roofline', {})
      if rl.get('per_class_us'):
        doc['per_class'] = rl['per_class_us']
        doc['whole_step'] = rl.get('whole_step')
        doc['dominant'] = {k: rl.get(k) for k in ('kernel', 'achieved', 'frac', 'kernel_ms_per_launch')}
        doc['bench_line'] = 'profiles/' + cand
  with open(os.path.join(prof, 'roofline_small.json'), 'w') as f:
    json.dump(doc, f, indent=1)
  for k, e in sorted(per_kernel.items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['launches_per_step']):
    print('%-70s x%-6s %7.2f us  mfma %s  fabric %s MB' % (k, e['launches_per_step'], e['avg_us'], e.get('mfma_util'),
                                                            round(e.get('fabric_bytes_per_launch', 0) / 1e6, 2)))


def main():
  if len(sys.argv) > 2 and sys.argv[2] == '--small':
    return small_record(sys.argv[1])
  tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
  s_valid = float(sys.argv[2]) if len(sys.argv) > 2 else 1100.0 + 256.0
  prof = os.path.join(ROOT, 'profiles')
  # PMC_TAG=<other tag>: take the counter passes of another round (same launches, same bytes) when a round had
  # GPU time for the kernel trace only; the result says so in 'source'
  pmc_tag = os.environ.get('PMC_TAG', tag)
  spec = msd_amd.config.preset('base_with_context')
  flops = bench.class_flops(spec, s_valid, 2)
  abytes = bench.class_bytes(spec, s_valid, 2)
  # the shared 32x32 residual template serves two classes with different M: weight by launch share (1:1)
  flops['gemm_attn_out+gemm_cross_out'] = 0.5 * (flops['gemm_attn_out'] + flops['gemm_cross_out'])
  abytes['gemm_attn_out+gemm_cross_out'] = 0.5 * (abytes['gemm_attn_out'] + abytes['gemm_cross_out'])
  # hoisted query projection (round 3): one launch = the out-projection (M = 2T) + [x (.) gamma | ao] . [Wq ; Wo g Wq]
  # (M = T, K = D + J); with it the 32 x 32 residual template only serves the cross-attention output projection
  flops['gemm_attn_out+cross_q'] = flops['gemm_attn_out'] + flops['gemm_cross_q']          # ALGORITHMIC: 2 T J D once
  abytes['gemm_attn_out+cross_q'] = abytes['gemm_attn_out'] + abytes['gemm_cross_q']
  flops1, abytes1 = bench.class_flops(spec, s_valid, 1), bench.class_bytes(spec, s_valid, 1)
  for c in ('gemm_qkv', 'gemm_attn_out'):   # layer 0's launches on ONE pass's rows (round 5, S5)
    flops[c + '_l0'] = flops1[c]
    abytes[c + '_l0'] = abytes1[c]
  flops.setdefault('attn_cross_merge', 0.0)
  flops.setdefault('sampler_step', 0.0)
  # a class may run as several instantiations (with / without the weight prefetch): call-weighted means
  out = {}
  stats_csv = os.path.join(prof, '%s_bench_kernel_stats.csv' % tag)
  step_kernels = step_kernel_names(stats_csv)
  with open(stats_csv) as f:
    for r in csv.DictReader(f):
      cls = classify(r['Name'], step_kernels)
      if not cls:
        continue
      e = out.setdefault(cls, {'kernel': [], 'calls': 0, '_ns': 0.0})
      e['kernel'].append(r['Name'].split('(')[0].replace('void msd::', '').replace('msd::', ''))
      e['calls'] += int(r['Calls'])
      e['_ns'] += float(r['TotalDurationNs'])
  for e in out.values():
    e['avg_us'] = round(e.pop('_ns') / e['calls'] / 1e3, 3)
    e['kernel'] = ' | '.join(e['kernel'])
  bench.fold_cross_q_work(flops, abytes, out)   # round 6 (S6): no cross-q launch -- its algorithmic work sits with attention-out

  def pmc(name):
    path = os.path.join(prof, '%s_pmc_%s.csv' % (pmc_tag, name))
    if not os.path.exists(path):
      return {}
    acc = {}
    with open(path) as f:
      for r in csv.DictReader(f):
        cls = classify(r['kernel'], step_kernels)
        if not cls:
          continue
        n = float(r['dispatches'])
        a = acc.setdefault(cls, {'_n': 0.0})
        a['_n'] += n
        for k, v in r.items():
          if k not in ('kernel', 'dispatches'):
            a[k] = a.get(k, 0.0) + n * float(v)
    return {cls: {k: v / a['_n'] for k, v in a.items() if k != '_n'} for cls, a in acc.items()}
  fetch, write, sq = pmc('fetch'), pmc('write'), pmc('sq')
  for cls, e in out.items():
    us = e['avg_us']
    e['algorithmic_gflop'] = round(flops.get(cls, 0.0) / 1e9, 4)
    e['tflops'] = round(flops.get(cls, 0.0) / (us * 1e-6) / 1e12, 2)
    e['frac'] = round(e['tflops'] / bench.PEAK_BF16_TFLOPS, 5)
    if cls in sq and 'SQ_VALU_MFMA_BUSY_CYCLES' in sq[cls]:
      busy = float(sq[cls]['SQ_VALU_MFMA_BUSY_CYCLES'])
      e['mfma_busy_cycles'] = busy
      e['mfma_util'] = round(busy / (SIMDS * us * 1e-6 * CLOCK_GHZ * 1e9), 4)
    if cls in fetch:
      fb = (2.0 * float(fetch[cls]['FETCH_SIZE']) + float(write.get(cls, {}).get('WRITE_SIZE', 0.0))) * 1024
      e['fabric_bytes_per_launch'] = int(round(fb))
      e['fabric_gbs'] = round(fb / (us * 1e-6) / 1e9, 1)
      ab = abytes.get(cls)
      if ab:
        e['algorithmic_bytes'] = int(ab)
        e['waste'] = round(fb / ab, 2)
  sha_path = os.path.join(prof, '%s_library_sha.txt' % tag)
  sha = open(sha_path).read().strip() if os.path.exists(sha_path) else 'unknown (%s predates the stamp)' % tag
  hoisted = 'gemm_attn_out+cross_q' in out
  if hoisted and 'gemm_attn_out+gemm_cross_out' in out:   # that template now runs the cross-attention output projection only
    e = out['gemm_attn_out+gemm_cross_out']
    e['algorithmic_gflop'] = round(flops['gemm_cross_out'] / 1e9, 4)
    e['tflops'] = round(flops['gemm_cross_out'] / (e['avg_us'] * 1e-6) / 1e12, 2)
    e['frac'] = round(e['tflops'] / bench.PEAK_BF16_TFLOPS, 5)
    if 'fabric_bytes_per_launch' in e:
      e['algorithmic_bytes'] = int(abytes['gemm_cross_out'])
      e['waste'] = round(e['fabric_bytes_per_launch'] / abytes['gemm_cross_out'], 2)
  launches = {'gemm_attn_out+gemm_cross_out': 12 if hoisted else 24, 'final_proj_f32': 1, 'in_proj_f32': 1, 'sampler_step': 1}
  if 'gemm_qkv_l0' in out:   # round 5: layer 0's self-attention block has classes of its own (attention: same kernel, 12 launches)
    launches.update({'gemm_qkv_l0': 1, 'gemm_attn_out_l0': 1, 'gemm_qkv': 11, 'gemm_attn_out': 11})
  step_us = sum(e['avg_us'] * launches.get(c, 12) for c, e in out.items())
  step_fabric = sum(e.get('fabric_bytes_per_launch', 0) * launches.get(c, 12) for c, e in out.items())
  step_alg = sum(e.get('algorithmic_bytes', 0) * launches.get(c, 12) for c, e in out.items())
  doc = {
      'tag': tag, 'library_sha': sha,
      'source': 'rocprofv3 --kernel-trace --stats and separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ group) over '
                'bench.py, base_with_context B=1 %s CFG (tools/profile_round.sh %s); profiles/%s_*.csv' % (
                    'f16x3', tag, tag) + ('' if pmc_tag == tag else
                '; COUNTER passes (MFMA busy cycles, FETCH_SIZE, WRITE_SIZE) from profiles/%s_pmc_*.csv, taken on the '
                'bfloat16-plane binary of the same kernels: same launches, same operand bytes, same MFMA count -- '
                'mfma_util and fabric GB/s are those counts over THIS trace\'s durations' % pmc_tag),
      'step_kernels': sorted(step_kernels) if step_kernels else None,
      'assumptions': {'clock_ghz': CLOCK_GHZ, 'simds': SIMDS, 'peak_bf16_tflops': bench.PEAK_BF16_TFLOPS,
                      's_valid_keys': s_valid,
                      'fetch_correction': 'FETCH_SIZE x2 (gfx950 counts 16 B/lane reads at half size); WRITE_SIZE as reported'},
      'whole_step': {'sum_kernel_us': round(step_us, 1), 'fabric_bytes': int(step_fabric), 'algorithmic_bytes': int(step_alg),
                     'waste': round(step_fabric / step_alg, 2) if step_alg else None,
                     'fabric_gbs': round(step_fabric / (step_us * 1e-6) / 1e9, 1) if step_us else None},
      'per_class': out,
  }
  with open(os.path.join(prof, 'roofline.json'), 'w') as f:
    json.dump(doc, f, indent=1)
  for c, e in sorted(out.items(), key=lambda kv: -kv[1]['avg_us']):
    print('%-30s %7.2f us  %7.1f TF  frac %.4f  mfma %s  fabric %s MB (%s GB/s)  waste %s' % (
        c, e['avg_us'], e['tflops'], e['frac'], e.get('mfma_util'), round(e.get('fabric_bytes_per_launch', 0) / 1e6, 1),
        e.get('fabric_gbs'), e.get('waste')))
  print('step: %.1f us kernels, fabric %.2f GB vs algorithmic %.2f GB' % (step_us, step_fabric / 1e9, step_alg / 1e9))


if __name__ == '__main__':
  main()
