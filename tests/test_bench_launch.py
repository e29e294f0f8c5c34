"""`python bench.py --gpus N` must run by ITSELF: started without torch.distributed.run it
becomes the launcher of its N ranks.  Exercised here on CPU: gloo backend and a stand-in model class
(tests/bench_stub.py) in place of the HIP synthesizer -- everything else is the product's bench.py: rendezvous on
127.0.0.1, barriers, the --mode drivers of sharding.py (replicas / chained / wavefront / masked), the hand-off probe,
max-over-ranks timing and exactly one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = ['--dist-backend', 'gloo', '--model-factory', 'tests.bench_stub:StubInferenceModel', '--no-cpu-baseline',
        '--batched-songs', '0', '--small-segments', '0']


def run_bench(*flags, env_extra=None, timeout=240):
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  env.update(env_extra or {})
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags) + STUB, cwd=ROOT, env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
  return p


def json_lines(stdout):
  return [json.loads(l) for l in stdout.splitlines() if l.startswith('{') and l.rstrip().endswith('}')]


@pytest.mark.parametrize('mode', ['replicas', 'chained', 'wavefront', 'masked'])
def test_plain_command_starts_its_own_ranks(mode):
  p = run_bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--mode', mode)
  assert p.returncode == 0, p.stderr[-2000:]
  lines = json_lines(p.stdout)
  assert len(lines) == 1, p.stdout            # rank 0 only
  d = lines[0]
  assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
  assert d['metric'] == 'mel-frames/sec' and d['value'] > 0 and d['higher_is_better'] is True
  assert d['config']['mode'] == mode
  # whole-job aggregate: 2 ranks x 3 segments x 256 frames in ms_per_step * 3
  assert abs(d['value'] - 2 * 3 * 256 / (d['ms_per_step'] * 3e-3)) / d['value'] < 1e-3
  h = d['handoff_check']                       # the 128 KiB context message went rank 0 -> 1, content verified
  assert h['ok'] is True and h['hops'] == 1 and h['message_bytes'] == 256 * 128 * 4
  assert 'cpu_baseline' not in d and 'batched' not in d    # N = 1 legs only
  # the census a SCALE record needs: every rank reported itself, distinct processes, and its own timed seconds
  seen = d['ranks_seen']
  assert [r['rank'] for r in seen] == [0, 1] and len({r['pid'] for r in seen}) == 2
  assert len(d['per_rank_seconds']) == 2 and all(t > 0 for t in d['per_rank_seconds'])
  assert abs(max(d['per_rank_seconds']) - d['ms_per_step'] * 3e-3) < 1e-3 * d['ms_per_step'] * 3e-3 + 2e-6   # `value` is over the slowest rank
  assert d['per_rank_precision'] == ['f16x3', 'f16x3'] and all(r['precision'] == 'f16x3' for r in seen)
  assert d['config']['precision'] == 'f16x3' and d['config']['precision_requested'] == 'f16x3'
  if mode == 'replicas':
    # BASELINE config 4 as a leg of the plain line: the 118-segment workload as a wavefront of songs x 2 segments
    # (capped at --handoff-max-segments per rank), every rank's idle fraction, finite outputs
    h4 = d['handoff']
    assert h4['ok'] is True and h4['mode'] == 'wavefront' and h4['segments_per_song'] == 2
    assert h4['songs'] == 16 and h4['segments'] == 32 and h4['value'] > 0
    assert [r['rank'] for r in h4['per_rank']] == [0, 1]
    assert all(0.0 <= r['idle_fraction'] <= 1.0 and r['busy_seconds'] <= r['seconds'] + 1e-6 for r in h4['per_rank'])
    assert abs(h4['ideal_efficiency'] - 16 / 17) < 1e-3
  else:
    assert 'handoff' not in d


def test_a_hung_handoff_leg_still_prints_the_replicas_line():
  """First contact with RCCL point-to-point must not cost the SCALE record: when the hand-off leg hangs (here: rank 1
  never returns from its segments), every rank's watchdog ends its process after --handoff-timeout and rank 0 prints
  the line it already has -- replicas figures intact, `handoff.error` set."""
  p = run_bench('--gpus', '2', '--steps', '2', '--warmup', '1', '--handoff-timeout', '6', env_extra={'MSD_STUB_HANG_RANK': '1'},
                timeout=120)
  lines = json_lines(p.stdout)
  assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
  # ... and the plain command's exit code says that the run did NOT end cleanly (round 6: the watchdog used to leave
  # with 0 -- a hung hand-off looked like a clean run to whoever only reads the exit code)
  assert p.returncode != 0, p.returncode
  d = lines[0]
  assert d['n_gpus'] == 2 and d['value'] > 0 and d['config']['mode'] == 'replicas'
  assert d['handoff']['ok'] is False and 'did not finish' in d['handoff']['error']
  assert d['handoff_check']['ok'] is True      # the probe ran before the leg hung


def test_single_rank_needs_no_launcher():
  p = run_bench('--gpus', '1', '--steps', '2', '--warmup', '1')
  assert p.returncode == 0, p.stderr[-2000:]
  d, = json_lines(p.stdout)
  assert d['n_gpus'] == 1 and 'handoff_check' not in d
  assert d['roofline']['kernel'] == 'gemm_mlp_in_geglu' and d['roofline']['bound'] == 'mfma'


def test_world_size_mismatch_is_an_error():
  p = run_bench('--gpus', '2', '--steps', '1', env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
  assert p.returncode != 0 and 'WORLD_SIZE=1' in (p.stderr + p.stdout)
