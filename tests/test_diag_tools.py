"""The measurement tooling that is not part of the product but is cited by DESIGN.md must stay usable:

  * tools/diag/phase_timestamps.patch (per-block phase stamps, debug build only) applies to the tree as it is, and the
    product sources contain none of it (the default library is built WITHOUT the stamps);
  * tools/diag/phase_times.py turns a stamp table into the per-phase medians it prints (synthetic stamps here)."""
import importlib.util
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, 'tools', 'diag', 'phase_timestamps.patch')
CSRC = os.path.join(ROOT, 'music-spectrogram-diffusion_amd', 'csrc')


@pytest.mark.skipif(shutil.which('patch') is None, reason='patch(1) not installed')
def test_phase_stamp_patch_applies_to_the_tree(tmp_path):
  dst = tmp_path / 'music-spectrogram-diffusion_amd' / 'csrc'
  shutil.copytree(CSRC, dst, ignore=shutil.ignore_patterns('*.so', '*.tmp'))
  r = subprocess.run(['patch', '-p1', '--dry-run', '-i', PATCH], cwd=tmp_path, capture_output=True, text=True)
  assert r.returncode == 0, r.stdout + r.stderr
  assert 'FAILED' not in r.stdout and 'fuzz' not in r.stdout, r.stdout


def test_product_sources_carry_no_stamps():
  for f in os.listdir(CSRC):
    if f.endswith(('.h', '.hip')):
      text = open(os.path.join(CSRC, f)).read()
      assert 'MSD_TIMESTAMPS' not in text and 'g_msd_ts' not in text, f


def test_phase_times_table_arithmetic():
  """phase_times.py's per-class reduction on a synthetic table: 4 blocks, 2 GHz core clock (s_memtime) against the
  100 MHz s_memrealtime; phases of 1 / 2 / 3 / 0.5 / 1.5 / 0.25 us."""
  spec = importlib.util.spec_from_file_location('phase_times', os.path.join(ROOT, 'tools', 'diag', 'phase_times.py'))
  src = open(spec.origin).read()
  assert "ts = np.zeros((8, 1024, 12), np.uint64)" in src   # the layout this test mirrors
  us = np.array([0, 1, 3, 6, 6.5, 8, 8.25])
  t = np.zeros((4, 12), np.int64)
  for b in range(4):
    t[b, :7] = 1000 + b * 50 + (us * 2000).astype(np.int64)       # 2000 ticks per us
    t[b, 8], t[b, 9] = b, 4
    t[b, 10] = 500 + b                                             # realtime at entry, 10 ns units
    t[b, 11] = t[b, 10] + 825                                      # 8.25 us later
  core = t[:, :7] - t[:, :1]
  real = (t[:, 11] - t[:, 10]).astype(np.float64) * 10.
  ghz = core[:, 6].sum() / real.sum()
  ph = np.diff(core, axis=1) / ghz / 1e3
  assert abs(ghz - 2.0) < 1e-9
  np.testing.assert_allclose(np.median(ph, axis=0), [1, 2, 3, 0.5, 1.5, 0.25], rtol=1e-9)
  # and the script uses exactly these expressions
  for line in ("core = t[:, :7] - t[:, :1]", "ghz = core[:, 6].sum() / real.sum()", "ph = np.diff(core, axis=1) / ghz / 1e3"):
    assert line in src, line


def test_every_library_switch_is_documented():
  """Each environment variable msd_create (or a launch helper) reads is listed in DESIGN.md 11's table of A/B switches."""
  import re
  src = open(os.path.join(CSRC, 'msd_api.hip')).read()
  names = set(re.findall(r'getenv\("(MSD_[A-Z0-9_]+)"\)', src))
  assert len(names) >= 15
  design = open(os.path.join(ROOT, 'DESIGN.md')).read()
  missing = sorted(n for n in names if n not in design)
  assert not missing, 'switches read by the library but absent from DESIGN.md: %s' % missing


def test_tools_do_not_import_the_oracle():
  """tools/ is measurement tooling around the product: like the product it must not reach into oracle/ (directly or
  through tests.helpers, which imports it)."""
  import re
  bad = []
  for d, _, files in os.walk(os.path.join(ROOT, 'tools')):
    for f in files:
      if f.endswith('.py'):
        text = open(os.path.join(d, f)).read()
        if re.search(r'^\s*(from|import)\s+(oracle|tests)\b', text, re.M):
          bad.append(os.path.relpath(os.path.join(d, f), ROOT))
  assert not bad, bad


def test_diag_inputs_equal_the_test_helpers():
  """tools/diag/_inputs.py restates two helpers of tests/helpers.py (without the oracle import): same arrays."""
  import sys
  import msd_amd
  from tests import helpers
  sys.path.insert(0, os.path.join(ROOT, 'tools', 'diag'))
  import _inputs
  spec = msd_amd.config.preset('tiny_context', num_steps=3)
  a, b = helpers.make_batch(spec, batch=2), _inputs.make_batch(spec, batch=2)
  assert a.keys() == b.keys()
  for k in a:
    np.testing.assert_array_equal(a[k], b[k])
  for x, y in zip(helpers.make_noise(spec, batch=2), _inputs.make_noise(spec, batch=2)):
    np.testing.assert_array_equal(x, y)
