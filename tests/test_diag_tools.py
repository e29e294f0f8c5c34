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


def test_phase_stamps_are_compiled_out_of_the_product():
  """The per-block phase stamps live in the sources behind MSD_TIMESTAMPS (round 4: no more patch file to keep in step
  with the tree); every stamp macro expands to NOTHING unless a debug build defines it, and the built product library
  has neither the table nor its read-back entry point."""
  text = open(os.path.join(CSRC, 'phase_stamps.h')).read()   # (round 5: the stamp macros have a header of their own)
  off = text[text.index('#else', text.index('#if MSD_TIMESTAMPS')):]
  off = off[:off.index('#endif')]
  macros = [l for l in off.splitlines() if l.startswith('#define MSD_TS')]
  assert len(macros) >= 4 and all(l.rstrip().endswith(')') for l in macros), macros   # "#define MSD_TS_X(...)" and nothing behind it
  lib = os.path.join(CSRC, 'libmsd_amd.so')
  if os.path.exists(lib):
    blob = open(lib, 'rb').read()
    assert b'msd_debug_timestamps' not in blob and b'g_msd_ts' not in blob


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


def test_product_library_reads_no_environment():
  """Every knob a caller may choose is a msd_config field; the product sources contain no getenv, no experiments /
  ablation conditional and no include that leaves csrc/ + include/ (round 5: the rejected kernels and their switches
  build from round 4's sources, which tools/ubench/exp/restore_src_r04.sh reconstructs from history -- a directory the product never names)."""
  import re
  for f in sorted(os.listdir(CSRC)):
    if not f.endswith(('.h', '.hip')):
      continue
    text = open(os.path.join(CSRC, f)).read()
    assert 'getenv' not in text, f
    for token in ('MSD_EXPERIMENTS', 'kExperiments', 'MSD_DMA_ABL', 'MSD_ATT_ABL'):
      assert token not in text, '%s in %s' % (token, f)
    for inc in re.findall(r'#include\s+"([^"]+)"', text):
      assert 'tools' not in inc, '%s includes %s' % (f, inc)
      target = os.path.normpath(os.path.join(CSRC, inc))
      assert target.startswith(CSRC) or target.startswith(os.path.join(ROOT, 'include')), '%s includes %s' % (f, inc)
  lib = os.path.join(CSRC, 'libmsd_amd.so')
  if os.path.exists(lib):
    nm = subprocess.run(['nm', '-D', '--undefined-only', lib], capture_output=True, text=True).stdout
    assert 'getenv' not in nm
    strings = subprocess.run(['strings', lib], capture_output=True, text=True).stdout
    assert not re.search(r'\bMSD_[A-Z]+(_[A-Z0-9]+)*=?\b', strings.replace('MSD_PREC_', 'x')), 'an MSD_* name survives in the product library'
    assert ROOT not in strings, 'the checkout path is inside the library (its hash would depend on where it was built)'


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
