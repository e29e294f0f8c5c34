"""tools/pin: the ready-to-run pins of the three "partial" rows cannot reach their real
artifacts here (no jax, no tensorflow, no released checkpoint) -- these tests run each script's own comparison code
against a synthetic stand-in (`--self-test`), and check that a planted difference is REPORTED, so that the scripts work
the day an artifact exists."""
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', 'pin', name + '.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_pin_jax_random_self_test_and_planted_difference(tmp_path, capsys):
  pin = _load('pin_jax_random')
  out = str(tmp_path / 'n5.json')
  assert pin.main(['--self-test', '--seeds', '0', '42', '--steps', '3', '--shape', '1', '64', '128', '--json', out]) == 0
  rec = json.load(open(out))
  assert rec['ok'] and len(rec['cases']) == 2 and all(c['init_z']['differing'] == 0 for c in rec['cases'])
  # one flipped mantissa bit in one element of every draw of "the other side" must be found, as 1 ulp
  assert pin.main(['--self-test-flip', '--seeds', '7', '--steps', '2', '--shape', '1', '64', '128', '--json', out]) == 1
  rec = json.load(open(out))
  c = rec['cases'][0]
  assert not rec['ok'] and c['init_z']['differing'] == 1 and c['init_z']['max_ulp'] == 1 and c['init_z']['first'] == [7]
  assert c['step_noise_worst']['differing'] >= 1
  # without jax (this container) the plain call says so with its own exit code
  try:
    import jax  # noqa: F401
    have_jax = True
  except Exception:
    have_jax = False
  if not have_jax:
    assert pin.main([]) == 2
    assert 'not importable' in capsys.readouterr().out


def test_ulp_report_orders_floats_across_zero():
  pin = _load('pin_jax_random')
  a = np.array([0.0, -0.0, 1.0, -1.0], np.float32)
  b = np.array([np.nextafter(np.float32(0), np.float32(1)), 0.0, np.nextafter(np.float32(1), np.float32(2)), -1.0], np.float32)
  r = pin.ulp_report(a, b)
  assert r['differing'] == 2 and r['max_ulp'] == 1            # (+0 and -0 are the same point; one step each elsewhere)


def test_pin_mel_filterbank_self_test(tmp_path):
  pin = _load('pin_mel_filterbank')
  out = str(tmp_path / 'n4.json')
  assert pin.main(['--self-test', '--json', out]) == 0
  rec = json.load(open(out))
  assert rec['ok'] and len(rec['cases']) == 3 and rec['cases'][0]['shape'] == [513, 128]
  # a bank drawn linear in HERTZ (what torchaudio / librosa do) is NOT tf.signal's: the comparison must say so
  def hertz_triangles(nm, nb, sr, lo, hi):
    mel = lambda f: 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
    edges_hz = 700.0 * np.expm1(np.linspace(mel(lo), mel(hi), nm + 2) / 1127.0)
    bins = np.linspace(0.0, sr / 2.0, nb)
    m = np.stack([np.interp(bins, edges_hz[j:j + 3], [0.0, 1.0, 0.0], left=0.0, right=0.0) for j in range(nm)], 1)
    m[0] = 0.0
    return m.astype(np.float32)
  bad = pin.compare(hertz_triangles)
  assert not bad['ok'] and bad['cases'][0]['max_abs_diff'] > 1e-3


def test_pin_t5x_checkpoint_self_test(tmp_path):
  pin = _load('pin_t5x_checkpoint')
  out = str(tmp_path / 'n3.json')
  assert pin.main(['--self-test', '--json', out]) == 0
  rec = json.load(open(out))
  assert rec['ok'] and rec['step'] == 7000 and rec['self_test_damaged_copy_caught'] is True
  assert not (rec['missing'] or rec['unexpected'] or rec['misshapen'])
  # the parameter counts the script expects of the released checkpoints are the model's (411.67 M / 85.0 M rounded)
  for preset, millions in pin.EXPECTED_MILLIONS.items():
    n = sum(int(np.prod(s)) for s in pin.expected_tree(preset).values())
    assert abs(n / 1e6 - millions) < 0.01, (preset, n)
