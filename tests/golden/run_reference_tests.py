"""Run the REFERENCE's own unit tests (music_spectrogram_diffusion/layers_test.py) against the reference's own
layers.py, both executed over the NumPy stand-in of jax / flax in ref_shim.py (build container only).

What it shows: the stand-in is faithful enough that the known-answer tests the reference ships -- multi-head
attention with explicit kernels, attention + bias, every mask-making helper, DenseGeneral with ones kernels, the
Embed module -- pass on it, in float64 ("wide") and in float32 arrays.  What it cannot run: the one test of the
autoregressive decoding cache (`self.variable('cache', ...)`, mutable collections): not on the inference path of
this model, not in the stand-in.  The outcome is committed as tests/golden/ref_layers_test_report.txt.

    python tests/golden/run_reference_tests.py
"""
import importlib.util
import io
import os
import sys
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

SKIP = {'test_multihead_dot_product_attention_caching': "needs flax's mutable 'cache' collection (autoregressive decoding)"}


def run(wide):
  ref_shim.WIDE = wide
  suite = unittest.TestSuite()
  loader = unittest.TestLoader()
  for cls_name in ('AttentionTest', 'EmbeddingTest', 'DenseTest'):
    cls = getattr(MOD, cls_name)
    for name in loader.getTestCaseNames(cls):
      if name in SKIP:
        continue
      suite.addTest(cls(name))
  buf = io.StringIO()
  res = unittest.TextTestRunner(stream=buf, verbosity=2).run(suite)
  return res, buf.getvalue()


if __name__ == '__main__':
  ref_shim.load_reference()
  ref_shim.install_absl_testing()
  path = os.path.join(ref_shim.REFERENCE_ROOT, 'music_spectrogram_diffusion', 'layers_test.py')
  spec = importlib.util.spec_from_file_location('music_spectrogram_diffusion.layers_test', path)
  MOD = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(MOD)
  lines = ['reference layers_test.py over tests/golden/ref_shim.py (reference code from %s)' % path]
  ok = True
  for wide in (True, False):
    res, text = run(wide)
    lines.append('')
    lines.append('== arrays in %s: ran %d, failures %d, errors %d' % (
        'float64 (WIDE)' if wide else 'float32', res.testsRun, len(res.failures), len(res.errors)))
    lines += [l for l in text.splitlines() if l.endswith(('ok', 'FAIL', 'ERROR')) or l.startswith(('FAIL', 'ERROR'))]
    for _, tb in res.failures + res.errors:
      lines.append(tb.strip().splitlines()[-1])
    ok = ok and res.wasSuccessful()
  lines.append('')
  lines += ['not run: %s -- %s' % kv for kv in SKIP.items()]
  report = '\n'.join(lines) + '\n'
  open(os.path.join(HERE, 'ref_layers_test_report.txt'), 'w').write(report)
  print(report)
  sys.exit(0 if ok else 1)
