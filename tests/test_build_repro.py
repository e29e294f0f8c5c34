"""The library hash stamped into profiles/ (tools/profile_round.sh -> profiles/<tag>_library_sha.txt -> roofline.json ->
bench.py `matches_binary`) must identify the SOURCES, not the directory they were built in: a copy of the tree at
another path builds a bit-identical libmsd_amd.so (the absolute source path used to leak in
through __FILE__ and through clang's compilation-unit id)."""
import hashlib
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'music-spectrogram-diffusion_amd')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _sha(path):
  return hashlib.sha256(open(path, 'rb').read()).hexdigest()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_library_hash_does_not_depend_on_the_checkout_path(tmp_path):
  import __graft_entry__
  __graft_entry__.build()                                  # the in-tree library is current
  clone = tmp_path / 'some' / 'other' / 'checkout'
  (clone / 'music-spectrogram-diffusion_amd').mkdir(parents=True)
  shutil.copytree(os.path.join(PKG, 'csrc'), clone / 'music-spectrogram-diffusion_amd' / 'csrc',
                  ignore=shutil.ignore_patterns('*.so', '*.tmp'))
  shutil.copytree(os.path.join(ROOT, 'include'), clone / 'include')
  for f in ('build_native.py', 'check_prefetch_regs.py'):
    shutil.copy(os.path.join(PKG, f), clone / 'music-spectrogram-diffusion_amd' / f)
  code = ("import importlib.util as u; s = u.spec_from_file_location('b', 'music-spectrogram-diffusion_amd/build_native.py'); "
          "m = u.module_from_spec(s); s.loader.exec_module(m); m.LIBS.pop('bf16'); print(m.build(force=True, verbose=False))")
  out = subprocess.run([sys.executable, '-c', code], cwd=clone, capture_output=True, text=True)
  assert out.returncode == 0, out.stdout + out.stderr
  other = out.stdout.strip().splitlines()[-1]
  assert str(clone) in other
  assert _sha(other) == _sha(os.path.join(PKG, 'csrc', 'libmsd_amd.so'))
  assert str(clone).encode() not in open(other, 'rb').read()
