"""Build the HIP/gfx950 libraries in-tree with hipcc.

The shared library is the product's compute path; it is built next to the
sources (music-spectrogram-diffusion_amd/csrc/libmsd_amd.so) so that it travels
with the repository snapshot to the GPU box.  Two builds of the same sources:
libmsd_amd.so (operand planes in IEEE half: precisions 'f16x3' / 'f16') and
libmsd_amd_bf16.so (-DMSD_PLANE_BF16=1, bfloat16 planes: 'bf16x3' / 'bf16'; csrc/common.h).  ``python -m`` cannot name this
package (dash), so run:  python music-spectrogram-diffusion_amd/build_native.py
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libmsd_amd.so')
LIBS = {'f16': (LIB, []), 'bf16': (os.path.join(CSRC, 'libmsd_amd_bf16.so'), ['-DMSD_PLANE_BF16=1'])}
SOURCES = ['msd_api.hip']
HEADERS = ['common.h', 'chain.h', 'gemm_h16.h', 'gemm_h16_pair.h', 'gemm_h16_wide.h', 'gemm_h16_ls.h', 'gemm_f32.h', 'attention.h', 'elementwise.h',
           os.path.join('..', '..', 'include', 'msd_amd.h')]


def _hipcc():
  for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if c and os.path.exists(c):
      return c
  raise RuntimeError('hipcc not found (ROCm toolchain required to build the HIP library)')


def needs_build(lib: str = LIB) -> bool:
  if not os.path.exists(lib):
    return True
  t = os.path.getmtime(lib)
  for f in SOURCES + HEADERS + [os.path.join('..', 'build_native.py')]:
    if os.path.getmtime(os.path.join(CSRC, f)) > t:
      return True
  return False


def check_prefetch_registers(listing: str) -> str:
  """The weight prefetch (csrc/gemm_h16.h prefetch_weights) loads from inline asm into registers the compiler does
  not know are written late; a compiler that copied or reused one of them would produce a library that corrupts
  its own epilogue.  tools/check_prefetch_regs.py verifies the DEVICE LISTING OF THE BINARY BEING BUILT (whatever
  ROCm version builds it); an unsafe listing fails the build."""
  tool = os.path.join(os.path.dirname(HERE), 'tools', 'check_prefetch_regs.py')
  out = subprocess.run([sys.executable, tool, listing], capture_output=True, text=True)
  if out.returncode != 0:
    raise RuntimeError('prefetch register check failed on %s:\n%s' % (listing, out.stdout[-3000:]))
  return out.stdout.strip().split('\n')[-1]


def build(force: bool = False, verbose: bool = True) -> str:
  """Builds both libraries (each only if older than its sources); returns the default one.  Each compile keeps
  its device listing (-save-temps, in a scratch directory) and runs the prefetch register check on it."""
  import tempfile
  for lib, defs in LIBS.values():
    if not force and not needs_build(lib):
      continue
    with tempfile.TemporaryDirectory(prefix='msd_build_') as tmp:
      out = os.path.join(tmp, os.path.basename(lib))
      # the output is named RELATIVE to the scratch cwd: an absolute path in a randomly named directory ends up
      # inside the library and would give every build of the same sources another sha256 (the profiles are stamped
      # with the hash of the library they ran on; built this way it is reproducible: same sources + compiler -> same hash)
      cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-save-temps',
             '-fno-gpu-rdc', '-Wno-unused-result', '-I', CSRC] + defs + ['-o', os.path.basename(lib)] + [os.path.join(CSRC, s) for s in SOURCES]
      if verbose:
        print('[build_native]', ' '.join(cmd), flush=True)
      subprocess.run(cmd, check=True, cwd=tmp)
      listings = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith('gfx950.s')]
      if not listings:
        raise RuntimeError('no device listing next to %s: cannot verify the prefetch registers' % out)
      for l in listings:
        verdict = check_prefetch_registers(l)
        if verbose:
          print('[build_native] prefetch registers:', verdict, flush=True)
      shutil.copyfile(out, lib + '.tmp')
      os.chmod(lib + '.tmp', 0o755)
      os.replace(lib + '.tmp', lib)
  return LIB


if __name__ == '__main__':
  build(force='--force' in sys.argv)
  print(LIB)
