"""Build the HIP/gfx950 libraries in-tree with hipcc.

The shared library is the product's compute path; it is built next to the
sources (music-spectrogram-diffusion_amd/csrc/libmsd_amd.so) so that it travels
with the repository snapshot to the GPU box.  Two builds of the same sources:
libmsd_amd.so (operand planes in IEEE half: precisions 'f16x3' / 'f16') and
libmsd_amd_bf16.so (-DMSD_PLANE_BF16=1, bfloat16 planes: 'bf16x3' / 'bf16'; csrc/common.h).  The measured-and-rejected
kernels of rounds 2 - 4 and their environment switches live OUTSIDE the product tree: tools/ubench/exp/src_r04 -- round 4's
sources (ABI 4), rebuilt from history by tools/ubench/exp/restore_src_r04.sh, not kept in the tree -- and ``build(experiments=True)`` / ``--experiments`` builds
tools/ubench/exp/libmsd_amd_exp.so from there, if that directory exists.  Nothing in csrc/ refers to it.
``python -m`` cannot name this package (dash), so run:  python music-spectrogram-diffusion_amd/build_native.py
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
EXP = os.path.join(ROOT, 'tools', 'ubench', 'exp')
LIB = os.path.join(CSRC, 'libmsd_amd.so')
LIB_EXP = os.path.join(EXP, 'libmsd_amd_exp.so')
LIBS = {'f16': (LIB, []), 'bf16': (os.path.join(CSRC, 'libmsd_amd_bf16.so'), ['-DMSD_PLANE_BF16=1'])}
EXP_SRC = os.path.join(EXP, 'src_r04')   # round 4's sources with the experiments still inside
EXP_LIBS = {'exp': (LIB_EXP, ['-DMSD_EXPERIMENTS=1'])}
SOURCES = ['msd_api.hip']
HEADERS = ['common.h', 'phase_stamps.h', 'gemm_h16.h', 'gemm_f32.h', 'attention.h', 'elementwise.h',
           os.path.join('..', '..', 'include', 'msd_amd.h')]
EXP_HEADERS = ['chain.h', 'gemm_h16_exp.h', 'gemm_splitk_exchange.inc', 'gemm_h16_pair.h', 'gemm_h16_wide.h', 'gemm_h16_ls.h']


def _hipcc():
  for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if c and os.path.exists(c):
      return c
  raise RuntimeError('hipcc not found (ROCm toolchain required to build the HIP library)')


def needs_build(lib: str = LIB) -> bool:
  if not os.path.exists(lib):
    return True
  t = os.path.getmtime(lib)
  if os.path.abspath(lib) == os.path.abspath(LIB_EXP):
    deps = [os.path.join(EXP_SRC, f) for f in os.listdir(EXP_SRC)] + [os.path.join(EXP, f) for f in EXP_HEADERS]
  else:
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS + [os.path.join('..', 'build_native.py')]]
  return any(os.path.getmtime(f) > t for f in deps)


def check_prefetch_registers(listing: str) -> str:
  """The weight prefetch (csrc/gemm_h16.h prefetch_weights) loads from inline asm into registers the compiler does
  not know are written late; a compiler that copied or reused one of them would produce a library that corrupts
  its own epilogue.  check_prefetch_regs.py (next to this file) verifies the DEVICE LISTING OF THE BINARY BEING BUILT (whatever
  ROCm version builds it); an unsafe listing fails the build."""
  tool = os.path.join(HERE, 'check_prefetch_regs.py')   # ships inside the package
  if not os.path.exists(tool):
    raise RuntimeError('%s is missing: the build cannot verify the prefetch registers of the library' % tool)
  out = subprocess.run([sys.executable, tool, listing], capture_output=True, text=True)
  if out.returncode != 0:
    raise RuntimeError('prefetch register check failed on %s:\n%s\n%s' % (listing, out.stdout[-3000:], out.stderr[-2000:]))
  return out.stdout.strip().split('\n')[-1]


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
  """Builds both product libraries (each only if older than its sources) and, with `experiments`, the A/B library of
  tools/ubench/exp; returns the default one.  Each compile keeps its device listing (-save-temps, in a scratch
  directory) and runs the prefetch register check on it."""
  import tempfile
  targets = [(lib, defs, CSRC) for lib, defs in LIBS.values()]
  if experiments:
    if not os.path.isdir(EXP_SRC):
      raise RuntimeError('%s is missing: run `bash tools/ubench/exp/restore_src_r04.sh` first (the experiments build needs round 4\'s sources)' % EXP_SRC)
    targets += [(lib, defs, EXP_SRC) for lib, defs in EXP_LIBS.values()]
  for lib, defs, src in targets:
    if not force and not needs_build(lib):
      continue
    with tempfile.TemporaryDirectory(prefix='msd_build_') as tmp:
      out = os.path.join(tmp, os.path.basename(lib))
      # the output is named RELATIVE to the scratch cwd: an absolute path in a randomly named directory ends up
      # inside the library and would give every build of the same sources another sha256 (the profiles are stamped
      # with the hash of the library they ran on; built this way it is reproducible: same sources + compiler -> same hash)
      # ... and -ffile-prefix-map keeps the checkout path out of __FILE__ / debug strings: the hash is the same from
      # any clone (tests/test_abi.py checks the built library for the path)
      cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-save-temps',
             '-fno-gpu-rdc', '-Wno-unused-result', '-ffile-prefix-map=%s=.' % ROOT, '-ffile-prefix-map=%s=.' % tmp,
             '-cuid=msd_amd',   # the compilation-unit id (__hip_cuid_*) is otherwise a hash of the source PATH
             '-I', src] + defs + ['-o', os.path.basename(lib)] + [os.path.join(src, s) for s in SOURCES]
      if verbose:
        print('[build_native]', ' '.join(cmd), flush=True)
      subprocess.run(cmd, check=True, cwd=tmp)
      listings = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith('.s') and 'gfx950' in f]
      if not listings:
        raise RuntimeError('no gfx950 device listing (*gfx950*.s, -save-temps) in %s: cannot verify the prefetch '
                           'registers; the scratch directory holds %s' % (tmp, sorted(os.listdir(tmp))))
      for l in listings:
        verdict = check_prefetch_registers(l)
        if verbose:
          print('[build_native] prefetch registers:', verdict, flush=True)
      shutil.copyfile(out, lib + '.tmp')
      os.chmod(lib + '.tmp', 0o755)
      os.replace(lib + '.tmp', lib)
  return LIB


if __name__ == '__main__':
  build(force='--force' in sys.argv, experiments='--experiments' in sys.argv)
  print(LIB)
