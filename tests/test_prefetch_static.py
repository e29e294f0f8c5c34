"""The weight prefetch (csrc/gemm_h16.h prefetch_weights) issues loads from inline asm into registers that the
compiler must neither copy nor reuse before the kernel ends -- it does not know they are loads.  That cannot be
expressed in the source; it is verified on the compiled device listing of the library (CPU only: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('defs', [[], ['-DMSD_PLANE_BF16=1']], ids=['half_planes', 'bfloat16_planes'])
def test_prefetch_destination_registers_are_never_rewritten(tmp_path, defs):
  src = os.path.join(ROOT, 'music-spectrogram-diffusion_amd', 'csrc', 'msd_api.hip')
  listing = str(tmp_path / 'msd.s')
  subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S'] + defs + ['-o', listing, src],
                 check=True, cwd=os.path.dirname(src), capture_output=True)
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'music-spectrogram-diffusion_amd', 'check_prefetch_regs.py'), listing],
                       capture_output=True, text=True)
  print(out.stdout[-400:])
  assert out.returncode == 0, out.stdout[-2000:]
  n_loads = int(out.stdout.strip().split('\n')[-1].split()[0])
  assert n_loads >= 20   # the prefetch is compiled in (GEMM + attention instantiations)
