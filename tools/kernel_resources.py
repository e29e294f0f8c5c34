"""Register / LDS / scratch budget of every kernel in csrc/libmsd_amd.so, read from the code object's metadata
(no GPU needed): extracts the gfx950 image from the .hip_fatbin offload bundle, runs llvm-readelf --notes, and
prints one row per kernel with the occupancy the unified 512-entry VGPR file allows (waves per SIMD =
512 // (vgpr + agpr rounded up to 8); the dynamic LDS of the GEMM / attention kernels is chosen at launch and
is what actually pins them to one block per CU).

  python tools/kernel_resources.py [path/to/libmsd_amd.so] > profiles/rNN_kernel_resources.txt
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'music-spectrogram-diffusion_amd', 'csrc', 'libmsd_amd.so')


def section(path, name):
  out = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '-S', '-W', path], text=True)
  for line in out.splitlines():
    m = re.search(r'\]\s+%s\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)' % re.escape(name), line)
    if m:
      return int(m.group(2), 16), int(m.group(3), 16)
  raise SystemExit('no %s section in %s' % (name, path))


def device_image(path):
  off, size = section(path, '.hip_fatbin')
  blob = open(path, 'rb').read()[off:off + size]
  magic = b'__CLANG_OFFLOAD_BUNDLE__'
  assert blob.startswith(magic), 'not an offload bundle'
  n, = struct.unpack_from('<Q', blob, len(magic))
  p = len(magic) + 8
  for _ in range(n):
    o, s, t = struct.unpack_from('<QQQ', blob, p)
    triple = blob[p + 24:p + 24 + t].decode()
    p += 24 + t
    if 'gfx950' in triple:
      return blob[o:o + s], triple
  raise SystemExit('no gfx950 image in the bundle')


def demangle(names):
  out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
  return out.splitlines()


def main():
  img, triple = device_image(lib)
  with tempfile.NamedTemporaryFile(suffix='.co') as f:
    f.write(img)
    f.flush()
    notes = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], text=True)
  kernels = []
  for blk in re.split(r'\n\s*- \.agpr_count:', notes)[1:]:
    blk = '.agpr_count:' + blk
    get = lambda k: re.search(r'\.%s:\s*(\S+)' % k, blk)
    name = get('name').group(1)
    row = dict(name=name, **{k: int(get(k).group(1)) for k in
                             ('agpr_count', 'vgpr_count', 'sgpr_count', 'group_segment_fixed_size',
                              'private_segment_fixed_size', 'vgpr_spill_count', 'sgpr_spill_count',
                              'max_flat_workgroup_size')})
    kernels.append(row)
  names = demangle([k['name'] for k in kernels])
  print('# %s (%s), %d kernels' % (os.path.relpath(lib, ROOT), triple, len(kernels)))
  print('# vgpr = arch VGPRs incl. AGPRs as the metadata counts them (unified file); waves/SIMD = 512 // ceil8(vgpr)')
  print('%5s %5s %5s %8s %8s %6s %6s %6s  %s' % ('vgpr', 'agpr', 'sgpr', 'lds(fix)', 'scratch', 'vspill', 'sspill', 'w/SIMD', 'kernel'))
  for k, n in sorted(zip(kernels, names), key=lambda kn: kn[1]):
    total = k['vgpr_count']
    occ = min(8, 512 // max(8, -(-total // 8) * 8))
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*\)$', '', n).replace('msd::', '').replace('(anonymous namespace)::', '')
    print('%5d %5d %5d %8d %8d %6d %6d %6d  %s' % (total, k['agpr_count'], k['sgpr_count'], k['group_segment_fixed_size'],
                                                  k['private_segment_fixed_size'], k['vgpr_spill_count'],
                                                  k['sgpr_spill_count'], occ, n))
  spills = [n for k, n in zip(kernels, names) if k['vgpr_spill_count'] or k['private_segment_fixed_size']]
  print('# kernels with scratch or VGPR spills: %d' % len(spills))


if __name__ == '__main__':
  main()
