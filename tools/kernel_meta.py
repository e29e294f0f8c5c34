"""Print per-kernel VGPR/AGPR/LDS/scratch from a hipcc -S (--cuda-device-only) .s file."""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
meta = txt[txt.index('amdhsa.kernels:'):]
for blk in meta.split('\n  - ')[1:]:
  g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
  name = g('name')
  if flt in name:
    print('%-110s vgpr %4s agpr %4s lds %6s scratch %4s sgpr %3s' % (
        name.replace('_ZN3msd', '')[:110], g('vgpr_count'), g('agpr_count'),
        g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('sgpr_count')))
