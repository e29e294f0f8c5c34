"""Static check of the weight-prefetch trick (csrc/gemm_h16.h prefetch_weights): the inline-asm loads target
registers the compiler must neither copy nor reuse before the kernel ends -- it does not know they are written
late.  Scans a device assembly listing (hipcc -save-temps / --cuda-device-only -S) and fails if a destination
register of such a load can be WRITTEN again on any path from the load to the end of the kernel (a read of a
register pair that merely contains it, e.g. a packed-math broadcast operand, is harmless: only a write could be
overwritten by the late load).

Paths are followed through the kernel's control flow (labels, s_branch, s_cbranch_*): every instruction reachable
from the load is checked, both sides of every conditional branch.  (Round 2's scan was linear in text order, which is
the same thing for a single straight-line epilogue but flags the OTHER arm of a kernel with two bodies --
gemm_h16_dual_kernel -- whose code merely follows in the listing.)

usage: python music-spectrogram-diffusion_amd/check_prefetch_regs.py /tmp/msd.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
# the prefetch touches carry the tag `msd_prefetch` in their asm text (other inline-asm loads are waited for normally)
idx = [i for i, l in enumerate(lines) if 'global_load_dword' in l and 'msd_prefetch' in l and i > 0 and 'ASMSTART' in lines[i - 1]]
NO_DST = ('global_store', 'ds_write', 'buffer_store', 'flat_store', 's_', 'v_cmp', 'ds_bpermute', 'scratch_store')


def writes(line, r):
  """does this instruction write VGPR r?"""
  l = line.split(';')[0].strip()
  ops = l.split(None, 1)
  if not ops or len(ops) < 2 or l.endswith(':') or l.startswith('.'):
    return False
  dst = ops[1].split(',')[0]
  if ops[0].startswith(NO_DST) and not ops[0].startswith('v_cmpx'):
    return False
  return bool(re.search(r'\bv%d\b' % r, dst)) or any(int(a) <= r <= int(b) for a, b in re.findall(r'v\[(\d+):(\d+)\]', dst))


bad = 0
by_func = {}
for i in idx:
  s = max(x for x in starts if x <= i)
  by_func.setdefault(s, []).append(i)
for s, ii in by_func.items():
  if 'chain_kernel' in lines[s]:
    continue   # several tiles per block: the chain kernels never enable the prefetch (pf.rows == 0 there)
  end = min([x for x in starts if x > ii[-1]] + [len(lines)])
  labels = {m.group(1): j for j in range(s, end) for m in [re.match(r'^(\.LBB\w+):', lines[j])] if m}
  for i in ii:
    r = int(re.search(r'global_load_dword v(\d+)', lines[i]).group(1))
    seen, work, hit = set(), [i + 1], None
    while work and hit is None:
      j = work.pop()
      while j < end and j not in seen:
        seen.add(j)
        l = lines[j].split(';')[0].strip()
        op = l.split(None, 1)[0] if l else ''
        # (another tagged touch into the same register is not a hazard: nobody reads these registers)
        if writes(lines[j], r) and not ('global_load_dword v%d,' % r in l and 'ASMSTART' in lines[j - 1]):
          hit = j
          break
        if op == 's_endpgm':
          break
        if op.startswith(('s_cbranch', 's_branch')):
          tgt = l.split()[-1]
          if tgt in labels:
            work.append(labels[tgt])
          if op == 's_branch':
            break
        j += 1
    if hit is not None:
      print('%s: v%d (loaded at line %d) is written again at line %d: %s' % (lines[s][:70], r, i, hit, lines[hit].strip()))
      bad += 1
print('%d prefetch loads in %d kernels, %d unsafe' % (len(idx), len(by_func), bad))
sys.exit(1 if bad else 0)
