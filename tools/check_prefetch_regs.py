"""Static check of the weight-prefetch trick (csrc/gemm_h16.h prefetch_weights): the inline-asm loads target
registers the compiler must neither copy nor reuse before the kernel ends.  Scans a device assembly listing
(hipcc --cuda-device-only -S) and fails if a destination register of such a load is mentioned again later in
the same kernel as a DESTINATION (a read of a register pair that merely contains it, e.g. a packed-math broadcast
operand, is harmless: only a write could be overwritten by the late load).   usage: python tools/check_prefetch_regs.py /tmp/msd.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
idx = [i for i, l in enumerate(lines) if 'global_load_dword' in l and i > 0 and 'ASMSTART' in lines[i - 1]]
bad = 0
by_func = {}
for i in idx:
  s = max(x for x in starts if x <= i)
  by_func.setdefault(s, []).append(i)
for s, ii in by_func.items():
  if 'chain_kernel' in lines[s]:
    continue   # several tiles per block: the chain kernels never enable the prefetch (pf.rows == 0 there)
  end = min([x for x in starts if x > ii[-1]] + [len(lines)])
  for i in ii:
    r = int(re.search(r'global_load_dword v(\d+)', lines[i]).group(1))
    for j in range(i + 1, end):
      l = lines[j].split(';')[0].strip()
      ops = l.split(None, 1)
      dst = ops[1].split(',')[0] if len(ops) > 1 else ''   # first operand = destination (stores / branches have none that matter)
      if ops and (ops[0].startswith(('global_store', 'ds_write', 'buffer_store', 'flat_store', 's_', 'v_cmp', 'ds_bpermute')) and not ops[0].startswith('v_cmpx')):
        dst = '' if not ops[0].startswith('ds_bpermute') else dst
      hit = re.search(r'\bv%d\b' % r, dst) or any(int(a) <= r <= int(b) for a, b in re.findall(r'v\[(\d+):(\d+)\]', dst))
      if hit and not ('global_load_dword v%d,' % r in l and 'ASMSTART' in lines[j - 1]):
        print('%s: v%d (loaded at line %d) is touched again at line %d: %s' % (lines[s][:70], r, i, j, l.strip()))
        bad += 1
        break
print('%d prefetch loads in %d kernels, %d unsafe' % (len(idx), len(by_func), bad))
sys.exit(1 if bad else 0)
