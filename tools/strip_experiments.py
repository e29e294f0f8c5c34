#!/usr/bin/env python3
"""One-off (round 5): remove the experiments / ablation conditionals from the product sources.

`#if MSD_EXPERIMENTS` is evaluated as 0, `#if defined(MSD_DMA_ABL) && ...` as 0; every other conditional is kept.
The sources as they were (with the experiments): tools/ubench/exp/restore_src_r04.sh rebuilds tools/ubench/exp/src_r04/ from history.
usage: strip_experiments.py FILE...   (rewrites in place)
"""
import re
import sys

FALSE = [re.compile(r'^\s*#\s*if\s+MSD_EXPERIMENTS\b'), re.compile(r'^\s*#\s*if\s+defined\(MSD_DMA_ABL\)')]
TRUE = [re.compile(r'^\s*#\s*if\s+!\s*MSD_EXPERIMENTS\b')]
IF = re.compile(r'^\s*#\s*(if|ifdef|ifndef)\b')
ELSE = re.compile(r'^\s*#\s*else\b')
ELIF = re.compile(r'^\s*#\s*elif\b')
ENDIF = re.compile(r'^\s*#\s*endif\b')


def strip(text):
  out = []
  stack = []   # entries: ['keep'] (ordinary conditional) or ['known', value_now]
  def emitting():
    return all(e[0] == 'keep' or e[1] for e in stack)
  for line in text.split('\n'):
    if any(r.match(line) for r in FALSE):
      stack.append(['known', False]); continue
    if any(r.match(line) for r in TRUE):
      stack.append(['known', True]); continue
    if IF.match(line):
      if emitting(): out.append(line)
      stack.append(['keep']); continue
    if ELSE.match(line):
      if stack[-1][0] == 'known':
        stack[-1][1] = not stack[-1][1]; continue
      if emitting(): out.append(line)
      continue
    if ELIF.match(line):
      assert stack[-1][0] == 'keep', line
      if emitting(): out.append(line)
      continue
    if ENDIF.match(line):
      e = stack.pop()
      if e[0] == 'keep' and emitting(): out.append(line)
      continue
    if emitting(): out.append(line)
  assert not stack
  return '\n'.join(out)


for f in sys.argv[1:]:
  src = open(f).read()
  dst = strip(src)
  open(f, 'w').write(dst)
  print(f, len(src.split('\n')), '->', len(dst.split('\n')))
