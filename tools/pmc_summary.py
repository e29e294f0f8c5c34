"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per
dispatch for every kernel (used to fill profiles/*.md).  Usage:
  python tools/pmc_summary.py <counter_collection.csv> [name-filter]"""
import collections
import csv
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else 'msd::'
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
with open(path) as f:
  for row in csv.DictReader(f):
    k = row['Kernel_Name']
    if flt not in k:
      continue
    k = k.split('(')[0].replace('void msd::', '').replace('msd::', '')
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    cnt[k][row['Counter_Name']] += 1
names = sorted({c for k in acc for c in acc[k]})
out = csv.writer(sys.stdout)   # kernel names contain commas (template arguments): quoted
out.writerow(['kernel', 'dispatches'] + names)
for k in sorted(acc):
  n = max(cnt[k].values())
  out.writerow([k, n] + ['%.1f' % (acc[k][c] / max(cnt[k][c], 1)) for c in names])
