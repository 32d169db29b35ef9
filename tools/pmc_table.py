"""Per-kernel averages of every counter in a rocprofv3 --pmc CSV (one pass), with the kernel's average duration from the
kernel trace of the same run:   python tools/pmc_table.py <counter_collection.csv> <kernel_trace.csv> [substring]"""
import collections
import csv
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
seen, n = set(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('void ', '').split('(')[0]
    tot[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id']))
        n[k] += 1
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r['Kernel_Name'].replace('void ', '').split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
want = sys.argv[3] if len(sys.argv) > 3 else ''
for k in sorted(tot, key=lambda k: -sum(dur[k])):
    if want not in k:
        continue
    us = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    print(f'{k}  launches {n[k]}  avg {us:.1f} us')
    for c, v in sorted(tot[k].items()):
        print(f'    {c:32s} {v / n[k]:16.0f}')
