"""Prints the kernel_stats CSV of the newest rocprofv3 run under a directory: name, calls, average us."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us")
