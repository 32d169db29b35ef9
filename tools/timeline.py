"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per-queue busy time, overlap between the two HIP
streams, and the idle gaps on the critical path.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --steps 6 --warmup 3 ...
    python tools/timeline.py gpurun_out/tl/**/*_kernel_trace.csv [step_kernel_substring]"""
import csv
import glob
import sys
from collections import defaultdict

path = sorted(glob.glob(sys.argv[1], recursive=True))[-1]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'silog_fwd'
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']))
rows.sort()
# a step = from one occurrence of the anchor kernel (the loss: once per step) to the next
marks = [i for i, r in enumerate(rows) if anchor in r[3]]
if len(marks) < 3:
    sys.exit(f'need at least three "{anchor}" launches, found {len(marks)}')
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
print(f'{path}\nstep window {1e-3 * (t1 - t0):.1f} us, {len(step)} kernels')
byq = defaultdict(list)
for s, e, q, n in step:
    byq[q].append((s, e, n))
for q, ks in byq.items():
    busy = sum(e - s for s, e, _ in ks)
    print(f'  queue {q}: {len(ks)} kernels, busy {1e-3 * busy:.1f} us')
# union busy / both busy
ev = []
for s, e, q, n in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, one, two = 0, t0, 0, 0
for t, d in ev:
    if depth == 1: one += t - last
    elif depth >= 2: two += t - last
    depth += d; last = t
print(f'  exactly one kernel running {1e-3 * one:.1f} us, two or more {1e-3 * two:.1f} us, none {1e-3 * (t1 - t0 - one - two):.1f} us')
# gaps on the main queue
mainq = max(byq, key=lambda q: len(byq[q]))
ks = byq[mainq]
gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2][:60], ks[i + 1][2][:60]) for i in range(len(ks) - 1)]
print(f'  main queue {mainq}: sum of gaps {1e-3 * sum(max(g[0], 0) for g in gaps):.1f} us; largest:')
for g in sorted(gaps, reverse=True)[:8]:
    print(f'    {1e-3 * g[0]:7.1f} us  after {g[1]}  before {g[2]}')
if '-v' in sys.argv:
    for s, e, q, n in step:
        print(f'  {1e-3 * (s - t0):8.1f} {1e-3 * (e - s):7.1f} q{q} {n[:90]}')
