"""Instruction mix of the hottest loop of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: loopmix.py file.s <substring of the mangled kernel name>"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith('_Z') and key in l.split(':')[0])
end = next(i for i in range(start + 1, len(src)) if src[i].startswith('.Lfunc_end'))
body = src[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r'\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))


def klass(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'):
        return 'lds_read'
    if op.startswith('ds_'):
        return 'lds_write'
    if op.startswith('buffer_load') or op.startswith('global_load'):
        return 'vmem_load'
    if op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_'):
        return 'vmem_other'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


best = None
for a, b in loops:
    c = Counter()
    for l in body[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].startswith('.') or t[0].startswith(';') or t[0].endswith(':'):
            continue
        c[klass(t[0])] += 1
    if best is None or c['mfma'] > best[2]['mfma']:
        best = (a, b, c)
    print(f'loop lines {a}-{b}: ' + ' '.join(f'{k}={v}' for k, v in sorted(c.items())))
a, b, c = best
n = c['mfma'] or 1
print('hottest loop: per MFMA ' + ' '.join(f'{k}={v / n:.2f}' for k, v in sorted(c.items())))
for l in body:
    if 'vgpr_count' in l or 'sgpr_count' in l or 'scratch' in l.lower() and 'size' in l.lower() or 'Occupancy' in l or 'LDSByteSize' in l:
        pass
tail = src[end:end + 60]
for l in tail:
    if any(k in l for k in ('NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy', 'NumSgprs', 'TotalNumVgprs')):
        print(l.strip())
