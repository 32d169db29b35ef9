import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for mb in (67, 201, 805):
    n=mb*1000*1000//4
    a=torch.empty(n,device='cuda'); b=torch.randn(n,device='cuda'); c=torch.randn(n,device='cuda')
    tz=t(lambda: a.zero_()); tc=t(lambda: a.copy_(b)); ta=t(lambda: torch.add(b,c,out=a)); tm=t(lambda: b.mul_(0.9)); ts=t(lambda: float(0) if False else b.sum())
    print(f'{mb} MB: fill {tz:.1f} us = {mb/tz*1e-3*1e3:.2f} TB/s | copy {tc:.1f} us = {2*mb/tc:.2f} TB/s total | add(2r1w) {ta:.1f} us = {3*mb/ta:.2f} | inplace mul (1r1w) {tm:.1f} = {2*mb/tm:.2f} | sum(read) {ts:.1f} = {mb/ts:.2f}')
