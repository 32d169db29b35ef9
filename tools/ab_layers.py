"""A/B of two library builds on the hot layers, planner's own picks: A3D_LIB=<lib> python tools/ab_layers.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops
from tools.sweep_igemm import LAYERS, timeit
B = int(os.environ.get('B', 32))
PREC = os.environ.get('PREC', 'fp32')
for name, h, w, c, k, ks, st, pad in LAYERS:
    if name not in ('conv2d_1', 'conv2d_2', 'conv2d_3', 'conv2d_4', 'fine2', 'conv2d_0', 'fine1'): continue
    d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad, precision=PREC)
    x = torch.randn((B, h, w, c), device='cuda'); wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
    bias = torch.zeros(k, device='cuda'); y = torch.empty((B, d.ho, d.wo, k), device='cuda'); dz = torch.randn_like(y)
    dx = torch.empty_like(x); dw = torch.empty_like(wt); db = torch.empty(k, device='cuda')
    modes = {'fwd': lambda: ops.conv2d_fwd(d, x, wt, bias, y, 'relu'), 'bwd_f': lambda: ops.conv2d_bwd_filter(d, x, dz, dw, db)}
    if c > 3: modes['bwd_d'] = lambda: ops.conv2d_bwd_data(d, dz, wt, dx, relu_mask=x)
    out = []
    for mode, fn in modes.items():
        timeit(fn, 3)
        out.append(f'{mode} {min(timeit(fn, 5) for _ in range(3)):7.1f}')
    print(f'{name:9s}', ' | '.join(out), flush=True)
