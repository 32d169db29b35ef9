"""Times one bf16-storage convolution (forward + bwd-data) of a given shape:  python tools/ring_probe.py n h w c k ks [st pad]
(study tool for igemm_ring.h: with the diagnostic build, A3D_LIB=tools/ab/liba3d_diag.so A3D_TUNING=1 A3D_DBG=7, the K
dependence of the timings separates a launch's fixed cost from its cost per k-tile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.bench_layers import timeit  # noqa: E402

n, h, w, c, k, ks = (int(a) for a in sys.argv[1:7])
st = int(sys.argv[7]) if len(sys.argv) > 7 else 1
pad = sys.argv[8] if len(sys.argv) > 8 else 'SAME'
X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
bf = torch.bfloat16
d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, precision='bf16')
x = torch.randn((n, h, w, c), device='cuda').to(bf)
wb = (torch.randn((ks, ks, c, k), device='cuda') * 0.01).to(bf)
bias = torch.zeros(k, device='cuda')
y = torch.empty((n, d.ho, d.wo, k), device='cuda', dtype=bf)
dz = torch.randn((n, d.ho, d.wo, k), device='cuda').to(bf)
dx = torch.empty_like(x)
dd = ops.with_storage(d, X | W | Y)
flops = 2.0 * n * d.ho * d.wo * k * ks * ks * c
for name, fn in (('fwd', lambda: ops.conv2d_fwd(dd, x, wb, bias, y, 'relu')),
                 ('bwd_d', lambda: ops.conv2d_bwd_data(dd, dz, wb, dx, relu_mask=x))):
    if name == 'bwd_d' and st != 1:
        continue
    t = timeit(fn, reps=30)
    print(f'{n}x{h}x{w}x{c}->{k} k{ks} {name:6s} {flops / 1e9:7.2f} GF {t:8.1f} us {flops / t / 1e6:7.1f} TF', flush=True)
