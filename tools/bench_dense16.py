"""Config 5's dense launches alone (bf16 x / dz / weight copies through the LDS-DMA kernel's 64-row weight stream), B = 64, with
a forced split-K factor:   A3D_TUNING=1 A3D_FORCE_SPLITK=<k> python tools/bench_dense16.py"""
import os
os.environ.setdefault('A3D_TUNING', '1')
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.bench_layers import timeit  # noqa: E402

B = int(os.environ.get('B', 64))
bf = torch.bfloat16
st = ops.STORE_W | ops.STORE_X
for name, k, n in (('dense_0', 12288, 4096), ('dense_1', 4096, 4072)):
    x = torch.randn((B, k), device='cuda').to(bf)
    w = (torch.randn((k, n), device='cuda') * 0.01).to(bf)
    b = torch.zeros((1, n), device='cuda')
    y = torch.empty((B, n), device='cuda')
    dz = torch.randn((B, n), device='cuda').to(bf)
    dx = torch.empty((B, k), device='cuda')
    mask = torch.randn((B, k), device='cuda')
    mb_w = k * n * 2 / 1e6
    for mode, fn in (('fwd', lambda: ops.dense_fwd_ex(x, w, b, y, 'relu', precision='bf16', storage=st)),
                     ('bwd_d', lambda: ops.dense_bwd_data_ex(dz, w, dx, mask=mask, scale=1.0, precision='bf16',
                                                             storage=ops.STORE_W | ops.STORE_Y))):
        t = timeit(fn)
        print(f'{name} {mode:6s} splitk {os.environ.get("A3D_FORCE_SPLITK", "auto"):>4s} {t:8.1f} us  {mb_w / t:6.2f} TB/s of the bf16 weights', flush=True)
