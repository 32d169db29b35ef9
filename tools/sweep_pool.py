"""Fused conv + ReLU + 2x2 max-pool forward (a3d_conv2d_pool_fwd) of the three MSDN layers that use it, per tile config."""
import os
import sys
os.environ['A3D_TUNING'] = '1'   # the library reads its A3D_FORCE_* switches per launch only then
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.sweep_igemm import LAYERS, CFGS, timeit  # noqa: E402

B = 32
for name, h, w, c, k, ks, st, pad in LAYERS:
    if name not in ('conv2d_0', 'conv2d_1', 'fine1'):
        continue
    d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad)
    x = torch.randn((B, h, w, c), device='cuda')
    wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
    bias = torch.zeros(k, device='cuda')
    ld = k + (1 if name == 'fine1' else 0)
    yp = torch.empty((B, d.ho // 2, d.wo // 2, ld), device='cuda')
    am = torch.empty((B, d.ho // 2, d.wo // 2, k), dtype=torch.uint8, device='cuda')
    fn = lambda: ops.conv2d_pool_fwd(d, x, wt, bias, yp, 'relu', argmax=am)
    os.environ.pop('A3D_FORCE_CFG', None)
    res = [('auto', timeit(fn))]
    for ci in range(9):
        os.environ['A3D_FORCE_CFG'] = str(ci)
        os.environ['A3D_FORCE_SPLITK'] = '1'
        try:
            res.append((CFGS[ci], timeit(fn, 3)))
        except Exception as e:      # noqa: BLE001
            res.append((CFGS[ci], float('nan')))
    os.environ.pop('A3D_FORCE_CFG', None)
    print(name, ' | '.join(f'{n} {t:.0f}' for n, t in res), flush=True)
