"""Times the dense-layer entry points (a3d_dense_*) at MSDN's shapes: [B, 12288] x [12288, 4096] and [B, 4096] x [4096, 4070]."""
import os
os.environ.setdefault('A3D_TUNING', '1')   # the library reads its A3D_FORCE_* switches per launch only then
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.bench_layers import timeit  # noqa: E402

B = int(os.environ.get('B', 32))
for name, k, n in (('dense_0', 12288, 4096), ('dense_1', 4096, 4070)):
    x = torch.randn((B, k), device='cuda')
    w = torch.randn((k, n), device='cuda') * 0.01
    b = torch.zeros(n, device='cuda')
    y = torch.empty((B, n), device='cuda')
    dz = torch.randn((B, n), device='cuda')
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    db = torch.empty_like(b)
    keep = (torch.rand((B, n), device='cuda') > 0.5).to(torch.uint8)
    mw, vw, mb, vb = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(b)
    mb_w = k * n * 4 / 1e6
    rows = [('fwd', lambda: ops.dense_fwd(x, w, b, y, 'relu', drop_keep=keep), mb_w),
            ('bwd_d', lambda: ops.dense_bwd_data(dz, w, dx, mask=x, scale=1.0), mb_w),
            ('bwd_f', lambda: ops.dense_bwd_filter(x, dz, dw, db), mb_w),
            ('bwd_f+adam', lambda: ops.dense_bwd_filter_adam_tf1(x, dz, w, mw, vw, b, mb, vb, 0.1, 0.9, 1.0, 0.9, 1.0, 1.0), 2 * mb_w)]
    for mode, fn, mbytes in rows:
        t = timeit(fn)
        print(f'{name} {mode:11s} {t:8.1f} us  {mbytes / t:6.2f} TB/s of weight-sized traffic', flush=True)
