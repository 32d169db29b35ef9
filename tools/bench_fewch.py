"""The few-channel layers' filter gradient alone: LDS-staged kernel (fewch.hip) vs the generic window-run form (A3D_FEWCH=0
in a tuning process), with and without the pool's gradient fused.   A3D_TUNING=1 [A3D_FEWCH=0] python tools/bench_fewch.py"""
import sys
import torch
sys.path.insert(0, '.')
from ann3depth_amd import ops

CASES = {'conv2d_0 B=32': (32, 228, 304, 3, 96, 11, 4, 96), 'fine/first B=32': (32, 228, 304, 3, 63, 9, 2, 64),
         'conv2d_0 B=64': (64, 228, 304, 3, 96, 11, 4, 96), 'fine/first B=64': (64, 228, 304, 3, 63, 9, 2, 64),
         'dcnf conv11 768': (768, 100, 100, 3, 64, 11, 1, 64)}


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


ONLY = sys.argv[1:]            # e.g. 'B=32': only the cases whose name contains one of these
for name, (n, h, w, c, k, ks, st, ld) in CASES.items():
    if ONLY and not any(o in name for o in ONLY):
        continue
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, 'VALID')
    x = torch.randn((n, h, w, c), device='cuda')
    dz = torch.randn((n, d.ho, d.wo, k), device='cuda')
    dw = torch.empty((ks, ks, c, k), device='cuda')
    db = torch.empty((k,), device='cuda')
    gf = 2.0 * ks * ks * c * k * n * d.ho * d.wo / 1e9
    us = t(lambda: ops.conv2d_bwd_filter(d, x, dz, dw, db))
    line = f'{name:18s} {gf:7.2f} GF  plain {us:8.1f} us {gf / us * 1e3:6.1f} TF'
    ph, pw = d.ho // 2, d.wo // 2
    pooled = torch.randn((n, ph, pw, ld), device='cuda')
    dpool = torch.randn((n, ph, pw, ld), device='cuda')
    arg = torch.randint(0, 4, (n, ph, pw, k), device='cuda', dtype=torch.uint8)
    if ops.conv2d_bwd_filter_pooled_supported(d):
        us = t(lambda: ops.conv2d_bwd_filter_pooled(d, x, dpool, pooled, arg, dw, db))
        line += f' | pool gradient fused {us:8.1f} us'
        if ld == k:
            us = t(lambda: (ops.maxpool2x2_bwd_idx(arg, pooled, dpool, dz, relu_mask=True), ops.conv2d_bwd_filter(d, x, dz, dw, db)))
            line += f' | two launches {us:8.1f} us'
    print(line, flush=True)
