"""Times every MSDN layer x direction (B = 32 unless B=..) with the planner's own choice, the way MSDNReplica.step calls
them (conv + pool fused where the step fuses them).  One line per op: us, TFLOP/s.  A/B two builds of the library on one
GPU box:  A3D_LIB=tools/ab/liba3d_old.so python tools/bench_layers.py ; python tools/bench_layers.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.sweep_igemm import LAYERS  # noqa: E402

B = int(os.environ.get('B', 32))
POOLED = {'conv2d_0', 'conv2d_1', 'fine1'}


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    only = [a for a in sys.argv[1:] if not a.startswith('--')] or None
    rows = []
    tot = 0.0
    for name, h, w, c, k, ks, st, pad in LAYERS:
        if only and name not in only:
            continue
        d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad)
        x = torch.randn((B, h, w, c), device='cuda')
        wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
        bias = torch.zeros(k, device='cuda')
        y = torch.empty((B, d.ho, d.wo, k), device='cuda')
        yp = torch.empty((B, d.ho // 2, d.wo // 2, k), device='cuda')
        am = torch.empty((B, d.ho // 2, d.wo // 2, k), dtype=torch.uint8, device='cuda')
        dz = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        db = torch.empty(k, device='cuda')
        flops = 2.0 * B * d.ho * d.wo * k * ks * ks * c
        modes = {}
        if name in POOLED:
            modes['fwd+pool'] = lambda: ops.conv2d_pool_fwd(d, x, wt, bias, yp, 'relu', am)
        else:
            modes['fwd'] = lambda: ops.conv2d_fwd(d, x, wt, bias, y, 'relu')
        modes['bwd_f'] = lambda: ops.conv2d_bwd_filter(d, x, dz, dw, db)
        if c > 3:
            modes['bwd_d'] = lambda: ops.conv2d_bwd_data(d, dz, wt, dx, relu_mask=x)
        # conv + pool in one launch enumerates whole pool windows only: the last odd row / column is never computed
        flops_pooled = 2.0 * B * (d.ho // 2) * (d.wo // 2) * 4 * k * ks * ks * c
        for mode, fn in modes.items():
            t = timeit(fn)
            tot += t
            f = flops_pooled if mode == 'fwd+pool' else flops      # the FLOPs of the GEMM the launch computes
            rows.append({'layer': name, 'mode': mode, 'us': round(t, 1), 'tflops': round(f / t / 1e6, 1)})
            print(f'{name:9s} {mode:9s} {f / 1e9:6.2f} GF {t:8.1f} us {f / t / 1e6:6.1f} TF', flush=True)
    print(f'total {tot:.1f} us   lib={os.environ.get("A3D_LIB", "in-tree")}')
    out = os.environ.get('OUT')
    if out:
        json.dump(rows, open(out, 'w'))


if __name__ == '__main__':
    main()
