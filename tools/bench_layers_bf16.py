"""Times the conv layers of BASELINE config 5's storage mode one by one (B = 64 unless B=..): bf16 activations, bf16 weight
copies, fp32 filter gradients, the storage bits MSDNReplica uses (models.py: store[...]).  One line per layer x
direction: us, TFLOP/s.  A/B two builds on one GPU box:  A3D_LIB=tools/ab/liba3d_x.so python tools/bench_layers_bf16.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.sweep_igemm import LAYERS  # noqa: E402
from tools.bench_layers import timeit  # noqa: E402

B = int(os.environ.get('B', 64))
NAMES = ('conv2d_1', 'conv2d_2', 'conv2d_3', 'conv2d_4', 'fine2')


def main():
    only = [a for a in sys.argv[1:] if not a.startswith('--')] or NAMES
    X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
    bf = torch.bfloat16
    tot = 0.0
    warm = torch.randn((4096, 4096), device='cuda')
    for _ in range(200):                                   # clocks up before the first measurement
        warm = torch.tanh(warm @ warm * 1e-4)
    torch.cuda.synchronize()
    for name, h, w, c, k, ks, st, pad in LAYERS:
        if name not in only:
            continue
        d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad, precision='bf16')
        fine = name == 'fine2'                              # f2 / df2 stay fp32 there
        bits = {'fwd': X | W if fine else X | W | Y, 'bwd_d': X | W if fine else X | W | Y, 'bwd_f': X if fine else X | Y}
        x = torch.randn((B, h, w, c), device='cuda').to(bf)
        zero = os.environ.get('ZERO') == '1'      # all-zero operands: the clock the chip holds without switching activity (DVFS check)
        wt = (torch.randn((ks, ks, c, k), device='cuda') * 0.01)
        wb = wt.to(bf)
        bias = torch.zeros(k, device='cuda')
        y = torch.empty((B, d.ho, d.wo, k), device='cuda', dtype=torch.float32 if fine else bf)
        dz = torch.randn((B, d.ho, d.wo, k), device='cuda').to(torch.float32 if fine else bf)
        if zero:
            x.zero_(), wt.zero_(), wb.zero_(), dz.zero_()
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        db = torch.empty(k, device='cuda')
        flops = 2.0 * B * d.ho * d.wo * k * ks * ks * c
        dd = {m: ops.with_storage(d, b) for m, b in bits.items()}
        modes = {'fwd': lambda: ops.conv2d_fwd(dd['fwd'], x, wb, bias, y, 'relu'),
                 'bwd_f': lambda: ops.conv2d_bwd_filter(dd['bwd_f'], x, dz, dw, db),
                 'bwd_d': lambda: ops.conv2d_bwd_data(dd['bwd_d'], dz, wb, dx, relu_mask=x)}
        for mode, fn in modes.items():
            t = timeit(fn)
            tot += t
            print(f'{name:9s} {mode:6s} {flops / 1e9:6.2f} GF {t:8.1f} us {flops / t / 1e6:6.1f} TF', flush=True)
    print(f'total {tot:.1f} us   lib={os.environ.get("A3D_LIB", "in-tree")}')


if __name__ == '__main__':
    main()
