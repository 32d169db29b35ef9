"""Times every MSDN layer (B=32) x direction under each tile config / split-K of the implicit-GEMM kernel.
Tuning aid for csrc/igemm_host.hip:plan_gemm; run on the GPU box:  python tools/sweep_igemm.py > gpurun_out/sweep.txt"""
import os
os.environ.setdefault('A3D_TUNING', '1')   # the library reads its A3D_FORCE_* switches per launch only then
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402

B = int(os.environ.get('B', 32))
PREC = os.environ.get('PREC', 'fp32')
LAYERS = [  # name, h, w, cin, cout, k, stride, pad
    ('conv2d_0', 228, 304, 3, 96, 11, 4, 'VALID'), ('conv2d_1', 27, 37, 96, 256, 5, 1, 'SAME'),
    ('conv2d_2', 13, 18, 256, 384, 3, 1, 'SAME'), ('conv2d_3', 13, 18, 384, 384, 3, 1, 'SAME'),
    ('conv2d_4', 13, 18, 384, 256, 3, 2, 'VALID'), ('fine1', 228, 304, 3, 63, 9, 2, 'VALID'),
    ('fine2', 55, 74, 64, 64, 5, 1, 'SAME'), ('fine3', 55, 74, 64, 1, 5, 1, 'SAME'),
    ('dense_0', 1, 1, 12288, 4096, 1, 1, 'VALID'), ('dense_1', 1, 1, 4096, 4070, 1, 1, 'VALID'),
]
CFGS = ['128x128', '128x96', '128x64', '128x32', '64x64', '32x128', '64x128', '128x128w8', '128x64w8', 'G128x128w8', 'G128x64w8', 'N128x128']       # N: the second-generation kernel (igemm2.h)


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


ALL = []


def main():
    only = [a for a in sys.argv[1:] if not a.startswith('--')] or None
    for name, h, w, c, k, ks, st, pad in LAYERS:
        if only and name not in only:
            continue
        d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad, precision=PREC)
        x = torch.randn((B, h, w, c), device='cuda')
        wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
        bias = torch.zeros(k, device='cuda')
        y = torch.empty((B, d.ho, d.wo, k), device='cuda')
        dz = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        flops = 2.0 * B * d.ho * d.wo * k * ks * ks * c
        modes = {'fwd': lambda: ops.conv2d_fwd(d, x, wt, bias, y, 'relu'),
                 'bwd_f': lambda: ops.conv2d_bwd_filter(d, x, dz, dw, None)}
        if c > 3:
            modes['bwd_d'] = lambda: ops.conv2d_bwd_data(d, dz, wt, dx)
        for mode, fn in modes.items():
            os.environ.pop('A3D_FORCE_CFG', None)
            os.environ.pop('A3D_FORCE_SPLITK', None)
            t_auto = timeit(fn)
            res = []
            if PREC != 'fp32':
                for bn in (64, 128):
                    for sk in (1, 2, 4, 8, 16, 32, 64):
                        os.environ['A3D_BF16_BN'] = str(bn)
                        os.environ['A3D_FORCE_SPLITK'] = str(sk)
                        res.append((timeit(fn, 3), f'bn{bn}', sk))
                os.environ.pop('A3D_BF16_BN', None)
                os.environ.pop('A3D_FORCE_SPLITK', None)
            for ci, cn in enumerate(CFGS if PREC == 'fp32' else []):
                for sk in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 20, 24, 28, 32, 40, 48, 64, 96, 128, 256):
                    os.environ['A3D_FORCE_CFG'] = str(ci)
                    os.environ['A3D_FORCE_SPLITK'] = str(sk)
                    try:
                        t = timeit(fn, 3)
                    except Exception as e:   # noqa
                        continue
                    res.append((t, cn, sk))
                    if t > 6 * t_auto and sk >= 64:
                        break
            ALL.append({'layer': name, 'mode': mode, 'M': {'fwd': B * d.ho * d.wo, 'bwd_d': B * h * w, 'bwd_f': ks * ks * c}[mode],
                        'N': {'fwd': k, 'bwd_d': c, 'bwd_f': k}[mode],
                        'K': {'fwd': ks * ks * c, 'bwd_d': ks * ks * k, 'bwd_f': B * d.ho * d.wo}[mode],
                        'auto_us': t_auto, 'results': [(cn, sk, t) for t, cn, sk in res]})
            res.sort()
            best = ' | '.join(f'{cn} sk{sk} {t:.0f}us {flops / t / 1e6:.0f}TF' for t, cn, sk in res[:6])
            print(f'{name:9s} {mode:6s} {flops / 1e9:6.2f}GF auto {t_auto:7.0f}us {flops / t_auto / 1e6:5.0f}TF || {best}', flush=True)
    os.environ.pop('A3D_FORCE_CFG', None)
    os.environ.pop('A3D_FORCE_SPLITK', None)
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'sweep_full.json')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(ALL, open(out, 'w'))


if __name__ == '__main__':
    main()
