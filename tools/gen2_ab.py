"""Same-box A/B of the first- and second-generation fp32 kernels, plans pinned, launches alternating (the chip's clock drifts
over the first seconds of load and differs from box to box: only alternating runs on one box compare).
    python tools/gen2_ab.py [layer ...]         layers of tools/sweep_igemm.py, plus 'gemm' = a 1x1 conv of 2400 -> 256 channels
                                                over 32000 pixels (the plain GEMM of tools/micro/gemm4.hip)"""
import os
import sys

os.environ['A3D_TUNING'] = '1'
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.sweep_igemm import LAYERS, timeit  # noqa: E402

SPECS = {l[0]: l[1:] for l in LAYERS}
SPECS['gemm'] = (25, 40, 2400, 256, 1, 1, 'VALID')
SPECS['gemmT'] = (25, 40, 256, 2400, 1, 1, 'VALID')      # its bwd-data is the same GEMM with both operands k-contiguous
# DCNF's unary stack at batch 16 = 768 patches (src/models.py:61-83): its 5x5 conv and the first of its 3x3 convs
SPECS['dcnf5'] = (45, 45, 64, 256, 5, 1, 'VALID')
SPECS['dcnf3'] = (20, 20, 256, 256, 3, 1, 'VALID')
BATCH = {'dcnf5': 768, 'dcnf3': 768}
only = sys.argv[1:] or ['gemm', 'conv2d_1', 'conv2d_2', 'conv2d_3', 'fine2']
PLANS = [('gen1 128x128w8', '7'), ('gen2 128x128w4', '11'), ('gen1 128x128w4', '0')]


def clear():
    for v in ('A3D_FORCE_CFG', 'A3D_FORCE_SPLITK', 'A3D_FORCE_STREAMK'):
        os.environ.pop(v, None)


_w = torch.randn((4096, 4096), device='cuda')
for _ in range(100):
    _w @ _w
torch.cuda.synchronize()
for name in only:
    h, w, c, k, ks, st, pad = SPECS[name]
    B = BATCH.get(name, 32)
    d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad)
    x = torch.randn((B, h, w, c), device='cuda'); wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
    bias = torch.zeros(k, device='cuda'); y = torch.empty((B, d.ho, d.wo, k), device='cuda'); dz = torch.randn_like(y)
    dx = torch.empty_like(x); dw = torch.empty_like(wt); db = torch.empty(k, device='cuda')
    flops = 2.0 * B * d.ho * d.wo * k * ks * ks * c
    modes = {'fwd': lambda: ops.conv2d_fwd(d, x, wt, bias, y, 'relu'), 'bwd_f': lambda: ops.conv2d_bwd_filter(d, x, dz, dw, db),
             'bwd_d': lambda: ops.conv2d_bwd_data(d, dz, wt, dx, relu_mask=x)}
    tiles = -(-B * d.ho * d.wo // 128) * -(-k // 128)
    for mode, fn in modes.items():
        if name == 'gemmT' and mode != 'bwd_d':
            continue
        kinds = [('sk', '1')] if ((mode == 'fwd' and tiles >= 400) or name == 'gemmT' or (tiles >= 2000 and mode != 'bwd_f')) else [('st', '512'), ('st', '256')]
        for kind, v in kinds:
            best = {}
            for rnd in range(4):
                for label, cfg in PLANS:
                    clear()
                    os.environ['A3D_FORCE_CFG'] = cfg
                    os.environ['A3D_FORCE_SPLITK' if kind == 'sk' else 'A3D_FORCE_STREAMK'] = v
                    t = timeit(fn, 8)
                    best[label] = min(best.get(label, 1e9), t)
            clear()
            line = ' | '.join(f'{lab} {t:7.1f} us {flops / t / 1e6:6.1f} TF' for lab, t in best.items())
            print(f'{name:9s} {mode:6s} {kind}{v:4s} {line}', flush=True)
