"""Cross-check of the second-generation kernel (csrc/igemm2.h, tile configuration 11) against torch float64 with its plan
pinned: plain launches, split-K factors, stream-K grids, the fused max pool.  Run on the GPU box:
    python tools/gen2_check.py            (exit code 1 and a FAIL line on a mismatch)
Every case has gathered channel counts that are multiples of 32 (forward: Cin, bwd-data: Cout), so all three directions run on
the new kernel; A3D_PLAN_LOG=1 prints the plan of every launch."""
import importlib.util
import os
import sys

os.environ['A3D_TUNING'] = '1'
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], '0', '5']            # fuzz_ops reads its budget / seed from argv at import
spec = importlib.util.spec_from_file_location('fuzz_ops_mod', os.path.join(ROOT, 'tools', 'fuzz_ops.py'))
fz = importlib.util.module_from_spec(spec)
fz.__name__ = 'fuzz_ops_mod'
try:
    spec.loader.exec_module(fz)          # (its own fuzz loop runs for the 0 seconds asked for, then exits)
except SystemExit:
    pass
from ann3depth_amd import ops  # noqa: E402

CASES = [
    # n, h, w, c, k, ks, st, pad
    (2, 27, 37, 96, 256, 5, 1, 'SAME'),        # conv2d_1 kind: halo, 75 k-tiles, tap-inner K order
    (3, 13, 18, 256, 384, 3, 1, 'SAME'),       # conv2d_2
    (2, 13, 18, 384, 384, 3, 1, 'SAME'),       # conv2d_3
    (2, 21, 30, 64, 64, 5, 1, 'SAME'),         # fine/second: one column tile half empty
    (5, 24, 24, 64, 256, 5, 1, 'VALID'),       # DCNF conv2d_1 kind
    (3, 9, 11, 32, 32, 3, 1, 'SAME'),          # one k-tile per tap, M tail, N tail of 96 columns
    (1, 7, 9, 64, 160, 1, 1, 'VALID'),         # 1 x 1: two k-tiles in all
    (4, 17, 19, 32, 36, 3, 2, 'SAME'),         # stride 2 forward / bwd-filter (bwd-data goes to the parity classes)
    (2, 16, 20, 128, 72, 2, 1, 'SAME'),        # even kernel: SAME pads right / below only
    (33, 6, 8, 96, 100, 3, 1, 'SAME'),         # 48 pixels per image: a row tile spans three images
    (2, 30, 41, 32, 64, 7, 1, 'SAME'),         # 49 taps
]
PLANS = [{'A3D_FORCE_SPLITK': '1'}, {'A3D_FORCE_SPLITK': '2'}, {'A3D_FORCE_SPLITK': '3'}, {'A3D_FORCE_SPLITK': '5'},
         {'A3D_FORCE_STREAMK': '3'}, {'A3D_FORCE_STREAMK': '8'}, {'A3D_FORCE_STREAMK': '64'}, {'A3D_FORCE_STREAMK': '256'},
         {'A3D_FORCE_STREAMK': '512'}, {'A3D_FORCE_STREAMK': '700'}]


def clear():
    for v in ('A3D_FORCE_CFG', 'A3D_FORCE_SPLITK', 'A3D_FORCE_STREAMK', 'A3D_FORCE_SK_SLICED'):
        os.environ.pop(v, None)


def pooled(n, h, w, c, k, ks, pad):
    """conv + bias + ReLU + 2x2 max pool in one launch against the two-launch path of the default plan (bit-identical values)."""
    d = ops.conv_desc(n, h, w, c, k, ks, ks, 1, pad)
    g = torch.Generator(device='cuda').manual_seed(h * w + c)
    x = torch.randn((n, h, w, c), device='cuda', generator=g)
    wt = torch.randn((ks, ks, c, k), device='cuda', generator=g) / np.sqrt(ks * ks * c)
    b = torch.randn((k,), device='cuda', generator=g)
    yp = torch.full((n, d.ho // 2, d.wo // 2, k), float('nan'), device='cuda')
    am = torch.empty((n, d.ho // 2, d.wo // 2, k), dtype=torch.uint8, device='cuda')
    clear()
    os.environ['A3D_FORCE_CFG'] = '11'
    ops.conv2d_pool_fwd(d, x, wt, b, yp, 'relu', am)
    clear()
    os.environ['A3D_NO_GEN2'] = '1'
    y = torch.empty((n, d.ho, d.wo, k), device='cuda')
    ops.conv2d_fwd(d, x, wt, b, y, 'relu')
    os.environ.pop('A3D_NO_GEN2')
    ref = torch.nn.functional.max_pool2d(y.permute(0, 3, 1, 2).double(), 2).permute(0, 2, 3, 1)
    return float((yp.double() - ref).norm() / ref.norm())


def main():
    worst, nrun, bad = 0.0, 0, 0
    for case in CASES:
        for plan in PLANS:
            clear()
            os.environ['A3D_FORCE_CFG'] = '11'
            os.environ.update(plan)
            err = fz.run_conv(*case)
            nrun += 1
            worst = max(worst, err)
            tag = ' '.join(f'{k[10:].lower()}{v}' for k, v in plan.items())
            if not (err < fz.TOL):
                bad += 1
                print(f'FAIL {case} {tag}: rel {err:.3e}', flush=True)
        print(f'{case}: ok so far, worst {worst:.2e}', flush=True)
    for case in [(2, 27, 37, 96, 256, 5, 'SAME'), (3, 12, 14, 32, 40, 3, 'VALID'), (2, 9, 21, 64, 64, 3, 'SAME')]:
        err = pooled(*case)
        nrun += 1
        if not (err < fz.TOL_POOL):
            bad += 1
            print(f'FAIL pooled {case}: rel {err:.3e}', flush=True)
        else:
            print(f'pooled {case}: {err:.2e}', flush=True)
    clear()
    print(f'{nrun} gen-2 checks, worst rel-L2 {worst:.2e}, {bad} failed')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
