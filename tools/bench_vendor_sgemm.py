"""Context for the fp32 roofline fraction: what the vendor's plain fp32 GEMM (torch.matmul -> rocBLAS / hipBLASLt)
reaches on this GPU for the GEMM shapes of the hot conv layers (explicit operands, no im2col gather) and for a large
square.  Not part of the product path.     python tools/bench_vendor_sgemm.py > gpurun_out/vendor_sgemm.json"""
import json

import torch

torch.backends.cuda.matmul.allow_tf32 = False
SHAPES = [  # (what, M, N, K)
    ('conv2d_1 fwd as GEMM', 31968, 256, 2400),
    ('conv2d_1 bwd-data as GEMM', 31968, 96, 6400),
    ('conv2d_1 bwd-filter as GEMM (A^T B)', 2400, 256, 31968),
    ('conv2d_3 fwd as GEMM', 7488, 384, 3456),
    ('square 8192', 8192, 8192, 8192),
]
out = []
for what, m, n, k in SHAPES:
    transposed = 'A^T' in what
    a = torch.randn((k, m) if transposed else (m, k), device='cuda')
    b = torch.randn((k, n), device='cuda')
    f = (lambda: torch.matmul(a.t(), b)) if transposed else (lambda: torch.matmul(a, b))
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    out.append({'what': what, 'm': m, 'n': n, 'k': k, 'us': round(best * 1e3, 1),
                'tflops': round(2.0 * m * n * k / best / 1e9, 1)})
print(json.dumps({'library': 'torch.matmul fp32 (rocBLAS / hipBLASLt as shipped with torch %s)' % torch.__version__,
                  'gemms': out}))
