"""Config 5's few-channel filter gradient alone (fewch16.hip: bf16 arithmetic, pool gradient fused), B = 64.  Durations from
the kernel trace:  rocprofv3 --kernel-trace --stats -- python3 tools/bench_fewch16.py"""
import sys
import torch
sys.path.insert(0, '.')
from ann3depth_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bf = torch.bfloat16
for name, (k, ks, st, ld) in {'conv2d_0': (96, 11, 4, 96), 'fine/first': (63, 9, 2, 64)}.items():
    d = ops.conv_desc(B, 228, 304, 3, k, ks, ks, st, 'VALID', precision='bf16')
    x = torch.randn((B, 228, 304, 3), device='cuda')
    ph, pw = d.ho // 2, d.wo // 2
    pooled = torch.randn((B, ph, pw, ld), device='cuda').to(bf)
    dpool = torch.randn((B, ph, pw, ld), device='cuda').to(bf)
    arg = torch.randint(0, 4, (B, ph, pw, k), device='cuda', dtype=torch.uint8)
    dw = torch.empty((ks, ks, 3, k), device='cuda')
    db = torch.empty((k,), device='cuda')
    for _ in range(10):
        ops.conv2d_bwd_filter_pooled(d, x, dpool, pooled, arg, dw, db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_bwd_filter_pooled(d, x, dpool, pooled, arg, dw, db)
    e1.record()
    torch.cuda.synchronize()
    gf = 2.0 * ks * ks * 3 * k * B * d.ho * d.wo / 1e9
    us = 1e3 * e0.elapsed_time(e1) / 20
    print(f'{name:12s} B={B} {gf:6.2f} GF  {us:7.1f} us (kernel + reduction)  {gf / us * 1e3:6.1f} TF', flush=True)
