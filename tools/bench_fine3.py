"""fine/third alone (src/models.py:250-251): forward, and both gradients in one pass (a3d_conv2d_bwd_both) beside the
two-launch path it replaces.   python tools/bench_fine3.py [batch]"""
import sys
import torch
sys.path.insert(0, '.')
from ann3depth_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = ops.conv_desc(B, 55, 74, 64, 1, 5, 5, 1, 'SAME')
x = torch.randn((B, 55, 74, 64), device='cuda')
w = torch.randn((5, 5, 64, 1), device='cuda') * 0.02
b = torch.zeros(1, device='cuda')
y = torch.empty((B, 55, 74, 1), device='cuda')
dz = torch.randn_like(y)
dw, db = torch.empty_like(w), torch.empty_like(b)
dx = torch.empty_like(x)
dx16 = torch.empty_like(x, dtype=torch.bfloat16)


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


mb = x.numel() * 4 / 1e6
us = t(lambda: ops.conv2d_fwd(d, x, w, b, y, None))
print(f'B={B} forward            {us:7.1f} us  {mb / us:6.2f} TB/s of {mb:.0f} MB')
us = t(lambda: (ops.conv2d_bwd_filter(d, x, dz, dw, db), ops.conv2d_bwd_data(d, dz, w, dx, relu_mask=x)))
print(f'B={B} bwd two launches   {us:7.1f} us')
us = t(lambda: ops.conv2d_bwd_both(d, x, dz, w, dw, db, dx))
print(f'B={B} bwd both (fp32 dx) {us:7.1f} us  {2 * mb / us:6.2f} TB/s of {2 * mb:.0f} MB')
us = t(lambda: ops.conv2d_bwd_both(d, x, dz, w, dw, db, dx16))
print(f'B={B} bwd both (bf16 dx) {us:7.1f} us  {1.5 * mb / us:6.2f} TB/s of {1.5 * mb:.0f} MB')
