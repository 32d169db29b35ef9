"""dense_1's forward / bwd-data / fused dW+Adam alone and beside fine/first's conv3 grid, for n = 4070 (the reference's
55 x 74 outputs: rows of W start at alternating 8-byte offsets) and n = 4072 (16-byte rows).  (study tool)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ann3depth_amd import ops  # noqa: E402

B = 32
d = ops.conv_desc(B, 228, 304, 3, 63, 9, 9, 2, 'VALID', hints=ops.HINT_SHARE_CU)      # fine/first (+ pool)
x = torch.randn((B, 228, 304, 3), device='cuda')
w = torch.randn((9, 9, 3, 63), device='cuda') * 0.01
b = torch.zeros(63, device='cuda')
cat = torch.empty((B, 55, 74, 64), device='cuda')
side = torch.cuda.Stream()


def conv(reps):
    for _ in range(reps):
        ops.conv2d_pool_fwd(d, x, w, b, cat, 'relu')


def measure(fn, reps=20, beside=False):
    fn(); fn()
    torch.cuda.synchronize()
    if beside:
        with torch.cuda.stream(side):
            conv(3 * reps // 10 + 4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


with torch.cuda.stream(side):
    conv(2)
torch.cuda.synchronize()
for n in (4070, 4072, 4096):
    k, m = 4096, B
    xx = torch.randn((m, k), device='cuda')
    W = torch.randn((k, n), device='cuda') * 0.01
    bias = torch.zeros(n, device='cuda')
    y = torch.empty((m, n), device='cuda')
    dz = torch.randn((m, n), device='cuda')
    dx = torch.empty((m, k), device='cuda')
    slots = [W, torch.zeros_like(W), torch.zeros_like(W), bias, torch.zeros_like(bias), torch.zeros_like(bias)]
    ops_ = {'fwd': lambda: ops.dense_fwd(xx, W, bias, y),
            'bwd_d': lambda: ops.dense_bwd_data(dz, W, dx),
            'dW+Adam': lambda: ops.dense_bwd_filter_adam_tf1(xx, dz, *slots, 0.1, 0.9, 1.0, 0.9, 1.0, 1.0)}
    for name, fn in ops_.items():
        a, bs = measure(fn), measure(fn, beside=True)
        print(f'n = {n}  {name:8s} alone {a:6.1f} us   beside fine/first {bs:6.1f} us', flush=True)

# the reverse of run.py: the GEMM grid is resident first, a thin streaming grid arrives beside it
import ctypes
import subprocess
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'libcorun.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-shared', '-fPIC', os.path.join(here, 'stream.hip'), '-o', so])
lib = ctypes.CDLL(so)
lib.corun_stream.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
big = torch.zeros(64 * 1024 * 1024 // 4, device='cuda')          # 64 MB read + written per pass: a dense_1-sized stream
for grid, block in ((256, 256), (512, 256), (1024, 256), (2048, 256), (8192, 256)):
    fn = lambda: lib.corun_stream(big.data_ptr(), big.numel() // 4, grid, block, 1, 1, torch.cuda.current_stream().cuda_stream)
    a, bs = measure(fn), measure(fn, beside=True)
    print(f'thin stream {grid:5d} x {block}: alone {a:6.1f} us ({2 * big.numel() * 4 / a / 1e6:5.2f} TB/s)   beside fine/first {bs:6.1f} us', flush=True)
