"""How much does a bandwidth-bound grid on another stream slow an MFMA-bound GEMM, by the grid's shape?  (study tool)
    python tools/corun/run.py      -> table: streaming grid (blocks x threads, unroll, nt) | GB/s alone | GEMM us alone / beside"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ann3depth_amd import ops  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'libcorun.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-shared', '-fPIC', os.path.join(here, 'stream.hip'), '-o', so])
lib = ctypes.CDLL(so)
lib.corun_stream.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

B = 32
d = ops.conv_desc(B, 55, 74, 64, 64, 5, 5, 1, 'SAME')            # fine/second
x = torch.randn((B, 55, 74, 64), device='cuda')
w = torch.randn((5, 5, 64, 64), device='cuda') * 0.01
b = torch.zeros(64, device='cuda')
y = torch.empty((B, 55, 74, 64), device='cuda')
big = torch.zeros(256 * 1024 * 1024 // 4, device='cuda')          # 256 MB: read + written = 512 MB per pass
n4 = big.numel() // 4
side = torch.cuda.Stream()


def gemm(reps=10):
    for _ in range(reps):
        ops.conv2d_fwd(d, x, w, b, y, 'relu')


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


gemm(3)
alone = timed(lambda: gemm(10)) / 10
print(f'fine/second forward alone: {alone:.1f} us')
for grid, block, unroll, nt in [(256, 64, 1, 1), (256, 256, 1, 1), (512, 256, 1, 1), (1024, 256, 1, 1), (2048, 256, 1, 1), (2048, 256, 4, 1),
                                (8192, 256, 1, 1), (2048, 256, 1, 0), (8192, 256, 1, 0), (32768, 256, 1, 0)]:
    def stream(reps):
        for _ in range(reps):
            lib.corun_stream(big.data_ptr(), n4, grid, block, unroll, nt, side.cuda_stream)
    stream(2)
    torch.cuda.synchronize()
    t = timed(lambda: stream(4)) / 4          # timed on the default stream's events: includes the wait below
    # proper: time the side stream alone
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        s0.record()
        stream(4)
        s1.record()
    torch.cuda.synchronize()
    gbs = 4 * 2 * big.numel() * 4 / (s0.elapsed_time(s1) * 1e-3) / 1e9
    # beside: keep the side stream busy for the whole GEMM measurement
    torch.cuda.synchronize()
    passes = max(4, int(12 * alone * 1e-6 * gbs * 1e9 / (2 * big.numel() * 4)) + 2)      # enough to outlast the GEMMs
    with torch.cuda.stream(side):
        s0.record()
        stream(passes)
        s1.record()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()                       # (no synchronize here: the side stream's passes are running)
    gemm(10)
    g1.record()
    torch.cuda.synchronize()
    beside = g0.elapsed_time(g1) * 1e3 / 10
    gbs_b = passes * 2 * big.numel() * 4 / (s0.elapsed_time(s1) * 1e-3) / 1e9
    print(f'stream {grid:5d} x {block:3d} unroll {unroll} nt {nt}: {gbs:7.0f} GB/s alone | GEMM {beside:6.1f} us beside ({beside / alone:.2f}x), '
          f'stream then {gbs_b:7.0f} GB/s')
