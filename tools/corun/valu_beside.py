"""Does a weight stream with its arithmetic on the VECTOR pipe keep its bandwidth beside an MFMA-bound GEMM?  (study tool)
    python tools/corun/valu_beside.py
Streams 256 MB (read-modify-write or read-only) with 0 / 16 / 32 scalar-times-vector FMAs per element — 32 is the dense
layers' ratio at batch 32 — alone and beside fine/second's forward GEMM (hinted A3D_HINT_SHARE_CU as in the train step)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ann3depth_amd import ops  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'libcorun_valu.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-shared', '-fPIC', os.path.join(here, 'valu.hip'), '-o', so])
lib = ctypes.CDLL(so)
lib.corun_valu.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

B = 32
d = ops.conv_desc(B, 55, 74, 64, 64, 5, 5, 1, 'SAME', hints=ops.HINT_SHARE_CU)            # fine/second, as the step launches it
x = torch.randn((B, 55, 74, 64), device='cuda')
w = torch.randn((5, 5, 64, 64), device='cuda') * 0.01
b = torch.zeros(64, device='cuda')
y = torch.empty((B, 55, 74, 64), device='cuda')
big = torch.randn(256 * 1024 * 1024 // 4, device='cuda') * 1e-3
sink = torch.zeros(1 << 22, device='cuda')
n4 = big.numel() // 4
side = torch.cuda.Stream()


def gemm(reps=10):
    for _ in range(reps):
        ops.conv2d_fwd(d, x, w, b, y, 'relu')


def ev():
    return torch.cuda.Event(enable_timing=True)


gemm(3)
torch.cuda.synchronize()
e0, e1 = ev(), ev()
e0.record(); gemm(10); e1.record(); torch.cuda.synchronize()
alone = e0.elapsed_time(e1) * 1e3 / 10
print(f'fine/second forward (hinted) alone: {alone:.1f} us')
for rmw in (1, 0):
    for fmas in (0, 16, 32):
        for grid in (1024, 2048):
            def stream(reps):
                for _ in range(reps):
                    lib.corun_valu(big.data_ptr(), n4, grid, fmas, rmw, sink.data_ptr(), side.cuda_stream)
            stream(2)
            torch.cuda.synchronize()
            s0, s1 = ev(), ev()
            with torch.cuda.stream(side):
                s0.record(); stream(4); s1.record()
            torch.cuda.synchronize()
            per_pass = big.numel() * 4 * (2 if rmw else 1)
            gbs = 4 * per_pass / (s0.elapsed_time(s1) * 1e-3) / 1e9
            passes = max(4, int(12 * alone * 1e-6 * gbs * 1e9 / per_pass) + 2)
            with torch.cuda.stream(side):
                s0.record(); stream(passes); s1.record()
            g0, g1 = ev(), ev()
            g0.record(); gemm(10); g1.record()
            torch.cuda.synchronize()
            beside = g0.elapsed_time(g1) * 1e3 / 10
            gbs_b = passes * per_pass / (s0.elapsed_time(s1) * 1e-3) / 1e9
            print(f'{"read-modify-write" if rmw else "read-only        "} {fmas:2d} FMAs/element grid {grid:5d}: alone {gbs:6.0f} GB/s | beside the GEMM '
                  f'{gbs_b:6.0f} GB/s (whole run, GEMMs cover part of it), GEMM {alone:.0f} -> {beside:.0f} us', flush=True)
