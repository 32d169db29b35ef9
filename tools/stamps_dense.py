"""Diagnostic: phase timeline of the fused dense dW + Adam kernel from the -DA3D_STAMPS build.
    A3D_LIB=tools/ab/liba3d_stamps.so python tools/stamps_dense.py [dense_0|dense_1]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import _lib, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'dense_0'
k, n = {'dense_0': (12288, 4096), 'dense_1': (4096, 4070)}[name]
B = 32
x = torch.randn((B, k), device='cuda')
w = torch.randn((k, n), device='cuda') * 0.01
b = torch.zeros(n, device='cuda')
dz = torch.randn((B, n), device='cuda')
mw, vw, mb, vb = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(b)
for _ in range(3):
    ops.dense_bwd_filter_adam_tf1(x, dz, w, mw, vw, b, mb, vb, 0.1, 0.9, 1.0, 0.9, 1.0, 1.0)
lib = _lib.load()
lib.a3d_debug_dense_stamps.restype = ctypes.c_int
lib.a3d_debug_dense_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(4096 * 4 * 8, np.uint64)
lib.a3d_debug_dense_stamps(buf.ctypes.data, buf.nbytes)
a = buf.reshape(4096, 4, 8).astype(np.int64)
nblk = min(4096, ((n + 511) // 512) * ((k + 31) // 32))
a = a[:nblk].reshape(-1, 8)
t0 = a[:, 0].min()
names = ['issue m loads', 'bias + acc init', 'contraction', 'lds write + barrier', 'wait for m', 'update + store (drained)']
for i, nm in enumerate(names):
    d = a[:, i + 1] - a[:, i]
    print(f'  {nm:28s} mean {d.mean():8.0f} cyc   p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}')
life = a[:, 5] - a[:, 0]
print(f'  wave life mean {life.mean():.0f} cyc; kernel span {a[:, 5].max() - t0} cyc (100 MHz ticks: x21 for core clocks?)')
print(f'  entries: p10 {np.percentile(a[:, 0] - t0, 10):.0f} p50 {np.percentile(a[:, 0] - t0, 50):.0f} p90 {np.percentile(a[:, 0] - t0, 90):.0f}')
