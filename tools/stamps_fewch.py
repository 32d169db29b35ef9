"""Diagnostic: where a row of the few-channel filter-gradient kernel (fewch.hip) spends its cycles, from the in-kernel stamps of
the -DA3D_STAMPS build (tools/ab/liba3d_stamps.so; see csrc/Makefile).  Not a timing tool: the stamps fence overlaps.
    A3D_LIB=tools/ab/liba3d_stamps.so python tools/stamps_fewch.py [conv2d_0|fine/first] [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import _lib, ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'conv2d_0'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
k, ks, st, ld = {'conv2d_0': (96, 11, 4, 96), 'fine/first': (63, 9, 2, 64)}[which]
d = ops.conv_desc(B, 228, 304, 3, k, ks, ks, st, 'VALID')
x = torch.randn((B, 228, 304, 3), device='cuda')
ph, pw = d.ho // 2, d.wo // 2
pooled = torch.randn((B, ph, pw, ld), device='cuda')
dpool = torch.randn((B, ph, pw, ld), device='cuda')
arg = torch.randint(0, 4, (B, ph, pw, k), device='cuda', dtype=torch.uint8)
dw = torch.empty((ks, ks, 3, k), device='cuda')
db = torch.empty((k,), device='cuda')
for _ in range(3):
    ops.conv2d_bwd_filter_pooled(d, x, dpool, pooled, arg, dw, db)
torch.cuda.synchronize()
lib = _lib.load()
lib.a3d_debug_fewch_stamps.restype = ctypes.c_int
lib.a3d_debug_fewch_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(1024 * 4 * 8, np.uint64)
assert lib.a3d_debug_fewch_stamps(buf.ctypes.data, buf.nbytes) == 0
a = buf.reshape(1024, 4, 8)
a = a[a[:, :, 4] > 0].reshape(-1, 8).astype(np.float64)
rows = a[:, 4]
print(f'{which} B={B}: {len(a)} waves, rows per wave {rows.min():.0f}..{rows.max():.0f}')
for i, n in enumerate(['barrier 1 (the other waves\' MFMAs)', 'commit (LDS writes, pool-gradient decode)', 'barrier 2', 'reads + MFMAs (+ next row\'s loads)']):
    per = a[:, i] / rows
    print(f'  {n:44s} {per.mean():8.0f} cyc/row   min {per.min():.0f} max {per.max():.0f}')
print(f'  entry -> loop end {a[:, 6].mean():.0f} cyc (max {a[:, 6].max():.0f}); in the loop {a[:, :4].sum(1).mean():.0f}; epilogue (slab stores, until they have landed) {a[:, 5].mean():.0f} (max {a[:, 5].max():.0f})')
print(f'  of the gradient row: waiting for its loads {np.mean(a[:, 7] / rows):.0f} cyc/row')
