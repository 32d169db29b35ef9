"""Diagnostic: where an iteration of the LDS-DMA bf16 kernel (igemm_ring.h) spends its cycles, from in-kernel stamps of the
-DA3D_STAMPS build (make -C ann3depth_amd/csrc OBJDIR=build_stamps TARGET=../../tools/ab/liba3d_stamps.so EXTRA=-DA3D_STAMPS).
Not a timing tool: a stamp drains the wave's outstanding LDS reads.
    A3D_LIB=tools/ab/liba3d_stamps.so python tools/stamps_ring.py conv2d_1 fwd"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import _lib, ops  # noqa: E402
from tools.sweep_igemm import LAYERS  # noqa: E402

name, mode = sys.argv[1], sys.argv[2]
B = int(os.environ.get('B', 64))
zero = os.environ.get('ZERO') == '1'
L = [l for l in LAYERS if l[0] == name][0]
_, h, w, c, k, ks, st, pad = L
X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
bf = torch.bfloat16
d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad, precision='bf16')
x = torch.randn((B, h, w, c), device='cuda').to(bf)
wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
wb = wt.to(bf)
bias = torch.zeros(k, device='cuda')
y = torch.empty((B, d.ho, d.wo, k), device='cuda', dtype=bf)
dz = torch.randn((B, d.ho, d.wo, k), device='cuda').to(bf)
if zero:
    x.zero_(), wb.zero_(), dz.zero_()
dx = torch.empty_like(x)
dw = torch.empty_like(wt)
db = torch.empty(k, device='cuda')
fn = {'fwd': lambda: ops.conv2d_fwd(ops.with_storage(d, X | W | Y), x, wb, bias, y, 'relu'),
      'bwd_f': lambda: ops.conv2d_bwd_filter(ops.with_storage(d, X | Y), x, dz, dw, db),
      'bwd_d': lambda: ops.conv2d_bwd_data(ops.with_storage(d, X | W | Y), dz, wb, dx, relu_mask=x)}[mode]
warm = torch.randn((4096, 4096), device='cuda')
for _ in range(100):
    warm = torch.tanh(warm @ warm * 1e-4)
for _ in range(20):
    fn()
lib = _lib.load()
lib.a3d_debug_stamps.restype = ctypes.c_int
lib.a3d_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8 << 17, np.uint64)
grid = lib.a3d_debug_stamps(buf.ctypes.data, buf.nbytes)
raw = buf[:grid * 8 * 16].reshape(grid, 8, 16)
raw = raw[raw[:, :, 5] > 0].reshape(-1, 16)
a = raw.astype(np.float64)
nkt = a[:, 5]
clock = a[:, 3] / a[:, 4] * 100e6
print(f'{name} {mode} B={B}{" zeros" if zero else ""}: grid {grid}, {len(a)} waves, k-tiles/wave {nkt.mean():.1f}')
print(f'  in-kernel clock over the loop: median {np.median(clock) / 1e9:.3f} GHz (min {clock.min() / 1e9:.3f} max {clock.max() / 1e9:.3f})')
print(f'  loop {np.mean(a[:, 3] / nkt):.0f} cycles per k-tile = {np.mean(a[:, 3] / nkt) / np.median(clock) * 1e6:.3f} us'
      f' (MFMA alone: 2 waves x 32 x 32 = 2048 cycles per SIMD and k-tile of a 256 x 256 block)')
for i, n in enumerate(['k-steps 0..2 (24 MFMA slots of this wave)', 'wait for the requests + barrier', 'last k-step (8 MFMA slots, requests, first reads)']):
    per = a[:, i] / nkt
    print(f'  {n:52s} {per.mean():7.0f} cycles per k-tile (min {per.min():.0f} max {per.max():.0f})')
t_entry, t_exit = raw[:, 8].astype(np.int64), raw[:, 9].astype(np.int64)
print(f'  prologue {a[:, 6].mean():.0f} cycles (min {a[:, 6].min():.0f} max {a[:, 6].max():.0f}), loop {a[:, 3].mean():.0f}, '
      f'epilogue {a[:, 7].mean():.0f} (min {a[:, 7].min():.0f} max {a[:, 7].max():.0f})')
print(f'  prologue parts: tables + first barrier {a[:, 10].mean():.0f}, lane constants + offsets + requests of tile 0 {a[:, 11].mean():.0f}, '
      f'fragment addresses + wait for tile 0 + barrier + first reads {(a[:, 6] - a[:, 10] - a[:, 11]).mean():.0f}')
print(f'  kernel span (first entry -> last exit) {t_exit.max() - t_entry.min()} cycles; entries spread over {t_entry.max() - t_entry.min()}')
