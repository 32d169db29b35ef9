"""Diagnostic: per-phase cycle shares of one K-tile iteration of the implicit-GEMM kernel, from the in-kernel stamps of
the -DA3D_STAMPS build (tools/ab/liba3d_stamps.so; see csrc/Makefile).  Not a timing tool: the stamps fence overlaps.
    A3D_LIB=tools/ab/liba3d_stamps.so python tools/stamps_layer.py conv2d_1 fwd [cfg [splitk]]"""
import ctypes
import os
os.environ.setdefault('A3D_TUNING', '1')   # the library reads its A3D_FORCE_* switches per launch only then
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_igemm import LAYERS, B  # noqa: E402
from ann3depth_amd import _lib, ops  # noqa: E402

name, mode = sys.argv[1], sys.argv[2]
if len(sys.argv) > 3 and int(sys.argv[3]) >= 0:
    os.environ['A3D_FORCE_CFG'] = sys.argv[3]
    os.environ['A3D_FORCE_SPLITK'] = sys.argv[4] if len(sys.argv) > 4 else '1'
EXTRA = [('gemm', 25, 40, 2400, 256, 1, 1, 'VALID'), ('gemmT', 25, 40, 256, 2400, 1, 1, 'VALID')]      # plain GEMMs as 1x1 convs (tools/gen2_ab.py)
_, h, w, c, k, ks, st, pad = next(l for l in LAYERS + EXTRA if l[0] == name)
d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad)
x = torch.randn((B, h, w, c), device='cuda')
wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
bias = torch.zeros(k, device='cuda')
y = torch.empty((B, d.ho, d.wo, k), device='cuda')
dz = torch.randn_like(y)
dx = torch.empty_like(x)
dw = torch.empty_like(wt)
yp = torch.empty((B, d.ho // 2, d.wo // 2, k), device='cuda')
am = torch.empty((B, d.ho // 2, d.wo // 2, k), dtype=torch.uint8, device='cuda')
fn = {'fwd': lambda: ops.conv2d_fwd(d, x, wt, bias, y, 'relu'),
      'fwdpool': lambda: ops.conv2d_pool_fwd(d, x, wt, bias, yp, 'relu', am),
      'fwdplain': lambda: ops.conv2d_fwd(d, x, wt, None, y, None),
      'bwd_f': lambda: ops.conv2d_bwd_filter(d, x, dz, dw, None),
      'bwd_d': lambda: ops.conv2d_bwd_data(d, dz, wt, dx)}[mode]
for _ in range(3):
    fn()
lib = _lib.load()
lib.a3d_debug_stamps.restype = ctypes.c_int
lib.a3d_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8 << 17, np.uint64)
W = 16
grid = lib.a3d_debug_stamps(buf.ctypes.data, buf.nbytes)
nw = 8
raw = buf[:grid * nw * W].reshape(grid, nw, W)
raw = raw[raw[:, :, 6] > 0].reshape(-1, W)
a = raw.astype(np.float64)
nkt = a[:, 6]
names = ['issue next tile loads (P)', 'chunks 0..2: lds reads + mfma', 'vmcnt wait + ds_write', 'last chunk mfma',
         'barrier wait']
tot = a[:, :5].sum(1)
print(f'{name} {mode}: {len(a)} waves with work, k-tiles/wave {nkt.mean():.1f}, loop cycles/tile {np.mean(a[:,5]/nkt):.0f}')
for i, n in enumerate(names):
    per = a[:, i] / nkt
    print(f'  {n:34s} {per.mean():8.0f} cyc/tile  ({100 * a[:, i].sum() / tot.sum():5.1f} %)   min {per.min():.0f} max {per.max():.0f}')
t_entry, t_end, t_exit = raw[:, 9].astype(np.int64), raw[:, 10].astype(np.int64), raw[:, 11].astype(np.int64)
t0 = t_entry.min()
print(f'  prologue {a[:, 8].mean():.0f} cyc (min {a[:, 8].min():.0f} max {a[:, 8].max():.0f}), loop {a[:, 5].mean():.0f}, '
      f'epilogue {(t_exit - t_end).mean():.0f} (min {(t_exit - t_end).min()} max {(t_exit - t_end).max()})')
print(f'  kernel span (first entry -> last exit) {t_exit.max() - t0} cyc; entries spread over {t_entry.max() - t0} cyc; '
      f'exit times: 10% {np.percentile(t_exit - t0, 10):.0f} 50% {np.percentile(t_exit - t0, 50):.0f} 90% {np.percentile(t_exit - t0, 90):.0f}')
rt = a[:, 12]
ok = rt > 0
if ok.any():
    print(f'  clock held during the loops: {np.mean(a[ok, 5] / rt[ok]) * 0.1:.3f} GHz (loop cycles / 100 MHz ticks; min {np.min(a[ok, 5] / rt[ok]) * 0.1:.3f} max {np.max(a[ok, 5] / rt[ok]) * 0.1:.3f}); '
          f'first loop start -> last loop end {(raw[ok, 14].astype(np.int64).max() - raw[ok, 13].astype(np.int64).min()) * 0.01:.1f} us')
if ok.any():
    rb = (raw[ok, 13].astype(np.int64) - raw[ok, 13].astype(np.int64).min()) * 0.01
    first = rb < 30.0
    print(f'  prologue parts (cycles): setup {a[:, 7].mean():.0f}, address + issue of tiles 0/1 {a[:, 15].mean():.0f}, wait + LDS store + barrier '
          f'{(a[:, 8] - a[:, 7] - a[:, 15]).mean():.0f}; loop starts of the first round ({first.sum()} waves): '
          f'10% {np.percentile(rb[first], 10):.1f} 50% {np.percentile(rb[first], 50):.1f} 90% {np.percentile(rb[first], 90):.1f} max {rb[first].max():.1f} us after the first')
    lb = (t_entry + raw[:, 8].astype(np.int64) - t0)[ok][first]
    print(f'  first-round loop starts, cycles after the first wave entered the kernel: 10% {np.percentile(lb, 10):.0f} 50% {np.percentile(lb, 50):.0f} 90% {np.percentile(lb, 90):.0f}; '
          f'entries of those waves: 50% {np.percentile((t_entry - t0)[ok][first], 50):.0f} 90% {np.percentile((t_entry - t0)[ok][first], 90):.0f}')
