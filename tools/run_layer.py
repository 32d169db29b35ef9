"""Runs one MSDN layer/direction a few times (optionally with a forced tile config) — target for rocprofv3 --pmc.
    python tools/run_layer.py conv2d_1 fwd [cfg [splitk [reps]]]"""
import os
os.environ.setdefault('A3D_TUNING', '1')   # the library reads its A3D_FORCE_* switches per launch only then
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_igemm import LAYERS, B  # noqa: E402
from ann3depth_amd import ops  # noqa: E402

name, mode = sys.argv[1], sys.argv[2]
if len(sys.argv) > 3 and int(sys.argv[3]) >= 0:
    os.environ['A3D_FORCE_CFG'] = sys.argv[3]
    os.environ['A3D_FORCE_SPLITK'] = sys.argv[4] if len(sys.argv) > 4 else '1'
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
_, h, w, c, k, ks, st, pad = next(l for l in LAYERS if l[0] == name)
d = ops.conv_desc(B, h, w, c, k, ks, ks, st, pad)
x = torch.randn((B, h, w, c), device='cuda')
wt = torch.randn((ks, ks, c, k), device='cuda') * 0.01
bias = torch.zeros(k, device='cuda')
y = torch.empty((B, d.ho, d.wo, k), device='cuda')
dz = torch.randn_like(y)
dx = torch.empty_like(x)
dw = torch.empty_like(wt)
fn = {'fwd': lambda: ops.conv2d_fwd(d, x, wt, bias, y, 'relu'),
      'bwd_f': lambda: ops.conv2d_bwd_filter(d, x, dz, dw, None),
      'bwd_d': lambda: ops.conv2d_bwd_data(d, dz, wt, dx)}[mode]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print('done', name, mode)
