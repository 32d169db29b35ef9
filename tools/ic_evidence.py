"""HBM or Infinity Cache?  The dominant kernel (bwd-filter of conv2d_1, stream-K with tile-major shares) reads 13-18x its
algorithmic bytes through the fabric (FETCH_SIZE): every XCD streams all of x and dz.  At batch 32 the two tensors are
45 MB and stay in the 256 MiB Infinity Cache; at batch 256 they are 360 MB and cannot (MI355X_MICROARCH.md, Infinity
Cache: scale past L3 before reading FETCH_SIZE as over-fetch evidence).  This script times the launch at a given batch,
tile-major (default) or K-sliced per XCD (A3D_SK_SLICED=1, read once per process); run it under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
for the byte counts (tools/profile_ic.sh runs exactly these passes and writes profiles/<round>_ic_evidence.txt).
    python tools/ic_evidence.py BATCH"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402
from tools.bench_layers import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = ops.conv_desc(B, 27, 37, 96, 256, 5, 5, 1, 'SAME')
x = torch.randn((B, 27, 37, 96), device='cuda')
dz = torch.randn((B, 27, 37, 256), device='cuda')
dw = torch.empty((5, 5, 96, 256), device='cuda')
db = torch.empty(256, device='cuda')
t = timeit(lambda: ops.conv2d_bwd_filter(d, x, dz, dw, db), reps=10)
flops = 2.0 * B * 27 * 37 * 256 * 2400
print(json.dumps({'batch': B, 'k_sliced': os.environ.get('A3D_SK_SLICED', '0') == '1', 'us': round(t, 1),
                  'tflops': round(flops / t / 1e6, 1), 'x_plus_dz_MB': round((x.numel() + dz.numel()) * 4 / 1e6, 1),
                  'algorithmic_MB': round((x.numel() + dz.numel() + dw.numel()) * 4 / 1e6, 1)}))
