"""Config 5 (precision 'bf16s'): what the error against the fp32 oracle actually IS, per tensor.  The
same quantities tests/test_gpu_msdn.py::test_msdn_bf16_storage_at_config5_batch bounds.   python tools/bf16s_errors.py"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from ann3depth_amd import models            # noqa: E402
from oracle import msdn as O                # noqa: E402
from test_gpu_msdn import gpu_activations, rel, synth   # noqa: E402

B = 64
img, dep, keep = synth(B, 6464)
params = O.init_params(3000)
args = [torch.from_numpy(a).cuda() for a in (img, dep, keep)]
table = {}
net = models.MSDNReplica(B, params=params, precision='bf16s')
net.step(*args)
torch.cuda.synchronize()
sl = [0, 1, 62, 63]
a = O.forward(params, img[sl], dep[sl], keep[sl])
table['depth/coarse'] = rel(net.coarse[sl].cpu().numpy(), a['coarse'])
table['depth/fine (coarse phase: fine/first on the bf16 image form)'] = rel(net.fine[sl].cpu().numpy(), a['fine'])
a_gpu = gpu_activations(net)
a_gpu['keep_mask'] = keep
for n, gref in O.backward_coarse(params, a_gpu).items():
    table['grad/' + n] = rel(net.grad(n).cpu().numpy(), gref)
for b2 in (8, 64):
    net2 = models.MSDNReplica(b2, params=params, precision='bf16s', global_step=2000000 // b2)
    a8 = [t[:b2] for t in args]
    net2.step(*a8)
    torch.cuda.synchronize()
    if b2 == 8:
        a2 = O.forward(params, img[:8], dep[:8], keep[:8])
        table['depth/fine (fine phase)'] = rel(net2.fine.cpu().numpy(), a2['fine'])
    a_gpu = gpu_activations(net2)
    for n, gref in O.backward_fine(params, a_gpu).items():
        table[f'grad/{n} (B={b2})'] = rel(net2.grad(n).cpu().numpy(), gref)
    del net2
for k, v in table.items():
    print(f'{k:75s} {v:.3e}')
print(json.dumps(table))
