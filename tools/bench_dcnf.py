"""BASELINE config 4: DCNF unary conv stack, batch 16 (768 patches of 100x100x3), one MI355X.
Times the forward (resize -> patches -> 5 conv / 3 pool / 3 dense), the unary backward from a synthetic dz, and the
whole `models.dcnf` train step (+ pairwise part, CRF loss, gradient descent).
    python tools/bench_dcnf.py [batch] > gpurun_out/dcnf.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import models  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(1000)
img = torch.from_numpy((rng.integers(0, 256, (B, 480, 640, 3)) / 255).astype(np.float32)).cuda()
dep = torch.from_numpy(rng.random((B, 55, 74, 1)).astype(np.float32)).cuda()
rep = models.DCNFReplica(B)
net = rep.unary
dz = torch.randn((net.P, 1), device='cuda')


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_fwd = timeit(lambda: net.forward(img))
t_bwd = timeit(lambda: net.backward(dz))
t_step = timeit(lambda: rep.step(img, dep))
t_crf = timeit(lambda: rep.forward_crf(dep))
gflop_patch = 2.672          # forward GFLOP per patch
fwd_tf = gflop_patch * net.P / t_fwd           # GFLOP / ms = TFLOP/s
print(json.dumps({'workload': f'DCNF unary, batch {B} -> {net.P} patches 100x100x3', 'forward_ms': round(t_fwd, 3),
                  'forward_images_per_s': round(B / t_fwd * 1e3, 1), 'forward_tflops': round(fwd_tf, 1),
                  'backward_ms': round(t_bwd, 3), 'dtype': 'f32',
                  'train_step_ms': round(t_step, 3), 'train_step_images_per_s': round(B / t_step * 1e3, 1),
                  'pairwise_and_crf_loss_ms': round(t_crf, 3),
                  'fwd_bwd_images_per_s': round(B / (t_fwd + t_bwd) * 1e3, 1)}))
